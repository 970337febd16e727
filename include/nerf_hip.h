/*
 * nerf_hip.h -- C ABI of libnerf_hip.so: the NeRF render/train hot path as
 * hand-written gfx950 (MI355X) HIP kernels.
 *
 * The reference (skowae/NeRF) is pure Python/PyTorch and has no FFI of its
 * own; each entry point below names the reference function whose body it
 * replaces (paths relative to the reference checkout).  Conventions:
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - all tensors are dense row-major float32 unless stated;
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library
 *     never allocates and keeps no state between calls;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *     default stream) and the call returns without synchronising;
 *   - the return value is 0 on success or a negative NERF_E* code, and
 *     nerf_last_error() gives the thread-local message of the last failure.
 * Thread safety: calls are re-entrant; they may come from any host thread
 * (PyTorch runs backward on its own thread).
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 4
#define NERF_OK 0
#define NERF_EINVAL (-22) /* bad argument (shape, null pointer, unsupported size) */
#define NERF_ELAUNCH (-5) /* the HIP runtime rejected a launch                    */

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* ---------------------------------------------------------------- rays ---- */

/* Camera.get_rays, src/Camera.py:49-72.  c2w: 12 floats, rows of the 3x4 pose.
 * Pixel id p = row*W + col.  pix == NULL renders all H*W pixels in order (R must
 * be H*W); otherwise pix holds R int32 pixel ids.  rays_o, rays_d: [R,3].  A pixel
 * id outside [0, H*W) yields a NaN ray.
 * The older free function get_rays(H, W, focal, c2w) of src/utils/get_rays.py:115-147
 * (+z forward, [W,H,3] output) is this call with the third pose column negated,
 * fx = fy = focal, cx = W/2, cy = H/2 and pix = the transposed pixel order. */
int nerf_raygen(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                const int32_t* pix, int64_t R, float* rays_o, float* rays_d, void* stream);

/* A ray batch drawn across the images of a dataset, which the reference forms one image at a
 * time (Camera.get_rays per DataLoader item, src/train_nerf_skow.py:132-146, src/train_nerf_llff.py:106-111).
 * cams: [n_cams][16] = 12 floats of the 3x4 pose, fx, fy, cx, cy; ray r looks through pixel pix[r]
 * (row*W + col, all images H x W) of camera cam_id[r].  images: [n_cams][H*W][3] or NULL; when given,
 * rgb [R,3] receives the pixels' colours.  A camera id outside [0, n_cams) or a pixel id
 * outside [0, H*W) reads nothing and yields NaN rays (and colours). */
int nerf_raygen_multi(const float* cams, int n_cams, int H, int W, const int32_t* cam_id,
                      const int32_t* pix, int64_t R, const float* images, float* rays_o,
                      float* rays_d, float* rgb, void* stream);

/* Camera.sample_points_along_rays, src/Camera.py:127-141.  z_lin: the N base
 * depths (torch.linspace(near, far, N), computed by the caller so that the
 * values are the reference's bit for bit).  u: [R,N] uniforms in [0,1) that the
 * reference would draw with torch.rand_like, or NULL for perturb=False.
 * Outputs z_vals [R,N] and pts [R,N,3] (either may be NULL).  u, z_vals and pts
 * must be 16-byte aligned (they are accessed 16 bytes at a time). */
int nerf_stratified_sample(const float* rays_o, const float* rays_d, const float* z_lin,
                           const float* u, int64_t R, int N, float* z_vals, float* pts,
                           void* stream);

/* Fused form of the two calls above: rays are generated and sampled without a
 * round trip through memory.  Any of rays_o/rays_d/z_vals/pts may be NULL; the
 * alignment rule of nerf_stratified_sample applies. */
int nerf_raygen_sample(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                       const int32_t* pix, int64_t R, const float* z_lin, const float* u, int N,
                       float* rays_o, float* rays_d, float* z_vals, float* pts, void* stream);

/* Hierarchical inverse-CDF sampling (NeRF 2020 section 5.2).  The reference has
 * no such stage; the semantics are those of oracle/nerf_oracle.py:sample_pdf.
 * z [R,Nc], weights [R,Nc], u [R,Nf] -> z_fine [R,Nf] (may be NULL), idx int32 [R,Nf] (may be
 * NULL), z_all [R,Nc+Nf] sorted, pts [R,Nc+Nf,3] (may be NULL).
 * Limits: 3 <= Nc <= 256, 1 <= Nf, Nc+Nf <= 1024. */
int nerf_pdf_sample(const float* rays_o, const float* rays_d, const float* z, const float* weights,
                    const float* u, int64_t R, int Nc, int Nf, float* z_fine, int32_t* idx,
                    float* z_all, float* pts, void* stream);

/* Forward-facing NDC ray transform (NeRF 2020 appendix C); not in the reference,
 * semantics of oracle/nerf_oracle.py:ndc_rays.  In/out [R,3]; in-place allowed. */
int nerf_ndc_rays(const float* rays_o, const float* rays_d, int64_t R, int H, int W, float focal,
                  float near_, float* out_o, float* out_d, void* stream);

/* view_dirs [R,3] = rays_d / ||rays_d||, the directions the reference's render helpers hand to the model
 * (src/train_nerf_llff.py:42, src/train_nerf_skow.py:77, src/train_nerf_ngp.py:92). */
int nerf_normalize_dirs(const float* rays_d, int64_t R, float* out, void* stream);

/* ------------------------------------------------------------ encoders ---- */

/* PositionalEncoding.forward, src/utils/positional_encoding.py:21-36 with
 * log-scale bands 2^k and include_input: x [P,D] -> out [P, D*(2L+1)]. */
int nerf_encode_pe(const float* x, int64_t P, int D, int L, float* out, void* stream);

/* HashGridEncoder.forward, src/utils/hash_grid_encoder.py:42-107.
 * tables: host array of `levels` device pointers, level l holding
 * sizes[l] rows of 2 floats; res, sizes: host int arrays.  out [P, 2*levels].
 * idx_dump (may be NULL): int32 [levels,8,P] hashed row of every corner in the
 * reference's order c000,c001,...,c111; w_dump (may be NULL): [levels,P,3]. */
int nerf_hashgrid_fwd(const float* x, int64_t P, const float* const* tables, const int* res,
                      const int* sizes, int levels, float* out, int32_t* idx_dump, float* w_dump,
                      void* stream);

/* Backward of the above w.r.t. the tables (the reference gets it from the
 * dense nn.Embedding gradient).  d_out [P, 2*levels]; d_tables: host array of
 * device pointers, ACCUMULATED into with float atomics (zero them first).
 * workspace: device scratch of nerf_hashgrid_bwd_workspace_bytes(P, sizes, levels) bytes (about
 * 1.4 KB per point; 16-byte aligned, 128 for whole-line writes), with which the updates of the
 * tables too large for LDS are first sorted by table segment and summed in LDS (4.5 times faster
 * at 4 M points); NULL (or a size of 0, which that function returns for small P) makes every
 * update a float atomic on the table.  Same sums either way, up to the order of the additions and,
 * for the high-x corner of a corner pair, a few units in the last place (its update is formed
 * as the low-x corner's times wx / (1 - wx)). */
size_t nerf_hashgrid_bwd_workspace_bytes(int64_t P, const int* sizes, int levels);
int nerf_hashgrid_bwd(const float* x, int64_t P, const float* d_out, const int* res,
                      const int* sizes, int levels, float* const* d_tables, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ----------------------------------------------------------- composite ---- */

/* Camera.volume_render, src/Camera.py:143-183.  rgb [R,N,3], sigma [R,N],
 * z [R,N] -> out_rgb [R,3], weights [R,N]; depth [R] = sum w*z (may be NULL; the
 * reference has no depth output).  2 <= N <= 1024 (for N = 1 the reference's
 * tail-distance concat yields empty tensors; that case is rejected here).
 * tail_dist: the distance appended after the last sample, 1e5 for Camera.volume_render
 * (src/Camera.py:154) and 1e10 for the older volume_render of src/utils/get_rays.py:198-236. */
#define NERF_TAIL_CAMERA 1e5f
#define NERF_TAIL_LEGACY 1e10f
int nerf_composite_fwd(const float* rgb, const float* sigma, const float* z, int64_t R, int N,
                       int white_bkgd, float tail_dist, float* out_rgb, float* weights, float* depth,
                       void* stream);

/* Backward of volume_render for a loss that reads out_rgb, weights and depth
 * (the reference's NGP loss uses both outputs, src/train_nerf_ngp.py:221-222).
 * d_out [R,3]; d_weights [R,N] and d_depth [R] may be NULL (zero).
 * Outputs d_rgb [R,N,3], d_sigma [R,N]. */
int nerf_composite_bwd(const float* rgb, const float* sigma, const float* z, const float* d_out,
                       const float* d_weights, const float* d_depth, int64_t R, int N,
                       int white_bkgd, float tail_dist, float* d_rgb, float* d_sigma, void* stream);

/* The training loss of the reference's loops and its gradient, one pass each:
 *   loss = mean((rgb - target)^2) [+ mean((rgb_fine - target)^2)] [+ reg * mean_r |1 - sum_n weights[r][n]|]
 * -- the MSE of the rendered colours (src/train_nerf_llff.py:116, src/train_nerf_skow.py:168), of both passes
 * of a hierarchical render when rgb_fine is given, plus the weight regulariser of the hash-grid loop
 * (src/train_nerf_ngp.py:221-222: reg = 0.05) when weights [R,N] is given.  rgb, rgb_fine, target [R,3].
 * workspace: nerf_render_loss_workspace_floats(R) floats; loss: one device float.  The sums are formed in a
 * fixed order.  bwd: d_loss = one device float (the gradient of the loss itself); d_rgb, d_rgb_fine [R,3],
 * d_weights [R,N] are WRITTEN (any of them may be NULL). */
size_t nerf_render_loss_workspace_floats(int64_t R);
int nerf_render_loss_fwd(const float* rgb, const float* rgb_fine, const float* target,
                         const float* weights, int64_t R, int N, float reg, float* workspace,
                         float* loss, void* stream);
int nerf_render_loss_bwd(const float* rgb, const float* rgb_fine, const float* target,
                         const float* weights, int64_t R, int N, float reg, const float* d_loss,
                         float* d_rgb, float* d_rgb_fine, float* d_weights, void* stream);

/* Occupancy-grid masking of per-sample densities, the step between the model and the compositor in
 * render_image (src/train_nerf_ngp.py:113-117) and render_from_model
 * (src/utils/analyze_capture_results.py:283-287): out[p] = values[p] where grid[cell(p)] == 1, else 0, with
 * cell(p) = x + 32 y + 1024 z of the truncated ((pts + 1) / 2) / (1 / grid_res) (xyz_to_idx keeps its default
 * res = 32, :230-231).  pts [P,3], values/out [P] (in place allowed), grid: grid_cells bytes.  A cell index
 * outside the grid (an IndexError in the reference) is clamped into it.  The same call applied to d_sigma
 * is the backward: masked samples receive no gradient. */
int nerf_occupancy_mask(const float* pts, const float* values, int64_t P, const uint8_t* grid,
                        int64_t grid_cells, int grid_res, float* out, void* stream);

/* Skipping the samples nerf_occupancy_mask would zero ("eventually skipping", the step after
 * src/train_nerf_ngp.py:113-117 and src/utils/analyze_capture_results.py:283-287, which run the model on
 * every sample and zero the densities of the empty cells afterwards): a sample with sigma = 0 has
 * weight 0, adds nothing to the colour and receives no gradient, so evaluating the model on the
 * occupied samples only renders the same image.
 * nerf_occupancy_compact: stream compaction of pts [P,3] in their own order.  kept [P] (first
 * *count entries valid): indices of the samples in occupied cells; slot [P]: a sample's position in
 * that list, -1 if dropped; pts_kept [P,3] / dirs_kept [P,3] (optional): the kept samples'
 * coordinates and their rays' view directions (dirs has one row per dirs_div consecutive samples);
 * count: one device int32.  workspace: nerf_occupancy_compact_workspace_ints(P) int32.  Cell lookup
 * exactly as nerf_occupancy_mask.  P < 2^31.
 * nerf_expand_kept: rgb [P,3], sigma [P] = the kept samples' model outputs at their places, zeros
 * at the dropped ones.  nerf_gather_kept: its backward (d_rgb / d_sigma may be null = zeros). */
size_t nerf_occupancy_compact_workspace_ints(int64_t P);
int nerf_occupancy_compact(const float* pts, int64_t P, const uint8_t* grid, int64_t grid_cells,
                           int grid_res, const float* dirs, int dirs_div, int32_t* workspace,
                           int32_t* kept, int32_t* slot, float* pts_kept, float* dirs_kept,
                           int32_t* count, void* stream);
int nerf_expand_kept(const float* rgb_kept, const float* sigma_kept, const int32_t* slot, int64_t P,
                     float* rgb, float* sigma, void* stream);
int nerf_gather_kept(const float* d_rgb, const float* d_sigma, const int32_t* kept, int64_t K,
                     float* d_rgb_kept, float* d_sigma_kept, void* stream);
/* The same for a caller that leaves the count on the device (no read-back: the step can then be captured
 * into a HIP graph): sample p hands its gradient to position slot[p] of d_rgb_kept [P,3] / d_sigma_kept [P]
 * (entries behind the count stay untouched). */
int nerf_gather_kept_by_slot(const float* d_rgb, const float* d_sigma, const int32_t* slot, int64_t P,
                             float* d_rgb_kept, float* d_sigma_kept, void* stream);

/* ---------------------------------------------------------- fused MLPs ---- */

/* Vanilla 8x256 NeRF MLP: NeRF.forward (src/models/NeRF.py:67-109) and
 * FastMLPNeRF.forward (src/models/FastMLPNeRF.py:47-70), which differ only in
 * the column order of the layer-4 weight (skip_h_first = 0: [x_enc | h],
 * 1: [h | x_enc]).
 *
 * Parameter block `params`: host array of 24 device pointers in the order
 *   W0,b0, ... ,W7,b7, Wsigma,bsigma, Wfeat,bfeat, Wrgb0,brgb0, Wrgb2,brgb2
 * with the nn.Linear shapes (out,in): (256,63) (256,256)x3 (256,319)
 * (256,256)x3 (1,256) (256,256) (128,283) (3,128).
 *
 * Numerics: fp32 in and out with fp32 accuracy (the forward pass agrees with a float64
 * evaluation to 1e-7).  Products run on the f16 matrix pipe as three MFMAs over two-piece
 * splits of the fp32 operands; the kernels keep the pieces in range themselves with exact
 * power-of-two factors per point and per saved matrix, so no loss scaling or clamping is
 * asked of the caller, whatever the magnitude of activations and gradients.
 */
#define NERF_VANILLA_NPARAM 24

/* Floats of packed-weight scratch needed by the forward / backward kernels. */
size_t nerf_mlp_vanilla_pack_floats(void);
/* Re-order the weights into the per-lane MFMA operand layout. */
int nerf_mlp_vanilla_pack(const float* const* params, int skip_h_first, float* packed,
                          void* stream);
/* Floats per point of the forward's activation workspace (kept for backward)
 * and of the backward's own scratch (per-layer pre-activation gradients). */
size_t nerf_mlp_vanilla_ws_floats_per_point(void);
size_t nerf_mlp_vanilla_ws2_floats_per_point(void);
/* x [P,3], dirs [ceil(P/dirs_div),3] -> rgb [P,3], sigma [P].  Point p uses
 * direction row p/dirs_div: dirs_div = 1 is the reference's per-point view_dirs,
 * dirs_div = samples-per-ray lets one row per ray serve all its samples (what
 * render_image expands by hand, src/train_nerf_llff.py:42-47).  ws: workspace of
 * ceil32(P)*ws_floats_per_point floats, or NULL for inference (nothing saved). */
int nerf_mlp_vanilla_fwd(const float* x, const float* dirs, int dirs_div, int64_t P,
                         const float* packed, float* rgb, float* sigma, float* ws, void* stream);
/* d_rgb [P,3], d_sigma [P] -> parameter gradients, ACCUMULATED (float atomics)
 * into `grads` (host array of 24 device pointers, same order/shapes as params;
 * zero them first).  ws is the forward's workspace; ws2: ceil32(P) *
 * ws2_floats_per_point floats of scratch.  phases: bit 0 runs the backward-data
 * chain (fills ws2), bit 1 the weight-gradient kernels (read ws and ws2); pass 3. */
int nerf_mlp_vanilla_bwd(const float* d_rgb, const float* d_sigma, int64_t P, const float* packed,
                         const float* rgb, const float* sigma, float* ws, float* ws2,
                         int skip_h_first, float* const* grads, int phases, void* stream);
/* Which ReLUs the forward pass found open (F.relu at src/models/NeRF.py:87 and inside rgb_layers, :58-63),
 * read back from its workspace for parity checks: out [P][NERF_VANILLA_MASK_ROW] bytes, 0 or 1;
 * out[p][256 l + n] = neuron n of trunk layer l (0..7), out[p][2048 + n] = neuron n of the colour hidden
 * layer (0..127). */
#define NERF_VANILLA_MASK_ROW 2176
int nerf_mlp_vanilla_relu_masks(const float* ws, int64_t P, uint8_t* out, void* stream);

/* Hash-grid NeRF, FastNGPNeRF.forward (src/models/FastNGPNeRF.py:79-110), fused:
 * 16-level hash encoding (src/utils/hash_grid_encoder.py:42-107) + density MLP +
 * colour MLP with the bias-free skip projection.  The kernel is specialised to
 * the reference's defaults: 16 levels x 2 features, hidden width 64.
 *
 * `params`: host array of 9 device pointers
 *   fc_density.0.weight (64,32), .bias (64), fc_density.2.weight (16,64), .bias (16),
 *   fc_color.0.weight (64,42), .bias (64), fc_color.2.weight (3,64), .bias (3),
 *   skip_proj.weight (64,15).
 * tables / res / sizes: as nerf_hashgrid_fwd, exactly 16 levels.
 */
#define NERF_NGP_NPARAM 9
size_t nerf_mlp_ngp_pack_floats(void);
size_t nerf_mlp_ngp_ws_floats_per_point(void);
size_t nerf_mlp_ngp_ws2_floats_per_point(void);
int nerf_mlp_ngp_pack(const float* const* params, float* packed, void* stream);
/* x [P,3]; dirs [ceil(P/dirs_div),3] un-normalised view directions (the kernel
 * normalises them like the model does) or NULL for "no directions" (colour input
 * zeros, src/models/FastNGPNeRF.py:100-103) -> rgb [P,3], sigma [P].
 * ws: ceil32(P)*ws_floats_per_point floats, or NULL for inference.
 * enc: scratch of ceil32(P)*32 floats that receives the encoding level by level (two launches: the
 * gather, every level on one XCD, then the MLPs). */
int nerf_mlp_ngp_fwd(const float* x, const float* dirs, int dirs_div, int64_t P,
                     const float* const* tables, const int* res, const int* sizes,
                     const float* packed, float* rgb, float* sigma, float* ws, float* enc,
                     void* stream);
/* Gradients of the 9 MLP parameters (`grads`, same order/shapes) and of the 16
 * tables (`d_tables`), all ACCUMULATED (added to what the buffers hold; the tables' by float atomics,
 * the parameters' by plain additions of per-workgroup sums in a fixed order): zero them first.
 * ws2: ceil32(P)*ws2_floats_per_point floats of scratch.  scatter_ws / scatter_ws_bytes: the
 * workspace of nerf_hashgrid_bwd (nerf_hashgrid_bwd_workspace_bytes(P, sizes, 16)), or NULL / 0.
 * phases: bit 0 runs the backward kernel -- the backward-data chain, which also forms the
 * weight-gradient sums of fc_density.0/.2, fc_color.0 and skip_proj per workgroup -- and the
 * table-gradient scatter: d_tables is complete when these launches are.  Bit 1 adds those sums to
 * `grads` and forms the gradient of the colour head fc_color.2 (needs the ws2 of bit 0, untouched in
 * between); 3 = both.  A data-parallel caller issues 1, starts the all-reduce of the table gradients
 * (41.7 MB) and issues 2.  (With fewer than 128 points bit 0 adds to `grads` directly.) */
int nerf_mlp_ngp_bwd(const float* x, const float* d_rgb, const float* d_sigma, int64_t P,
                     const int* res, const int* sizes, const float* packed, const float* rgb,
                     const float* sigma, float* ws, float* ws2, float* const* d_tables,
                     float* const* grads, void* scatter_ws, size_t scatter_ws_bytes, int phases,
                     void* stream);
/* The two calls above on the first *count points of arrays sized for P_max, count being one int32 in
 * device memory (nerf_occupancy_compact's): the model on the samples that survive the occupancy grid
 * (src/train_nerf_ngp.py:113-117 evaluates all of them) without the host ever learning how many they are.
 * Every buffer, workspace and size query is that of P_max points; grids are launched for P_max and the
 * workgroups behind the count leave at once; outputs behind the count are not written, inputs there not
 * read.  count > P_max is taken as P_max, count <= 0 as nothing to do. */
int nerf_mlp_ngp_fwd_counted(const float* x, const float* dirs, int dirs_div, int64_t P_max,
                             const int32_t* count, const float* const* tables, const int* res,
                             const int* sizes, const float* packed, float* rgb, float* sigma,
                             float* ws, float* enc, void* stream);
int nerf_mlp_ngp_bwd_counted(const float* x, const float* d_rgb, const float* d_sigma, int64_t P_max,
                             const int32_t* count, const int* res, const int* sizes,
                             const float* packed, const float* rgb, const float* sigma, float* ws,
                             float* ws2, float* const* d_tables, float* const* grads,
                             void* scatter_ws, size_t scatter_ws_bytes, int phases, void* stream);

/* One Adam step over `count` tensors in one launch: torch.optim.Adam (no amsgrad), the optimiser
 * of the reference's training loops (src/train_nerf_llff.py:82 and src/train_nerf_skow.py:47: lr 5e-4,
 * weight_decay 1e-6; src/train_nerf_ngp.py:55: lr 1e-2, eps 1e-15).  weight_decay is torch's L2 form
 * (grad + weight_decay * param enters the moments).
 * Host arrays of device pointers params/grads/exp_avg/exp_avg_sq [count] and element
 * counts numel [count]; step = 1 for the first step.  Updates params, exp_avg and
 * exp_avg_sq in place. */
int nerf_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, int64_t step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, void* stream);
/* The same step with the step count AND the learning rate in device memory, for a training step
 * that is captured into a HIP graph and replayed: a count or a rate passed by value would be frozen
 * into the graph, and the reference's loops change the rate every epoch (ExponentialLR
 * src/train_nerf.py:87,204; StepLR src/train_nerf_llff.py:83,183; ReduceLROnPlateau and
 * CosineAnnealingLR src/train_nerf_ngp.py:57-62,373-379).  The mode itself has no counterpart in the
 * reference, whose loops step torch.optim.Adam eagerly (torch's own name for it is
 * Adam(capturable=True)).  step_state: 4 device floats, [0] = steps taken so far (0 before the first
 * step; integers are exact in a float up to 2^24 steps), [3] = the learning rate, written by
 * nerf_adam_set_lr.  Every call adds 1 to [0] (a one-thread launch in front of the update, which
 * also leaves sqrt(1 - beta2^step) in [1] and -lr / (1 - beta1^step) in [2]) and applies the update
 * for that step.  Neither the host nor the call reads anything back. */
int nerf_adam_step_capturable(int count, float* const* params, const float* const* grads,
                              float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                              float* step_state, float beta1, float beta2, float eps,
                              float weight_decay, void* stream);
/* step_state[3] = lr, by a one-thread launch on `stream` (what an LR scheduler's step amounts to
 * for a captured optimiser: optimizer.param_groups[k]["lr"] = ..., src/train_nerf.py:204).  Call it
 * before the first step and whenever the rate changes, in front of a graph replay on the replay's
 * stream -- not inside a capture, where the value would become part of the graph. */
int nerf_adam_set_lr(float* step_state, float lr, void* stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) with the 2-norm (src/train_nerf.py:165):
 * total = the norm of all gradients together, every gradient is multiplied by
 * min(1, max_norm / (total + 1e-6)) in place.  grads/numel: host arrays [count]; total_norm: one
 * device float that receives total (may be NULL).  workspace: device scratch of
 * nerf_clip_grad_norm_workspace_floats(count, numel) floats; nothing is read back to the host. */
size_t nerf_clip_grad_norm_workspace_floats(int count, const int64_t* numel);
int nerf_clip_grad_norm(int count, float* const* grads, const int64_t* numel, float max_norm,
                        float* workspace, float* total_norm, void* stream);

/* nerf_clip_grad_norm without the workspace argument.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving or
 * training loops, harnesses that forbid allocation).  The scratch is a 256 KiB array built into the
 * library's device image (each GPU the library runs on holds its own copy until the process ends);
 * the call allocates and frees nothing.  Limit: the gradients together may span at most 65,536 chunks
 * of 4,096 elements (each tensor rounds up to whole chunks; 268,435,456 elements at most) -- more is
 * rejected with NERF_EINVAL.  Every call on a GPU uses that GPU's one array: calls that overlap on
 * the same GPU (two streams, two threads) race; calls on one stream, or on different GPUs, do not.
 * Same results as nerf_clip_grad_norm, bit for bit. */
int nerf_clip_grad_norm_static(int count, float* const* grads, const int64_t* numel, float max_norm,
                               float* total_norm, void* stream);

/* The gradient exchange of ray-sharded data-parallel training (no counterpart in the reference,
 * which is single-process: rays are independent, every rank renders its own batch with replicated
 * parameters and the only exchange of a step is the sum of the parameter gradients; the same
 * partitioning nerf_amd/dist.py implements over torch.distributed).
 * nerf_allreduce_sum: buf[0..n) <- sum over the ranks of rccl_comm (an ncclComm_t), in place, float32,
 * enqueued on `stream`; one call per gradient bucket (flat buffer) per step, issued as soon as the
 * kernels filling the bucket are enqueued on that stream.  The communicator may come from the caller's
 * own RCCL calls or from nerf_comm_init_rank: rank 0 fills a NERF_COMM_ID_BYTES buffer with
 * nerf_comm_unique_id, hands it to the other ranks by any means, and every rank (one process per
 * GPU, its device current) calls nerf_comm_init_rank(id, world, rank, &comm); nerf_comm_destroy
 * ends it.  librccl is resolved at first use -- the copy already loaded into the process (PyTorch
 * ships one) or else librccl.so.1 -- so the library has no link-time dependency on it; a
 * communicator must be used with the RCCL copy that created it, which holds when both sides go
 * through these entry points or through the process's one loaded librccl.
 * nerf_rccl_version: RCCL's version code (e.g. 22606), -1 if RCCL is unavailable. */
#define NERF_COMM_ID_BYTES 128
int nerf_rccl_version(void);
int nerf_comm_unique_id(void* id128);
int nerf_comm_init_rank(const void* id128, int world, int rank, void** comm);
int nerf_comm_destroy(void* comm);
int nerf_allreduce_sum(float* buf, size_t n, void* rccl_comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */
