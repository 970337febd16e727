// Ray generation, stratified sampling, hierarchical (inverse-CDF) sampling and
// the NDC transform for gfx950.  All kernels here are bandwidth-bound; this file
// is compiled with -ffp-contract=off so that every multiply/add rounds exactly
// like the separate PyTorch ops of the reference (bit-exact depths).
#include "common.h"

namespace nerf {

struct Pose {
    float m[12];
};

__device__ __forceinline__ void pixel_ray(const float* __restrict__ c2w, float fx, float fy, float cx, float cy,
                                          int W, int pix, float o[3], float d[3]) {
    // src/Camera.py:62-70: dirs = ((i-cx)/fx, -(j-cy)/fy, -1); rays_d = normalize(R * dirs)
    const int row = pix / W, col = pix - row * W;
    const float dx = ((float)col - cx) / fx;
    const float dy = -((float)row - cy) / fy;
    const float dz = -1.0f;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = (dx * c2w[4 * k + 0] + dy * c2w[4 * k + 1]) + dz * c2w[4 * k + 2];
    const float n = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = r[k] / n;
        o[k] = c2w[4 * k + 3];
    }
}

__global__ void raygen_kernel(const float* __restrict__ c2w, float fx, float fy, float cx, float cy, int W, int HW,
                              const int32_t* __restrict__ pix, int64_t R, float* __restrict__ rays_o,
                              float* __restrict__ rays_d) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float o[3], d[3];
    const int px = pix ? pix[r] : (int)r;
    pixel_ray(c2w, fx, fy, cx, cy, W, px, o, d);
    const bool ok = (unsigned)px < (unsigned)HW;   // a pixel id outside the image yields a NaN ray
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[3 * r + k] = ok ? o[k] : __builtin_nanf("");
        rays_d[3 * r + k] = ok ? d[k] : __builtin_nanf("");
    }
}

// Rays of a batch drawn across many cameras: ray r looks through pixel pix[r] of camera cam_id[r].
// cams: [n][16] = the 12 floats of the 3x4 pose, then fx, fy, cx, cy.  With images ([n][H*W][3]) the
// pixel's colour is gathered alongside (the training target), so a batch costs one pass.
// A camera or pixel id out of range reads nothing and yields NaN rays and colours.
__global__ void raygen_multi_kernel(const float* __restrict__ cams, int n_cams, int W, int64_t HW, const int32_t* __restrict__ cam_id,
                                    const int32_t* __restrict__ pix, int64_t R, const float* __restrict__ images,
                                    float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ rgb) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int c = cam_id[r], px = pix[r];
    if ((unsigned)c >= (unsigned)n_cams || (unsigned)px >= (unsigned)HW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rays_o[3 * r + k] = __builtin_nanf("");
            rays_d[3 * r + k] = __builtin_nanf("");
            if (images) rgb[3 * r + k] = __builtin_nanf("");
        }
        return;
    }
    const float* cam = cams + 16 * (int64_t)c;
    float pose[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pose[k] = cam[k];
    float o[3], d[3];
    pixel_ray(pose, cam[12], cam[13], cam[14], cam[15], W, px, o, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[3 * r + k] = o[k];
        rays_d[3 * r + k] = d[k];
    }
    if (images) {
        const float* src = images + 3 * ((int64_t)c * HW + px);
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[3 * r + k] = src[k];
    }
}

// Depth of sample n: src/Camera.py:133-137 (midpoint strata, lower + (upper-lower)*u).
__device__ __forceinline__ float stratified_depth(const float* __restrict__ z_lin, int N, int n, const float* u,
                                                  int64_t un) {
    const float zc = z_lin[n];
    if (!u) return zc;
    const float lower = (n == 0) ? zc : 0.5f * (zc + z_lin[n - 1]);
    const float upper = (n == N - 1) ? zc : 0.5f * (z_lin[n + 1] + zc);
    return lower + (upper - lower) * u[un];
}

// Four consecutive (ray, sample) pairs per thread, 1024 per workgroup.  A thread derives the
// ray(s) of its samples itself (RAYGEN: from the pixel, forty instructions; else a load) -- at
// most two when N >= 4: one division finds the first, a comparison the switch -- forms its four
// depths and twelve point coordinates in registers and parks the coordinates in LDS as 48
// contiguous bytes, so that the points then leave as fully coalesced 16-byte stores (flat float
// f of the workgroup's span is coordinate f%3 of sample f/3).  One barrier per workgroup.
constexpr int SMP_THREADS = 256, SMP_PER_THREAD = 4, SMP_BLOCK = SMP_THREADS * SMP_PER_THREAD;
template <bool RAYGEN>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const float* __restrict__ c2w, float fx, float fy, float cx,
                                                             float cy, int W, const int32_t* __restrict__ pix,
                                                             const float* __restrict__ in_o, const float* __restrict__ in_d,
                                                             const float* __restrict__ z_lin, const float* __restrict__ u,
                                                             int64_t R, int N, float* __restrict__ rays_o,
                                                             float* __restrict__ rays_d, float* __restrict__ z_vals,
                                                             float* __restrict__ pts) {
    extern __shared__ __attribute__((aligned(16))) float s_pts[];   // the workgroup's 1024 points
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * SMP_BLOCK, total = R * N;
    const int here = (int)min((int64_t)SMP_BLOCK, total - t0);         // pairs in this workgroup
    const int first = SMP_PER_THREAD * tid;
    const bool whole = first + SMP_PER_THREAD <= here;   // all four pairs exist: 16-byte accesses (t0 is a multiple of 1024)
    const int64_t r_first = t0 / N;                      // the same for the whole workgroup
    const int n_first = (int)(t0 - r_first * N);
    const int k_first = (n_first + first) / N;           // 32-bit division: n_first < N, first < 1024
    int64_t ray = r_first + k_first;                     // ray and sample index of the thread's first pair
    int n = (n_first + first) - k_first * N;
    float zz[4] = {0.f, 0.f, 0.f, 0.f}, pc[12];
    if (first < here) {
        float4 uu = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u && whole) uu = *reinterpret_cast<const float4*>(u + t0 + first);
        const float ua[4] = {uu.x, uu.y, uu.z, uu.w};
        float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool exists = first + k < here;
            if (exists && (k == 0 || n == 0)) {   // a new ray: derive it from its pixel (cheaper than a trip through LDS), or load it
                if (RAYGEN) {
                    pixel_ray(c2w, fx, fy, cx, cy, W, pix ? pix[ray] : (int)ray, o, d);
                    if (n == 0) {   // the thread that holds a ray's first sample publishes the ray
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if (rays_o) rays_o[3 * ray + c] = o[c];
                            if (rays_d) rays_d[3 * ray + c] = d[c];
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        o[c] = in_o[3 * ray + c];
                        d[c] = in_d[3 * ray + c];
                    }
                }
            }
            if (exists) zz[k] = whole ? stratified_depth(z_lin, N, n, u ? ua : nullptr, k)
                                      : stratified_depth(z_lin, N, n, u, t0 + first + k);
#pragma unroll
            for (int c = 0; c < 3; ++c) pc[3 * k + c] = o[c] + d[c] * zz[k];   // src/Camera.py:140
            if (++n == N) {   // the next pair starts the next ray
                n = 0;
                ++ray;
            }
        }
        if (z_vals) {
            if (whole) *reinterpret_cast<float4*>(z_vals + t0 + first) = make_float4(zz[0], zz[1], zz[2], zz[3]);
            else
                for (int k = 0; k < 4 && first + k < here; ++k) z_vals[t0 + first + k] = zz[k];
        }
    }
    if (!pts) return;
    if (first < here) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            *reinterpret_cast<float4*>(s_pts + 12 * tid + 4 * j) = make_float4(pc[4 * j], pc[4 * j + 1], pc[4 * j + 2], pc[4 * j + 3]);
    }
    __syncthreads();
    float* out = pts + t0 * 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = 4 * (j * SMP_THREADS + tid);
        if (f + 4 <= 3 * here) {
            *reinterpret_cast<float4*>(out + f) = *reinterpret_cast<const float4*>(s_pts + f);
        } else {
            for (int k = 0; k < 4 && f + k < 3 * here; ++k) out[f + k] = s_pts[f + k];
        }
    }
}

// LDS of a sample_kernel workgroup: its 1024 points.
inline size_t sample_lds_bytes(int) { return 3 * SMP_BLOCK * sizeof(float); }

// ---------------------------------------------------------------------------
// Hierarchical sampling: one wavefront per ray, 4 rays per workgroup; the waves
// never meet (each works in its own LDS region, and LDS operations of one wave
// complete in order).  Per ray: log-step scan of the 62 interior weights (in registers by wavefront
// shuffles when they fit one per lane), a binary
// search per fine sample, a bitonic sort of the fine depths in registers (EPL per
// lane), and a merge with the already sorted coarse depths by rank: a coarse depth
// moves up by the number of fine depths below it, a fine depth by the number of
// coarse depths not above it.  Only values leave, so the order of equal depths is
// immaterial.
// ---------------------------------------------------------------------------
constexpr int PDF_MAX_NC = 256;
constexpr int PDF_MAX_ALL = 1024;
constexpr int PDF_WAVES = 4;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Value of lane ^ J.
template <int J>
__device__ __forceinline__ float xor_lane(float v) {
    if constexpr (J == 1) return dpp_move<0xB1>(v, v);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_move<0x4E>(v, v);   // quad_perm [2,3,0,1]
    else if constexpr (J == 8) return dpp_move<0x128>(v, v);  // row_ror:8 (rows of 16 lanes)
    else if constexpr (J == 4) {   // banks 0 and 2 of each row read four lanes up, banks 1 and 3 four lanes down
        const int up = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_ROW_SHL + 4, 0xf, 0x5, false);
        return __int_as_float(__builtin_amdgcn_update_dpp(up, __float_as_int(v), DPP_ROW_SHR + 4, 0xf, 0xA, false));
    } else if constexpr (J == 16) {   // the swap exchanges the odd rows of its first operand with the even rows of its second
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
    } else {                          // 32: the upper half of the first operand with the lower half of the second
        static_assert(J == 32, "partner distances within a wavefront");
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
    }
}

// The lanes that keep the smaller value in the exchange at distance j < 64 of the stage forming sorted runs of `size`
// (register k of the lane: element k*64 + lane; the last stage, size == total, sorts everything upwards).
constexpr unsigned long long bitonic_min_lanes(int size, int j, int k, int total) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const bool asc = ((k * 64 + l) & size) == 0 || size == total, lower = (l & j) == 0;
        if (lower == asc) m |= 1ull << l;
    }
    return m;
}

// Ascending bitonic sort of 64*EPL values; element e = k*64 + lane lives in v[k] of the lane.
template <int EPL>
__device__ __forceinline__ void wave_bitonic_sort(float (&v)[EPL], int lane) {
    static_for<1, 7 + (EPL == 1 ? 0 : EPL == 2 ? 1 : EPL == 4 ? 2 : EPL == 8 ? 3 : 4)>([&](auto cs) {
        constexpr int SIZE = 1 << decltype(cs)::value;   // length of the sorted runs being formed
        static_for<0, decltype(cs)::value>([&](auto cj) {
            constexpr int J = SIZE >> (1 + decltype(cj)::value);   // partner distance
            static_for<0, EPL>([&](auto ck) {
                constexpr int k = decltype(ck)::value;
                if constexpr (J < 64) {
                    // One compare and one select per exchange: a lane keeps its own value where "mine is smaller" agrees
                    // with "this position takes the smaller one".  Which positions do is a constant of the step, so the
                    // two lane masks are combined on the scalar unit (asm: the compiler would rebuild the constant mask
                    // from the lane index, two vector instructions per step).
                    const float o = xor_lane<J>(v[k]);
                    constexpr unsigned long long TAKES_MIN = bitonic_min_lanes(SIZE, J, k, 64 * EPL);
                    unsigned long long mine_smaller;
                    asm("v_cmp_lt_f32 %0, %1, %2" : "=s"(mine_smaller) : "v"(v[k]), "v"(o));
                    const unsigned long long keep = ~(mine_smaller ^ TAKES_MIN);
                    // (the wait states a DPP read of a freshly written register needs: the compiler inserts them only
                    // behind instructions it knows, and the next exchange may read this register through DPP)
                    asm("v_cndmask_b32 %0, %1, %2, %3\n\ts_nop 1" : "=v"(v[k]) : "v"(o), "v"(v[k]), "s"(keep));
                } else {
                    constexpr int KJ = J / 64;
                    if constexpr ((k & KJ) == 0) {   // handle the pair (k, k + KJ) once: every lane the same direction
                        constexpr bool asc = ((k * 64) & SIZE) == 0 || SIZE == 64 * EPL;
                        const float lo = fminf(v[k], v[k ^ KJ]), hi = fmaxf(v[k], v[k ^ KJ]);
                        v[k] = asc ? lo : hi;
                        v[k ^ KJ] = asc ? hi : lo;
                    }
                }
            });
        });
    });
}

struct __attribute__((packed, aligned(4))) Float3 {   // one 12-byte store
    float x, y, z;
};

// Number of entries of the non-decreasing a[0..n) that are <= v (LE) or < v: a branch-free binary search whose steps
// are the same for every lane (top = the largest power of two <= n).
template <bool LE>
__device__ __forceinline__ int count_below(const float* a, int n, int top, float v) {
    int lo = 0;
    for (int step = top; step >= 1; step >>= 1) {
        const int mid = lo + step;
        const float t = a[min(mid, n) - 1];
        if (mid <= n && (LE ? t <= v : t < v)) lo = mid;
    }
    return lo;
}
__device__ __forceinline__ int pow2_floor(int n) { return 1 << (31 - __builtin_clz(n)); }
// The same count for a length known at compile time, 2^k or 2^k - 1: the classic halving search over 2^k - 1 entries
// needs no bounds checks (k steps of read, compare, add), and the entry a power of two has on top counts only when all
// the others did.
template <bool LE, int N>
__device__ __forceinline__ int count_below_fixed(const float* a, float v) {
    constexpr bool POW2 = (N & (N - 1)) == 0;
    constexpr int M = POW2 ? N - 1 : N;   // 2^k - 1
    static_assert(((M + 1) & M) == 0, "a length of 2^k or 2^k - 1");
    int lo = 0;
#pragma unroll
    for (int step = (M + 1) / 2; step >= 1; step >>= 1) {
        const float t = a[lo + step - 1];
        if (LE ? t <= v : t < v) lo += step;
    }
    if constexpr (POW2) {
        const float t = a[N - 1];
        if (lo == M && (LE ? t <= v : t < v)) lo = N;
    }
    return lo;
}


// NC_MAX: the coarse depths a wave's LDS arrays hold (128 or PDF_MAX_NC).  The kernel lives on its LDS round trips
// (scan, three rounds of binary searches): with arrays sized for 256 + 1024 depths four workgroups fit a CU, sized for
// the usual 64 + 128 eight do, the most its waves allow.
// NC_FIX, NF_FIX: the shape when it is known at compile time (64 coarse + 128 fine, the reference configuration: every
// loop bound and search depth is then a constant), 0 for a shape taken from the arguments.
template <int EPL, int NC_MAX, int NC_FIX = 0, int NF_FIX = 0>
__global__ __launch_bounds__(PDF_WAVES * 64, NC_FIX ? 8 : (EPL <= 2 && NC_MAX <= 128) ? 6 : 1) void pdf_sample_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ z,
    const float* __restrict__ weights, const float* __restrict__ u, int64_t R, int Nc_arg, int Nf_arg,
    float* __restrict__ z_fine, int32_t* __restrict__ idx_out, float* __restrict__ z_all, float* __restrict__ pts) {
    const int Nc = NC_FIX ? NC_FIX : Nc_arg, Nf = NF_FIX ? NF_FIX : Nf_arg;
    __shared__ float s_scan[PDF_WAVES][2][NC_MAX];
    __shared__ float s_bins[PDF_WAVES][NC_MAX];
    __shared__ float s_zc[PDF_WAVES][NC_MAX];
    __shared__ float s_fine[PDF_WAVES][64 * EPL];
    __shared__ float s_sorted[PDF_WAVES][64 * EPL + NC_MAX];
    // the wave index as a scalar: every per-ray address below is then formed on the scalar unit
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * PDF_WAVES + wv;
    if (ray >= R) return;   // whole wave
    const float* zr = z + ray * Nc;
    const float* wr = weights + ray * Nc;
    const int nb = Nc - 1;  // bins and cdf entries
    const int nw = Nc - 2;  // interior weights

    // every global load of the wave goes out now: issued where they are used, the uniforms and the ray would each add
    // a round trip to memory to a chain that already waits for the depths and weights
    float uu[EPL], ray_o[3], ray_d[3];
#pragma unroll
    for (int k = 0; k < EPL; ++k) uu[k] = k * 64 + lane < Nf ? u[ray * Nf + k * 64 + lane] : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ray_o[k] = pts ? rays_o[3 * ray + k] : 0.0f;
        ray_d[k] = pts ? rays_d[3 * ray + k] : 0.0f;
    }
    for (int k = lane; k < Nc; k += 64) s_zc[wv][k] = zr[k];
    float* cdf;
    if (nb <= 64) {
        // One weight and one cdf entry per lane (nb = nw + 1 entries: with nw = 64 the last entry would have no lane): the log-step inclusive scan runs in registers, each step one wavefront shuffle
        // (lane k adds the value of lane k - d).  The association order is part of the contract: the CPU
        // oracle performs the same additions.
        float v = lane < nw ? wr[lane + 1] + 1e-5f : 0.0f;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const float below = __shfl_up(v, dd, 64);
            if (lane >= dd) v = v + below;
        }
        const float total = lane_value(v, nw - 1);
        const float before = __shfl_up(v, 1, 64);   // c[k-1]
        cdf = s_scan[wv][0];
        if (lane < nb) cdf[lane] = lane == 0 ? 0.0f : before / total;   // cdf[0] = 0, cdf[k+1] = c[k] / total
        wave_lds_sync();
        for (int k = lane; k < nb; k += 64) s_bins[wv][k] = 0.5f * (s_zc[wv][k + 1] + s_zc[wv][k]);
    } else {
        for (int k = lane; k < nw; k += 64) s_scan[wv][0][k] = wr[k + 1] + 1e-5f;
        wave_lds_sync();
        for (int k = lane; k < nb; k += 64) s_bins[wv][k] = 0.5f * (s_zc[wv][k + 1] + s_zc[wv][k]);
        // The same scan for more than 64 weights, through two LDS arrays.
        int cur = 0;
        for (int dd = 1; dd < nw; dd <<= 1) {
            for (int k = lane; k < nw; k += 64) {
                const float v = s_scan[wv][cur][k];
                s_scan[wv][cur ^ 1][k] = (k >= dd) ? v + s_scan[wv][cur][k - dd] : v;
            }
            wave_lds_sync();
            cur ^= 1;
        }
        const float total = s_scan[wv][cur][nw - 1];
        cdf = s_scan[wv][cur ^ 1];
        for (int k = lane; k < nb; k += 64) cdf[k] = (k == 0) ? 0.0f : s_scan[wv][cur][k - 1] / total;
    }
    wave_lds_sync();

    const int top_nb = pow2_floor(nb), top_nc = pow2_floor(Nc), top_nf = pow2_floor(Nf);
    float zf[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        const int f = k * 64 + lane;
        zf[k] = __builtin_inff();   // padding sorts to the end
        if (f < Nf) {
            const float uk = uu[k];
            int lo;   // first entry with cdf > the uniform
            if constexpr (NC_FIX != 0) lo = count_below_fixed<true, NC_FIX - 1>(cdf, uk);
            else lo = count_below<true>(cdf, nb, top_nb, uk);
            const int below = max(lo - 1, 0), above = min(lo, nb - 1);
            const float c_lo = cdf[below], c_hi = cdf[above];
            const float b_lo = s_bins[wv][below], b_hi = s_bins[wv][above];
            float denom = c_hi - c_lo;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uk - c_lo) / denom;
            zf[k] = b_lo + t * (b_hi - b_lo);
            if (z_fine) z_fine[ray * Nf + f] = zf[k];
            if (idx_out) idx_out[ray * Nf + f] = lo;
        }
    }
    // The depths come out in the order of their uniforms (the inverse cdf is monotone): a caller that passes sorted
    // uniforms -- an evenly spaced or stratified set, as deterministic renders do -- needs no sort, a third of this
    // kernel's instructions.  Checked on the depths themselves, so nothing is assumed: element e against e + 1.
    bool in_order = true;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        float next = dpp_move<DPP_WAVE_SHL1>(0.0f, zf[k]);   // lane i reads lane i + 1 (lane 63 is set below)
        if (lane == 63) next = k + 1 < EPL ? lane_value(zf[k + 1 < EPL ? k + 1 : k], 0) : __builtin_inff();
        in_order = in_order && zf[k] <= next;   // false for a NaN: the sort decides
    }
    if (!__all(in_order)) wave_bitonic_sort<EPL>(zf, lane);
#pragma unroll
    for (int k = 0; k < EPL; ++k) s_fine[wv][k * 64 + lane] = zf[k];
    wave_lds_sync();
    const int na = Nc + Nf;
    for (int i = lane; i < Nc; i += 64) {   // coarse depth i: fine depths strictly below it come first
        const float v = s_zc[wv][i];
        int below_it;
        if constexpr (NF_FIX != 0) below_it = count_below_fixed<false, NF_FIX>(s_fine[wv], v);
        else below_it = count_below<false>(s_fine[wv], Nf, top_nf, v);
        s_sorted[wv][i + below_it] = v;
    }
#pragma unroll
    for (int k = 0; k < EPL; ++k) {         // fine depth e: coarse depths up to and including it come first
        const int e = k * 64 + lane;
        if (e < Nf) {
            int up_to;
            if constexpr (NC_FIX != 0) up_to = count_below_fixed<true, NC_FIX>(s_zc[wv], zf[k]);
            else up_to = count_below<true>(s_zc[wv], Nc, top_nc, zf[k]);
            s_sorted[wv][e + up_to] = zf[k];
        }
    }
    wave_lds_sync();
    if (z_all)
        for (int i = lane; i < na; i += 64) z_all[ray * na + i] = s_sorted[wv][i];
    if (pts) {
        const float (&o)[3] = ray_o, (&d)[3] = ray_d;
        // a lane per sample, 12 contiguous bytes each: 64 lanes store 768 contiguous bytes per instruction
        float* out = pts + ray * na * 3;
        for (int smp = lane; smp < na; smp += 64) {
            const float t = s_sorted[wv][smp];
            *reinterpret_cast<Float3*>(out + 3 * smp) = Float3{o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t};
        }
    }
}

__global__ void ndc_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int64_t R, float ax,
                           float ay, float near_, float* __restrict__ out_o, float* __restrict__ out_d) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = rays_o[3 * r + k];
        d[k] = rays_d[3 * r + k];
    }
    const float t = -(near_ + o[2]) / d[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = o[k] + t * d[k];
    const float oxz = o[0] / o[2], oyz = o[1] / o[2];
    out_o[3 * r + 0] = ax * oxz;
    out_o[3 * r + 1] = ay * oyz;
    out_o[3 * r + 2] = 1.0f + 2.0f * near_ / o[2];
    out_d[3 * r + 0] = ax * (d[0] / d[2] - oxz);
    out_d[3 * r + 1] = ay * (d[1] / d[2] - oyz);
    out_d[3 * r + 2] = -2.0f * near_ / o[2];
}

// view_dirs = rays_d / ||rays_d|| (what the reference's render helpers feed the model, src/train_nerf_llff.py:42,
// src/train_nerf_ngp.py:92): one launch instead of a reduction and a division.
__global__ void normalize_kernel(const float* __restrict__ d, int64_t R, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float x = d[3 * r], y = d[3 * r + 1], z = d[3 * r + 2];
    const float n = sqrtf((x * x + y * y) + z * z);
    out[3 * r] = x / n;
    out[3 * r + 1] = y / n;
    out[3 * r + 2] = z / n;
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_normalize_dirs(const float* rays_d, int64_t R, float* out, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_d && out && R > 0, "nerf_normalize_dirs: null pointer or R=%lld", (long long)R);
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, rays_d, R, out);
    return check_launch("nerf_normalize_dirs");
}

extern "C" int nerf_raygen(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                           const int32_t* pix, int64_t R, float* rays_o, float* rays_d, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(c2w && rays_o && rays_d, "nerf_raygen: null pointer");
    NERF_REQUIRE(H > 0 && W > 0 && R >= 0, "nerf_raygen: bad size H=%d W=%d R=%lld", H, W, (long long)R);
    NERF_REQUIRE(pix || R == (int64_t)H * W, "nerf_raygen: R=%lld must equal H*W when pix is NULL", (long long)R);
    NERF_REQUIRE((int64_t)H * W < (1ll << 31), "nerf_raygen: image too large");
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, c2w, fx, fy,
                       cx, cy, W, H * W, pix, R, rays_o, rays_d);
    return check_launch("nerf_raygen");
}

extern "C" int nerf_raygen_multi(const float* cams, int n_cams, int H, int W, const int32_t* cam_id, const int32_t* pix,
                                 int64_t R, const float* images, float* rays_o, float* rays_d, float* rgb, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(cams && cam_id && pix && rays_o && rays_d, "nerf_raygen_multi: null pointer");
    NERF_REQUIRE(n_cams > 0 && H > 0 && W > 0 && R >= 0, "nerf_raygen_multi: bad size n=%d H=%d W=%d R=%lld", n_cams, H, W,
                 (long long)R);
    NERF_REQUIRE((int64_t)H * W < (1ll << 31), "nerf_raygen_multi: image too large");
    NERF_REQUIRE(!images || rgb, "nerf_raygen_multi: images given without an rgb output");
    hipLaunchKernelGGL(raygen_multi_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, cams, n_cams, W,
                       (int64_t)H * W, cam_id, pix, R, images, rays_o, rays_d, rgb);
    return check_launch("nerf_raygen_multi");
}

extern "C" int nerf_stratified_sample(const float* rays_o, const float* rays_d, const float* z_lin, const float* u,
                                      int64_t R, int N, float* z_vals, float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && z_lin, "nerf_stratified_sample: null pointer");
    NERF_REQUIRE(R >= 0 && N >= 1, "nerf_stratified_sample: bad size R=%lld N=%d", (long long)R, N);
    NERF_REQUIRE(((size_t)u | (size_t)z_vals | (size_t)pts) % 16 == 0, "nerf_stratified_sample: u, z_vals and pts must be 16-byte aligned");
    hipLaunchKernelGGL(sample_kernel<false>, dim3((unsigned)ceil_div(R * N, SMP_BLOCK)), dim3(SMP_THREADS), sample_lds_bytes(N), (hipStream_t)stream,
                       nullptr, 0.f, 0.f, 0.f, 0.f, 0, nullptr, rays_o, rays_d, z_lin, u, R, N, nullptr, nullptr,
                       z_vals, pts);
    return check_launch("nerf_stratified_sample");
}

extern "C" int nerf_raygen_sample(const float* c2w, float fx, float fy, float cx, float cy, int H, int W,
                                  const int32_t* pix, int64_t R, const float* z_lin, const float* u, int N,
                                  float* rays_o, float* rays_d, float* z_vals, float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(c2w && z_lin, "nerf_raygen_sample: null pointer");
    NERF_REQUIRE(H > 0 && W > 0 && R >= 0 && N >= 1, "nerf_raygen_sample: bad size");
    NERF_REQUIRE(pix || R == (int64_t)H * W, "nerf_raygen_sample: R must equal H*W when pix is NULL");
    NERF_REQUIRE((int64_t)H * W < (1ll << 31), "nerf_raygen_sample: image too large");
    NERF_REQUIRE(((size_t)u | (size_t)z_vals | (size_t)pts) % 16 == 0, "nerf_raygen_sample: u, z_vals and pts must be 16-byte aligned");
    hipLaunchKernelGGL(sample_kernel<true>, dim3((unsigned)ceil_div(R * N, SMP_BLOCK)), dim3(SMP_THREADS), sample_lds_bytes(N), (hipStream_t)stream, c2w,
                       fx, fy, cx, cy, W, pix, nullptr, nullptr, z_lin, u, R, N, rays_o, rays_d, z_vals, pts);
    return check_launch("nerf_raygen_sample");
}

extern "C" int nerf_pdf_sample(const float* rays_o, const float* rays_d, const float* z, const float* weights,
                               const float* u, int64_t R, int Nc, int Nf, float* z_fine, int32_t* idx, float* z_all,
                               float* pts, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && z && weights && u, "nerf_pdf_sample: null pointer");
    NERF_REQUIRE(Nc >= 3 && Nc <= PDF_MAX_NC, "nerf_pdf_sample: Nc=%d outside [3,%d]", Nc, PDF_MAX_NC);
    NERF_REQUIRE(Nf >= 1 && Nc + Nf <= PDF_MAX_ALL, "nerf_pdf_sample: Nc+Nf=%d exceeds %d", Nc + Nf, PDF_MAX_ALL);
    NERF_REQUIRE(R >= 0, "nerf_pdf_sample: R=%lld", (long long)R);
    const dim3 grid((unsigned)ceil_div(R, PDF_WAVES)), block(PDF_WAVES * 64);
#define PDF_LAUNCH(EPL)                                                                                                      \
    do {                                                                                                                     \
        if (Nc <= 128)                                                                                                       \
            hipLaunchKernelGGL((pdf_sample_kernel<EPL, 128>), grid, block, 0, (hipStream_t)stream, rays_o, rays_d, z, weights, \
                               u, R, Nc, Nf, z_fine, idx, z_all, pts);                                                       \
        else                                                                                                                 \
            hipLaunchKernelGGL((pdf_sample_kernel<EPL, PDF_MAX_NC>), grid, block, 0, (hipStream_t)stream, rays_o, rays_d, z,   \
                               weights, u, R, Nc, Nf, z_fine, idx, z_all, pts);                                              \
    } while (0)
    if (Nc == 64 && Nf == 128)
        hipLaunchKernelGGL((pdf_sample_kernel<2, 128, 64, 128>), grid, block, 0, (hipStream_t)stream, rays_o, rays_d, z, weights, u,
                           R, Nc, Nf, z_fine, idx, z_all, pts);
    else if (Nf <= 64) PDF_LAUNCH(1);
    else if (Nf <= 128) PDF_LAUNCH(2);
    else if (Nf <= 256) PDF_LAUNCH(4);
    else if (Nf <= 512) PDF_LAUNCH(8);
    else PDF_LAUNCH(16);
#undef PDF_LAUNCH
    return check_launch("nerf_pdf_sample");
}

extern "C" int nerf_ndc_rays(const float* rays_o, const float* rays_d, int64_t R, int H, int W, float focal,
                             float near_, float* out_o, float* out_d, void* stream) {
    if (R == 0) return NERF_OK;
    NERF_REQUIRE(rays_o && rays_d && out_o && out_d, "nerf_ndc_rays: null pointer");
    NERF_REQUIRE(R >= 0 && H > 0 && W > 0, "nerf_ndc_rays: bad size");
    const float ax = (float)(-(double)focal / (0.5 * W)), ay = (float)(-(double)focal / (0.5 * H));
    hipLaunchKernelGGL(ndc_kernel, dim3((unsigned)ceil_div(R, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       R, ax, ay, near_, out_o, out_d);
    return check_launch("nerf_ndc_rays");
}
