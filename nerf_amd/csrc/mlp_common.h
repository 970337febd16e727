// Building blocks of the fused MLP kernels (gfx950; exact-f32 and split-f16 MFMA).
//
// Data layout idea.  v_mfma_f32_32x32x2_f32 computes D[32x32] += A[32x2] B[2x32]:
//   A: lane l holds A[i = l&31][k = l>>5]      B: lane l holds B[k = l>>5][j = l&31]
//   D: lane l, register r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// A wavefront owns 32 points (the MFMA column j) and keeps a whole activation
// vector per point in registers: tile t, register r of lane (p, h) is neuron
// 32t + rowmap(r, h) of point p.  Register r of such a tile is then exactly the B
// operand of one k-step of the next layer (its two k values are the rows held by
// the two lane halves), so activations flow from one layer's accumulators into
// the next layer's MFMAs without touching LDS.  The weights are re-ordered once
// per step ("packed") so that the A operand of every MFMA is one coalesced
// 256-byte read: pack[kt][r][lane][ot] = W[32*ot + (lane&31)][32*kt + rowmap(r, lane>>5)].
#pragma once
#include "common.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// acc[ot] += W_tile(ot, kt) * in[kt] for all tiles on the exact-f32 MFMA, weights straight from memory
// (the small layers of the hash-grid model).  wp: packed [KT][16][64][OT].
template <int KT, int OT>
__device__ __forceinline__ void dense(const float* __restrict__ wp, const f32x16 (&in)[KT], f32x16 (&acc)[OT],
                                      int lane) {
    static_assert(OT % 4 == 0 || OT == 1 || OT == 2, "OT must be 1, 2 or a multiple of 4");
    typedef float wvec __attribute__((ext_vector_type(OT >= 4 ? 4 : OT)));  // one load: 16, 8 or 4 bytes per lane
    constexpr int W = OT >= 4 ? 4 : OT;
    constexpr int V = OT / W;
    constexpr int S = KT * 16;       // k-steps; one step = OT MFMAs sharing one B register
    constexpr int D = S < 4 ? S : 4;  // steps of weight prefetch kept in flight (L2 latency >> one step)
    const wvec* __restrict__ w4 = reinterpret_cast<const wvec*>(wp) + lane * V;
    wvec ring[D][V];
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int v = 0; v < V; ++v) ring[s][v] = w4[(size_t)(s * 64) * V + v];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float a[OT];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const wvec t = ring[s % D][v];
#pragma unroll
            for (int j = 0; j < W; ++j) a[W * v + j] = t[j];
        }
        if (s + D < S) {
#pragma unroll
            for (int v = 0; v < V; ++v) ring[s % D][v] = w4[(size_t)((s + D) * 64) * V + v];
        }
        // Pin the order "issue the prefetch, then this step's MFMAs": left alone the
        // scheduler sinks every load to just before its use and exposes its latency.
        __builtin_amdgcn_sched_barrier(0);
        const float b = in[s / 16][s % 16];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ot], b, acc[ot], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------ shared weight stream ----
// Weights that arrive through vector loads cost the matrix pipe time: every KiB
// returned to a SIMD (into registers or, by a direct-to-LDS load, into LDS) holds
// its MFMAs back by ~22 cycles, which is 7 of 71 cycles per MFMA when each of the
// four waves of a workgroup loads every weight itself.  LDS reads are free.  So the
// workgroup copies the weights into LDS once, each wave a quarter, and every wave
// reads its operands from there.
//
// The packed weights of consecutive dense_h3() calls form one stream of 16-KiB chunks through a ring of
// chunk buffers.  In the middle of chunk g every wave waits for its own copies of chunk g+1, the workgroup
// meets at a barrier (so chunk g+1 is complete, and nobody reads chunk g-1 any more), and the copies of
// chunk g+RING-1 go out into the buffer chunk g-1 occupied.  The barrier sits between two MFMA groups with the
// next operands already requested, so the pipe only sees the skew between the waves.  All four waves must make
// the same calls.  (Details with dense_h3 below.)
//
// Every LDS access here is asm: the compiler makes each LDS access it knows of
// wait for all direct-to-LDS copies in flight, and orders loads by its own model.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
}
template <int OFF_BYTES>
__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF_BYTES));
    return r;
}
template <int OFF_BYTES>
__device__ __forceinline__ void lds_write4(unsigned addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF_BYTES) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int TILE_LD = 36;                                // row pitch of the 32 x 32 transposition buffer
constexpr int TILE_LDS_FLOATS = 32 * TILE_LD;

__device__ __forceinline__ int wrap3(int x) { return x >= 3 ? x - 3 : x; }

struct WeightStream {
    float* lds;   // the chunk buffers shared by the workgroup
    int g3;       // ring buffer of the chunk the next dense layer starts with
    int wave, lane;
    const float* next = nullptr;   // the next chunk to copy, and where the packed layers end
    const float* end = nullptr;
};
// A wave's 32 x 36 float transposition buffer, for writing register tiles (32 points x
// 32 neurons) to a row-major array as whole 128-byte row segments: ds_write_b128 in the
// register layout, ds_read_b128 + one 16-byte store per lane in the row layout (lane L:
// row L/8 of each group of 8 rows, floats 4(L%8)..+3).  Stored straight from the register
// layout a store instruction touches 32 different 128-byte lines, 32 bytes each.
struct TileBuf {
    unsigned wr, rd;   // this lane's LDS byte addresses in the two layouts
    int row, col;      // this lane's row within a group of 8 and first column, row layout
};
__device__ __forceinline__ TileBuf tile_buf(float* lds_wave, int lane) {
    return TileBuf{lds_addr(lds_wave + (lane & 31) * TILE_LD + 4 * (lane >> 5)),
                   lds_addr(lds_wave + (lane >> 3) * TILE_LD + 4 * (lane & 7)), lane >> 3, 4 * (lane & 7)};
}
// Rows to write underneath a dense_h3() call: its input tiles, to rows0[32][ld].
struct TileSave {
    TileBuf tb;
    float* rows0;   // row 0 of the tile's 32 rows, already offset by this lane's (row, col)
    int ld;
    uint4* relu_bits_out = nullptr;   // PRE_RELU: one bit per input neuron (1 = positive) goes here, in relu_bits() order
    uint4 bits = {0u, 0u, 0u, 0u};    // PRE_MASK: the inputs to keep, in the same order
    float out_scale = 1.0f;           // PRE_OUTSCALE: the rows are written multiplied by this
};
__device__ __forceinline__ TileSave tile_save(const TileBuf& tb, float* tile_rows0, int ld = 256) {
    return TileSave{tb, tile_rows0 + (size_t)tb.row * ld + tb.col, ld};
}

// Write NT register tiles to rows0[32][ld] now (not overlapped with anything).
template <int NT>
__device__ __forceinline__ void save_tiles(const TileBuf& tb, float* __restrict__ rows0, int ld, const f32x16 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // The tile may be the destination of MFMAs issued just before: a 32 x 32 MFMA delivers its result up to 18
        // cycles after it issues, the compiler spaces only the readers it knows, and the readers here are asm.  (Seen
        // as a saved matrix that held partial sums while the registers, read later, were right.)
        f32x16 v16 = a[t];
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(v16));
        static_for<0, 4>([&](auto cq) {
            constexpr int q = decltype(cq)::value;
            lds_write4<32 * q>(tb.wr, f32x4{v16[4 * q], v16[4 * q + 1], v16[4 * q + 2], v16[4 * q + 3]});
        });
        f32x4 v[4];
        static_for<0, 4>([&](auto ck) { v[ck] = lds_read4<8 * decltype(ck)::value * TILE_LD * 4>(tb.rd); });
        lgkm_wait();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("" : "+v"(v[k]));
            __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(rows0 + (size_t)(8 * k + tb.row) * ld + 32 * t + tb.col));
        }
    }
}

__device__ __forceinline__ void ws_copy_piece(const WeightStream& ws, const float* chunk, float* buf, int q) {
    const int piece = 4 * q + ws.wave;   // the four waves take pieces 4q..4q+3 together
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(chunk + piece * 256 + ws.lane * 4),
                                     (__attribute__((address_space(3))) void*)(buf + piece * 256), 16, 0, 0);
}

// YOUNGER: vector memory operations this wave has issued after its last copy (they complete in
// issue order, so the wait may leave that many in flight; waiting for a store just issued costs
// its whole trip to memory).
template <int YOUNGER = 0>
__device__ __forceinline__ void ws_ready() {
    vm_wait<YOUNGER>();
    __builtin_amdgcn_s_barrier();
}

// ------------------------------------------------------- f16 x 3 dense ----
// The same contraction at five times the rate of the exact-f32 MFMA and fp32 accuracy (measured: max error /
// sum|terms| 0.5-1.5e-7 at K = 256, an fp32 fma chain 1.3-3.7e-7): every fp32 operand is two f16 numbers,
// v = h + l with h = f16(v) and l = f16(v - h), which together hold 22-24 bits of v as long as neither piece
// leaves the f16 range (normal numbers 6e-5 .. 65504; subnormal inputs pass through the MFMA unflushed, with
// an absolute error of 2^-25).  A product is the three f16 MFMAs that matter, a_h b_h + a_h b_l + a_l b_h (a_l b_l
// is below 2^-22), accumulated in fp32 by v_mfma_f32_32x32x16_f16.  The residuals are kept in range by powers of
// two: the weights store l' = 2^11 (w - h), the activations supply h'' = 2^-11 h next to h, so that
// (a_l') (b_h'') = a_l b_h; and every point's activation vector is scaled by 2^e, chosen so that its largest
// entry lies in [2^13, 2^14), which the epilogue of the layer removes again (PointScale).  Entries smaller than
// 2^-17 of the largest lose relative, not absolute, accuracy: error / largest entry stays below 2^-38.
// The weights are split once per optimiser step by the pack kernel; the activations on the fly, in the shadow of
// the previous k-step's MFMAs.
//
// A k-step is K = 16: the B operand of lane (point p, half h) is eight of its registers of the input tile
// (r = 8u..8u+7 for k-step 2t+u), i.e. neurons 32t + rowmap(r, h); the packed weights follow the same order:
// h3[k-step][out tile][piece h, l'][lane][8 f16], one k-step = 2 KiB per out tile.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int H3_CHUNK_FLOATS = 8 * 2 * 256;              // a chunk: a k-step of 8 out tiles, or two of 4 (16 KiB)
constexpr int h3_layer_floats(int KT, int OT) { return 2 * KT * OT * 2 * 256; }

// The three conversions below are volatile asm: left to the scheduler they drift between an asm LDS read and the
// wait that completes it, where the extra register pressure makes the compiler copy registers whose data has
// not arrived yet.
// {f16(a), f16(b)} rounded to nearest even, a in the low half
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x - (low or high half of hp as f32): exact in fp32 when that half is f16(x)
template <int HALF>
__device__ __forceinline__ float sub_f16(float x, unsigned hp) {
    float r;
    if constexpr (HALF == 0) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    else asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    return r;
}
// both f16 halves times 2^-11 (exact unless the result is subnormal)
__device__ __forceinline__ unsigned pk_f16_down11(unsigned hp) {
    unsigned r;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(hp), "v"(0x10001000u));
    return r;
}
template <int OFF_BYTES>
__device__ __forceinline__ f16x8 lds_read_h8(unsigned addr) {
    f16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF_BYTES));
    return r;
}

// The power of two that brings a point's activation vector into the f16 pieces' range, and its inverse.
struct PointScale {
    float s, inv;
};
// Lanes l and l ^ 32 exchange a value without LDS (a cross-lane LDS operation known to the compiler would wait
// for every weight copy in flight): after the swap a holds lane l's value in the low half and lane l-32's in the
// high half, b the reverse, so a (op) b combines the pair in both lanes.
__device__ __forceinline__ void swap_halves(float& a, float& b) {
    // the builtin, not asm: the instruction needs wait states after a VALU write of its operands, which the compiler
    // only inserts for instructions it knows
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// max_bits: the bit pattern of the largest |entry| (a non-negative float).
__device__ __forceinline__ PointScale scale_for_max(unsigned max_bits) {
    const int e = (int)(max_bits >> 23);                     // biased exponent; 255 for inf / nan
    int se = 267 - e;                                        // largest entry -> [2^13, 2^14)
    se = se > 253 ? 253 : se;
    if (e == 0) se = 127;                                    // all zero (or subnormal): leave as it is
    return PointScale{__uint_as_float((unsigned)se << 23), __uint_as_float((unsigned)(254 - se) << 23)};
}
// m: the largest |entry| this lane holds; the other half of the point's vector is in lane ^ 32.
__device__ __forceinline__ PointScale point_scale(float m) {
    float mo = m;
    swap_halves(m, mo);
    return scale_for_max(__float_as_uint(fmaxf(m, mo)));
}
// The largest of a value over the wavefront (values compared as unsigned: non-negative floats order like their
// bit patterns), in every lane; no LDS.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));    // lanes 1,0,3,2 of each quad
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));    // lanes 2,3,0,1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));   // row half mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));   // row mirror: rows uniform
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(r0, r1), max(r2, r3));
}
template <int NT>
__device__ __forceinline__ float abs_max(const f32x16 (&a)[NT], float m = 0.0f) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(fabsf(a[t][r]), fabsf(a[t][r + 1])), m);
    return m;
}

// The same over the positive parts (the largest output of a ReLU that has not been applied yet).
template <int NT>
__device__ __forceinline__ float pos_max(const f32x16 (&a)[NT], float m = 0.0f) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(a[t][r], a[t][r + 1]), m);
    return m;
}

// The weight stream of the f16 x 3 layers.  The packed layers of a kernel lie in memory in the order they are
// used and every chunk is 16 KiB (a k-step of 8 out tiles, or two k-steps of 4), so the stream is one linear
// run of chunks through RING rotating LDS buffers: while chunk c is multiplied, chunk c+1 has landed, chunks
// c+2 .. c+RING-2 are on their way and the copies of chunk c+RING-1 go out (in the middle of chunk c, after the
// barrier that frees the buffer of chunk c-1).  The wait for chunk c+1 in the middle of chunk c so concerns
// copies issued RING-2 chunks earlier: vector memory operations complete in issue order, and a shorter ring would
// make that wait also cover the row stores of the previous k-step, whose acknowledgement takes longer than a
// k-step.  Past the end of the stream the last chunk is copied again, so that the count of operations in flight
// stays what the waits assume.
constexpr int h3_lds_floats(int ring) { return ring * H3_CHUNK_FLOATS; }
constexpr int H3_PIECES = H3_CHUNK_FLOATS / 1024;   // 1 KiB pieces per wave per chunk
__device__ __forceinline__ const float* h3_next_chunk(const WeightStream& ws) {
    return ws.next < ws.end ? ws.next : ws.end - H3_CHUNK_FLOATS;
}
__device__ __forceinline__ void h3_copy_chunk(WeightStream& ws, int buf) {
    const float* src = h3_next_chunk(ws);
#pragma unroll
    for (int q = 0; q < H3_PIECES; ++q) ws_copy_piece(ws, src, ws.lds + buf * H3_CHUNK_FLOATS, q);
    ws.next += H3_CHUNK_FLOATS;
}
// Start the stream at `first`: request its first RING-1 chunks.  ws_ready() must follow before the first dense_h3().
template <int RING>
__device__ __forceinline__ void h3_prime(WeightStream& ws, const float* first, const float* end) {
    ws.next = first;
    ws.end = end;
#pragma unroll
    for (int c = 0; c < RING - 1; ++c) h3_copy_chunk(ws, c);
    ws.g3 = 0;
}

// Row stores a layer issues in its k-step k (none in k-step 0, and nothing is assumed about the previous layer),
// and in k-steps k0 .. k1.
template <bool DRAIN>
constexpr int h3_stores(int k) { return DRAIN && k > 0 ? 2 : 0; }
template <bool DRAIN>
constexpr int h3_stores(int k0, int k1) {
    int n = 0;
    for (int k = k0; k <= k1; ++k) n += h3_stores<DRAIN>(k);
    return n;
}

// What a layer does to its input tiles on their way in -- in place, as part of the operand split, i.e. in the
// shadow of the MFMAs instead of in the previous layer's epilogue:
//   PRE_RELU      in = max(in, 0); with DRAIN the ReLU bits are recorded and stored to sv.relu_bits_out
//   PRE_MASK      in = sv.bits ? in : 0
//   PRE_OUTSCALE  the rows DRAIN writes are in * sv.out_scale (the tiles themselves stay as they are)
constexpr int PRE_NONE = 0, PRE_RELU = 1, PRE_MASK = 2, PRE_OUTSCALE = 4;

// acc[ot] (+)= W_tile(ot, kt) * (scale * in[kt]), the layer's weights being the next KT * OT / 4 chunks of the
// stream.  A chunk is two groups of 12 MFMAs (4 out tiles x 3 terms): the two halves of a k-step (OT = 8) or two
// k-steps (OT = 4); one barrier per chunk, between its groups.
// ZERO: the accumulators start from zero (their contents on entry are ignored).
// DRAIN: also write the input tiles (unscaled) to sv.rows0, tile t during k-steps 2t and 2t+1.
template <int RING, int KT, int OT, bool DRAIN = false, bool ZERO = false, int PRE = PRE_NONE>
__device__ __forceinline__ void dense_h3(WeightStream& ws, f32x16 (&in)[KT], float scale, f32x16 (&acc)[OT],
                                         TileSave sv = TileSave{TileBuf{0, 0, 0, 0}, nullptr, 0}) {
    static_assert(OT == 4 || OT == 8, "out tiles are processed four at a time, two groups per chunk");
    constexpr int S = 2 * KT, G = OT / 4, NCH = S * G / 2;
    constexpr int NM = 12, NSPLIT = 11, SLOTS = NM * G;   // MFMAs per group; steps of the operand split; MFMAs per k-step
    constexpr int RMASK = RING - 1;
    static_assert(RING >= 4 && (RING & RMASK) == 0, "the ring is indexed with a mask");
    constexpr bool RELU = (PRE & PRE_RELU) != 0, MASK = (PRE & PRE_MASK) != 0, OUTSCALE = (PRE & PRE_OUTSCALE) != 0;
    constexpr bool BITS_OUT = RELU && DRAIN;
    static_assert(!(RELU && MASK) && KT <= 8, "one input transform; 8 tiles of ReLU bits fit 128 bits");
    f16x8 a[2][8];                   // operands of a group of 4 out tiles (tile*2 + piece), double-buffered
    u32x4 b[3];                      // this k-step's B pieces (h, h'', l); the next one's are built in nh, nh2, nl
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};   // two row blocks on their way out
    u32x4 nh, nh2, nl;
    float sr0[4], sr1[4];
    unsigned mw[4] = {0u, 0u, 0u, 0u};   // BITS_OUT: the ReLU bits so far
    unsigned mb[8], mp[4], mq[2], mf = 0u;
    int kp[8];
    const unsigned keep_words[4] = {sv.bits.x, sv.bits.y, sv.bits.z, sv.bits.w};
    // The split of the next k-step in eleven steps of at most six independent instructions, one step every few
    // MFMAs: an instruction issued right behind the one it depends on waits ~9 clocks with the MFMA behind it, and
    // more than five or six between two MFMAs no longer hide behind the first.  (Plain multiplies: a packed fp32
    // instruction beside an MFMA costs more than the two it replaces.)  Value j of the k-step is register 8u + j of
    // tile t (k-step 2t + u), bit 16 (t & 1) + 8u + j of word t / 2 of the ReLU bits.
    float xs[8];
    auto split_chunk = [&](auto cs, auto ck) {
        constexpr int s = decltype(cs)::value, k = decltype(ck)::value, t = s / 2, r0 = 8 * (s % 2), pos = 16 * (t & 1) + r0;
        if constexpr (k < 2) {
#pragma unroll
            for (int j = 4 * k; j < 4 * k + 4; ++j) {
                if constexpr (RELU) in[t][r0 + j] = fmaxf(in[t][r0 + j], 0.0f);
                if constexpr (MASK) kp[j] = __builtin_amdgcn_sbfe((int)keep_words[t >> 1], pos + j, 1);   // 0 or -1
            }
        } else if constexpr (k < 4) {
#pragma unroll
            for (int j = 4 * (k - 2); j < 4 * (k - 2) + 4; ++j) {
                if constexpr (BITS_OUT) mb[j] = min(__float_as_uint(in[t][r0 + j]), 1u);
                if constexpr (MASK) in[t][r0 + j] = __uint_as_float(__float_as_uint(in[t][r0 + j]) & (unsigned)kp[j]);
            }
        } else if constexpr (k < 6) {
#pragma unroll
            for (int j = 4 * (k - 4); j < 4 * (k - 4) + 4; ++j) xs[j] = in[t][r0 + j] * scale;
            if constexpr (BITS_OUT) {
                mp[2 * (k - 4)] = (mb[4 * (k - 4) + 1] << 1) | mb[4 * (k - 4)];
                mp[2 * (k - 4) + 1] = (mb[4 * (k - 4) + 3] << 1) | mb[4 * (k - 4) + 2];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (k == 6) nh[i] = cvt_pk_f16(xs[2 * i], xs[2 * i + 1]);
                else if constexpr (k == 7) sr0[i] = sub_f16<0>(xs[2 * i], nh[i]);
                else if constexpr (k == 8) sr1[i] = sub_f16<1>(xs[2 * i + 1], nh[i]);
                else if constexpr (k == 9) nl[i] = cvt_pk_f16(sr0[i], sr1[i]);
                else nh2[i] = pk_f16_down11(nh[i]);
            }
            if constexpr (BITS_OUT && k == 6) {
                mq[0] = (mp[1] << 2) | mp[0];
                mq[1] = (mp[3] << 2) | mp[2];
            }
            if constexpr (BITS_OUT && k == 7) mf = (mq[1] << 4) | mq[0];
            if constexpr (BITS_OUT && k == 8) mw[t >> 1] = (mf << pos) | mw[t >> 1];
        }
    };
    auto take_split = [&]() {
        b[0] = nh;
        b[1] = nh2;
        b[2] = nl;
    };
    auto read_one = [&](unsigned rd, auto cg, auto ce, f16x8 (&o)[8]) {   // group 0 or 1 of the chunk at rd
        constexpr int g = decltype(cg)::value, e = decltype(ce)::value;
        o[e] = lds_read_h8<((4 * g + e / 2) * 2 + e % 2) * 1024>(rd);
    };
    auto read_group = [&](unsigned rd, auto cg, f16x8 (&o)[8]) {
        static_for<0, 8>([&](auto ce) { read_one(rd, cg, ce, o); });
    };
    static_for<0, NSPLIT>([&](auto ck) { split_chunk(std::integral_constant<int, 0>{}, ck); });
    take_split();
    {
        const unsigned rd0 = lds_addr(ws.lds + ws.g3 * H3_CHUNK_FLOATS) + ws.lane * 16;
        read_group(rd0, std::integral_constant<int, 0>{}, a[0]);
    }
    static_for<0, S>([&](auto cs) {
        constexpr int s = decltype(cs)::value;
        static_for<0, G>([&](auto cg) {
            constexpr int g = decltype(cg)::value;
            constexpr int q = s * G + g, ch = q / 2, gg = q % 2;   // group q of the layer: half gg of its chunk ch
            constexpr int par = q % 2;
            const int b0 = (ws.g3 + ch) & RMASK;
            const unsigned rd = lds_addr(ws.lds + b0 * H3_CHUNK_FLOATS) + ws.lane * 16;
            const unsigned rd_next = lds_addr(ws.lds + ((b0 + 1) & RMASK) * H3_CHUNK_FLOATS) + ws.lane * 16;
            f16x8(&cur)[8] = a[par];
            f16x8(&nxt)[8] = a[par ^ 1];
            lgkm_wait();
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(cur[e]));
            if constexpr (DRAIN) {
                asm volatile("" : "+v"(d0));
                asm volatile("" : "+v"(d1));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* copy_src = nullptr;
            // the 12 MFMAs of the group, smallest terms first: l' * h'', h * l, h * h
            static_for<0, NM>([&](auto cm) {
                constexpr int m = decltype(cm)::value, term = m / 4, o = m % 4;
                constexpr int as = term == 0 ? 1 : 0;
                constexpr int bs = term == 0 ? 1 : term == 1 ? 2 : 0;
                if constexpr (ZERO && s == 0 && term == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[4 * g + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * o + as], __builtin_bit_cast(f16x8, b[bs]), zero, 0, 0, 0);
                } else {
                    acc[4 * g + o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[2 * o + as], __builtin_bit_cast(f16x8, b[bs]), acc[4 * g + o], 0, 0, 0);
                }
                asm volatile("" : "+a"(acc[4 * g + o]));
                if constexpr (m == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (gg == 1) {   // the middle of the chunk: chunk ch+1 must have landed
                        // Vector memory operations this wave has certainly issued after the copies of chunk ch+1
                        // (RING-2 chunks ago) may stay in flight: the copies of the chunks after it and the row
                        // stores since, which follow their k-step's copies.  This layer's k-step 0 has no stores;
                        // at the start of a layer only the previous layer's copies are counted.
                        constexpr int YOUNGER = (RING - 3) * H3_PIECES +
                                                (ch == 0 ? 0 : G == 2 ? h3_stores<DRAIN>(ch - (RING - 3), ch)
                                                                      : h3_stores<DRAIN>(2 * (ch - (RING - 2)) + 1, 2 * ch));
                        static_assert(YOUNGER < 64, "s_waitcnt vmcnt holds 6 bits");
                        ws_ready<YOUNGER>();
                        copy_src = h3_next_chunk(ws);
                        ws.next += H3_CHUNK_FLOATS;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // One memory request per MFMA where possible: the four waves run in step, and bursts from each of
                // them at the same moment queue up.  The next group's eight operands first,
                if constexpr (m >= 1 && m <= 8) {
                    if constexpr (gg == 0) read_one(rd, std::integral_constant<int, 1>{}, std::integral_constant<int, m - 1>{}, nxt);
                    else if constexpr (q + 1 < S * G) read_one(rd_next, std::integral_constant<int, 0>{}, std::integral_constant<int, m - 1>{}, nxt);
                }
                // between them the copies of chunk ch+RING-1 (after the barrier of this group),
                if constexpr (gg == 1 && m >= 2 && m % 2 == 0 && m <= 2 * H3_PIECES)
                    ws_copy_piece(ws, copy_src, ws.lds + ((b0 + RING - 1) & RMASK) * H3_CHUNK_FLOATS, m / 2 - 1);
                // then the two row blocks read during the previous k-step leave for memory (non-temporal: they are
                // read again only by another kernel, after gigabytes of other traffic),
                if constexpr (DRAIN && g == 0 && s > 0 && (m == 9 || m == 10)) {
                    constexpr int t = (s - 1) / 2, k0 = 2 * ((s - 1) % 2);
                    if constexpr (m == 9) __builtin_nontemporal_store(d0, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)(8 * k0) * sv.ld + 32 * t));
                    else __builtin_nontemporal_store(d1, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)(8 * (k0 + 1)) * sv.ld + 32 * t));
                }
                // an input tile goes into the transposition buffer and two of its row blocks come back out
                if constexpr (DRAIN && g == 0 && s % 2 == 0 && m >= 7 && m <= 10) {
                    constexpr int qq = m - 7;
                    if constexpr (OUTSCALE)
                        lds_write4<32 * qq>(sv.tb.wr, f32x4{in[s / 2][4 * qq] * sv.out_scale, in[s / 2][4 * qq + 1] * sv.out_scale,
                                                            in[s / 2][4 * qq + 2] * sv.out_scale, in[s / 2][4 * qq + 3] * sv.out_scale});
                    else
                        lds_write4<32 * qq>(sv.tb.wr, f32x4{in[s / 2][4 * qq], in[s / 2][4 * qq + 1], in[s / 2][4 * qq + 2], in[s / 2][4 * qq + 3]});
                }
                if constexpr (DRAIN && g == G - 1 && m == 11) {
                    constexpr int k0 = 2 * (s % 2);
                    d0 = lds_read4<8 * k0 * TILE_LD * 4>(sv.tb.rd);
                    d1 = lds_read4<8 * (k0 + 1) * TILE_LD * 4>(sv.tb.rd);
                }
                if constexpr (s + 1 >= S) __builtin_amdgcn_sched_barrier(0);
                if constexpr (s + 1 < S) {
                    constexpr int slot = NM * g + m;
                    static_for<(slot * NSPLIT + SLOTS - 1) / SLOTS, ((slot + 1) * NSPLIT + SLOTS - 1) / SLOTS>(
                        [&](auto ck) { split_chunk(std::integral_constant<int, s + 1>{}, ck); });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (s + 1 < S) take_split();
    });
    if constexpr (DRAIN) {   // the last two row blocks
        lgkm_wait();
        asm volatile("" : "+v"(d0));
        asm volatile("" : "+v"(d1));
        constexpr int t = (S - 1) / 2;
        __builtin_nontemporal_store(d0, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)16 * sv.ld + 32 * t));
        __builtin_nontemporal_store(d1, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)24 * sv.ld + 32 * t));
    }
    if constexpr (BITS_OUT) *sv.relu_bits_out = make_uint4(mw[0], mw[1], mw[2], mw[3]);
    ws.g3 = (ws.g3 + NCH) & RMASK;
}

// dot_rows() below with the vector in LDS (byte address v_lds): a vector-memory load at that point would wait
// for every weight copy in flight.  Half of the tiles per round trip.
template <int NT>
__device__ __forceinline__ float dot_rows_lds(const f32x16 (&a)[NT], unsigned v_lds, int h) {
    static_assert(NT % 2 == 0, "two halves");
    const unsigned a0 = v_lds + 16 * h;
    float s = 0.0f;
    static_for<0, 2>([&](auto chalf) {
        constexpr int t0 = decltype(chalf)::value * (NT / 2);
        f32x4 w[NT / 2][4];
        static_for<0, NT / 2>([&](auto ct) {
            static_for<0, 4>([&](auto cq) {
                w[ct][cq] = lds_read4<(32 * (t0 + decltype(ct)::value) + 8 * decltype(cq)::value) * 4>(a0);
            });
        });
        lgkm_wait();
#pragma unroll
        for (int t = 0; t < NT / 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("" : "+v"(w[t][q]));
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fmaf(a[t0 + t][4 * q + e], w[t][q][e], s);
            }
    });
    float so = s;
    swap_halves(s, so);
    return s + so;
}

// acc tiles <- per-neuron vector v[32*t + rowmap(r,h)] (bias as the initial accumulator).
template <int NT>
__device__ __forceinline__ void fill_rows(f32x16 (&acc)[NT], const float* __restrict__ v, int h) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(v + 32 * t + 8 * q + 4 * h);
            acc[t][4 * q + 0] = b.x;
            acc[t][4 * q + 1] = b.y;
            acc[t][4 * q + 2] = b.z;
            acc[t][4 * q + 3] = b.w;
        }
    }
}

// Epilogue of a dense_h3 layer: a = a * inv + bias[32t + rowmap(r, h)] (bias in LDS at byte address bias_lds;
// asm reads, see the note on LDS accesses above), optionally through a ReLU.
template <int NT, bool RELU>
__device__ __forceinline__ void finish_tiles(f32x16 (&a)[NT], float inv, unsigned bias_lds, int h) {
    const unsigned a0 = bias_lds + 16 * h;
    static_for<0, NT / 2>([&](auto cpair) {
        constexpr int t0 = 2 * decltype(cpair)::value;
        f32x4 bv[2][4];
        static_for<0, 2>([&](auto ct) {
            static_for<0, 4>([&](auto cq) {
                bv[ct][cq] = lds_read4<(32 * (t0 + decltype(ct)::value) + 8 * decltype(cq)::value) * 4>(a0);
            });
        });
        lgkm_wait();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("" : "+v"(bv[t][q]));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaf(a[t0 + t][4 * q + e], inv, bv[t][q][e]);
                    a[t0 + t][4 * q + e] = RELU ? fmaxf(v, 0.0f) : v;
                }
            }
    });
}
template <int NT>
__device__ __forceinline__ void scale_tiles(f32x16 (&a)[NT], float inv) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] *= inv;
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = fmaxf(a[t][r], 0.0f);
}

// Store the lane's part of a point's activation row: row -> this point's row
// (row-major [P][width]); each (t,q) group is 4 consecutive neurons = one float4.
template <int NT>
__device__ __forceinline__ void store_row(float* __restrict__ row, const f32x16 (&a)[NT], int h, bool valid) {
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(row + 32 * t + 8 * q + 4 * h) =
                make_float4(a[t][4 * q], a[t][4 * q + 1], a[t][4 * q + 2], a[t][4 * q + 3]);
}

// One bit per neuron: 1 where the (post-ReLU) activation is > 0.  Word t>>1, bit (t&1)*16 + r.
template <int NT>
__device__ __forceinline__ uint4 relu_bits(const f32x16 (&a)[NT]) {
    static_assert(NT <= 8, "at most 8 tiles fit 128 bits");
    unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[t >> 1] |= min(__float_as_uint(a[t][r]), 1u) << ((t & 1) * 16 + r);
    return make_uint4(m[0], m[1], m[2], m[3]);
}

// a[t][r] = bit ? a[t][r] : 0
template <int NT>
__device__ __forceinline__ void apply_bits(f32x16 (&a)[NT], uint4 bits) {
    const unsigned m[4] = {bits.x, bits.y, bits.z, bits.w};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int keep = ((int)(m[t >> 1] << (31 - ((t & 1) * 16 + r)))) >> 31;  // 0 or -1
            a[t][r] = __uint_as_float(__float_as_uint(a[t][r]) & (unsigned)keep);
        }
}

// Positional encoding of a 3-vector with L frequency bands as register tiles: element e of
// [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (each block 3 wide; 0 beyond the end) sits in
// tile e/32, register r, lane half h with e%32 = rowmap(r, h).
template <int NT, int L>
__device__ __forceinline__ void pe_tiles(f32x16 (&out)[NT], float x0, float x1, float x2, int h) {
    constexpr int WIDTH = 3 * (2 * L + 1);
    float val[32 * NT + 4];
    {
        float sn[3][L], cs[3][L];
        sincos_octaves<L>(x0, sn[0], cs[0]);
        sincos_octaves<L>(x1, sn[1], cs[1]);
        sincos_octaves<L>(x2, sn[2], cs[2]);
#pragma unroll
        for (int e = 0; e < 32 * NT + 4; ++e) {
            if (e >= WIDTH) val[e] = 0.0f;
            else if (e < 3) val[e] = e == 0 ? x0 : e == 1 ? x1 : x2;
            else {
                const int q = e - 3, f = q / 6, rr = q % 6;
                val[e] = rr >= 3 ? cs[rr - 3][f < L ? f : 0] : sn[rr][f < L ? f : 0];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[t][r] = h ? val[32 * t + rowmap(r, 0) + 4] : val[32 * t + rowmap(r, 0)];
}

__device__ __forceinline__ float softplus_f(float x) {  // F.softplus, beta 1, threshold 20
    return x > 20.0f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum over neurons of a[t][r] * v[32t + rowmap(r,h)], reduced across the two lane halves.
template <int NT>
__device__ __forceinline__ float dot_rows(const f32x16 (&a)[NT], const float* __restrict__ v, int h) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(v + 32 * t + 8 * q + 4 * h);
            s = fmaf(a[t][4 * q + 0], w.x, s);
            s = fmaf(a[t][4 * q + 1], w.y, s);
            s = fmaf(a[t][4 * q + 2], w.z, s);
            s = fmaf(a[t][4 * q + 3], w.w, s);
        }
    return s + __shfl_xor(s, 32, 64);
}

// ---------------------------------------------------------------------------
// Weight packing jobs (run once per optimiser step).
// ---------------------------------------------------------------------------
struct PackJob {
    const float* src;  // nn.Linear weight [N][ld] (row = output neuron)
    int dst;           // float offset in the packed buffer
    int ld, N;         // source leading dimension and row count
    int mode;          // 0: forward pack, 1: backward (transposed) pack, 2: plain copy of `count` floats, 3: zero fill
    int hash_perm;     // hash-grid first layer: tile position rowmap(r,h) stands for feature 16h + r
    int KT, OT;        // tiles of the contraction / output dimension of the packed operand
    int h3;            // write the operand as two f16 pieces in the dense_h3 order (the same size)
    int seg_eff[2], seg_len[2], seg_src[2];  // forward: effective input k in [eff, eff+len) reads column src + (k-eff)
    int col_begin, col_count;                 // backward: output m in [m_first, m_first+col_count) reads column col_begin + (m - m_first)
    int m_first;
    int count;                                // copy mode
    const float* add_src;                     // optional second weight added element-wise (same rows; forward: segment 0 only)
    int add_ld;
};
constexpr int MAX_PACK_JOBS = 40;
struct PackJobs {
    PackJob job[MAX_PACK_JOBS];
    int n;
};

int launch_pack(const PackJobs& jobs, float* packed, hipStream_t stream);

// ---------------------------------------------------------------------------
// Weight-gradient work list: dW[n][col_off + k] += sum_p dY[p][n] * X[p][k].
// ---------------------------------------------------------------------------
struct WgradPart {
    const float* dY;   // [P32][ldY], columns 0 .. N
    const float* X;    // [P32][ldX], columns 0 .. 32*KT
    const float* X2;   // optional second input block [P32][ldX2], columns 0 .. 32*KT2
    float* dW;         // nn.Linear gradient [N][ldW]
    float* dbias;      // [N] or null
    int ldY, ldX, ldX2, ldW;
    int col_off, col_off2;   // gradient column of input 0 of the X / X2 block (may be negative with k_first)
    int k_first;             // X inputs below k_first are skipped (not written)
    int k_valid, k_valid2;   // inputs >= k_valid within a block are padding and are not written
    int n_valid;             // output rows >= n_valid are padding and are not written
    const unsigned* amaxY;   // the largest |entry| of dY, X, X2 as float bit patterns in device memory (the split-f16
    const unsigned* amaxX;   // shapes scale their operands by them); null: the matrix is used as it is
    const unsigned* amaxX2;
    const float* sk_d;       // optional rider on a split-f16 part: a per-point scalar d[P32] whose product with X is
    float* sk_dW;            // accumulated alongside, sk_dW[k] += sum_p d[p] X[p][k], sk_db[0] += sum_p d[p] (the gradient
    float* sk_db;            // of a one-output head that reads the same X, in plain fp32 on the lanes that load X)
    int shape;         // index into the (NTW, KT, KT2) instantiation list
    int slabs;         // point slabs this part is split into
    int first_block;   // prefix sum of slabs
};
constexpr int MAX_WGRAD_PARTS = 24;
struct WgradParts {
    WgradPart part[MAX_WGRAD_PARTS];
    int n;
    int total_blocks;
    long long P32;
};

// Gradients of heads with at most 4 outputs (sigma, rgb): dW[c][k] += sum_p dY[p][c0+c] X[p][k].
struct SkinnyPart {
    const float* dY;  // [P32][ldY]
    const float* X;   // [P32][ldX]
    float* dW;        // [C][K]
    float* dbias;     // [C]
    int ldY, ldX, c0, C, K;
    const int* count = nullptr;   // rows in use, in device memory (null: all P32 of them)
};

// Largest-|entry| bookkeeping of the split-f16 weight gradients: every wavefront of the kernel that writes a
// matrix stores the largest |entry| of its 32 rows (float bits) at per_wave[16 * tile + slot]; this reduces the 16
// slots over all tiles into out[0..15].  (Plain stores and one small launch: atomics on 16 shared words from
// 25,000 wavefronts cost the writing kernel 0.8 ms.)
constexpr int AMAX_SLOTS = 16;
int launch_amax_reduce(const unsigned* per_wave, long long tiles, unsigned* out, hipStream_t stream);

// Fills slabs/first_block/total_blocks and launches one workgroup per slab.
int launch_wgrad(WgradParts& parts, long long P32, hipStream_t stream);
int launch_skinny(const SkinnyPart& part, long long P32, hipStream_t stream);

inline PackJob pack_fwd_job(const float* src, int ld, int N, int dst, int KT, int OT, int eff0, int len0, int src0,
                            int eff1 = 0, int len1 = 0, int src1 = 0) {
    PackJob j{};
    j.src = src; j.dst = dst; j.ld = ld; j.N = N; j.mode = 0; j.KT = KT; j.OT = OT;
    j.seg_eff[0] = eff0; j.seg_len[0] = len0; j.seg_src[0] = src0;
    j.seg_eff[1] = eff1; j.seg_len[1] = len1; j.seg_src[1] = src1;
    return j;
}
inline PackJob pack_bwd_job(const float* src, int ld, int N, int dst, int KT, int OT, int col_begin, int col_count) {
    PackJob j{};
    j.src = src; j.dst = dst; j.ld = ld; j.N = N; j.mode = 1; j.KT = KT; j.OT = OT;
    j.col_begin = col_begin; j.col_count = col_count;
    return j;
}
inline PackJob pack_copy_job(const float* src, int dst, int count) {
    PackJob j{};
    j.src = src; j.dst = dst; j.mode = 2; j.count = count;
    return j;
}
inline PackJob pack_zero_job(int dst, int count) {
    PackJob j{};
    j.dst = dst; j.mode = 3; j.count = count;
    return j;
}
inline WgradPart wgrad_part_desc(const float* dY, int ldY, const float* X, int ldX, float* dW, int ldW, int col_off,
                                 int k_valid, float* dbias, int shape, const float* X2 = nullptr, int ldX2 = 0,
                                 int col_off2 = 0, int k_valid2 = 0) {
    WgradPart p{};
    p.dY = dY; p.ldY = ldY; p.X = X; p.ldX = ldX; p.dW = dW; p.ldW = ldW; p.col_off = col_off; p.k_valid = k_valid;
    p.dbias = dbias; p.shape = shape; p.X2 = X2; p.ldX2 = ldX2; p.col_off2 = col_off2; p.k_valid2 = k_valid2;
    p.k_first = 0; p.n_valid = 1 << 30;
    return p;
}

}  // namespace nerf
