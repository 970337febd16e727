// Building blocks of the fused MLP kernels (gfx950, exact-f32 MFMA).
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
// The packed weights of consecutive dense() calls form one stream of chunks of 16
// k-steps (at most 32 KiB).  Three chunk buffers rotate.  In the middle of chunk g
// every wave waits for its own copies of chunk g+1, the workgroup meets at a
// barrier (so chunk g+1 is complete, and nobody reads chunk g-1 any more), and the
// copies of chunk g+2 go out into the buffer chunk g-1 occupied.  The barrier sits
// between two MFMA groups with the next operands already requested, so the pipe
// only sees the skew between the waves.  All four waves must make the same calls.
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
    int g3;       // buffer of the chunk the next dense layer starts with
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
// Rows to write underneath a dense_x6() call: its input tiles, to rows0[32][ld].
struct TileSave {
    TileBuf tb;
    float* rows0;   // row 0 of the tile's 32 rows, already offset by this lane's (row, col)
    int ld;
};
__device__ __forceinline__ TileSave tile_save(const TileBuf& tb, float* tile_rows0, int ld = 256) {
    return TileSave{tb, tile_rows0 + (size_t)tb.row * ld + tb.col, ld};
}

// Write NT register tiles to rows0[32][ld] now (not overlapped with anything).
template <int NT>
__device__ __forceinline__ void save_tiles(const TileBuf& tb, float* __restrict__ rows0, int ld, const f32x16 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        static_for<0, 4>([&](auto cq) {
            constexpr int q = decltype(cq)::value;
            lds_write4<32 * q>(tb.wr, f32x4{a[t][4 * q], a[t][4 * q + 1], a[t][4 * q + 2], a[t][4 * q + 3]});
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

// ------------------------------------------------------ bf16 x 6 dense ----
// The same contraction at 2.6 times the rate of the exact-f32 MFMA and the same accuracy
// (measured: max error / sum|terms| 1.3e-7 for K = 256, both): every fp32 operand is
// split into three bf16 numbers, v = h + m + l with |m| <= 2^-9 |v| and |l| <= 2^-18 |v|,
// and a product is the six bf16 MFMAs that matter (hh, hm, mh, mm, hl, lh; the rest is
// below 2^-26), smallest terms first, accumulated in fp32 by v_mfma_f32_32x32x16_bf16
// (2.5 PFLOP/s dense: 16x the fp32 matrix rate, so 6 of them cost 3/8 of one fp32 MFMA
// pass over the same K).  The weights are split once per optimiser step by the pack
// kernel; the activations on the fly, in the shadow of the previous k-step's MFMAs.
//
// A k-step is K = 16: the B operand of lane (point p, half h) is eight of its registers
// of the input tile (r = 8u..8u+7 for k-step 2t+u), i.e. neurons 32t + rowmap(r, h); the
// packed weights follow the same order: x6[k-step][out tile][split][lane][8 bf16], one
// k-step = 3 KiB per out tile = one chunk of the weight stream.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X6_CHUNK_FLOATS = 8 * 3 * 256;              // largest chunk: a k-step of 8 out tiles (24 KiB)
constexpr int x6_layer_floats(int KT, int OT) { return 2 * KT * OT * 3 * 256; }

__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hh = (__bf16)v[i];
        const float r1 = v[i] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[i] = hh;
        m[i] = mm;
        l[i] = (__bf16)(r1 - (float)mm);
    }
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// {bf16(a), bf16(b)} rounded to nearest even, a in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int OFF_BYTES>
__device__ __forceinline__ bf16x8 lds_read_bf8(unsigned addr) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF_BYTES));
    return r;
}
// The weight stream of the bf16 x 6 layers.  The packed layers of a kernel lie in memory in the order they are
// used and every chunk is 24 KiB (a k-step of 8 out tiles, or two k-steps of 4), so the stream is one linear
// run of chunks through four rotating LDS buffers: while chunk c is multiplied, chunk c+1 has landed, chunk
// c+2 is on its way and the copies of chunk c+3 go out (in the middle of chunk c, after the barrier that frees
// the buffer of chunk c-1).  The wait for chunk c+1 in the middle of chunk c so concerns copies issued two
// chunks earlier: vector memory operations complete in issue order, and with a ring of three that wait also
// covered the row stores of the previous k-step, whose acknowledgement takes longer than a k-step.
// Past the end of the stream the last chunk is copied again, so that the count of operations in flight
// stays what the waits assume.
constexpr int X6_RING = 4;
constexpr int X6_LDS_FLOATS = X6_RING * X6_CHUNK_FLOATS;
constexpr int X6_PIECES = X6_CHUNK_FLOATS / 1024;   // 1 KiB pieces per wave per chunk
__device__ __forceinline__ const float* x6_next_chunk(const WeightStream& ws) {
    return ws.next < ws.end ? ws.next : ws.end - X6_CHUNK_FLOATS;
}
__device__ __forceinline__ void x6_copy_chunk(WeightStream& ws, int buf) {
    const float* src = x6_next_chunk(ws);
#pragma unroll
    for (int q = 0; q < X6_PIECES; ++q) ws_copy_piece(ws, src, ws.lds + buf * X6_CHUNK_FLOATS, q);
    ws.next += X6_CHUNK_FLOATS;
}
// Start the stream at `first`: request its first three chunks.  ws_ready() must follow before the first dense_x6().
__device__ __forceinline__ void x6_prime(WeightStream& ws, const float* first, const float* end) {
    ws.next = first;
    ws.end = end;
    x6_copy_chunk(ws, 0);
    x6_copy_chunk(ws, 1);
    x6_copy_chunk(ws, 2);
    ws.g3 = 0;
}

// Row stores a layer issues in its k-step k (none in k-step 0, and nothing is assumed about the previous layer).
template <bool DRAIN>
constexpr int x6_stores(int k) { return DRAIN && k > 0 ? 2 : 0; }

// acc[ot] += W_tile(ot, kt) * in[kt], the layer's weights being the next KT * OT / 4 chunks of the stream.
// A chunk is two groups of 24 MFMAs (4 out tiles x 6 terms): the two halves of a k-step (OT = 8) or two
// k-steps (OT = 4); one barrier per chunk, between its groups.
// DRAIN: also write the input tiles to sv.rows0, tile t during k-steps 2t and 2t+1.
template <int KT, int OT, bool DRAIN = false>
__device__ __forceinline__ void dense_x6(WeightStream& ws, const f32x16 (&in)[KT], f32x16 (&acc)[OT],
                                         TileSave sv = TileSave{TileBuf{0, 0, 0, 0}, nullptr, 0}) {
    static_assert(OT == 4 || OT == 8, "out tiles are processed four at a time, two groups per chunk");
    constexpr int S = 2 * KT, G = OT / 4, NCH = S * G / 2;
    bf16x8 a[2][12];                 // operands of a group of 4 out tiles (tile*3 + split), double-buffered
    bf16x8 b[3];                     // this k-step's B splits (h, m, l); the next one's are built in nh, nm, nl
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};   // two row blocks on their way out
    auto split_step = [&](auto cs, bf16x8 (&o)[3]) {
        constexpr int s = decltype(cs)::value;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = in[s / 2][8 * (s % 2) + i];
        split3(v, o[0], o[1], o[2]);
    };
    // The same split for the next k-step in eleven chunks of four independent instructions, one chunk every few
    // MFMAs: an instruction issued right behind the one it depends on waits ~9 clocks with the MFMA behind it.
    u32x4 nh, nm, nl;
    float sr0[4], sr1[4], sf0[4], sf1[4];
    auto split_chunk = [&](auto cs, auto ck) {
        constexpr int s = decltype(cs)::value, k = decltype(ck)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x0 = in[s / 2][8 * (s % 2) + 2 * i], x1 = in[s / 2][8 * (s % 2) + 2 * i + 1];
            if constexpr (k == 0) nh[i] = cvt_pk_bf16(x0, x1);
            else if constexpr (k == 1) sf0[i] = __uint_as_float(nh[i] << 16);
            else if constexpr (k == 2) sf1[i] = __uint_as_float(nh[i] & 0xffff0000u);
            else if constexpr (k == 3) sr0[i] = x0 - sf0[i];
            else if constexpr (k == 4) sr1[i] = x1 - sf1[i];
            else if constexpr (k == 5) nm[i] = cvt_pk_bf16(sr0[i], sr1[i]);
            else if constexpr (k == 6) sf0[i] = __uint_as_float(nm[i] << 16);
            else if constexpr (k == 7) sf1[i] = __uint_as_float(nm[i] & 0xffff0000u);
            else if constexpr (k == 8) sr0[i] = sr0[i] - sf0[i];
            else if constexpr (k == 9) sr1[i] = sr1[i] - sf1[i];
            else nl[i] = cvt_pk_bf16(sr0[i], sr1[i]);
        }
    };
    auto read_one = [&](unsigned rd, auto cg, auto ce, bf16x8 (&o)[12]) {   // group 0 or 1 of the chunk at rd
        constexpr int g = decltype(cg)::value, e = decltype(ce)::value;
        o[e] = lds_read_bf8<((4 * g + e / 3) * 3 + e % 3) * 1024>(rd);
    };
    auto read_group = [&](unsigned rd, auto cg, bf16x8 (&o)[12]) {
        static_for<0, 12>([&](auto ce) { read_one(rd, cg, ce, o); });
    };
    split_step(std::integral_constant<int, 0>{}, b);
    {
        const unsigned rd0 = lds_addr(ws.lds + ws.g3 * X6_CHUNK_FLOATS) + ws.lane * 16;
        read_group(rd0, std::integral_constant<int, 0>{}, a[0]);
    }
    static_for<0, S>([&](auto cs) {
        constexpr int s = decltype(cs)::value;
        static_for<0, G>([&](auto cg) {
            constexpr int g = decltype(cg)::value;
            constexpr int q = s * G + g, ch = q / 2, gg = q % 2;   // group q of the layer: half gg of its chunk ch
            constexpr int par = q % 2;
            const int b0 = (ws.g3 + ch) & 3;
            const unsigned rd = lds_addr(ws.lds + b0 * X6_CHUNK_FLOATS) + ws.lane * 16;
            const unsigned rd_next = lds_addr(ws.lds + ((b0 + 1) & 3) * X6_CHUNK_FLOATS) + ws.lane * 16;
            bf16x8(&cur)[12] = a[par];
            bf16x8(&nxt)[12] = a[par ^ 1];
            lgkm_wait();
#pragma unroll
            for (int e = 0; e < 12; ++e) asm volatile("" : "+v"(cur[e]));
            if constexpr (DRAIN) {
                asm volatile("" : "+v"(d0));
                asm volatile("" : "+v"(d1));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[4 * g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[2], b[0], acc[4 * g], 0, 0, 0);   // l * h of tile 0
            asm volatile("" : "+a"(acc[4 * g]));
            if constexpr (s + 1 < S) {
                static_for<(24 * g * 11 + 24 * G - 1) / (24 * G), ((24 * g + 1) * 11 + 24 * G - 1) / (24 * G)>(
                    [&](auto ck) { split_chunk(std::integral_constant<int, s + 1>{}, ck); });
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* copy_src = nullptr;
            if constexpr (gg == 1) {   // the middle of the chunk: chunk ch+1 must have landed
                // Vector memory operations this wave has certainly issued after the copies of chunk ch+1 (two
                // chunks ago) may stay in flight: the copies of chunk ch+2 and the row stores since, which follow
                // their k-step's copies (MFMAs 13 and 15).  This layer's k-step 0 has no stores; at the start of a
                // layer only the previous layer's copies are counted.
                constexpr int YOUNGER = X6_PIECES + (ch == 0 ? 0 : G == 2 ? x6_stores<DRAIN>(ch - 1) + x6_stores<DRAIN>(ch)
                                                                   : x6_stores<DRAIN>(2 * ch - 3) + x6_stores<DRAIN>(2 * ch - 2) +
                                                                     x6_stores<DRAIN>(2 * ch - 1) + x6_stores<DRAIN>(2 * ch));
                ws_ready<YOUNGER>();
                copy_src = x6_next_chunk(ws);
                ws.next += X6_CHUNK_FLOATS;
            }
            __builtin_amdgcn_sched_barrier(0);
            // the other 23 MFMAs of the group, smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h
            static_for<1, 24>([&](auto cm) {
                constexpr int m = decltype(cm)::value, term = m / 4, o = m % 4;
                constexpr int as = term == 0 ? 2 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 1 : 0;
                constexpr int bs = term == 0 ? 0 : term == 1 ? 2 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 1 : 0;
                acc[4 * g + o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[3 * o + as], b[bs], acc[4 * g + o], 0, 0, 0);
                asm volatile("" : "+a"(acc[4 * g + o]));
                // One memory request per MFMA: the four waves run in step, and bursts from each of them at the same
                // moment queue up.  The next group's twelve operands first,
                if constexpr (m <= 12) {
                    if constexpr (gg == 0) read_one(rd, std::integral_constant<int, 1>{}, std::integral_constant<int, m - 1>{}, nxt);
                    else if constexpr (q + 1 < S * G) read_one(rd_next, std::integral_constant<int, 0>{}, std::integral_constant<int, m - 1>{}, nxt);
                }
                // between them the copies of chunk ch+3 (after the barrier of this group),
                if constexpr (gg == 1 && m % 2 == 0 && m <= 12)
                    ws_copy_piece(ws, copy_src, ws.lds + ((b0 + 3) & 3) * X6_CHUNK_FLOATS, m / 2 - 1);
                // then the two row blocks read during the previous k-step leave for memory (non-temporal: they are
                // read again only by another kernel, after gigabytes of other traffic),
                if constexpr (DRAIN && g == 0 && s > 0 && (m == 13 || m == 15)) {
                    constexpr int t = (s - 1) / 2, k0 = 2 * ((s - 1) % 2);
                    if constexpr (m == 13) __builtin_nontemporal_store(d0, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)(8 * k0) * sv.ld + 32 * t));
                    else __builtin_nontemporal_store(d1, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)(8 * (k0 + 1)) * sv.ld + 32 * t));
                }
                // an input tile goes into the transposition buffer and two of its row blocks come back out
                if constexpr (DRAIN && g == 0 && s % 2 == 0 && m >= 13 && m <= 16) {
                    constexpr int qq = m - 13;
                    lds_write4<32 * qq>(sv.tb.wr, f32x4{in[s / 2][4 * qq], in[s / 2][4 * qq + 1], in[s / 2][4 * qq + 2], in[s / 2][4 * qq + 3]});
                }
                if constexpr (DRAIN && g == G - 1 && (m == 17 || m == 18)) {
                    constexpr int k0 = 2 * (s % 2);
                    if constexpr (m == 17) d0 = lds_read4<8 * k0 * TILE_LD * 4>(sv.tb.rd);
                    else d1 = lds_read4<8 * (k0 + 1) * TILE_LD * 4>(sv.tb.rd);
                }
                if constexpr (s + 1 >= S) __builtin_amdgcn_sched_barrier(0);
                if constexpr (s + 1 < S) {
                    constexpr int slot = 24 * g + m;
                    static_for<(slot * 11 + 24 * G - 1) / (24 * G), ((slot + 1) * 11 + 24 * G - 1) / (24 * G)>(
                        [&](auto ck) { split_chunk(std::integral_constant<int, s + 1>{}, ck); });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (s + 1 < S) {
            b[0] = __builtin_bit_cast(bf16x8, nh);
            b[1] = __builtin_bit_cast(bf16x8, nm);
            b[2] = __builtin_bit_cast(bf16x8, nl);
        }
    });
    if constexpr (DRAIN) {   // the last two row blocks
        lgkm_wait();
        asm volatile("" : "+v"(d0));
        asm volatile("" : "+v"(d1));
        constexpr int t = (S - 1) / 2;
        __builtin_nontemporal_store(d0, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)16 * sv.ld + 32 * t));
        __builtin_nontemporal_store(d1, reinterpret_cast<f32x4*>(sv.rows0 + (size_t)24 * sv.ld + 32 * t));
    }
    ws.g3 = (ws.g3 + NCH) & 3;
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
    return s + __shfl_xor(s, 32, 64);
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
    int x6;            // write the operand as three bf16 splits in the dense_x6 order (1.5x the size)
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
};

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
