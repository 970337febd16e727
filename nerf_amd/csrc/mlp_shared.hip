// Kernels shared by the fused MLPs: weight packing, the weight-gradient GEMMs of the 8x256 MLP
// (contraction over points on split-f16 MFMAs) and the small-head gradients.
#include "mlp_common.h"

namespace nerf {

// ------------------------------------------------------------- packing ----
__global__ void pack_kernel(PackJobs jobs, float* __restrict__ packed) {
    const PackJob& jb = jobs.job[blockIdx.y];
    const int total = jb.mode >= 2 ? jb.count : jb.KT * 16 * 64 * jb.OT;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (jb.mode == 2) {
            v = jb.src[idx];
        } else if (jb.mode == 3) {
            v = 0.0f;
        } else {
            const int ot = idx % jb.OT;
            const int lane = (idx / jb.OT) % 64;
            const int r = (idx / (jb.OT * 64)) % 16;
            const int kt = idx / (jb.OT * 64 * 16);
            int out_i = 32 * ot + (lane & 31);                 // row of the packed operand (MFMA row i)
            int k = 32 * kt + rowmap(r, lane >> 5);            // contraction index
            if (jb.hash_perm) {   // tile position v = rowmap(r,h) <-> feature 16h + r
                int& v32 = jb.mode == 0 ? k : out_i;
                v32 = 16 * ((v32 >> 2) & 1) + (v32 & 3) + 4 * (v32 >> 3);
            }
            if (jb.mode == 0) {  // forward: operand[n][k_eff] = W[n][col(k_eff)]
                int col = -1;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (k >= jb.seg_eff[s] && k < jb.seg_eff[s] + jb.seg_len[s]) col = jb.seg_src[s] + (k - jb.seg_eff[s]);
                if (out_i < jb.N && col >= 0) {
                    v = jb.src[(size_t)out_i * jb.ld + col];
                    if (jb.add_src && k >= jb.seg_eff[0] && k < jb.seg_eff[0] + jb.seg_len[0])
                        v += jb.add_src[(size_t)out_i * jb.add_ld + col];
                }
            } else {             // backward: operand[m][n] = W[n][col_begin + (m - m_first)]
                const int m = out_i - jb.m_first;
                if (k < jb.N && m >= 0 && m < jb.col_count) {
                    v = jb.src[(size_t)k * jb.ld + jb.col_begin + m];
                    if (jb.add_src) v += jb.add_src[(size_t)k * jb.add_ld + jb.col_begin + m];
                }
            }
        }
        if (jb.h3 && jb.mode < 2) {
            // two f16 pieces for dense_h3: [k-step = 2*kt + r/8][out tile][h, l'][lane][r%8], l' = 2^11 (v - h)
            const int ot = idx % jb.OT, lane = (idx / jb.OT) % 64, r = (idx / (jb.OT * 64)) % 16, kt = idx / (jb.OT * 64 * 16);
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)((v - (float)hh) * 2048.0f);
            _Float16* out = reinterpret_cast<_Float16*>(packed + jb.dst);
            const size_t base = ((size_t)((2 * kt + (r >> 3)) * jb.OT + ot) * 2 * 64 + lane) * 8 + (r & 7);
            out[base] = hh;
            out[base + 64 * 8] = ll;
        } else {
            packed[jb.dst + idx] = v;
        }
    }
}

int launch_pack(const PackJobs& jobs, float* packed, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3(64, jobs.n), dim3(256), 0, stream, jobs, packed);
    return check_launch("pack");
}

// ------------------------------------------------------- operand maxima ----
__global__ __launch_bounds__(1024) void amax_reduce_kernel(const unsigned* __restrict__ per_wave, long long tiles,
                                                           unsigned* __restrict__ out) {
    __shared__ unsigned part[16];
    const int slot = blockIdx.x;
    unsigned m = 0;
    for (long long t = threadIdx.x; t < tiles; t += blockDim.x) m = max(m, per_wave[t * AMAX_SLOTS + slot]);
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) m = max(m, part[w]);
        out[slot] = m;
    }
}

int launch_amax_reduce(const unsigned* per_wave, long long tiles, unsigned* out, hipStream_t stream) {
    hipLaunchKernelGGL(amax_reduce_kernel, dim3(AMAX_SLOTS), dim3(1024), 0, stream, per_wave, tiles, out);
    return check_launch("amax_reduce");
}

// ---------------------------------------------------------------- wgrad ----
// One workgroup (4 waves) owns one point slab of one part; wave w owns n-tiles
// [w*NTW, (w+1)*NTW) and all k-tiles (KT from X plus KT2 from X2), i.e.
// NTW*(KT+KT2) accumulator tiles (up to 256 registers).
constexpr int WG_LDS_FLOATS = 3 * 16 * (3 * 256 + 4);   // three k-steps of 16 split operand tiles (145 KiB)

// Adds a wave's accumulator tiles (and bias sums) to the gradient tensors.
// dW[n0 + 32nt + rowmap(r,s)][col + i]: per register two 128-byte row segments
// scale, scale2: factors applied to the accumulators of the X and X2 blocks on their way out.
template <int NTW, int KT, int KT2>
__device__ __forceinline__ void wgrad_finish(const WgradPart& pt, f32x16 (&acc)[NTW][KT + KT2], float (&bsum)[NTW], int n0,
                                             int i, int s, bool with_bias = true, float scale = 1.0f, float scale2 = 1.0f) {
    constexpr int KTT = KT + KT2;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt) {
            const int k = (kt < KT ? 32 * kt : 32 * (kt - KT)) + i;
            const int col = (kt < KT ? pt.col_off : pt.col_off2) + k;
            if (k < (kt < KT ? pt.k_valid : pt.k_valid2) && (kt >= KT || k >= pt.k_first)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 32 * nt + rowmap(r, s);
                    float v = acc[nt][kt][r];
                    asm volatile("" : "+v"(v)::"memory");   // one product at a time: hoisted above the atomics they all spill
                    if (n < pt.n_valid) atomicAdd(pt.dW + (size_t)n * pt.ldW + col, v * (kt < KT ? scale : scale2));
                }
            }
        }
    if (pt.dbias && with_bias) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
            if (s == 0 && n0 + 32 * nt + i < pt.n_valid) atomicAdd(pt.dbias + n0 + 32 * nt + i, b);
        }
    }
}

// ---- The product on split-f16 MFMAs, the operands split once per workgroup --------------------------
// K = 16 points per MFMA: lane (i, hb) holds points 8hb..8hb+7 of column i of dY (A operand) or of X (B operand).
// If every wave split each X tile it multiplies with, a workgroup would do that work four times.  Here a k-step
// (16 points) of dY and X is loaded into registers once, split there into two f16 pieces (see dense_h3 in
// mlp_common.h: v = h + l, three MFMAs per product) and stored to LDS as ready operands:
// [tile][h, l][operand lane][8 points], which every wave then reads with one 16-byte load per operand.
// Both matrices are first scaled by a power of two that brings their largest |entry| (found by the kernels that
// wrote them) into [2^13, 2^14): both residuals then stay f16 normal numbers for every entry within 2^-16 of
// the largest, and smaller entries keep an absolute error below 2^-38 of the largest.  The product of the two
// scales is removed when the accumulators are added to the gradient.
// Three buffers of one k-step rotate: k-step s is multiplied while k-step s+2 is split and stored and
// k-step s+4 is on its way from memory, so the one barrier per k-step only has skew to wait for.
// A tile's stride is padded by 16 bytes so that the four tiles a wave stores at once fall on different banks.
constexpr int PK_TILE_BYTES = 2 * 1024 + 16;
constexpr int PK_BUF_BYTES = 16 * PK_TILE_BYTES;
constexpr int PK_CHUNKS = 40;   // steps of the operand split of a k-step (see wgrad_part_pk)
static_assert(3 * PK_BUF_BYTES <= WG_LDS_FLOATS * 4, "operand buffers exceed the LDS allocation");

template <int OFF_BYTES>
__device__ __forceinline__ void lds_write_u4(unsigned addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF_BYTES) : "memory");
}

__device__ __forceinline__ PointScale part_scale(const unsigned* amax) {
    return amax ? scale_for_max(*amax) : PointScale{1.0f, 1.0f};
}

template <int NTW, int KT, int KT2, int GROUP, bool SK = false>
__device__ __forceinline__ void wgrad_part_pk(const WgradPart& pt, long long p_begin, long long p_end, int wave, int lane,
                                              float* lds) {
    constexpr int NT = 4 * NTW, KTT = KT + KT2, TILES = NT + KTT, J = KTT / GROUP, NM = 3 * NTW * GROUP;
    static_assert(TILES <= 16 && KTT % GROUP == 0 && (NTW + GROUP) * 2 <= NM, "shape");
    constexpr int PTAIL = 6;   // the last 1/PTAIL of a k-step's MFMAs carry no split work
    const int i = lane & 31, s = lane >> 5;
    const int n0 = wave * NTW * 32;
    const PointScale scY = part_scale(pt.amaxY), scX = part_scale(pt.amaxX), scX2 = part_scale(pt.amaxX2);
    f32x16 acc[NTW][KTT];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][kt][r] = 0.0f;

    // producer role: columns 4pj .. 4pj+3 of tile T (dY tiles first, then X, then X2), points 8ps .. 8ps+7 of a k-step
    // lane bits: 0 = column quad bit 0, 1-2 = tile, 3-4 = column quad bits 1-2, 5 = point half: any eight (and
    // sixteen) consecutive lanes store to different banks (tile stride = 129 x 16 B, column quad = 4 x 16 B)
    const int pj = (lane & 1) | ((lane >> 2) & 6), ptl = (lane >> 1) & 3, ps = lane >> 5;
    const int T = 4 * wave + ptl;
    const bool active = T < TILES;
    const float* src = pt.dY + 4 * pj;   // lanes without a tile load (and store) something harmless
    long long ld = pt.ldY;
    float c1 = scY.s;                    // the scale of this lane's matrix
    if (T < NT) {
        src = pt.dY + 32 * T + 4 * pj;
        ld = pt.ldY;
    } else if (T < NT + KT) {
        src = pt.X + 32 * (T - NT) + 4 * pj;
        ld = pt.ldX;
        c1 = scX.s;
    } else if (active) {
        src = pt.X2 + 32 * (T - NT - KT) + 4 * pj;
        ld = pt.ldX2;
        c1 = scX2.s;
    }
    src += 8 * ps * ld;
    const unsigned lds0 = lds_addr(lds);
    const unsigned wr_off = T * PK_TILE_BYTES + (4 * pj + 32 * ps) * 16;
    const unsigned rd_a_off = NTW * wave * PK_TILE_BYTES + lane * 16, rd_b_off = lane * 16;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of this lane's dY columns (the bias gradient)
    float4 R[2][8];
    // the rider (see WgradPart::sk_d): this lane's eight points of the per-point scalar, and its sums
    constexpr bool sk = SK;
    const float* sk_src = sk ? pt.sk_d + 8 * ps : nullptr;
    float4 D[2][2];
    float sk_acc[4] = {0.f, 0.f, 0.f, 0.f}, sk_sum = 0.0f, dl0 = 0.0f, dl1 = 0.0f;
    auto load_d = [&](auto cu, long long p) {
        if constexpr (sk) {
            const long long pc = p < p_end ? p : p_end - 16;
            D[decltype(cu)::value][0] = *reinterpret_cast<const float4*>(sk_src + pc);
            D[decltype(cu)::value][1] = *reinterpret_cast<const float4*>(sk_src + pc + 4);
        }
    };
    // Request row q of a k-step.  Never skipped, so that the compiler's count of requests in flight stays exact:
    // past the end the last k-step is read again and ignored.
    auto load_row = [&](auto cu, auto cq, long long p) {
        const long long pc = p < p_end ? p : p_end - 16;
        const f32x4 ld4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (pc + decltype(cq)::value) * ld));
        R[decltype(cu)::value][decltype(cq)::value] = make_float4(ld4[0], ld4[1], ld4[2], ld4[3]);
    };
    auto load_raw = [&](auto cu, long long p) {
        static_for<0, 8>([&](auto cq) { load_row(cu, cq, p); });
        load_d(cu, p);
    };
    // Splitting a k-step is 40 chunks of at most six mutually independent instructions, one chunk per MFMA:
    // a dependent instruction issued right behind its producer waits ~9 clocks and holds up the MFMA queued
    // behind it, whereas up to six independent ones issue under an MFMA for free.  Chunk 8e + 4g + level
    // is one level of the split of rows 2e, 2e+1 of columns 2g, 2g+1 (scale, f16 pair h, its residual, pair l);
    // chunks 32-39 store the two operands of the four finished columns.  After the last chunk of a row pair its
    // rows of the k-step after next are requested, so the requests spread over the k-step.
    unsigned PH[4][4], PL[4][4];   // [column][row pair], two f16 each
    float t_x0[2], t_x1[2], t_sum[2];
    auto chunk = [&](auto cu, auto ck, unsigned wr, float live, long long p_next) {
        constexpr int u = decltype(cu)::value, k = decltype(ck)::value;
        if constexpr (k < 32) {
            constexpr int e = k / 8, g = (k % 8) / 4, lv = k % 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = 2 * g + j;
                const float x0 = c == 0 ? R[u][2 * e].x : c == 1 ? R[u][2 * e].y : c == 2 ? R[u][2 * e].z : R[u][2 * e].w;
                const float x1 = c == 0 ? R[u][2 * e + 1].x : c == 1 ? R[u][2 * e + 1].y : c == 2 ? R[u][2 * e + 1].z : R[u][2 * e + 1].w;
                if constexpr (lv == 0) {
                    t_x0[j] = x0 * c1;
                    t_x1[j] = x1 * c1;
                    t_sum[j] = x0 + x1;
                } else if constexpr (lv == 1) {
                    PH[c][e] = cvt_pk_f16(t_x0[j], t_x1[j]);
                    bsum[c] = fmaf(live, t_sum[j], bsum[c]);
                } else if constexpr (lv == 2) {
                    t_x0[j] = sub_f16<0>(t_x0[j], PH[c][e]);
                    t_x1[j] = sub_f16<1>(t_x1[j], PH[c][e]);
                } else {
                    PL[c][e] = cvt_pk_f16(t_x0[j], t_x1[j]);
                    if constexpr (sk) sk_acc[c] = fmaf(x0, dl0, fmaf(x1, dl1, sk_acc[c]));
                }
            }
            if constexpr (sk && g == 0) {   // the rider's two points of this row pair, zero past the end
                const float4 dq = D[u][e / 2];
                if constexpr (lv == 1) {
                    dl0 = live * (e % 2 == 0 ? dq.x : dq.z);
                    dl1 = live * (e % 2 == 0 ? dq.y : dq.w);
                } else if constexpr (lv == 2) {
                    sk_sum += dl0 + dl1;
                }
            }
            if constexpr (k % 8 == 7) {
                load_row(cu, std::integral_constant<int, 2 * e>{}, p_next);
                load_row(cu, std::integral_constant<int, 2 * e + 1>{}, p_next);
                if constexpr (e == 3) load_d(cu, p_next);
            }
        } else {   // one store per chunk: the four waves are in step and their stores would arrive together
            constexpr int c = (k - 32) / 2, part = (k - 32) % 2;
            if constexpr (part == 0) lds_write_u4<c * 16>(wr, u32x4{PH[c][0], PH[c][1], PH[c][2], PH[c][3]});
            else lds_write_u4<c * 16 + 1024>(wr, u32x4{PL[c][0], PL[c][1], PL[c][2], PL[c][3]});
        }
    };
    f16x8 as[2][NTW][2], bs[2][GROUP][2];
    auto read_a1 = [&](unsigned buf, auto ce, f16x8 (&o)[NTW][2]) {
        constexpr int e = decltype(ce)::value;
        o[e / 2][e % 2] = lds_read_h8<(e / 2) * PK_TILE_BYTES + (e % 2) * 1024>(buf + rd_a_off);
    };
    auto read_b1 = [&](unsigned buf, auto cc, auto ce, f16x8 (&o)[GROUP][2]) {
        constexpr int k0 = GROUP * decltype(cc)::value, e = decltype(ce)::value;
        o[e / 2][e % 2] = lds_read_h8<(NT + k0 + e / 2) * PK_TILE_BYTES + (e % 2) * 1024>(buf + rd_b_off);
    };
    auto read_a = [&](unsigned buf, f16x8 (&o)[NTW][2]) {
        static_for<0, NTW * 2>([&](auto ce) { read_a1(buf, ce, o); });
    };
    auto read_b = [&](unsigned buf, auto cc, f16x8 (&o)[GROUP][2]) {
        static_for<0, GROUP * 2>([&](auto ce) { read_b1(buf, cc, ce, o); });
    };

    // prologue: k-steps 0 and 1 into buffers 0 and 1, k-steps 2 and 3 requested
    load_raw(std::integral_constant<int, 0>{}, p_begin);
    load_raw(std::integral_constant<int, 1>{}, p_begin + 16);
    static_for<0, PK_CHUNKS>([&](auto ck) { chunk(std::integral_constant<int, 0>{}, ck, lds0 + wr_off, 1.0f, p_begin + 32); });
    static_for<0, PK_CHUNKS>([&](auto ck) {
        chunk(std::integral_constant<int, 1>{}, ck, lds0 + PK_BUF_BYTES + wr_off, 1.0f, p_begin + 48);
    });
    lgkm_wait();
    __syncthreads();
    read_a(lds0, as[0]);
    read_b(lds0, std::integral_constant<int, 0>{}, bs[0]);
    int b = 0;
#pragma unroll 1
    for (long long p = p_begin; p < p_end; p += 32) {
        static_for<0, 2>([&](auto cu) {
            constexpr int u = decltype(cu)::value;
            const unsigned cur = lds0 + b * PK_BUF_BYTES, nxt = lds0 + wrap3(b + 1) * PK_BUF_BYTES;
            const unsigned wr = lds0 + wrap3(b + 2) * PK_BUF_BYTES + wr_off;
            const float live = p + (u + 2) * 16 < p_end ? 1.0f : 0.0f;   // is there a k-step to store
            static_for<0, J>([&](auto cc) {
                constexpr int c = decltype(cc)::value, par = (u * J + c) % 2, k0 = GROUP * c;
                __builtin_amdgcn_sched_barrier(0);
                lgkm_wait();
#pragma unroll
                for (int g = 0; g < GROUP; ++g)
#pragma unroll
                    for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(bs[par][g][e]));
                if constexpr (c == 0) {
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                        for (int e = 0; e < 2; ++e) asm volatile("" : "+v"(as[u][nt][e]));
                }
                __builtin_amdgcn_sched_barrier(0);
                // MFMAs of item c, smallest terms first: l * h, h * l, h * h
                static_for<0, NM>([&](auto cm) {
                    constexpr int m = decltype(cm)::value, term = m / (NTW * GROUP), nt = (m / GROUP) % NTW, g = m % GROUP;
                    constexpr int ai = term == 0 ? 1 : 0;
                    constexpr int bi = term == 1 ? 1 : 0;
                    acc[nt][k0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[u][nt][ai], bs[par][g][bi], acc[nt][k0 + g], 0, 0, 0);
                    asm volatile("" : "+a"(acc[nt][k0 + g]));
                    // The next item's operands, one request per MFMA (the four waves run in step: bursts from all of
                    // them at once queue up in the LDS); after the last item the next k-step's dY tiles and first
                    // item, whose buffer was complete a barrier ago.
                    if constexpr (c + 1 < J) {
                        if constexpr (m < GROUP * 2) read_b1(cur, std::integral_constant<int, c + 1>{}, cm, bs[par ^ 1]);
                    } else {
                        if constexpr (m < NTW * 2) read_a1(nxt, cm, as[u ^ 1]);
                        else if constexpr (m < NTW * 2 + GROUP * 2)
                            read_b1(nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, m - NTW * 2>{}, bs[par ^ 1]);
                    }
                    // position within the k-step: its share of the chunks, which end PSLOTS MFMAs in so that
                    // the last stores have landed when the k-step's barrier comes
                    constexpr int step = c * NM + m, PSLOTS = J * NM - (J * NM) / PTAIL;
                    constexpr int k_lo = step < PSLOTS ? (step * PK_CHUNKS + PSLOTS - 1) / PSLOTS : PK_CHUNKS;
                    constexpr int k_hi = step + 1 < PSLOTS ? ((step + 1) * PK_CHUNKS + PSLOTS - 1) / PSLOTS : PK_CHUNKS;
                    static_for<k_lo, k_hi>([&](auto ck) { chunk(cu, ck, wr, live, p + (u + 4) * 16); });
                    __builtin_amdgcn_sched_barrier(0);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            lgkm_wait();
            __syncthreads();
            b = wrap3(b + 1);
        });
    }
    float zero[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) zero[nt] = 0.0f;
    wgrad_finish<NTW, KT, KT2>(pt, acc, zero, n0, i, s, false, scY.inv * scX.inv, scY.inv * scX2.inv);   // the scales come off
    if (pt.dbias && T < NT) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (32 * T + 4 * pj + c < pt.n_valid) atomicAdd(pt.dbias + 32 * T + 4 * pj + c, bsum[c]);
    }
    if (sk && pt.sk_d && T >= NT && T < NT + KT) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (32 * (T - NT) + 4 * pj + c < pt.k_valid) atomicAdd(pt.sk_dW + 32 * (T - NT) + 4 * pj + c, sk_acc[c]);
        if (pt.sk_db && T == NT && pj == 0) atomicAdd(pt.sk_db, sk_sum);
    }
}

// shape ids (NTW,KT,KT2): 0 = <2,8,0>  1 = <1,8,1>  2 = <2,2,0>, all on split-f16 MFMAs (a 320-register <2,8,2> tile set
// does not allocate without spills).  The hash-grid model's small layers form their weight gradients inside its
// backward kernel (mlp_ngp.hip).
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParts parts) {
    __shared__ __attribute__((aligned(16))) float lds[WG_LDS_FLOATS];
    int pi = 0;
#pragma unroll 1
    while (pi + 1 < parts.n && (int)blockIdx.x >= parts.part[pi + 1].first_block) ++pi;
    const WgradPart& pt = parts.part[pi];
    const int slab = blockIdx.x - pt.first_block;
    // slab boundaries in units of 32 points
    const long long tiles = parts.P32 / 32;
    const long long t0 = tiles * slab / pt.slabs, t1 = tiles * (slab + 1) / pt.slabs;
    if (t0 >= t1) return;   // uniform for the whole workgroup
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    switch (pt.shape) {
        case 0:
            if (pt.sk_d) wgrad_part_pk<2, 8, 0, 2, true>(pt, t0 * 32, t1 * 32, wave, lane, lds);
            else wgrad_part_pk<2, 8, 0, 2>(pt, t0 * 32, t1 * 32, wave, lane, lds);
            break;
        case 1: wgrad_part_pk<1, 8, 1, 3>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;
        default: wgrad_part_pk<2, 2, 0, 2>(pt, t0 * 32, t1 * 32, wave, lane, lds); break;   // 2
    }
}

// Relative time per point of a workgroup of each shape (from sweeps of the ratios at 786,432 points: 3.62 ms with
// 32:24:17 for :22, 3.21 ms with 24:19:22; the 64-wide shape then moved from a per-wave split to the shared one:
// 3.16 ms with :17): the cost model of the slab split.
static const int kShapeCost[3] = {24, 19, 17};

int launch_wgrad(WgradParts& parts, long long P32, hipStream_t stream) {
    // A workgroup takes a whole CU (512 registers per lane), so the grid is sized to exactly two equal-cost rounds of
    // the 256 CUs, slabs proportional to cost.
    long long total_cost = 0;
    for (int i = 0; i < parts.n; ++i) {
        if (parts.part[i].shape < 0 || parts.part[i].shape > 2) return fail(NERF_EINVAL, "wgrad: unknown shape %d", parts.part[i].shape);
        total_cost += kShapeCost[parts.part[i].shape];
    }
    const long long tiles = P32 / 32;
    const int target_blocks = 512;
    int used = 0;
    for (int i = 0; i < parts.n; ++i) {
        long long s = target_blocks * (long long)kShapeCost[parts.part[i].shape] / total_cost;
        if (s < 1) s = 1;
        if (s > tiles) s = tiles;
        parts.part[i].slabs = (int)s;
        used += (int)s;
    }
    // hand the remaining blocks to the parts with the most work per slab
    while (used < target_blocks) {
        int best = -1;
        double best_load = 0.0;
        for (int i = 0; i < parts.n; ++i) {
            const double load = (double)kShapeCost[parts.part[i].shape] / parts.part[i].slabs;
            if (parts.part[i].slabs < tiles && load > best_load) { best = i; best_load = load; }
        }
        if (best < 0) break;
        ++parts.part[best].slabs;
        ++used;
    }
    int first = 0;
    for (int i = 0; i < parts.n; ++i) {
        parts.part[i].first_block = first;
        first += parts.part[i].slabs;
    }
    parts.total_blocks = first;
    parts.P32 = P32;
    hipLaunchKernelGGL(wgrad_kernel, dim3(first), dim3(256), 0, stream, parts);
    return check_launch("wgrad");
}

// -------------------------------------------------------- small heads ----
// dW[c][k] += sum_p dY[p][c0+c] X[p][k] for heads of at most 4 outputs: bandwidth
// work (X is read once).  A block owns a point range; K/4 threads cover one row
// of X with 16-byte loads and the 4096/K row groups of the block stride through
// the range, several loads in flight each; the row groups are then summed in LDS.
// At most one workgroup of 1024 threads per CU: every workgroup ends with atomics on the same few cache
// lines of dW, and atomics on one line are executed one after the other at the memory side (~2.7 ns each) --
// with 2048 workgroups of 256 threads they, not the reads, set the time (65 us for a 67 MB read).
constexpr int SKINNY_THREADS = 1024;
template <int K>
__global__ __launch_bounds__(SKINNY_THREADS) void skinny_wgrad_kernel(SkinnyPart pt, long long P32_max) {
    constexpr int L = K / 4, R = SKINNY_THREADS / L;     // threads per row, rows per pass
    __shared__ float red[R][4][K];
    // rows up to the end of the last tile in use: the rows behind the last point hold zeros in dY
    const long long P32 = (points_in_use(P32_max, pt.count) + 31) / 32 * 32;
    const int c4 = threadIdx.x % L, rs = threadIdx.x / L;
    const long long per = (P32 + gridDim.x - 1) / gridDim.x;
    const long long p0 = per * blockIdx.x, p1 = min(P32, p0 + per);
    float4 acc[4];
    float bs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc[c] = make_float4(0.f, 0.f, 0.f, 0.f); bs[c] = 0.f; }
    const float* xp = pt.X + 4 * c4;
#pragma unroll 4
    for (long long p = p0 + rs; p < p1; p += R) {
        const float4 x = *reinterpret_cast<const float4*>(xp + p * pt.ldX);
        const float4 g4 = *reinterpret_cast<const float4*>(pt.dY + p * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < pt.C) {
                const int ch = pt.c0 + c;
                const float gc = ch == 0 ? g4.x : ch == 1 ? g4.y : ch == 2 ? g4.z : g4.w;
                acc[c].x = fmaf(gc, x.x, acc[c].x);
                acc[c].y = fmaf(gc, x.y, acc[c].y);
                acc[c].z = fmaf(gc, x.z, acc[c].z);
                acc[c].w = fmaf(gc, x.w, acc[c].w);
                bs[c] += gc;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(&red[rs][c][4 * c4]) = acc[c];
    __syncthreads();
    for (int e = threadIdx.x; e < pt.C * K; e += SKINNY_THREADS) {
        const int c = e / K, k = e % K;
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < R; ++r) sum += red[r][c][k];
        atomicAdd(pt.dW + c * K + k, sum);
    }
    if (pt.dbias) {
        __syncthreads();
        if (c4 == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) red[rs][c][0] = bs[c];
        }
        __syncthreads();
        if ((int)threadIdx.x < pt.C) {
            float sum = 0.f;
            for (int r = 0; r < R; ++r) sum += red[r][threadIdx.x][0];
            atomicAdd(pt.dbias + threadIdx.x, sum);
        }
    }
}

int launch_skinny(const SkinnyPart& pt, long long P32, hipStream_t stream) {
    NERF_REQUIRE(pt.ldY == 4 && pt.C >= 1 && pt.c0 + pt.C <= 4, "skinny_wgrad: dY must be rows of 4 floats");
    NERF_REQUIRE(pt.ldX % 4 == 0 && ((size_t)pt.X % 16) == 0 && ((size_t)pt.dY % 16) == 0, "skinny_wgrad: rows must be 16-byte aligned");
    const long long want = P32 / 256;   // at least 256 points per workgroup
    const int blocks = (int)(want < 1 ? 1 : want > 512 ? 512 : want);   // two workgroups per CU at most (64 KiB of LDS each)
    const dim3 block(SKINNY_THREADS);
    switch (pt.K) {
        case 256: hipLaunchKernelGGL(skinny_wgrad_kernel<256>, dim3(blocks), block, 0, stream, pt, P32); break;
        case 128: hipLaunchKernelGGL(skinny_wgrad_kernel<128>, dim3(blocks), block, 0, stream, pt, P32); break;
        case 64: hipLaunchKernelGGL(skinny_wgrad_kernel<64>, dim3(blocks), block, 0, stream, pt, P32); break;
        default: return fail(NERF_EINVAL, "skinny_wgrad: unsupported width");
    }
    return check_launch("skinny_wgrad");
}

}  // namespace nerf
