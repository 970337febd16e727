// Input encoders for gfx950: sin/cos positional encoding and the
// multiresolution hash grid (forward gather + trilinear interpolation, backward
// scatter with float atomics).  Compiled with -ffp-contract=off so the
// interpolation rounds like the reference's separate PyTorch ops.
#include <atomic>

#include "common.h"
#include "hashgrid.h"

namespace nerf {

// ------------------------------------------------------------------ PE ----
// out[p][0:D] = x; out[p][D + 2*D*f + D*s + c] = (s ? cos : sin)(2^f * x[p][c])
// (src/utils/positional_encoding.py:30-36).  One thread per (point, coordinate) computes
// the whole column family of its coordinate (sincos_octaves; bands 20 and up, where angle
// doubling in double has lost too many bits, by library calls), the workgroup's rows are
// assembled in LDS and leave fully coalesced.
constexpr int PE_THREADS = 256;
constexpr int PE_CHAIN = 20;
template <int LC>   // bands computed by the octave recurrence = min(L, LC)
__global__ __launch_bounds__(PE_THREADS) void encode_pe_kernel(const float* __restrict__ x, int64_t P, int D, int L,
                                                               float* __restrict__ out) {
    extern __shared__ float s_rows[];   // [points per workgroup][width]
    const int width = D * (2 * L + 1);
    const int ppb = PE_THREADS / D;                    // points per workgroup
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int here = (int)min((int64_t)ppb, P - p0);
    const int tid = threadIdx.x;
    if (tid < here * D) {
        const int pl = tid / D, c = tid - pl * D;
        const float xv = x[p0 * D + tid];
        float* row = s_rows + pl * width;
        row[c] = xv;
        float sn[LC], cs[LC];
        sincos_octaves<LC>(xv, sn, cs);
#pragma unroll
        for (int f = 0; f < LC; ++f) {
            if (f < L) {
                row[D + 2 * D * f + c] = sn[f];
                row[D + 2 * D * f + D + c] = cs[f];
            }
        }
        for (int f = LC; f < L; ++f) {
            const float arg = ldexpf(xv, f);  // 2^f * x, exact like the reference's fp32 product
            row[D + 2 * D * f + c] = sinf(arg);
            row[D + 2 * D * f + D + c] = cosf(arg);
        }
    }
    __syncthreads();
    const int n = here * width;
    float* dst = out + p0 * width;
    for (int i = tid; i < n; i += PE_THREADS) dst[i] = s_rows[i];
}

// ----------------------------------------------------------- hash grid ----
// grid.y = a group of HG_FWD_GROUP consecutive levels, one thread per point: a workgroup gathers from a few
// tables, so the small ones stay in L2 and the large ones in the Infinity Cache, and a thread writes its
// group's features as one 32-byte piece of the point's row (8 bytes per level written one level at a time
// cost four times the bytes at the memory side: a 32-byte sector is the unit of a partial write).
constexpr int HG_FWD_GROUP = 4;
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(const float* __restrict__ x, int64_t P, HashLevels lv, int levels,
                                                           float* __restrict__ out, int32_t* __restrict__ idx_dump,
                                                           float* __restrict__ w_dump) {
    const int l0 = blockIdx.y * HG_FWD_GROUP;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float2 r[HG_FWD_GROUP];
#pragma unroll
    for (int k = 0; k < HG_FWD_GROUP; ++k) {
        const int l = l0 + k;
        if (l >= levels) break;
        const Cell c = locate(x, p, lv.res[l]);
        const float2* __restrict__ tab = reinterpret_cast<const float2*>(lv.table[l]);
        float2 f[8];
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {  // 000,001,...,111 = (x,y,z) offsets
            const unsigned row = corner_row(c, corner, lv.size[l]);
            f[corner] = tab[row];
            if (idx_dump) idx_dump[((int64_t)l * 8 + corner) * P + p] = (int32_t)row;
        }
        if (w_dump) {
#pragma unroll
            for (int j = 0; j < 3; ++j) w_dump[((int64_t)l * P + p) * 3 + j] = c.w[j];
        }
        r[k] = trilerp(f, c.w);   // src/utils/hash_grid_encoder.py:95-103
    }
    float* dst = out + p * (2 * levels) + 2 * l0;
    if (l0 + HG_FWD_GROUP <= levels && (2 * levels) % 8 == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(r[0].x, r[0].y, r[1].x, r[1].y);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(r[2].x, r[2].y, r[3].x, r[3].y);
    } else {
#pragma unroll
        for (int k = 0; k < HG_FWD_GROUP; ++k)
            if (l0 + k < levels) *reinterpret_cast<float2*>(dst + 2 * k) = r[k];
    }
}

// The same gather with every level pinned to one XCD: workgroups whose linear id is congruent mod 8 share an
// XCD and its 4 MiB L2, and a 2^19-row table is 4 MiB, so XCD g takes levels g, g + 8, ... one after the other
// and its L2 holds the table being gathered from.  Output level-major, out_t[level][P][2]: consecutive lanes
// write consecutive 8-byte features.
template <bool LEVEL_MAJOR>
__global__ __launch_bounds__(256) void hashgrid_fwd_pinned_kernel(const float* __restrict__ x, int64_t P, HashLevels lv,
                                                                  int levels, int nblk, float* __restrict__ out,
                                                                  int64_t level_stride) {
    const int g = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int l = g + 8 * (i / nblk);
    if (l >= levels) return;
    const int64_t p = (int64_t)(i % nblk) * 256 + threadIdx.x;
    if (p >= P) return;
    const Cell c = locate(x, p, lv.res[l]);
    float2 f[8];
    gather_corners(lv.table[l], c, lv.size[l], f);
    const float2 r = trilerp(f, c.w);   // src/utils/hash_grid_encoder.py:95-103
    if (LEVEL_MAJOR) *reinterpret_cast<float2*>(out + 2 * ((int64_t)l * level_stride + p)) = r;
    else *reinterpret_cast<float2*>(out + p * (2 * levels) + 2 * l) = r;
}

// d table[row(corner)][f] += weight(corner) * d_out[f], weight = product of (w or 1-w).
// Scattered float atomics are executed line by line at the memory side, about 20 G lines/s
// however few lanes of an instruction share a line.  Four lanes share a point: lane bit 0 is
// the feature, bit 1 the x offset of the corner, so the (up to) four adds that fall into one
// 16-byte stretch of the table (both features of rows r and r+1: the hash is linear in x)
// travel in one instruction and, unless the row wraps, in one line.
__global__ void hashgrid_bwd_kernel(const float* __restrict__ x, int64_t P, const float* __restrict__ d_out,
                                    HashGradLevels lv, int levels) {
    const int j = blockIdx.y, l = lv.id[j];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t >> 2;
    if (p >= P) return;
    const int f = (int)t & 1, cx = ((int)t >> 1) & 1;
    const Cell c = locate(x, p, lv.res[j]);
    const float g = d_out[p * (2 * levels) + 2 * l + f];
    float* __restrict__ tab = lv.table[j] + f;
    const unsigned size = lv.size[j];
    const unsigned px = cx ? c.hi[0] : c.lo[0];
    const float wx = cx ? c.w[0] : 1.0f - c.w[0];
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
        const unsigned row = hash_row(px, (yz & 2) ? c.hi[1] : c.lo[1], (yz & 1) ? c.hi[2] : c.lo[2], size);
        const float wgt = (wx * ((yz & 2) ? c.w[1] : 1.0f - c.w[1])) * ((yz & 1) ? c.w[2] : 1.0f - c.w[2]);
        atomicAdd(tab + 2 * (size_t)row, wgt * g);
    }
}

// The same for a level whose whole gradient table fits in LDS (the coarsest levels): each
// workgroup adds its share of the points into an LDS copy and then adds the copy to the
// table once, contiguously (contiguous atomics run ~16 times faster than scattered ones).
// The copy is double: LDS float atomics are executed lane by lane on this chip, double ones are not.
constexpr int HG_LDS_THREADS = 1024;
constexpr unsigned HG_LDS_MAX_ROWS = 8192;   // 128 KiB of LDS
__global__ __launch_bounds__(HG_LDS_THREADS) void hashgrid_bwd_lds_kernel(const float* __restrict__ x, int64_t P,
                                                                          const float* __restrict__ d_out, float* tab,
                                                                          int res, unsigned size, int l, int levels) {
    extern __shared__ double s_tab[];   // [size][2]
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) s_tab[i] = 0.0;
    __syncthreads();
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p_begin = per * blockIdx.x, p_end = min(P, p_begin + per);
    for (int64_t t = 4 * p_begin + tid; t < 4 * p_end; t += HG_LDS_THREADS) {
        const int64_t p = t >> 2;
        const int f = (int)t & 1, cx = ((int)t >> 1) & 1;
        const Cell c = locate(x, p, res);
        const float g = d_out[p * (2 * levels) + 2 * l + f];
        const unsigned px = cx ? c.hi[0] : c.lo[0];
        const float wx = cx ? c.w[0] : 1.0f - c.w[0];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const unsigned row = hash_row(px, (yz & 2) ? c.hi[1] : c.lo[1], (yz & 1) ? c.hi[2] : c.lo[2], size);
            const float wgt = (wx * ((yz & 2) ? c.w[1] : 1.0f - c.w[1])) * ((yz & 1) ? c.w[2] : 1.0f - c.w[2]);
            atomicAdd(s_tab + 2 * row + f, (double)(wgt * g));
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) {
        const float v = (float)s_tab[i];
        if (v != 0.0f) atomicAdd(tab + i, v);
    }
}


// ---------------------------------------------------------------------------
// Binned scatter for the levels whose table does not fit in LDS.  A batch updates every row of a table
// many times over (4.2 M points x 8 corners into 524,288 rows: 64 updates per row), but scattered float
// atomics are executed one line at a time at the memory side (~20 G transactions/s) whoever shares the
// row.  So the updates are first sorted by table segment and then summed in LDS:
//   pass 1 (bin): a workgroup of 512 points turns every (point, level) into 4 records -- one per (y, z)
//     corner pair: row of the low-x corner within its 16,384-row segment + the four values for rows r and
//     r+1 (the hash is linear in x) -- counting-sorts them by segment in LDS and appends each run to that
//     segment's queue in the workspace (one reservation atomic per segment, coalesced stores);
//   pass 2 (reduce): a workgroup zeroes a 128 KiB LDS image of one 8,192-row segment, adds a slice of its
//     queue with LDS atomics and adds the image to the table contiguously.
// The LDS image is DOUBLE: measured on this chip, ds_add_f32 is executed one lane at a time (192 cycles per
// wave instruction whatever the addresses; ds_add_rtn_f32 the same) while ds_add_f64 takes 8-26 and
// ds_add_u32 5-8.  Float sums in LDS are therefore kept as doubles, which also makes them exact enough
// that the order of the additions stops mattering.
// Records whose queue is full (the capacity is 1.25 x the even share) and high-x corners that do not sit
// in the row after their partner (x wrap, segment boundary) go to the table directly.
// ---------------------------------------------------------------------------
constexpr int BIN_SEG_ROWS = 8192;    // 128 KiB of double2
constexpr int BIN_PTS = 512;
constexpr int BIN_REC = 4 * BIN_PTS;
constexpr int BIN_MAX_SEGS = 64;
constexpr unsigned BIN_NO_DEST = 0xffffffffu;
constexpr long long BIN_SLICE_RECORDS = 131072;
struct BinLevels {
    float* table[HG_MAX_LEVELS];
    long long q_off[HG_MAX_LEVELS];     // dword offset of the level's queues: 5 component arrays of nseg * cap dwords
    int res[HG_MAX_LEVELS];
    unsigned size[HG_MAX_LEVELS];
    int id[HG_MAX_LEVELS];              // position of the level in the encoding
    int nseg[HG_MAX_LEVELS];
    int seg_first[HG_MAX_LEVELS + 1];   // prefix sum of nseg: index of the level's first counter
    int slices[HG_MAX_LEVELS];
    int wg_first[HG_MAX_LEVELS + 1];    // prefix sum of nseg * slices: the level's first reduce workgroup
    unsigned cap[HG_MAX_LEVELS];        // records per segment queue
    int n;
};

__global__ __launch_bounds__(BIN_PTS) void hashgrid_bin_kernel(const float* __restrict__ x, int64_t P,
                                                               const float* __restrict__ d_out, int levels, BinLevels lv,
                                                               unsigned* __restrict__ counters, float* __restrict__ queues) {
    __shared__ unsigned s_hist[BIN_MAX_SEGS], s_start[BIN_MAX_SEGS + 1], s_gbase[BIN_MAX_SEGS];
    __shared__ unsigned s_dest[BIN_REC];
    __shared__ float s_rec[5][BIN_REC];
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * BIN_PTS + tid;
    const bool valid = p < P;
    const int64_t pl = valid ? p : P - 1;
#pragma unroll 1
    for (int j = 0; j < lv.n; ++j) {
        const unsigned size = lv.size[j], cap = lv.cap[j];
        const int nseg = lv.nseg[j];
        float* __restrict__ tab = lv.table[j];
        if (tid < BIN_MAX_SEGS) s_hist[tid] = 0u;
        __syncthreads();
        const Cell c = locate(x, pl, lv.res[j]);
        const float2 g = *reinterpret_cast<const float2*>(d_out + pl * (2 * levels) + 2 * lv.id[j]);
        unsigned seg[4], pos[4], loc[4];
        float v[4][4];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const unsigned cy = (yz & 2) ? c.hi[1] : c.lo[1], cz = (yz & 1) ? c.hi[2] : c.lo[2];
            const float wy = (yz & 2) ? c.w[1] : 1.0f - c.w[1], wz = (yz & 1) ? c.w[2] : 1.0f - c.w[2];
            const unsigned r_lo = hash_row(c.lo[0], cy, cz, size), r_hi = hash_row(c.hi[0], cy, cz, size);
            const float w_lo = ((1.0f - c.w[0]) * wy) * wz, w_hi = (c.w[0] * wy) * wz;
            seg[yz] = r_lo / BIN_SEG_ROWS;
            loc[yz] = r_lo % BIN_SEG_ROWS;
            const bool pair = r_hi == r_lo + 1 && loc[yz] != BIN_SEG_ROWS - 1;
            v[yz][0] = w_lo * g.x;
            v[yz][1] = w_lo * g.y;
            v[yz][2] = pair ? w_hi * g.x : 0.0f;
            v[yz][3] = pair ? w_hi * g.y : 0.0f;
            if (valid && !pair) {
                atomicAdd(tab + 2 * (size_t)r_hi, w_hi * g.x);
                atomicAdd(tab + 2 * (size_t)r_hi + 1, w_hi * g.y);
            }
            pos[yz] = valid ? atomicAdd(&s_hist[seg[yz]], 1u) : 0u;
        }
        __syncthreads();
        if (tid < nseg) {
            const unsigned cnt = s_hist[tid];
            s_gbase[tid] = cnt ? atomicAdd(&counters[lv.seg_first[j] + tid], cnt) : 0u;
        }
        if (tid == 2 * BIN_MAX_SEGS) {   // another wave than those reserving queue space
            unsigned run = 0u;
            for (int k = 0; k < nseg; ++k) {
                s_start[k] = run;
                run += s_hist[k];
            }
            s_start[BIN_MAX_SEGS] = run;
        }
        __syncthreads();
        if (valid) {
#pragma unroll
            for (int yz = 0; yz < 4; ++yz) {
                const unsigned i = s_start[seg[yz]] + pos[yz], gq = s_gbase[seg[yz]] + pos[yz];
                if (gq < cap) {
                    s_dest[i] = seg[yz] * cap + gq;
                    s_rec[0][i] = __uint_as_float(loc[yz]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) s_rec[1 + k][i] = v[yz][k];
                } else {   // the segment's queue is full
                    s_dest[i] = BIN_NO_DEST;
                    float* row = tab + 2 * ((size_t)seg[yz] * BIN_SEG_ROWS + loc[yz]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (v[yz][k] != 0.0f) atomicAdd(row + k, v[yz][k]);
                }
            }
        }
        __syncthreads();
        const unsigned n_rec = s_start[BIN_MAX_SEGS];
        float* __restrict__ q = queues + lv.q_off[j];
        const size_t comp = (size_t)nseg * cap;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            for (unsigned i = tid; i < n_rec; i += BIN_PTS) {
                const unsigned d = s_dest[i];
                if (d != BIN_NO_DEST) q[k * comp + d] = s_rec[k][i];
            }
        __syncthreads();
    }
}

constexpr int RED_THREADS = 1024;
__global__ __launch_bounds__(RED_THREADS) void hashgrid_reduce_kernel(BinLevels lv, const unsigned* __restrict__ counters,
                                                                      const float* __restrict__ queues) {
    extern __shared__ double s_tab[];   // [BIN_SEG_ROWS + 2][2]: a record's second row may be the one past the segment
    int j = 0;
#pragma unroll 1
    while (j + 1 < lv.n && (int)blockIdx.x >= lv.wg_first[j + 1]) ++j;
    const int w = blockIdx.x - lv.wg_first[j], slices = lv.slices[j];
    const int sg = w / slices, slice = w - sg * slices;
    const unsigned cap = lv.cap[j];
    const unsigned have = counters[lv.seg_first[j] + sg];
    const unsigned long long cnt = have < cap ? have : cap;
    const unsigned lo = (unsigned)(cnt * slice / slices), hi = (unsigned)(cnt * (slice + 1) / slices);
    if (lo >= hi) return;   // the whole workgroup
    const unsigned row0 = (unsigned)sg * BIN_SEG_ROWS;
    const unsigned rows = min((unsigned)BIN_SEG_ROWS, lv.size[j] - row0);
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < 2 * rows + 2; i += RED_THREADS) s_tab[i] = 0.0;
    __syncthreads();
    const size_t comp = (size_t)lv.nseg[j] * cap;
    const float* __restrict__ q = queues + lv.q_off[j] + (size_t)sg * cap;
    // Eight records per thread and round, all forty loads first: one workgroup per CU (the LDS image) has only its
    // own loads to cover the memory latency with.
    constexpr int UNR = 8;
    for (unsigned i0 = lo + tid; i0 < hi; i0 += UNR * RED_THREADS) {
        unsigned loc[UNR];
        float v[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned i = i0 + u * RED_THREADS;
            const bool ok = i < hi;
            const unsigned ic = ok ? i : i0;
            loc[u] = ok ? __float_as_uint(__builtin_nontemporal_load(q + ic)) : BIN_SEG_ROWS;   // the spare row pair
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = __builtin_nontemporal_load(q + (k + 1) * comp + ic);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            double* row = s_tab + 2 * loc[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(row + k, (double)v[u][k]);
        }
    }
    __syncthreads();
    float* __restrict__ tab = lv.table[j] + 2 * (size_t)row0;
    for (unsigned i = tid; i < 2 * rows; i += RED_THREADS) {
        const float v = (float)s_tab[i];
        if (v != 0.0f) atomicAdd(tab + i, v);
    }
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_encode_pe(const float* x, int64_t P, int D, int L, float* out, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && out, "nerf_encode_pe: null pointer");
    NERF_REQUIRE(P >= 0 && D >= 1 && L >= 0 && L <= 32, "nerf_encode_pe: bad size P=%lld D=%d L=%d", (long long)P, D, L);
    NERF_REQUIRE(D <= PE_THREADS, "nerf_encode_pe: D=%d exceeds %d", D, PE_THREADS);
    const int ppb = PE_THREADS / D;
    const size_t lds = (size_t)ppb * D * (2 * L + 1) * sizeof(float);   // <= 256 * 65 floats
    const dim3 grid((unsigned)ceil_div(P, ppb)), block(PE_THREADS);
    if (L <= 4) hipLaunchKernelGGL(encode_pe_kernel<4>, grid, block, lds, (hipStream_t)stream, x, P, D, L, out);
    else if (L <= 10) hipLaunchKernelGGL(encode_pe_kernel<10>, grid, block, lds, (hipStream_t)stream, x, P, D, L, out);
    else hipLaunchKernelGGL(encode_pe_kernel<PE_CHAIN>, grid, block, lds, (hipStream_t)stream, x, P, D, L, out);
    return check_launch("nerf_encode_pe");
}

static int fill_levels(const int* res, const int* sizes, int levels, int* res_out, unsigned* size_out) {
    for (int l = 0; l < levels; ++l) {
        if (res[l] < 1 || sizes[l] < 1) return fail(NERF_EINVAL, "hashgrid: level %d has res=%d size=%d", l, res[l], sizes[l]);
        res_out[l] = res[l];
        size_out[l] = (unsigned)sizes[l];
    }
    return NERF_OK;
}

int nerf::launch_hashgrid_gather(const float* x, int64_t P, const float* const* tables, const int* res, const int* sizes,
                                 int levels, float* out, bool level_major, int64_t level_stride, hipStream_t stream) {
    HashLevels lv;
    if (int rc = fill_levels(res, sizes, levels, lv.res, lv.size)) return rc;
    for (int l = 0; l < levels; ++l) {
        if (!tables[l]) return fail(NERF_EINVAL, "hashgrid: table %d is null", l);
        lv.table[l] = tables[l];
    }
    const int64_t nblk = ceil_div(P, 256), grid = 8 * ceil_div(levels, 8) * nblk;
    if (grid >= (1ll << 31)) return fail(NERF_EINVAL, "hashgrid: %lld points are too many for one launch", (long long)P);
    if (level_major)
        hipLaunchKernelGGL(hashgrid_fwd_pinned_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, x, P, lv, levels,
                           (int)nblk, out, level_stride);
    else
        hipLaunchKernelGGL(hashgrid_fwd_pinned_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, x, P, lv, levels,
                           (int)nblk, out, level_stride);
    return check_launch("hashgrid_gather");
}

extern "C" int nerf_hashgrid_fwd(const float* x, int64_t P, const float* const* tables, const int* res,
                                 const int* sizes, int levels, float* out, int32_t* idx_dump, float* w_dump,
                                 void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && tables && res && sizes && out, "nerf_hashgrid_fwd: null pointer");
    NERF_REQUIRE(levels >= 1 && levels <= HG_MAX_LEVELS, "nerf_hashgrid_fwd: levels=%d outside [1,%d]", levels, HG_MAX_LEVELS);
    NERF_REQUIRE(P >= 0, "nerf_hashgrid_fwd: P=%lld", (long long)P);
    HashLevels lv;
    if (int rc = fill_levels(res, sizes, levels, lv.res, lv.size)) return rc;
    for (int l = 0; l < levels; ++l) {
        NERF_REQUIRE(tables[l], "nerf_hashgrid_fwd: table %d is null", l);
        lv.table[l] = tables[l];
    }
    if (!idx_dump && !w_dump)   // every level on one XCD; the grouped kernel below serves the parity dumps
        return launch_hashgrid_gather(x, P, tables, res, sizes, levels, out, false, 0, (hipStream_t)stream);
    NERF_REQUIRE((size_t)out % 16 == 0, "nerf_hashgrid_fwd: out must be 16-byte aligned");
    hipLaunchKernelGGL(hashgrid_fwd_kernel, dim3((unsigned)ceil_div(P, 256), (unsigned)ceil_div(levels, HG_FWD_GROUP)),
                       dim3(256), 0, (hipStream_t)stream, x, P, lv, levels, out, idx_dump, w_dump);
    return check_launch("nerf_hashgrid_fwd");
}

// Raise a kernel's dynamic-LDS limit once per device (the limit belongs to (kernel, device); setting it twice
// is harmless, so two threads racing on the flag only repeat the call).
static int raise_lds_limit(const void* kernel, std::atomic<bool>* done, int bytes) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail(NERF_ELAUNCH, "hashgrid_scatter: no current device");
    if (device < 0 || device >= 64 || !done[device].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return fail(NERF_ELAUNCH, "hashgrid_scatter: cannot raise the LDS limit");
        if (device >= 0 && device < 64) done[device].store(true, std::memory_order_release);
    }
    return NERF_OK;
}

constexpr int64_t BIN_MIN_POINTS = 65536;   // below this the per-segment zero and flush cost more than the atomics saved
constexpr size_t BIN_COUNTER_BYTES = 4096;  // up to 1024 segment counters in front of the queues

static bool level_is_binned(int size) { return (unsigned)size > HG_LDS_MAX_ROWS; }
static unsigned bin_capacity(int64_t P, int nseg) {
    const int64_t even = ceil_div(4 * P, nseg);
    const int64_t cap = even + even / 4 + 8192;
    return (unsigned)(cap < 4 * P ? cap : 4 * P);
}

size_t nerf::hashgrid_scatter_workspace_bytes(int64_t P, const int* sizes, int levels) {
    if (P < BIN_MIN_POINTS || 4 * P >= (1ll << 31)) return 0;
    size_t dwords = 0;
    for (int l = 0; l < levels; ++l)
        if (level_is_binned(sizes[l])) {
            const int nseg = (int)ceil_div(sizes[l], BIN_SEG_ROWS);
            dwords += (size_t)5 * nseg * bin_capacity(P, nseg);
        }
    return dwords ? BIN_COUNTER_BYTES + 4 * dwords : 0;
}

int nerf::launch_hashgrid_scatter(const float* x, int64_t P, const float* d_enc, const int* res, const int* sizes,
                                  int levels, float* const* d_tables, void* workspace, size_t workspace_bytes,
                                  hipStream_t stream) {
    static std::atomic<bool> lds_enabled[64], red_enabled[64];
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hashgrid_bwd_lds_kernel), lds_enabled,
                                 (int)(HG_LDS_MAX_ROWS * 2 * sizeof(double)))) return rc;
    const size_t need = hashgrid_scatter_workspace_bytes(P, sizes, levels);
    const bool binned = workspace && need;
    if (binned && workspace_bytes < need)
        return fail(NERF_EINVAL, "hashgrid_scatter: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    HashGradLevels lv;
    BinLevels bl;
    bl.n = 0;
    bl.seg_first[0] = bl.wg_first[0] = 0;
    long long q_off = 0;
    int big = 0;
    for (int l = 0; l < levels; ++l) {
        if (res[l] < 1 || sizes[l] < 1) return fail(NERF_EINVAL, "hashgrid: level %d has res=%d size=%d", l, res[l], sizes[l]);
        if ((unsigned)sizes[l] <= HG_LDS_MAX_ROWS && P >= 65536) {
            // enough points per workgroup to pay for adding the whole table once per workgroup
            const int blocks = (int)(P / 2048 < 256 ? P / 2048 : 256);
            hipLaunchKernelGGL(hashgrid_bwd_lds_kernel, dim3(blocks), dim3(HG_LDS_THREADS), (size_t)sizes[l] * 2 * sizeof(double),
                               stream, x, P, d_enc, d_tables[l], res[l], (unsigned)sizes[l], l, levels);
            if (int rc = check_launch("hashgrid_scatter(lds)")) return rc;
        } else if (binned && level_is_binned(sizes[l])) {
            const int j = bl.n++, nseg = (int)ceil_div(sizes[l], BIN_SEG_ROWS);
            if (nseg > BIN_MAX_SEGS) return fail(NERF_EINVAL, "hashgrid_scatter: level %d has more than %d segments", l, BIN_MAX_SEGS);
            bl.table[j] = d_tables[l];
            bl.res[j] = res[l];
            bl.size[j] = (unsigned)sizes[l];
            bl.id[j] = l;
            bl.nseg[j] = nseg;
            bl.cap[j] = bin_capacity(P, nseg);
            bl.q_off[j] = q_off;
            q_off += (long long)5 * nseg * bl.cap[j];
            bl.slices[j] = (int)ceil_div(ceil_div(4 * P, nseg), BIN_SLICE_RECORDS);
            bl.seg_first[j + 1] = bl.seg_first[j] + nseg;
            bl.wg_first[j + 1] = bl.wg_first[j] + nseg * bl.slices[j];
        } else {
            lv.table[big] = d_tables[l];
            lv.res[big] = res[l];
            lv.size[big] = (unsigned)sizes[l];
            lv.id[big] = l;
            ++big;
        }
    }
    if (bl.n) {
        if ((size_t)bl.seg_first[bl.n] * 4 > BIN_COUNTER_BYTES) return fail(NERF_EINVAL, "hashgrid_scatter: too many segments");
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hashgrid_reduce_kernel), red_enabled,
                                     (int)((BIN_SEG_ROWS + 2) * 2 * sizeof(double)))) return rc;
        unsigned* counters = reinterpret_cast<unsigned*>(workspace);
        float* queues = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + BIN_COUNTER_BYTES);
        if (hipMemsetAsync(counters, 0, BIN_COUNTER_BYTES, stream) != hipSuccess)
            return fail(NERF_ELAUNCH, "hashgrid_scatter: cannot clear the queue counters");
        hipLaunchKernelGGL(hashgrid_bin_kernel, dim3((unsigned)ceil_div(P, BIN_PTS)), dim3(BIN_PTS), 0, stream, x, P, d_enc,
                           levels, bl, counters, queues);
        if (int rc = check_launch("hashgrid_scatter(bin)")) return rc;
        hipLaunchKernelGGL(hashgrid_reduce_kernel, dim3(bl.wg_first[bl.n]), dim3(RED_THREADS),
                           (size_t)(BIN_SEG_ROWS + 2) * 2 * sizeof(double), stream, bl, counters, queues);
        if (int rc = check_launch("hashgrid_scatter(reduce)")) return rc;
    }
    if (big == 0) return NERF_OK;
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3((unsigned)ceil_div(4 * P, 256), big), dim3(256), 0, stream, x, P, d_enc, lv,
                       levels);
    return check_launch("hashgrid_scatter");
}

extern "C" size_t nerf_hashgrid_bwd_workspace_bytes(int64_t P, const int* sizes, int levels) {
    if (!sizes || levels < 1 || levels > HG_MAX_LEVELS) return 0;
    return hashgrid_scatter_workspace_bytes(P, sizes, levels);
}

extern "C" int nerf_hashgrid_bwd(const float* x, int64_t P, const float* d_out, const int* res, const int* sizes,
                                 int levels, float* const* d_tables, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && d_out && res && sizes && d_tables, "nerf_hashgrid_bwd: null pointer");
    NERF_REQUIRE(levels >= 1 && levels <= HG_MAX_LEVELS, "nerf_hashgrid_bwd: levels=%d outside [1,%d]", levels, HG_MAX_LEVELS);
    NERF_REQUIRE(P >= 0, "nerf_hashgrid_bwd: P=%lld", (long long)P);
    for (int l = 0; l < levels; ++l) NERF_REQUIRE(d_tables[l], "nerf_hashgrid_bwd: gradient table %d is null", l);
    return launch_hashgrid_scatter(x, P, d_out, res, sizes, levels, d_tables, workspace, workspace_bytes, (hipStream_t)stream);
}
