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
                                                                  int64_t level_stride, const int* __restrict__ count) {
    const int g = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int l = g + 8 * (i / nblk);
    if (l >= levels) return;
    const int64_t p = (int64_t)(i % nblk) * 256 + threadIdx.x;
    if (p >= points_in_use(P, count)) return;
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
// The gradient of feature f of level l of point p: d_out is row-major [P][2 * levels] (level_stride 0) or
// level-major [levels][level_stride][2].
__device__ __forceinline__ int64_t denc_index(int64_t p, int l, int levels, int64_t level_stride) {
    return level_stride ? 2 * ((int64_t)l * level_stride + p) : p * (2 * levels) + 2 * l;
}

__global__ void hashgrid_bwd_kernel(const float* __restrict__ x, int64_t P, const float* __restrict__ d_out,
                                    HashGradLevels lv, int levels, int64_t level_stride, const int* __restrict__ count) {
    const int j = blockIdx.y, l = lv.id[j];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = t >> 2;
    if (p >= points_in_use(P, count)) return;
    const int f = (int)t & 1, cx = ((int)t >> 1) & 1;
    const Cell c = locate(x, p, lv.res[j]);
    const float g = d_out[denc_index(p, l, levels, level_stride) + f];
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
constexpr int HG_LDS_MAX_LEVELS = 8;
struct LdsLevels {   // the levels of one launch: blockIdx.y picks one (at 262,144 points a level by itself fills half the chip)
    float* table[HG_LDS_MAX_LEVELS];
    int res[HG_LDS_MAX_LEVELS];
    unsigned size[HG_LDS_MAX_LEVELS];
    int id[HG_LDS_MAX_LEVELS];   // position of the level in the encoding
};
__global__ __launch_bounds__(HG_LDS_THREADS) void hashgrid_bwd_lds_kernel(const float* __restrict__ x, int64_t P_max,
                                                                          const float* __restrict__ d_out, LdsLevels lv,
                                                                          int levels, int64_t level_stride,
                                                                          const int* __restrict__ count) {
    extern __shared__ double s_tab[];   // [2][size]: feature-major (two atomics on the adjacent doubles of a row collide)
    const int tid = threadIdx.x;
    const int64_t P = points_in_use(P_max, count);
    float* tab = lv.table[blockIdx.y];
    const int res = lv.res[blockIdx.y], l = lv.id[blockIdx.y];
    const unsigned size = lv.size[blockIdx.y];
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) s_tab[i] = 0.0;
    __syncthreads();
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p_begin = per * blockIdx.x, p_end = min(P, p_begin + per);
    // 256 points per round, four threads each.  Consecutive points are consecutive samples of a ray and share a
    // cell at these resolutions: taken in order, the 16 points of a wave instruction would add to the same rows
    // (same-address LDS atomics run one lane after the other).  Within a round the points are taken in the order of
    // a transposed 16 x 16 block, so that a wave holds points 16 apart.
    static_assert(HG_LDS_THREADS / 4 == 256, "the transposed order below is that of a 16 x 16 block of points");
    for (int64_t t0 = p_begin; t0 < p_end; t0 += HG_LDS_THREADS / 4) {
        const int li = tid >> 2;
        const int64_t p = t0 + ((li & 15) * 16 + (li >> 4));
        if (p >= p_end) continue;
        const int f = tid & 1, cx = (tid >> 1) & 1;
        const Cell c = locate(x, p, res);
        const float g = d_out[denc_index(p, l, levels, level_stride) + f];
        const unsigned px = cx ? c.hi[0] : c.lo[0];
        const float wx = cx ? c.w[0] : 1.0f - c.w[0];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const unsigned row = hash_row(px, (yz & 2) ? c.hi[1] : c.lo[1], (yz & 1) ? c.hi[2] : c.lo[2], size);
            const float wgt = (wx * ((yz & 2) ? c.w[1] : 1.0f - c.w[1])) * ((yz & 1) ? c.w[2] : 1.0f - c.w[2]);
            atomicAdd(s_tab + f * size + row, (double)(wgt * g));
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) {
        const float v = (float)s_tab[(i & 1) * size + (i >> 1)];
        if (v != 0.0f) atomicAdd(tab + i, v);
    }
}


// ---------------------------------------------------------------------------
// Binned scatter for the levels whose table does not fit in LDS.  A batch updates every row of a table
// many times over (4.2 M points x 8 corners into 524,288 rows: 64 updates per row), but scattered float
// atomics are executed one line at a time at the memory side (~20 G transactions/s) whoever shares the
// row.  So the updates are first sorted by table segment and then summed in LDS:
//   pass 1 (bin): a workgroup of 512 points turns every (point, level) into 4 records of 16 bytes -- one per
//     (y, z) corner pair: the row r of the low-x corner within its 8,192-row segment, the two updates of row r
//     (a, b) and the ratio rho = wx / (1 - wx) of the high-x to the low-x weight, so that row r + 1 (the hash is
//     linear in x) receives (rho a, rho b) -- counting-sorts them by segment in LDS and appends each run to that
//     segment's queue in the workspace.  A run is padded with empty records to whole 128-byte lines and the
//     queues start on line boundaries: every line of a queue is written once, whole, by one instruction (runs that
//     start anywhere cost a read-modify-write of their first and last lines at the memory side).  A segment has
//     several queues, a workgroup appending to queue (its index % nsub): the space of a run is reserved by a
//     returning atomic on the queue's counter, atomics on one cache line are executed one after the other at the
//     memory side, and with one queue per segment every workgroup of the grid queues up at the same two lines.
//   pass 2 (reduce): a workgroup zeroes a 128 KiB LDS image of one 8,192-row segment, adds a slice of its
//     queue with LDS atomics and adds the image to the table contiguously.
// The LDS image is DOUBLE: measured on this chip, ds_add_f32 is executed one lane at a time (192 cycles per
// wave instruction whatever the addresses; ds_add_rtn_f32 the same) while ds_add_f64 takes 8-26 and
// ds_add_u32 5-8.  Float sums in LDS are therefore kept as doubles, which also makes them exact enough
// that the order of the additions stops mattering.  (rho a is formed in double from the two floats: it differs
// from the float product of the high-x weight and the gradient by a few units in the last place of a float.)
// Records whose queue is full (the capacity is 1.25 x the even share plus the padding) and high-x corners that
// do not sit in the row after their partner (x wrap, segment boundary) go to the table directly.
// ---------------------------------------------------------------------------
constexpr int BIN_SEG_ROWS = 8192;    // 128 KiB of double2
constexpr int BIN_PTS = 512;          // 8 waves (the point-to-thread mapping of the bin kernel is written for that)
constexpr int BIN_REC = 4 * BIN_PTS;
constexpr int BIN_MAX_SEGS = 64;
constexpr int BIN_RUN = 8;            // records per 128-byte line
constexpr int BIN_SLOTS = BIN_REC + BIN_MAX_SEGS * (BIN_RUN - 1);   // a workgroup's records of a level, every run padded
constexpr int BIN_MIN_SUB = 8;        // queues per segment at least; a reduce workgroup takes at most this many
constexpr unsigned BIN_NO_DEST = 0xffffffffu;
constexpr long long BIN_SLICE_RECORDS = 131072, BIN_SLICE_MIN = 32768;   // records per reduce workgroup of a full segment: at most, at least
struct BinLevels {
    float* table[HG_MAX_LEVELS];
    long long q_off[HG_MAX_LEVELS];     // record offset of the level's queues
    int res[HG_MAX_LEVELS];
    unsigned size[HG_MAX_LEVELS];
    unsigned magic[HG_MAX_LEVELS];      // mod_magic_of(size)
    int id[HG_MAX_LEVELS];              // position of the level in the encoding
    int nseg[HG_MAX_LEVELS];
    int nsub[HG_MAX_LEVELS];            // queues per segment: a bin workgroup appends to queue (its index % nsub)
    int cnt_first[HG_MAX_LEVELS + 1];   // prefix sum of nseg * nsub: the level's first counter, laid out [sub][segment]
    int slices[HG_MAX_LEVELS];          // reduce workgroups per segment; workgroup r takes queues r, r + slices, ...
    int wg_first[HG_MAX_LEVELS + 1];    // prefix sum of nseg * slices: the level's first reduce workgroup
    unsigned cap[HG_MAX_LEVELS];        // records per queue, a multiple of BIN_RUN; queues laid out [segment][sub][cap]
    int n;
};

// locate() of hashgrid.h on coordinates already in registers
__device__ __forceinline__ Cell locate3(const float (&xyz)[3], int res) {
    Cell c;
    const float inv_res = 1.0f / (float)res;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pos = xyz[k] * (float)res;
        const float fl = floorf(pos);
        c.w[k] = pos - fl;
        if (fabsf(fl) < 16777216.0f) {
            c.lo[k] = wrap_small(fl, res, inv_res);
            c.hi[k] = c.lo[k] + 1 == (unsigned)res ? 0u : c.lo[k] + 1;
        } else {
            const long long p0 = (long long)fl;
            c.lo[k] = wrap(p0, res);
            c.hi[k] = wrap(p0 + 1, res);
        }
    }
    return c;
}

__global__ __launch_bounds__(BIN_PTS, 6) void hashgrid_bin_kernel(const float* __restrict__ x, int64_t P_max,
                                                               const float* __restrict__ d_out, int levels,
                                                               int64_t level_stride, BinLevels lv,
                                                               unsigned* __restrict__ counters, float4* __restrict__ queues,
                                                               const int* __restrict__ count) {
    const int64_t P = points_in_use(P_max, count);
    if ((int64_t)blockIdx.x * BIN_PTS >= P) return;   // the whole workgroup (the grid covers P_max)
    // The kernel is bound by latency, not by a unit (vector ALU and LDS ~30 % busy each, waves waiting 60 % of their
    // time): what counts per level is the number of barriers and what the returning atomic that reserves queue space
    // has to overlap with.  Three barriers per level: B (histogram complete) -- reserve, prefix sums, the records'
    // values -- C -- records to their sorted places -- D -- copy-out.  The histogram alternates between two arrays so
    // that the next level's counting needs no barrier behind the copy-out: the other array was cleared between B and C.
    __shared__ unsigned s_hist[2][BIN_MAX_SEGS], s_start[BIN_MAX_SEGS + 1], s_gbase[BIN_MAX_SEGS];
    __shared__ unsigned s_dest[BIN_SLOTS];
    __shared__ float4 s_rec[BIN_SLOTS];
    const int tid = threadIdx.x;
    // Which of the workgroup's 512 points a thread takes: consecutive points are consecutive samples of a ray and,
    // on the coarser levels, fall into the same cell -- their records would sit side by side in a queue and the
    // reduce pass would add them to the same row in one instruction, the slowest case of an LDS atomic.  Wave w
    // takes the points w, w + 8, w + 16, ... of each of eight runs of 64: neighbours end up in different waves,
    // whose records enter the queues at different times.
    // The coordinates are loaded coalesced (thread t loads point t) and handed to their owners through the record
    // buffer, idle at that point; the gradient of a level is loaded by the owner itself, a level ahead (8 bytes per
    // lane at a stride of 64: every line of the workgroup's 4 KiB is touched by all eight waves, which the L1 absorbs).
    static_assert(BIN_PTS == 512, "the mapping below permutes 8 waves x 64 lanes");
    const int pt_in_wg = ((tid & 63) >> 3) * 64 + (tid >> 6) + 8 * (tid & 7);
    const int64_t p = (int64_t)blockIdx.x * BIN_PTS + pt_in_wg;
    const bool valid = p < P;
    const int64_t p_own = valid ? p : P - 1;
    const int64_t p_load = min((int64_t)blockIdx.x * BIN_PTS + tid, P - 1);   // the point whose coordinates this thread loads
    float* s_xch = reinterpret_cast<float*>(s_rec);
#pragma unroll
    for (int k = 0; k < 3; ++k) s_xch[3 * tid + k] = x[3 * p_load + k];
    float2 g_next = *reinterpret_cast<const float2*>(d_out + denc_index(p_own, lv.id[0], levels, level_stride));
    if (tid < 2 * BIN_MAX_SEGS) (&s_hist[0][0])[tid] = 0u;
    __syncthreads();
    const float xyz[3] = {s_xch[3 * pt_in_wg], s_xch[3 * pt_in_wg + 1], s_xch[3 * pt_in_wg + 2]};
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < lv.n; ++j) {
        const unsigned size = lv.size[j], magic = lv.magic[j], cap = lv.cap[j];
        const int nseg = lv.nseg[j], nsub = lv.nsub[j], sub = blockIdx.x % nsub;
        float* __restrict__ tab = lv.table[j];
        unsigned* hist = s_hist[j & 1];
        const float2 g = g_next;   // requested an iteration ago
        if (j + 1 < lv.n) g_next = *reinterpret_cast<const float2*>(d_out + denc_index(p_own, lv.id[j + 1], levels, level_stride));
        const Cell c = locate3(xyz, lv.res[j]);
        unsigned seg[4], pos[4], loc[4];
        bool pair[4];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const unsigned cy = (yz & 2) ? c.hi[1] : c.lo[1], cz = (yz & 1) ? c.hi[2] : c.lo[2];
            // the hash is linear in x: unless x wraps around the grid or the 32-bit sum around zero, the high-x
            // corner's sum is the low-x one's plus one, and so is its remainder
            const unsigned h_yz = cy * 2654435761u + cz * 805459861u, h_lo = c.lo[0] + h_yz;
            const unsigned r_lo = row_of_sum(h_lo, size, magic);
            unsigned r_hi = r_lo + 1 == size ? 0u : r_lo + 1;
            if (c.hi[0] != c.lo[0] + 1 || h_lo == 0xffffffffu) r_hi = row_of_sum(c.hi[0] + h_yz, size, magic);
            seg[yz] = r_lo / BIN_SEG_ROWS;
            loc[yz] = r_lo % BIN_SEG_ROWS;
            // w = pos - floor(pos) is a fractional part, but for pos in [-2^-25, 0) it rounds to exactly 1: rho = inf
            // there, and such a corner pair never takes the (a, rho a) record form
            pair[yz] = r_hi == r_lo + 1 && loc[yz] != BIN_SEG_ROWS - 1 && c.w[0] < 1.0f;
            if (valid && !pair[yz]) {
                const float wy = (yz & 2) ? c.w[1] : 1.0f - c.w[1], wz = (yz & 1) ? c.w[2] : 1.0f - c.w[2];
                const float w_hi = (c.w[0] * wy) * wz;
                atomicAdd(tab + 2 * (size_t)r_hi, w_hi * g.x);
                atomicAdd(tab + 2 * (size_t)r_hi + 1, w_hi * g.y);
            }
            pos[yz] = valid ? atomicAdd(&hist[seg[yz]], 1u) : 0u;
        }
        __syncthreads();   // B
        if (tid < nseg) {
            const unsigned cnt = hist[tid], padded = (cnt + BIN_RUN - 1) & ~(unsigned)(BIN_RUN - 1);
            s_gbase[tid] = cnt ? atomicAdd(&counters[lv.cnt_first[j] + sub * nseg + tid], padded) : 0u;
        }
        if (tid >= BIN_MAX_SEGS && tid < 2 * BIN_MAX_SEGS) s_hist[(j + 1) & 1][tid - BIN_MAX_SEGS] = 0u;   // last read a level ago
        if (tid >= 2 * BIN_MAX_SEGS && tid < 3 * BIN_MAX_SEGS) {   // another wave than the one reserving queue space:
            const int k = tid - 2 * BIN_MAX_SEGS;                   // prefix sums of the padded run lengths, one lane per segment
            const unsigned len = k < nseg ? (hist[k] + BIN_RUN - 1) & ~(unsigned)(BIN_RUN - 1) : 0u;
            unsigned incl = len;
#pragma unroll
            for (int d = 1; d < 64; d *= 2) {
                const unsigned up = __shfl_up(incl, d, 64);
                if (k >= d) incl += up;
            }
            s_start[k] = incl - len;
            if (k == BIN_MAX_SEGS - 1) s_start[BIN_MAX_SEGS] = incl;
        }
        // the records' values, while the reservation is on its way
        const float rho = c.w[0] / (1.0f - c.w[0]);
        float va[4], vb[4], vr[4];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const float wy = (yz & 2) ? c.w[1] : 1.0f - c.w[1], wz = (yz & 1) ? c.w[2] : 1.0f - c.w[2];
            const float w_lo = ((1.0f - c.w[0]) * wy) * wz;
            va[yz] = w_lo * g.x;
            vb[yz] = w_lo * g.y;
            vr[yz] = pair[yz] ? rho : 0.0f;
        }
        __syncthreads();   // C
        if (valid) {
#pragma unroll
            for (int yz = 0; yz < 4; ++yz) {
                const unsigned i = s_start[seg[yz]] + pos[yz], gq = s_gbase[seg[yz]] + pos[yz];
                if (gq < cap) {
                    s_dest[i] = (seg[yz] * nsub + sub) * cap + gq;
                    s_rec[i] = make_float4(__uint_as_float(loc[yz]), va[yz], vb[yz], vr[yz]);
                } else {   // the segment's queue is full
                    s_dest[i] = BIN_NO_DEST;
                    float* row = tab + 2 * ((size_t)seg[yz] * BIN_SEG_ROWS + loc[yz]);
                    if (va[yz] != 0.0f) atomicAdd(row, va[yz]);
                    if (vb[yz] != 0.0f) atomicAdd(row + 1, vb[yz]);
                    if (vr[yz] != 0.0f) {   // a pair: the high-x corner's update, formed as on the direct path
                        const float wy = (yz & 2) ? c.w[1] : 1.0f - c.w[1], wz = (yz & 1) ? c.w[2] : 1.0f - c.w[2];
                        const float w_hi = (c.w[0] * wy) * wz;
                        atomicAdd(row + 2, w_hi * g.x);
                        atomicAdd(row + 3, w_hi * g.y);
                    }
                }
            }
        }
        if (tid < nseg) {   // empty records up to the end of the run's last line (row BIN_SEG_ROWS is the image's spare row)
            const unsigned cnt = hist[tid], padded = (cnt + BIN_RUN - 1) & ~(unsigned)(BIN_RUN - 1);
            for (unsigned k = cnt; k < padded; ++k) {
                const unsigned gq = s_gbase[tid] + k;
                s_dest[s_start[tid] + k] = gq < cap ? (tid * nsub + sub) * cap + gq : BIN_NO_DEST;
                s_rec[s_start[tid] + k] = make_float4(__uint_as_float((unsigned)BIN_SEG_ROWS), 0.0f, 0.0f, 0.0f);
            }
        }
        __syncthreads();   // D
        const unsigned n_rec = s_start[BIN_MAX_SEGS];
        float4* __restrict__ q = queues + lv.q_off[j];
        for (unsigned i = tid; i < n_rec; i += BIN_PTS) {
            const unsigned d = s_dest[i];
            if (d != BIN_NO_DEST) q[d] = s_rec[i];
        }
        // no barrier here: whatever the next level writes before its barrier B (the other histogram) is not read by
        // this copy-out, and nothing passes B before every wave has finished it
    }
}

constexpr int RED_THREADS = 1024;
constexpr unsigned RED_F1 = BIN_SEG_ROWS + 2;   // doubles from a row's feature 0 to its feature 1 in the LDS image
__global__ __launch_bounds__(RED_THREADS) void hashgrid_reduce_kernel(BinLevels lv, const unsigned* __restrict__ counters,
                                                                      const float4* __restrict__ queues) {
    // The image is feature-major, [2][BIN_SEG_ROWS + 2] (a record's second row may be the one past the segment): two
    // atomics on the adjacent doubles of an interleaved row cost 69 ticks where two on independent doubles cost 52
    // (tools/microbench/lds_atomic.hip).
    extern __shared__ double s_tab[];
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    int j = 0;
#pragma unroll 1
    while (j + 1 < lv.n && (int)blockIdx.x >= lv.wg_first[j + 1]) ++j;
    const int w = blockIdx.x - lv.wg_first[j], slices = lv.slices[j], nseg = lv.nseg[j], nsub = lv.nsub[j];
    const int sg = w / slices, first_sub = w - sg * slices;
    const unsigned cap = lv.cap[j];
    // This workgroup's queues (at most BIN_MIN_SUB of them) as one run of records: queue m starts at pre[m].
    const unsigned* __restrict__ cnt_of = counters + lv.cnt_first[j] + sg;   // [sub * nseg]
    unsigned pre[BIN_MIN_SUB + 1];
    pre[0] = 0u;
#pragma unroll
    for (int m = 0; m < BIN_MIN_SUB; ++m) {
        const int sub = first_sub + m * slices;
        unsigned have = sub < nsub ? cnt_of[sub * nseg] : 0u;
        have = have < cap ? have : cap;
        pre[m + 1] = pre[m] + have;
    }
    const unsigned total = pre[BIN_MIN_SUB];
    if (total == 0) return;   // the whole workgroup
    const unsigned row0 = (unsigned)sg * BIN_SEG_ROWS;
    const unsigned rows = min((unsigned)BIN_SEG_ROWS, lv.size[j] - row0);
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < 2 * RED_F1; i += RED_THREADS) s_tab[i] = 0.0;
    __syncthreads();
    const f32x4n* __restrict__ q = reinterpret_cast<const f32x4n*>(queues + lv.q_off[j] + ((size_t)sg * nsub + first_sub) * cap);
    const size_t q_stride = (size_t)slices * cap;   // from one of this workgroup's queues to the next
    // Eight records per thread and round, all loads first: one workgroup per CU (the LDS image) has only its own
    // loads to cover the memory latency with.  (What bounds this kernel is the LDS double atomics, about 1.25
    // lanes per clock at random rows: loading a round ahead of the atomics changed nothing.)  Padding records and
    // lanes past the end skip theirs.
    constexpr int UNR = 8;
    for (unsigned k0 = tid; k0 < total; k0 += UNR * RED_THREADS) {
        f32x4n r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned k = k0 + u * RED_THREADS, kc = k < total ? k : k0;
            unsigned m = 0, start = 0;
#pragma unroll
            for (int t = 1; t < BIN_MIN_SUB; ++t)
                if (kc >= pre[t]) {
                    m = t;
                    start = pre[t];
                }
            r[u] = __builtin_nontemporal_load(q + m * q_stride + (kc - start));
            if (k >= total) r[u][0] = __uint_as_float((unsigned)BIN_SEG_ROWS);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned loc = __float_as_uint(r[u][0]);
            if (loc < BIN_SEG_ROWS) {
                double* row = s_tab + loc;   // feature 0 of the row; feature 1 is RED_F1 doubles on
                const double a = (double)r[u][1], b = (double)r[u][2], rho = (double)r[u][3];
                atomicAdd(row, a);
                atomicAdd(row + RED_F1, b);
                atomicAdd(row + 1, a * rho);
                atomicAdd(row + RED_F1 + 1, b * rho);
            }
        }
    }
    __syncthreads();
    float* __restrict__ tab = lv.table[j] + 2 * (size_t)row0;
    for (unsigned i = tid; i < 2 * rows; i += RED_THREADS) {
        const float v = (float)s_tab[(i & 1) * RED_F1 + (i >> 1)];
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
                                 int levels, float* out, bool level_major, int64_t level_stride, hipStream_t stream,
                                 const int* count) {
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
                           (int)nblk, out, level_stride, count);
    else
        hipLaunchKernelGGL(hashgrid_fwd_pinned_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, x, P, lv, levels,
                           (int)nblk, out, level_stride, count);
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
constexpr size_t BIN_COUNTER_BYTES = 65536;  // up to 16,384 queue counters in front of the queues

static bool level_is_binned(int size) { return (unsigned)size > HG_LDS_MAX_ROWS; }
// How a level's updates are queued.  A segment's share of the 4 P records is its share of the rows (rows are hit
// uniformly; a table of 10,648 rows has a segment of 8,192 and one of 2,456).  slices = reduce workgroups per
// segment (about BIN_SLICE_RECORDS records of a full segment each), nsub = queues per segment (at least 8), cap =
// records per queue: 1.25 x its share, the padding of every run appended to it, a floor; a multiple of BIN_RUN so
// that every queue starts on a line boundary.
struct BinPlan {
    int nseg, nsub, slices;
    unsigned cap;
};
static BinPlan bin_plan(int64_t P, int size) {
    BinPlan pl;
    pl.nseg = (int)ceil_div(size, BIN_SEG_ROWS);
    const int64_t seg_rows = size < BIN_SEG_ROWS ? size : BIN_SEG_ROWS;
    const int64_t share = ceil_div(4 * P * seg_rows, (int64_t)size);
    // A reduce workgroup's time is its records' (4 LDS atomics each) plus a fixed zero-and-flush of its image, and the
    // launch takes as long as its longest workgroup: a slice is about 1/512 of the records of a 14-level batch (two
    // rounds of the 256 CUs) within [BIN_SLICE_MIN, BIN_SLICE_RECORDS].  (At 262,144 points slices of 131,072 records
    // left the few-segment levels to 10-14 workgroups that ran three times as long as all the others.)
    // (swept at 262,144 and 1,048,576 points: P/9 with a floor of 32,768 and P/4 with 65,536 tie; floors of 16,384
    // and 8,192 lose 1-4 % to the images' zero-and-flush)
    int64_t slice = P / 9;
    slice = slice < BIN_SLICE_MIN ? BIN_SLICE_MIN : slice > BIN_SLICE_RECORDS ? BIN_SLICE_RECORDS : slice;
    pl.slices = (int)ceil_div(share, slice);
    pl.nsub = pl.slices > BIN_MIN_SUB ? pl.slices : BIN_MIN_SUB;
    const int64_t wgs = ceil_div(ceil_div(P, BIN_PTS), pl.nsub);           // bin workgroups per queue
    const int64_t sub_share = ceil_div(share, pl.nsub), pads = (BIN_RUN - 1) * wgs, most = 4 * wgs * BIN_PTS + pads;
    int64_t cap = sub_share + sub_share / 4 + pads + 1024;
    if (cap > most) cap = most;
    pl.cap = (unsigned)ceil_div(cap, BIN_RUN) * BIN_RUN;
    return pl;
}

int nerf::hashgrid_scatter_counter_words() { return (int)(BIN_COUNTER_BYTES / 4); }

size_t nerf::hashgrid_scatter_workspace_bytes(int64_t P, const int* sizes, int levels) {
    if (P < BIN_MIN_POINTS || 5 * P >= (1ll << 31)) return 0;
    size_t records = 0;
    for (int l = 0; l < levels; ++l)
        if (level_is_binned(sizes[l])) {
            const BinPlan pl = bin_plan(P, sizes[l]);
            records += (size_t)pl.nseg * pl.nsub * pl.cap;
        }
    return records ? BIN_COUNTER_BYTES + 16 * records : 0;
}

int nerf::launch_hashgrid_scatter(const float* x, int64_t P, const float* d_enc, int64_t level_stride, const int* res,
                                  const int* sizes, int levels, float* const* d_tables, void* workspace,
                                  size_t workspace_bytes, hipStream_t stream, const int* count, bool counters_cleared) {
    static std::atomic<bool> lds_enabled[64], red_enabled[64];
    if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hashgrid_bwd_lds_kernel), lds_enabled,
                                 (int)(HG_LDS_MAX_ROWS * 2 * sizeof(double)))) return rc;
    const size_t need = hashgrid_scatter_workspace_bytes(P, sizes, levels);
    const bool binned = workspace && need;
    if (binned && workspace_bytes < need)
        return fail(NERF_EINVAL, "hashgrid_scatter: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    if (binned && (size_t)workspace % 16)
        return fail(NERF_EINVAL, "hashgrid_scatter: the workspace must be 16-byte aligned (128 for whole-line writes)");
    HashGradLevels lv;
    LdsLevels ll;
    int n_lds = 0;
    unsigned lds_rows = 0;
    auto flush_lds_levels = [&]() -> int {   // one launch for the LDS-resident levels collected so far
        if (n_lds == 0) return NERF_OK;
        const int blocks = (int)(P / 2048 < 256 ? P / 2048 : 256);
        hipLaunchKernelGGL(hashgrid_bwd_lds_kernel, dim3(blocks, n_lds), dim3(HG_LDS_THREADS), (size_t)lds_rows * 2 * sizeof(double),
                           stream, x, P, d_enc, ll, levels, level_stride, count);
        n_lds = 0;
        lds_rows = 0;
        return check_launch("hashgrid_scatter(lds)");
    };
    BinLevels bl;
    bl.n = 0;
    bl.cnt_first[0] = bl.wg_first[0] = 0;
    long long q_off = 0;
    int big = 0;
    for (int l = 0; l < levels; ++l) {
        if (res[l] < 1 || sizes[l] < 1) return fail(NERF_EINVAL, "hashgrid: level %d has res=%d size=%d", l, res[l], sizes[l]);
        if ((unsigned)sizes[l] <= HG_LDS_MAX_ROWS && P >= 65536) {
            // enough points per workgroup to pay for adding the whole table once per workgroup
            if (n_lds == HG_LDS_MAX_LEVELS)
                if (int rc = flush_lds_levels()) return rc;
            ll.table[n_lds] = d_tables[l];
            ll.res[n_lds] = res[l];
            ll.size[n_lds] = (unsigned)sizes[l];
            ll.id[n_lds] = l;
            lds_rows = lds_rows > (unsigned)sizes[l] ? lds_rows : (unsigned)sizes[l];
            ++n_lds;
        } else if (binned && level_is_binned(sizes[l])) {
            const BinPlan pl = bin_plan(P, sizes[l]);
            const int j = bl.n++;
            if (pl.nseg > BIN_MAX_SEGS) return fail(NERF_EINVAL, "hashgrid_scatter: level %d has more than %d segments", l, BIN_MAX_SEGS);
            bl.table[j] = d_tables[l];
            bl.res[j] = res[l];
            bl.size[j] = (unsigned)sizes[l];
            bl.magic[j] = mod_magic_of((unsigned)sizes[l]);
            bl.id[j] = l;
            bl.nseg[j] = pl.nseg;
            bl.nsub[j] = pl.nsub;
            bl.slices[j] = pl.slices;
            bl.cap[j] = pl.cap;
            bl.q_off[j] = q_off;
            q_off += (long long)pl.nseg * pl.nsub * pl.cap;
            bl.cnt_first[j + 1] = bl.cnt_first[j] + pl.nseg * pl.nsub;
            bl.wg_first[j + 1] = bl.wg_first[j] + pl.nseg * pl.slices;
        } else {
            lv.table[big] = d_tables[l];
            lv.res[big] = res[l];
            lv.size[big] = (unsigned)sizes[l];
            lv.id[big] = l;
            ++big;
        }
    }
    if (int rc = flush_lds_levels()) return rc;
    if (bl.n) {
        if ((size_t)bl.cnt_first[bl.n] * 4 > BIN_COUNTER_BYTES) return fail(NERF_EINVAL, "hashgrid_scatter: too many queues");
        if (int rc = raise_lds_limit(reinterpret_cast<const void*>(hashgrid_reduce_kernel), red_enabled,
                                     (int)((BIN_SEG_ROWS + 2) * 2 * sizeof(double)))) return rc;
        unsigned* counters = reinterpret_cast<unsigned*>(workspace);
        float4* queues = reinterpret_cast<float4*>(reinterpret_cast<char*>(workspace) + BIN_COUNTER_BYTES);
        if (!counters_cleared) {
            launch_zero_words(counters, (int)(BIN_COUNTER_BYTES / 4), stream);
            if (int rc = check_launch("hashgrid_scatter(clear)")) return rc;
        }
        hipLaunchKernelGGL(hashgrid_bin_kernel, dim3((unsigned)ceil_div(P, BIN_PTS)), dim3(BIN_PTS), 0, stream, x, P, d_enc,
                           levels, level_stride, bl, counters, queues, count);
        if (int rc = check_launch("hashgrid_scatter(bin)")) return rc;
        hipLaunchKernelGGL(hashgrid_reduce_kernel, dim3(bl.wg_first[bl.n]), dim3(RED_THREADS),
                           (size_t)(BIN_SEG_ROWS + 2) * 2 * sizeof(double), stream, bl, counters, queues);
        if (int rc = check_launch("hashgrid_scatter(reduce)")) return rc;
    }
    if (big == 0) return NERF_OK;
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3((unsigned)ceil_div(4 * P, 256), big), dim3(256), 0, stream, x, P, d_enc, lv,
                       levels, level_stride, count);
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
    return launch_hashgrid_scatter(x, P, d_out, 0, res, sizes, levels, d_tables, workspace, workspace_bytes, (hipStream_t)stream);
}
