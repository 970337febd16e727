// Device helpers of the multiresolution hash grid shared by the stand-alone
// encoder kernels and the fused hash-grid MLP (src/utils/hash_grid_encoder.py).
#pragma once
#include "common.h"

namespace nerf {

constexpr int HG_MAX_LEVELS = 32;

struct HashLevels {
    const float* table[HG_MAX_LEVELS];
    int res[HG_MAX_LEVELS];
    unsigned size[HG_MAX_LEVELS];
};
struct HashGradLevels {   // a subset of the levels; id = position of the level in the encoding
    float* table[HG_MAX_LEVELS];
    int res[HG_MAX_LEVELS];
    unsigned size[HG_MAX_LEVELS];
    int id[HG_MAX_LEVELS];
};

// Floor-modulo of a possibly negative cell coordinate (Python's % on int64,
// src/utils/hash_grid_encoder.py:68-69).
__device__ __forceinline__ unsigned wrap(long long v, int res) {
    long long m = v % res;
    if (m < 0) m += res;
    return (unsigned)m;
}

// The same for a coordinate given as an integer-valued float of magnitude below 2^24 (every
// sample within 32768 grid extents of the origin): the quotient is estimated in float, which
// can be off by one, and corrected; a 64-bit integer division costs ~100 instructions.
__device__ __forceinline__ unsigned wrap_small(float fl, int res, float inv_res) {
    const int v = (int)fl;
    int m = v - (int)floorf(fl * inv_res) * res;
    if (m < 0) m += res;
    if (m >= res) m -= res;
    return (unsigned)m;
}

// ((x*1 + y*2654435761 + z*805459861) & 0xFFFFFFFF) % size in the reference's
// int64 arithmetic equals the 32-bit wrap-around sum (src/utils/hash_grid_encoder.py:39-40).
__device__ __forceinline__ unsigned hash_row(unsigned cx, unsigned cy, unsigned cz, unsigned size) {
    const unsigned h = cx + cy * 2654435761u + cz * 805459861u;
    return (size & (size - 1)) == 0 ? h & (size - 1) : h % size;   // uniform over a level
}

// h % size for a size that is not a power of two, given magic = floor(2^32 / size): mulhi(h, magic) is the
// quotient or one less for every 32-bit h (h magic / 2^32 lies in (h / size - 1, h / size]), so one correction
// completes it -- 6 instructions where the division by a run-time value takes 15, five of them quarter-rate multiplies.
__host__ __device__ __forceinline__ unsigned mod_magic_of(unsigned size) {
    return (size & (size - 1)) ? (unsigned)((1ull << 32) / size) : 0u;   // 0: a power of two, masked instead
}
__device__ __forceinline__ unsigned row_of_sum(unsigned h, unsigned size, unsigned magic) {
    if (magic == 0u) return h & (size - 1);   // uniform over a level
    const unsigned r = h - __umulhi(h, magic) * size;
    return r >= size ? r - size : r;
}

struct Cell {
    unsigned lo[3], hi[3];
    float w[3];
};

__device__ __forceinline__ Cell locate(const float* __restrict__ x, int64_t p, int res) {
    Cell c;
    const float inv_res = 1.0f / (float)res;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float pos = x[p * 3 + k] * (float)res;   // :57
        const float fl = floorf(pos);
        c.w[k] = pos - fl;                             // :64 (before wrapping)
        if (fabsf(fl) < 16777216.0f) {
            c.lo[k] = wrap_small(fl, res, inv_res);   // :60, :68
            c.hi[k] = c.lo[k] + 1 == (unsigned)res ? 0u : c.lo[k] + 1;   // (p0 + 1) % res, :69
        } else {
            const long long p0 = (long long)fl;
            c.lo[k] = wrap(p0, res);
            c.hi[k] = wrap(p0 + 1, res);
        }
    }
    return c;
}

// Trilinear interpolation of the 8 corner rows in the reference's order: along
// x, then y, then z (src/utils/hash_grid_encoder.py:95-103), channel by channel.
__device__ __forceinline__ float2 trilerp(const float2 (&f)[8], const float (&w)[3]) {
    const float wx = w[0], wy = w[1], wz = w[2];
    float r[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const float v000 = ch ? f[0].y : f[0].x, v001 = ch ? f[1].y : f[1].x, v010 = ch ? f[2].y : f[2].x,
                    v011 = ch ? f[3].y : f[3].x, v100 = ch ? f[4].y : f[4].x, v101 = ch ? f[5].y : f[5].x,
                    v110 = ch ? f[6].y : f[6].x, v111 = ch ? f[7].y : f[7].x;
        const float x00 = v000 * (1.0f - wx) + v100 * wx;
        const float x01 = v001 * (1.0f - wx) + v101 * wx;
        const float x10 = v010 * (1.0f - wx) + v110 * wx;
        const float x11 = v011 * (1.0f - wx) + v111 * wx;
        const float y0 = x00 * (1.0f - wy) + x10 * wy;
        const float y1 = x01 * (1.0f - wy) + x11 * wy;
        r[ch] = y0 * (1.0f - wz) + y1 * wz;
    }
    return make_float2(r[0], r[1]);
}

// The eight corner rows of a cell, gathered in four 16-byte loads where that is possible: the hash is linear
// in x, so the high-x corner of a (y, z) pair normally sits in the row after the low-x one.  Where it does not
// (x wraps around the grid, or the sum wraps around the table) the high-x row is loaded by itself.
// f[corner] in the reference's order, corner bits = (x, y, z) offsets.
__device__ __forceinline__ void gather_corners(const float* __restrict__ table, const Cell& c, unsigned size, float2 (&f)[8],
                                               unsigned (*rows)[8] = nullptr) {
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(8)));
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
        const unsigned cy = (yz & 2) ? c.hi[1] : c.lo[1], cz = (yz & 1) ? c.hi[2] : c.lo[2];
        const unsigned r_lo = hash_row(c.lo[0], cy, cz, size), r_hi = hash_row(c.hi[0], cy, cz, size);
        if (rows) {
            (*rows)[yz] = r_lo;
            (*rows)[4 + yz] = r_hi;
        }
        if (r_hi == r_lo + 1) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(table + 2 * (size_t)r_lo);
            f[yz] = make_float2(v[0], v[1]);
            f[4 + yz] = make_float2(v[2], v[3]);
        } else {
            f[yz] = *reinterpret_cast<const float2*>(table + 2 * (size_t)r_lo);
            f[4 + yz] = *reinterpret_cast<const float2*>(table + 2 * (size_t)r_hi);
        }
    }
}

__device__ __forceinline__ unsigned corner_row(const Cell& c, int corner, unsigned size) {  // corner bits = (x,y,z) offsets
    return hash_row((corner & 4) ? c.hi[0] : c.lo[0], (corner & 2) ? c.hi[1] : c.lo[1], (corner & 1) ? c.hi[2] : c.lo[2], size);
}
__device__ __forceinline__ float corner_weight(const Cell& c, int corner) {
    return (((corner & 4) ? c.w[0] : 1.0f - c.w[0]) * ((corner & 2) ? c.w[1] : 1.0f - c.w[1])) *
           ((corner & 1) ? c.w[2] : 1.0f - c.w[2]);
}

// The encoding of P points: out[p][2l + f] (row-major) or out[(l * level_stride + p) * 2 + f] (level-major); every
// level is gathered by the workgroups of one XCD (defined with the encoder kernels).
// count (may be null): the number of points in use, in device memory; P is then the upper bound the grids are sized for.
int launch_hashgrid_gather(const float* x, int64_t P, const float* const* tables, const int* res, const int* sizes, int levels,
                           float* out, bool level_major, int64_t level_stride, hipStream_t stream, const int* count = nullptr);

// d_tables[l][row][f] += corner weight * d_enc[p][2l + f] over all points and corners
// (the backward of the hash-grid gather); defined with the encoder kernels.  d_enc is row-major [P][2 * levels]
// (level_stride 0) or level-major [levels][level_stride][2].  With a workspace of
// hashgrid_scatter_workspace_bytes() the updates of the large tables are binned by table segment and summed
// in LDS; without one (or when that size is 0: few points) every update is a float atomic on the table.
size_t hashgrid_scatter_workspace_bytes(int64_t P, const int* sizes, int levels);
int launch_hashgrid_scatter(const float* x, int64_t P, const float* d_enc, int64_t level_stride, const int* res,
                            const int* sizes, int levels, float* const* d_tables, void* workspace, size_t workspace_bytes,
                            hipStream_t stream, const int* count = nullptr, bool counters_cleared = false);
// The first words of a binned scatter's workspace are its queue counters; a caller that clears them itself (in a kernel
// it launches anyway, just before) says so with counters_cleared.
int hashgrid_scatter_counter_words();

}  // namespace nerf
