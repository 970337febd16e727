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
// grid.y = level, one thread per point: a workgroup gathers from one table, so
// the small tables stay in L2 and the large ones in the Infinity Cache.
__global__ void hashgrid_fwd_kernel(const float* __restrict__ x, int64_t P, HashLevels lv, int levels,
                                    float* __restrict__ out, int32_t* __restrict__ idx_dump,
                                    float* __restrict__ w_dump) {
    const int l = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const Cell c = locate(x, p, lv.res[l]);
    const float2* __restrict__ tab = reinterpret_cast<const float2*>(lv.table[l]);
    float2 f[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {  // 000,001,...,111 = (x,y,z) offsets
        const unsigned row = hash_row((corner & 4) ? c.hi[0] : c.lo[0], (corner & 2) ? c.hi[1] : c.lo[1],
                                      (corner & 1) ? c.hi[2] : c.lo[2], lv.size[l]);
        f[corner] = tab[row];
        if (idx_dump) idx_dump[((int64_t)l * 8 + corner) * P + p] = (int32_t)row;
    }
    if (w_dump) {
#pragma unroll
        for (int k = 0; k < 3; ++k) w_dump[((int64_t)l * P + p) * 3 + k] = c.w[k];
    }
    const float wx = c.w[0], wy = c.w[1], wz = c.w[2];
    float r[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const float v000 = ch ? f[0].y : f[0].x, v001 = ch ? f[1].y : f[1].x, v010 = ch ? f[2].y : f[2].x,
                    v011 = ch ? f[3].y : f[3].x, v100 = ch ? f[4].y : f[4].x, v101 = ch ? f[5].y : f[5].x,
                    v110 = ch ? f[6].y : f[6].x, v111 = ch ? f[7].y : f[7].x;
        const float x00 = v000 * (1.0f - wx) + v100 * wx;  // :95-98
        const float x01 = v001 * (1.0f - wx) + v101 * wx;
        const float x10 = v010 * (1.0f - wx) + v110 * wx;
        const float x11 = v011 * (1.0f - wx) + v111 * wx;
        const float y0 = x00 * (1.0f - wy) + x10 * wy;      // :100-101
        const float y1 = x01 * (1.0f - wy) + x11 * wy;
        r[ch] = y0 * (1.0f - wz) + y1 * wz;                 // :103
    }
    *reinterpret_cast<float2*>(out + p * (2 * levels) + 2 * l) = make_float2(r[0], r[1]);
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
constexpr int HG_LDS_THREADS = 1024;
constexpr unsigned HG_LDS_MAX_ROWS = 16384;   // 128 KiB of LDS
__global__ __launch_bounds__(HG_LDS_THREADS) void hashgrid_bwd_lds_kernel(const float* __restrict__ x, int64_t P,
                                                                          const float* __restrict__ d_out, float* tab,
                                                                          int res, unsigned size, int l, int levels) {
    extern __shared__ float s_tab[];   // [size][2]
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) s_tab[i] = 0.0f;
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
            atomicAdd(s_tab + 2 * row + f, wgt * g);
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < 2 * size; i += HG_LDS_THREADS) {
        const float v = s_tab[i];
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
    hipLaunchKernelGGL(hashgrid_fwd_kernel, dim3((unsigned)ceil_div(P, 256), levels), dim3(256), 0, (hipStream_t)stream, x,
                       P, lv, levels, out, idx_dump, w_dump);
    return check_launch("nerf_hashgrid_fwd");
}

int nerf::launch_hashgrid_scatter(const float* x, int64_t P, const float* d_enc, const int* res, const int* sizes,
                                  int levels, float* const* d_tables, hipStream_t stream) {
    // The raised dynamic-LDS limit belongs to (kernel, device): remembered per device.  Setting it twice is
    // harmless, so two threads racing on a flag only repeat the call.
    static std::atomic<bool> lds_enabled[64];
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail(NERF_ELAUNCH, "hashgrid_scatter: no current device");
    if (device < 0 || device >= 64 || !lds_enabled[device].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(hashgrid_bwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(HG_LDS_MAX_ROWS * 2 * sizeof(float))) != hipSuccess)
            return fail(NERF_ELAUNCH, "hashgrid_scatter: cannot raise the LDS limit");
        if (device >= 0 && device < 64) lds_enabled[device].store(true, std::memory_order_release);
    }
    HashGradLevels lv;
    int big = 0;
    for (int l = 0; l < levels; ++l) {
        if (res[l] < 1 || sizes[l] < 1) return fail(NERF_EINVAL, "hashgrid: level %d has res=%d size=%d", l, res[l], sizes[l]);
        if ((unsigned)sizes[l] <= HG_LDS_MAX_ROWS && P >= 65536) {
            // enough points per workgroup to pay for adding the whole table once per workgroup
            const int blocks = (int)(P / 2048 < 256 ? P / 2048 : 256);
            hipLaunchKernelGGL(hashgrid_bwd_lds_kernel, dim3(blocks), dim3(HG_LDS_THREADS), (size_t)sizes[l] * 2 * sizeof(float),
                               stream, x, P, d_enc, d_tables[l], res[l], (unsigned)sizes[l], l, levels);
            if (int rc = check_launch("hashgrid_scatter(lds)")) return rc;
        } else {
            lv.table[big] = d_tables[l];
            lv.res[big] = res[l];
            lv.size[big] = (unsigned)sizes[l];
            lv.id[big] = l;
            ++big;
        }
    }
    if (big == 0) return NERF_OK;
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3((unsigned)ceil_div(4 * P, 256), big), dim3(256), 0, stream, x, P, d_enc, lv,
                       levels);
    return check_launch("hashgrid_scatter");
}

extern "C" int nerf_hashgrid_bwd(const float* x, int64_t P, const float* d_out, const int* res, const int* sizes,
                                 int levels, float* const* d_tables, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && d_out && res && sizes && d_tables, "nerf_hashgrid_bwd: null pointer");
    NERF_REQUIRE(levels >= 1 && levels <= HG_MAX_LEVELS, "nerf_hashgrid_bwd: levels=%d outside [1,%d]", levels, HG_MAX_LEVELS);
    NERF_REQUIRE(P >= 0, "nerf_hashgrid_bwd: P=%lld", (long long)P);
    for (int l = 0; l < levels; ++l) NERF_REQUIRE(d_tables[l], "nerf_hashgrid_bwd: gradient table %d is null", l);
    return launch_hashgrid_scatter(x, P, d_out, res, sizes, levels, d_tables, (hipStream_t)stream);
}
