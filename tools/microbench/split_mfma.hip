// Accuracy of split-operand MFMA products (32x32 tile, K = 256) against float64, and whether f16 subnormal inputs
// pass through the MFMA:   hipcc -O2 --offload-arch=gfx950 -o split_mfma split_mfma.hip && ./split_mfma
// Result that shaped the MLP kernels: two f16 pieces per operand and three MFMAs (residuals kept in range by
// powers of two) are as accurate as three bf16 pieces and six MFMAs, and as an fp32 fma chain:
// max|err| / sum|terms| 0.5-1.5e-7 against 1.3-2.7e-7 and 1.3-3.7e-7; subnormal f16 inputs are not flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split_bf(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
    for (int i = 0; i < 8; ++i) {
        __bf16 hh = (__bf16)v[i]; float r1 = v[i] - (float)hh; __bf16 mm = (__bf16)r1;
        h[i] = hh; m[i] = mm; l[i] = (__bf16)(r1 - (float)mm);
    }
}
__device__ inline void split_h(const float* v, float scale, f16x8& h, f16x8& l) {
    for (int i = 0; i < 8; ++i) {
        const float x = v[i] * scale;
        _Float16 hh = (_Float16)x; float r = (x - (float)hh) * 2048.0f;
        h[i] = hh; l[i] = (_Float16)r;
    }
}
// A [32][K] row-major, B [K][32] row-major, out C [32][32]; mode 0 bf16x6, 1 f16 two accumulators, 2 f16 single accumulator (third A piece)
__global__ void k(const float* A, const float* B, float* C, int K, int mode, float sa, float sb) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc = {0}, corr = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        for (int e = 0; e < 8; ++e) { a[e] = A[i * K + k0 + 8 * h + e]; b[e] = B[(k0 + 8 * h + e) * 32 + i]; }
        if (mode == 0) {
            bf16x8 ah, am, al, bh, bm, bl;
            split_bf(a, ah, am, al); split_bf(b, bh, bm, bl);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        } else if (mode == 1) {
            f16x8 ah, al, bh, bl;
            split_h(a, sa, ah, al); split_h(b, sb, bh, bl);
            corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, corr, 0, 0, 0);
            corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, corr, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        } else {
            f16x8 ah, al, bh, bl, ah2, al2;
            split_h(a, sa, ah, al); split_h(b, sb, bh, bl);
            for (int e = 0; e < 8; ++e) { ah2[e] = (_Float16)((float)ah[e] * (1.0f / 2048.0f)); al2[e] = (_Float16)((float)al[e] * (1.0f / 2048.0f)); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al2, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah2, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[r];
        if (mode == 1) v = fmaf(corr[r], 1.0f / 2048.0f, v);
        C[row * 32 + i] = v * inv;
    }
}
__global__ void denorm_probe(float* out) {
    const int l = threadIdx.x;
    f16x8 a, b;
    const _Float16 tiny = (_Float16)(ldexpf(1.0f, -20));   // subnormal in f16
    for (int e = 0; e < 8; ++e) { a[e] = tiny; b[e] = (_Float16)1.0f; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    f32x16 acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
    if (l == 0) { out[0] = acc[0]; out[1] = acc2[0]; out[2] = (float)tiny; }
}
int main() {
    float* dout; hipMalloc(&dout, 16);
    denorm_probe<<<1, 64>>>(dout);
    float ho[3]; hipMemcpy(ho, dout, 12, hipMemcpyDeviceToHost);
    printf("f16 subnormal through MFMA: A-side %g  B-side %g  expected %g (tiny=%g)\n", ho[0], ho[1], 16 * ldexp(1.0, -20), ho[2]);
    std::mt19937 g(1);
    const int K = 256;
    struct Case { const char* name; float amag, bmag; int kind; float sa, sb; };
    const Case cases[] = {
        {"weights U(-1/16,1/16) x relu acts N(0,1)", 1.f / 16, 1.f, 0, 4096.f, 1.f},
        {"weights U(-1/16,1/16) x sin/cos", 1.f / 16, 1.f, 1, 4096.f, 1.f},
        {"weights x grads ~1e-7 scaled by 2^24", 1.f / 16, 1e-7f, 0, 4096.f, 16777216.f},
        {"weights x grads log-uniform 1e-12..1e-6 scaled 2^24", 1.f / 16, 1e-6f, 2, 4096.f, 16777216.f},
        {"weights x acts log-uniform 1e-6..1e2", 1.f / 16, 1e2f, 2, 4096.f, 1.f},
        {"unscaled weights (sa=1) x acts N(0,1)", 1.f / 16, 1.f, 0, 1.f, 1.f},
    };
    for (const Case& c : cases) {
        std::vector<float> A(32 * K), B(K * 32);
        std::uniform_real_distribution<float> U(-1.f, 1.f);
        std::normal_distribution<float> N(0.f, 1.f);
        for (auto& v : A) v = c.amag * U(g);
        for (auto& v : B) {
            if (c.kind == 0) v = c.bmag * std::max(0.f, N(g));
            else if (c.kind == 1) v = sinf(100.f * U(g));
            else v = c.bmag * powf(10.f, -6.f * (0.5f + 0.5f * U(g))) * (U(g) > 0 ? 1.f : -1.f);
        }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s\n", c.name);
        for (int mode = 0; mode < 3; ++mode) {
            k<<<1, 64>>>(dA, dB, dC, K, mode, mode == 0 ? 1.f : c.sa, mode == 0 ? 1.f : c.sb);
            std::vector<float> C(32 * 32);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, worst32 = 0;
            for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
                double s = 0, sabs = 0; float s32 = 0;
                for (int kk = 0; kk < K; ++kk) { double t = (double)A[i * K + kk] * B[kk * 32 + j]; s += t; sabs += fabs(t); s32 = fmaf(A[i * K + kk], B[kk * 32 + j], s32); }
                if (sabs > 0) { worst = std::max(worst, fabs(C[i * 32 + j] - s) / sabs); worst32 = std::max(worst32, fabs((double)s32 - s) / sabs); }
            }
            printf("   %-28s max|err|/sum|terms| = %.3g   (f32 fma chain: %.3g)\n", mode == 0 ? "bf16 x 6" : mode == 1 ? "f16 x 3, two accumulators" : "f16 x 3, one accumulator", worst, worst32);
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
