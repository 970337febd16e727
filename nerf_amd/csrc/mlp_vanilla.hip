// Fused 8x256 NeRF MLP for gfx950 (NeRF.forward, src/models/NeRF.py:67-109;
// FastMLPNeRF.forward, src/models/FastMLPNeRF.py:47-70).
//
// Forward: one wavefront per 32 points; positional encodings are computed in
// registers, every layer is a chain of split-f16 MFMAs (three per product, fp32-accurate) whose accumulators
// become the next layer's B operands (see mlp_common.h), weights stream from the
// packed buffer (L2-resident, 5 MB) through LDS, copied once per workgroup.  For training the activations are written
// row-major for the weight-gradient GEMMs plus one ReLU bit per neuron for the
// backward-data kernel.
// Backward: (1) a fused backward-data chain in the same register layout that
// emits d(pre-activation) of every layer, (2) one work-list launch of
// weight-gradient GEMMs contracting over points, (3) two small-head reductions.
#include "mlp_common.h"

namespace nerf {
namespace vp {  // float offsets into the packed buffer
constexpr int wfloats(int KT, int OT) { return h3_layer_floats(KT, OT); }
constexpr int T88 = wfloats(8, 8);
constexpr int F_L0 = 0;                     // 63(64) -> 256
constexpr int F_L1 = F_L0 + wfloats(2, 8);  // layers 1,2,3
constexpr int F_L4X = F_L1 + 3 * T88;       // layer 4, x_enc columns
constexpr int F_L4H = F_L4X + wfloats(2, 8);
constexpr int F_L5 = F_L4H + T88;           // layers 5,6,7
constexpr int F_FEAT = F_L5 + 3 * T88;
constexpr int F_RGB0F = F_FEAT + T88;       // rgb hidden layer, feature columns (256 -> 128)
constexpr int F_RGB0D = F_RGB0F + wfloats(8, 4);  // view-direction columns (27(32) -> 128)
constexpr int B_RGB0 = F_RGB0D + wfloats(1, 4);   // transposed: 128 -> 256 (feature columns only)
constexpr int B_FEAT = B_RGB0 + wfloats(4, 8);
constexpr int B_L7 = B_FEAT + T88;          // layers 7,6,5,4(h columns),3,2,1
constexpr int S_BIAS = B_L7 + 7 * T88;      // b0..b7
constexpr int S_BF = S_BIAS + 8 * 256;
constexpr int S_B1 = S_BF + 256;
constexpr int S_WS = S_B1 + 128;
constexpr int S_BS = S_WS + 256;
constexpr int S_W2 = S_BS + 4;
constexpr int S_B2 = S_W2 + 3 * 128;
constexpr int CONST_FLOATS = 13 * 256;       // the biases and head weights from S_BIAS on, padded to whole 1 KiB pieces:
constexpr int TOTAL = S_BIAS + CONST_FLOATS; // the kernels keep a copy in LDS
static_assert(S_B2 + 4 <= TOTAL, "constant block");
}  // namespace vp

namespace vw {  // workspace layout, floats per point (multiply by P32 for the section offset)
constexpr int XE = 0;        // [P32][64]
constexpr int DE = 64;       // [P32][32]
constexpr int H = 96;        // 8 x [P32][256]
constexpr int FEAT = H + 8 * 256;
constexpr int HR = FEAT + 256;  // [P32][128]
constexpr int MASK = HR + 128;  // [P32/32][9][64][4] u32
constexpr int AMAX = MASK + 72; // [P32/32][16] u32: largest |entry| per wavefront of each saved matrix (float bits, slots
constexpr int PER_POINT = AMAX + 4;   // below); their maxima over the wavefronts in the 16 words at float offset 2 * P32
// slots of the forward maxima: x_enc, h0..h7, feat, hr, d_enc
constexpr int SLOT_XE = 0, SLOT_H = 1, SLOT_FEAT = 9, SLOT_HR = 10, SLOT_DE = 11;
// backward workspace
constexpr int DY = 0;           // 8 x [P32][256]
constexpr int DFEAT = 8 * 256;
constexpr int DYR = DFEAT + 256;   // [P32][128]
constexpr int DOUT4 = DYR + 128;   // [P32][4] = (d sigma_pre, d rgb_pre[3])
constexpr int AMAX2 = DOUT4 + 4;   // the same for the gradient matrices: slots d y0..d y7, d feat, d hr
constexpr int DPS = AMAX2 + 4;     // [P32] d sigma_pre once more, contiguous (the weight-gradient kernel reads 8 points at a time)
constexpr int PER_POINT2 = DPS + 1;
}  // namespace vw

struct TileIdx {
    long long tile, gp, lp, P32;
    int lane, p, h, wave;
    bool valid;
};

// One wave per 32 points.  The waves of a workgroup meet at barriers inside the
// dense layers, so a wave whose tile lies beyond the last one does not leave: it
// recomputes the last tile and stores the same values to the same places.
__device__ __forceinline__ void tile_index(long long P, TileIdx& ix) {
    ix.lane = threadIdx.x & 63;
    ix.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long last = (P - 1) / 32;
    const long long tile = (long long)blockIdx.x * 4 + ix.wave;
    ix.tile = tile < last ? tile : last;
    ix.p = ix.lane & 31;
    ix.h = ix.lane >> 5;
    ix.gp = ix.tile * 32 + ix.p;
    ix.valid = ix.gp < P;
    ix.lp = ix.valid ? ix.gp : P - 1;
    ix.P32 = (P + 31) / 32 * 32;
}

// LDS of the two kernels: [weight ring][4 transposition tiles][biases and head weights][backward: ReLU bits].
// The forward kernel has room for a ring of 8 chunks: the wait for a chunk then concerns copies issued six chunks
// earlier, and the row stores of the five k-steps since may still be in flight.
constexpr int FWD_RING = 8, BWD_RING = 4;
constexpr int tiles_lds_off(int ring) { return h3_lds_floats(ring); }
constexpr int const_lds_off(int ring) { return tiles_lds_off(ring) + 4 * TILE_LDS_FLOATS; }
constexpr int FWD_LDS_FLOATS = const_lds_off(FWD_RING) + vp::CONST_FLOATS;
constexpr int MASK_LDS_OFF = const_lds_off(BWD_RING) + vp::CONST_FLOATS;   // backward: the ReLU bits of the wave's tile,
constexpr int BWD_LDS_FLOATS = MASK_LDS_OFF + 4 * 9 * 256;                 // 9 layers x 1 KiB per wave
static_assert(FWD_LDS_FLOATS * 4 <= 160 * 1024 && BWD_LDS_FLOATS * 4 <= 160 * 1024, "LDS of a workgroup");

// The biases and head weights go to LDS with the first chunks of the weight stream; complete after ws_ready().
// Returns the LDS byte address of packed offset S_BIAS.
__device__ __forceinline__ unsigned copy_consts(float* lds_consts, const float* __restrict__ pk, int wave, int lane) {
    const WeightStream w{lds_consts, 0, wave, lane};
#pragma unroll
    for (int q = 0; q < (vp::CONST_FLOATS / 256 + 3) / 4; ++q)
        if (4 * q + wave < vp::CONST_FLOATS / 256) ws_copy_piece(w, pk + vp::S_BIAS, lds_consts, q);
    return lds_addr(lds_consts);
}
__device__ __forceinline__ float lds_read1(unsigned addr) {
    float r;
    asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr));
    lgkm_wait();
    asm volatile("" : "+v"(r));
    return r;
}

template <bool SAVE>
__global__ __launch_bounds__(256) void vanilla_fwd_kernel(const float* __restrict__ x, const float* __restrict__ dirs,
                                                          long long P, const float* __restrict__ pk,
                                                          float* __restrict__ rgb, float* __restrict__ sigma,
                                                          float* __restrict__ ws, int dirs_div) {
    constexpr int RING = FWD_RING;
    __shared__ __attribute__((aligned(16))) float lds[FWD_LDS_FLOATS];
    TileIdx ix;
    tile_index(P, ix);
    const int lane = ix.lane, h = ix.h;
    WeightStream wst{lds, 0, ix.wave, lane};
    const TileBuf tb = tile_buf(lds + tiles_lds_off(RING) + ix.wave * TILE_LDS_FLOATS, lane);
    // the weight stream runs through the packed layers in the order they are used
    const unsigned consts = copy_consts(lds + const_lds_off(RING), pk, ix.wave, lane);
    auto const_at = [&](int packed_off) { return consts + 4 * (packed_off - vp::S_BIAS); };
    h3_prime<RING>(wst, pk + vp::F_L0, pk + vp::B_RGB0);
    const float x0 = x[3 * ix.lp], x1 = x[3 * ix.lp + 1], x2 = x[3 * ix.lp + 2];
    const long long dr = ix.lp / dirs_div;  // one direction per ray when dirs_div = samples per ray
    const float dir0 = dirs[3 * dr], dir1 = dirs[3 * dr + 1], dir2 = dirs[3 * dr + 2];
    uint4* mask_out = SAVE ? reinterpret_cast<uint4*>(ws + ix.P32 * vw::MASK) + (ix.tile * 9) * 64 + lane : nullptr;
    // Training: the wave's largest |entry| of every matrix it saves, from which the weight-gradient kernel derives
    // the scale of its operands (see launch_amax_reduce).
    unsigned* amax = SAVE ? reinterpret_cast<unsigned*>(ws + ix.P32 * vw::AMAX) + ix.tile * AMAX_SLOTS : nullptr;
    auto publish_max = [&](float m, int slot) {
        if (SAVE) {
            const unsigned wm = wave_max_u32(__float_as_uint(m));
            if (lane == 0) amax[slot] = wm;
        }
    };

    f32x16 xe[2];
    pe_tiles<2, 10>(xe, x0, x1, x2, h);
    const float mx = abs_max<2>(xe);
    publish_max(mx, vw::SLOT_XE);
    ws_ready();

    f32x16 hc[8], acc[8];
    // Training: the activations of layer l are the input tiles of layer l+1, which applies the ReLU to them and
    // writes them out row by row, and their ReLU bits, underneath its MFMAs (see TileSave, PRE_RELU).
    auto rows_of = [&](int section, int bits_layer) {
        TileSave sv = tile_save(tb, ws + ix.P32 * section + ix.tile * 32 * 256);
        sv.relu_bits_out = mask_out + bits_layer * 64;
        return sv;
    };
    float mh = 0.0f;   // largest activation of this lane's half of the current layer
    auto finish_layer = [&](int l, float inv) {   // bias; the ReLU is left to the next layer
        finish_tiles<8, false>(acc, inv, const_at(vp::S_BIAS + 256 * l), h);
#pragma unroll
        for (int t = 0; t < 8; ++t) hc[t] = acc[t];
        mh = pos_max<8>(hc);
        publish_max(mh, vw::SLOT_H + l);
    };
    // layer 0
    PointScale sc = point_scale(mx);
    dense_h3<RING, 2, 8, SAVE, true>(wst, xe, sc.s, acc, tile_save(tb, ws + ix.P32 * vw::XE + ix.tile * 32 * 64, 64));
    finish_layer(0, sc.inv);
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        sc = point_scale(mh);
        dense_h3<RING, 8, 8, SAVE, true, PRE_RELU>(wst, hc, sc.s, acc, rows_of(vw::H + 256 * (l - 1), l - 1));
        finish_layer(l, sc.inv);
    }
    // skip connection: the layer-4 input is [x_enc | h] (either column order, see the packing), one scale for both
    sc = point_scale(fmaxf(mx, mh));
    dense_h3<RING, 2, 8, false, true>(wst, xe, sc.s, acc);
    dense_h3<RING, 8, 8, SAVE, false, PRE_RELU>(wst, hc, sc.s, acc, rows_of(vw::H + 256 * 3, 3));
    finish_layer(4, sc.inv);
#pragma unroll 1
    for (int l = 5; l < 8; ++l) {
        sc = point_scale(mh);
        dense_h3<RING, 8, 8, SAVE, true, PRE_RELU>(wst, hc, sc.s, acc, rows_of(vw::H + 256 * (l - 1), l - 1));
        finish_layer(l, sc.inv);
    }
    // feature layer (no activation)      (NeRF.py:95); it also applies layer 7's ReLU
    sc = point_scale(mh);
    dense_h3<RING, 8, 8, SAVE, true, PRE_RELU>(wst, hc, sc.s, acc, rows_of(vw::H + 256 * 7, 7));
    finish_tiles<8, false>(acc, sc.inv, const_at(vp::S_BF), h);
    // density head: softplus(w.h + b)   (NeRF.py:90)
    const float sp = dot_rows_lds<8>(hc, const_at(vp::S_WS), h) + lds_read1(const_at(vp::S_BS));
    if (ix.valid && h == 0) sigma[ix.gp] = softplus_f(sp);
    // colour head: sigmoid(W2 relu(W1 [feat | pe(dir)] + b1) + b2)   (NeRF.py:94-97)
    f32x16 de[1];
    pe_tiles<1, 4>(de, dir0, dir1, dir2, h);
    {
        const float mf = abs_max<8>(acc), md = abs_max<1>(de);
        publish_max(mf, vw::SLOT_FEAT);
        publish_max(md, vw::SLOT_DE);
        sc = point_scale(fmaxf(mf, md));
    }
    f32x16 hr[4];
    dense_h3<RING, 8, 4, SAVE, true>(wst, acc, sc.s, hr, tile_save(tb, ws + ix.P32 * vw::FEAT + ix.tile * 32 * 256));
    dense_h3<RING, 1, 4, SAVE>(wst, de, sc.s, hr, tile_save(tb, ws + ix.P32 * vw::DE + ix.tile * 32 * 32, 32));
    finish_tiles<4, true>(hr, sc.inv, const_at(vp::S_B1), h);
    if (SAVE) {
        publish_max(abs_max<4>(hr), vw::SLOT_HR);
        save_tiles<4>(tb, ws + ix.P32 * vw::HR + ix.tile * 32 * 128, 128, hr);
        mask_out[8 * 64] = relu_bits<4>(hr);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = dot_rows_lds<4>(hr, const_at(vp::S_W2 + 128 * c), h) + lds_read1(const_at(vp::S_B2 + c));
        if (ix.valid && h == 0) rgb[3 * ix.gp + c] = sigmoid_f(v);
    }
}

// Backward-data chain.  Emits d(pre-activation) of every layer row-major into ws2, and per wavefront the largest
// |entry| of each of those matrices (as the bit pattern of a non-negative float) into amax[16 * tile + 0..9]:
// layer 0..7, feature layer, colour hidden layer.
__global__ __launch_bounds__(256) void vanilla_bwd_data_kernel(const float* __restrict__ d_rgb,
                                                               const float* __restrict__ d_sigma, long long P,
                                                               const float* __restrict__ pk,
                                                               const float* __restrict__ rgb,
                                                               const float* __restrict__ sigma,
                                                               const float* __restrict__ ws, float* __restrict__ ws2,
                                                               unsigned* __restrict__ amax) {
    constexpr int RING = BWD_RING;
    __shared__ __attribute__((aligned(16))) float lds[BWD_LDS_FLOATS];
    TileIdx ix;
    tile_index(P, ix);
    const int lane = ix.lane, h = ix.h;
    WeightStream wst{lds, 0, ix.wave, lane};
    const TileBuf tb = tile_buf(lds + tiles_lds_off(RING) + ix.wave * TILE_LDS_FLOATS, lane);
    const unsigned consts = copy_consts(lds + const_lds_off(RING), pk, ix.wave, lane);
    auto const_at = [&](int packed_off) { return consts + 4 * (packed_off - vp::S_BIAS); };
    // the ReLU bits of this wave's tile, [layer][lane] 16 bytes each, also through LDS (a load from memory in the
    // middle of the chain would wait for every weight copy in flight)
    float* mask_lds = lds + MASK_LDS_OFF + ix.wave * 9 * 256;
    {
        const uint4* mask_in = reinterpret_cast<const uint4*>(ws + ix.P32 * vw::MASK) + (ix.tile * 9) * 64 + lane;
#pragma unroll
        for (int l = 0; l < 9; ++l)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mask_in + l * 64),
                                             (__attribute__((address_space(3))) void*)(mask_lds + l * 256), 16, 0, 0);
    }
    const unsigned mask_rd = lds_addr(mask_lds) + lane * 16;
    auto bits_of = [&](int l) {
        const f32x4 v = lds_read4<0>(mask_rd + l * 1024);
        lgkm_wait();
        f32x4 w = v;
        asm volatile("" : "+v"(w));
        return make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    };
    h3_prime<RING>(wst, pk + vp::B_RGB0, pk + vp::S_BIAS);
    float dpre[3], dps = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float o = rgb[3 * ix.lp + c];
        dpre[c] = ix.valid ? d_rgb[3 * ix.lp + c] * o * (1.0f - o) : 0.0f;   // sigmoid'
    }
    if (ix.valid) dps = d_sigma[ix.lp] * (1.0f - expf(-sigma[ix.lp]));          // softplus' = 1 - exp(-softplus)
    if (h == 0) {
        *reinterpret_cast<float4*>(ws2 + ix.P32 * vw::DOUT4 + ix.gp * 4) = make_float4(dps, dpre[0], dpre[1], dpre[2]);
        ws2[ix.P32 * vw::DPS + ix.gp] = dps;
    }
    ws_ready();

    // The wave's largest |entry| of each gradient matrix (see launch_amax_reduce).
    unsigned* amax_w = amax + ix.tile * AMAX_SLOTS;
    auto publish_max = [&](float m, int slot) {
        const unsigned wm = wave_max_u32(__float_as_uint(m));
        if (lane == 0) amax_w[slot] = wm;
    };

    // d hr = W2^T d rgb_pre, masked by the ReLU of the colour hidden layer
    f32x16 dyr[4];
    {
        const unsigned w2 = const_at(vp::S_W2) + 16 * h;
        static_for<0, 4>([&](auto ct) {
            constexpr int t = decltype(ct)::value;
            f32x4 w[3][4];
            static_for<0, 3>([&](auto cc) {
                static_for<0, 4>([&](auto cq) {
                    w[cc][cq] = lds_read4<(128 * decltype(cc)::value + 32 * t + 8 * decltype(cq)::value) * 4>(w2);
                });
            });
            lgkm_wait();
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sum = 0.0f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        asm volatile("" : "+v"(w[c][q]));
                        sum = fmaf(w[c][q][e], dpre[c], sum);
                    }
                    dyr[t][4 * q + e] = sum;
                }
        });
    }

    // Each 256-wide gradient tile set is the input of the next layer down, which applies the ReLU bits to it and
    // writes it out underneath its MFMAs.  The accumulators are not rescaled in between: `carry` is the factor
    // that turns the current tiles into gradients, folded into the next layer's operand scale and row stores.
    // (The published maxima are those before the ReLU bits: upper bounds.)
    auto rows_of = [&](int section, int bits_layer, float carry) {
        TileSave sv = tile_save(tb, ws2 + ix.P32 * section + ix.tile * 32 * 256);
        if (bits_layer >= 0) sv.bits = bits_of(bits_layer);
        sv.out_scale = carry;
        return sv;
    };
    f32x16 a[8], b[8];
    float m = abs_max<4>(dyr);
    publish_max(m, 9);
    PointScale sc = point_scale(m);
    {
        TileSave sv = tile_save(tb, ws2 + ix.P32 * vw::DYR + ix.tile * 32 * 128, 128);
        sv.bits = bits_of(8);
        dense_h3<RING, 4, 8, true, true, PRE_MASK>(wst, dyr, sc.s, a, sv);   // d feat (times sc.s)
    }
    float carry = sc.inv;
    m = abs_max<8>(a) * carry;
    publish_max(m, 8);
    sc = point_scale(m);
    dense_h3<RING, 8, 8, true, true, PRE_OUTSCALE>(wst, a, sc.s * carry, b, rows_of(vw::DFEAT, -1, carry));   // d h7 through the feature layer ...
    {
        const unsigned wsig = const_at(vp::S_WS) + 16 * h;
        static_for<0, 2>([&](auto chalf) {                                 // ... plus the density head
            constexpr int t0 = 4 * decltype(chalf)::value;
            f32x4 w[4][4];
            static_for<0, 4>([&](auto ct) {
                static_for<0, 4>([&](auto cq) {
                    w[ct][cq] = lds_read4<(32 * (t0 + decltype(ct)::value) + 8 * decltype(cq)::value) * 4>(wsig);
                });
            });
            lgkm_wait();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(w[t][q]));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        b[t0 + t][4 * q + e] = fmaf(w[t][q][e], dps, b[t0 + t][4 * q + e] * sc.inv);
                }
        });
    }
    carry = 1.0f;
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {                 // d h_{l-1} = W_l^T d y_l (h columns only at l = 4)
        m = abs_max<8>(b) * carry;
        publish_max(m, l);
        sc = point_scale(m);
        dense_h3<RING, 8, 8, true, true, PRE_MASK | PRE_OUTSCALE>(wst, b, sc.s * carry, a, rows_of(vw::DY + l * 256, l, carry));
        carry = sc.inv;
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = a[t];
    }
    apply_bits<8>(b, bits_of(0));
    scale_tiles<8>(b, carry);
    publish_max(abs_max<8>(b), 0);
    save_tiles<8>(tb, ws2 + ix.P32 * vw::DY + ix.tile * 32 * 256, 256, b);
}

// The ReLU bits the forward pass saved, one byte per neuron: out[p][256 l + n] for the trunk layers l = 0..7,
// out[p][2048 + n] for the 128 neurons of the colour hidden layer.
constexpr int MASK_ROW = 8 * 256 + 128;
__global__ __launch_bounds__(256) void relu_mask_dump_kernel(const float* __restrict__ ws, long long P, long long P32,
                                                             uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * MASK_ROW) return;
    const long long gp = i / MASK_ROW;
    const int c = (int)(i - gp * MASK_ROW), l = c >> 8, n = c & 255;
    const int t = n >> 5, e = n & 31, h = (e >> 2) & 1, r = (e & 3) + 4 * (e >> 3);   // rowmap(r, h) == e
    const unsigned* words = reinterpret_cast<const unsigned*>(ws + P32 * vw::MASK) + (((gp >> 5) * 9 + l) * 64 + (gp & 31) + 32 * h) * 4;
    out[i] = (uint8_t)((words[t >> 1] >> ((t & 1) * 16 + r)) & 1u);
}

}  // namespace nerf

using namespace nerf;

extern "C" int nerf_mlp_vanilla_relu_masks(const float* ws, int64_t P, uint8_t* out, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(ws && out && P > 0, "nerf_mlp_vanilla_relu_masks: null pointer or P=%lld", (long long)P);
    const long long total = (long long)P * MASK_ROW;
    hipLaunchKernelGGL(relu_mask_dump_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, ws,
                       (long long)P, (long long)((P + 31) / 32 * 32), out);
    return check_launch("nerf_mlp_vanilla_relu_masks");
}

extern "C" size_t nerf_mlp_vanilla_pack_floats(void) { return vp::TOTAL; }
extern "C" size_t nerf_mlp_vanilla_ws_floats_per_point(void) { return vw::PER_POINT; }
extern "C" size_t nerf_mlp_vanilla_ws2_floats_per_point(void) { return vw::PER_POINT2; }

extern "C" int nerf_mlp_vanilla_pack(const float* const* params, int skip_h_first, float* packed, void* stream) {
    NERF_REQUIRE(params && packed, "nerf_mlp_vanilla_pack: null pointer");
    for (int i = 0; i < NERF_VANILLA_NPARAM; ++i) NERF_REQUIRE(params[i], "nerf_mlp_vanilla_pack: parameter %d is null", i);
    const float* const* W = params;
    auto Wl = [&](int l) { return W[2 * l]; };
    auto bl = [&](int l) { return W[2 * l + 1]; };
    const float *Ws = W[16], *bs = W[17], *Wf = W[18], *bf = W[19], *W1 = W[20], *b1 = W[21], *W2 = W[22], *b2 = W[23];
    // layer-4 column ranges: NeRF has [x_enc(63) | h(256)], FastMLPNeRF has [h(256) | x_enc(63)]
    const int xe_col = skip_h_first ? 256 : 0, h_col = skip_h_first ? 0 : 63;
    PackJobs J{};
    int n = 0;
    J.job[n++] = pack_fwd_job(Wl(0), 63, 256, vp::F_L0, 2, 8, 0, 63, 0);
    for (int l = 1; l <= 3; ++l) J.job[n++] = pack_fwd_job(Wl(l), 256, 256, vp::F_L1 + (l - 1) * vp::T88, 8, 8, 0, 256, 0);
    J.job[n++] = pack_fwd_job(Wl(4), 319, 256, vp::F_L4X, 2, 8, 0, 63, xe_col);
    J.job[n++] = pack_fwd_job(Wl(4), 319, 256, vp::F_L4H, 8, 8, 0, 256, h_col);
    for (int l = 5; l <= 7; ++l) J.job[n++] = pack_fwd_job(Wl(l), 256, 256, vp::F_L5 + (l - 5) * vp::T88, 8, 8, 0, 256, 0);
    J.job[n++] = pack_fwd_job(Wf, 256, 256, vp::F_FEAT, 8, 8, 0, 256, 0);
    J.job[n++] = pack_fwd_job(W1, 283, 128, vp::F_RGB0F, 8, 4, 0, 256, 0);
    J.job[n++] = pack_fwd_job(W1, 283, 128, vp::F_RGB0D, 1, 4, 0, 27, 256);
    J.job[n++] = pack_bwd_job(W1, 283, 128, vp::B_RGB0, 4, 8, 0, 256);
    J.job[n++] = pack_bwd_job(Wf, 256, 256, vp::B_FEAT, 8, 8, 0, 256);
    for (int l = 7; l >= 1; --l)
        J.job[n++] = pack_bwd_job(Wl(l), l == 4 ? 319 : 256, 256, vp::B_L7 + (7 - l) * vp::T88, 8, 8, l == 4 ? h_col : 0, 256);
    for (int i = 0; i < n; ++i) J.job[i].h3 = 1;   // every matrix above feeds a dense layer
    for (int l = 0; l < 8; ++l) J.job[n++] = pack_copy_job(bl(l), vp::S_BIAS + 256 * l, 256);
    J.job[n++] = pack_copy_job(bf, vp::S_BF, 256);
    J.job[n++] = pack_copy_job(b1, vp::S_B1, 128);
    J.job[n++] = pack_copy_job(Ws, vp::S_WS, 256);
    J.job[n++] = pack_copy_job(bs, vp::S_BS, 1);
    J.job[n++] = pack_copy_job(W2, vp::S_W2, 384);
    J.job[n++] = pack_copy_job(b2, vp::S_B2, 3);
    J.n = n;
    static_assert(MAX_PACK_JOBS >= 36, "pack job table too small");
    return launch_pack(J, packed, (hipStream_t)stream);
}

extern "C" int nerf_mlp_vanilla_fwd(const float* x, const float* dirs, int dirs_div, int64_t P, const float* packed,
                                    float* rgb, float* sigma, float* ws, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(x && dirs && packed && rgb && sigma, "nerf_mlp_vanilla_fwd: null pointer");
    NERF_REQUIRE(P >= 0 && dirs_div >= 1, "nerf_mlp_vanilla_fwd: P=%lld dirs_div=%d", (long long)P, dirs_div);
    const dim3 grid((unsigned)ceil_div(P, 128)), block(256);
    if (ws) {
        hipLaunchKernelGGL(vanilla_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, x, dirs, (long long)P, packed, rgb,
                           sigma, ws, dirs_div);
        if (int rc = check_launch("nerf_mlp_vanilla_fwd")) return rc;
        const long long P32 = (P + 31) / 32 * 32;
        unsigned* amax = reinterpret_cast<unsigned*>(ws + P32 * vw::AMAX);
        return launch_amax_reduce(amax, P32 / 32, amax + 2 * P32, (hipStream_t)stream);
    } else
        hipLaunchKernelGGL(vanilla_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, x, dirs, (long long)P, packed,
                           rgb, sigma, ws, dirs_div);
    return check_launch("nerf_mlp_vanilla_fwd");
}

extern "C" int nerf_mlp_vanilla_bwd(const float* d_rgb, const float* d_sigma, int64_t P, const float* packed,
                                    const float* rgb, const float* sigma, float* ws, float* ws2, int skip_h_first,
                                    float* const* grads, int phases, void* stream) {
    if (P == 0) return NERF_OK;
    NERF_REQUIRE(d_rgb && d_sigma && packed && rgb && sigma && ws && ws2 && grads, "nerf_mlp_vanilla_bwd: null pointer");
    for (int i = 0; i < NERF_VANILLA_NPARAM; ++i) NERF_REQUIRE(grads[i], "nerf_mlp_vanilla_bwd: gradient %d is null", i);
    NERF_REQUIRE(P >= 0, "nerf_mlp_vanilla_bwd: P=%lld", (long long)P);
    hipStream_t st = (hipStream_t)stream;
    if (phases & 1) {
        const long long P32 = (P + 31) / 32 * 32;
        unsigned* amax = reinterpret_cast<unsigned*>(ws2 + P32 * vw::AMAX2);
        hipLaunchKernelGGL(vanilla_bwd_data_kernel, dim3((unsigned)ceil_div(P, 128)), dim3(256), 0, st, d_rgb, d_sigma,
                           (long long)P, packed, rgb, sigma, ws, ws2, amax);
        if (int rc = check_launch("nerf_mlp_vanilla_bwd(data)")) return rc;
        if (int rc = launch_amax_reduce(amax, P32 / 32, amax + 2 * P32, st)) return rc;
    }
    if (!(phases & 2)) return NERF_OK;

    const long long P32 = (P + 31) / 32 * 32;
    auto H = [&](int l) { return ws + P32 * (vw::H + 256 * l); };
    auto DY = [&](int l) { return ws2 + P32 * (vw::DY + 256 * l); };
    const float* XE = ws + P32 * vw::XE;
    const float* DE = ws + P32 * vw::DE;
    const float* FEAT = ws + P32 * vw::FEAT;
    const float* HR = ws + P32 * vw::HR;
    const float* DFEAT = ws2 + P32 * vw::DFEAT;
    const float* DYR = ws2 + P32 * vw::DYR;
    const float* DOUT4 = ws2 + P32 * vw::DOUT4;
    const int xe_col = skip_h_first ? 256 : 0, h_col = skip_h_first ? 0 : 63;
    const unsigned* amax_f = reinterpret_cast<const unsigned*>(ws + P32 * vw::AMAX) + 2 * P32;     // forward maxima, see vw::SLOT_*
    const unsigned* amax_b = reinterpret_cast<const unsigned*>(ws2 + P32 * vw::AMAX2) + 2 * P32;   // d y0..7, d feat, d hr
    auto with_max = [&](WgradPart p, int slot_y, int slot_x, int slot_x2 = -1) {
        p.amaxY = amax_b + slot_y;
        p.amaxX = amax_f + slot_x;
        p.amaxX2 = slot_x2 >= 0 ? amax_f + slot_x2 : nullptr;
        return p;
    };
    WgradParts parts{};
    int n = 0;
    // The narrow parts (4 MFMAs per k-step) wait on memory rather than on the MFMA pipe
    // and run longer than their MFMA count says: dispatch them first, the wide ones after.
    parts.part[n++] = with_max(wgrad_part_desc(DY(0), 256, XE, 64, grads[0], 63, 0, 63, grads[1], 2), 0, vw::SLOT_XE);
    parts.part[n++] = with_max(wgrad_part_desc(DY(4), 256, XE, 64, grads[8], 319, xe_col, 63, nullptr, 2), 4, vw::SLOT_XE);
    parts.part[n++] = with_max(wgrad_part_desc(DYR, 128, FEAT, 256, grads[20], 283, 0, 256, grads[21], 1, DE, 32, 256, 27), 9,
                               vw::SLOT_FEAT, vw::SLOT_DE);
    for (int l = 1; l < 8; ++l)
        parts.part[n++] = with_max(wgrad_part_desc(DY(l), 256, H(l - 1), 256, grads[2 * l], l == 4 ? 319 : 256,
                                                   l == 4 ? h_col : 0, 256, grads[2 * l + 1], 0), l, vw::SLOT_H + l - 1);
    {   // the feature layer's part also forms the density head's gradient: the same h7 rows times d sigma_pre
        WgradPart pf = with_max(wgrad_part_desc(DFEAT, 256, H(7), 256, grads[18], 256, 0, 256, grads[19], 0), 8, vw::SLOT_H + 7);
        pf.sk_d = ws2 + P32 * vw::DPS;
        pf.sk_dW = grads[16];
        pf.sk_db = grads[17];
        parts.part[n++] = pf;
    }
    parts.n = n;
    if (int rc = launch_wgrad(parts, P32, st)) return rc;

    SkinnyPart rg{DOUT4, HR, grads[22], grads[23], 4, 128, 1, 3, 128};
    return launch_skinny(rg, P32, st);
}
