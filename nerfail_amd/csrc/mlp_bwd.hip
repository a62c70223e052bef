// K4b: backward of the fused NeRF MLP (autograd of run_nerf_helpers.py:100-123 inside loss.backward(), RN:791).
//
// Two kernel families, both on v_mfma_f32_32x32x2_f32, both reading the activations that nerfail_mlp_fwd_train saved as
// channel-major tiles (mlp_layout.h):
//
//  1. nerf_mlp_bwd_data_kernel: the backward-data chain dX = W^T dZ, register resident exactly like the
//     forward: dZ of a layer sits in the accumulator layout (sample on the lane, channel on the register), which
//     is the B operand of the next (earlier) layer's MFMA with A = a 32-row slab of W^T (packed once by
//     nerfail_mlp_pack_T). ReLU masks come from the saved post-ReLU activations (h > 0 <=> pre-activation > 0).
//     Every dZ is stored (fragment layout) for kernel 2. No gradient w.r.t. points/dirs is needed (RN:394 detaches
//     z_samples; rays are data), so the chain stops at layer 1.
//
//  2. the weight gradients dW = dZ X^T live in mlp_dw.hip.
#include <cstdlib>
#include "mlp_layout.h"

namespace nerfail {

// ------------------------------------------------------------------------------------- W^T packing
// [quad][in-tile t][lane (i = l&31 -> input channel 32t+i, h = l>>5)][e]: W[o = out channel of k-step 4q+e in half h][col0 + 32t + i]
__global__ void pack_layer_T_kernel(const float* __restrict__ w, int out_f, int in_f, int col0, int NT, int total,
                                    float* __restrict__ wq) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int e = g & 3, lane = (g >> 2) & 63, rest = g >> 8;
    const int t = rest % NT, q = rest / NT;
    const int s = 4 * q + e, hh = lane >> 5;
    const int o = 32 * (s / 16) + acc_channel(s % 16, hh);
    const int i = 32 * t + (lane & 31);
    wq[g] = (o < out_f) ? w[(long)o * in_f + col0 + i] : 0.f;
}

// ------------------------------------------------------------------------------------- backward data
template <int NT>
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_data_kernel(BwdArgs a) {
    constexpr int OTV = NT / 2;
    const int lane = threadIdx.x & 63;
    // wave id made PROVABLY wave-uniform: tile bases then live in SGPRs and every access is scalar-base + 32-bit
    // lane offset instead of a 64-bit VGPR pair per address (which spilled hundreds of registers)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const MlpLayout& L = a.lay;
    const TrainLayout& TL = a.tl;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    for (long rnd = 0; rnd < nrounds; ++rnd) {
        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;
        if (tile >= ntiles) break;
        const bool second = tile >= a.split;                     // wave-uniform: the images stay scalar bases
        const float* __restrict__ P = second ? a.packed2 : a.packed;
        const float* __restrict__ PT = second ? a.packedT2 : a.packedT;
        const long sraw = tile * 32 + j;
        const float* __restrict__ A = a.acts + (size_t)tile * TL.a_slots * 1024;
        float* __restrict__ Z = a.dz + (size_t)tile * TL.z_slots * 1024;
        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sraw < a.M) dr = reinterpret_cast<const float4*>(a.d_raw)[sraw];   // padded samples carry zero gradient

        // ---- ZR: d_raw as a tile (channels 0..3 live in half 0, registers 0..3)
        {
            f32x16 zr[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[0][r] = 0.f;
            if (h == 0) { zr[0][0] = dr.x; zr[0][1] = dr.y; zr[0][2] = dr.z; zr[0][3] = dr.w; }
            store_tiles<1>(Z + TL.z_ZR * 1024, zr, lane);
        }
        // ---- rgb_linear backward: dZ_v = (W_rgb^T d_rgb) * [hv > 0]
        f32x16 dzv[OTV];
        {
            const float* wr = P + L.rgb_off;
            const TileMask<OTV> mhv = load_mask<OTV>(A + TL.a_MASK * 1024, L.D, lane);
#pragma unroll
            for (int t = 0; t < OTV; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g = wr[((0 * OTV + t) * 2 + h) * 16 + r] * dr.x + wr[((1 * OTV + t) * 2 + h) * 16 + r] * dr.y +
                                    wr[((2 * OTV + t) * 2 + h) * 16 + r] * dr.z;
                    dzv[t][r] = mask_apply<OTV>(mhv, t, r, g);
                }
            }
            store_tiles<OTV>(Z + TL.z_ZV * 1024, dzv, lane);
        }
        // ---- views_linears[0] backward (feature columns): d_feature = Wv[:, :W]^T dZ_v
        f32x16 cur[NT], nxt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[t][r] = 0.f;
        mfma_part<NT, OTV * 4>(cur, PT + a.layT.w_off[L.D + 1], lane, [&](int q, int e) { return dzv[q >> 2][4 * (q & 3) + e]; });
        store_tiles<NT>(Z + TL.z_ZF * 1024, cur, lane);         // feature_linear has no activation: dZ_F = d_feature
        // ---- feature_linear + alpha_linear backward: d_h = Wf^T d_feature + w_alpha * d_sigma
        {
            const float* wa = P + L.alpha_off;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[t][r] = wa[(t * 2 + h) * 16 + r] * dr.w;
        }
        mfma_acts<NT, NT>(nxt, PT + a.layT.w_off[L.D], lane, cur);
        // ---- pts_linears[D-1 .. 0]
#pragma unroll 1
        for (int i = L.D - 1; i >= 0; --i) {
            // dZ_i = d_h_{i+1} * [h_{i+1} > 0]
            const TileMask<NT> mk = load_mask<NT>(A + TL.a_MASK * 1024, i, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[t][r] = mask_apply<NT>(mk, t, r, nxt[t][r]);
            store_tiles<NT>(Z + (TL.z_Z0 + i * NT) * 1024, cur, lane);
            if (i == 0) break;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) nxt[t][r] = 0.f;
            mfma_acts<NT, NT>(nxt, PT + a.layT.w_off[i], lane, cur);     // d_h_i = W_i[:, h-part]^T dZ_i
        }
    }
}


static int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}

}  // namespace nerfail

using namespace nerfail;

// 0 = automatic (LDS-ring form where it applies), 1 = register-streamed form, 2 = LDS-ring form or NERFAIL_EINVAL
static int bwd_select_from_env() {
    const char* e = getenv("NERFAIL_BWD_KERNEL");
    if (e == nullptr) return 0;
    return e[0] == 'r' ? 1 : (e[0] == 'l' ? 2 : 0);
}
static int g_bwd_select = bwd_select_from_env();

extern "C" int nerfail_mlp_bwd_select(int which) {
    const int prev = g_bwd_select;
    if (which >= 0 && which <= 2) g_bwd_select = which;
    return prev;
}

extern "C" size_t nerfail_mlp_packed_T_floats(int D, int W, int skip) {
    MlpLayout L;
    if (!make_layout(D, W, skip, L)) return 0;
    MlpLayoutT T;
    make_layout_T(D, L.NT, T);
    return (size_t)T.total;
}

extern "C" size_t nerfail_mlp_train_dz_floats(int D, int W, int64_t M) {
    MlpLayout L;
    if (!make_layout(D, W, -1, L) || M < 0) return 0;
    return (size_t)((M + 31) / 32) * make_train_layout(D, W).z_slots * 1024;
}

extern "C" int nerfail_mlp_pack_T(const nerfail_mlp_params* p, float* packedT, void* stream) {
    NF_REQUIRE(p != nullptr && packedT != nullptr, "NULL pointer");
    MlpLayout L;
    NF_REQUIRE(make_layout(p->D, p->W, p->skip, L), "unsupported (D, W)");
    MlpLayoutT T;
    make_layout_T(p->D, L.NT, T);
    hipStream_t s = as_stream(stream);
    const int W = p->W, NT = L.NT;
    for (int l = 1; l <= p->D + 1; ++l) {
        const float* w;
        int out_f, in_f, col0 = 0;
        if (l < p->D) {
            NF_REQUIRE(p->pts_w[l] != nullptr, "pts_linears pointer is NULL");
            const bool emb = layer_has_emb(l, L.skip);
            w = p->pts_w[l]; out_f = W; in_f = emb ? W + kPtsCh : W; col0 = emb ? kPtsCh : 0;
        } else if (l == p->D) {
            NF_REQUIRE(p->feature_w != nullptr, "feature_linear pointer is NULL");
            w = p->feature_w; out_f = W; in_f = W;
        } else {
            NF_REQUIRE(p->views_w != nullptr, "views_linears pointer is NULL");
            w = p->views_w; out_f = W / 2; in_f = W + kDirCh;
        }
        const int total = (int)(((l == p->D + 1) ? (NT / 2) * 4 : NT * 4) * NT * 256);
        pack_layer_T_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(w, out_f, in_f, col0, NT, total, packedT + T.w_off[l]);
        NF_LAUNCHED("pack_layer_T_kernel");
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_bwd_data(const float* packed, const float* packedT, int D, int W, int skip, const float* d_raw,
                                    const float* acts, int64_t M, float* dz, void* stream) {
    return nerfail_mlp_bwd_data2(packed, packedT, M, nullptr, nullptr, 0, D, W, skip, d_raw, acts, dz, stream);
}

extern "C" int nerfail_mlp_bwd_data2(const float* packed0, const float* packedT0, int64_t M0, const float* packed1,
                                     const float* packedT1, int64_t M1, int D, int W, int skip, const float* d_raw,
                                     const float* acts, float* dz, void* stream) {
    NF_REQUIRE(M0 >= 0 && M1 >= 0, "M is negative");
    BwdArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    const int64_t M = M0 + M1;
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(d_raw && acts && dz, "NULL pointer");
    NF_REQUIRE(M0 == 0 || (packed0 && packedT0), "NULL weight image");
    NF_REQUIRE(M1 == 0 || (packed1 && packedT1), "NULL weight image of the second network");
    NF_REQUIRE(M1 == 0 || M0 % 32 == 0, "with two networks the first one's sample count must be a multiple of 32");
    make_layout_T(D, a.lay.NT, a.layT);
    a.tl = make_train_layout(D, W);
    a.packed = packed0; a.packedT = packedT0; a.packed2 = packed1; a.packedT2 = packedT1; a.split = (M0 + 31) / 32;
    a.d_raw = d_raw; a.acts = acts; a.dz = dz; a.M = M;
    const long ntiles = (M + 31) / 32;
    long blocks = (ntiles + 3) / 4;
    if (blocks > cu_count()) blocks = cu_count();
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = as_stream(stream);
    a.blocks0 = 0;
    // W = 256, even depth <= 8: the LDS-ring form (mlp_lds.hip); g_bwd_select / NERFAIL_BWD_KERNEL=reg forces the register form
    if (g_bwd_select != 1 && W == 256 && !(D & 1) && D >= 2 && D <= 8) return launch_bwd_data_lds(a, W, cu_count(), s);
    if (g_bwd_select == 2) { set_error("nerfail_mlp_bwd_data: the LDS-ring kernel does not cover this shape"); return NERFAIL_EINVAL; }
    switch (W) {
        case 256: nerf_mlp_bwd_data_kernel<8><<<grid, block, 0, s>>>(a); break;
        case 128: nerf_mlp_bwd_data_kernel<4><<<grid, block, 0, s>>>(a); break;
        case 64: nerf_mlp_bwd_data_kernel<2><<<grid, block, 0, s>>>(a); break;
        default: set_error("nerfail_mlp_bwd_data: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_bwd_data_kernel");
    return NERFAIL_OK;
}
