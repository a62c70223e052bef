// K3 + K4: fused positional encoding + NeRF MLP forward on the exact-f32 matrix cores.
// Replaces run_network (run_nerf.py:37-51), Embedder.embed (run_nerf_helpers.py:15-50) and
// NeRF.forward (run_nerf_helpers.py:100-123); ~99 % of the render path's FLOPs.
//
// Design (CDNA4 / gfx950, wave64, v_mfma_f32_32x32x2_f32 = bit-exact f32 FMA chain):
//   * Every layer is computed TRANSPOSED: H^T[out, sample] = W[out, in] * X^T[in, sample]. The MFMA
//     A operand is a 32-row slab of W, the B operand is the activation, so the 32x32 accumulator
//     tile holds "sample on the lane, output channel on the register": lane l (h = l>>5, j = l&31)
//     owns sample j and channels c(r,h) = (r&3) + 8*(r>>2) + 4*h of the tile in registers r=0..15.
//   * That is exactly the B-operand shape of the NEXT layer (B[k][j]: lane half h supplies one k per
//     step, lane j the sample), so after bias+ReLU the accumulators feed the next layer's MFMAs
//     directly. Activations never leave the register file: no LDS, no HBM, no barriers. A wave owns
//     32 samples x all W channels (128 accumulator registers in + 128 out at W = 256, which is why
//     the kernel runs one wave per SIMD with the 512-entry unified VGPR/AGPR file).
//   * The k order a layer consumes is therefore "whatever the previous accumulator layout holds";
//     the weights are re-packed ONCE (nerfail_mlp_pack) into that k order and into the A-fragment
//     lane order, so each weight read is one fully coalesced 16-byte-per-lane load covering 4 MFMA
//     k-steps. The 2.4 MB image stays L2-resident; all waves stream it in the same order.
//   * Positional encoding is computed per lane in registers: MFMA step s of the encoding part needs,
//     for lane half 0 / 1, sin / cos of the SAME argument x_d*2^f: one (sin, cos) pair per step, and all bands of
//     a coordinate share ONE double-precision argument reduction (SinCosBands, common.h).
//   * alpha_linear (W->1) and rgb_linear (W/2->3) are too thin for a 32-wide MFMA tile: VALU dot
//     products on the accumulator registers + one cross-half shuffle.
// Bound: f32 MFMA (157 TFLOP/s dense on MI355X); algorithmic work 1 186 816 FLOP per sample (D8 W256).
#include <stdlib.h>
#include "mlp_layout.h"

namespace nerfail {

// ------------------------------------------------------------------------------------- packing
// One launch per MFMA layer: writes the A-fragment image [quad][tile][lane][4] and the bias image.
__global__ void pack_layer_kernel(const float* __restrict__ w, const float* __restrict__ b, int out_f, int in_f,
                                  int OT, int NT, int emb_col0, int h_col0, int dir_col0, float* __restrict__ wq,
                                  float* __restrict__ bq, int total_w) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < OT * 32) {   // bias image [OT][2][16]
        const int t = g / 32, hh = (g / 16) & 1, r = g & 15;
        const int ch = 32 * t + acc_channel(r, hh);
        bq[g] = (ch < out_f) ? b[ch] : 0.f;
    }
    if (g >= total_w) return;
    const int e = g & 3, lane = (g >> 2) & 63, rest = g >> 8;
    const int t = rest % OT;
    int q = rest / OT;
    const int hh = lane >> 5, row = 32 * t + (lane & 31);
    int col = -1;
    if (emb_col0 >= 0) {
        if (q < kEmbQuads) { const int c = enc_channel(4 * q + e, hh, 10); col = c < 0 ? -1 : emb_col0 + c; q = -1; }
        else q -= kEmbQuads;
    }
    if (q >= 0 && h_col0 >= 0) {
        if (q < NT * 4) { const int s = 4 * q + e; col = h_col0 + 32 * (s / 16) + acc_channel(s % 16, hh); q = -1; }
        else q -= NT * 4;
    }
    if (q >= 0 && dir_col0 >= 0) {
        const int c = enc_channel(4 * q + e, hh, 4); col = c < 0 ? -1 : dir_col0 + c;
    }
    wq[g] = (row < out_f && col >= 0) ? w[(long)row * in_f + col] : 0.f;
}

// All MFMA layers in ONE launch (the image is re-packed after every optimizer step): a table of per-layer
// descriptors passed by value; a block works on one layer (blockIdx.y), grid-striding over its elements.
struct PackLayerDesc {
    const float* w; const float* b;
    int out_f, in_f, OT, emb0, h0, dir0, total_w;
    unsigned w_off, b_off;
};
struct PackTable { int n, NT; PackLayerDesc l[NERFAIL_MAX_DEPTH + 2]; };

__global__ void pack_all_layers_kernel(PackTable t, float* __restrict__ packed) {
    const PackLayerDesc& d = t.l[blockIdx.y];
    const int n = d.total_w > d.OT * 32 ? d.total_w : d.OT * 32;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) {
        if (g < d.OT * 32) {
            const int tt = g / 32, hh = (g / 16) & 1, r = g & 15;
            const int ch = 32 * tt + acc_channel(r, hh);
            packed[d.b_off + g] = (ch < d.out_f) ? d.b[ch] : 0.f;
        }
        if (g >= d.total_w) continue;
        const int e = g & 3, lane = (g >> 2) & 63, rest = g >> 8;
        const int tt = rest % d.OT;
        int q = rest / d.OT;
        const int hh = lane >> 5, row = 32 * tt + (lane & 31);
        int col = -1;
        if (d.emb0 >= 0) {
            if (q < kEmbQuads) { const int c = enc_channel(4 * q + e, hh, 10); col = c < 0 ? -1 : d.emb0 + c; q = -1; }
            else q -= kEmbQuads;
        }
        if (q >= 0 && d.h0 >= 0) {
            if (q < t.NT * 4) { const int s_ = 4 * q + e; col = d.h0 + 32 * (s_ / 16) + acc_channel(s_ % 16, hh); q = -1; }
            else q -= t.NT * 4;
        }
        if (q >= 0 && d.dir0 >= 0) {
            const int c = enc_channel(4 * q + e, hh, 4); col = c < 0 ? -1 : d.dir0 + c;
        }
        packed[d.w_off + g] = (row < d.out_f && col >= 0) ? d.w[(long)row * d.in_f + col] : 0.f;
    }
}

// alpha image [NT][2][16] + bias, rgb image [3][OTV][2][16] + 3 biases
__global__ void pack_heads_kernel(const float* __restrict__ aw, const float* __restrict__ ab,
                                  const float* __restrict__ rw, const float* __restrict__ rb, int W,
                                  float* __restrict__ aq, float* __restrict__ rq) {
    const int NT = W / 32, OTV = NT / 2;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < NT * 32) {
        const int t = g / 32, hh = (g / 16) & 1, r = g & 15;
        aq[g] = aw[32 * t + acc_channel(r, hh)];
    } else if (g < NT * 32 + 4) {
        aq[g] = (g == NT * 32) ? ab[0] : 0.f;
    }
    if (g < 3 * OTV * 32) {
        const int c = g / (OTV * 32), rem = g % (OTV * 32);
        const int t = rem / 32, hh = (rem / 16) & 1, r = rem & 15;
        rq[g] = rw[c * (W / 2) + 32 * t + acc_channel(r, hh)];
    } else if (g < 3 * OTV * 32 + 4) {
        const int c = g - 3 * OTV * 32;
        rq[g] = (c < 3) ? rb[c] : 0.f;
    }
}

// Training re-packs BOTH images (forward + transposed) after every optimizer step: 1 + 1 + (D + 1) launches of ~5 us each
// per network were 1.3 % of a training step. One launch instead: blockIdx.y walks the forward layers, then the heads,
// then the transposed layers ([quad][in-tile t][lane (i = l&31 -> input channel 32t+i, h = l>>5)][e]:
// W[o = out channel of k-step 4q+e in half h][col0 + 32t + i], as pack_layer_T_kernel in mlp_bwd.hip).
struct PackTDesc { const float* w; int out_f, in_f, col0, total; unsigned off; };
struct PackTrainTable {
    PackTable fwd;
    int nT, W;
    const float* aw; const float* ab; const float* rw; const float* rb;
    unsigned alpha_off, rgb_off;
    PackTDesc t[NERFAIL_MAX_DEPTH + 2];
};
__global__ void pack_train_kernel(PackTrainTable t, float* __restrict__ packed, float* __restrict__ packedT) {
    const int job = blockIdx.y, NT = t.fwd.NT;
    const int stride = gridDim.x * blockDim.x, g0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (job < t.fwd.n) {
        const PackLayerDesc& d = t.fwd.l[job];
        const int n = d.total_w > d.OT * 32 ? d.total_w : d.OT * 32;
        for (int g = g0; g < n; g += stride) {
            if (g < d.OT * 32) {
                const int tt = g / 32, hh = (g / 16) & 1, r = g & 15;
                const int ch = 32 * tt + acc_channel(r, hh);
                packed[d.b_off + g] = (ch < d.out_f) ? d.b[ch] : 0.f;
            }
            if (g >= d.total_w) continue;
            const int e = g & 3, lane = (g >> 2) & 63, rest = g >> 8;
            const int tt = rest % d.OT;
            int q = rest / d.OT;
            const int hh = lane >> 5, row = 32 * tt + (lane & 31);
            int col = -1;
            if (d.emb0 >= 0) {
                if (q < kEmbQuads) { const int c = enc_channel(4 * q + e, hh, 10); col = c < 0 ? -1 : d.emb0 + c; q = -1; }
                else q -= kEmbQuads;
            }
            if (q >= 0 && d.h0 >= 0) {
                if (q < NT * 4) { const int s_ = 4 * q + e; col = d.h0 + 32 * (s_ / 16) + acc_channel(s_ % 16, hh); q = -1; }
                else q -= NT * 4;
            }
            if (q >= 0 && d.dir0 >= 0) {
                const int c = enc_channel(4 * q + e, hh, 4); col = c < 0 ? -1 : d.dir0 + c;
            }
            packed[d.w_off + g] = (row < d.out_f && col >= 0) ? d.w[(long)row * d.in_f + col] : 0.f;
        }
    } else if (job == t.fwd.n) {
        const int W = t.W, OTV = NT / 2;
        float* __restrict__ aq = packed + t.alpha_off;
        float* __restrict__ rq = packed + t.rgb_off;
        for (int g = g0; g < NT * 32 + 4 || g < 3 * OTV * 32 + 4; g += stride) {
            if (g < NT * 32) {
                const int tt = g / 32, hh = (g / 16) & 1, r = g & 15;
                aq[g] = t.aw[32 * tt + acc_channel(r, hh)];
            } else if (g < NT * 32 + 4) {
                aq[g] = (g == NT * 32) ? t.ab[0] : 0.f;
            }
            if (g < 3 * OTV * 32) {
                const int c = g / (OTV * 32), rem = g % (OTV * 32);
                const int tt = rem / 32, hh = (rem / 16) & 1, r = rem & 15;
                rq[g] = t.rw[c * (W / 2) + 32 * tt + acc_channel(r, hh)];
            } else if (g < 3 * OTV * 32 + 4) {
                const int c = g - 3 * OTV * 32;
                rq[g] = (c < 3) ? t.rb[c] : 0.f;
            }
        }
    } else {
        const PackTDesc& d = t.t[job - t.fwd.n - 1];
        for (int g = g0; g < d.total; g += stride) {
            const int e = g & 3, lane = (g >> 2) & 63, rest = g >> 8;
            const int tt = rest % NT, q = rest / NT;
            const int s_ = 4 * q + e, hh = lane >> 5;
            const int o = 32 * (s_ / 16) + acc_channel(s_ % 16, hh);
            const int i = 32 * tt + (lane & 31);
            packedT[d.off + g] = (o < d.out_f) ? d.w[(long)o * d.in_f + d.col0 + i] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------- device side
template <int NT, bool TRAIN>
__global__ __launch_bounds__(256, 1) void nerf_mlp_fwd_kernel(MlpArgs a) {
    constexpr int OTV = NT / 2;
    const int lane = threadIdx.x & 63;
    // wave id made PROVABLY wave-uniform: tile bases then live in SGPRs and every access is scalar-base + 32-bit
    // lane offset instead of a 64-bit VGPR pair per address (which spilled hundreds of registers)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ P = a.packed;
    const MlpLayout& L = a.lay;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    for (long rnd = 0; rnd < nrounds; ++rnd) {
        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;
        if (tile >= ntiles) break;     // wave-uniform; waves are fully independent (no barriers)
        const long sraw = tile * 32 + j;
        const long s = sraw < a.M ? sraw : a.M - 1;

        // ---- B operands of the encoding parts, in registers
        float emb[4 * kEmbQuads], demb[4 * kDirQuads];
        encode_sample(a, s, h, emb, demb);

        f32x16 act[NT], acc[NT];
        float* __restrict__ A = nullptr;     // this tile's activation slots (training)
        if (TRAIN) {
            A = a.acts + (size_t)tile * (train_a_slots(L.D, NT) * 1024);
            store_enc<10, 4 * kEmbQuads>(A, emb, lane);              // E0 E1: 63 channels
            store_enc<4, 4 * kDirQuads>(A + 2 * 1024, demb, lane);   // V: 27 channels
        }
        // ---- layer 0: 63 -> W
        load_bias<NT>(acc, P + L.b_off[0], h);
        mfma_scalars<NT, kEmbQuads>(acc, P + L.w_off[0], lane, emb);
        relu_to<NT>(act, acc, true);
        if (TRAIN) {
            store_tiles<NT>(A + 3 * 1024, act, lane);
            store_mask<NT>(A + train_mask_slot0(L.D, NT) * 1024, 0, mask_of<NT>(act), lane);
        }

        // ---- layers 1..D-1 (pts_linears, ReLU) and D (feature_linear, no activation)
        float alpha = 0.f;
#pragma unroll 1
        for (int l = 1; l <= L.D; ++l) {
            if (l == L.D) {   // alpha_linear on the last pts activation (RH:110)
                const float* wa = P + L.alpha_off;
                float sacc = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4* p = reinterpret_cast<const f32x4*>(wa + (t * 2 + h) * 16);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 wv = p[r4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) sacc = fmaf(wv[e], act[t][4 * r4 + e], sacc);
                    }
                }
                sacc += __shfl_xor(sacc, 32, 64);
                alpha = sacc + wa[NT * 32];
            }
            load_bias<NT>(acc, P + L.b_off[l], h);
            const float* w = P + L.w_off[l];
            if (l == L.skip + 1 && L.skip >= 0) {   // h = cat([input_pts, h]) (RH:106-107)
                if (TRAIN) {
                    // training: the encoding was saved to `acts`; reloading it here (instead of keeping 32 registers
                    // live across the layer loop) leaves room for the double-buffered weight stream
                    float e2[4 * kEmbQuads];
                    load_enc<10, 4 * kEmbQuads>(A, e2, lane);
                    mfma_scalars<NT, kEmbQuads>(acc, w, lane, e2);
                } else {
                    mfma_scalars<NT, kEmbQuads>(acc, w, lane, emb);
                }
                w += kEmbQuads * NT * 256;
            }
            mfma_acts<NT, NT>(acc, w, lane, act);
            relu_to<NT>(act, acc, l < L.D);
            if (TRAIN) {
                store_tiles<NT>(A + (3 + l * NT) * 1024, act, lane);   // H_{l+1} for l < D, F for l == D
                if (l < L.D) store_mask<NT>(A + train_mask_slot0(L.D, NT) * 1024, l, mask_of<NT>(act), lane);
            }
        }

        // ---- views_linears[0]: cat([feature, embedded dirs]) -> W/2, ReLU (RH:112-116)
        f32x16 hv[OTV];
        load_bias<OTV>(hv, P + L.b_off[L.D + 1], h);
        mfma_acts<OTV, NT>(hv, P + L.w_off[L.D + 1], lane, act);
        if (TRAIN) {
            float d2[4 * kDirQuads];
            load_enc<4, 4 * kDirQuads>(A + 2 * 1024, d2, lane);
            mfma_scalars<OTV, kDirQuads>(hv, P + L.w_off[L.D + 1] + NT * 4 * OTV * 256, lane, d2);
        } else {
            mfma_scalars<OTV, kDirQuads>(hv, P + L.w_off[L.D + 1] + NT * 4 * OTV * 256, lane, demb);
        }
        if (TRAIN) {
            f32x16 hvr[OTV];
            relu_to<OTV>(hvr, hv, true);
            store_tiles<OTV>(A + (3 + (L.D + 1) * NT) * 1024, hvr, lane);
            store_mask<OTV>(A + train_mask_slot0(L.D, NT) * 1024, L.D, mask_of<OTV>(hvr), lane);
        }

        // ---- rgb_linear: W/2 -> 3 (RH:118)
        const float* wr = P + L.rgb_off;
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < OTV; ++t) {
                const f32x4* p = reinterpret_cast<const f32x4*>(wr + ((c * OTV + t) * 2 + h) * 16);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 wv = p[r4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc = fmaf(wv[e], fmaxf(hv[t][4 * r4 + e], 0.f), sacc);
                }
            }
            sacc += __shfl_xor(sacc, 32, 64);
            rgb[c] = sacc + wr[3 * OTV * 32 + c];
        }
        if (h == 0 && sraw < a.M)
            reinterpret_cast<float4*>(a.raw)[sraw] = make_float4(rgb[0], rgb[1], rgb[2], alpha);
    }
}

// Inference goes to the LDS-streaming kernel (mlp_lds.hip) when it covers the shape (even depth <= 8); the training
// forward (activations saved) and the other shapes run the register-streamed kernel below. nerfail_mlp_fwd_select /
// NERFAIL_FWD_KERNEL=reg|lds force one of them (A/B timing, parity test of one against the other).
static int g_fwd_select = [] { const char* e = getenv("NERFAIL_FWD_KERNEL"); return e ? (e[0] == 'r' ? 1 : (e[0] == 'l' ? 2 : 0)) : 0; }();
static bool use_lds_kernel(const MlpArgs& a) {
    if (g_fwd_select == 1) return false;
    if (g_fwd_select == 2) return true;
    return !(a.lay.D & 1) && a.lay.D <= 8;
}

static int launch_mlp(const MlpArgs& a, int W, hipStream_t s) {
    if (use_lds_kernel(a)) return launch_mlp_lds(a, W, s);
    const long ntiles = (a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;      // persistent: one 4-wave workgroup per CU, one wave per SIMD
    const dim3 grid((unsigned)blocks), block(256);
    const bool train = a.acts != nullptr;
    switch (W) {
        case 256: if (train) nerf_mlp_fwd_kernel<8, true><<<grid, block, 0, s>>>(a); else nerf_mlp_fwd_kernel<8, false><<<grid, block, 0, s>>>(a); break;
        case 128: if (train) nerf_mlp_fwd_kernel<4, true><<<grid, block, 0, s>>>(a); else nerf_mlp_fwd_kernel<4, false><<<grid, block, 0, s>>>(a); break;
        case 64: if (train) nerf_mlp_fwd_kernel<2, true><<<grid, block, 0, s>>>(a); else nerf_mlp_fwd_kernel<2, false><<<grid, block, 0, s>>>(a); break;
        default: set_error("nerfail_mlp_fwd: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_fwd_kernel");
    return NERFAIL_OK;
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_mlp_fwd_select(int which) {
    const int prev = g_fwd_select;
    if (which >= 0 && which <= 2) g_fwd_select = which;
    return prev;
}

extern "C" size_t nerfail_mlp_packed_floats(int D, int W, int skip) {
    MlpLayout L;
    return make_layout(D, W, skip, L) ? (size_t)L.total : 0;
}

static int fill_pack_table(const nerfail_mlp_params* p, MlpLayout& L, PackTable& tab) {
    NF_REQUIRE(p->input_ch == kPtsCh && p->input_ch_views == kDirCh, "only multires=10 / multires_views=4 (63 + 27 channels)");
    NF_REQUIRE(make_layout(p->D, p->W, p->skip, L), "unsupported (D, W): W in {64,128,256}, 2 <= D <= 16");
    for (int i = 0; i < p->D; ++i) NF_REQUIRE(p->pts_w[i] != nullptr && p->pts_b[i] != nullptr, "pts_linears pointer is NULL");
    NF_REQUIRE(p->views_w && p->views_b && p->feature_w && p->feature_b && p->alpha_w && p->alpha_b && p->rgb_w && p->rgb_b,
               "head pointer is NULL");
    const int W = p->W, NT = L.NT, OTV = NT / 2;
    tab.n = p->D + 2; tab.NT = NT;
    for (int l = 0; l <= p->D + 1; ++l) {
        const bool emb = l <= p->D - 1 && layer_has_emb(l, L.skip);
        PackLayerDesc& d = tab.l[l];
        d.OT = NT; d.emb0 = -1; d.h0 = -1; d.dir0 = -1;
        if (l < p->D) {
            d.w = p->pts_w[l]; d.b = p->pts_b[l]; d.out_f = W;
            d.in_f = (l == 0) ? kPtsCh : (emb ? W + kPtsCh : W);
            if (emb) d.emb0 = 0;
            if (l > 0) d.h0 = emb ? kPtsCh : 0;
        } else if (l == p->D) {
            d.w = p->feature_w; d.b = p->feature_b; d.out_f = W; d.in_f = W; d.h0 = 0;
        } else {
            d.w = p->views_w; d.b = p->views_b; d.out_f = W / 2; d.in_f = W + kDirCh; d.OT = OTV; d.h0 = 0; d.dir0 = W;
        }
        d.total_w = (int)L.w_count[l];
        d.w_off = L.w_off[l]; d.b_off = L.b_off[l];
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_pack(const nerfail_mlp_params* p, float* packed, void* stream) {
    NF_REQUIRE(p != nullptr && packed != nullptr, "NULL pointer");
    MlpLayout L;
    PackTable tab;
    const int rc = fill_pack_table(p, L, tab);
    if (rc != NERFAIL_OK) return rc;
    hipStream_t s = as_stream(stream);
    const int W = p->W, NT = L.NT, OTV = NT / 2;
    pack_all_layers_kernel<<<dim3(64, (unsigned)tab.n), dim3(256), 0, s>>>(tab, packed);
    NF_LAUNCHED("pack_all_layers_kernel");
    const int nh = (NT * 32 + 4) > (3 * OTV * 32 + 4) ? (NT * 32 + 4) : (3 * OTV * 32 + 4);
    pack_heads_kernel<<<dim3((nh + 255) / 256), dim3(256), 0, s>>>(p->alpha_w, p->alpha_b, p->rgb_w, p->rgb_b, W,
                                                                  packed + L.alpha_off, packed + L.rgb_off);
    NF_LAUNCHED("pack_heads_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_pack_train(const nerfail_mlp_params* p, float* packed, float* packedT, void* stream) {
    NF_REQUIRE(p != nullptr && packed != nullptr && packedT != nullptr, "NULL pointer");
    MlpLayout L;
    PackTrainTable t;
    const int rc = fill_pack_table(p, L, t.fwd);
    if (rc != NERFAIL_OK) return rc;
    MlpLayoutT T;
    make_layout_T(p->D, L.NT, T);
    const int W = p->W, NT = L.NT;
    t.W = W; t.aw = p->alpha_w; t.ab = p->alpha_b; t.rw = p->rgb_w; t.rb = p->rgb_b;
    t.alpha_off = L.alpha_off; t.rgb_off = L.rgb_off;
    t.nT = 0;
    for (int l = 1; l <= p->D + 1; ++l) {
        PackTDesc& d = t.t[t.nT++];
        if (l < p->D) {
            const bool emb = layer_has_emb(l, L.skip);
            d.w = p->pts_w[l]; d.out_f = W; d.in_f = emb ? W + kPtsCh : W; d.col0 = emb ? kPtsCh : 0;
        } else if (l == p->D) {
            d.w = p->feature_w; d.out_f = W; d.in_f = W; d.col0 = 0;
        } else {
            d.w = p->views_w; d.out_f = W / 2; d.in_f = W + kDirCh; d.col0 = 0;
        }
        d.total = (int)(((l == p->D + 1) ? (NT / 2) * 4 : NT * 4) * NT * 256);
        d.off = T.w_off[l];
    }
    pack_train_kernel<<<dim3(32, (unsigned)(t.fwd.n + 1 + t.nT)), dim3(256), 0, as_stream(stream)>>>(t, packed, packedT);
    NF_LAUNCHED("pack_train_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_fwd(const float* packed, int D, int W, int skip, const float* pts, const float* viewdirs,
                               int64_t M, int samples_per_ray, float* raw, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    NF_REQUIRE(samples_per_ray >= 1, "samples_per_ray must be positive");
    MlpArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed != nullptr && pts != nullptr && viewdirs != nullptr && raw != nullptr, "NULL pointer");
    a.packed = packed; a.pts = pts; a.viewdirs = viewdirs; a.xemb = nullptr; a.rays = nullptr; a.z = nullptr; a.raw = raw; a.acts = nullptr; a.M = M; a.spr = samples_per_ray;
    return launch_mlp(a, W, as_stream(stream));
}

extern "C" int nerfail_mlp_fwd_embedded(const float* packed, int D, int W, int skip, const float* x, int64_t M, float* raw,
                                        void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    MlpArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed != nullptr && x != nullptr && raw != nullptr, "NULL pointer");
    a.packed = packed; a.pts = nullptr; a.viewdirs = nullptr; a.xemb = x; a.rays = nullptr; a.z = nullptr; a.raw = raw; a.acts = nullptr; a.M = M; a.spr = 1;
    return launch_mlp(a, W, as_stream(stream));
}

extern "C" size_t nerfail_mlp_train_acts_floats(int D, int W, int64_t M) {
    MlpLayout L;
    if (!make_layout(D, W, -1, L) || M < 0) return 0;
    return (size_t)((M + 31) / 32) * make_train_layout(D, W).a_slots * 1024;   // activation tiles + ReLU bit masks
}

extern "C" int nerfail_mlp_fwd_train(const float* packed, int D, int W, int skip, const float* pts, const float* viewdirs,
                                     int64_t M, int samples_per_ray, float* raw, float* acts, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    NF_REQUIRE(samples_per_ray >= 1, "samples_per_ray must be positive");
    MlpArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed != nullptr && pts != nullptr && viewdirs != nullptr && raw != nullptr && acts != nullptr, "NULL pointer");
    a.packed = packed; a.pts = pts; a.viewdirs = viewdirs; a.xemb = nullptr; a.rays = nullptr; a.z = nullptr; a.raw = raw; a.acts = acts; a.M = M; a.spr = samples_per_ray;
    return launch_mlp(a, W, as_stream(stream));
}

// north-star form of the two entry points above: the sample points are formed inside the kernel from the packed rays and the
// depths (pts = o + d * z, RN:381 / :399) instead of being read from a [M,3] tensor that the sampling kernels wrote.
extern "C" int nerfail_mlp_fwd_rays(const float* packed, int D, int W, int skip, const float* rays, const float* z_vals,
                                    int64_t n_rays, int samples_per_ray, float* raw, float* acts, void* stream) {
    NF_REQUIRE(n_rays >= 0, "n_rays is negative");
    NF_REQUIRE(samples_per_ray >= 1, "samples_per_ray must be positive");
    MlpArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    const int64_t M = n_rays * samples_per_ray;
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed != nullptr && rays != nullptr && z_vals != nullptr && raw != nullptr, "NULL pointer");
    a.packed = packed; a.pts = nullptr; a.viewdirs = nullptr; a.xemb = nullptr; a.rays = rays; a.z = z_vals; a.raw = raw; a.acts = acts;
    a.M = M; a.spr = samples_per_ray;
    return launch_mlp(a, W, as_stream(stream));
}
