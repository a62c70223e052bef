// K3 + K4, split-precision variant ("f16x3"): the fused encode + NeRF MLP forward on the fp16 matrix cores with
// fp32-equivalent results. Opt-in alternative to the exact-f32 kernel of mlp.hip (same inputs, same outputs).
//
//   a*w  ~  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          a_hi = fp16(a), a_lo = fp16(a - a_hi)   (same for w)
// Each product of two fp16 numbers is exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16, the dropped
// a_lo*w_lo term is ~2^-21 relative, so every layer is accurate to fp32-rounding level while running on the
// 16x-faster fp16 MFMA: 3 MFMAs of 32 cycles per 16 k instead of 8 f32 MFMAs of 64 cycles (5.3x fewer MFMA cycles).
// Range: weights are packed pre-scaled by 2^10 (so the fp16 "lo" parts of small weights stay normal numbers; the
// accumulator is initialised with bias * 2^10 and rescaled by 2^-10 when it is turned into the next layer's
// operands); activations of a NeRF are O(1..100), far inside fp16 range.
//
// Structure = mlp.hip: one wave owns 32 samples x all channels, layers computed transposed so the accumulator
// layout (sample on lane, channel on register) IS the next B operand: registers 8s..8s+7 of a tile, converted and
// packed, are the 8-element fp16 fragment of k16-step s (hi) and its residual (lo). 128 packed operand registers
// + 128 accumulators per wave.
// Weights: the A operand must be re-streamed for every 32 samples, 32 B per lane per 3 MFMAs (96 cycles): 4 waves
// would ask the L1 for 85 B/clk/CU. So a workgroup streams the image ONCE into an LDS ring (coalesced 16-byte
// loads, register-staged, two k16-steps ahead) and the 4 waves read their fragments from LDS (ds_read_b128,
// conflict-free lane-linear layout); one barrier per k16-step (24 MFMAs per wave).
#include "mlp_layout.h"

namespace nerfail {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kWScale = 1024.0f, kWInv = 1.0f / 1024.0f;
constexpr int kEmbK16 = 4, kDirK16 = 2;      // 64 / 32 padded encoding channels

// fp16 image: per MFMA layer, per k16-step ks, per out tile t: hi fragment then lo fragment, each [64 lanes][8 halfs]
// (1 KB). Offsets in units of 16 bytes (one lane fragment).
struct F16Layout {
    int NT, D, skip;
    unsigned off[NERFAIL_MAX_DEPTH + 2];      // [0..D-1] pts, [D] feature, [D+1] views; units: 16 B
    unsigned nk16[NERFAIL_MAX_DEPTH + 2];     // k16-steps of the layer
    unsigned total;                           // 16-byte units
};

static bool make_f16_layout(int D, int W, int skip, F16Layout& L) {
    MlpLayout M;
    if (!make_layout(D, W, skip, M)) return false;
    L.NT = M.NT; L.D = D; L.skip = M.skip;
    unsigned off = 0;
    for (int l = 0; l <= D + 1; ++l) {
        const int OT = (l == D + 1) ? M.NT / 2 : M.NT;
        int k16 = 0;
        if (l <= D - 1 && layer_has_emb(l, M.skip)) k16 += kEmbK16;
        if (l > 0) k16 += 2 * M.NT;
        if (l == D + 1) k16 += kDirK16;
        L.off[l] = off; L.nk16[l] = (unsigned)k16;
        off += (unsigned)k16 * OT * 2 * 64;
    }
    L.total = off;
    return true;
}

__device__ __forceinline__ unsigned pack_rtz(float a, float b) {
    const hp2 v = __builtin_amdgcn_cvt_pkrtz(a, b);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float half_lo(unsigned p) { return (float)__builtin_bit_cast(hp2, p)[0]; }
__device__ __forceinline__ float half_hi(unsigned p) { return (float)__builtin_bit_cast(hp2, p)[1]; }

// split 8 floats into the hi / lo fp16 fragments (4 packed registers each)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned h = pack_rtz(v[2 * p], v[2 * p + 1]);
        hi[p] = h;
        lo[p] = pack_rtz(v[2 * p] - half_lo(h), v[2 * p + 1] - half_hi(h));
    }
}

// ------------------------------------------------------------------------------------- packing
// one thread per (k16, tile, lane, element j): writes hi and lo halfs of W[row][col] * 2^10
__global__ void pack_f16_layer_kernel(const float* __restrict__ w, int out_f, int in_f, int OT, int NT, int emb_col0,
                                      int h_col0, int dir_col0, _Float16* __restrict__ img, int total) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int j = g & 7, lane = (g >> 3) & 63, rest = g >> 9;
    const int t = rest % OT;
    int ks = rest / OT;
    const int hh = lane >> 5, row = 32 * t + (lane & 31);
    int col = -1;
    if (emb_col0 >= 0) {
        if (ks < kEmbK16) { const int c = enc_channel(8 * ks + j, hh, 10); col = c < 0 ? -1 : emb_col0 + c; ks = -1; }
        else ks -= kEmbK16;
    }
    if (ks >= 0 && h_col0 >= 0) {
        if (ks < 2 * NT) { col = h_col0 + 32 * (ks >> 1) + acc_channel(8 * (ks & 1) + j, hh); ks = -1; }
        else ks -= 2 * NT;
    }
    if (ks >= 0 && dir_col0 >= 0) {
        const int c = enc_channel(8 * ks + j, hh, 4); col = c < 0 ? -1 : dir_col0 + c;
    }
    const float v = (row < out_f && col >= 0) ? w[(long)row * in_f + col] * kWScale : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    // g enumerates [ks][t][lane][j]; the image is [ks][t][hi|lo][lane][j]
    const long base = ((long)(g >> 9) * 2) * 512 + (lane * 8 + j);
    img[base] = hi;
    img[base + 512] = lo;
}

// ------------------------------------------------------------------------------------- forward kernel
struct F16Args {
    const float* packed;        // fp32 image of mlp.hip: biases and the alpha / rgb head weights
    const u32x4* img;           // fp16 hi/lo image
    const float* pts;
    const float* viewdirs;
    float* raw;
    long M;
    int spr;
    MlpLayout lay;
    F16Layout l16;
};

constexpr int kRing = 2;        // LDS ring depth in k16-steps: chunk ks+2 is only written after every wave passed barrier ks+1

template <int OT>
__device__ __forceinline__ void load_bias_scaled(f32x16 (&acc)[OT], const float* __restrict__ b, int h) {
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        const f32x4* p = reinterpret_cast<const f32x4*>(b + (t * 2 + h) * 16);
        const f32x4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        acc[t] = (f32x16){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3],
                          v2[0], v2[1], v2[2], v2[3], v3[0], v3[1], v3[2], v3[3]} * kWScale;
    }
}

// One part of a layer: NK k16-steps. The workgroup streams chunk (k16-step) after chunk of the image through the LDS
// ring; bsel(ks, hi, lo) yields this wave's B fragments of step ks.
//   ring[(slot)][t][hi|lo][lane] as u32x4; chunk = OT*2*64 u32x4 = OT KB * 2
template <int OT, int NK, typename BSel>
__device__ __forceinline__ void f16_part(f32x16 (&acc)[OT], const u32x4* __restrict__ img, u32x4* __restrict__ ring,
                                         int tid, int lane, BSel bsel) {
    constexpr int CH = OT * 2 * 64;               // u32x4 per chunk
    constexpr int PER_T = (CH + 255) / 256;       // u32x4 each thread moves per chunk (OT*128/256 = OT/2)
    const bool mover = (CH % 256 == 0) || tid < CH;   // W = 64: the 1-tile views chunk is 128 fragments
    u32x4 stage[2][PER_T];                        // register staging: two chunks in flight
    // prologue: chunks 0 and 1 -> registers; chunk 0 -> LDS
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (c < NK) {
#pragma unroll
            for (int i = 0; i < PER_T; ++i) stage[c][i] = img[(long)c * CH + (mover ? i * 256 + tid : 0)];
        }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        u32x4* slot = ring + (ks % kRing) * CH;
        // chunk ks: registers -> LDS (its global loads were issued two steps ago), then refill the stage with ks + 2
#pragma unroll
        for (int i = 0; i < PER_T; ++i)
            if (mover) slot[i * 256 + tid] = stage[ks & 1][i];
        if (ks + 2 < NK) {
#pragma unroll
            for (int i = 0; i < PER_T; ++i) stage[ks & 1][i] = img[(long)(ks + 2) * CH + (mover ? i * 256 + tid : 0)];
        }
        __syncthreads();                          // chunk ks visible to all 4 waves; the other slot is free for ks + 1
        u32x4 bhi, blo;
        bsel(ks, bhi, blo);
        const h8 bh = __builtin_bit_cast(h8, bhi), bl = __builtin_bit_cast(h8, blo);
#pragma unroll
        for (int t = 0; t < OT; ++t) {
            const h8 ah = __builtin_bit_cast(h8, slot[(t * 2 + 0) * 64 + lane]);
            const h8 al = __builtin_bit_cast(h8, slot[(t * 2 + 1) * 64 + lane]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                              // everyone is done with the ring before the next part refills it
}

template <int NT>
__global__ __launch_bounds__(256, 1) void nerf_mlp_fwd_f16_kernel(F16Args a) {
    constexpr int OTV = NT / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* ring = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ P = a.packed;
    const MlpLayout& L = a.lay;
    const F16Layout& L16 = a.l16;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    for (long rnd = 0; rnd < nrounds; ++rnd) {
        // every wave of the workgroup runs every round (barriers inside): out-of-range tiles compute on a clamped
        // sample and skip the store
        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;
        const long sraw = tile * 32 + j;
        const long s = sraw < a.M ? sraw : a.M - 1;

        float emb[32], demb[16];
        {
            const float px[3] = {a.pts[3 * s], a.pts[3 * s + 1], a.pts[3 * s + 2]};
            const float* vd = a.viewdirs + 3 * (s / a.spr);
            const float vx[3] = {vd[0], vd[1], vd[2]};
#pragma unroll
            for (int f = 0; f < 10; ++f)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float sn, cs;
                    sincosf(__fmul_rn(px[d], (float)(1 << f)), &sn, &cs);
                    emb[3 * f + d] = h ? cs : sn;
                }
            emb[30] = h ? px[1] : px[0];
            emb[31] = h ? 0.f : px[2];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float sn, cs;
                    sincosf(__fmul_rn(vx[d], (float)(1 << f)), &sn, &cs);
                    demb[3 * f + d] = h ? cs : sn;
                }
            demb[12] = h ? vx[1] : vx[0];
            demb[13] = h ? 0.f : vx[2];
            demb[14] = 0.f; demb[15] = 0.f;
        }
        // encoding operands, split once
        u32x4 ehi[kEmbK16], elo[kEmbK16], dhi[kDirK16], dlo[kDirK16];
#pragma unroll
        for (int e = 0; e < kEmbK16; ++e) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = emb[8 * e + q];
            split8(v, ehi[e], elo[e]);
        }
#pragma unroll
        for (int e = 0; e < kDirK16; ++e) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = demb[8 * e + q];
            split8(v, dhi[e], dlo[e]);
        }

        f32x16 acc[NT];
        u32x4 bh[NT][2], bl[NT][2];           // packed operands of the current layer input (hi / lo per k16-step)
        auto to_operands = [&](bool relu) {   // acc (scaled by 2^10) -> next layer's B fragments
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float x = acc[t][8 * sgrp + q] * kWInv;
                        v[q] = relu ? fmaxf(x, 0.f) : x;
                    }
                    split8(v, bh[t][sgrp], bl[t][sgrp]);
                }
        };

        // ---- layer 0
        load_bias_scaled<NT>(acc, P + L.b_off[0], h);
        f16_part<NT, kEmbK16>(acc, a.img + L16.off[0], ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = ehi[ks]; lo = elo[ks]; });
        to_operands(true);

        float alpha = 0.f;
#pragma unroll 1
        for (int l = 1; l <= L.D; ++l) {
            if (l == L.D) {   // alpha_linear on the last pts activation, fp32 VALU (operands re-expanded from hi + lo)
                const float* wa = P + L.alpha_off;
                float sacc = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int sgrp = 0; sgrp < 2; ++sgrp)
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int r = 8 * sgrp + 2 * p;
                            const float x0 = half_lo(bh[t][sgrp][p]) + half_lo(bl[t][sgrp][p]);
                            const float x1 = half_hi(bh[t][sgrp][p]) + half_hi(bl[t][sgrp][p]);
                            sacc = fmaf(wa[(t * 2 + h) * 16 + r], x0, sacc);
                            sacc = fmaf(wa[(t * 2 + h) * 16 + r + 1], x1, sacc);
                        }
                sacc += __shfl_xor(sacc, 32, 64);
                alpha = sacc + wa[NT * 32];
            }
            load_bias_scaled<NT>(acc, P + L.b_off[l], h);
            const u32x4* img = a.img + L16.off[l];
            if (l == L.skip + 1 && L.skip >= 0) {
                f16_part<NT, kEmbK16>(acc, img, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = ehi[ks]; lo = elo[ks]; });
                img += kEmbK16 * NT * 2 * 64;
            }
            f16_part<NT, 2 * NT>(acc, img, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
            to_operands(l < L.D);
        }

        // ---- views_linears[0]
        f32x16 hv[OTV];
        load_bias_scaled<OTV>(hv, P + L.b_off[L.D + 1], h);
        {
            const u32x4* img = a.img + L16.off[L.D + 1];
            f16_part<OTV, 2 * NT>(hv, img, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
            f16_part<OTV, kDirK16>(hv, img + 2 * NT * OTV * 2 * 64, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = dhi[ks]; lo = dlo[ks]; });
        }
        // ---- rgb_linear on VALU (fp32)
        const float* wr = P + L.rgb_off;
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sacc = 0.f;
#pragma unroll
            for (int t = 0; t < OTV; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc = fmaf(wr[((c * OTV + t) * 2 + h) * 16 + r], fmaxf(hv[t][r] * kWInv, 0.f), sacc);
            sacc += __shfl_xor(sacc, 32, 64);
            rgb[c] = sacc + wr[3 * OTV * 32 + c];
        }
        if (h == 0 && sraw < a.M)
            reinterpret_cast<float4*>(a.raw)[sraw] = make_float4(rgb[0], rgb[1], rgb[2], alpha);
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" size_t nerfail_mlp_f16_image_bytes(int D, int W, int skip) {
    F16Layout L;
    return make_f16_layout(D, W, skip, L) ? (size_t)L.total * 16 : 0;
}

extern "C" int nerfail_mlp_pack_f16(const nerfail_mlp_params* p, void* image, void* stream) {
    NF_REQUIRE(p != nullptr && image != nullptr, "NULL pointer");
    NF_REQUIRE(p->input_ch == kPtsCh && p->input_ch_views == kDirCh, "only multires=10 / multires_views=4 (63 + 27 channels)");
    F16Layout L;
    NF_REQUIRE(make_f16_layout(p->D, p->W, p->skip, L), "unsupported (D, W)");
    hipStream_t s = as_stream(stream);
    const int W = p->W, NT = L.NT, OTV = NT / 2;
    for (int l = 0; l <= p->D + 1; ++l) {
        const bool emb = l <= p->D - 1 && layer_has_emb(l, L.skip);
        const float* w;
        int out_f, in_f, OT = NT, emb0 = -1, h0 = -1, dir0 = -1;
        if (l < p->D) {
            NF_REQUIRE(p->pts_w[l] != nullptr, "pts_linears pointer is NULL");
            w = p->pts_w[l]; out_f = W;
            in_f = (l == 0) ? kPtsCh : (emb ? W + kPtsCh : W);
            if (emb) emb0 = 0;
            if (l > 0) h0 = emb ? kPtsCh : 0;
        } else if (l == p->D) {
            NF_REQUIRE(p->feature_w != nullptr, "feature_linear pointer is NULL");
            w = p->feature_w; out_f = W; in_f = W; h0 = 0;
        } else {
            NF_REQUIRE(p->views_w != nullptr, "views_linears pointer is NULL");
            w = p->views_w; out_f = W / 2; in_f = W + kDirCh; OT = OTV; h0 = 0; dir0 = W;
        }
        const int total = (int)L.nk16[l] * OT * 512;     // one thread per (k16, tile, lane, j)
        pack_f16_layer_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(
            w, out_f, in_f, OT, NT, emb0, h0, dir0, reinterpret_cast<_Float16*>(image) + (size_t)L.off[l] * 8, total);
        NF_LAUNCHED("pack_f16_layer_kernel");
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_fwd_f16(const float* packed, const void* image, int D, int W, int skip, const float* pts,
                                   const float* viewdirs, int64_t M, int samples_per_ray, float* raw, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    NF_REQUIRE(samples_per_ray >= 1, "samples_per_ray must be positive");
    F16Args a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay) && make_f16_layout(D, W, skip, a.l16), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed && image && pts && viewdirs && raw, "NULL pointer");
    a.packed = packed; a.img = reinterpret_cast<const u32x4*>(image); a.pts = pts; a.viewdirs = viewdirs; a.raw = raw;
    a.M = M; a.spr = samples_per_ray;
    const long ntiles = (M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = as_stream(stream);
    const size_t lds = (size_t)kRing * a.lay.NT * 2 * 64 * 16;      // ring of kRing chunks of NT*2 KB
    switch (W) {
        case 256: nerf_mlp_fwd_f16_kernel<8><<<grid, block, lds, s>>>(a); break;
        case 128: nerf_mlp_fwd_f16_kernel<4><<<grid, block, lds, s>>>(a); break;
        case 64: nerf_mlp_fwd_f16_kernel<2><<<grid, block, lds, s>>>(a); break;
        default: set_error("nerfail_mlp_fwd_f16: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_fwd_f16_kernel");
    return NERFAIL_OK;
}
