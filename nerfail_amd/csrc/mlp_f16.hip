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
// would ask the L1 for 85 B/clk/CU. So the WORKGROUP streams the image once through LDS and its 4 waves read their
// fragments from there (ds_read_b128, lane-linear = conflict-free). The image is one continuous sequence of
// fixed-size chunks (NT*2 KB = one k16-step x NT out tiles, or two steps x NT/2 tiles for the views layer) in
// consumption order over the WHOLE network; the stream never stops at a layer or tile boundary:
//     step c, phase A:  ds_read 2nd half of chunk c | ds_write chunk c+1 | global loads of chunk c+3
//                       MFMAs on the 1st half of chunk c (fragments read during the previous step)     -> barrier
//             phase B:  ds_read 1st half of chunk c+1 (now visible) | MFMAs on the 2nd half of chunk c
// so every LDS read has 12 MFMAs (~400 cycles) of cover and the LDS writes / global loads ride under the MFMAs.
// 2-slot LDS ring, 2 register stages; every part has an even chunk count (views padded), so ring slot and stage
// index are compile-time constants.
#include "mlp_layout.h"

namespace nerfail {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kWScale = 1024.0f, kWInv = 1.0f / 1024.0f;
constexpr int kEmbK16 = 4, kDirK16 = 2;      // 64 / 32 padded encoding channels

// fp16 image: a sequence of chunks of NT*2*64 fragments (16 B each). A chunk = [step-in-chunk][tile][hi|lo][lane],
// steps per chunk = 1 for the NT-tile layers and 2 for the NT/2-tile views layer. Per layer the chunk count is padded
// to an even number (zero chunks that are streamed but not computed).
struct F16Layout {
    int NT, D, skip;
    unsigned chunk0[NERFAIL_MAX_DEPTH + 2];   // first chunk of layer l ([0..D-1] pts, [D] feature, [D+1] views)
    unsigned nchunks[NERFAIL_MAX_DEPTH + 2];  // padded chunk count of layer l
    unsigned total_chunks;
};

static bool make_f16_layout(int D, int W, int skip, F16Layout& L) {
    MlpLayout M;
    if (!make_layout(D, W, skip, M)) return false;
    L.NT = M.NT; L.D = D; L.skip = M.skip;
    unsigned c = 0;
    for (int l = 0; l <= D + 1; ++l) {
        int k16 = 0;
        if (l <= D - 1 && layer_has_emb(l, M.skip)) k16 += kEmbK16;
        if (l > 0) k16 += 2 * M.NT;
        if (l == D + 1) k16 += kDirK16;
        int n = (l == D + 1) ? (k16 + 1) / 2 : k16;
        n = (n + 1) & ~1;
        L.chunk0[l] = c; L.nchunks[l] = (unsigned)n;
        c += (unsigned)n;
    }
    L.total_chunks = c;
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
// one thread per (k16-step, tile, lane, element j) of ONE layer: writes hi and lo halfs of W[row][col] * 2^10.
// spc = steps per chunk (1: OT == NT, 2: OT == NT/2); chunk bytes = NT*2 KB either way.
__global__ void pack_f16_layer_kernel(const float* __restrict__ w, int out_f, int in_f, int OT, int NT, int emb_col0,
                                      int h_col0, int dir_col0, _Float16* __restrict__ img, int total) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int j = g & 7, lane = (g >> 3) & 63, rest = g >> 9;
    const int t = rest % OT;
    const int step = rest / OT;
    int ks = step;
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
    // image position: chunk = step / spc; inside the chunk [step % spc][t][hi|lo][lane][j]
    const int spc = NT / OT;
    const long chunk = step / spc;
    const long in_chunk = ((long)(step % spc) * OT + t) * 2;
    const long base = (chunk * NT * 2 + in_chunk) * 512 + (lane * 8 + j);    // halfs
    img[base] = hi;
    img[base + 512] = lo;
}

// the whole image in ONE launch: per-layer descriptors by value, a block row (blockIdx.y) per layer
struct PackF16Desc { const float* w; int out_f, in_f, OT, emb0, h0, dir0, total; unsigned long dst_halfs; };
struct PackF16Table { int n, NT; PackF16Desc l[NERFAIL_MAX_DEPTH + 2]; };

__global__ void pack_f16_all_kernel(PackF16Table t, _Float16* __restrict__ image) {
    const PackF16Desc& d = t.l[blockIdx.y];
    _Float16* __restrict__ img = image + d.dst_halfs;
    const int NT = t.NT, OT = d.OT;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < d.total; g += gridDim.x * blockDim.x) {
        const int j = g & 7, lane = (g >> 3) & 63, rest = g >> 9;
        const int tt = rest % OT;
        const int step = rest / OT;
        int ks = step;
        const int hh = lane >> 5, row = 32 * tt + (lane & 31);
        int col = -1;
        if (d.emb0 >= 0) {
            if (ks < kEmbK16) { const int c = enc_channel(8 * ks + j, hh, 10); col = c < 0 ? -1 : d.emb0 + c; ks = -1; }
            else ks -= kEmbK16;
        }
        if (ks >= 0 && d.h0 >= 0) {
            if (ks < 2 * NT) { col = d.h0 + 32 * (ks >> 1) + acc_channel(8 * (ks & 1) + j, hh); ks = -1; }
            else ks -= 2 * NT;
        }
        if (ks >= 0 && d.dir0 >= 0) {
            const int c = enc_channel(8 * ks + j, hh, 4); col = c < 0 ? -1 : d.dir0 + c;
        }
        const float v = (row < d.out_f && col >= 0) ? d.w[(long)row * d.in_f + col] * kWScale : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const int spc = NT / OT;
        const long chunk = step / spc;
        const long in_chunk = ((long)(step % spc) * OT + tt) * 2;
        const long base = (chunk * NT * 2 + in_chunk) * 512 + (lane * 8 + j);
        img[base] = hi;
        img[base + 512] = lo;
    }
}

// ------------------------------------------------------------------------------------- forward kernel
struct F16Args {
    const float* packed;        // fp32 image of mlp.hip: biases and the alpha / rgb head weights
    const u32x4* img;           // fp16 hi/lo image
    const float* pts;
    const float* viewdirs;
    float* raw;
    float* acts;                // training: channel-major activation tiles, same layout as nerfail_mlp_fwd_train
    long M;
    int spr;
    MlpLayout lay;
    F16Layout l16;
};

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

// The weight stream of one workgroup. CH = fragments per chunk, PER_T = fragments each thread moves per chunk.
// A chunk holds NT (hi, lo) fragment pairs: pair p = [step-in-chunk][tile]; "first half" = pairs 0..NT/2-1.
template <int NT>
struct WStream {
    static constexpr int CH = NT * 2 * 64;
    static constexpr int PER_T = CH / 256;
    static constexpr int HP = NT / 2;      // fragment pairs per half chunk
    const u32x4* gp;          // this lane's global read position: chunk (c+3) when step c starts
    const u32x4* img;         // image start (lane-adjusted)
    unsigned left;            // chunks until the image wraps
    unsigned total;
    u32x4 st[2][PER_T];       // register stages: st[x] holds the chunk with parity x that is next to be written
    u32x4 fa[HP][2];          // first-half fragments of the CURRENT chunk (prefetched during the previous step)
};

template <int NT>
__device__ __forceinline__ void stream_load(WStream<NT>& w, u32x4 (&dst)[WStream<NT>::PER_T]) {
#pragma unroll
    for (int i = 0; i < WStream<NT>::PER_T; ++i) dst[i] = w.gp[i * 256];
    w.gp += WStream<NT>::CH;
    if (--w.left == 0) { w.gp = w.img; w.left = w.total; }       // next tile starts over (workgroup-uniform)
}

template <int NT>
__device__ __forceinline__ void read_half(u32x4 (&dst)[NT / 2][2], const u32x4* __restrict__ rp, int slot, int half) {
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
        dst[p][0] = rp[slot * WStream<NT>::CH + ((half * (NT / 2) + p) * 2 + 0) * 64];
        dst[p][1] = rp[slot * WStream<NT>::CH + ((half * (NT / 2) + p) * 2 + 1) * 64];
    }
}

// One part of a layer: NCH chunks (even), each holding SPC k16-steps for OT = NT/SPC out tiles; the last PADC chunks
// are zero padding (streamed, not computed). Entry invariant = exit invariant: chunk 0 of the part is visible in LDS
// slot 0, its first-half fragments are in w.fa, st[1] holds chunk 1, st[0] holds chunk 2 (loads in flight).
template <int NT, int SPC, int NCH, int PADC, typename BSel>
__device__ __forceinline__ void f16_part(f32x16 (&acc)[NT / SPC], WStream<NT>& w, u32x4* __restrict__ ring, int tid, int lane,
                                         BSel bsel) {
    constexpr int OT = NT / SPC, CH = WStream<NT>::CH, PER_T = WStream<NT>::PER_T, HP = NT / 2;
    static_assert(NCH % 2 == 0, "parts must have an even chunk count");
    static_assert(SPC == 1 || SPC == 2, "one or two k16-steps per chunk");
    u32x4* __restrict__ wp = ring + tid;            // lane pointers: every LDS / global access is pointer + constant
    const u32x4* __restrict__ rp = ring + lane;
    // pair p of a chunk -> (step-in-chunk, tile): SPC == 1: (0, p); SPC == 2: (p / OT, p % OT) with OT == HP
    auto mfma_half = [&](int c, int half, const u32x4 (&fr)[HP][2]) {
#pragma unroll
        for (int x = 0; x < 3; ++x)                 // x-outer: 3 independent rounds over the tiles (hh, hl, lh)
#pragma unroll
            for (int p = 0; p < HP; ++p) {
                const int pair = half * HP + p;
                const int sp = (SPC == 1) ? 0 : pair / OT, t = (SPC == 1) ? pair : pair % OT;
                u32x4 bhi, blo;
                bsel(c * SPC + sp, bhi, blo);
                const h8 ah = __builtin_bit_cast(h8, fr[p][0]), al = __builtin_bit_cast(h8, fr[p][1]);
                const h8 bh = __builtin_bit_cast(h8, bhi), bl = __builtin_bit_cast(h8, blo);
                if (x == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                else if (x == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            }
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bool live = c < NCH - PADC;
        // ---- phase A
        u32x4 fb[HP][2];
        read_half<NT>(fb, rp, c & 1, 1);            // second half of chunk c
#pragma unroll
        for (int i = 0; i < PER_T; ++i) wp[((c + 1) & 1) * CH + i * 256] = w.st[(c + 1) & 1][i];   // chunk c+1 -> LDS
        stream_load<NT>(w, w.st[(c + 1) & 1]);      // chunk c+3 -> the stage just freed
        if (live) mfma_half(c, 0, w.fa);
        __syncthreads();                        // chunk c+1 visible; nobody reads slot (c+1)&1's old content any more
        // ---- phase B
        read_half<NT>(w.fa, rp, (c + 1) & 1, 0);    // first half of chunk c+1 (possibly the next part's chunk 0)
        if (live) mfma_half(c, 1, fb);
    }
}

template <int NT, bool TRAIN>
__global__ __launch_bounds__(256, 1) void nerf_mlp_fwd_f16_kernel(F16Args a) {
    constexpr int OTV = NT / 2;
    __shared__ __attribute__((aligned(16))) u32x4 ring_s[2 * WStream<NT>::CH];        // 2-slot weight ring
    // The 12 encoding operands of a lane (pts: 4 k16-steps hi + lo, dirs: 2 + 2) are needed at layer 0, again at the
    // skip layer and at the view layer. Parked in LDS in between they free 48 VGPRs inside the layer loop, where the
    // compiler otherwise spilled ~30 registers per iteration to scratch - and every scratch reload queues behind the
    // weight stream's global loads (vmcnt retires in order).
    __shared__ __attribute__((aligned(16))) u32x4 s_enc[4][2 * kEmbK16 + 2 * kDirK16][64];
    u32x4* ring = ring_s;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int kViewsSteps = 2 * NT + kDirK16;                     // k16-steps of the views layer
    constexpr int kViewsChunks = (((kViewsSteps + 1) / 2) + 1) & ~1;  // 2 steps per chunk, padded to even
    constexpr int kViewsPad = kViewsChunks - (kViewsSteps + 1) / 2;
    static_assert(kViewsSteps % 2 == 0, "views k16-steps fill whole chunks");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ P = a.packed;
    const MlpLayout& L = a.lay;
    const F16Layout& L16 = a.l16;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    // ---- start the weight stream: chunk 0 -> LDS slot 0, chunks 1 and 2 -> register stages
    WStream<NT> ws;
    ws.img = a.img + tid; ws.gp = ws.img; ws.total = L16.total_chunks; ws.left = ws.total;
    stream_load<NT>(ws, ws.st[0]);
#pragma unroll
    for (int i = 0; i < WStream<NT>::PER_T; ++i) ring[tid + i * 256] = ws.st[0][i];
    stream_load<NT>(ws, ws.st[1]);
    stream_load<NT>(ws, ws.st[0]);
    __syncthreads();
    read_half<NT>(ws.fa, ring + lane, 0, 0);

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
            for (int d = 0; d < 3; ++d) {
                const SinCosBands sc(px[d]);
#pragma unroll
                for (int f = 0; f < 10; ++f) {
                    float sn, cs;
                    sc.band(f, sn, cs);
                    emb[3 * f + d] = h ? cs : sn;
                }
            }
            emb[30] = h ? px[1] : px[0];
            emb[31] = h ? 0.f : px[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const SinCosBands sc(vx[d]);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float sn, cs;
                    sc.band(f, sn, cs);
                    demb[3 * f + d] = h ? cs : sn;
                }
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
#pragma unroll
        for (int e = 0; e < kEmbK16; ++e) { s_enc[wave][e][lane] = ehi[e]; s_enc[wave][kEmbK16 + e][lane] = elo[e]; }
#pragma unroll
        for (int e = 0; e < kDirK16; ++e) { s_enc[wave][2 * kEmbK16 + e][lane] = dhi[e]; s_enc[wave][2 * kEmbK16 + kDirK16 + e][lane] = dlo[e]; }

        f32x16 acc[NT];
        u32x4 bh[NT][2], bl[NT][2];           // packed operands of the current layer input (hi / lo per k16-step)
        const bool save = TRAIN && tile < ntiles;                    // wave-uniform
        float* __restrict__ A = nullptr;                             // this tile's activation slots (training)
        float* __restrict__ lpA = nullptr;                           // lane pointer for accumulator-layout stores
        if (TRAIN && save) {
            A = a.acts + (size_t)tile * (train_a_slots(L.D, NT) * 1024);
            lpA = A + acc_lane_off(lane);
            store_enc<10, 32>(A, emb, lane);
            store_enc<4, 16>(A + 2 * 1024, demb, lane);
        }
        // acc (scaled by 2^10) -> next layer's B fragments; training also saves the fp32 activation (slot0 = first slot)
        auto to_operands = [&](bool relu, int slot0) {
            TileMask<NT> mk;
#pragma unroll
            for (int i = 0; i < (NT + 1) / 2; ++i) mk.w[i] = 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float x = acc[t][8 * sgrp + q] * kWInv;
                        v[q] = relu ? fmaxf(x, 0.f) : x;
                        if (TRAIN && save) {
                            const int r = 8 * sgrp + q;
                            slot_store(lpA + (slot0 + t) * 1024 + acc_reg_off(r), v[q]);
                            mk.w[t >> 1] |= relu_bit(v[q]) << (16 * (t & 1) + r);
                        }
                    }
                    split8(v, bh[t][sgrp], bl[t][sgrp]);
                }
            if (TRAIN && save && relu) store_mask<NT>(A + train_mask_slot0(L.D, NT) * 1024, (slot0 - 3) / NT, mk, lane);
        };

        // ---- layer 0
        load_bias_scaled<NT>(acc, P + L.b_off[0], h);
        f16_part<NT, 1, kEmbK16, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = ehi[ks]; lo = elo[ks]; });
        to_operands(true, 3);

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
            if (l == L.skip + 1 && L.skip >= 0)
                f16_part<NT, 1, kEmbK16, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = s_enc[wave][ks][lane]; lo = s_enc[wave][kEmbK16 + ks][lane]; });
            f16_part<NT, 1, 2 * NT, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
            to_operands(l < L.D, 3 + l * NT);      // H_{l+1} for l < D, F for l == D
        }

        // ---- views_linears[0]
        f32x16 hv[OTV];
        load_bias_scaled<OTV>(hv, P + L.b_off[L.D + 1], h);
        f16_part<NT, 2, kViewsChunks, kViewsPad>(hv, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) {
            if (ks < 2 * NT) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; }
            else { hi = s_enc[wave][2 * kEmbK16 + ks - 2 * NT][lane]; lo = s_enc[wave][2 * kEmbK16 + kDirK16 + ks - 2 * NT][lane]; }
        });
        if (TRAIN && save) {
            TileMask<OTV> mk;
#pragma unroll
            for (int i = 0; i < (OTV + 1) / 2; ++i) mk.w[i] = 0u;
#pragma unroll
            for (int t = 0; t < OTV; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = fmaxf(hv[t][r] * kWInv, 0.f);
                    slot_store(lpA + (3 + (L.D + 1) * NT + t) * 1024 + acc_reg_off(r), x);
                    mk.w[t >> 1] |= relu_bit(x) << (16 * (t & 1) + r);
                }
            store_mask<OTV>(A + train_mask_slot0(L.D, NT) * 1024, L.D, mk, lane);
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

// ------------------------------------------------------------------------------------- backward data (f16x3)
// dX = W^T dZ with the same split-precision scheme and the same LDS weight stream. The W^T image is a chunk sequence
// in BACKWARD order: views^T (K = W/2), feature^T, pts_{D-1}^T .. pts_1^T (K = W each), all NT in-tiles wide.
// Gradients are tiny (d_raw ~ 1e-5), so each wave scales its d_raw by an exact power of two S (max |d_raw| -> [1,2));
// every dZ is stored divided by S again, i.e. unscaled and bit-comparable with the f32 kernel's layout.
struct F16LayoutT {
    unsigned chunk0[NERFAIL_MAX_DEPTH + 2];   // index: 0 = views^T, 1 = feature^T, 2 + k = pts_{D-1-k}^T (k = 0 .. D-2)
    unsigned total_chunks;
};
static void make_f16_layout_T(int D, int NT, F16LayoutT& L) {
    unsigned c = 0;
    L.chunk0[0] = c; c += 2 * (NT / 2);
    for (int k = 1; k <= D; ++k) { L.chunk0[k] = c; c += 2 * NT; }
    L.total_chunks = c;
}

// one thread per (k16-step, in-tile, lane, j): A[row = input channel 32t + lane&31][k = out channel of slot (ks, hh, j)]
__global__ void pack_f16_layer_T_kernel(const float* __restrict__ w, int out_f, int in_f, int col0, int NT,
                                        _Float16* __restrict__ img, int total) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int j = g & 7, lane = (g >> 3) & 63, rest = g >> 9;
    const int t = rest % NT, ks = rest / NT;
    const int hh = lane >> 5;
    const int o = 32 * (ks >> 1) + acc_channel(8 * (ks & 1) + j, hh);
    const int i = 32 * t + (lane & 31);
    const float v = (o < out_f) ? w[(long)o * in_f + col0 + i] * kWScale : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const long base = ((long)ks * NT * 2 + (long)t * 2) * 512 + (lane * 8 + j);
    img[base] = hi;
    img[base + 512] = lo;
}

struct F16BwdArgs {
    const float* packed;        // fp32 image (alpha / rgb head weights)
    const u32x4* imgT;          // fp16 hi/lo image of the transposed weights
    const float* d_raw;
    const float* acts;
    float* dz;
    long M;
    unsigned total_chunks;
    MlpLayout lay;
    TrainLayout tl;
};

template <int NT>
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_data_f16_kernel(F16BwdArgs a) {
    constexpr int OTV = NT / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* ring = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ P = a.packed;
    const MlpLayout& L = a.lay;
    const TrainLayout& TL = a.tl;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    WStream<NT> ws;
    ws.img = a.imgT + tid; ws.gp = ws.img; ws.total = a.total_chunks; ws.left = ws.total;
    stream_load<NT>(ws, ws.st[0]);
#pragma unroll
    for (int i = 0; i < WStream<NT>::PER_T; ++i) ring[tid + i * 256] = ws.st[0][i];
    stream_load<NT>(ws, ws.st[1]);
    stream_load<NT>(ws, ws.st[0]);
    __syncthreads();
    read_half<NT>(ws.fa, ring + lane, 0, 0);

    for (long rnd = 0; rnd < nrounds; ++rnd) {
        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;
        const bool live = tile < ntiles;                                   // wave-uniform; dead waves still run the barriers
        const long tl_ = live ? tile : ntiles - 1;
        const long sraw = tl_ * 32 + j;
        const float* __restrict__ A = a.acts + (size_t)tl_ * TL.a_slots * 1024;
        float* __restrict__ Z = a.dz + (size_t)tl_ * TL.z_slots * 1024;
        float* __restrict__ lpZ = Z + acc_lane_off(lane);
        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && sraw < a.M) dr = reinterpret_cast<const float4*>(a.d_raw)[sraw];
        // Exact power-of-two scale PER SAMPLE (a sample is a column of the B operand, so scaling it scales the same
        // column of the result), renewed at every layer: gradients shrink from layer to layer and differ by orders
        // of magnitude between the samples of a ray, and an fp16 lo half below 2^-14 loses its bits to the subnormal
        // grid. A fixed per-wave scale cost a factor ~2 of gradient accuracy per layer (tools/grad_accuracy.py).
        int E = 0;                                                         // S = 2^E, kept inside [2^-100, 2^100]
        {
            const float m = fmaxf(fmaxf(fabsf(dr.x), fabsf(dr.y)), fmaxf(fabsf(dr.z), fabsf(dr.w)));
            int ex = 10;
            if (m > 0.f) (void)frexpf(m, &ex);                             // m = f * 2^ex, f in [0.5, 1)
            E = 10 - ex;                                                   // max |d_raw| -> [2^9, 2^10): 2^6 of headroom for W_rgb
            E = E > 100 ? 100 : (E < -100 ? -100 : E);
        }
        float S = ldexpf(1.0f, E), Sinv = ldexpf(1.0f, -E);
        if (live) {   // ZR: d_raw as a tile (channels 0..3 in half 0, registers 0..3)
            f32x16 zr[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[0][r] = 0.f;
            if (h == 0) { zr[0][0] = dr.x; zr[0][1] = dr.y; zr[0][2] = dr.z; zr[0][3] = dr.w; }
            store_tiles<1>(Z + TL.z_ZR * 1024, zr, lane);
        }
        u32x4 bh[NT][2], bl[NT][2];
        f32x16 acc[NT];
        // ---- rgb_linear backward (fp32 VALU): dZ_v = (W_rgb^T d_rgb) * [hv > 0]; operands of the views^T part
        {
            const float* wr = P + L.rgb_off;
            const TileMask<OTV> mhv = load_mask<OTV>(A + TL.a_MASK * 1024, L.D, lane);
#pragma unroll
            for (int t = 0; t < OTV; ++t) {
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * sgrp + q;
                        const float g = wr[((0 * OTV + t) * 2 + h) * 16 + r] * dr.x + wr[((1 * OTV + t) * 2 + h) * 16 + r] * dr.y +
                                        wr[((2 * OTV + t) * 2 + h) * 16 + r] * dr.z;
                        const float dzv = mask_apply<OTV>(mhv, t, r, g);
                        if (live) slot_store(lpZ + (TL.z_ZV + t) * 1024 + acc_reg_off(r), dzv);
                        v[q] = dzv * S;
                    }
                    split8(v, bh[t][sgrp], bl[t][sgrp]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        f16_part<NT, 1, 2 * OTV, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
        // acc = d_feature * S * 2^10: store dZ_F (unscaled), operands for feature^T
        auto emit = [&](int slot0, int mask_entry) {   // acc -> (masked) dZ: store unscaled, split rescaled; mask_entry < 0: none
            float m = 0.f;
            TileMask<NT> mk;
            if (mask_entry >= 0) mk = load_mask<NT>(A + TL.a_MASK * 1024, mask_entry, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[t][r] * kWInv;                            // dZ * S
                    if (mask_entry >= 0) x = mask_apply<NT>(mk, t, r, x);
                    acc[t][r] = x;
                    m = fmaxf(m, fabsf(x));
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32));                                // the two lanes holding this sample's channels
            int ex = 14;                                                    // all-zero sample: keep the scale
            if (m > 0.f) (void)frexpf(m, &ex);
            int k = 14 - ex;                                                // sample max -> [2^13, 2^14)
            k = E + k > 100 ? 100 - E : (E + k < -100 ? -100 - E : k);
            E += k;
            const float up = ldexpf(1.0f, k), Sinv_old = Sinv;
            S = ldexpf(1.0f, E); Sinv = ldexpf(1.0f, -E);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int sgrp = 0; sgrp < 2; ++sgrp) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * sgrp + q;
                        const float x = acc[t][r];
                        if (live) slot_store(lpZ + (slot0 + t) * 1024 + acc_reg_off(r), x * Sinv_old);
                        v[q] = x * up;
                    }
                    split8(v, bh[t][sgrp], bl[t][sgrp]);
                }
        };
        emit(TL.z_ZF, -1);
        // ---- feature^T + alpha: d_h = Wf^T d_feature + w_alpha * d_sigma
        {
            const float* wa = P + L.alpha_off;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = wa[(t * 2 + h) * 16 + r] * (dr.w * S * kWScale);
        }
        f16_part<NT, 1, 2 * NT, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
        // ---- pts_linears[D-1 .. 0]
#pragma unroll 1
        for (int i = L.D - 1; i >= 0; --i) {
            emit(TL.z_Z0 + i * NT, i);                                    // dZ_i = d_h_{i+1} * [h_{i+1} > 0]
            if (i == 0) break;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            f16_part<NT, 1, 2 * NT, 0>(acc, ws, ring, tid, lane, [&](int ks, u32x4& hi, u32x4& lo) { hi = bh[ks >> 1][ks & 1]; lo = bl[ks >> 1][ks & 1]; });
        }
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" size_t nerfail_mlp_f16_image_bytes(int D, int W, int skip) {
    F16Layout L;
    return make_f16_layout(D, W, skip, L) ? (size_t)L.total_chunks * L.NT * 2 * 64 * 16 : 0;
}

extern "C" int nerfail_mlp_pack_f16(const nerfail_mlp_params* p, void* image, void* stream) {
    NF_REQUIRE(p != nullptr && image != nullptr, "NULL pointer");
    NF_REQUIRE(p->input_ch == kPtsCh && p->input_ch_views == kDirCh, "only multires=10 / multires_views=4 (63 + 27 channels)");
    F16Layout L;
    NF_REQUIRE(make_f16_layout(p->D, p->W, p->skip, L), "unsupported (D, W)");
    hipStream_t s = as_stream(stream);
    const int W = p->W, NT = L.NT, OTV = NT / 2;
    {   // padding chunks must be zero (they are streamed, never multiplied, but keep the image deterministic)
        hipError_t e = hipMemsetAsync(image, 0, (size_t)L.total_chunks * NT * 2 * 64 * 16, s);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    }
    PackF16Table tab;
    tab.n = p->D + 2; tab.NT = NT;
    for (int l = 0; l <= p->D + 1; ++l) {
        const bool emb = l <= p->D - 1 && layer_has_emb(l, L.skip);
        PackF16Desc& d = tab.l[l];
        d.OT = NT; d.emb0 = -1; d.h0 = -1; d.dir0 = -1;
        if (l < p->D) {
            NF_REQUIRE(p->pts_w[l] != nullptr, "pts_linears pointer is NULL");
            d.w = p->pts_w[l]; d.out_f = W;
            d.in_f = (l == 0) ? kPtsCh : (emb ? W + kPtsCh : W);
            if (emb) d.emb0 = 0;
            if (l > 0) d.h0 = emb ? kPtsCh : 0;
        } else if (l == p->D) {
            NF_REQUIRE(p->feature_w != nullptr, "feature_linear pointer is NULL");
            d.w = p->feature_w; d.out_f = W; d.in_f = W; d.h0 = 0;
        } else {
            NF_REQUIRE(p->views_w != nullptr, "views_linears pointer is NULL");
            d.w = p->views_w; d.out_f = W / 2; d.in_f = W + kDirCh; d.OT = OTV; d.h0 = 0; d.dir0 = W;
        }
        int k16 = 0;
        if (d.emb0 >= 0) k16 += kEmbK16;
        if (d.h0 >= 0) k16 += 2 * NT;
        if (d.dir0 >= 0) k16 += kDirK16;
        d.total = k16 * d.OT * 512;                      // one element per (k16-step, tile, lane, j)
        d.dst_halfs = (unsigned long)L.chunk0[l] * NT * 2 * 512;
    }
    pack_f16_all_kernel<<<dim3(64, (unsigned)tab.n), dim3(256), 0, s>>>(tab, reinterpret_cast<_Float16*>(image));
    NF_LAUNCHED("pack_f16_all_kernel");
    return NERFAIL_OK;
}

static int launch_f16(F16Args& a, int W, hipStream_t s) {
    const long ntiles = (a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = 0;                                           // ring and parking lot are static LDS of the kernel
    const bool train = a.acts != nullptr;
    switch (W) {
        case 256: if (train) nerf_mlp_fwd_f16_kernel<8, true><<<grid, block, lds, s>>>(a); else nerf_mlp_fwd_f16_kernel<8, false><<<grid, block, lds, s>>>(a); break;
        case 128: if (train) nerf_mlp_fwd_f16_kernel<4, true><<<grid, block, lds, s>>>(a); else nerf_mlp_fwd_f16_kernel<4, false><<<grid, block, lds, s>>>(a); break;
        case 64: if (train) nerf_mlp_fwd_f16_kernel<2, true><<<grid, block, lds, s>>>(a); else nerf_mlp_fwd_f16_kernel<2, false><<<grid, block, lds, s>>>(a); break;
        default: set_error("nerfail_mlp_fwd_f16: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_fwd_f16_kernel");
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
    a.acts = nullptr; a.M = M; a.spr = samples_per_ray;
    return launch_f16(a, W, as_stream(stream));
}

extern "C" int nerfail_mlp_fwd_f16_train(const float* packed, const void* image, int D, int W, int skip, const float* pts,
                                         const float* viewdirs, int64_t M, int samples_per_ray, float* raw, float* acts,
                                         void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    NF_REQUIRE(samples_per_ray >= 1, "samples_per_ray must be positive");
    F16Args a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay) && make_f16_layout(D, W, skip, a.l16), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed && image && pts && viewdirs && raw && acts, "NULL pointer");
    a.packed = packed; a.img = reinterpret_cast<const u32x4*>(image); a.pts = pts; a.viewdirs = viewdirs; a.raw = raw;
    a.acts = acts; a.M = M; a.spr = samples_per_ray;
    return launch_f16(a, W, as_stream(stream));
}

extern "C" size_t nerfail_mlp_f16_image_T_bytes(int D, int W, int skip) {
    F16Layout L;
    if (!make_f16_layout(D, W, skip, L)) return 0;
    F16LayoutT T;
    make_f16_layout_T(D, L.NT, T);
    return (size_t)T.total_chunks * L.NT * 2 * 64 * 16;
}

extern "C" int nerfail_mlp_pack_f16_T(const nerfail_mlp_params* p, void* image, void* stream) {
    NF_REQUIRE(p != nullptr && image != nullptr, "NULL pointer");
    F16Layout L;
    NF_REQUIRE(make_f16_layout(p->D, p->W, p->skip, L), "unsupported (D, W)");
    F16LayoutT T;
    make_f16_layout_T(p->D, L.NT, T);
    hipStream_t s = as_stream(stream);
    const int W = p->W, NT = L.NT;
    for (int part = 0; part <= p->D; ++part) {
        const float* w;
        int out_f, in_f, col0 = 0;
        if (part == 0) {
            NF_REQUIRE(p->views_w != nullptr, "views_linears pointer is NULL");
            w = p->views_w; out_f = W / 2; in_f = W + kDirCh;
        } else if (part == 1) {
            NF_REQUIRE(p->feature_w != nullptr, "feature_linear pointer is NULL");
            w = p->feature_w; out_f = W; in_f = W;
        } else {
            const int l = p->D - 1 - (part - 2);           // pts layer D-1 .. 1
            NF_REQUIRE(p->pts_w[l] != nullptr, "pts_linears pointer is NULL");
            const bool emb = layer_has_emb(l, L.skip);
            w = p->pts_w[l]; out_f = W; in_f = emb ? W + kPtsCh : W; col0 = emb ? kPtsCh : 0;
        }
        const int k16 = (part == 0) ? 2 * (NT / 2) : 2 * NT;
        const int total = k16 * NT * 512;
        pack_f16_layer_T_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(
            w, out_f, in_f, col0, NT, reinterpret_cast<_Float16*>(image) + (size_t)T.chunk0[part] * NT * 2 * 512, total);
        NF_LAUNCHED("pack_f16_layer_T_kernel");
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_mlp_bwd_data_f16(const float* packed, const void* imageT, int D, int W, int skip, const float* d_raw,
                                        const float* acts, int64_t M, float* dz, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    F16BwdArgs a;
    F16Layout L16;
    NF_REQUIRE(make_layout(D, W, skip, a.lay) && make_f16_layout(D, W, skip, L16), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed && imageT && d_raw && acts && dz, "NULL pointer");
    F16LayoutT T;
    make_f16_layout_T(D, a.lay.NT, T);
    a.tl = make_train_layout(D, W);
    a.packed = packed; a.imgT = reinterpret_cast<const u32x4*>(imageT); a.d_raw = d_raw; a.acts = acts; a.dz = dz; a.M = M;
    a.total_chunks = T.total_chunks;
    const long ntiles = (M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;
    const dim3 grid((unsigned)blocks), block(256);
    const size_t lds = (size_t)2 * a.lay.NT * 2 * 64 * 16;
    hipStream_t s = as_stream(stream);
    switch (W) {
        case 256: nerf_mlp_bwd_data_f16_kernel<8><<<grid, block, lds, s>>>(a); break;
        case 128: nerf_mlp_bwd_data_f16_kernel<4><<<grid, block, lds, s>>>(a); break;
        case 64: nerf_mlp_bwd_data_f16_kernel<2><<<grid, block, lds, s>>>(a); break;
        default: set_error("nerfail_mlp_bwd_data_f16: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_bwd_data_f16_kernel");
    return NERFAIL_OK;
}
