// Shared layout definitions of the fused NeRF MLP kernels (forward, backward-data, weight-gradient).
#pragma once
#include "common.h"

namespace nerfail {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPtsCh = 63, kDirCh = 27;
constexpr int kEmbQuads = 8;   // 32 k-steps: 30 sin/cos pairs + (x|y) + (z|pad)
constexpr int kDirQuads = 4;   // 16 k-steps: 12 sin/cos pairs + (x|y) + (z|pad) + 2 zero steps

// Input channel of a positional encoding (RH:47-50 order: x(3), then per band sin(3), cos(3))
// consumed by k-step s in lane half h; -1 = zero padding. `bands` = 10 (pts) or 4 (dirs).
__host__ __device__ inline int enc_channel(int s, int h, int bands) {
    if (s < 3 * bands) return 3 + 6 * (s / 3) + 3 * h + (s % 3);
    if (s == 3 * bands) return h;               // x | y
    if (s == 3 * bands + 1) return h ? -1 : 2;  // z | pad
    return -1;
}
// Channel of a 32-channel accumulator tile held in register r of lane half h (32x32 C/D layout).
__host__ __device__ inline int acc_channel(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Packed f32 image (nerfail_mlp_pack), in 1 KB "pieces" of 256 floats:
//   [0, w_total)        the WEIGHT STREAM: layer after layer in consumption order (pts_linears 0..D-1, feature_linear,
//                       views_linears), each layer [quad][out-tile][lane][4] = A fragments, one piece per (quad, tile);
//                       padded at the end to a whole number of ring groups (4*NT pieces) - the LDS-streaming forward
//                       kernel (mlp_lds.hip) copies exactly this range through its ring, piece after piece;
//   [b_off[0], total)   the CONSTANT area: one piece per layer bias ([OT][2][16], rest of the piece unused), then the
//                       alpha and rgb heads, each padded to whole pieces; copied once per workgroup into LDS.
struct MlpLayout {
    int NT, D, skip;
    unsigned w_off[NERFAIL_MAX_DEPTH + 2];   // [0..D-1] pts layers, [D] feature, [D+1] views
    unsigned b_off[NERFAIL_MAX_DEPTH + 2];
    unsigned w_count[NERFAIL_MAX_DEPTH + 2]; // floats of layer l's weights (quads * OT * 256)
    unsigned alpha_off, rgb_off, total;
    unsigned w_total;                        // floats of the weight stream incl. its end padding
    unsigned stream_pad;                     // padding pieces after the views weights
};
constexpr int kPiece = 256;                  // floats per piece (one wave-wide 16-byte access = 1 KB)

static inline bool layer_has_emb(int l, int skip) { return l == 0 || (skip >= 0 && l == skip + 1); }

static bool make_layout(int D, int W, int skip, MlpLayout& L) {
    if (!(W == 64 || W == 128 || W == 256)) return false;
    if (D < 2 || D > NERFAIL_MAX_DEPTH) return false;
    if (skip >= D - 1) skip = -1;    // `if i in skips` never fires for the last layer's successor (RH:106)
    L.NT = W / 32; L.D = D; L.skip = skip;
    const int NT = L.NT, OTV = NT / 2;
    unsigned off = 0;
    for (int l = 0; l <= D + 1; ++l) {
        const int OT = (l == D + 1) ? OTV : NT;
        int quads = 0;
        if (l <= D - 1 && layer_has_emb(l, skip)) quads += kEmbQuads;
        if (l > 0) quads += NT * 4;
        if (l == D + 1) quads += kDirQuads;
        L.w_off[l] = off; L.w_count[l] = (unsigned)quads * OT * 256; off += L.w_count[l];
    }
    const unsigned group = 4u * NT * kPiece;                  // every part but the last is a multiple of this already
    const unsigned pad = (group - off % group) % group;
    L.stream_pad = pad / kPiece; off += pad; L.w_total = off;
    for (int l = 0; l <= D + 1; ++l) { L.b_off[l] = off; off += kPiece; }
    auto up = [](unsigned n) { return (n + kPiece - 1) / kPiece * kPiece; };
    L.alpha_off = off; off += up((unsigned)NT * 32 + 4);
    L.rgb_off = off; off += up(3u * OTV * 32 + 4);
    L.total = off;
    return true;
}


// ---- training buffers: [32-sample tile][slot][32 channels][32 samples] ("channel-major tiles") -----------
// A "slot" is one 32-channel x 32-sample tile, 4 KB, element (channel c, sample j) at float offset c*32 + j.
//   * producers hold it in accumulator layout (lane = sample j + 32*half h, register r = channel acc_channel(r,h)):
//     register r of a wave is two full 128-byte rows (channels acc_channel(r,0) and acc_channel(r,1)) -> one
//     coalesced dword store / load instruction per register;
//   * the weight-gradient kernel needs "channel on the lane, sample on the MFMA k index": with the k-step mapping
//     (step st, half kh) <-> sample 16*kh + st, lane (c, kh) reads 16 CONTIGUOUS floats c*32 + 16*kh .. +15, i.e.
//     four 16-byte loads cover all 16 k-steps of the tile.
//   acts slots: E0 E1 (the 63 pts-encoding channels in RH:47-50 order, channel 63 = 0), V (27 dir-encoding
//               channels, rest 0), H_1..H_D (NT each, post-ReLU
//               outputs of pts_linears), F (NT, feature_linear output), HV (NT/2, post-ReLU views output)
//   dz   slots: Z_0..Z_{D-1} (NT each, gradient w.r.t. the pre-activation of pts_linears[i]), ZF (NT), ZV (NT/2),
//               ZR (1: d_raw, channels 0..3 = rgb, sigma)
__host__ __device__ inline int train_mask_slots(int D) { return (D + 1 + 3) / 4; }      // (D+1) entries of 1 KB
__host__ __device__ inline int train_mask_slot0(int D, int NT) { return 3 + (D + 1) * NT + NT / 2; }
__host__ __device__ inline int train_a_slots(int D, int NT) { return train_mask_slot0(D, NT) + train_mask_slots(D); }

struct TrainLayout {
    int NT, D, OTV;
    int a_slots, a_E, a_V, a_H1, a_F, a_HV, a_MASK;
    int z_slots, z_Z0, z_ZF, z_ZV, z_ZR;
};
static inline TrainLayout make_train_layout(int D, int W) {
    TrainLayout t;
    t.NT = W / 32; t.D = D; t.OTV = t.NT / 2;
    t.a_E = 0; t.a_V = 2; t.a_H1 = 3; t.a_F = 3 + D * t.NT; t.a_HV = t.a_F + t.NT; t.a_MASK = t.a_HV + t.OTV; t.a_slots = t.a_MASK + train_mask_slots(D);
    t.z_Z0 = 0; t.z_ZF = D * t.NT; t.z_ZV = t.z_ZF + t.NT; t.z_ZR = t.z_ZV + t.OTV; t.z_slots = t.z_ZR + 1;
    return t;
}

// arguments of the fused forward kernels (mlp.hip: register-streamed, also the training forward; mlp_lds.hip: LDS ring)
struct MlpArgs {
    const float* packed;
    const float* pts;        // [M,3]      (NULL when xemb is given)
    const float* viewdirs;   // [rays,3]
    const float* xemb;       // [M,90] already embedded input, or NULL
    const float* rays;       // [rays,11] packed rays (o, d, near, far, viewdir) + z [M]: the point of sample s is formed HERE,
    const float* z;          //   pts = o + d * z (RN:381 rounding: multiply, then add), and never written to HBM (pts == NULL)
    float* raw;              // [M,4]
    float* acts;             // training only: [tiles][TrainLayout::a_slots][64][16] activations for the backward
    long M;
    int spr;                 // samples per ray
    MlpLayout lay;
};
int launch_mlp_lds(const MlpArgs& a, int W, hipStream_t s);      // mlp_lds.hip; NERFAIL_EINVAL when the shape is not covered

// B operands of the two encoding parts of sample s for lane half h (k-step st <-> channel enc_channel(st, h, bands)):
// computed in registers from the raw point / view direction (one shared double-precision argument reduction per
// coordinate, SinCosBands), or gathered from an already embedded input row (NeRF.forward's contract).
__device__ __forceinline__ void encode_sample(const MlpArgs& a, long s, int h, float (&emb)[4 * kEmbQuads], float (&demb)[4 * kDirQuads]) {
    if (a.xemb == nullptr) {
        float px[3], vx[3];
        if (a.rays != nullptr) {         // north-star form: the sample's point from its ray and depth, in registers
            const float* ray = a.rays + NERFAIL_RAY_FLOATS * (s / a.spr);
            const float zz = a.z[s];
#pragma unroll
            for (int d = 0; d < 3; ++d) { px[d] = mul_add_rn(ray[3 + d], zz, ray[d]); vx[d] = ray[8 + d]; }
        } else {
            const float* vd = a.viewdirs + 3 * (s / a.spr);
#pragma unroll
            for (int d = 0; d < 3; ++d) { px[d] = a.pts[3 * s + d]; vx[d] = vd[d]; }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const SinCosBands sc(px[d]);
#pragma unroll
            for (int f = 0; f < 10; ++f) emb[3 * f + d] = sc.band_sel(f, h);
        }
        emb[30] = h ? px[1] : px[0];
        emb[31] = h ? 0.f : px[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const SinCosBands sc(vx[d]);
#pragma unroll
            for (int f = 0; f < 4; ++f) demb[3 * f + d] = sc.band_sel(f, h);
        }
        demb[12] = h ? vx[1] : vx[0];
        demb[13] = h ? 0.f : vx[2];
        demb[14] = 0.f; demb[15] = 0.f;
    } else {
        const float* x = a.xemb + (kPtsCh + kDirCh) * s;
#pragma unroll
        for (int st = 0; st < 4 * kEmbQuads; ++st) {
            const int c = enc_channel(st, h, 10);
            emb[st] = c >= 0 ? x[c] : 0.f;
        }
#pragma unroll
        for (int st = 0; st < 4 * kDirQuads; ++st) {
            const int c = enc_channel(st, h, 4);
            demb[st] = c >= 0 ? x[kPtsCh + c] : 0.f;
        }
    }
}

// transposed-weight image for the backward-data pass: per layer [quad][in-tile][lane][4] (the piece layout of the forward
// image). The layers lie in the order the backward pass CONSUMES them - views, feature, pts D-1 .. 1 - so that the image is one
// contiguous stream for the LDS weight ring (round 5, nerf_mlp_bwd_data_lds_kernel); every layer is a whole number of
// 4*NT-piece ring groups. Every user addresses a layer through w_off[l].
struct MlpLayoutT {
    unsigned w_off[NERFAIL_MAX_DEPTH + 2];   // index l = 1..D-1 pts layers, D feature, D+1 views
    unsigned total;
};
static inline void make_layout_T(int D, int NT, MlpLayoutT& L) {
    unsigned off = 0;
    L.w_off[0] = 0;
    for (int l = D + 1; l >= 1; --l) {
        L.w_off[l] = off;
        const int quads = (l == D + 1) ? (NT / 2) * 4 : NT * 4;
        off += (unsigned)quads * NT * 256;
    }
    L.total = off;
}

// arguments of the backward-data kernels (mlp_bwd.hip: register-streamed form; mlp_lds.hip: LDS weight ring)
struct BwdArgs {
    const float* packed;     // forward image (alpha / rgb head weights)
    const float* packedT;    // transposed image
    const float* packed2;    // the same two images of a SECOND network of the same architecture: tiles >= split use them
    const float* packedT2;   // (coarse + fine network of one training step in one launch; RN:394 makes them independent)
    long split;              // first 32-sample tile of the second network (= number of tiles when there is none)
    const float* d_raw;      // [M,4]
    const float* acts;       // saved activations
    float* dz;               // out: all dZ
    long M;
    int blocks0;             // LDS-ring form: workgroups [0, blocks0) serve the first network's tiles, the others the second's
    MlpLayout lay;
    MlpLayoutT layT;
    TrainLayout tl;
};
int launch_bwd_data_lds(const BwdArgs& a, int W, int cus, hipStream_t s);      // mlp_lds.hip; NERFAIL_EINVAL when the shape is not covered

// ---- device helpers shared by the forward and backward kernels -------------------------------------------
template <int OT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[OT], const float* __restrict__ b, int h) {
#pragma unroll
    for (int t = 0; t < OT; ++t) {
        const f32x4* p = reinterpret_cast<const f32x4*>(b + (t * 2 + h) * 16);
        const f32x4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        acc[t] = (f32x16){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3],
                          v2[0], v2[1], v2[2], v2[3], v3[0], v3[1], v3[2], v3[3]};
    }
}

// One part of a layer: NQ quads of 4 k-steps. Quad q's A fragments (one 16-byte load per out-tile and
// lane) are requested PFQ (default NF_MLP_PF = 1) quads AHEAD of the 4*OT MFMAs that consume them, so ~2048 MFMA
// cycles per quad of distance cover the L2 latency (one wave per SIMD: nothing else would hide it).
// bsel(q, e) yields the B operand (a register of the previous layer / of the encoding) for k-step 4q+e;
// q and e are compile-time constants after unrolling, so it is a plain register reference.
#ifndef NF_MLP_PF
#define NF_MLP_PF 1          // quads of A fragments in flight ahead of the one being multiplied
#endif
// What the exact-f32 forward kernel loses against the 154.8 TFLOP/s the bare MFMA loop sustains on this chip (clock
// 2.387 GHz inside the kernel, no throttling - tools/clockprobe, tools/fwd_clock.py), measured by ablation
// (round-1 experiment builds, see tools/README.md) at 1.57 M samples, 14.14 ms = 132 TFLOP/s:
//   weight stream served from L1 instead of L2 (re-reading quads 0/1)   13.13 ms   -> 6.5 % is L2 -> CU delivery
//   no weight stream at all (registers rotate)                          13.23 ms      (i.e. not instruction issue)
//   no positional encoding                                              13.92 ms   -> 2 %
//   no bias loads / no exposed first quad of a part                     14.08 / 14.11 ms -> 1 % each
//   rest (ReLU, accumulator shuffles at layer boundaries, heads)        ~5 %
// Tried against the 6.5 %: a ring 2 quads ahead (no change even with the encodings parked in LDS so that nothing
// spills in the loop: the limit is delivery rate, not latency) and lock-stepping the 4 waves with one s_barrier per
// quad so that one L2 fetch serves all four from L1 (17.9 ms: a barrier couples every wave to the slowest one; one
// barrier per layer: 16.2 ms - aligned waves miss L1 together, free-running ones already share it by drifting).
// Tried against the layer-boundary shuffles: a two-layer loop body in which the two register arrays swap roles
// (no copies, in-place ReLU): no change for inference (14.15 ms), 6 % slower with the training stores.
template <int OT, int NQ, int PFQ = NF_MLP_PF, typename BSel>
__device__ __forceinline__ void mfma_part(f32x16 (&acc)[OT], const float* __restrict__ w, int lane, BSel bsel) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + lane;
    constexpr int PF = PFQ < NQ ? PFQ : (NQ > 1 ? NQ - 1 : 1), RING = PF + 1;
    f32x4 ring[RING][OT];      // indices are compile-time constants after unrolling: plain registers, no copies
#pragma unroll
    for (int p = 0; p < PF && p < NQ; ++p)
#pragma unroll
        for (int t = 0; t < OT; ++t) ring[p][t] = wp[(p * OT + t) * 64];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + PF < NQ) {
#pragma unroll
            for (int t = 0; t < OT; ++t)
                ring[(q + PF) % RING][t] = wp[((q + PF) * OT + t) * 64];
            // Pin the software pipeline: nothing may be scheduled across this point, so the loads of quad q+PF stay
            // AHEAD of the 4*OT MFMAs of quad q (under register pressure the scheduler otherwise sinks them next to
            // their use and every quad waits vmcnt(0) on an exposed L2 round trip).
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < OT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[q % RING][t][e], bsel(q, e), acc[t], 0, 0, 0);
        if (q + PF < NQ) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int OT, int NQ, int PFQ = NF_MLP_PF>
__device__ __forceinline__ void mfma_scalars(f32x16 (&acc)[OT], const float* __restrict__ w, int lane,
                                             const float (&bsrc)[4 * NQ]) {
    mfma_part<OT, NQ, PFQ>(acc, w, lane, [&](int q, int e) { return bsrc[4 * q + e]; });
}

// NT*4 quads whose B operands are the previous layer's accumulator registers
template <int OT, int NT, int PFQ = NF_MLP_PF>
__device__ __forceinline__ void mfma_acts(f32x16 (&acc)[OT], const float* __restrict__ w, int lane,
                                          const f32x16 (&in)[NT]) {
    mfma_part<OT, NT * 4, PFQ>(acc, w, lane, [&](int q, int e) { return in[q >> 2][4 * (q & 3) + e]; });
}

template <int NT>
__device__ __forceinline__ void relu_to(f32x16 (&dst)[NT], const f32x16 (&src)[NT], bool relu) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[t][r] = relu ? fmaxf(src[t][r], 0.f) : src[t][r];
}

// A saved slot (1024 floats) is [2 halves of 16 samples][32 channels][16 samples]: element (channel ch, sample j) lives
// at slot_pos(j) + 16*ch. One k16-step of the weight-gradient GEMM (k = sample) is then a contiguous 2 KB half slot
// (every 128-byte line is consumed completely by the step that touches it), and an accumulator register still
// stores as four full 64-byte segments per wave instruction.
__device__ __forceinline__ int slot_pos(int j) { return (j >> 4) * 512 + (j & 15); }
constexpr int kSlotCh = 16;                // floats between consecutive channels

// store / load accumulator-layout tiles to / from channel-major slots: one dword instruction per register.
// Every address is "ONE per-lane pointer + compile-time constant":
// acc_channel(r, h) = const(r) + 4h and enc_channel(s, h) = const(s) + 3h, so the half-dependent part lives in the
// lane pointer and the rest folds into the instruction's immediate offset (computed per access, the addresses
// otherwise get hoisted, spilled, and every store ends up behind a vmcnt(0)).
__device__ __forceinline__ int acc_lane_off(int lane) { return slot_pos(lane & 31) + (lane >> 5) * 4 * kSlotCh; }
__device__ __forceinline__ constexpr int acc_reg_off(int r) { return ((r & 3) + 8 * (r >> 2)) * kSlotCh; }

#ifndef NF_NT_STORE
#define NF_NT_STORE 1
#endif
// One saved activation / gradient value: written once, read by a LATER kernel, 2 GB per pass. Stored non-temporal so
// the stream does not push the 2.4 MB weight image (re-read by every workgroup for every tile) out of L2:
// train-forward 16.1 -> 14.9 ms (f32) and 7.75 -> 7.15 ms (f16x3) at 1.57 M samples.
__device__ __forceinline__ void slot_store(float* p, float v) {
    if (NF_NT_STORE) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int NTILES>
__device__ __forceinline__ void store_tiles(float* __restrict__ base, const f32x16 (&a)[NTILES], int lane) {
    float* __restrict__ lp = base + acc_lane_off(lane);
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot_store(lp + t * 1024 + acc_reg_off(r), a[t][r]);
}
__device__ __forceinline__ f32x16 load_tile(const float* __restrict__ base, int lane) {
    const float* __restrict__ lp = base + acc_lane_off(lane);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lp[acc_reg_off(r)];
    return v;
}
// ReLU masks. The backward-data pass needs only the SIGN of every saved activation; reading the fp32 tiles for it
// doubled that kernel's HBM reads. The forward therefore also writes one bit per activation: entry e of a tile
// (e = 0..D-1 for H_{e+1}, e = D for HV; 1 KB each, in the a_MASK slots) is [64 lanes][4 dwords]; dword w of a lane
// holds its accumulator tiles 2w (bits 0-15) and 2w+1 (bits 16-31), bit r = (register r > 0).
template <int NTILES>
struct TileMask { unsigned w[(NTILES + 1) / 2]; };

// 1 where a post-ReLU value is > +0, else 0 - as integer clamp of the float's bit pattern (v_med3_i32, no trip through
// VCC; -0 and +0 give 0). A compare + select per element measured 7 % of the training forward.
__device__ __forceinline__ unsigned relu_bit(float a) {
    unsigned b;      // inline asm: hipcc rewrites min(max(x, 0), 1) into compare + select
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(a));
    return b;
}
template <int NTILES>
__device__ __forceinline__ TileMask<NTILES> mask_of(const f32x16 (&a)[NTILES]) {
    TileMask<NTILES> m;
#pragma unroll
    for (int i = 0; i < (NTILES + 1) / 2; ++i) m.w[i] = 0u;
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m.w[t >> 1] |= relu_bit(a[t][r]) << (16 * (t & 1) + r);
    return m;
}
template <int NTILES>
__device__ __forceinline__ void store_mask(float* __restrict__ mask_slots, int entry, const TileMask<NTILES>& m, int lane) {
    unsigned* __restrict__ q = reinterpret_cast<unsigned*>(mask_slots) + entry * 256 + lane * 4;
    if constexpr (NTILES == 8) {
        *reinterpret_cast<uint4*>(q) = make_uint4(m.w[0], m.w[1], m.w[2], m.w[3]);
    } else {
#pragma unroll
        for (int i = 0; i < (NTILES + 1) / 2; ++i) q[i] = m.w[i];
    }
}
template <int NTILES>
__device__ __forceinline__ TileMask<NTILES> load_mask(const float* __restrict__ mask_slots, int entry, int lane) {
    const unsigned* __restrict__ q = reinterpret_cast<const unsigned*>(mask_slots) + entry * 256 + lane * 4;
    TileMask<NTILES> m;
    if constexpr (NTILES == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(q);
        m.w[0] = v.x; m.w[1] = v.y; m.w[2] = v.z; m.w[3] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < (NTILES + 1) / 2; ++i) m.w[i] = q[i];
    }
    return m;
}
// x where the activation of (tile t, register r) was positive, else +0: one v_bfe_i32 (0 / all ones) + one v_and
template <int NTILES>
__device__ __forceinline__ float mask_apply(const TileMask<NTILES>& m, int t, int r, float x) {
    const int k = __builtin_amdgcn_sbfe((int)m.w[t >> 1], 16 * (t & 1) + r, 1);
    return __uint_as_float(__float_as_uint(x) & (unsigned)k);
}

// positional-encoding k-step values (per lane: step s -> channel enc_channel(s, h)) <-> channel-major slots.
// BANDS = 10 (pts: 63 channels in 2 slots) or 4 (dirs: 27 channels in 1 slot); padding channels are zero-filled so
// the weight-gradient kernel never reads uninitialised memory. Channel c of an encoding lives in slot c / 32.
__device__ __forceinline__ constexpr int enc_ch_off(int c) { return (c >> 5) * 1024 + (c & 31) * kSlotCh; }

template <int BANDS, int NSTEPS>
__device__ __forceinline__ void store_enc(float* __restrict__ base, const float (&e)[NSTEPS], int lane) {
    const int h = lane >> 5, j = lane & 31;
    float* __restrict__ lpz = base + slot_pos(j);
    float* __restrict__ lp3 = lpz + h * 3 * kSlotCh;   // + 3h channels
    // channel 29 (+3h) is the one pair that straddles the slot boundary (29 | 32): its own per-lane pointer
    float* __restrict__ lpx = lpz + (h ? enc_ch_off(32) : enc_ch_off(29));
#pragma unroll
    for (int s = 0; s < 3 * BANDS; ++s) {
        const int c0 = 3 + 6 * (s / 3) + (s % 3);
        if (c0 == 29) lpx[0] = e[s];
        else lp3[enc_ch_off(c0)] = e[s];
    }
    lpz[h * kSlotCh] = e[3 * BANDS];                   // x | y  (channels 0, 1)
    constexpr int nch = 3 + 6 * BANDS, cap = ((nch + 31) / 32) * 32;
    // half 0: z (channel 2) and the even padding channels; half 1: the odd padding channels
    if (h == 0) lpz[2 * kSlotCh] = e[3 * BANDS + 1];
#pragma unroll
    for (int c = nch; c < cap; ++c)
        if (((c - nch) & 1) == h) lpz[enc_ch_off(c)] = 0.f;
}
template <int BANDS, int NSTEPS>
__device__ __forceinline__ void load_enc(const float* __restrict__ base, float (&e)[NSTEPS], int lane) {
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ lpz = base + slot_pos(j);
    const float* __restrict__ lp3 = lpz + h * 3 * kSlotCh;
    const float* __restrict__ lpx = lpz + (h ? enc_ch_off(32) : enc_ch_off(29));
#pragma unroll
    for (int s = 0; s < 3 * BANDS; ++s) {
        const int c0 = 3 + 6 * (s / 3) + (s % 3);
        e[s] = (c0 == 29) ? lpx[0] : lp3[enc_ch_off(c0)];
    }
    e[3 * BANDS] = lpz[h * kSlotCh];
    const float z = lpz[2 * kSlotCh];
    e[3 * BANDS + 1] = h ? 0.f : z;
#pragma unroll
    for (int s = 3 * BANDS + 2; s < NSTEPS; ++s) e[s] = 0.f;
}

}  // namespace nerfail
