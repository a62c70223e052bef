// K4b: backward of the fused NeRF MLP (autograd of run_nerf_helpers.py:100-123 inside loss.backward(), RN:791).
//
// Two kernels, both on v_mfma_f32_32x32x2_f32, both reading the activations that nerfail_mlp_fwd_train saved as
// channel-major tiles (mlp_layout.h):
//
//  1. nerf_mlp_bwd_data_kernel: the backward-data chain dX = W^T dZ, register resident exactly like the
//     forward: dZ of a layer sits in the accumulator layout (sample on the lane, channel on the register), which
//     is the B operand of the next (earlier) layer's MFMA with A = a 32-row slab of W^T (packed once by
//     nerfail_mlp_pack_T). ReLU masks come from the saved post-ReLU activations (h > 0 <=> pre-activation > 0).
//     Every dZ is stored (fragment layout) for kernel 2. No gradient w.r.t. points/dirs is needed (RN:394 detaches
//     z_samples; rays are data), so the chain stops at layer 1.
//
//  2. nerf_mlp_bwd_weights_kernel: dW[o][i] = sum_samples dZ[o][s] X[i][s], a contraction over SAMPLES, i.e. the
//     MFMA k index is the sample. Both operands are needed as "channel on the lane, sample on k" - the transpose of
//     how the producers hold them - but the channel-major tile layout (mlp_layout.h) makes that transpose free: with
//     the k-step mapping (step st, half kh) <-> sample 16*kh + st, lane (channel c, kh) reads 16 contiguous floats,
//     four 16-byte loads for all 16 k-steps of a tile. Operands go global -> VGPR -> MFMA, no LDS, no shuffles.
//     A wave owns a 128 x 128 block of one layer's dW (4x4 accumulator tiles, 256 registers) over a chunk of
//     sample tiles and adds it to the gradient with float atomics shaped as two 128-byte runs per instruction;
//     bias gradients fall out of the A operands (row sums) for free.
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
struct BwdArgs {
    const float* packed;     // forward image (alpha / rgb head weights)
    const float* packedT;    // transposed image
    const float* d_raw;      // [M,4]
    const float* acts;       // saved activations
    float* dz;               // out: all dZ
    long M;
    MlpLayout lay;
    MlpLayoutT layT;
    TrainLayout tl;
};

template <int NT>
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_data_kernel(BwdArgs a) {
    constexpr int OTV = NT / 2;
    const int lane = threadIdx.x & 63;
    // wave id made PROVABLY wave-uniform: tile bases then live in SGPRs and every access is scalar-base + 32-bit
    // lane offset instead of a 64-bit VGPR pair per address (which spilled hundreds of registers)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const float* __restrict__ P = a.packed;
    const float* __restrict__ PT = a.packedT;
    const MlpLayout& L = a.lay;
    const TrainLayout& TL = a.tl;
    const long ntiles = (a.M + 31) / 32;
    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);

    for (long rnd = 0; rnd < nrounds; ++rnd) {
        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;
        if (tile >= ntiles) break;
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

// ------------------------------------------------------------------------------------- backward weights
struct XPart {
    int slot0, ntiles, kind;   // kind is informational (0 activations, 1 pts encoding, 2 dir encoding): all slots are channel-major
    int col0, ncols;           // destination columns [col0, col0 + ncols) of the weight gradient
};
struct LinDesc {
    int dz_slot0, dz_tiles;    // dZ slots (out tiles)
    int row0, row1;            // valid out rows (within the dZ tiles) -> gradient rows row - row0
    int in_f;                  // row stride of the gradient
    int nparts;
    XPart parts[2];
    float* gw;
    float* gb;
};
constexpr int kMaxDesc = 14, kMaxTasks = 72;
struct WTask { unsigned char desc, ob, part, ib; };
struct WArgs {
    const float* acts;
    const float* dz;
    long ntiles;               // 32-sample tiles
    int a_slots, z_slots;
    int ndesc, ntasks, ngroups;            // a group = 4 consecutive tasks = the 4 waves of a workgroup
    int bf16x3;                            // 0: exact f32 MFMA, 1: bf16 hi/lo split (dw_task_bf16)
    int group_cost[kMaxTasks / 4 + 1];     // MFMAs per k-step of the group's heaviest task
    long cum[kMaxTasks / 4 + 2];           // prefix sums of group_cost * ntiles (work units)
    LinDesc desc[kMaxDesc];
    WTask tasks[kMaxTasks];
};

// One task: an (MA x NB)-tile block of one layer's dW over one chunk of sample tiles. MA / NB are the numbers of
// VALID out / in tiles of the block (4x4 for the 256-wide layers, 4x2 for the encoding columns, 4x1 for the view
// encoding, 1x4 for the rgb / alpha heads), so no MFMA is spent on padding tiles.
template <int MA, int NB>
__device__ __forceinline__ void dw_task(const WArgs& a, const WTask tk, int lane, long t_begin, long t_end) {
    const LinDesc& d = a.desc[tk.desc];
    const XPart& xp = d.parts[tk.part];
    const int c = lane & 31, kh = lane >> 5;
    // per-lane operand offsets (floats) inside one sample tile: lane (c, kh) owns 16 contiguous floats (k-steps 0..15).
    // Loads are UNCONDITIONAL (a padding channel reads its zero-filled row; a column beyond ncols is never written
    // out): a conditional load makes hipcc branch around it and wait vmcnt(0), which kills the software pipeline.
    int offA[MA], offB[NB];
#pragma unroll
    for (int m = 0; m < MA; ++m) offA[m] = (d.dz_slot0 + 4 * tk.ob + m) * 1024 + kh * 512 + c * kSlotCh;
#pragma unroll
    for (int n = 0; n < NB; ++n) offB[n] = (xp.slot0 + 4 * tk.ib + n) * 1024 + kh * 512 + c * kSlotCh;
    f32x16 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float rowsum[MA];
#pragma unroll
    for (int m = 0; m < MA; ++m) rowsum[m] = 0.f;
    const bool do_bias = (d.gb != nullptr) && tk.part == 0 && tk.ib == 0;

    if (t_begin >= t_end) return;
    const float* __restrict__ zb = a.dz + (size_t)t_begin * a.z_slots * 1024;
    const float* __restrict__ xb = a.acts + (size_t)t_begin * a.a_slots * 1024;
    // Operand ring in units of QUADS (4 k-steps = one 16-byte load per operand and lane = 4*MA*NB MFMAs ~ 4000
    // cycles for a 4x4 block). acts / dz stream from HBM (each slot is read by one workgroup only), so the ring runs
    // PFQ = 2 quads ahead; it costs 4*(MA+NB) registers per stage.
    constexpr int PFQ = 2;
    f32x4 av[PFQ][MA], bv[PFQ][NB];
#pragma unroll
    for (int p = 0; p < PFQ; ++p) {
#pragma unroll
        for (int m = 0; m < MA; ++m) av[p][m] = *reinterpret_cast<const f32x4*>(zb + offA[m] + 4 * p);
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[p][n] = *reinterpret_cast<const f32x4*>(xb + offB[n] + 4 * p);
    }
    for (long ts = t_begin; ts < t_end; ++ts) {
        const bool last_tile = ts + 1 >= t_end;
        const float* __restrict__ zn = last_tile ? zb : zb + (size_t)a.z_slots * 1024;   // next tile (or a harmless re-read)
        const float* __restrict__ xn = last_tile ? xb : xb + (size_t)a.a_slots * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = q % PFQ;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][m][e], bv[p][n][e], acc[m][n], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int m = 0; m < MA; ++m) rowsum[m] += av[p][m][e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // refill this ring stage with quad q + PFQ (rolling into the next tile)
#pragma unroll
            for (int m = 0; m < MA; ++m)
                av[p][m] = *reinterpret_cast<const f32x4*>((q + PFQ < 4) ? zb + offA[m] + 4 * (q + PFQ) : zn + offA[m] + 4 * (q + PFQ - 4));
#pragma unroll
            for (int n = 0; n < NB; ++n)
                bv[p][n] = *reinterpret_cast<const f32x4*>((q + PFQ < 4) ? xb + offB[n] + 4 * (q + PFQ) : xn + offB[n] + 4 * (q + PFQ - 4));
            __builtin_amdgcn_sched_barrier(0);
        }
        zb = zn;
        xb = xn;
    }
    // ---- add the block into the gradient: lane = column (input channel), registers = rows (output channels)
#pragma unroll
    for (int m = 0; m < MA; ++m) {
        const int tt = 4 * tk.ob + m;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = 32 * (4 * tk.ib + n) + c;
            if (col >= xp.ncols) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * tt + acc_channel(r, kh);
                if (row >= d.row0 && row < d.row1)
                    atomicAdd(d.gw + (long)(row - d.row0) * d.in_f + xp.col0 + col, acc[m][n][r]);
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            const float sres = rowsum[m] + __shfl_xor(rowsum[m], 32, 64);
            const int row = 32 * (4 * tk.ob + m) + c;
            if (kh == 0 && row >= d.row0 && row < d.row1) atomicAdd(d.gb + (row - d.row0), sres);
        }
    }
}

// ---- split-precision form of the same task (opt-in, WArgs::bf16x3): the operands are converted IN REGISTERS to bf16
// hi / lo pairs (a = a_hi + a_lo, 16 significant bits, fp32's exponent range - so no scaling is needed for gradients of
// any magnitude) and every product block is ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16: 3 MFMAs of 32 cycles
// per 16 samples instead of 8 f32 MFMAs of 64 cycles. The dropped lo*lo term is ~2^-16 relative per product. k16-step mapping: (step ks, half kh, element j) <-> sample 16*ks + 8*kh + j,
// i.e. each lane reads 8 contiguous floats (two 16-byte loads) per operand and step.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// a = hi + lo + O(2^-18 |a|), both halves rounded to nearest (v_cvt_pk_bf16_f32, two elements per instruction): the
// residual is unbiased - a truncating split leaves every product short by the same sign, which does not average out
// over the non-negative post-ReLU activations.
__device__ __forceinline__ void split_bf16(const f32x4& v0, const f32x4& v1, u32x4b& hi, u32x4b& lo) {
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {                                         // element 2p in the low half, 2p+1 in the high half
        const f32x2 x = {v[2 * p], v[2 * p + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf2));
        const f32x2 r = {x[0] - __uint_as_float(hu << 16), x[1] - __uint_as_float(hu & 0xffff0000u)};
        hi[p] = hu;
        lo[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf2));
    }
}

template <int MA, int NB>
__device__ __forceinline__ void dw_task_bf16(const WArgs& a, const WTask tk, int lane, long t_begin, long t_end) {
    const LinDesc& d = a.desc[tk.desc];
    const XPart& xp = d.parts[tk.part];
    const int c = lane & 31, kh = lane >> 5;
    int offA[MA], offB[NB];
#pragma unroll
    for (int m = 0; m < MA; ++m) offA[m] = (d.dz_slot0 + 4 * tk.ob + m) * 1024 + c * kSlotCh + 8 * kh;
#pragma unroll
    for (int n = 0; n < NB; ++n) offB[n] = (xp.slot0 + 4 * tk.ib + n) * 1024 + c * kSlotCh + 8 * kh;
    f32x16 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float rowsum[MA];
#pragma unroll
    for (int m = 0; m < MA; ++m) rowsum[m] = 0.f;
    const bool do_bias = (d.gb != nullptr) && tk.part == 0 && tk.ib == 0;
    if (t_begin >= t_end) return;
    const float* __restrict__ zb = a.dz + (size_t)t_begin * a.z_slots * 1024;
    const float* __restrict__ xb = a.acts + (size_t)t_begin * a.a_slots * 1024;
    f32x4 ra[MA][2], rb[NB][2], na[MA][2], nb[NB][2];       // raw fp32 operands of the current / next k16-step
#pragma unroll
    for (int m = 0; m < MA; ++m) { ra[m][0] = *reinterpret_cast<const f32x4*>(zb + offA[m]); ra[m][1] = *reinterpret_cast<const f32x4*>(zb + offA[m] + 4); }
#pragma unroll
    for (int n = 0; n < NB; ++n) { rb[n][0] = *reinterpret_cast<const f32x4*>(xb + offB[n]); rb[n][1] = *reinterpret_cast<const f32x4*>(xb + offB[n] + 4); }
    for (long ts = t_begin; ts < t_end; ++ts) {
        const bool last_tile = ts + 1 >= t_end;
        const float* __restrict__ zn = last_tile ? zb : zb + (size_t)a.z_slots * 1024;
        const float* __restrict__ xn = last_tile ? xb : xb + (size_t)a.a_slots * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // prefetch the next k16-step (step 1 of this tile, or step 0 of the next tile)
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                const float* src = (ks == 0) ? zb + offA[m] + 512 : zn + offA[m];
                na[m][0] = *reinterpret_cast<const f32x4*>(src); na[m][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const float* src = (ks == 0) ? xb + offB[n] + 512 : xn + offB[n];
                nb[n][0] = *reinterpret_cast<const f32x4*>(src); nb[n][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            u32x4b ah[MA], al[MA], bh[NB], bl[NB];
#pragma unroll
            for (int m = 0; m < MA; ++m) {
                split_bf16(ra[m][0], ra[m][1], ah[m], al[m]);
                if (do_bias) rowsum[m] += (ra[m][0][0] + ra[m][0][1]) + (ra[m][0][2] + ra[m][0][3]) + (ra[m][1][0] + ra[m][1][1]) + (ra[m][1][2] + ra[m][1][3]);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) split_bf16(rb[n][0], rb[n][1], bh[n], bl[n]);
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const bf8 A_ = __builtin_bit_cast(bf8, x == 2 ? al[m] : ah[m]);
                        const bf8 B_ = __builtin_bit_cast(bf8, x == 1 ? bl[n] : bh[n]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[m][n], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MA; ++m) { ra[m][0] = na[m][0]; ra[m][1] = na[m][1]; }
#pragma unroll
            for (int n = 0; n < NB; ++n) { rb[n][0] = nb[n][0]; rb[n][1] = nb[n][1]; }
        }
        zb = zn;
        xb = xn;
    }
#pragma unroll
    for (int m = 0; m < MA; ++m) {
        const int tt = 4 * tk.ob + m;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = 32 * (4 * tk.ib + n) + c;
            if (col >= xp.ncols) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * tt + acc_channel(r, kh);
                if (row >= d.row0 && row < d.row1)
                    atomicAdd(d.gw + (long)(row - d.row0) * d.in_f + xp.col0 + col, acc[m][n][r]);
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            const float sres = rowsum[m] + __shfl_xor(rowsum[m], 32, 64);
            const int row = 32 * (4 * tk.ob + m) + c;
            if (kh == 0 && row >= d.row0 && row < d.row1) atomicAdd(d.gb + (row - d.row0), sres);
        }
    }
}

// Persistent grid (one workgroup per CU: 256 accumulator registers per lane leave room for one wave per SIMD).
// The work "group g over sample tile t" costs group_cost[g] MFMAs per k-step; the flattened (group-major) sequence of
// all such items is cut into gridDim.x equal-cost intervals, so every workgroup computes the same number of MFMAs
// and adds its accumulators to the gradient once per (group, interval) segment - at most a handful of times.
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_weights_kernel(WArgs a) {
    const int lane = threadIdx.x & 63;
    // wave id made PROVABLY wave-uniform: tile bases then live in SGPRs and every access is scalar-base + 32-bit
    // lane offset instead of a 64-bit VGPR pair per address (which spilled hundreds of registers)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long total = a.cum[a.ngroups];
    const long lo = total / gridDim.x * blockIdx.x + (total % gridDim.x) * blockIdx.x / gridDim.x;
    const long hi = total / gridDim.x * (blockIdx.x + 1) + (total % gridDim.x) * (blockIdx.x + 1) / gridDim.x;
    for (int g = 0; g < a.ngroups; ++g) {
        const long g0 = a.cum[g], g1 = a.cum[g + 1];
        if (hi <= g0 || lo >= g1) continue;
        const long c = a.group_cost[g];
        const long s_ = (lo > g0 ? lo : g0) - g0, e_ = (hi < g1 ? hi : g1) - g0;
        const long t_begin = (s_ + c - 1) / c, t_end = (e_ + c - 1) / c;     // same rounding at both ends: exact partition
        const int task_id = g * 4 + wave;
        if (task_id >= a.ntasks || t_begin >= t_end) continue;
        const WTask tk = a.tasks[task_id];
        const LinDesc& d = a.desc[tk.desc];
        int ma = d.dz_tiles - 4 * tk.ob, nb = d.parts[tk.part].ntiles - 4 * tk.ib;
        ma = ma > 4 ? 4 : ma;
        nb = nb > 4 ? 4 : nb;
        // wave-uniform dispatch on the block shape (the shapes a NeRF produces; anything else takes the padded path)
        if (a.bf16x3) {
            if (ma == 4 && nb == 4) dw_task_bf16<4, 4>(a, tk, lane, t_begin, t_end);
            else if (ma == 4 && nb == 2) dw_task_bf16<4, 2>(a, tk, lane, t_begin, t_end);
            else if (ma == 4 && nb == 1) dw_task_bf16<4, 1>(a, tk, lane, t_begin, t_end);
            else if (ma == 1 && nb == 4) dw_task_bf16<1, 4>(a, tk, lane, t_begin, t_end);
            else if (ma == 2 && nb == 2) dw_task_bf16<2, 2>(a, tk, lane, t_begin, t_end);
            else if (ma == 2 && nb == 1) dw_task_bf16<2, 1>(a, tk, lane, t_begin, t_end);
            else if (ma == 1 && nb == 2) dw_task_bf16<1, 2>(a, tk, lane, t_begin, t_end);
            else if (ma == 1 && nb == 1) dw_task_bf16<1, 1>(a, tk, lane, t_begin, t_end);
            else if (ma == 2 && nb == 4) dw_task_bf16<2, 4>(a, tk, lane, t_begin, t_end);
            else dw_task_bf16<4, 4>(a, tk, lane, t_begin, t_end);
            continue;
        }
        if (ma == 4 && nb == 4) dw_task<4, 4>(a, tk, lane, t_begin, t_end);
        else if (ma == 4 && nb == 2) dw_task<4, 2>(a, tk, lane, t_begin, t_end);
        else if (ma == 4 && nb == 1) dw_task<4, 1>(a, tk, lane, t_begin, t_end);
        else if (ma == 1 && nb == 4) dw_task<1, 4>(a, tk, lane, t_begin, t_end);
        else if (ma == 2 && nb == 2) dw_task<2, 2>(a, tk, lane, t_begin, t_end);
        else if (ma == 2 && nb == 1) dw_task<2, 1>(a, tk, lane, t_begin, t_end);
        else if (ma == 1 && nb == 2) dw_task<1, 2>(a, tk, lane, t_begin, t_end);
        else if (ma == 1 && nb == 1) dw_task<1, 1>(a, tk, lane, t_begin, t_end);
        else if (ma == 2 && nb == 4) dw_task<2, 4>(a, tk, lane, t_begin, t_end);
        else dw_task<4, 4>(a, tk, lane, t_begin, t_end);   // unreachable for W in {64,128,256}
    }
}

// ------------------------------------------------------------------------------------- LDS-staged bf16x3 weight gradients
// The register-fed kernel above leaves the bf16 MFMAs waiting: a wave can keep only one k16-step of fp32 operands in
// flight next to its 256 accumulator registers (16 KB, ~64 KB per CU - latency bound), the two waves that share an
// operand tile fetch it separately, and each instruction touches 32 half-lines. Here (W = 256 only) a workgroup owns
// one whole layer-part at a time ("group": an LA-tile run of dZ slots x an LB-tile run of activation slots, split
// evenly over its 4 waves as MA x NB blocks) and every operand tile is fetched ONCE per workgroup, straight into LDS
// by LDS-DMA (global_load_lds_dwordx4: 1 KB = 16 full 64-byte channel rows per wave instruction, no VGPRs), through a
// 4-stage ring of k16-steps: 3 stages = 96 KB per CU stay in flight behind the step being multiplied. One raw
// s_barrier per step; LDS-DMA completion is counted with s_waitcnt vmcnt(2G) so the younger stages keep flying
// across the barrier (a __syncthreads() would drain them).
//   LDS image of a slot-step (2 KB = [32 channels][16 samples] fp32): the DMA writes lane-linear (wave base + 16 B x
// lane), so the XOR swizzle that makes the readers' ds_read_b128 conflict-free is applied to the per-lane SOURCE
// address: 16-byte piece (channel c, quarter q) sits at position 4c + (q ^ ((c >> 2) & 3)). A reader lane (c, kh)
// takes quarters 2kh and 2kh+1 = samples 8kh .. 8kh+7 of the step, exactly dw_task_bf16's k mapping.
#ifndef NF_DW_AUX
#define NF_DW_AUX 0             // cache policy bits of the LDS-DMA loads (2 = nt)
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

struct LGroup { int desc, part, dz_slot0, x_slot0, shape; };
constexpr int kMaxLGroups = 16, kLdsStages = 4, kLdsStageFloats = 8192;      // 4 x 32 KB
struct LArgs {
    const float* acts;
    const float* dz;
    long ntiles;
    int a_slots, z_slots, ngroups;
    int cost[kMaxLGroups];                 // estimated cycles per k16-step (MFMA + staging), for the partition
    long cum[kMaxLGroups + 1];
    LGroup grp[kMaxLGroups];
    LinDesc desc[kMaxDesc];
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool BF16, int MA, int NB, int LA, int LB>
__device__ __forceinline__ void dw_group_lds(const LArgs& a, const LGroup& g, float* smem, int lane, int wave,
                                             long t_begin, long t_end) {
    constexpr int NPIECE = 2 * (LA + LB), G = (NPIECE + 3) / 4, NS = kLdsStages;
    constexpr int NBK = LB / NB;
    static_assert((LA / MA) * NBK == 4 && LA % MA == 0 && LB % NB == 0, "a group is split evenly over 4 waves");
    static_assert(G >= 2 && G * 1024 <= kLdsStageFloats, "stage does not fit");
    const LinDesc& d = a.desc[g.desc];
    const XPart& xp = d.parts[g.part];
    const int a0 = (wave / NBK) * MA, b0 = (wave % NBK) * NB;           // this wave's block: A tiles a0.., B tiles b0..
    const int c = lane & 31, kh = lane >> 5;
    const int p0 = 4 * c + ((2 * kh) ^ ((c >> 2) & 3));
    const int rd0 = p0 * 4, rd1 = (p0 ^ 1) * 4;                         // float offsets of the two quarters inside a slot-step
    // DMA source: this wave moves pieces wave, wave+4, ... (all of parity wave&1 = channel half of the slot-step)
    const int cs = (wave & 1) * 16 + (lane >> 2);
    const int lane_src = cs * kSlotCh + (((lane & 3) ^ ((cs >> 2) & 3)) * 4);
    f32x16 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float rowsum[MA];
#pragma unroll
    for (int m = 0; m < MA; ++m) rowsum[m] = 0.f;
    const bool do_bias = (d.gb != nullptr) && g.part == 0 && b0 == 0;
    const long S = 2 * (t_end - t_begin);                               // k16-steps of this segment
    if (S <= 0) return;                                                 // workgroup-uniform

    auto issue = [&](long s, int rs) {                                  // stage s -> ring slot rs (steps past the end re-read the last one: uniform vmcnt)
        const long sc = s < S ? s : S - 1;
        const long t = t_begin + (sc >> 1);
        const int ks = (int)(sc & 1);
        const float* __restrict__ zb = a.dz + ((size_t)t * a.z_slots + g.dz_slot0) * 1024 + ks * 512 + lane_src;
        const float* __restrict__ xb = a.acts + ((size_t)t * a.a_slots + g.x_slot0) * 1024 + ks * 512 + lane_src;
        float* dst = smem + rs * kLdsStageFloats;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int pc = wave + 4 * i;
            const int ps = pc < NPIECE ? pc : pc - 4;                   // padding piece: same parity, valid source
            const float* src = ps < 2 * LA ? zb + (ps >> 1) * 1024 : xb + ((ps - 2 * LA) >> 1) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(dst + pc * 256), 16, 0, NF_DW_AUX);
        }
    };
    // Operands of a step are read from LDS into registers one step AHEAD (raw[..]), so the ds_read latency and the
    // barrier hide behind the MFMAs of the previous step; a stage therefore has to land one step earlier.
    f32x4 rawA[MA][2], rawB[NB][2];
    auto fetch = [&](int rs) {                                          // LDS -> registers: this wave's MA + NB tiles of one stage
        const float* __restrict__ st = smem + rs * kLdsStageFloats;
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            rawA[m][0] = *reinterpret_cast<const f32x4*>(st + (a0 + m) * 512 + rd0);
            rawA[m][1] = *reinterpret_cast<const f32x4*>(st + (a0 + m) * 512 + rd1);
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            rawB[n][0] = *reinterpret_cast<const f32x4*>(st + (LA + b0 + n) * 512 + rd0);
            rawB[n][1] = *reinterpret_cast<const f32x4*>(st + (LA + b0 + n) * 512 + rd1);
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s, s);
    wait_vmcnt<(NS - 2) * G>();                                         // stage 0 (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    fetch(0);
    int rs_next = 1, rs_issue = NS - 1;
    for (long s = 0; s < S; ++s) {
        // operands of step s leave the raw registers (bf16: split into hi / lo)
        u32x4b ah[BF16 ? MA : 1], al[BF16 ? MA : 1], bh[BF16 ? NB : 1], bl[BF16 ? NB : 1];
        f32x4 av[BF16 ? 1 : MA][2], bv[BF16 ? 1 : NB][2];
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            if (do_bias) rowsum[m] += (rawA[m][0][0] + rawA[m][0][1]) + (rawA[m][0][2] + rawA[m][0][3]) +
                                      (rawA[m][1][0] + rawA[m][1][1]) + (rawA[m][1][2] + rawA[m][1][3]);
            if constexpr (BF16) {
                split_bf16(rawA[m][0], rawA[m][1], ah[m], al[m]);
            } else { av[m][0] = rawA[m][0]; av[m][1] = rawA[m][1]; }
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            if constexpr (BF16) {
                split_bf16(rawB[n][0], rawB[n][1], bh[n], bl[n]);
            } else { bv[n][0] = rawB[n][0]; bv[n][1] = rawB[n][1]; }
        }
        wait_vmcnt<(NS - 3) * G>();                                     // this wave's pieces of stage s+1 have landed
        __builtin_amdgcn_s_barrier();                                   // ... everyone's have; stage s is in everybody's registers
        issue(s + NS - 1, rs_issue);                                    // into the slot stage s-1 occupied
        fetch(rs_next);                                                 // stage s+1 (past the end: a harmless re-read)
        if constexpr (BF16) {
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const bf8 A_ = __builtin_bit_cast(bf8, x == 2 ? al[m] : ah[m]);
                        const bf8 B_ = __builtin_bit_cast(bf8, x == 1 ? bl[n] : bh[n]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[m][n], 0, 0, 0);
                    }
        } else {
            // exact-f32 form: the 8 samples a lane holds are 8 k-steps of v_mfma_f32_32x32x2_f32 (k = (step, kh) <-> sample
            // 8*kh + step inside the k16-step: any bijection works as long as A and B use the same one)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][e >> 2][e & 3], bv[n][e >> 2][e & 3], acc[m][n], 0, 0, 0);
        }
        rs_next = (rs_next + 1) & (NS - 1);
        rs_issue = (rs_issue + 1) & (NS - 1);
    }
    wait_vmcnt<0>();                                                    // drain the padding stages before the ring is reused
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < MA; ++m) {
        const int tt = a0 + m;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = 32 * (b0 + n) + c;
            if (col >= xp.ncols) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * tt + acc_channel(r, kh);
                if (row >= d.row0 && row < d.row1)
                    atomicAdd(d.gw + (long)(row - d.row0) * d.in_f + xp.col0 + col, acc[m][n][r]);
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            const float sres = rowsum[m] + __shfl_xor(rowsum[m], 32, 64);
            const int row = 32 * (a0 + m) + c;
            if (kh == 0 && row >= d.row0 && row < d.row1) atomicAdd(d.gb + (row - d.row0), sres);
        }
    }
}

// group shapes of a W = 256 network: <MA, NB, LA, LB>
//   0 full 256x256 layer-part <4,4,8,8>   1 views (128 x 256) <4,2,4,8>   2 encoding columns (256 x 63) <2,2,8,2>
//   3 view-direction columns (128 x 27) <1,1,4,1>   4 rgb head (3 x 128) <1,1,1,4>   5 alpha head (1 x 256) <1,2,1,8>
template <bool BF16>
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_weights_lds_kernel(LArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[kLdsStages * kLdsStageFloats];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long total = a.cum[a.ngroups];
    const long lo = total / gridDim.x * blockIdx.x + (total % gridDim.x) * blockIdx.x / gridDim.x;
    const long hi = total / gridDim.x * (blockIdx.x + 1) + (total % gridDim.x) * (blockIdx.x + 1) / gridDim.x;
    for (int g = 0; g < a.ngroups; ++g) {
        const long g0 = a.cum[g], g1 = a.cum[g + 1];
        if (hi <= g0 || lo >= g1) continue;
        const long cst = a.cost[g];
        const long s_ = (lo > g0 ? lo : g0) - g0, e_ = (hi < g1 ? hi : g1) - g0;
        const long t_begin = (s_ + cst - 1) / cst, t_end = (e_ + cst - 1) / cst;
        if (t_begin >= t_end) continue;
        const LGroup& grp = a.grp[g];
        switch (grp.shape) {
            case 0: dw_group_lds<BF16, 4, 4, 8, 8>(a, grp, smem, lane, wave, t_begin, t_end); break;
            case 1: dw_group_lds<BF16, 4, 2, 4, 8>(a, grp, smem, lane, wave, t_begin, t_end); break;
            case 2: dw_group_lds<BF16, 2, 2, 8, 2>(a, grp, smem, lane, wave, t_begin, t_end); break;
            case 3: dw_group_lds<BF16, 1, 1, 4, 1>(a, grp, smem, lane, wave, t_begin, t_end); break;
            case 4: dw_group_lds<BF16, 1, 1, 1, 4>(a, grp, smem, lane, wave, t_begin, t_end); break;
            default: dw_group_lds<BF16, 1, 2, 1, 8>(a, grp, smem, lane, wave, t_begin, t_end); break;
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
    NF_REQUIRE(M >= 0, "M is negative");
    BwdArgs a;
    NF_REQUIRE(make_layout(D, W, skip, a.lay), "unsupported (D, W)");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(packed && packedT && d_raw && acts && dz, "NULL pointer");
    make_layout_T(D, a.lay.NT, a.layT);
    a.tl = make_train_layout(D, W);
    a.packed = packed; a.packedT = packedT; a.d_raw = d_raw; a.acts = acts; a.dz = dz; a.M = M;
    const long ntiles = (M + 31) / 32;
    long blocks = (ntiles + 3) / 4;
    if (blocks > cu_count()) blocks = cu_count();
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = as_stream(stream);
    switch (W) {
        case 256: nerf_mlp_bwd_data_kernel<8><<<grid, block, 0, s>>>(a); break;
        case 128: nerf_mlp_bwd_data_kernel<4><<<grid, block, 0, s>>>(a); break;
        case 64: nerf_mlp_bwd_data_kernel<2><<<grid, block, 0, s>>>(a); break;
        default: set_error("nerfail_mlp_bwd_data: unsupported W"); return NERFAIL_EINVAL;
    }
    NF_LAUNCHED("nerf_mlp_bwd_data_kernel");
    return NERFAIL_OK;
}

static int bwd_weights_impl(int D, int W, int skip, const float* acts, const float* dz, int64_t M,
                            const nerfail_mlp_params* grads_host, int bf16x3, void* stream);

extern "C" int nerfail_mlp_bwd_weights(int D, int W, int skip, const float* acts, const float* dz, int64_t M,
                                       const nerfail_mlp_params* grads_host, void* stream) {
    return bwd_weights_impl(D, W, skip, acts, dz, M, grads_host, 0, stream);
}

extern "C" int nerfail_mlp_bwd_weights_bf16x3(int D, int W, int skip, const float* acts, const float* dz, int64_t M,
                                              const nerfail_mlp_params* grads_host, void* stream) {
    return bwd_weights_impl(D, W, skip, acts, dz, M, grads_host, 1, stream);
}

static int bwd_weights_impl(int D, int W, int skip, const float* acts, const float* dz, int64_t M,
                            const nerfail_mlp_params* grads_host, int bf16x3, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    MlpLayout L;
    NF_REQUIRE(make_layout(D, W, skip, L), "unsupported (D, W)");
    NF_REQUIRE(grads_host != nullptr, "grads_host is NULL");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(acts && dz, "NULL pointer");
    const nerfail_mlp_params& g = *grads_host;
    const TrainLayout TL = make_train_layout(D, W);
    const int NT = L.NT, OTV = NT / 2;
    WArgs a;
    a.acts = acts; a.dz = dz; a.ntiles = (M + 31) / 32; a.a_slots = TL.a_slots; a.z_slots = TL.z_slots; a.bf16x3 = bf16x3;
    int nd = 0;
    auto add = [&](int dz_slot0, int dz_tiles, int row0, int row1, int in_f, float* gw, float* gb) -> LinDesc& {
        LinDesc& d = a.desc[nd++];
        d.dz_slot0 = dz_slot0; d.dz_tiles = dz_tiles; d.row0 = row0; d.row1 = row1; d.in_f = in_f; d.gw = gw; d.gb = gb;
        d.nparts = 0;
        return d;
    };
    auto part = [&](LinDesc& d, int slot0, int ntiles, int kind, int col0, int ncols) {
        XPart& p = d.parts[d.nparts++];
        p.slot0 = slot0; p.ntiles = ntiles; p.kind = kind; p.col0 = col0; p.ncols = ncols;
    };
    NF_REQUIRE(D + 4 <= kMaxDesc, "network too deep for the weight-gradient descriptor table");
    for (int i = 0; i < D; ++i) {
        NF_REQUIRE(g.pts_w[i] != nullptr && g.pts_b[i] != nullptr, "pts_linears gradient pointer is NULL");
        const bool emb = layer_has_emb(i, L.skip);
        const int in_f = (i == 0) ? kPtsCh : (emb ? W + kPtsCh : W);
        LinDesc& d = add(TL.z_Z0 + i * NT, NT, 0, W, in_f, (float*)g.pts_w[i], (float*)g.pts_b[i]);
        if (emb) part(d, TL.a_E, 2, 1, 0, kPtsCh);          // 63 encoding channels live in slots E0 E1 (channel 63 = 0)
        if (i > 0) part(d, TL.a_H1 + (i - 1) * NT, NT, 0, emb ? kPtsCh : 0, W);
    }
    NF_REQUIRE(g.feature_w && g.feature_b && g.views_w && g.views_b && g.alpha_w && g.alpha_b && g.rgb_w && g.rgb_b,
               "head gradient pointer is NULL");
    {
        LinDesc& d = add(TL.z_ZF, NT, 0, W, W, (float*)g.feature_w, (float*)g.feature_b);
        part(d, TL.a_H1 + (D - 1) * NT, NT, 0, 0, W);
    }
    {
        LinDesc& d = add(TL.z_ZV, OTV, 0, W / 2, W + kDirCh, (float*)g.views_w, (float*)g.views_b);
        part(d, TL.a_F, NT, 0, 0, W);
        part(d, TL.a_V, 1, 2, W, kDirCh);
    }
    {   // rgb_linear: rows 0..2 of d_raw against hv
        LinDesc& d = add(TL.z_ZR, 1, 0, 3, W / 2, (float*)g.rgb_w, (float*)g.rgb_b);
        part(d, TL.a_HV, OTV, 0, 0, W / 2);
    }
    {   // alpha_linear: row 3 of d_raw against the last pts activation
        LinDesc& d = add(TL.z_ZR, 1, 3, 4, W, (float*)g.alpha_w, (float*)g.alpha_b);
        part(d, TL.a_H1 + (D - 1) * NT, NT, 0, 0, W);
    }
    a.ndesc = nd;
    // Kernel choice: W = 256 runs on the LDS-staged kernel in both precisions (exact f32 2.41 -> 2.01 ms, bf16x3
    // 1.12 -> 1.05 ms at 196 608 samples once its partition used fitted step costs); other widths and
    // NERFAIL_DW_KERNEL=reg take the register-fed kernel (kept for A/B timing and parity-tested alongside).
    const char* dwk = getenv("NERFAIL_DW_KERNEL");
    const bool use_lds = dwk != nullptr ? dwk[0] == 'l' : true;
    if (NT == 8 && use_lds) {   // LDS-staged kernel: one group per layer-part, in descriptor order
        LArgs la;
        la.acts = acts; la.dz = dz; la.ntiles = a.ntiles; la.a_slots = a.a_slots; la.z_slots = a.z_slots;
        for (int di = 0; di < nd; ++di) la.desc[di] = a.desc[di];
        int ng = 0;
        la.cum[0] = 0;
        for (int di = 0; di < nd; ++di)
            for (int p = 0; p < a.desc[di].nparts; ++p) {
                const LinDesc& d = a.desc[di];
                const int LA = d.dz_tiles, LB = d.parts[p].ntiles;
                int shape, ma, nb;
                if (LA == 8 && LB == 8) { shape = 0; ma = 4; nb = 4; }
                else if (LA == 4 && LB == 8) { shape = 1; ma = 4; nb = 2; }
                else if (LA == 8 && LB == 2) { shape = 2; ma = 2; nb = 2; }
                else if (LA == 4 && LB == 1) { shape = 3; ma = 1; nb = 1; }
                else if (LA == 1 && LB == 4) { shape = 4; ma = 1; nb = 1; }
                else if (LA == 1 && LB == 8) { shape = 5; ma = 1; nb = 2; }
                else { set_error("nerfail_mlp_bwd_weights: unexpected layer shape for the LDS-staged kernel"); return NERFAIL_EINVAL; }
                NF_REQUIRE(ng < kMaxLGroups, "too many weight-gradient groups");
                LGroup& g = la.grp[ng];
                g.desc = di; g.part = p; g.dz_slot0 = d.dz_slot0; g.x_slot0 = d.parts[p].slot0; g.shape = shape;
                const int G = (2 * (LA + LB) + 3) / 4;
                // ns per k16-step of each group shape, FITTED to per-workgroup busy times (tools/dw_balance.py): the
                // obvious model (MFMA cycles + a staging constant) under-estimated the small shapes by 15-35 % and left
                // the workgroups that own them 25-36 % over the mean
                static const int kStepNs[2][6] = {{4460, 2450, 1486, 496, 483, 732}, {2030, 1427, 987, 551, 539, 648}};
                (void)ma; (void)nb; (void)G;
                la.cost[ng] = kStepNs[bf16x3 ? 1 : 0][shape];
                la.cum[ng + 1] = la.cum[ng] + (long)la.cost[ng] * la.ntiles;
                ++ng;
            }
        la.ngroups = ng;
        long wgs = cu_count();
        const long min_units = (bf16x3 ? 2023L : 4444L) * 4;       // at least ~4 full-layer tiles per workgroup
        if (wgs > la.cum[ng] / min_units) wgs = la.cum[ng] / min_units > 0 ? la.cum[ng] / min_units : 1;
        if (bf16x3) nerf_mlp_bwd_weights_lds_kernel<true><<<dim3((unsigned)wgs), dim3(256), 0, as_stream(stream)>>>(la);
        else nerf_mlp_bwd_weights_lds_kernel<false><<<dim3((unsigned)wgs), dim3(256), 0, as_stream(stream)>>>(la);
        NF_LAUNCHED("nerf_mlp_bwd_weights_lds_kernel");
        return NERFAIL_OK;
    }
    int nt = 0;
    for (int di = 0; di < nd; ++di) {
        const LinDesc& d = a.desc[di];
        for (int ob = 0; ob < (d.dz_tiles + 3) / 4; ++ob)
            for (int p = 0; p < d.nparts; ++p)
                for (int ib = 0; ib < (d.parts[p].ntiles + 3) / 4; ++ib) {
                    NF_REQUIRE(nt < kMaxTasks, "too many weight-gradient tasks");
                    a.tasks[nt].desc = (unsigned char)di; a.tasks[nt].ob = (unsigned char)ob;
                    a.tasks[nt].part = (unsigned char)p; a.tasks[nt].ib = (unsigned char)ib;
                    ++nt;
                }
    }
    a.ntasks = nt;
    // cost of a task = MFMAs per k-step = valid out tiles x valid in tiles; heaviest first, so the 4 waves of a
    // workgroup carry equal work and the light tasks (encoding columns, heads) fill the tail
    auto cost = [&](const WTask& t) {
        const LinDesc& d = a.desc[t.desc];
        int ma = d.dz_tiles - 4 * t.ob, nb = d.parts[t.part].ntiles - 4 * t.ib;
        return (ma > 4 ? 4 : ma) * (nb > 4 ? 4 : nb);
    };
    for (int i = 1; i < nt; ++i) {      // insertion sort, stable
        const WTask t = a.tasks[i];
        int j = i - 1;
        while (j >= 0 && cost(a.tasks[j]) < cost(t)) { a.tasks[j + 1] = a.tasks[j]; --j; }
        a.tasks[j + 1] = t;
    }
    a.ngroups = (nt + 3) / 4;
    a.cum[0] = 0;
    for (int g = 0; g < a.ngroups; ++g) {
        int c = 1;
        for (int w = 0; w < 4 && g * 4 + w < nt; ++w) { const int cw = cost(a.tasks[g * 4 + w]); c = cw > c ? cw : c; }
        a.group_cost[g] = c;
        a.cum[g + 1] = a.cum[g] + (long)c * a.ntiles;
    }
    long wgs = cu_count();
    if (wgs > a.cum[a.ngroups] / 16) wgs = a.cum[a.ngroups] / 16 > 0 ? a.cum[a.ngroups] / 16 : 1;   // tiny problems
    const dim3 grid((unsigned)wgs), block(256);
    nerf_mlp_bwd_weights_kernel<<<grid, block, 0, as_stream(stream)>>>(a);
    NF_LAUNCHED("nerf_mlp_bwd_weights_kernel");
    return NERFAIL_OK;
}
