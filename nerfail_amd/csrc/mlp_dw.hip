// K4b, weight gradients: dW[o][i] = sum_samples dZ[o][s] X[i][s] for every layer of the fused NeRF MLP (autograd of
// run_nerf_helpers.py:100-123 inside loss.backward(), RN:791). The contraction runs over SAMPLES, i.e. the MFMA k index
// is the sample; both operands are read from the channel-major tiles the forward / backward-data kernels saved
// (mlp_layout.h), where lane (channel c, half kh) finds its samples as contiguous floats.
//
//  * nerf_mlp_dw_lds_kernel (W = 256, the training configuration): LDS-staged, software-pipelined, DETERMINISTIC.
//    A workgroup owns one layer-part ("group": LA dZ tiles x LB activation tiles, split over its 4 waves as MA x NB
//    blocks, 256 accumulator registers per lane) over a run of sample tiles; every operand tile is fetched ONCE per
//    workgroup by LDS-DMA through a 4-stage ring of k16-steps. The groups of BOTH networks of a training step (coarse
//    and fine: independent, RN:394 detaches z_samples) are served by ONE launch; the flattened (group x tile) work is
//    cut into equal-cost intervals, one per workgroup. A workgroup's partial block goes to its own slab in a caller-
//    provided scratch (plain full-line stores), and nerf_mlp_dw_combine_kernel adds the slabs of a group in workgroup
//    order: no float atomics, every order fixed -> the gradient is bitwise reproducible.
//  * nerf_mlp_bwd_weights_kernel (other widths, NERFAIL_DW_KERNEL=reg): register-fed, float atomics (kept for the
//    D=4 W=64 plumbing configuration and as an A/B partner in the parity tests).
#include <cstdlib>
#include <type_traits>
#include "mlp_layout.h"

namespace nerfail {

// ------------------------------------------------------------------------------------- descriptors shared by both kernels
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
// ------------------------------------------------------------------------------------- LDS-staged, deterministic kernel
// LDS image of a slot-step (2 KB = [32 channels][16 samples] fp32): the DMA writes lane-linear (wave base + 16 B x lane),
// so the XOR swizzle that makes the readers' ds_read_b128 conflict-free is applied to the per-lane SOURCE address:
// 16-byte piece (channel c, quarter q) sits at position 4c + (q ^ ((c >> 2) & 3)). A reader lane (c, kh) takes quarters
// 2kh and 2kh+1 = samples 8kh .. 8kh+7 of the step; for the exact-f32 MFMA these are 8 k-steps (k = (step, kh) <-> sample
// 8*kh + step: any bijection works as long as A and B use the same one), for the split form one k16-step.
//
// The step loop is SOFTWARE-PIPELINED BY HAND and pinned with sched_barrier(0): one piece of side work (the stage's
// barrier, one LDS-DMA of the stage three steps ahead, one ds_read_b128 of the NEXT step's operands, a bias row sum)
// stands behind each MFMA, so the matrix pipe never waits for it. The round-2 form of this loop left that order to
// hipcc, which put the LDS reads, their lgkmcnt(0), the barrier, ~100 scalar address instructions (with a kernarg
// s_load + wait) and the eight DMAs in FRONT of the step's 128 MFMAs: 4.46 us per full-layer step against 3.43 us of
// MFMA time (tools/dw_balance.py fit).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kDwMaxGroups = 32, kDwStages = 4, kDwStageFloats = 8192;      // ring: 4 x 32 KB

// group shapes of a W = 256 network: <MA, NB, LA, LB>
//   0 full 256x256 layer-part <4,4,8,8>   1 views (128 x 256) <4,2,4,8>   2 encoding columns (256 x 63) <2,2,8,2>
//   3 view-direction columns (128 x 27) <1,1,4,1>   4 rgb head (3 x 128) <1,1,1,4>   5 alpha head (1 x 256) <1,2,1,8>
__host__ __device__ inline void dw_shape(int sh, int& MA, int& NB, int& LA, int& LB) {
    switch (sh) {
        case 0: MA = 4; NB = 4; LA = 8; LB = 8; break;
        case 1: MA = 4; NB = 2; LA = 4; LB = 8; break;
        case 2: MA = 2; NB = 2; LA = 8; LB = 2; break;
        case 3: MA = 1; NB = 1; LA = 4; LB = 1; break;
        case 4: MA = 1; NB = 1; LA = 1; LB = 4; break;
        default: MA = 1; NB = 2; LA = 1; LB = 8; break;
    }
}
// floats of one workgroup's partial of a group: the LA x LB tile block in accumulator order
// [wave][m][n][register][lane], then the bias row sums [wave][m][lane]
__host__ __device__ inline int dw_seg_floats(int sh) {
    int MA, NB, LA, LB;
    dw_shape(sh, MA, NB, LA, LB);
    return LA * LB * 1024 + 4 * MA * 64;
}
// The interval [lo, hi) of workgroup `wg` in the flattened cost sequence, and its tile range inside the group that
// spans [g0, g1) at `cost` units per tile. Used by the kernel AND by the host's slab planning: identical arithmetic.
__host__ __device__ inline void dw_interval(long total, int wgs, int wg, long& lo, long& hi) {
    lo = total / wgs * wg + (total % wgs) * wg / wgs;
    hi = total / wgs * (wg + 1) + (total % wgs) * (wg + 1) / wgs;
}
__host__ __device__ inline bool dw_tiles(long lo, long hi, long g0, long g1, long cost, int& t_begin, int& t_end) {
    if (hi <= g0 || lo >= g1) return false;
    const long s_ = (lo > g0 ? lo : g0) - g0, e_ = (hi < g1 ? hi : g1) - g0;
    t_begin = (int)((s_ + cost - 1) / cost);                    // same rounding at both ends: exact partition
    t_end = (int)((e_ + cost - 1) / cost);
    return t_begin < t_end;
}

struct DwGroup {
    long slab;                 // float offset of the group's first segment in the scratch
    int dz_slot0, x_slot0;     // first dZ / activation slot inside a tile
    int shape, bias;           // bias: the waves with b0 == 0 also deliver the bias gradient's row sums
    int t0, t1;                // tile range of the group's network inside acts / dz
    int first_wg, seg_floats;  // first workgroup that holds a segment of this group; floats per segment
};
struct DwArgs {
    const float* acts;
    const float* dz;
    float* slab;
    int a_slots, z_slots, ngroups, pad;
    long cum[kDwMaxGroups + 1];            // prefix sums of cost * tiles
    int cost[kDwMaxGroups];                // ns per tile (two k16-steps) of the group's shape: the partition's unit
    DwGroup grp[kDwMaxGroups];
};

#ifndef NF_DW_DMA_BUF
#define NF_DW_DMA_BUF 0    // measured round 5: the buffer form removes 155 SGPR-spill lane moves and 150 64-bit address adds from the code and is SLOWER here (2.75 vs 2.55 ms; the forward kernel gains from it) - left off
#endif
template <int N> __device__ __forceinline__ void dw_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool BF16, int MA, int NB, int LA, int LB>
__device__ __forceinline__ void dw_group_run(const DwArgs& a, const DwGroup& g, float* smem, int lane, int wave,
                                             int t_begin, int t_end, float* __restrict__ out) {
    constexpr int NPIECE = 2 * (LA + LB), G = (NPIECE + 3) / 4, NS = kDwStages, NBK = LB / NB, NOP = MA + NB;
    static_assert((LA / MA) * NBK == 4 && LA % MA == 0 && LB % NB == 0, "a group is split evenly over 4 waves");
    static_assert(G >= 2 && G * 1024 <= kDwStageFloats, "stage does not fit");
    const int a0 = (wave / NBK) * MA, b0 = (wave % NBK) * NB;           // this wave's block: A tiles a0.., B tiles b0..
    const int c = lane & 31, kh = lane >> 5;
    const int p0 = 4 * c + ((2 * kh) ^ ((c >> 2) & 3));
    // byte offsets of this lane's two quarters inside a slot-step, plus the wave's first A tile (LDS reads are then
    // "one VGPR + 16-bit immediate": tile index and B-block base fold into the immediate / one more VGPR)
    const int rdA0 = (p0 * 4 + a0 * 512) * 4, rdA1 = ((p0 ^ 1) * 4 + a0 * 512) * 4;
    const int rdB0 = (p0 * 4 + (LA + b0) * 512) * 4, rdB1 = ((p0 ^ 1) * 4 + (LA + b0) * 512) * 4;
    // DMA source: this wave moves pieces wave, wave+4, ... (all of parity wave&1 = channel half of the slot-step)
    const int cs = (wave & 1) * 16 + (lane >> 2);
    const unsigned lane_src = (unsigned)(cs * kSlotCh + (((lane & 3) ^ ((cs >> 2) & 3)) * 4)) * 4u;     // bytes
    unsigned voff[G];
    bool isz[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int pc = wave + 4 * i;
        const int ps = pc < NPIECE ? pc : pc - 4;                       // padding piece: same parity, valid source
        isz[i] = ps < 2 * LA;
        voff[i] = lane_src + (unsigned)(isz[i] ? (ps >> 1) : ((ps - 2 * LA) >> 1)) * 4096u;
    }
    f32x16 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    f32x4 rowsum[MA];                                                   // per-lane partial row sums of the A operands (bias gradient)
#pragma unroll
    for (int m = 0; m < MA; ++m) rowsum[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int S = 2 * (t_end - t_begin);                                // k16-steps of this segment (even, >= 2)

    // stage cursor: base addresses of the NEXT stage to issue (wave-uniform), advanced step by step; past the last
    // stage it stays there (the padding issues re-read the last stage: uniform vmcnt counts, harmless)
    const long ztile = (long)a.z_slots * 4096, xtile = (long)a.a_slots * 4096;
    const char* zc = reinterpret_cast<const char*>(a.dz) + ((long)(g.t0 + t_begin) * a.z_slots + g.dz_slot0) * 4096;
    const char* xc = reinterpret_cast<const char*>(a.acts) + ((long)(g.t0 + t_begin) * a.a_slots + g.x_slot0) * 4096;
    int si = 0;
#if NF_DW_DMA_BUF
    // Round 5: the LDS-DMA in the buffer form (SGPR resource + SGPR stage offset + the lane's 32-bit piece offset) instead of
    // global_load_lds with a 64-bit address pair per lane: no 64-bit vector add per piece and a cheaper issue (the forward
    // kernel's measurement: a DMA of the global form outlasts the 64 cycles of the MFMA it stands behind). The resources start at
    // this segment's first stage, so the stage offsets stay far below 4 GB whatever the size of the buffers.
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)zc, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xc, 0, 0x7fffffff, 0x00020000);
    unsigned zo = 0u, xo = 0u;
    auto dma = [&](int i, int rs) {                                     // piece i of the stage at the cursor -> ring slot rs
        lds_void_t* dst = (lds_void_t*)(smem + rs * kDwStageFloats + (wave + 4 * i) * 256);
        // (selected, not branched: isz is wave-uniform - five s_cselect; a branch would cut the pinned schedule)
        // (pieces wave, wave + 4, ...: with 2 LA a multiple of 4 "piece i is a dZ piece" does not depend on the wave - a
        // compile-time choice between the two resources; left to a run-time select hipcc kept one descriptor per piece in
        // SGPRs and spilled them to VGPR lanes)
        const bool zi = ((2 * LA) % 4 == 0) ? (i < LA / 2) : isz[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(zi ? rz : rx, dst, 16, (int)voff[i], (int)(zi ? zo : xo), 0, 0);
    };
    auto advance = [&]() {                                              // branch-free: a branch here would cut the pinned schedule
        const unsigned go = 0u - (unsigned)(si < S - 1);                // all ones while there is a next stage
        const bool odd = si & 1;
        zo += (odd ? (unsigned)ztile - 2048u : 2048u) & go;
        xo += (odd ? (unsigned)xtile - 2048u : 2048u) & go;
        si += (int)(go & 1u);
    };
#else
    auto dma = [&](int i, int rs) {                                     // piece i of the stage at the cursor -> ring slot rs
        const char* base = isz[i] ? zc : xc;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + voff[i]),
                                         (lds_void_t*)(smem + rs * kDwStageFloats + (wave + 4 * i) * 256), 16, 0, 0);
    };
    auto advance = [&]() {                                              // branch-free: a branch here would cut the pinned schedule
        const long go = -(long)(si < S - 1);                            // all ones while there is a next stage
        const bool odd = si & 1;
        zc += (odd ? ztile - 2048 : 2048L) & go;
        xc += (odd ? xtile - 2048 : 2048L) & go;
        si -= (int)go;
    };
#endif
    // operands of two consecutive steps (ping-pong): R[p][t][hf], t < MA: A tiles, then NB B tiles
    f32x4 R[2][NOP][2];
    auto fetch_one = [&](int p, int idx, const char* sbase) {           // LDS -> registers: one ds_read_b128
        const int t = idx >> 1, hf = idx & 1;
        const int off = t < MA ? (hf ? rdA1 : rdA0) + t * 2048 : (hf ? rdB1 : rdB0) + (t - MA) * 2048;
        R[p][t][hf] = *reinterpret_cast<const f32x4*>(sbase + off);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int i = 0; i < G; ++i) dma(i, s);
        advance();
    }
    dw_wait_vmcnt<(NS - 2) * G>();                                      // stage 0 (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int idx = 0; idx < 2 * NOP; ++idx) fetch_one(0, idx, reinterpret_cast<const char*>(smem));

    constexpr int NMF = BF16 ? 3 * MA * NB : 8 * MA * NB;               // MFMAs per step
    constexpr int NSIDE = 1 + G + 2 * NOP + MA;                         // side-work items per step
    int rs_issue = NS - 1, rs_next = 1;                                 // ring slots: stage s+3 goes in, stage s+1 comes out
    auto step = [&](auto PP) {
        constexpr int P = decltype(PP)::value;
        const char* const sbase = reinterpret_cast<const char*>(smem + rs_next * kDwStageFloats);
        auto side = [&](int k) {
            if (k == 0) {
                dw_wait_vmcnt<(NS - 3) * G>();                          // this wave's pieces of stage s+1 have landed
                __builtin_amdgcn_s_barrier();                           // ... everyone's have; stage s-1's slot is free
            } else if (k <= G) {
                dma(k - 1, rs_issue);
                if (k == G) advance();
            } else if (k <= G + 2 * NOP) {
                fetch_one(P ^ 1, k - G - 1, sbase);
            } else if (k < NSIDE) {
                const int m = k - G - 2 * NOP - 1;
                rowsum[m] = rowsum[m] + (R[P][m][0] + R[P][m][1]);
            }
        };
        int k = 0;
        if constexpr (BF16) {
            u32x4b ah[MA], al[MA], bh[NB], bl[NB];
#pragma unroll
            for (int m = 0; m < MA; ++m) split_bf16(R[P][m][0], R[P][m][1], ah[m], al[m]);
#pragma unroll
            for (int n = 0; n < NB; ++n) split_bf16(R[P][MA + n][0], R[P][MA + n][1], bh[n], bl[n]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const bf8 A_ = __builtin_bit_cast(bf8, x == 2 ? al[m] : ah[m]);
                        const bf8 B_ = __builtin_bit_cast(bf8, x == 1 ? bl[n] : bh[n]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, acc[m][n], 0, 0, 0);
                        side(k);
                        ++k;
                        __builtin_amdgcn_sched_barrier(0);
                    }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int m = 0; m < MA; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(R[P][m][e >> 2][e & 3], R[P][MA + n][e >> 2][e & 3],
                                                                         acc[m][n], 0, 0, 0);
                        side(k);
                        ++k;
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
#pragma unroll
        for (int k2 = NMF; k2 < NSIDE; ++k2) side(k2);                  // thin shapes: fewer MFMAs than side items
        __builtin_amdgcn_sched_barrier(0);
        rs_next = (rs_next + 1) & (NS - 1);
        rs_issue = (rs_issue + 1) & (NS - 1);
    };
    for (int s = 0; s < S; s += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    dw_wait_vmcnt<0>();                                                 // drain the padding stages before the ring is reused
    __builtin_amdgcn_s_barrier();
    // ---- this workgroup's partial of the group -> its slab (one 256-byte run per register; summed by the combine kernel)
    float* __restrict__ o = out + wave * (MA * NB * 1024) + lane;
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((m * NB + n) * 16 + r) * 64] = acc[m][n][r];
    float* __restrict__ ob = out + LA * LB * 1024 + wave * (MA * 64) + lane;
#pragma unroll
    for (int m = 0; m < MA; ++m) ob[m * 64] = (rowsum[m][0] + rowsum[m][1]) + (rowsum[m][2] + rowsum[m][3]);
}

template <bool BF16>
__global__ __launch_bounds__(256, 1) void nerf_mlp_dw_lds_kernel(DwArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[kDwStages * kDwStageFloats];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long lo, hi;
    dw_interval(a.cum[a.ngroups], (int)gridDim.x, (int)blockIdx.x, lo, hi);
    for (int g = 0; g < a.ngroups; ++g) {
        int t_begin, t_end;
        if (!dw_tiles(lo, hi, a.cum[g], a.cum[g + 1], a.cost[g], t_begin, t_end)) continue;
        const DwGroup& grp = a.grp[g];
        float* out = a.slab + grp.slab + (long)((int)blockIdx.x - grp.first_wg) * grp.seg_floats;
        switch (grp.shape) {
            case 0: dw_group_run<BF16, 4, 4, 8, 8>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
            case 1: dw_group_run<BF16, 4, 2, 4, 8>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
            case 2: dw_group_run<BF16, 2, 2, 8, 2>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
            case 3: dw_group_run<BF16, 1, 1, 4, 1>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
            case 4: dw_group_run<BF16, 1, 1, 1, 4>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
            default: dw_group_run<BF16, 1, 2, 1, 8>(a, grp, smem, lane, wave, t_begin, t_end, out); break;
        }
    }
}

// ---- fixed-order sum of the workgroup partials of every group -> the gradient tensors (no atomics)
struct DwCombineGroup {
    long slab;
    float* gw;
    float* gb;                 // NULL: this group delivers no bias gradient
    int nseg, seg_floats, shape;
    int row0, row1, in_f, col0, ncols;
};
struct DwCombineArgs {
    const float* slab;
    int ngroups, accumulate;
    DwCombineGroup grp[kDwMaxGroups];
};

__global__ __launch_bounds__(256) void nerf_mlp_dw_combine_kernel(DwCombineArgs a) {
    const DwCombineGroup& g = a.grp[blockIdx.y];
    int MA, NB, LA, LB;
    dw_shape(g.shape, MA, NB, LA, LB);
    const int NBK = LB / NB, nblock = LA * LB * 1024, nall = nblock + 4 * MA * 64;
    const float* __restrict__ base = a.slab + g.slab;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nall; e += gridDim.x * 256) {
        if (e < nblock) {
            const int lane = e & 63, r = (e >> 6) & 15, mn = (e >> 10) % (MA * NB), wave = e / (MA * NB * 1024);
            const int m = mn / NB, n = mn % NB;
            const int a0 = (wave / NBK) * MA, b0 = (wave % NBK) * NB;
            const int row = 32 * (a0 + m) + acc_channel(r, lane >> 5), col = 32 * (b0 + n) + (lane & 31);
            if (col >= g.ncols || row < g.row0 || row >= g.row1) continue;
            // workgroup order: fixed. Round 6: 8 loads in flight per thread (one dependent load after the other made this kernel
            // latency bound: 37 us for 65 MB); missing slabs add +0, which leaves s (never -0: it starts at +0) unchanged.
            float s = 0.f;
            for (int k0 = 0; k0 < g.nseg; k0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = k0 + u < g.nseg ? __builtin_nontemporal_load(base + (long)(k0 + u) * g.seg_floats + e) : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            float* dst = g.gw + (long)(row - g.row0) * g.in_f + g.col0 + col;
            *dst = a.accumulate ? *dst + s : s;
        } else if (g.gb != nullptr) {
            const int i = e - nblock, lane = i & 63, m = (i >> 6) % MA, wave = i / (MA * 64);
            if (lane >= 32 || (wave % NBK) != 0) continue;
            const int row = 32 * ((wave / NBK) * MA + m) + lane;
            if (row < g.row0 || row >= g.row1) continue;
            float s = 0.f;
            for (int k = 0; k < g.nseg; ++k) {
                const float* p = base + (long)k * g.seg_floats + e;
                s += p[0] + p[32];                                                       // the two lane halves of the row
            }
            float* dst = g.gb + (row - g.row0);
            *dst = a.accumulate ? *dst + s : s;
        }
    }
}

struct ZeroTable { int n; float* p[2 * (NERFAIL_MAX_DEPTH + 4) * 2]; int len[2 * (NERFAIL_MAX_DEPTH + 4) * 2]; };
__global__ void dw_zero_kernel(ZeroTable t) {
    float* __restrict__ p = t.p[blockIdx.y];
    const int n = t.len[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
}

static int cu_count_dw() {
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

namespace {

// The linear layers of one network as (dZ slots, activation parts, gradient tensors) - in the order the kernels walk them.
int build_descs(int D, int W, const MlpLayout& L, const TrainLayout& TL, const nerfail_mlp_params* g, LinDesc* desc, int& nd) {
    const int NT = L.NT, OTV = NT / 2;
    nd = 0;
    auto add = [&](int dz_slot0, int dz_tiles, int row0, int row1, int in_f, const float* gw, const float* gb) -> LinDesc& {
        LinDesc& d = desc[nd++];
        d.dz_slot0 = dz_slot0; d.dz_tiles = dz_tiles; d.row0 = row0; d.row1 = row1; d.in_f = in_f;
        d.gw = const_cast<float*>(gw); d.gb = const_cast<float*>(gb);
        d.nparts = 0;
        return d;
    };
    auto part = [&](LinDesc& d, int slot0, int ntiles, int kind, int col0, int ncols) {
        XPart& p = d.parts[d.nparts++];
        p.slot0 = slot0; p.ntiles = ntiles; p.kind = kind; p.col0 = col0; p.ncols = ncols;
    };
    NF_REQUIRE(D + 4 <= kMaxDesc, "network too deep for the weight-gradient descriptor table");
    for (int i = 0; i < D; ++i) {
        NF_REQUIRE(g == nullptr || (g->pts_w[i] != nullptr && g->pts_b[i] != nullptr), "pts_linears gradient pointer is NULL");
        const bool emb = layer_has_emb(i, L.skip);
        const int in_f = (i == 0) ? kPtsCh : (emb ? W + kPtsCh : W);
        LinDesc& d = add(TL.z_Z0 + i * NT, NT, 0, W, in_f, g ? g->pts_w[i] : nullptr, g ? g->pts_b[i] : nullptr);
        if (emb) part(d, TL.a_E, 2, 1, 0, kPtsCh);          // 63 encoding channels live in slots E0 E1 (channel 63 = 0)
        if (i > 0) part(d, TL.a_H1 + (i - 1) * NT, NT, 0, emb ? kPtsCh : 0, W);
    }
    NF_REQUIRE(g == nullptr || (g->feature_w && g->feature_b && g->views_w && g->views_b && g->alpha_w && g->alpha_b && g->rgb_w && g->rgb_b),
               "head gradient pointer is NULL");
    {
        LinDesc& d = add(TL.z_ZF, NT, 0, W, W, g ? g->feature_w : nullptr, g ? g->feature_b : nullptr);
        part(d, TL.a_H1 + (D - 1) * NT, NT, 0, 0, W);
    }
    {
        LinDesc& d = add(TL.z_ZV, OTV, 0, W / 2, W + kDirCh, g ? g->views_w : nullptr, g ? g->views_b : nullptr);
        part(d, TL.a_F, NT, 0, 0, W);
        part(d, TL.a_V, 1, 2, W, kDirCh);
    }
    {   // rgb_linear: rows 0..2 of d_raw against hv
        LinDesc& d = add(TL.z_ZR, 1, 0, 3, W / 2, g ? g->rgb_w : nullptr, g ? g->rgb_b : nullptr);
        part(d, TL.a_HV, OTV, 0, 0, W / 2);
    }
    {   // alpha_linear: row 3 of d_raw against the last pts activation
        LinDesc& d = add(TL.z_ZR, 1, 3, 4, W, g ? g->alpha_w : nullptr, g ? g->alpha_b : nullptr);
        part(d, TL.a_H1 + (D - 1) * NT, NT, 0, 0, W);
    }
    return NERFAIL_OK;
}

// ns per TILE (two k16-steps) of each group shape, FITTED to per-workgroup busy times (tools/dw_balance.py): the
// partition's cost unit. [0] exact f32, [1] bf16x3.
const int kTileNs[2][6] = {{7000, 3800, 2200, 800, 780, 1200}, {2643, 1717, 1361, 933, 924, 1115}};

struct DwPlan {
    DwArgs ka;
    DwCombineArgs ca;
    int wgs;
    long slab_floats;
};

// Groups of both networks, the cost partition over the persistent grid and every group's slab range.
int plan_dw(int D, int W, int skip, int64_t M0, const nerfail_mlp_params* g0, int64_t M1, const nerfail_mlp_params* g1,
            int bf16x3, bool need_pointers, DwPlan& P) {
    MlpLayout L;
    NF_REQUIRE(make_layout(D, W, skip, L), "unsupported (D, W)");
    NF_REQUIRE(L.NT == 8, "the LDS-staged weight-gradient kernel covers W = 256");
    const TrainLayout TL = make_train_layout(D, W);
    DwArgs& ka = P.ka;
    DwCombineArgs& ca = P.ca;
    ka.a_slots = TL.a_slots; ka.z_slots = TL.z_slots; ka.pad = 0;
    int ng = 0;
    ka.cum[0] = 0;
    const long tiles[2] = {(long)((M0 + 31) / 32), (long)((M1 + 31) / 32)};
    for (int net = 0; net < 2; ++net) {
        if (tiles[net] == 0) continue;
        const nerfail_mlp_params* gp = net == 0 ? g0 : g1;
        NF_REQUIRE(!need_pointers || gp != nullptr, "gradient table is NULL");
        LinDesc desc[kMaxDesc];
        int nd = 0;
        const int rc = build_descs(D, W, L, TL, need_pointers ? gp : nullptr, desc, nd);
        if (rc != NERFAIL_OK) return rc;
        for (int di = 0; di < nd; ++di)
            for (int p = 0; p < desc[di].nparts; ++p) {
                const LinDesc& d = desc[di];
                const int LA = d.dz_tiles, LB = d.parts[p].ntiles;
                int shape;
                if (LA == 8 && LB == 8) shape = 0;
                else if (LA == 4 && LB == 8) shape = 1;
                else if (LA == 8 && LB == 2) shape = 2;
                else if (LA == 4 && LB == 1) shape = 3;
                else if (LA == 1 && LB == 4) shape = 4;
                else if (LA == 1 && LB == 8) shape = 5;
                else { set_error("nerfail_mlp_bwd_weights: unexpected layer shape for the LDS-staged kernel"); return NERFAIL_EINVAL; }
                NF_REQUIRE(ng < kDwMaxGroups, "too many weight-gradient groups");
                DwGroup& g = ka.grp[ng];
                g.dz_slot0 = d.dz_slot0; g.x_slot0 = d.parts[p].slot0; g.shape = shape;
                g.bias = (p == 0 && (d.gb != nullptr || !need_pointers)) ? 1 : 0;
                g.t0 = net == 0 ? 0 : (int)tiles[0];
                g.t1 = g.t0 + (int)tiles[net];
                g.seg_floats = dw_seg_floats(shape);
                DwCombineGroup& c = ca.grp[ng];
                c.gw = d.gw; c.gb = (p == 0) ? d.gb : nullptr; c.shape = shape; c.seg_floats = g.seg_floats;
                c.row0 = d.row0; c.row1 = d.row1; c.in_f = d.in_f; c.col0 = d.parts[p].col0; c.ncols = d.parts[p].ncols;
                ka.cost[ng] = kTileNs[bf16x3 ? 1 : 0][shape];
                ka.cum[ng + 1] = ka.cum[ng] + (long)ka.cost[ng] * tiles[net];
                ++ng;
            }
    }
    ka.ngroups = ng; ca.ngroups = ng;
    long wgs = cu_count_dw();
    const long min_units = (long)kTileNs[bf16x3 ? 1 : 0][0] * 4;       // at least ~4 full-layer tiles per workgroup
    if (wgs > ka.cum[ng] / min_units) wgs = ka.cum[ng] / min_units > 0 ? ka.cum[ng] / min_units : 1;
    P.wgs = (int)wgs;
    // segments: which workgroups hold a partial of which group (the kernel's own arithmetic), slab ranges
    for (int g = 0; g < ng; ++g) { ka.grp[g].first_wg = -1; ca.grp[g].nseg = 0; }
    for (int wg = 0; wg < P.wgs; ++wg) {
        long lo, hi;
        dw_interval(ka.cum[ng], P.wgs, wg, lo, hi);
        for (int g = 0; g < ng; ++g) {
            int tb, te;
            if (!dw_tiles(lo, hi, ka.cum[g], ka.cum[g + 1], ka.cost[g], tb, te)) continue;
            if (ka.grp[g].first_wg < 0) ka.grp[g].first_wg = wg;
            if (wg != ka.grp[g].first_wg + ca.grp[g].nseg) {
                set_error("nerfail_mlp_bwd_weights: a group's workgroups are not contiguous (partition too fine)");
                return NERFAIL_EINVAL;
            }
            ++ca.grp[g].nseg;
        }
    }
    long off = 0;
    for (int g = 0; g < ng; ++g) {
        if (ca.grp[g].nseg == 0) { set_error("nerfail_mlp_bwd_weights: a group received no workgroup"); return NERFAIL_EINVAL; }
        ka.grp[g].slab = off; ca.grp[g].slab = off;
        off += (long)ca.grp[g].nseg * ka.grp[g].seg_floats;
    }
    P.slab_floats = off;
    return NERFAIL_OK;
}

bool use_lds_dw(int W) {
    const char* dwk = getenv("NERFAIL_DW_KERNEL");
    return W == 256 && (dwk == nullptr || dwk[0] == 'l');
}

// register-fed kernel (float atomics): one network
int launch_dw_reg(int D, int W, int skip, const float* acts, const float* dz, int64_t M, const nerfail_mlp_params* grads,
                  int bf16x3, hipStream_t stream) {
    MlpLayout L;
    NF_REQUIRE(make_layout(D, W, skip, L), "unsupported (D, W)");
    const TrainLayout TL = make_train_layout(D, W);
    WArgs a;
    a.acts = acts; a.dz = dz; a.ntiles = (M + 31) / 32; a.a_slots = TL.a_slots; a.z_slots = TL.z_slots; a.bf16x3 = bf16x3;
    int nd = 0;
    const int rc = build_descs(D, W, L, TL, grads, a.desc, nd);
    if (rc != NERFAIL_OK) return rc;
    a.ndesc = nd;
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
    long wgs = cu_count_dw();
    if (wgs > a.cum[a.ngroups] / 16) wgs = a.cum[a.ngroups] / 16 > 0 ? a.cum[a.ngroups] / 16 : 1;   // tiny problems
    nerf_mlp_bwd_weights_kernel<<<dim3((unsigned)wgs), dim3(256), 0, stream>>>(a);
    NF_LAUNCHED("nerf_mlp_bwd_weights_kernel");
    return NERFAIL_OK;
}

int zero_grads(int D, const nerfail_mlp_params* g, int W, int skip, hipStream_t stream) {
    MlpLayout L;
    NF_REQUIRE(make_layout(D, W, skip, L), "unsupported (D, W)");
    ZeroTable t;
    t.n = 0;
    auto put = [&](const float* p, int n) { t.p[t.n] = const_cast<float*>(p); t.len[t.n] = n; ++t.n; };
    for (int i = 0; i < D; ++i) {
        const bool emb = layer_has_emb(i, L.skip);
        put(g->pts_w[i], W * ((i == 0) ? kPtsCh : (emb ? W + kPtsCh : W)));
        put(g->pts_b[i], W);
    }
    put(g->views_w, (W / 2) * (W + kDirCh)); put(g->views_b, W / 2);
    put(g->feature_w, W * W); put(g->feature_b, W);
    put(g->alpha_w, W); put(g->alpha_b, 1);
    put(g->rgb_w, 3 * (W / 2)); put(g->rgb_b, 3);
    dw_zero_kernel<<<dim3(16, (unsigned)t.n), dim3(256), 0, stream>>>(t);
    NF_LAUNCHED("dw_zero_kernel");
    return NERFAIL_OK;
}

}  // namespace

extern "C" size_t nerfail_mlp_bwd_weights_scratch_bytes(int D, int W, int skip, int64_t M0, int64_t M1, int flags) {
    if (M0 < 0 || M1 < 0 || (M0 == 0 && M1 == 0)) return 0;
    if (!use_lds_dw(W)) return 0;                                      // the register-fed kernel needs none
    DwPlan P;
    if (plan_dw(D, W, skip, M0, nullptr, M1, nullptr, flags & NERFAIL_DW_BF16X3, false, P) != NERFAIL_OK) return 0;
    return (size_t)P.slab_floats * sizeof(float);
}

extern "C" int nerfail_mlp_bwd_weights(int D, int W, int skip, const float* acts, const float* dz, int64_t M0,
                                       const nerfail_mlp_params* grads0, int64_t M1, const nerfail_mlp_params* grads1,
                                       int flags, void* scratch, size_t scratch_bytes, void* stream) {
    NF_REQUIRE(M0 >= 0 && M1 >= 0, "M is negative");
    if (M0 == 0 && M1 == 0) return NERFAIL_OK;
    NF_REQUIRE(acts && dz, "NULL pointer");
    NF_REQUIRE(M1 == 0 || M0 % 32 == 0, "with two networks the first one's sample count must be a multiple of 32");
    NF_REQUIRE((M0 == 0 || grads0 != nullptr) && (M1 == 0 || grads1 != nullptr), "gradient table is NULL");
    const int bf16x3 = (flags & NERFAIL_DW_BF16X3) ? 1 : 0, accumulate = (flags & NERFAIL_DW_ACCUMULATE) ? 1 : 0;
    hipStream_t s = as_stream(stream);
    if (!use_lds_dw(W)) {
        MlpLayout L;
        NF_REQUIRE(make_layout(D, W, skip, L), "unsupported (D, W)");
        const TrainLayout TL = make_train_layout(D, W);
        const long tiles0 = (M0 + 31) / 32;
        for (int net = 0; net < 2; ++net) {
            const int64_t M = net == 0 ? M0 : M1;
            if (M == 0) continue;
            const nerfail_mlp_params* g = net == 0 ? grads0 : grads1;
            int rc = NERFAIL_OK;
            if (!accumulate) rc = zero_grads(D, g, W, skip, s);
            if (rc != NERFAIL_OK) return rc;
            const long t0 = net == 0 ? 0 : tiles0;
            rc = launch_dw_reg(D, W, skip, acts + (size_t)t0 * TL.a_slots * 1024, dz + (size_t)t0 * TL.z_slots * 1024, M, g, bf16x3, s);
            if (rc != NERFAIL_OK) return rc;
        }
        return NERFAIL_OK;
    }
    DwPlan P;
    const int rc = plan_dw(D, W, skip, M0, grads0, M1, grads1, bf16x3, true, P);
    if (rc != NERFAIL_OK) return rc;
    NF_REQUIRE(scratch != nullptr && scratch_bytes >= (size_t)P.slab_floats * sizeof(float),
               "scratch is smaller than nerfail_mlp_bwd_weights_scratch_bytes()");
    P.ka.acts = acts; P.ka.dz = dz; P.ka.slab = static_cast<float*>(scratch);
    P.ca.slab = static_cast<const float*>(scratch); P.ca.accumulate = accumulate;
    if (bf16x3) nerf_mlp_dw_lds_kernel<true><<<dim3((unsigned)P.wgs), dim3(256), 0, s>>>(P.ka);
    else nerf_mlp_dw_lds_kernel<false><<<dim3((unsigned)P.wgs), dim3(256), 0, s>>>(P.ka);
    NF_LAUNCHED("nerf_mlp_dw_lds_kernel");
    nerf_mlp_dw_combine_kernel<<<dim3(64, (unsigned)P.ka.ngroups), dim3(256), 0, s>>>(P.ca);
    NF_LAUNCHED("nerf_mlp_dw_combine_kernel");
    return NERFAIL_OK;
}
