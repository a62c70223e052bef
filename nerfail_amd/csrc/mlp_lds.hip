// K3 + K4, inference form: fused positional encoding + NeRF MLP forward on the exact-f32 matrix cores with the weight
// stream staged ONCE per workgroup through an LDS ring (run_nerf.py:37-51, run_nerf_helpers.py:15-50, :100-123).
//
// Same arithmetic and the same register-resident activation scheme as mlp.hip (every layer transposed, the 32x32
// accumulator of one layer is the B operand of the next; v_mfma_f32_32x32x2_f32 = exact f32 FMA chain; one wave per
// SIMD, 32 samples x all channels per wave). What differs is how the A operands (weights) reach the MFMAs:
//   * mlp.hip: every wave loads every weight fragment itself, global -> VGPR, one quad ahead: 4 x 2.4 MB per CU and
//     tile round from L2, a 64-register ring, 70 spilled registers; measured cost of that delivery: 6.5 % + 1 %.
//   * here: the 4 waves of a workgroup share ONE copy of the stream. The packed image's weight range is copied piece by
//     piece (1 KB = one wave-wide 16-byte access = the A fragments of one (quad, out-tile)) by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no VALU) into a ring of RP pieces cut into S groups; wave w moves pieces
//     w, w+4, ... of every group. A group becomes readable at ONE s_barrier per group (after each wave's counted
//     s_waitcnt vmcnt, which leaves the S-2 younger groups in flight); the slot freed at that barrier is refilled at
//     once, the refill's DMA instructions interleaved with the following MFMAs. The stream never stops: it runs across
//     layer and tile boundaries (the ring does not know about layers), so no part of a layer starts on an exposed load.
//     Barrier positions are compile-time (every layer part is a whole number of groups; the image is padded for it).
//   * the MFMA order inside a quad is tile-major (4 k-steps of one out-tile back to back: the dependent-accumulator
//     latency of this instruction equals its issue time), so a fragment lives for 4 MFMAs and the next step's
//     fragments are read from LDS (ds_read_b128, conflict-free) while the current step's 16 MFMAs run: 32 VGPRs of
//     fragments instead of 64, no spills.
//   * ReLU is applied to a B operand in place right before its quad uses it (hidden between MFMAs) and the two
//     activation arrays swap roles from layer to layer: no copy, no ReLU pass at a layer boundary.
//   * biases and the two thin heads live in a constant LDS area loaded once per workgroup.
// Bound: f32 MFMA (157.3 TFLOP/s dense); 1 186 816 FLOP per sample (D8 W256). LDS: 12*NT KB ring + 14 KB constants + 48 KB parked operands.
#include <type_traits>
#include "mlp_layout.h"

namespace nerfail {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;

#ifndef NF_LDS_SPREAD
#define NF_LDS_SPREAD 1     // 1: one piece of side work per MFMA shadow (WRing::step, inference kernels); 0: round-2 form
#endif
#ifndef NF_LDS_VGPR_P
#define NF_LDS_VGPR_P 1       // activation array P in arch VGPRs (VGPR-form MFMAs by inline asm; inference, W = 256); 0: both in AGPRs
#endif
#ifndef NF_LDS_RING_FIRST
#define NF_LDS_RING_FIRST 1
#endif
#ifndef NF_LDS_BWD_SP1
#define NF_LDS_BWD_SP1 1      // the backward-data ring kernel on the one-piece-of-side-work-per-shadow step form
#endif
#ifndef NF_LDS_TRAIN_NEWDMA
#define NF_LDS_TRAIN_NEWDMA 1 // 1: the training kernel on the buffer-form / per-step refill as well
#endif
#ifndef NF_LDS_TRAIN_SP1
#define NF_LDS_TRAIN_SP1 1    // the training forward on the one-piece-of-side-work-per-shadow step form too (round 6; 0: round-2 form)
#endif
#ifndef NF_LDS_TRAIN_K0
#define NF_LDS_TRAIN_K0 8     // first MFMA of a quad's first step whose shadow carries an activation store (4 stores: K0 .. K0 + 3;
                              // 8: behind the fragment reads of MFMAs 4..7 - measured 6.93 -> 6.89 ms at 786 432 samples; 4: round-3 position)
#endif
#ifndef NF_LDS_MID_SPLIT
#define NF_LDS_MID_SPLIT 0  // 1: the next quad's operand preparation in two halves behind two MFMAs
#endif
#ifndef NF_LDS_DMA_BUF
#define NF_LDS_DMA_BUF 1    // 1: LDS-DMA as buffer_load_dwordx4 ... lds (SGPR base + SGPR piece offset + one 32-bit lane offset)
#endif                      //    0: global_load_lds_dwordx4 (a 64-bit address pair per lane)
#ifndef NF_LDS_DMA_SPREAD
#define NF_LDS_DMA_SPREAD 1 // 1: one refill DMA per step, spread over the group interval (W = 256 only; LdsCfg::kDmaSpread)
#endif
#ifndef NF_LDS_GPM
#define NF_LDS_GPM 4        // pieces per ring group = NF_LDS_GPM * NT (4: one barrier per 4 quads of a W-wide layer)
#endif

template <int N> __device__ __forceinline__ void lds_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ f32x4 lds_read4(const float* p) { return *(lds_cf4*)p; }

template <int NT>
struct LdsCfg {
    static constexpr int HS = NT >= 4 ? 4 : NT;                 // pieces (= out tiles) per step at full width
    static constexpr int GP = NF_LDS_GPM * NT;                  // pieces per group
    static constexpr int RP = 12 * NT;                          // ring pieces (96 KB at W = 256: 3 groups of 32)
    static constexpr int S = RP / GP;                           // groups in the ring
    static constexpr int GPW = GP / 4;                          // LDS-DMA instructions per wave and group
    static constexpr int kViewsPieces = (4 * NT + kDirQuads) * (NT / 2);
    static constexpr int kStreamPad = (4 * NT - kViewsPieces % (4 * NT)) % (4 * NT);   // = MlpLayout::stream_pad
    static constexpr int kAlphaFloats = (NT * 32 + 4 + kPiece - 1) / kPiece * kPiece;
    static constexpr int kRgbFloats = (3 * (NT / 2) * 32 + 4 + kPiece - 1) / kPiece * kPiece;
    static constexpr int kMaxDepth = 8;                         // deeper nets run on the register-streamed kernel (mlp.hip)
    static constexpr int kConstMax = (kMaxDepth + 2) * kPiece + kAlphaFloats + kRgbFloats;
    static constexpr int kParkQuads = kEmbQuads + kDirQuads;    // the 32 + 16 encoding operands of every lane, parked in LDS
    static constexpr int kParkFloats = 4 * 64 * 4 * kParkQuads;
    static_assert(GP % 4 == 0 && RP % GP == 0 && S >= 3, "ring geometry");
    static_assert((S - 2) * GPW <= 48, "vmcnt is a 6-bit counter");
    static_assert(GPW <= 8 && (GPW % 4 == 0 || GPW < 4), "dma() covers 8 pieces per wave and group, in blocks of 4");
    // Round 5: the refill of a freed ring slot is spread over the whole group interval - ONE LDS-DMA per wave and step instead
    // of the group's GPW DMAs in a burst behind the first MFMAs of the boundary step. Per-step clock stamps (tools/lds_steps.py)
    // showed the boundary step 180-400 cycles over its 1024 and an ordinary step with the DMAs removed: with the four waves of
    // the workgroup each issuing 8 KB at the same moment the vector-memory path backs up and a DMA's issue outlasts the 64
    // cycles of the MFMA it stands behind. Needs every part of the net to run HS-piece steps (W = 256: 8 and 4 out tiles),
    // so that "step j of the interval" is a compile-time position everywhere in the stream.
    static constexpr bool kDmaSpread = NF_LDS_DMA_SPREAD && NT == 8 && HS == 4 && (GP / HS) == GPW;
    static constexpr int SPG = GP / HS;                         // steps per group (full-width steps)
};

// The weight stream of one workgroup. Every member but fr / gsrc / rl is wave-uniform (SGPRs).
// SP1: the step form with ONE piece of side work per MFMA shadow (round 5: inference and backward-data kernels; round 6: the
// training forward too - with the buffer-form refill it no longer spills: 227 VGPRs, scratch 0).
template <int NT, bool SP1 = false, bool NEWDMA = false>
struct WRing {
    using C = LdsCfg<NT>;
    // NEWDMA (round 5): the refill as buffer_load ... lds, one DMA per step (LdsCfg::kDmaSpread). All three ring kernels use it
    // (NF_LDS_TRAIN_NEWDMA = 0 restores the round-2 form of the training forward: global_load_lds in a burst behind the boundary step).
    static constexpr bool kBufDma = NEWDMA && NF_LDS_DMA_BUF;
    static constexpr bool kSpreadDma = NEWDMA && C::kDmaSpread;
    const float* gsrc;       // packed + lane*4: this lane's 16 bytes of stream piece 0
    __amdgpu_buffer_rsrc_t rsrc;   // the packed image as a raw buffer (NF_LDS_DMA_BUF)
    int voff;                // lane * 16: this lane's byte offset inside a piece
    float* ring;             // LDS ring base
    const float* rl;         // ring + lane*4
    int total;               // stream length in pieces (multiple of GP)
    int src;                 // next group's first source piece
    int slot;                // ring group the next refill goes to
    int rd;                  // ring piece the next step reads
    int wave;
    f32x4 fr[C::HS];         // fragments of the NEXT step (always C::HS pieces from rd; a narrower step uses the first ones)

    // This wave's i-th piece of the group being refilled. A wave moves GPW CONSECUTIVE pieces (consecutive in the image
    // and in the ring), so one address pair serves four of them through the instruction's immediate offset (applied to
    // the global AND the LDS address): per piece no address arithmetic at all - behind an MFMA that matters, the issue
    // of an LDS-DMA plus five scalar and one 64-bit vector instruction did not fit into the 64 cycles of its shadow.
    template <int I>
    __device__ __forceinline__ void dma_at() const {
        constexpr int B4 = I / 4;                                // which block of 4 pieces (one base each)
        const int first = wave * C::GPW + 4 * B4;
        if constexpr (kBufDma) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(ring + (slot * C::GP + first) * kPiece), 16, voff,
                                                     (src + first) * (kPiece * 4), (I % 4) * kPiece * 4, 0);
        } else {
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc + (size_t)(src + first) * kPiece),
                                             (lds_void_t*)(ring + (slot * C::GP + first) * kPiece), 16, (I % 4) * kPiece * 4, 0);
        }
    }
    __device__ __forceinline__ void dma(int i) const {           // i is a constant after unrolling
        switch (i) {
            case 0: dma_at<0>(); break;   case 1: dma_at<1>(); break;   case 2: dma_at<2>(); break;   case 3: dma_at<3>(); break;
            case 4: dma_at<4>(); break;   case 5: dma_at<5>(); break;   case 6: dma_at<6>(); break;   case 7: dma_at<7>(); break;
            default: break;
        }
    }
    __device__ __forceinline__ void group_issued() {
        src += C::GP;
        if (src >= total) src = 0;
        slot = (slot + 1 == C::S) ? 0 : slot + 1;
    }
    // Group boundary: every wave's pieces of the next group have landed (counted wait: the S-2 younger groups stay
    // in flight), every wave's reads of the group just finished have returned -> one barrier makes the first
    // readable for all and the second's slot free for all.
    // EXTRA: vector-memory operations that are NOT LDS-DMAs (training stores) and may stay in flight across the boundary.
    // vmcnt retires in issue order and counts every vector-memory operation, so "vmcnt(N)" = "everything but the N youngest has
    // completed". Call INTERVAL i the steps from boundary i (inclusive: the SYNC step that crossed it) to boundary i + 1. In
    // interval i this wave issues exactly GPW refill DMAs - the pieces of group i + 2 into the slot boundary i freed - whatever the
    // refill form: burst (all GPW behind the first MFMAs of the SYNC step) or spread (kSpreadDma: DMA J behind MFMA 0 of step J,
    // J = 0 .. SPG - 1 = GPW - 1; the SYNC step's own DMA is issued AFTER it crossed the boundary) - plus the interval's X stores,
    // interleaved with them in any order. Boundary i + 1 needs group i + 1 landed: its DMAs were issued in interval i - 1, i.e. they
    // are older than EVERY operation of interval i. Hence N = GPW + X is exact, any N <= GPW + X is safe (waits for more than
    // necessary), and N > GPW + X would let a DMA of the group about to be read stay in flight. EXTRA must therefore be a LOWER
    // bound of the stores of every interval that ends in the part (lds_part derives it; it does not depend on the refill form
    // nor on where inside a step the stores stand - NF_LDS_TRAIN_K0).
    template <int EXTRA = 0>
    __device__ __forceinline__ void boundary() const {
        static_assert((C::S - 2) * C::GPW + EXTRA <= 63, "vmcnt is a 6-bit counter");
        lds_wait_vmcnt<(C::S - 2) * C::GPW + EXTRA>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int t = 0; t < C::HS; ++t) fr[t] = lds_read4(rl + (rd + t) * kPiece);
    }
    __device__ __forceinline__ void start() {                   // S-1 groups in flight, group 0 readable, ring full
#pragma unroll
        for (int g = 0; g < C::S - 1; ++g) {
#pragma unroll
            for (int i = 0; i < C::GPW; ++i) dma(i);
            group_issued();
        }
        boundary();
        if constexpr (kSpreadDma) {
            dma(0);                  // the state right behind a boundary step: DMA 0 of the next group issued, 1.. follow step by step
        } else {
#pragma unroll
            for (int i = 0; i < C::GPW; ++i) dma(i);
            group_issued();
        }
        prefetch();
    }
    // One step: consume HSP pieces (out tiles) x 4 k-steps; mf(t, e, a) issues the MFMA of tile t, k-step e with A
    // operand a. SYNC: this step ends a group - cross the boundary BEFORE prefetching the next step's fragments (they
    // belong to the next group) and refill the freed slot behind the first MFMAs.
    // The schedule is pinned with sched_barrier(0): left alone, hipcc sinks the prefetch next to its use (every step then
    // starts on an exposed ds_read), hoists a whole layer's lazy ReLUs to the front (259 spilled registers) and puts all
    // refill DMAs in front of the MFMAs. The prefetch sits BEHIND the first tile's 4 MFMAs: hipcc answers the first use of
    // `cur` with s_waitcnt lgkmcnt(0), which is free only while the reads of the NEXT step have not been issued yet.
    // pre() runs before the MFMAs (work the step itself needs), mid() behind the first tile's MFMAs and the prefetch
    // (work for LATER steps: the next quad's B operands).
    // post(k) runs behind MFMA k of the step (training: ONE activation store per MFMA - four stores issued back to back
    // drained the matrix pipe: a global store takes about as long to issue as an MFMA runs).
    // J (kDmaSpread): position of this step in the refill interval that began at the last boundary (0 = the SYNC step itself):
    // it issues DMA J of the group, the interval's last step closes the group.
    // (J is a constant after unrolling, like dma()'s argument.)
    template <int HSP, bool SYNC, int EXTRA = 0, class PRE, class MID, class MF, class POST>
    __device__ __forceinline__ void step(const int J, PRE&& pre, MID&& mid, MF&& mf, POST&& post) {
        constexpr bool SPREAD = kSpreadDma;
        static_assert(!SPREAD || HSP == C::HS, "spread refill: full-width steps");
        f32x4 cur[HSP];
#pragma unroll
        for (int t = 0; t < HSP; ++t) cur[t] = fr[t];
        rd += HSP;
        if (rd >= C::RP) rd = 0;
        if (SYNC) boundary<EXTRA>();
        __builtin_amdgcn_sched_barrier(0);
        pre();
        int k = 0;
        if constexpr (SP1 && HSP == 4 && C::HS == 4) {
            // Round 5: ONE piece of side work per MFMA shadow. The round-2 form put the whole prefetch (address arithmetic + four
            // ds_read_b128) and the next quad's operand preparation behind MFMA 3: ten-odd instructions whose issue takes longer
            // than the 64 cycles MFMA 4 needs the pipe for, so MFMA 5 started late (measured: tools/lds_steps.py). Here every
            // MFMA is pinned, tile-major (a dependent MFMA issues when its predecessor leaves the pipe), and its shadow carries at
            // most one of: a refill DMA (SYNC steps: behind MFMAs 0..GPW-1), one fragment read of the next step, the operand
            // preparation. The fragment reads start behind MFMA 3 at the earliest: the first use of `cur` is answered with
            // s_waitcnt lgkmcnt(0), free only while no read of the NEXT step has been issued.
            constexpr int R0 = (SYNC && !SPREAD) ? (C::GPW > 4 ? C::GPW : 4) : 4;      // first MFMA whose shadow takes a fragment read
            static_assert(R0 + 4 < 16, "fragment reads and operand preparation fit behind this step's MFMAs");
#pragma unroll
            for (int t = 0; t < HSP; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mf(t, e, cur[t][e]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (SPREAD ? k == 0 : (SYNC && k < C::GPW)) dma(SPREAD ? J : k);
                    post(k);
                    if (k >= R0 && k < R0 + 4) fr[k - R0] = lds_read4(rl + (rd + (k - R0)) * kPiece);
#if NF_LDS_MID_SPLIT
                    if (k == R0 + 4) mid(-2);                             // operands 0, 1 of the next quad
                    if (k == R0 + 5) mid(-3);                             // operands 2, 3
#else
                    if (k == R0 + 4) mid(-1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    ++k;
                }
            }

        } else {
#pragma unroll
        for (int t = 0; t < HSP; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mf(t, e, cur[t][e]);
                if (SPREAD ? k == 0 : (SYNC && k < C::GPW)) {   // burst form: one refill DMA behind each of the first MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    dma(SPREAD ? J : k);
                    __builtin_amdgcn_sched_barrier(0);
                }
                post(k);
                ++k;
            }
            if (NF_LDS_SPREAD == 2 && HSP == 4 && C::HS == 4) {     // the prefetch in two halves: behind tile 0 and behind tile 1
                if (t == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    fr[0] = lds_read4(rl + (rd + 0) * kPiece);
                    fr[1] = lds_read4(rl + (rd + 1) * kPiece);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (t == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    fr[2] = lds_read4(rl + (rd + 2) * kPiece);
                    fr[3] = lds_read4(rl + (rd + 3) * kPiece);
                    mid(-1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (t == 0) {
                __builtin_amdgcn_sched_barrier(0);
                prefetch();
                mid(-1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        if constexpr (SPREAD) {
            if (J == C::SPG - 1) group_issued();
        } else if (SYNC) {
#pragma unroll
            for (int i = 4 * HSP; i < C::GPW; ++i) dma(i);
            group_issued();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int HSP, bool SYNC>
    __device__ __forceinline__ void skip(const int J) {          // padding pieces: no MFMAs
        rd += HSP;
        if (rd >= C::RP) rd = 0;
        if (SYNC) boundary();
        if constexpr (kSpreadDma) {
            static_assert(HSP == C::HS, "spread refill: full-width steps");
            dma(J);
            if (J == C::SPG - 1) group_issued();
        } else if (SYNC) {
#pragma unroll
            for (int i = 0; i < C::GPW; ++i) dma(i);
            group_issued();
        }
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
    }
};

// One part of a layer: NQ quads of 4 k-steps over OT out tiles, then PAD padding pieces.
//   bprep(q, b): the 4 B operands of quad q (k-steps 4q..4q+3) into b[] - for an activation array that is the lazy ReLU
//                of 4 accumulator registers. Prepared ONE QUAD AHEAD, behind the first MFMAs of the previous quad's last
//                step: an accumulator register reaches the VALU only through v_accvgpr_read, which goes through the
//                matrix pipe (~10 cycles each even when nothing waits for it, tools/clockprobe/mfma_patterns.hip); read
//                right in front of the MFMA that needs it, it cost 24 (70.0 instead of 66.5 cycles per MFMA).
//   hook(q):     side work at the start of quad q (bias tiles of the next layer into dead registers).
//   tstore(q, e, b): training only - operand e of quad q leaves for HBM (saved activations) and enters the ReLU bit mask.
//                Issued behind MFMAs 4..7 of the quad's FIRST step, one per MFMA: b[] is live for the whole quad anyway,
//                and a store issued there is old enough at the next group boundary (the boundary's counted vmcnt leaves
//                only the 8 youngest vector-memory operations in flight; stores count like LDS-DMAs, in order).
struct NoStore { __device__ __forceinline__ void operator()(int, int, const float (&)[4]) const {} };
// ZERO0: the accumulators are not read - the first MFMA of every out tile takes C = 0 (the backward-data layers: no zero fill).
// VG (round 5): the output array lives in arch VGPRs - its MFMAs are issued in the VGPR form, so that the NEXT layer reads its B
// operands without v_accvgpr_read (an accumulator read costs ~9-13 cycles that nothing hides). hipcc selects ONE form per
// function, hence inline asm. What the compiler then no longer knows is that the statement is an MFMA; the hazards it would
// have covered do not arise in this use: the accumulate chain reads SrcC = its own vDst back to back (forwarded by the
// hardware, as in every VGPR-form GEMM), A / B operands come from ds_read (waited for as asm inputs) and v_max (no VALU ->
// XDL SrcA/B hazard on gfx940+), and a VALU / LDS access to an MFMA result follows its last MFMA by at least a whole step
// (the 19 wait states a 16-pass XDL write needs are 76 cycles). Round 2's attempt put two wait states in front of every such
// MFMA and lost what the form gains; test_lds_streaming_kernel_equals_register_streamed_kernel holds the bits.
template <int NT, int OT, int NQ, int PAD, bool ZERO0 = false, bool VG = false, int NIN = 0, class Ring, class Hook, class BPrep, class TStore = NoStore>
__device__ __forceinline__ void lds_part(Ring& st, f32x16 (&acc)[NIN], Hook hook, BPrep bprep, TStore tstore = TStore()) {
    using C = LdsCfg<NT>;
    constexpr int HSP = OT >= 4 ? 4 : OT, SPQ = OT / HSP;
    static_assert((NQ * OT + PAD) % C::GP == 0, "a part is a whole number of ring groups");
    float bq[NQ + 1][4];                                              // per-quad operands: plain registers after unrolling
    bprep(0, bq[0]);                                                  // (the first quad's cannot be early: its source is the layer before)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int sp = 0; sp < SPQ; ++sp) {
            const int done = (q * SPQ + sp + 1) * HSP;
            auto mf = [&](int t, int e, float a) {
                if (ZERO0 && q == 0 && e == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[sp * HSP + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[q][e], zero, 0, 0, 0);
                } else if constexpr (VG) {
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[sp * HSP + t]) : "v"(a), "v"(bq[q][e]));
                } else {
                    acc[sp * HSP + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[q][e], acc[sp * HSP + t], 0, 0, 0);
                }
            };
            auto pr = [&]() { if (sp == 0) hook(q); };
            auto mid = [&](int e) {                                  // e < 0: all four operands of the next quad; else operand e only
                if (sp == SPQ - 1 && q + 1 < NQ) {
                    if (e == -1) bprep(q + 1, bq[q + 1]);
                    else if (e < -1) {                               // halves: -2 -> operands 0, 1; -3 -> operands 2, 3
                        float tmp[4];
                        bprep(q + 1, tmp);
                        const int e0 = e == -2 ? 0 : 2;
                        bq[q + 1][e0] = tmp[e0];
                        bq[q + 1][e0 + 1] = tmp[e0 + 1];
                    } else {
                        float tmp[4];                                // (the three unused elements are dead code after inlining)
                        bprep(q + 1, tmp);
                        bq[q + 1][e] = tmp[e];
                    }
                }
            };
            constexpr int K0 = 4 * HSP >= 8 ? (4 * HSP >= NF_LDS_TRAIN_K0 + 4 ? NF_LDS_TRAIN_K0 : 4) : 0;   // behind MFMAs 4..7 (a one-tile step has only 0..3)
            auto post = [&](int k) {
                if constexpr (!std::is_same<TStore, NoStore>::value) {
                    if (sp == 0 && k >= K0 && k < K0 + 4) {
                        tstore(q, k - K0, bq[q]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            // Training stores of a part: 4 per quad, on the quad's first step. They may stay in flight across a boundary
            // together with the youngest group's DMAs (WRing::boundary: vmcnt(GPW + EXTRA), EXTRA = a lower bound of the stores
            // issued in the interval that ends at the boundary) - otherwise every boundary would wait for stores issued ~1.7 us
            // earlier to be acknowledged.
            // Counting them (re-derived for the spread refill in round 6 - the count is the same for both refill
            // forms): the interval that ends at a boundary of this part runs from the SYNC step that crossed the previous
            // boundary up to (not including) the SYNC step at this one. SPQ == 2 (W = 256 full-width layers): the SYNC step is
            // the store-less second step of a quad, the GP/OT quads in between each carry 4 stores on their first step -> 4*GP/OT
            // whatever part came before. SPQ == 1 (W <= 128, and the W/2-wide views part at W = 256): every step carries stores,
            // also the SYNC step itself - which belongs to the PREVIOUS part at a part's first boundary, and that part may be a
            // store-less encoding part (or the store-less second step of a full-width quad): only GP/OT - 1 quads of stores are
            // guaranteed. One quad fewer is allowed to stay in flight there (4 stores issued ~3 steps earlier must have been
            // acknowledged - free in practice), so that never a DMA of the group about to be read is still outstanding.
            constexpr int EXTRA = std::is_same<TStore, NoStore>::value ? 0 : 4 * (C::GP / OT - (SPQ == 1 ? 1 : 0));
            static_assert(EXTRA >= 0, "a store-carrying part spans at least one quad per group");
            static_assert(std::is_same<TStore, NoStore>::value || SPQ == 1 || (C::GP / OT) * SPQ == C::SPG || !Ring::kSpreadDma,
                          "spread refill: an interval is SPG full-width steps = GP / OT quads of SPQ steps, 4 stores on each quad's first");

            const int J = Ring::kSpreadDma ? (done / HSP) % C::SPG : 0;        // (done / HSP - 1 is this step's index in the part)
            if (done % C::GP == 0) st.template step<HSP, true, EXTRA>(J, pr, mid, mf, post);
            else st.template step<HSP, false>(J, pr, mid, mf, post);
        }
    }
    // VG: nothing may read an MFMA result in VGPRs for 19 wait states after a 16-pass MFMA was issued, and hipcc - which does not
    // know the asm statements are MFMAs - is free to put a register copy of the array right behind the part's last one (the
    // backward-data kernel with its X array in VGPRs, tried in round 5, produced wrong bits; the cause was not isolated, such
    // a copy is the suspected path). 20 wait states close it.
    if constexpr (VG) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int p = 0; p < PAD / HSP; ++p) {
        const int done = NQ * OT + (p + 1) * HSP;
        const int J = Ring::kSpreadDma ? (done / HSP) % C::SPG : 0;
        if (done % C::GP == 0) st.template skip<HSP, true>(J);
        else st.template skip<HSP, false>(J);
    }
}

// max(x, 0) as ONE integer instruction on the bit pattern (negative floats are negative integers; -0 -> +0). fmaxf costs
// two: hipcc first canonicalises an accumulator value (v_max x, x) before the IEEE-mode v_max with 0.
__device__ __forceinline__ float relu_bits(float x) {
    const int i = __float_as_int(x);
    return __int_as_float(i > 0 ? i : 0);
}

// ---- training stores. Written as inline asm in the "SGPR base + 32-bit lane offset + immediate" form: ONE VGPR (the lane's
// offset inside a slot) serves every saved value of the kernel. Left to hipcc, each destination became a 64-bit per-lane
// pointer pair that was hoisted and spilled, and a scratch reload drains the LDS-DMA queue (vmcnt retires in order).
// (The register's byte offset rides in the instruction's immediate: an "i" operand, constant once the quad loops are
// unrolled - "n" would demand a front-end constant, a 16-way switch per store kept hipcc from unrolling, and adding the
// offset in a VGPR cost a v_mov + v_add + s_nop per store in an MFMA shadow that is 64 cycles long.)
__device__ __forceinline__ void st_acc_reg(const float* sbase, unsigned voff, int r, float v) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(v), "s"(sbase), "i"(acc_reg_off(r) * 4) : "memory");
}
// the 16 ReLU bits of one accumulator tile: entry's [64 lanes][4 dwords] image, tile t = 16-bit field t of the lane
__device__ __forceinline__ void st_mask16(const float* sentry, unsigned vlane16, int t, unsigned bits) {
    asm volatile("global_store_short %0, %1, %2 offset:%3" ::"v"(vlane16), "v"(bits), "s"(sentry), "i"(2 * t) : "memory");
}

// dot product of a thin head's weights (LDS, accumulator order [OT][2][16]) with relu(x): VALU + one cross-half shuffle.
// One tile at a time (sched_barrier): left alone, hipcc reads all 128 accumulators into VGPRs first and the register
// allocator answers by spilling the positional encodings across the whole layer loop.
template <int OT, int NIN>
__device__ __forceinline__ float lds_head(const f32x16 (&x)[NIN], const float* w, int h) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 wv = lds_read4(w + (t * 2 + h) * 16 + 4 * r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = x[t][4 * r4 + e];
                asm("" : "+v"(v));     // opaque copy: otherwise hipcc shares these ReLUs with the next layer's operand
                                       // preparation and keeps all 128 results alive in between (spills)
                s = fmaf(wv[e], relu_bits(v), s);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return s + __shfl_xor(s, 32, 64);
}

// Three heads over the same input (rgb_linear's three rows): every accumulator register is read ONCE (round 5: three calls of
// lds_head moved each of the 64 registers to a VGPR three times - an accumulator read goes through the matrix pipe, ~13 cycles).
// Each sum is the same fma chain in the same order as lds_head's: the bits do not change.
template <int OT, int NIN>
__device__ __forceinline__ void lds_head3(const f32x16 (&x)[NIN], const float* w, int stride, int h, float (&out)[3]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float* p = w + (t * 2 + h) * 16 + 4 * r4;
            const f32x4 w0 = lds_read4(p), w1 = lds_read4(p + stride), w2 = lds_read4(p + 2 * stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = x[t][4 * r4 + e];
                asm("" : "+v"(v));
                v = relu_bits(v);
                s0 = fmaf(w0[e], v, s0);
                s1 = fmaf(w1[e], v, s1);
                s2 = fmaf(w2[e], v, s2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[0] = s0 + __shfl_xor(s0, 32, 64);
    out[1] = s1 + __shfl_xor(s1, 32, 64);
    out[2] = s2 + __shfl_xor(s2, 32, 64);
}

// SKIP: where the skip connection's extra part is compiled in: 0 nowhere, 1 first / 2 second layer of a pair.
// TRAIN: additionally saves what the backward needs (mlp_layout.h: encodings, every layer's post-ReLU tile, the feature
// tile, ReLU bit masks) - same values, same slots as the register-streamed nerf_mlp_fwd_kernel<NT, true>.
template <int NT, int SKIP, bool TRAIN>
__global__ __launch_bounds__(256, 1) void nerf_mlp_fwd_lds_kernel(MlpArgs a) {
    using C = LdsCfg<NT>;
    constexpr int OTV = NT / 2;
    // ONE object (a second __shared__ array beside an LDS-DMA target makes hipcc drain vmcnt before LDS reads). Constants
    // first: their addresses are then 'lane part + immediate offset' (16-bit ds offsets) instead of one hoisted - and
    // spilled - VGPR per bias tile.
    __shared__ __attribute__((aligned(16))) float smem[C::kConstMax + C::kParkFloats + C::RP * kPiece];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const MlpLayout& L = a.lay;
#if NF_LDS_RING_FIRST
    // the ring at offset 0: a step's four fragment reads are then ONE address (lane * 16 + ring piece * 1024) + immediates
    // 0 .. 3072 - with the ring behind the constants its last two reads fell outside the 16-bit offset and every step paid a
    // second vector add (580 per tile; vector instructions are never hidden beside the f32 MFMA)
    float* const ring0 = smem;
    float* const cst = smem + C::RP * kPiece;
    float* const park = cst + C::kConstMax + wave * (64 * 4 * C::kParkQuads) + lane * 4;
#else
    float* const cst = smem;
    float* const park = smem + C::kConstMax + wave * (64 * 4 * C::kParkQuads) + lane * 4;
    float* const ring0 = smem + C::kConstMax + C::kParkFloats;
#endif
    {   // constant area: biases (one piece per layer), alpha head, rgb head
        const int n = (int)(L.total - L.b_off[0]);
        const float* __restrict__ g = a.packed + L.b_off[0];
        for (int i = tid * 4; i < n; i += 1024) *reinterpret_cast<f32x4*>(cst + i) = *reinterpret_cast<const f32x4*>(g + i);
    }
    __syncthreads();
    const float* const c_alpha = cst + (L.alpha_off - L.b_off[0]);
    const float* const c_rgb = cst + (L.rgb_off - L.b_off[0]);

    WRing<NT, NF_LDS_SPREAD == 1 && (!TRAIN || NF_LDS_TRAIN_SP1), !TRAIN || NF_LDS_TRAIN_NEWDMA> st;
    st.gsrc = a.packed + lane * 4; st.ring = ring0; st.rl = ring0 + lane * 4;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, (int)(L.w_total * 4), 0x00020000);   // raw buffer: the weight range
    st.voff = lane * 16;
    st.total = (int)(L.w_total / kPiece); st.src = 0; st.slot = 0; st.rd = 0; st.wave = wave;
    st.start();

    // Two activation arrays swap roles from layer to layer (no copies, no ReLU pass). A layer's bias is written into
    // its output array while that array is still the INPUT of the layer before: tile k is dead once that layer has
    // consumed its quads 4k..4k+3, so the 4 LDS reads + 16 accumulator writes of a bias tile hide between its MFMAs
    // instead of standing in front of the next layer (bias loads were 1 % of the register-streamed kernel).
    f32x16 P[NT], Q[NT];
    constexpr bool VGP = NF_LDS_VGPR_P && !TRAIN && NT == 8;                             // array P in arch VGPRs (experiment)
    auto bias_tile = [&](f32x16 (&dst)[NT], int l, int t, const bool in_vgpr = false) {  // dst[t] = bias of layer l, tile t
        const float* p = cst + l * kPiece + (t * 2 + h) * 16;
        const f32x4 v0 = lds_read4(p), v1 = lds_read4(p + 4), v2 = lds_read4(p + 8), v3 = lds_read4(p + 12);
        dst[t] = (f32x16){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3],
                          v2[0], v2[1], v2[2], v2[3], v3[0], v3[1], v3[2], v3[3]};
        // pinned into accumulator registers HERE (round 5; the LDS read then targets them directly): left alone hipcc keeps bias
        // tiles in VGPRs whenever it has some to spare and moves them over with v_accvgpr_write in front of a layer's first
        // MFMAs (280-320 such moves in the kernel's code, 0 with the pin). Vector instructions are never free beside the
        // f32 MFMA (see the note on the encoding-overlap experiment in DESIGN.md): every one removed is time gained.
        if (VGP && in_vgpr) asm volatile("" : "+v"(dst[t]));
        else asm volatile("" : "+a"(dst[t]));
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) bias_tile(P, 0, t, true);                               // later rounds: written by the views layer

    // 32-bit tile counters (the launcher checks M < 2^36): a 64-bit "tile < ntiles" has no scalar compare, so ntiles
    // was copied into a VGPR pair that then lived - in scratch - through the whole kernel
    const int ntiles = (int)((a.M + 31) / 32);
    const int nrounds = (int)((ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4));
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        // every wave of the workgroup walks the whole stream every round (it moves a quarter of it): a wave without a
        // tile of its own recomputes the last tile and stores nothing
        const int tile_own = (int)(((long)rnd * gridDim.x + blockIdx.x) * 4 + wave);
        const int tile = tile_own < ntiles ? tile_own : ntiles - 1;
        // (lane-derived values are taken through an opaque copy per tile: hoisted out of the tile loop - "M - j", byte
        // offsets of the lane, a zero for address extension - they had to live across it and were spilled to scratch)
        int jj = j;
        asm volatile("" : "+v"(jj));
        const long sraw = (long)tile * 32 + jj;
        const long s = sraw < a.M ? sraw : a.M - 1;

        float emb[4 * kEmbQuads], demb[4 * kDirQuads];
        int hh = h;
        asm volatile("" : "+v"(hh));
        encode_sample(a, s, hh, emb, demb);
        // The encoding operands are needed at layer 0 (now), at the skip layer (points) and at the views layer (directions),
        // 5 and 9 layers from here. Kept in registers they were spilled to scratch, and a scratch reload drains the
        // LDS-DMA queue (vmcnt retires in order): parked in LDS instead, one conflict-free 16-byte access per quad.
#pragma unroll
        for (int k = 0; k < kEmbQuads; ++k)
            *reinterpret_cast<f32x4*>(park + k * 256) = (f32x4){emb[4 * k], emb[4 * k + 1], emb[4 * k + 2], emb[4 * k + 3]};
#pragma unroll
        for (int k = 0; k < kDirQuads; ++k)
            *reinterpret_cast<f32x4*>(park + (kEmbQuads + k) * 256) = (f32x4){demb[4 * k], demb[4 * k + 1], demb[4 * k + 2], demb[4 * k + 3]};
        // training: this tile's slots. A wave without a tile of its own recomputes the last tile and writes the SAME bytes
        // again (no branch in the pinned schedule).
        float* __restrict__ A = nullptr;                                                  // wave-uniform
        unsigned voff = 0, vlane16 = 0;                                                   // lane offsets (bytes) in a slot / a mask entry
        if (TRAIN) {
            A = a.acts + (size_t)tile * (train_a_slots(L.D, NT) * 1024);
            int l2 = lane;
            asm volatile("" : "+v"(l2));                                                  // recomputed per tile (see jj above)
            voff = (unsigned)acc_lane_off(l2) * 4u;
            vlane16 = (unsigned)l2 * 16u;
            store_enc<10, 4 * kEmbQuads>(A, emb, lane);                                   // E0 E1: 63 channels
            store_enc<4, 4 * kDirQuads>(A + 2 * 1024, demb, lane);                        // V: 27 channels
        }
        const float* const Amask = TRAIN ? A + train_mask_slot0(L.D, NT) * 1024 : nullptr;
        auto b_park = [&](int q, float (&b)[4]) {                                         // quad q of the parked operands
            const f32x4 v = lds_read4(park + q * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = v[e];
        };
        // layer 0: 63 -> W into P (its bias is already there); Q (dead) receives the bias of layer 1 meanwhile
        lds_part<NT, NT, kEmbQuads, 0, false, VGP>(st, P, [&](int q) { if (q < NT) bias_tile(Q, 1, q); },
            [&](int q, float (&b)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = emb[4 * q + e];
            });

        // pts_linears[l] (l < D) / feature_linear (l == D): out += W_l relu(in). While it runs, `in` receives the bias of
        // layer l+1 tile by tile as its tiles die (for l == D that is the views layer: its W/2 channels use the first
        // tiles, the others get unused padding of the piece).
        float alpha = 0.f;
        auto layer = [&](f32x16 (&in)[NT], f32x16 (&out)[NT], int l, bool may_skip, bool may_be_last, auto out_is_p) {
            constexpr bool OVG = VGP && decltype(out_is_p)::value;                        // out == P: VGPR-form MFMAs; else in == P
            if (may_be_last && l == L.D) alpha = lds_head<NT>(in, c_alpha, h) + c_alpha[NT * 32];   // alpha_linear on relu(h) (RH:110)
            if (may_skip && l == L.skip + 1)                                              // h = cat([input_pts, h]) (RH:106-107)
                lds_part<NT, NT, kEmbQuads, 0, false, OVG>(st, out, [](int) {}, b_park);
            unsigned mk16 = 0u;                                                           // ReLU bits of the tile being consumed
            const float* const Hl = TRAIN ? A + (3 + (l - 1) * NT) * 1024 : nullptr;      // H_l = relu(in): layer l's input
            const float* const Ml = TRAIN ? Amask + (l - 1) * 256 : nullptr;              // its bit-mask entry
            auto bp = [&](int q, float (&b)[4]) {                                         // lazy ReLU of quad q's 4 accumulator registers
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = relu_bits(in[q >> 2][4 * (q & 3) + e]);
            };
            auto hk = [&](int q) { if ((q & 3) == 0 && q > 0) bias_tile(in, l + 1, (q >> 2) - 1, !decltype(out_is_p)::value); };
            if constexpr (TRAIN) {
                lds_part<NT, NT, 4 * NT, 0>(st, out, hk, bp, [&](int q, int e, const float (&b)[4]) {
                    const int r = 4 * (q & 3) + e;
                    st_acc_reg(Hl + (q >> 2) * 1024, voff, r, b[e]);
                    // this value's ReLU bit appended: compare into VCC, then mk16 = 2 mk16 + carry (the 16 bits of a tile arrive in
                    // the order r = 0 .. 15, so they end up reversed: one v_bfrev per tile). Round 6: 0.8 % of the training forward
                    // against v_med3_i32 + v_lshl_or_b32 (profiles/r06_train_mask_ab.log); same bits as relu_bit() << r.
                    asm volatile("v_cmp_lt_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mk16) : "v"(b[e]) : "vcc");
                    if (r == 15) {
                        st_mask16(Ml, vlane16, q >> 2, __builtin_bitreverse32(mk16) >> 16);
                        mk16 = 0u;
                    }
                });
            } else {
                lds_part<NT, NT, 4 * NT, 0, false, OVG>(st, out, hk, bp);
            }
            bias_tile(in, l + 1, NT - 1, !decltype(out_is_p)::value);
        };
#pragma unroll 1
        for (int l = 1; l < L.D; l += 2) {                                                // D is even (host check): whole pairs
            layer(P, Q, l, SKIP == 1, false, std::false_type{});
            layer(Q, P, l + 1, SKIP == 2, true, std::true_type{});
        }
        // views_linears[0]: cat([feature, embedded dirs]) -> W/2 into Q's first tiles (RH:112-116; its ReLU is applied by
        // the rgb head); P, its input, receives the bias of the NEXT tile's layer 0 as it dies
        {
            auto hk = [&](int q) { if ((q & 3) == 0 && q > 0) bias_tile(P, 0, (q >> 2) - 1, true); };
            auto bp = [&](int q, float (&b)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = P[q >> 2][4 * (q & 3) + e];
            };
            if constexpr (TRAIN) {
                const float* const Fl = A + (3 + L.D * NT) * 1024;                        // F = feature_linear's output (no activation)
                lds_part<NT, OTV, 4 * NT, 0>(st, Q, hk, bp, [&](int q, int e, const float (&b)[4]) {
                    st_acc_reg(Fl + (q >> 2) * 1024, voff, 4 * (q & 3) + e, b[e]);
                });
            } else {
                lds_part<NT, OTV, 4 * NT, 0>(st, Q, hk, bp);
            }
        }
        bias_tile(P, 0, NT - 1, true);
        lds_part<NT, OTV, kDirQuads, C::kStreamPad>(st, Q, [](int) {}, [&](int q, float (&b)[4]) { b_park(kEmbQuads + q, b); });
        if constexpr (TRAIN) {                                                            // HV = relu(views output) + its bit mask
            const float* const Vl = A + (3 + (L.D + 1) * NT) * 1024;
#pragma unroll
            for (int t = 0; t < OTV; ++t) {
                unsigned mv = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = Q[t][r];
                    asm("" : "+v"(v));
                    v = relu_bits(v);
                    st_acc_reg(Vl + t * 1024, voff, r, v);
                    asm volatile("v_cmp_lt_i32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mv) : "v"(v) : "vcc");   // (as in layer())
                }
                st_mask16(Amask + L.D * 256, vlane16, t, __builtin_bitreverse32(mv) >> 16);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (OTV & 1) st_mask16(Amask + L.D * 256, vlane16, OTV, 0u);                 // the dword's unused half, as store_mask writes it
        }
        float rgb[3];                                                                     // rgb_linear: W/2 -> 3 (RH:118)
        lds_head3<OTV>(Q, c_rgb, OTV * 32, h, rgb);
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] += c_rgb[3 * OTV * 32 + c];
        int je = j;                                  // recomputed: the sample index need not live through the tile
        asm volatile("" : "+v"(je));
        const long sout = (long)tile * 32 + je;
        if (h == 0 && sout < a.M && tile_own < ntiles)
            reinterpret_cast<float4*>(a.raw)[sout] = make_float4(rgb[0], rgb[1], rgb[2], alpha);
    }
    lds_wait_vmcnt<0>();       // no LDS-DMA may be in flight when the workgroup's LDS is released
}

// ---------------------------------------------------------------------------------------------------------------------
// K4b backward-data on the LDS weight ring (round 5): dX = W^T dZ through all layers, the structure of the training forward
// mirrored (mlp_bwd.hip holds the register-streamed form; both write the same bits). What changed against that form:
//   * the transposed image (laid out in consumption order: views, feature, pts D-1 .. 1; every layer a whole number of ring
//     groups) reaches the MFMAs through the workgroup's LDS ring - one copy per CU instead of four L2 streams;
//   * NO pass between layers. The register form masked a layer's 128 accumulator registers in place, stored them, and zeroed the
//     next 128: ~400 accumulator moves and 128 exposed stores per layer with the matrix pipe idle. Here dZ_i = d_h_{i+1} *
//     [h_{i+1} > 0] is formed LAZILY where it is consumed - the B operand of quad q is four accumulator registers ANDed with
//     their ReLU bits (the lazy ReLU of the forward) - and leaves for HBM from those VGPRs behind MFMAs 4..7 of the quad's
//     first step (the training forward's activation stores); the output array starts from C = 0 in its first MFMA;
//   * a workgroup serves ONE network (the ring holds one image): workgroups [0, blocks0) walk the first network's tiles, the
//     others the second's.
// Same products, same order of additions per accumulator: the stored dZ are bitwise those of nerf_mlp_bwd_data_kernel.
template <int NT>
__global__ __launch_bounds__(256, 1) void nerf_mlp_bwd_data_lds_kernel(BwdArgs a) {
    using C = LdsCfg<NT>;
    constexpr int OTV = NT / 2;
    constexpr int kHeadFloats = (NT * 32 + 3 * OTV * 32 + 3) / 4 * 4;       // alpha head, rgb head (forward image order)
    __shared__ __attribute__((aligned(16))) float smem[kHeadFloats + C::RP * kPiece];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const MlpLayout& L = a.lay;
    const TrainLayout& TL = a.tl;
    const bool second = (int)blockIdx.x >= a.blocks0;                       // workgroup-uniform
    const float* __restrict__ P = second ? a.packed2 : a.packed;
    const float* __restrict__ PT = second ? a.packedT2 : a.packedT;
    const long ntiles_all = (a.M + 31) / 32;
    const long t_lo = second ? a.split : 0, t_n = second ? ntiles_all - a.split : a.split;
    const int nb = second ? (int)gridDim.x - a.blocks0 : a.blocks0, bi = second ? (int)blockIdx.x - a.blocks0 : (int)blockIdx.x;
    float* const c_alpha = smem;
    float* const c_rgb = smem + NT * 32;
    float* const ring0 = smem + kHeadFloats;
    for (int i = tid; i < NT * 32; i += 256) c_alpha[i] = P[L.alpha_off + i];
    for (int i = tid; i < 3 * OTV * 32; i += 256) c_rgb[i] = P[L.rgb_off + i];
    __syncthreads();

    WRing<NT, NF_LDS_BWD_SP1, true> st;
    st.gsrc = PT + lane * 4; st.ring = ring0; st.rl = ring0 + lane * 4;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)PT, 0, (int)(a.layT.total * 4), 0x00020000);
    st.voff = lane * 16;
    st.total = (int)(a.layT.total / kPiece); st.src = 0; st.slot = 0; st.rd = 0; st.wave = wave;
    st.start();

    f32x16 X[NT], Y[NT];                                                    // two gradient arrays swap roles from layer to layer
    const int nrounds = (int)((t_n + (long)nb * 4 - 1) / ((long)nb * 4));
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        // every wave walks the whole stream every round; one without a tile of its own recomputes the range's last tile and
        // writes the same bytes again (no branch in the pinned schedule)
        const long own = ((long)rnd * nb + bi) * 4 + wave;
        const long tile = t_lo + (own < t_n ? own : t_n - 1);
        int jj = j, hh = h, l2 = lane;
        asm volatile("" : "+v"(jj), "+v"(hh), "+v"(l2));                    // lane values recomputed per tile (not hoisted + spilled)
        const long sraw = tile * 32 + jj;
        const float* __restrict__ A = a.acts + (size_t)tile * TL.a_slots * 1024;
        float* __restrict__ Z = a.dz + (size_t)tile * TL.z_slots * 1024;
        const float* const Amask = A + TL.a_MASK * 1024;
        const unsigned voff = (unsigned)acc_lane_off(l2) * 4u;
        float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sraw < a.M) dr = reinterpret_cast<const float4*>(a.d_raw)[sraw];   // padded samples carry zero gradient
        TileMask<NT> mk = load_mask<NT>(Amask, L.D - 1, l2);                // ReLU bits of h_D: the first pts layer of the chain
        // ---- ZR: d_raw as a tile (channels 0..3 live in half 0, registers 0..3)
        {
            f32x16 zr[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[0][r] = 0.f;
            if (hh == 0) { zr[0][0] = dr.x; zr[0][1] = dr.y; zr[0][2] = dr.z; zr[0][3] = dr.w; }
            store_tiles<1>(Z + TL.z_ZR * 1024, zr, l2);
        }
        // ---- rgb_linear backward: dZ_v = (W_rgb^T d_rgb) * [hv > 0]   (VALU; operands of the views part)
        f32x16 dzv[OTV];
        {
            const TileMask<OTV> mhv = load_mask<OTV>(Amask, L.D, l2);
#pragma unroll
            for (int t = 0; t < OTV; ++t) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 w0 = lds_read4(c_rgb + ((0 * OTV + t) * 2 + hh) * 16 + 4 * r4);
                    const f32x4 w1 = lds_read4(c_rgb + ((1 * OTV + t) * 2 + hh) * 16 + 4 * r4);
                    const f32x4 w2 = lds_read4(c_rgb + ((2 * OTV + t) * 2 + hh) * 16 + 4 * r4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = w0[e] * dr.x + w1[e] * dr.y + w2[e] * dr.z;
                        dzv[t][4 * r4 + e] = mask_apply<OTV>(mhv, t, 4 * r4 + e, g);
                    }
                }
            }
            store_tiles<OTV>(Z + TL.z_ZV * 1024, dzv, l2);
        }
        // ---- views_linears[0] backward (feature columns): X = d_feature = Wv[:, :W]^T dZ_v. Meanwhile Y (dead) receives
        // w_alpha * d_sigma, the alpha head's share of d_h_D, tile by tile.
        lds_part<NT, NT, OTV * 4, 0, true>(st, X,
            [&](int q) {
                if (q < NT) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 w = lds_read4(c_alpha + (q * 2 + hh) * 16 + 4 * r4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) Y[q][4 * r4 + e] = w[e] * dr.w;
                    }
                }
            },
            [&](int q, float (&b)[4]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = dzv[q >> 2][4 * (q & 3) + e];
            });
        // ---- feature_linear backward: Y += Wf^T d_feature; dZ_F = d_feature leaves as it is consumed (no activation)
        {
            const float* const ZF = Z + TL.z_ZF * 1024;
            lds_part<NT, NT, 4 * NT, 0>(st, Y, [](int) {},
                [&](int q, float (&b)[4]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = X[q >> 2][4 * (q & 3) + e];
                },
                [&](int q, int e, const float (&b)[4]) { st_acc_reg(ZF + (q >> 2) * 1024, voff, 4 * (q & 3) + e, b[e]); });
        }
        // ---- pts_linears[D-1 .. 1]: out = W_i^T (in * [h_{i+1} > 0]); the masked operand is dZ_i and is stored from the VGPRs
        // it is consumed from. The NEXT layer's bit mask is requested at the head of the layer before.
        auto layer = [&](f32x16 (&in)[NT], f32x16 (&out)[NT], int i) {
            const float* const Zi = Z + (TL.z_Z0 + i * NT) * 1024;
            const TileMask<NT> m = mk;
            lds_part<NT, NT, 4 * NT, 0, true>(st, out,
                [&](int q) { if (q == 0) mk = load_mask<NT>(Amask, i - 1, l2); },
                [&](int q, float (&b)[4]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = mask_apply<NT>(m, q >> 2, 4 * (q & 3) + e, in[q >> 2][4 * (q & 3) + e]);
                },
                [&](int q, int e, const float (&b)[4]) { st_acc_reg(Zi + (q >> 2) * 1024, voff, 4 * (q & 3) + e, b[e]); });
        };
#pragma unroll 1
        for (int i = L.D - 1; i >= 2; i -= 2) {                             // D is even (host check): pairs, then layer 1
            layer(Y, X, i);
            layer(X, Y, i - 1);
        }
        layer(Y, X, 1);
        // ---- dZ_0 = d_h_1 * [h_1 > 0]: stored only (the chain stops here: no gradient w.r.t. the inputs, RN:394)
        {
            const float* const Z0 = Z + TL.z_Z0 * 1024;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = X[t][r];
                    asm("" : "+v"(v));
                    st_acc_reg(Z0 + t * 1024, voff, r, mask_apply<NT>(mk, t, r, v));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    lds_wait_vmcnt<0>();       // (the ring's last DMAs have landed before the LDS is released)
}

int launch_bwd_data_lds(const BwdArgs& a, int W, int cus, hipStream_t s) {
    using C = LdsCfg<8>;
    const long ntiles = (a.M + 31) / 32, t0 = a.split, t1 = ntiles - a.split;
    if (W != 256 || (a.lay.D & 1) || a.lay.D < 2 || a.lay.D > C::kMaxDepth || (a.layT.total / kPiece) % C::GP != 0 ||
        a.layT.total / kPiece < (unsigned)C::GP || ntiles >= (1L << 31)) {
        set_error("nerf_mlp_bwd_data_lds_kernel: shape not covered");
        return NERFAIL_EINVAL;
    }
    // workgroups per network in proportion to its tiles (at least one where there are tiles, whole 4-tile rounds)
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;
    long b0 = t1 == 0 ? blocks : (t0 == 0 ? 0 : (blocks * t0 + ntiles / 2) / ntiles);
    if (t0 > 0 && b0 < 1) b0 = 1;
    if (t1 > 0 && b0 > blocks - 1) b0 = blocks - 1;
    if (t0 > 0 && t1 > 0 && blocks < 2) { blocks = 2; b0 = 1; }
    if (b0 > (t0 + 3) / 4) b0 = (t0 + 3) / 4;                                // never more workgroups than 4-tile rounds
    long b1 = blocks - b0;
    if (b1 > (t1 + 3) / 4) b1 = (t1 + 3) / 4;
    BwdArgs k = a;
    k.blocks0 = (int)b0;
    nerf_mlp_bwd_data_lds_kernel<8><<<dim3((unsigned)(b0 + b1)), dim3(256), 0, s>>>(k);
    NF_LAUNCHED("nerf_mlp_bwd_data_lds_kernel");
    return NERFAIL_OK;
}

template <int NT>
static int launch_lds(const MlpArgs& a, unsigned blocks, hipStream_t s) {
    using C = LdsCfg<NT>;
    if ((int)a.lay.stream_pad != C::kStreamPad || a.lay.w_total / kPiece < (unsigned)C::GP || (a.lay.w_total / kPiece) % C::GP != 0 ||
        a.lay.total - a.lay.b_off[0] > (unsigned)C::kConstMax) {
        set_error("nerf_mlp_fwd_lds_kernel: packed layout does not match the ring geometry");
        return NERFAIL_EINVAL;
    }
    if ((a.lay.D & 1) || a.lay.D > C::kMaxDepth) { set_error("nerf_mlp_fwd_lds_kernel: depth not covered"); return NERFAIL_EINVAL; }
    const int skip_layer = a.lay.skip >= 0 ? a.lay.skip + 1 : -1;                         // layer that takes the extra part
    const bool train = a.acts != nullptr;
    if (skip_layer < 0) {
        if (train) nerf_mlp_fwd_lds_kernel<NT, 0, true><<<dim3(blocks), dim3(256), 0, s>>>(a);
        else nerf_mlp_fwd_lds_kernel<NT, 0, false><<<dim3(blocks), dim3(256), 0, s>>>(a);
    } else if (skip_layer & 1) {
        if (train) nerf_mlp_fwd_lds_kernel<NT, 1, true><<<dim3(blocks), dim3(256), 0, s>>>(a);
        else nerf_mlp_fwd_lds_kernel<NT, 1, false><<<dim3(blocks), dim3(256), 0, s>>>(a);
    } else {
        if (train) nerf_mlp_fwd_lds_kernel<NT, 2, true><<<dim3(blocks), dim3(256), 0, s>>>(a);
        else nerf_mlp_fwd_lds_kernel<NT, 2, false><<<dim3(blocks), dim3(256), 0, s>>>(a);
    }
    NF_LAUNCHED("nerf_mlp_fwd_lds_kernel");
    return NERFAIL_OK;
}

int launch_mlp_lds(const MlpArgs& a, int W, hipStream_t s) {
    if (a.M >= (1L << 36)) { set_error("nerfail_mlp_fwd: M must be below 2^36 samples per call"); return NERFAIL_EINVAL; }
    const long ntiles = (a.M + 31) / 32;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    long blocks = (ntiles + 3) / 4;
    if (blocks > cus) blocks = cus;      // persistent: one 4-wave workgroup per CU, one wave per SIMD
    switch (W) {
        case 256: return launch_lds<8>(a, (unsigned)blocks, s);
        case 128: return launch_lds<4>(a, (unsigned)blocks, s);
        case 64: return launch_lds<2>(a, (unsigned)blocks, s);
        default: set_error("nerfail_mlp_fwd: unsupported W"); return NERFAIL_EINVAL;
    }
}

}  // namespace nerfail
