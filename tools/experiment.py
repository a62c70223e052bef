#!/usr/bin/env python3
"""Timing experiments WITHOUT experiment code in the product sources: builds libnerfail_hip_exp_<name>.so from a PATCHED
COPY of one csrc file (text substitutions listed below, applied in a temp dir; every pattern must match exactly once) plus
the product objects. Run on the GPU with  tools/microbench_mlp.py --lib nerfail_amd/lib/libnerfail_hip_exp_<name>.so.
Most variants compute WRONG results on purpose (they remove work to price it) - timing only. The variant libraries are
git-ignored; delete them afterwards (they travel with every gpurun snapshot).

    python tools/experiment.py <name> [<name> ...]      |  python tools/experiment.py --list
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfail_amd import build as B  # noqa: E402

# shader-clock stamp at the start of EVERY step, kept in the 64 lanes of one VGPR (v_writelane-like select: no memory traffic);
# dumped at kernel end: raw[(block*4 + wave)*66 + lane] = stamp of step (idx - 64 + lane), [64] = idx
LDS_STEPS = [
    ('    f32x4 fr[C::HS];         // fragments of the NEXT step',
     '    unsigned st_v = 0; int st_i = 0;\n    f32x4 fr[C::HS];         // fragments of the NEXT step'),
    ('        pre();\n        int k = 0;\n',
     '        pre();\n        { const unsigned st_t = (unsigned)clock64(); st_v = ((int)(threadIdx.x & 63) == (st_i & 63) && st_i >= NF_ST_LO && st_i < NF_ST_LO + 64) ? st_t : st_v; ++st_i; }\n        int k = 0;\n'),
    ('    lds_wait_vmcnt<0>();       // no LDS-DMA may be in flight',
     '    { unsigned* o_ = reinterpret_cast<unsigned*>(a.raw) + (blockIdx.x * 4 + wave) * 66; o_[lane] = st.st_v; if (lane == 0) o_[64] = st.st_i; }\n'
     '    lds_wait_vmcnt<0>();       // no LDS-DMA may be in flight')]

# name -> (source file, [(old, new), ...], extra compiler flags)
EXPERIMENTS = {
    # round 4, K11: the per-wave LDS slice doubled = half the waves per CU (how much does the segmented reduce depend on occupancy?)
    'seg_lds2': ('gauss_csr.hip', [], ['-DNF_SEG_LDS_MULT=2']),
    # (the round-2/3 ablations of the old entry-strided reduce - seg_noscan / _nogather / _noemit / _u16 / _u4 - went with that kernel)
    'k10_early_index': ('gauss.hip', [], ['-DNF_K10_EARLY_INDEX=1']),   # (until round 6: 'k10_late_index' with =0, the product default)
    'lds_base': ('mlp_lds.hip', [], []),
    'lds_gpm2': ('mlp_lds.hip', [], ['-DNF_LDS_GPM=2']),
    'lds_noenc': ('mlp_lds.hip', [('        encode_sample(a, s, hh, emb, demb);\n',
                                   '        for (int i_ = 0; i_ < 4 * kEmbQuads; ++i_) emb[i_] = 0.25f * (float)(lane & 3) + (float)s * 1e-9f;\n'
                                   '        for (int i_ = 0; i_ < 4 * kDirQuads; ++i_) demb[i_] = 0.5f;\n')], []),
    'lds_nodma': ('mlp_lds.hip', [('            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(ring + (slot * C::GP + first) * kPiece), 16, voff,\n'
                                   '                                                     (src + first) * (kPiece * 4), (I % 4) * kPiece * 4, 0);\n',
                                   '            asm volatile("" :: "s"(first));\n')], []),
    'lds_dma_global': ('mlp_lds.hip', [], ['-DNF_LDS_DMA_BUF=0']),
    'lds_dma_burst': ('mlp_lds.hip', [], ['-DNF_LDS_DMA_SPREAD=0']),
    'lds_dma_global_burst': ('mlp_lds.hip', [], ['-DNF_LDS_DMA_BUF=0', '-DNF_LDS_DMA_SPREAD=0']),
    'lds_nobarrier': ('mlp_lds.hip', [('        __builtin_amdgcn_s_barrier();\n        asm volatile("" ::: "memory");\n',
                                       '        asm volatile("" ::: "memory");\n')], []),
    # shader-clock stamps of one tile's phases -> raw[(block*4 + wave)] as 4 uint32 deltas (timing only, outputs destroyed)
    'lds_clock': ('mlp_lds.hip', [
        ('        const long s = sraw < a.M ? sraw : a.M - 1;\n',
         '        const long s = sraw < a.M ? sraw : a.M - 1;\n        const unsigned long long c0_ = clock64();\n'),
        ('        // layer 0: 63 -> W into P (its bias is already there); Q (dead) receives the bias of layer 1 meanwhile\n',
         '        const unsigned long long c1_ = clock64();\n'),
        ('        // pts_linears[l] (l < D) / feature_linear (l == D): out += W_l relu(in).',
         '        const unsigned long long c2_ = clock64();\n        // pts_linears[l] (l < D) / feature_linear (l == D): out += W_l relu(in).'),
        ('        // views_linears[0]: cat([feature, embedded dirs]) -> W/2 into Q\'s first tiles',
         '        const unsigned long long c3_ = clock64();\n        // views_linears[0]: cat([feature, embedded dirs]) -> W/2 into Q\'s first tiles'),
        ('        if (h == 0 && sout < a.M && tile_own < ntiles)\n'
         '            reinterpret_cast<float4*>(a.raw)[sout] = make_float4(rgb[0], rgb[1], rgb[2], alpha);\n',
         '        const unsigned long long c4_ = clock64();\n'
         '        if (rnd == nrounds - 2 && lane == 0) {\n'
         '            unsigned* o_ = reinterpret_cast<unsigned*>(a.raw) + (blockIdx.x * 4 + wave) * 8;\n'
         '            o_[0] = (unsigned)(c1_ - c0_); o_[1] = (unsigned)(c2_ - c1_); o_[2] = (unsigned)(c3_ - c2_); o_[3] = (unsigned)(c4_ - c3_);\n'
         '            o_[4] = __float_as_uint(rgb[0] + rgb[1] + rgb[2] + alpha); o_[5] = (unsigned)wall_clock64();\n'
         '        }\n')], []),
    # wall-clock (100 MHz) stamps of a launch's phases per wave -> raw as 32 uint64 per wave: [0] kernel entry, [1] constants in LDS,
    # [2] ring started, [3 + r] end of round r's tile (r < 12), [15] kernel end; [16 + i]: the shader clock (s_memtime) at the same
    # points i = 2, 3 + r, 15 (tools/lds_timeline.py; outputs destroyed)
    'lds_timeline': ('mlp_lds.hip', [
        ('    const MlpLayout& L = a.lay;\n#if NF_LDS_RING_FIRST\n',
         '    const MlpLayout& L = a.lay;\n    const unsigned long long tl0_ = wall_clock64();\n#if NF_LDS_RING_FIRST\n'),
        ('    __syncthreads();\n    const float* const c_alpha = cst + (L.alpha_off - L.b_off[0]);\n',
         '    __syncthreads();\n    const unsigned long long tl1_ = wall_clock64();\n    const float* const c_alpha = cst + (L.alpha_off - L.b_off[0]);\n'),
        ('    st.start();\n\n    // Two activation arrays swap roles',
         '    st.start();\n    const unsigned long long tl2_ = wall_clock64(), tc2_ = clock64();\n\n    // Two activation arrays swap roles'),
        ('        if (h == 0 && sout < a.M && tile_own < ntiles)\n'
         '            reinterpret_cast<float4*>(a.raw)[sout] = make_float4(rgb[0], rgb[1], rgb[2], alpha);\n',
         '        if (lane == 0 && rnd < 12 && sout < a.M + 64) {\n'
         '            unsigned long long* o_ = reinterpret_cast<unsigned long long*>(a.raw) + (blockIdx.x * 4 + wave) * 32;\n'
         '            o_[3 + rnd] = wall_clock64() + (rgb[0] + rgb[1] + rgb[2] + alpha == 123.25f ? 1 : 0);\n'
         '            o_[16 + 3 + rnd] = clock64();\n        }\n'),
        ('    lds_wait_vmcnt<0>();       // no LDS-DMA may be in flight',
         '    lds_wait_vmcnt<0>();       // no LDS-DMA may be in flight\n'
         '    if (lane == 0) { unsigned long long* o_ = reinterpret_cast<unsigned long long*>(a.raw) + (blockIdx.x * 4 + wave) * 32; o_[0] = tl0_; o_[1] = tl1_; o_[2] = tl2_; o_[15] = wall_clock64(); o_[16 + 2] = tc2_; o_[16 + 15] = clock64(); }\n    //')], []),
    # round 6, pricing of VERDICT r5 item 6 (the per-ray-constant view channels folded into the views layer's bias): the 64 MFMAs of
    # the direction part become ring skips, and a stand-in for the fold's own work is added per tile - all 12 direction bands
    # evaluated as (sin, cos) pairs, a 2-channel x 27 dot product per lane with its weights read from LDS. The lane's own 12 direction band values are still computed (the real fold would not): the build UNDERSTATES the gain by
    # ~1 k of a tile's 633 k cycles. Wrong results on purpose - timing only.
    'lds_fold_price': ('mlp_lds.hip', [
        ('        lds_part<NT, OTV, kDirQuads, C::kStreamPad>(st, Q, [](int) {}, [&](int q, float (&b)[4]) { b_park(kEmbQuads + q, b); });\n',
         '        lds_part<NT, OTV, 0, kDirQuads * OTV + C::kStreamPad>(st, Q, [](int) {}, [&](int q, float (&b)[4]) { b_park(kEmbQuads + q, b); });\n'),
        ('        encode_sample(a, s, hh, emb, demb);\n',
         '        encode_sample(a, s, hh, emb, demb);\n'
         '        if (a.rays != nullptr) {\n'
         '            const float* ray_ = a.rays + NERFAIL_RAY_FLOATS * (s / a.spr);\n'
         '            float gm_[27];\n'
         '#pragma unroll\n'
         '            for (int d = 0; d < 3; ++d) {\n'
         '                const SinCosBands sc_(ray_[8 + d]);\n'
         '                gm_[d] = ray_[8 + d];\n'
         '#pragma unroll\n'
         '                for (int f = 0; f < 4; ++f) sc_.band(f, gm_[3 + 6 * f + d], gm_[6 + 6 * f + d]);\n'
         '            }\n'
         '            float c0_ = cst[lane], c1_ = cst[64 + lane];\n'
         '#pragma unroll\n'
         '            for (int jx = 0; jx < 27; ++jx) {\n'
         '                const float2 w_ = *reinterpret_cast<const float2*>(cst + 256 + jx * 128 + 2 * lane);\n'
         '                c0_ = fmaf(w_.x, gm_[jx], c0_); c1_ = fmaf(w_.y, gm_[jx], c1_);\n'
         '            }\n'
         '            emb[31] += (c0_ + c1_) * 1e-30f;      // (kept alive: the stand-in must not be optimised away)\n'
         '        }\n')], []),
    'lds_nobias': ('mlp_lds.hip', [('            auto hk = [&](int q) { if ((q & 3) == 0 && q > 0) bias_tile(in, l + 1, (q >> 2) - 1, !decltype(out_is_p)::value); };\n', '            auto hk = [&](int q) {};\n'),
                                   ('            bias_tile(in, l + 1, NT - 1, !decltype(out_is_p)::value);\n', '')], []),
    'lds_norelu': ('mlp_lds.hip', [('b[e] = relu_bits(in[q >> 2][4 * (q & 3) + e]);', 'b[e] = in[q >> 2][4 * (q & 3) + e];')], []),
    'lds_noread': ('mlp_lds.hip', [('        for (int t = 0; t < C::HS; ++t) fr[t] = lds_read4(rl + (rd + t) * kPiece);\n    }\n    __device__ __forceinline__ void start()',
                                    '        for (int t = 0; t < C::HS; ++t) asm volatile("" : "+v"(fr[t]));\n    }\n    __device__ __forceinline__ void start()')], []),
    # shader-clock stamp at the end of EVERY step, kept in the 64 lanes of one VGPR (v_writelane: no memory traffic);
    # dumped at kernel end: raw[(block*4 + wave)*66 + lane] = stamp of step (idx - 64 + lane), [64] = idx
    'lds_steps': ('mlp_lds.hip', LDS_STEPS, ['-DNF_ST_LO=27776']),          # the last 64 steps of the last of 48 tiles (8192 x 192 samples)
    'lds_steps_l2': ('mlp_lds.hip', LDS_STEPS, ['-DNF_ST_LO=27340']),       # tile 47, steps 80..143 = pts_linears[2]
    'lds_spread_steps_l2': ('mlp_lds.hip', LDS_STEPS, ['-DNF_LDS_SPREAD=1', '-DNF_ST_LO=27340']),
    'lds_agpr_p': ('mlp_lds.hip', [], ['-DNF_LDS_VGPR_P=0']),
    'lds_ring_last': ('mlp_lds.hip', [], ['-DNF_LDS_RING_FIRST=0']),
    'bwd_sp0': ('mlp_lds.hip', [], ['-DNF_LDS_BWD_SP1=0']),
    'lds_train_newdma': ('mlp_lds.hip', [], ['-DNF_LDS_TRAIN_NEWDMA=1']),
    'lds_midsplit': ('mlp_lds.hip', [], ['-DNF_LDS_MID_SPLIT=1']),
    # round 6: the training forward on the one-piece-of-side-work-per-shadow step form; activation stores behind MFMAs K0 .. K0 + 3
    'lds_train_sp1': ('mlp_lds.hip', [], ['-DNF_LDS_TRAIN_SP1=1']),
    'lds_train_sp1_k8': ('mlp_lds.hip', [], ['-DNF_LDS_TRAIN_SP1=1', '-DNF_LDS_TRAIN_K0=8']),
    'lds_train_sp1_k12': ('mlp_lds.hip', [], ['-DNF_LDS_TRAIN_SP1=1', '-DNF_LDS_TRAIN_K0=12']),
    'lds_train_k8': ('mlp_lds.hip', [], ['-DNF_LDS_TRAIN_K0=8']),
    # round 6, the ReLU bit masks of the training forward: the product appends a value's bit with v_cmp_lt_i32 + v_addc_co_u32
    # (bits arrive reversed, one v_bfrev per tile). lds_mask_med3 = the idiom of rounds 3-5 (v_med3_i32 + v_lshl_or_b32): the A/B
    # partner of profiles/r06_train_mask_ab.log (6.836 against 6.782 ms at 786 432 samples; same bits)
    'lds_mask_med3': ('mlp_lds.hip', [
        ('                    asm volatile("v_cmp_lt_i32 vcc, 0, %1\\n\\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mk16) : "v"(b[e]) : "vcc");\n'
         '                    if (r == 15) {\n                        st_mask16(Ml, vlane16, q >> 2, __builtin_bitreverse32(mk16) >> 16);\n',
         '                    mk16 |= relu_bit(b[e]) << r;\n'
         '                    if (r == 15) {\n                        st_mask16(Ml, vlane16, q >> 2, mk16);\n')], []),
    # round 6: the backward-data ring kernel's lazy masking with the bits shifted out through VCC (v_add_co + v_cndmask) instead of
    # v_bfe_i32 + v_and_b32
    'bwd_mask_carry': ('mlp_lds.hip', [
        ('            const TileMask<NT> m = mk;\n',
         '            const TileMask<NT> m = mk;\n            unsigned mcur_ = 0u;\n'),
        ('#pragma unroll\n                    for (int e = 0; e < 4; ++e) b[e] = mask_apply<NT>(m, q >> 2, 4 * (q & 3) + e, in[q >> 2][4 * (q & 3) + e]);\n',
         '                    if ((q & 7) == 0) mcur_ = __builtin_bitreverse32(m.w[q >> 3]);\n'
         '#pragma unroll\n'
         '                    for (int e = 0; e < 4; ++e) {\n'
         '                        float x_ = in[q >> 2][4 * (q & 3) + e];\n'
         '                        asm volatile("v_add_co_u32 %0, vcc, %0, %0\\n\\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(mcur_), "+v"(x_) : : "vcc");\n'
         '                        b[e] = x_;\n'
         '                    }\n')], []),
    'lds_spread0': ('mlp_lds.hip', [], ['-DNF_LDS_SPREAD=0']),
    'lds_spread2': ('mlp_lds.hip', [], ['-DNF_LDS_SPREAD=2']),
    'lds_spread_steps': ('mlp_lds.hip', LDS_STEPS, ['-DNF_LDS_SPREAD=1', '-DNF_ST_LO=27776']),
    # clock probes of the register-streamed forward kernel and of the LDS-staged weight-gradient kernel (tools/fwd_clock.py,
    # tools/dw_balance.py read the stamps; outputs destroyed)
    'fwd_clock': ('mlp.hip', [
        ('    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);\n\n    for (long rnd = 0; rnd < nrounds; ++rnd) {\n        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;\n        if (tile >= ntiles) break;     // wave-uniform; waves are fully independent (no barriers)\n',
         '    const long nrounds = (ntiles + (long)gridDim.x * 4 - 1) / ((long)gridDim.x * 4);\n    const unsigned long long nf_c0 = clock64(), nf_w0 = wall_clock64();\n\n    for (long rnd = 0; rnd < nrounds; ++rnd) {\n        const long tile = (rnd * gridDim.x + blockIdx.x) * 4 + wave;\n        if (tile >= ntiles) break;     // wave-uniform; waves are fully independent (no barriers)\n'),
        ('            reinterpret_cast<float4*>(a.raw)[sraw] = make_float4(rgb[0], rgb[1], rgb[2], alpha);\n    }\n}\n',
         '            reinterpret_cast<float4*>(a.raw)[sraw] = make_float4(rgb[0], rgb[1], rgb[2], alpha);\n    }\n'
         '    if (threadIdx.x == 0) {\n'
         '        reinterpret_cast<unsigned long long*>(a.raw)[2 * blockIdx.x] = clock64() - nf_c0;\n'
         '        reinterpret_cast<unsigned long long*>(a.raw)[2 * blockIdx.x + 1] = wall_clock64() - nf_w0;\n    }\n}\n')], []),
    'dw_clock': ('mlp_dw.hip', [
        ('    __shared__ __attribute__((aligned(16))) float smem[kDwStages * kDwStageFloats];\n    const int lane = threadIdx.x & 63;\n',
         '    __shared__ __attribute__((aligned(16))) float smem[kDwStages * kDwStageFloats];\n    const unsigned long long nf_w0 = wall_clock64();\n    const int lane = threadIdx.x & 63;\n'),
        ('            default: dw_group_run<BF16, 1, 2, 1, 8>(a, grp, smem, lane, wave, t_begin, t_end, out); break;\n        }\n    }\n',
         '            default: dw_group_run<BF16, 1, 2, 1, 8>(a, grp, smem, lane, wave, t_begin, t_end, out); break;\n        }\n    }\n'
         '    if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.dz))[blockIdx.x] = wall_clock64() - nf_w0;\n')], []),
    # pricing of the training stores of the LDS-ring forward (mlp_lds.hip, TRAIN): default cache policy instead of nt / no
    # activation stores at all (backward then reads garbage: timing only) / no ReLU bit masks
    'lds_train_nont': ('mlp_lds.hip', [('    asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(v), "s"(sbase), "i"(acc_reg_off(r) * 4) : "memory");\n',
                                       '    asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(sbase), "i"(acc_reg_off(r) * 4) : "memory");\n')], []),
    'lds_train_nostore': ('mlp_lds.hip', [('    asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(v), "s"(sbase), "i"(acc_reg_off(r) * 4) : "memory");\n',
                                          '    asm volatile("" ::"v"(voff), "v"(v), "s"(sbase) : "memory");\n')], []),
    'lds_train_nomask': ('mlp_lds.hip', [('    asm volatile("global_store_short %0, %1, %2 offset:%3" ::"v"(vlane16), "v"(bits), "s"(sentry), "i"(2 * t) : "memory");\n',
                                         '    asm volatile("" ::"v"(vlane16), "v"(bits), "s"(sentry) : "memory");\n'),
                                        ('                    asm volatile("v_cmp_lt_i32 vcc, 0, %1\\n\\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mk16) : "v"(b[e]) : "vcc");\n', '')], []),
    # pricing of the weight-gradient step loop (mlp_dw.hip): no LDS-DMA / no per-step barrier / no LDS operand reads / no row sums
    'dw_nodma': ('mlp_dw.hip', [('        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + voff[i]),\n                                         (lds_void_t*)(smem + rs * kDwStageFloats + (wave + 4 * i) * 256), 16, 0, 0);\n',
                                 '        asm volatile("" :: "v"(base + voff[i]), "s"(rs));\n')], []),
    'dw_dma_buf': ('mlp_dw.hip', [], ['-DNF_DW_DMA_BUF=1']),
    'dw_nobarrier': ('mlp_dw.hip', [('                dw_wait_vmcnt<(NS - 3) * G>();                          // this wave\'s pieces of stage s+1 have landed\n                __builtin_amdgcn_s_barrier();',
                                     '                dw_wait_vmcnt<(NS - 3) * G>();                          // this wave\'s pieces of stage s+1 have landed\n')], []),
    'dw_nofetch': ('mlp_dw.hip', [('        R[p][t][hf] = *reinterpret_cast<const f32x4*>(sbase + off);\n', '        if (off == -12345) R[p][t][hf] = *reinterpret_cast<const f32x4*>(sbase + off);\n')], []),
    'dw_norowsum': ('mlp_dw.hip', [('                rowsum[m] = rowsum[m] + (R[P][m][0] + R[P][m][1]);\n', '')], []),
    # K11 segmented reduce pricing: no wave scan (wrong sums) / no gathers (index stream only)
    # K8 grid resolution: cells per axis = scale * cbrt(n) (1.5: 7.6 ms, 2.0: 7.7 ms per rendered-view map against 8.3 ms; shell
    # points 2.0 / 1.0 ms against 1.3 ms - no clear winner, the search is latency bound)
    # K8 far search: wave-level counters -> stats[2..9] (tools/debug/knn_profile.py)
    'knn_profile': ('knn_grid.hip', [
        ('    bool valid;\n    long qi;\n', '    bool valid;\n    long qi;\n    const unsigned long long pf_start = wall_clock64();\n'),
        ('            if (pending) done = shell_walk();\n', '            const unsigned long long psh = wall_clock64();\n            if (pending) done = shell_walk();\n            pf_shell += wall_clock64() - psh;\n'),
        ('    auto wave_search = [&](const bool far) {\n', '    unsigned long long pf_far = 0;\n    auto wave_search = [&](const bool far) {\n        pf_far |= __ballot(far);\n'),
        ('    const int Gc = (G + kCoarse - 1) / kCoarse, Gs = (Gc + kSuper - 1) / kSuper;\n    // ---- scattered', '    unsigned long long pf_pts = 0, pf_flush = 0, pf_cells = 0, pf_coarse = 0, pf_super = 0, pf_scan_clk = 0, pf_load_clk = 0, pf_loads = 0, pf_slow = 0, pf_slow_clk = 0, pf_eval = 0, pf_shell = 0, pf_t0 = wall_clock64();\n    const int Gc = (G + kCoarse - 1) / kCoarse, Gs = (Gc + kSuper - 1) / kSuper;\n    // ---- scattered'),
        ('        auto scan_uniform = [&](int b, int e) {\n            for (int p0 = b; p0 < e; p0 += 64) {',
         '        auto scan_uniform = [&](int b, int e) {\n            pf_pts += (unsigned)(e - b); pf_cells += 1; const unsigned long long c0 = wall_clock64();\n            for (int p0 = b; p0 < e; p0 += 64) {'),
        ('                if (far) examined += (unsigned)__popcll(m);\n', '                if (far) examined += (unsigned)__popcll(m);\n                pf_eval += (unsigned)__popcll(m);\n'),
        ('            n_pend = 0;\n', '            n_pend = 0;\n            pf_flush += 1;\n'),
        ('                const float4 mine = sorted[min(p0 + lane, e - 1)];\n',
         '                const unsigned long long pl0 = wall_clock64();\n                float4 mine = sorted[min(p0 + lane, e - 1)];\n                asm volatile("" : "+v"(mine.x), "+v"(mine.w));\n                pf_load_clk += wall_clock64() - pl0; pf_loads += 1;\n'),
        ('                    if (__ballot(far & any) == 0ull) continue;\n',
         '                    if (__ballot(far & any) == 0ull) continue;\n                    pf_slow += 1; const unsigned long long ps0 = wall_clock64();\n'),
        ('                        park(far & (key < fk[7]), key);\n                    }\n',
         '                        park(far & (key < fk[7]), key);\n                    }\n                    pf_slow_clk += wall_clock64() - ps0;\n'),
        ('            if (__ballot(n_pend != 0) != 0ull) flush();\n', '            if (__ballot(n_pend != 0) != 0ull) flush();\n            pf_scan_clk += wall_clock64() - c0;\n'),
        ('            const int fx0 = X * kCoarse, fy0 = Y * kCoarse, fz0 = Z * kCoarse;\n', '            const int fx0 = X * kCoarse, fy0 = Y * kCoarse, fz0 = Z * kCoarse;\n            pf_coarse += 1;\n'),
        ('        auto visit_super = [&](int S) {                                       // lane = coarse cell of the block\n',
         '        auto visit_super = [&](int S) {                                       // lane = coarse cell of the block\n            pf_super += 1;\n'),
        ('    if (stats != nullptr) {             // [0] candidates examined',
         '    unsigned long long pf_ex = examined;\n    for (int o = 32; o > 0; o >>= 1) pf_ex += __shfl_xor(pf_ex, o, 64);\n    if (stats != nullptr && (threadIdx.x & 63) == 0) {\n'
         '        atomicAdd(stats + 2, pf_pts); atomicAdd(stats + 3, pf_flush); atomicAdd(stats + 4, pf_cells); atomicAdd(stats + 5, pf_coarse);\n'
         '        atomicAdd(stats + 6, pf_super); atomicAdd(stats + 7, pf_scan_clk); atomicAdd(stats + 8, (unsigned long long)(wall_clock64() - pf_t0));\n'
         '        atomicAdd(stats + 9, pf_far != 0ull ? 1ull : 0ull); atomicAdd(stats + 10, pf_load_clk); atomicAdd(stats + 11, pf_loads); atomicAdd(stats + 12, pf_slow); atomicAdd(stats + 13, pf_slow_clk); atomicAdd(stats + 14, pf_eval);\n        const long gw = blockIdx.x, nw = gridDim.x;\n        stats[16 + gw] = (unsigned long long)(wall_clock64() - pf_start); stats[16 + nw + gw] = pf_pts; stats[16 + 2 * nw + gw] = __popcll(pf_far); stats[16 + 3 * nw + gw] = pf_start; stats[16 + 6 * nw + gw] = pf_shell; stats[16 + 5 * nw + gw] = pf_ex; stats[16 + 7 * nw + gw] = pf_cells | (pf_coarse << 20) | (pf_super << 40); stats[16 + 8 * nw + gw] = pf_flush | (pf_scan_clk << 24); stats[16 + 4 * nw + gw] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);\n    }\n'
         '    if (stats != nullptr) {             // [0] candidates examined')], []),
    'knn_g075': ('knn_grid.hip', [('    int G = (int)lround(cbrt((double)n));', '    int G = (int)lround(0.75 * cbrt((double)n));')], []),
    'knn_g05': ('knn_grid.hip', [('    int G = (int)lround(cbrt((double)n));', '    int G = (int)lround(0.5 * cbrt((double)n));')], []),
    'knn_g15': ('knn_grid.hip', [('    int G = (int)lround(cbrt((double)n));', '    int G = (int)lround(1.5 * cbrt((double)n));')], []),
    'knn_g20': ('knn_grid.hip', [('    int G = (int)lround(cbrt((double)n));', '    int G = (int)lround(2.0 * cbrt((double)n));')], []),
    # K6 sample_fine pricing: no output stores / no merge search / no inverse-cdf search / no double-precision scan
    'sf_nostore': ('sampling.hip', [('    auto emit = [&](int rank, float z) {\n', '    auto emit = [&](int rank, float z) {\n        if (rank != -12345) return;\n')], []),
    'sf_nomerge': ('sampling.hip', [('                if (p <= n && (is_c ? (o < v) : (o <= v))) lo = p;\n            }\n', '                if (p == -5 && (is_c ? (o < v) : (o <= v))) lo = p;\n                break;\n            }\n')], []),
    'sf_noinvert': ('sampling.hip', [('        if (p <= n && (STRICT ? o < v : o <= v)) pos = p;\n    }\n', '        if (p <= n && (STRICT ? o < v : o <= v)) pos = p;\n        break;\n    }\n')], []),
    'sf_noscan': ('sampling.hip', [('        const double incl = wave_scan_add_f64((double)pdf, lane) + carry;\n', '        const double incl = (double)pdf * (double)(lane + 1) + carry;\n')], []),
}


def main():
    if '--list' in sys.argv:
        print('\n'.join(EXPERIMENTS))
        return
    B.build()
    for name in sys.argv[1:]:
        parts = [EXPERIMENTS[n] for n in name.split('+')]             # a+b: both patch sets on the same source
        assert len({p_[0] for p_ in parts}) == 1
        src, patches, flags = parts[0][0], sum((p_[1] for p_ in parts), []), sum((p_[2] for p_ in parts), [])
        text = open(os.path.join(B.CSRC, src)).read()
        for old, new in patches:
            assert text.count(old) == 1, 'experiment %s: pattern does not match exactly once:\n%s' % (name, old)
            text = text.replace(old, new)
        tmp = tempfile.mkdtemp(prefix='nf_exp_')
        try:
            for f in os.listdir(B.CSRC):
                if f.endswith('.h'):
                    shutil.copy(os.path.join(B.CSRC, f), tmp)
            open(os.path.join(tmp, src), 'w').write(text.replace('"../../include/', '"' + B.INCLUDE + '/'))
            for f in os.listdir(tmp):
                if f.endswith('.h'):
                    t = open(os.path.join(tmp, f)).read()
                    open(os.path.join(tmp, f), 'w').write(t.replace('"../../include/', '"' + B.INCLUDE + '/'))
            o = os.path.join(B.OBJDIR, 'exp_%s.o' % name)
            subprocess.check_call([B.HIPCC] + B.CFLAGS + B.FILE_FLAGS.get(src, []) + flags + ['-c', os.path.join(tmp, src), '-o', o])
        finally:
            shutil.rmtree(tmp)
        objs = [os.path.join(B.OBJDIR, f[:-4] + '.o') for f in B._sources() if f != src] + [o]
        lib = os.path.join(B.LIBDIR, 'libnerfail_hip_exp_%s.so' % name)
        subprocess.check_call([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs)
        print(lib, flush=True)


if __name__ == '__main__':
    main()
