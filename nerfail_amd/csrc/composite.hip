// K5 alpha compositing (+ K7 pts_max epilogue): raw2outputs, run_nerf.py:262-305, and the
// argmax-weight point of nerf_to_coord.py:418-423.
//
// One wavefront per ray. Lane l owns IPL consecutive samples (IPL = ceil(N/64): 1 for the 64 coarse
// samples, 3 for the 192 fine ones), so every global access is a contiguous run per lane and a
// contiguous span per wave. The exclusive transmittance cumprod is a lane-local product followed by a
// 64-lane scan; the ray sums are wave totals. No LDS - round 4: not the LDS crossbar either (the scans and totals run on
// DPP moves instead of __shfl_*: ~36 ds_bpermute trips per ray gone), and the rgb sigmoid 1 / (1 + exp(-x)) uses v_exp_f32 and
// v_rcp_f32 (1-2 ulp each: 1e-7 of a colour that is compared at 1e-4) instead of the correctly rounded expf and division,
// which were two thirds of the kernel's ~90 vector instructions per sample. alpha = 1 - exp(-sigma dist) keeps the accurate
// expf: it cancels for faint samples, where one ulp of exp is 1e-4 of the weight.
//
// HBM-bound: algorithmic bytes per ray = 24*N + 36 (+12*N+12 when pts_max is requested: it reads one
// point per ray, 12 B, the figure quoted in DESIGN.md uses the raw2outputs form).
#include "common.h"

#include <stdlib.h>

namespace nerfail {

// 1 / (1 + exp(-x)) on the hardware transcendentals: v_exp_f32 (2^x, 1 ulp) of -x * log2(e), v_rcp_f32 (1 ulp). exp(-x) -> inf
// for x << 0 gives rcp(inf) = 0, like the reference's sigmoid; NaN propagates. (__frcp_rn / __expf compile to the correctly
// rounded division and the full expf in this HIP build.)
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

// FULL: N == 64 * IPL (the shipped configurations: 64 coarse, 192 fine samples) - every lane owns IPL valid samples, so the
// per-sample range checks (one exec-mask branch each) are compiled out, a lane's depths and weights move as ONE load / store of
// IPL dwords, and the depth behind a lane's last sample comes from the next lane's register (DPP wave_shl:1) instead of a
// fourth load. Same operations in the same order: the outputs are bitwise those of the general form.
template <int IPL, bool FULL>
__global__ __launch_bounds__(256) void composite_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays,
    const float* __restrict__ noise, long n_rays, int N, int white_bkgd,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ weights, float* __restrict__ depth_map,
    const float* __restrict__ pts, float* __restrict__ pts_max) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;   // whole wave exits together

    const float* rr = rays + NERFAIL_RAY_FLOATS * ray;
    const float dx = rr[3], dy = rr[4], dz = rr[5];
    const float nrm = sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));

    const long base = ray * N;
    const int i0 = lane * IPL;

    float z[IPL + 1];
    float alpha[IPL], rgbr[IPL], rgbg[IPL], rgbb[IPL];
    if constexpr (FULL) {
        const float* zp = z_vals + base + i0;
#pragma unroll
        for (int k = 0; k < IPL; ++k) z[k] = zp[k];                  // IPL consecutive dwords: one load
        z[IPL] = dpp_move_f<0x130, 0xF>(0.0f, z[0]);                 // wave_shl:1 - the next lane's first depth (lane 63: unused)
    } else {
#pragma unroll
        for (int k = 0; k <= IPL; ++k) {
            const int i = i0 + k;
            z[k] = (i < N) ? z_vals[base + i] : 0.0f;
        }
    }
    float tprod = 1.0f;          // product of this lane's (1-alpha+1e-10)
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = i0 + k;
        alpha[k] = 0.0f; rgbr[k] = rgbg[k] = rgbb[k] = 0.0f;
        if (FULL || i < N) {
            const float4 rw = raw[base + i];
            float dist = (i < N - 1) ? __fsub_rn(z[k + 1], z[k]) : 1e10f;
            dist = __fmul_rn(dist, nrm);
            float sigma = rw.w;
            if (noise != nullptr) sigma = __fadd_rn(sigma, noise[base + i]);
            sigma = fmaxf(sigma, 0.0f);
            alpha[k] = __fsub_rn(1.0f, expf(-__fmul_rn(sigma, dist)));
            rgbr[k] = fast_sigmoid(rw.x);
            rgbg[k] = fast_sigmoid(rw.y);
            rgbb[k] = fast_sigmoid(rw.z);
            tprod = __fmul_rn(tprod, __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f));
        }
    }
    // exclusive scan of lane products -> transmittance in front of this lane's first sample
    const float incl = wave_scan_mul_dpp(tprod);
    float T = dpp_move_f<0x138, 0xF>(1.0f, incl);            // wave_shr:1 - lane l receives lane l - 1, lane 0 the identity

    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    float best_w = -1.0f;
    int best_i = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = i0 + k;
        if (FULL || i < N) {
            const float w = __fmul_rn(alpha[k], T);
            if (weights != nullptr) weights[base + i] = w;
            sr += w * rgbr[k]; sg += w * rgbg[k]; sb += w * rgbb[k];
            sd += w * z[k];
            sa += w;
            if (w > best_w) { best_w = w; best_i = i; }          // first maximum inside the lane
            T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f));
        }
    }
    sr = wave_total_dpp(sr); sg = wave_total_dpp(sg); sb = wave_total_dpp(sb); sd = wave_total_dpp(sd); sa = wave_total_dpp(sa);

    if (pts_max != nullptr) {   // torch.argmax returns the FIRST maximal index (NC:418)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ow = __shfl_xor(best_w, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (ow > best_w || (ow == best_w && oi < best_i)) { best_w = ow; best_i = oi; }
        }
        if (best_i >= N) best_i = 0;   // all-NaN weights: never index outside the ray
        if (lane < 3) {
            // the sample point of the largest weight: gathered from pts, or formed from the ray (o + d z, RN:399 rounding)
            // when the pipeline no longer materialises the points (round 3)
            pts_max[3 * ray + lane] = pts != nullptr ? pts[3 * (base + best_i) + lane]
                                                     : mul_add_rn(rays[NERFAIL_RAY_FLOATS * ray + 3 + lane], z_vals[base + best_i],
                                                                  rays[NERFAIL_RAY_FLOATS * ray + lane]);
        }
    }

    if (lane == 0) {
        const float ratio = __fdiv_rn(sd, sa);
        // torch.max(1e-10, ratio) propagates NaN (acc == 0 -> 0/0): RN:299
        const float disp = (ratio != ratio) ? ratio : __fdiv_rn(1.0f, fmaxf(1e-10f, ratio));
        if (white_bkgd) {
            const float bg = __fsub_rn(1.0f, sa);
            sr += bg; sg += bg; sb += bg;
        }
        rgb_map[3 * ray + 0] = sr; rgb_map[3 * ray + 1] = sg; rgb_map[3 * ray + 2] = sb;
        disp_map[ray] = disp;
        acc_map[ray] = sa;
        if (depth_map != nullptr) depth_map[ray] = sd;
    }
}

// Two rays per wavefront, COALESCED (round 4): lanes 0-31 own ray 2w, lanes 32-63 ray 2w + 1; lane l owns the samples
// l, l + 32, l + 64, ... (IPL = N / 32 of them), so every load and store instruction covers two contiguous 512-byte (raw) or
// 128-byte (z, weights) runs. The one-ray-per-wave form gives a lane IPL CONSECUTIVE samples: its loads stride 48 bytes from
// lane to lane, every instruction touches all of a ray's lines and the L1 has to keep them for the next two instructions -
// measured alone on 640 000 x 192 samples (tools/debug/composite_time.py): 0.62 ms; two rays per wave with 6 consecutive
// samples per lane (96-byte stride, half the per-ray vector instructions) 0.68 ms - slower: the bound is the access pattern,
// not the instruction count; this form 0.545 ms = 5.45 TB/s (N = 64: 0.197 ms, 5.1 TB/s). The transmittance is scanned block by block (32 samples of each ray per DPP scan, the running
// product carried from block to block through lanes 31 / 63); the per-ray totals, the argmax and the epilogue serve both rays
// per instruction. Same per-sample arithmetic; sums in another order than the one-ray form (last bits). N == 32 * IPL.
template <int IPL>
__global__ __launch_bounds__(256) void composite2_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays,
    const float* __restrict__ noise, long n_rays, int N, int white_bkgd,
    float* __restrict__ rgb_map, float* __restrict__ disp_map, float* __restrict__ acc_map,
    float* __restrict__ weights, float* __restrict__ depth_map,
    const float* __restrict__ pts, float* __restrict__ pts_max) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const bool upper = lane >= 32;
    const long ray_raw = 2 * ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + (upper ? 1 : 0);
    if (ray_raw - (upper ? 1 : 0) >= n_rays) return;                   // whole wave exits together
    const bool live = ray_raw < n_rays;                                // (an odd ray count: the last wave's second half idles)
    const long ray = live ? ray_raw : n_rays - 1;                      // loads stay in range, stores are masked

    const float* rr = rays + NERFAIL_RAY_FLOATS * ray;
    const float dx = rr[3], dy = rr[4], dz = rr[5];
    const float nrm = sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const long base = ray * N;

    float4 rw[IPL];
    float z[IPL], nz[IPL];
#pragma unroll
    for (int k = 0; k < IPL; ++k) {                                    // every load of the ray up front
        rw[k] = raw[base + 32 * k + l];
        z[k] = z_vals[base + 32 * k + l];
        nz[k] = (noise != nullptr) ? noise[base + 32 * k + l] : 0.0f;
    }
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    float best_w = -1.0f;
    int best_i = 0x7fffffff;
    float carry = 1.0f;                                                // transmittance in front of the block (per ray)
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = 32 * k + l;
        // the next sample's depth: the next lane's (wave_shl:1); for the block's last lane the first lane's depth of block k + 1
        float zn = dpp_move_f<0x130, 0xF>(0.0f, z[k]);
        if (k + 1 < IPL) {
            const float f0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[k + 1 < IPL ? k + 1 : k]), 0));
            const float f1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[k + 1 < IPL ? k + 1 : k]), 32));
            if (l == 31) zn = upper ? f1 : f0;
        }
        float dist = (i < N - 1) ? __fsub_rn(zn, z[k]) : 1e10f;
        dist = __fmul_rn(dist, nrm);
        float sigma = rw[k].w;
        if (noise != nullptr) sigma = __fadd_rn(sigma, nz[k]);
        sigma = fmaxf(sigma, 0.0f);
        const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sigma, dist)));
        const float p = __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f);
        // inclusive product scan over the block's 32 samples of each ray (row_shr 1 2 4 8, row_bcast:15 inside the halves)
        float incl = p;
        incl *= dpp_move_f<0x111, 0xF>(1.f, incl);
        incl *= dpp_move_f<0x112, 0xF>(1.f, incl);
        incl *= dpp_move_f<0x114, 0xF>(1.f, incl);
        incl *= dpp_move_f<0x118, 0xF>(1.f, incl);
        incl *= dpp_move_f<0x142, 0xA>(1.f, incl);
        float excl = dpp_move_f<0x138, 0xF>(1.0f, incl);               // wave_shr:1
        if (l == 0) excl = 1.0f;                                       // (lane 32 received lane 31: the other ray)
        const float T = __fmul_rn(carry, excl);
        const float w = __fmul_rn(alpha, T);
        if (weights != nullptr && live) weights[base + i] = w;
        sr += w * fast_sigmoid(rw[k].x); sg += w * fast_sigmoid(rw[k].y); sb += w * fast_sigmoid(rw[k].z);
        sd += w * z[k];
        sa += w;
        if (w > best_w) { best_w = w; best_i = i; }                    // a lane's samples ascend with k: first maximum
        const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 31));
        const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
        carry = __fmul_rn(carry, upper ? t1 : t0);
    }
    // ray totals: inclusive sum scan inside the halves; lanes 31 and 63 end up with their ray's sums
    auto half_scan = [](float v) {
        v += dpp_move_f<0x111, 0xF>(0.f, v);
        v += dpp_move_f<0x112, 0xF>(0.f, v);
        v += dpp_move_f<0x114, 0xF>(0.f, v);
        v += dpp_move_f<0x118, 0xF>(0.f, v);
        v += dpp_move_f<0x142, 0xA>(0.f, v);
        return v;
    };
    sr = half_scan(sr); sg = half_scan(sg); sb = half_scan(sb); sd = half_scan(sd); sa = half_scan(sa);

    if (pts_max != nullptr) {   // torch.argmax returns the FIRST maximal index (NC:418); xor offsets < 32 stay inside a half
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ow = __shfl_xor(best_w, o, 64);
            const int oi = __shfl_xor(best_i, o, 64);
            if (ow > best_w || (ow == best_w && oi < best_i)) { best_w = ow; best_i = oi; }
        }
        if (best_i >= N) best_i = 0;   // all-NaN weights: never index outside the ray
        if (l < 3 && live) {
            pts_max[3 * ray + l] = pts != nullptr ? pts[3 * (base + best_i) + l]
                                                  : mul_add_rn(rays[NERFAIL_RAY_FLOATS * ray + 3 + l], z_vals[base + best_i],
                                                               rays[NERFAIL_RAY_FLOATS * ray + l]);
        }
    }
    if (l == 31 && live) {
        const float ratio = __fdiv_rn(sd, sa);
        // torch.max(1e-10, ratio) propagates NaN (acc == 0 -> 0/0): RN:299
        const float disp = (ratio != ratio) ? ratio : __fdiv_rn(1.0f, fmaxf(1e-10f, ratio));
        if (white_bkgd) {
            const float bg = __fsub_rn(1.0f, sa);
            sr += bg; sg += bg; sb += bg;
        }
        rgb_map[3 * ray + 0] = sr; rgb_map[3 * ray + 1] = sg; rgb_map[3 * ray + 2] = sb;
        disp_map[ray] = disp;
        acc_map[ray] = sa;
        if (depth_map != nullptr) depth_map[ray] = sd;
    }
}

}  // namespace nerfail

using namespace nerfail;

// 0 = automatic (two rays per wave for multiples of 32 samples), 1 = the one-ray-per-wave form everywhere. Initialised ONCE
// from NERFAIL_COMPOSITE_KERNEL (getenv per call races with setenv from other threads); tests and A/B runs switch
// it through nerfail_composite_select.
static int composite_select_from_env() {
    const char* sel = getenv("NERFAIL_COMPOSITE_KERNEL");
    return (sel != nullptr && sel[0] == '1') ? 1 : 0;
}
static int g_composite_select = composite_select_from_env();

extern "C" int nerfail_composite_select(int which) {
    const int prev = g_composite_select;
    if (which == 0 || which == 1) g_composite_select = which;
    return prev;
}

extern "C" int nerfail_composite(const float* raw, const float* z_vals, const float* rays, const float* noise,
                                 int64_t n_rays, int n_samples, int white_bkgd, float* rgb_map, float* disp_map,
                                 float* acc_map, float* weights, float* depth_map, const float* pts, float* pts_max,
                                 void* stream) {
    NF_REQUIRE(n_rays >= 0, "n_rays is negative");
    NF_REQUIRE(n_samples >= 2 && n_samples <= 256, "n_samples must be in [2, 256] (the reference fails on a single sample: empty dists, RN:277-278)");
    if (n_rays == 0) return NERFAIL_OK;
    NF_REQUIRE(raw != nullptr && z_vals != nullptr && rays != nullptr, "raw / z_vals / rays is NULL");
    NF_REQUIRE(rgb_map != nullptr && disp_map != nullptr && acc_map != nullptr, "rgb_map / disp_map / acc_map is NULL");
    NF_REQUIRE(pts == nullptr || pts_max != nullptr, "pts is only read for pts_max");
    hipStream_t s = as_stream(stream);
    const bool one_ray_form = g_composite_select == 1;
    if (n_samples % 32 == 0 && !one_ray_form) {              // two rays per wave (64, 192 and every other multiple of 32)
        const dim3 block2(256), grid2((unsigned)((n_rays + 7) / 8));
#define NF_COMPOSITE2(IPL)                                                                                                \
    composite2_kernel<IPL><<<grid2, block2, 0, s>>>((const float4*)raw, z_vals, rays, noise, n_rays, n_samples, white_bkgd, \
                                                    rgb_map, disp_map, acc_map, weights, depth_map, pts, pts_max)
        switch (n_samples / 32) {
            case 1: NF_COMPOSITE2(1); break;  case 2: NF_COMPOSITE2(2); break;  case 3: NF_COMPOSITE2(3); break;
            case 4: NF_COMPOSITE2(4); break;  case 5: NF_COMPOSITE2(5); break;  case 6: NF_COMPOSITE2(6); break;
            case 7: NF_COMPOSITE2(7); break;  case 8: NF_COMPOSITE2(8); break;
            default: set_error("nerfail_composite: n_samples not covered by the two-ray kernel"); return NERFAIL_EINVAL;
        }
#undef NF_COMPOSITE2
        NF_LAUNCHED("composite2_kernel");
        return NERFAIL_OK;
    }
    const dim3 block(256), grid((unsigned)((n_rays + 3) / 4));
    const int ipl = (n_samples + 63) / 64;
    const bool full = n_samples == 64 * ipl;
#define NF_COMPOSITE(IPL)                                                                                                  \
    do {                                                                                                                   \
        if (full) composite_kernel<IPL, true><<<grid, block, 0, s>>>((const float4*)raw, z_vals, rays, noise, n_rays, n_samples, \
                                                 white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map, pts, pts_max); \
        else composite_kernel<IPL, false><<<grid, block, 0, s>>>((const float4*)raw, z_vals, rays, noise, n_rays, n_samples, \
                                                 white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map, pts, pts_max); \
    } while (0)
    switch (ipl) {
        case 1: NF_COMPOSITE(1); break;
        case 2: NF_COMPOSITE(2); break;
        case 3: NF_COMPOSITE(3); break;
        default: NF_COMPOSITE(4); break;
    }
#undef NF_COMPOSITE
    NF_LAUNCHED("composite_kernel");
    return NERFAIL_OK;
}
