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

namespace nerfail {

// 1 / (1 + exp(-x)) on the hardware transcendentals: v_exp_f32 (2^x, 1 ulp) of -x * log2(e), v_rcp_f32 (1 ulp). exp(-x) -> inf
// for x << 0 gives rcp(inf) = 0, like the reference's sigmoid; NaN propagates. (__frcp_rn / __expf compile to the correctly
// rounded division and the full expf in this HIP build.)
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}

template <int IPL>
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
#pragma unroll
    for (int k = 0; k <= IPL; ++k) {
        const int i = i0 + k;
        z[k] = (i < N) ? z_vals[base + i] : 0.0f;
    }
    float tprod = 1.0f;          // product of this lane's (1-alpha+1e-10)
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = i0 + k;
        alpha[k] = 0.0f; rgbr[k] = rgbg[k] = rgbb[k] = 0.0f;
        if (i < N) {
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
        if (i < N) {
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

}  // namespace nerfail

using namespace nerfail;

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
    const dim3 block(256), grid((unsigned)((n_rays + 3) / 4));
    const int ipl = (n_samples + 63) / 64;
    hipStream_t s = as_stream(stream);
#define NF_COMPOSITE(IPL)                                                                                     \
    composite_kernel<IPL><<<grid, block, 0, s>>>((const float4*)raw, z_vals, rays, noise, n_rays, n_samples,   \
                                                 white_bkgd, rgb_map, disp_map, acc_map, weights, depth_map,   \
                                                 pts, pts_max)
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
