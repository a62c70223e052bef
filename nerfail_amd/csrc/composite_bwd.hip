// K5b: backward of raw2outputs (run_nerf.py:262-305), i.e. what loss.backward() (RN:791) pushes into `raw`.
//
// One wavefront per ray, same sample->lane mapping as the forward (lane owns IPL consecutive samples). The
// forward quantities (alpha, exp term, transmittance, weights, sigmoid colours) are recomputed, then
//     gw_i      = dL/dw_i = <g_rgb, c_i> - [white_bkgd] sum(g_rgb) + g_acc + g_depth * z_i + g_weights_i
//     dL/dalpha = gw_i * T_i - (sum_{k>i} gw_k w_k) / (1 - alpha_i + 1e-10)         (cumprod backward)
//     dL/dsigma = dL/dalpha * dist_i * exp(-sigma_i dist_i) * [sigma_i > 0]
//     dL/draw_c = w_i * g_rgb_c * c (1 - c)
// with the suffix sum as a reverse 64-lane shuffle scan in double precision (as torch's CPU cumsum accumulates). g_disp is folded into g_depth / g_acc through
// disp = 1/max(1e-10, depth/acc). HBM-bound: reads 20N+..., writes 16N bytes per ray.
#include "common.h"

namespace nerfail {

// inclusive suffix sum across lanes: out[l] = sum_{m >= l} v[m]. Double precision (round 5): torch's cumprod backward is
// reversed_cumsum(grad * output) / input, and the CPU cumsum accumulates fp32 inputs in double (acc_type<float, false>),
// so the reference's own fp32 path carries this sum wider than fp32; an fp32 shuffle scan was ~20x noisier than the
// reference's fp32-vs-fp64 spread on alpha_linear's gradient. The kernel is 0.1 % of a training step.
__device__ __forceinline__ double wave_suffix_sum_f64(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    return v;
}

// inclusive product scan across lanes in double. torch's CPU cumprod (RN:295) keeps its running product in double as well
// (acc_type<float, false>) and rounds it to fp32 per position: with the double scan T_i here IS that value, and the
// cancellation in dL/dalpha = gw_i T_i - S_i / (1 - alpha_i + 1e-10) no longer amplifies a product-tree rounding of T.
__device__ __forceinline__ double wave_scan_mul_f64(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}

template <int IPL>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays,
    const float* __restrict__ noise, long n_rays, int N, int white_bkgd,
    const float* __restrict__ g_rgb, const float* __restrict__ g_disp, const float* __restrict__ g_acc,
    const float* __restrict__ g_depth, const float* __restrict__ g_weights, float4* __restrict__ d_raw) {
    const int lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* rr = rays + NERFAIL_RAY_FLOATS * ray;
    const float dx = rr[3], dy = rr[4], dz = rr[5];
    const float nrm = sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const long base = ray * N;
    const int i0 = lane * IPL;

    float z[IPL + 1], alpha[IPL], ex[IPL], dist[IPL], sig[IPL], cr[IPL], cg[IPL], cb[IPL], tt[IPL];
#pragma unroll
    for (int k = 0; k <= IPL; ++k) z[k] = (i0 + k < N) ? z_vals[base + i0 + k] : 0.0f;
    double tprod = 1.0;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        const int i = i0 + k;
        alpha[k] = 0.f; ex[k] = 1.f; dist[k] = 0.f; sig[k] = 0.f; cr[k] = cg[k] = cb[k] = 0.f; tt[k] = 1.f;
        if (i < N) {
            const float4 rw = raw[base + i];
            float d = (i < N - 1) ? __fsub_rn(z[k + 1], z[k]) : 1e10f;
            d = __fmul_rn(d, nrm);
            float s = rw.w;
            if (noise != nullptr) s = __fadd_rn(s, noise[base + i]);
            sig[k] = s;
            dist[k] = d;
            ex[k] = expf(-__fmul_rn(fmaxf(s, 0.f), d));
            alpha[k] = __fsub_rn(1.0f, ex[k]);
            cr[k] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw.x)));
            cg[k] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw.y)));
            cb[k] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-rw.z)));
            tt[k] = __fadd_rn(__fsub_rn(1.0f, alpha[k]), 1e-10f);
            tprod *= (double)tt[k];
        }
    }
    const double incl = wave_scan_mul_f64(tprod, lane);
    double T0 = __shfl_up(incl, 1, 64);
    if (lane == 0) T0 = 1.0;

    float gr = 0.f, gg = 0.f, gb = 0.f;
    if (g_rgb != nullptr) { gr = g_rgb[3 * ray]; gg = g_rgb[3 * ray + 1]; gb = g_rgb[3 * ray + 2]; }
    float ga = (g_acc != nullptr) ? g_acc[ray] : 0.f;
    float gd = (g_depth != nullptr) ? g_depth[ray] : 0.f;
    if (white_bkgd) ga -= (gr + gg + gb);                  // rgb_map += 1 - acc
    float T[IPL], w[IPL];
    {
        double Tk = T0;
        float sd = 0.f, sa = 0.f;
#pragma unroll
        for (int k = 0; k < IPL; ++k) {
            T[k] = (float)Tk;
            w[k] = __fmul_rn(alpha[k], T[k]);
            if (i0 + k < N) { sd += w[k] * z[k]; sa += w[k]; }
            Tk *= (double)tt[k];
        }
        if (g_disp != nullptr) {                           // disp = 1 / max(1e-10, depth / acc)
            sd = wave_sum(sd); sa = wave_sum(sa);
            const float ratio = sd / sa;
            if (ratio > 1e-10f) {                          // false for NaN and for the clamped branch
                const float gratio = -g_disp[ray] / (ratio * ratio);
                gd += gratio / sa;
                ga -= gratio * sd / (sa * sa);
            }
        }
    }
    float gw[IPL], gp[IPL];
    double gww = 0.0;
#pragma unroll
    for (int k = 0; k < IPL; ++k) {
        gw[k] = 0.f; gp[k] = 0.f;
        if (i0 + k < N) {
            gw[k] = gr * cr[k] + gg * cg[k] + gb * cb[k] + ga + gd * z[k];
            if (g_weights != nullptr) gw[k] += g_weights[base + i0 + k];
            gp[k] = __fmul_rn(__fmul_rn(gw[k], alpha[k]), T[k]);     // grad_T * T, an fp32 product as autograd forms it
            gww += (double)gp[k];
        }
    }
    // S for this lane's LAST sample = sum over later lanes; walk backwards inside the lane (all of it in double)
    double S = wave_suffix_sum_f64(gww, lane) - gww;       // contributions of lanes > this one
#pragma unroll
    for (int k = IPL - 1; k >= 0; --k) {
        const int i = i0 + k;
        if (i < N) {
            const float dalpha = __fsub_rn(__fmul_rn(gw[k], T[k]), __fdiv_rn((float)S, tt[k]));
            const float dsig = (sig[k] > 0.f) ? dalpha * dist[k] * ex[k] : 0.f;
            d_raw[base + i] = make_float4(w[k] * gr * cr[k] * (1.f - cr[k]), w[k] * gg * cg[k] * (1.f - cg[k]),
                                          w[k] * gb * cb[k] * (1.f - cb[k]), dsig);
            S += (double)gp[k];
        }
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_composite_bwd(const float* raw, const float* z_vals, const float* rays, const float* noise,
                                     int64_t n_rays, int n_samples, int white_bkgd, const float* g_rgb_map,
                                     const float* g_disp_map, const float* g_acc_map, const float* g_depth_map,
                                     const float* g_weights, float* d_raw, void* stream) {
    NF_REQUIRE(n_rays >= 0, "n_rays is negative");
    NF_REQUIRE(n_samples >= 2 && n_samples <= 256, "n_samples must be in [2, 256]");
    if (n_rays == 0) return NERFAIL_OK;
    NF_REQUIRE(raw != nullptr && z_vals != nullptr && rays != nullptr && d_raw != nullptr, "NULL pointer");
    const dim3 block(256), grid((unsigned)((n_rays + 3) / 4));
    hipStream_t s = as_stream(stream);
#define NF_CB(IPL)                                                                                                  \
    composite_bwd_kernel<IPL><<<grid, block, 0, s>>>((const float4*)raw, z_vals, rays, noise, n_rays, n_samples,     \
                                                     white_bkgd, g_rgb_map, g_disp_map, g_acc_map, g_depth_map,      \
                                                     g_weights, (float4*)d_raw)
    switch ((n_samples + 63) / 64) {
        case 1: NF_CB(1); break;
        case 2: NF_CB(2); break;
        case 3: NF_CB(3); break;
        default: NF_CB(4); break;
    }
#undef NF_CB
    NF_LAUNCHED("composite_bwd_kernel");
    return NERFAIL_OK;
}
