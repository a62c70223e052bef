// K6 hierarchical sampling: sample_pdf (run_nerf_helpers.py:200-243) and the fused fine-sample
// step of render_rays (run_nerf.py:392-397 merge/sort, :412 z_std).
//
// One wavefront per ray; the ray's cdf / bins / merged z live in LDS (per-wave slice, < 2 KB).
//   - pdf normalisation: wave shuffle sum.
//   - cdf: torch's CPU cumsum accumulates in double and rounds each prefix to float; the wave scan
//     here runs in double too, so prefixes agree to the last float bit for equal pdf inputs.
//   - searchsorted(right=True): 6-step binary search in LDS.
//   - sort(cat(z_coarse, z_samples)): rank sort (each element counts smaller elements; ties by
//     position), conflict-free broadcast reads; values only, so tie order cannot change the output.
#include "common.h"

namespace nerfail {

constexpr int kMaxBins = 256;     // cdf entries per ray (n_coarse - 1 <= 255)
constexpr int kMaxMerged = 512;   // n_coarse + n_fine

// Inverse-CDF sample for one u (RH:227-241). cdf[0..nb-1] (cdf[0] = 0), bins[0..nb-1] in LDS.
__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int nb, float u) {
    // inds = number of cdf entries <= u (searchsorted right=True)
    int lo = 0, hi = nb;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = max(0, lo - 1);
    const int above = min(nb - 1, lo);
    const float cb = cdf[below], ca = cdf[above];
    const float bb = bins[below], ba = bins[above];
    float denom = __fsub_rn(ca, cb);
    if (denom < 1e-5f) denom = 1.0f;
    const float t = __fdiv_rn(__fsub_rn(u, cb), denom);
    return __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));
}

// Builds cdf (nb entries) in LDS from nb-1 weights supplied by `wfn(i)`; all 64 lanes participate.
template <typename WFn>
__device__ __forceinline__ void build_cdf(float* cdf, int nb, int lane, WFn wfn) {
    const int nw = nb - 1;
    // pass 1: sum of (w + 1e-5)
    float part = 0.f;
    for (int i = lane; i < nw; i += 64) part += __fadd_rn(wfn(i), 1e-5f);
    const float total = wave_sum(part);
    // pass 2: running prefix in double across 64-wide chunks
    double carry = 0.0;
    for (int c = 0; c < nw; c += 64) {
        const int i = c + lane;
        const float pdf = (i < nw) ? __fdiv_rn(__fadd_rn(wfn(i), 1e-5f), total) : 0.0f;
        const double incl = wave_scan_add_f64((double)pdf, lane) + carry;
        if (i < nw) cdf[i + 1] = (float)incl;
        carry = __shfl(incl, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.0f;
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights,
                                                         long n_rays, int nb, const float* __restrict__ u, int u_is_row,
                                                         int n, float* __restrict__ samples) {
    __shared__ float s_cdf[4][kMaxBins];
    __shared__ float s_bins[4][kMaxBins];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;
    float* cdf = s_cdf[wv];
    float* bn = s_bins[wv];
    const float* w = weights + ray * (nb - 1);
    for (int i = lane; i < nb; i += 64) bn[i] = bins[ray * nb + i];
    build_cdf(cdf, nb, lane, [&](int i) { return w[i]; });
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave visible to its reads
    for (int k = lane; k < n; k += 64) {
        const float uk = u_is_row ? u[k] : u[ray * n + k];
        samples[ray * n + k] = invert_cdf(cdf, bn, nb, uk);
    }
}

__global__ __launch_bounds__(256) void sample_fine_kernel(const float* __restrict__ rays, long n_rays,
                                                          const float* __restrict__ z_coarse,
                                                          const float* __restrict__ weights, int nc,
                                                          const float* __restrict__ u, int u_is_row, int nf,
                                                          float* __restrict__ z_samples, float* __restrict__ z_fine,
                                                          float* __restrict__ pts, float* __restrict__ z_std) {
    __shared__ float s_cdf[4][kMaxBins];
    __shared__ float s_bins[4][kMaxBins];
    __shared__ __attribute__((aligned(16))) float s_in[4][kMaxMerged];
    __shared__ float s_out[4][kMaxMerged];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;
    float* cdf = s_cdf[wv];
    float* bn = s_bins[wv];
    float* zin = s_in[wv];
    float* zout = s_out[wv];
    const int nb = nc - 1;            // z_vals_mid has nc-1 entries; weights[...,1:-1] has nc-2
    const int nt = nc + nf;
    const float* zc = z_coarse + ray * nc;
    const float* w = weights + ray * nc;

    for (int i = lane; i < nc; i += 64) zin[i] = zc[i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int i = lane; i < nb; i += 64) bn[i] = __fmul_rn(0.5f, __fadd_rn(zin[i + 1], zin[i]));   // RN:392
    build_cdf(cdf, nb, lane, [&](int i) { return w[i + 1]; });                                    // weights[...,1:-1]
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    float ssum = 0.f;
    for (int k = lane; k < nf; k += 64) {
        const float uk = u_is_row ? u[k] : u[ray * nf + k];
        const float zs = invert_cdf(cdf, bn, nb, uk);
        zin[nc + k] = zs;
        if (z_samples != nullptr) z_samples[ray * nf + k] = zs;
        ssum += zs;
    }
    // z_std = std(z_samples, unbiased=False) (RN:412): two-pass mean / variance
    const float mean = wave_sum(ssum) / (float)nf;
    float svar = 0.f;
    for (int k = lane; k < nf; k += 64) {
        const float dlt = zin[nc + k] - mean;   // own writes, same lane
        svar += dlt * dlt;
    }
    svar = wave_sum(svar);
    if (lane == 0) z_std[ray] = sqrt_rn(svar / (float)nf);
    for (int i = nt + lane; i < ((nt + 3) & ~3); i += 64) zin[i] = INFINITY;   // pad for the float4 sweep
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // rank sort: rank(e) = #{j : z_j < z_e or (z_j == z_e and j < e)}
    for (int e = lane; e < nt; e += 64) {
        const float v = zin[e];
        int rank = 0;
        const float4* z4 = reinterpret_cast<const float4*>(zin);
        for (int j4 = 0; j4 < (nt + 3) / 4; ++j4) {
            const float4 q = z4[j4];
            const int j = 4 * j4;
            rank += (q.x < v) || (q.x == v && j + 0 < e);
            rank += (q.y < v) || (q.y == v && j + 1 < e);
            rank += (q.z < v) || (q.z == v && j + 2 < e);
            rank += (q.w < v) || (q.w == v && j + 3 < e);
        }
        zout[rank] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    const float* rr = rays + NERFAIL_RAY_FLOATS * ray;
    const float ox = rr[0], oy = rr[1], oz = rr[2], dx = rr[3], dy = rr[4], dz = rr[5];
    for (int i = lane; i < nt; i += 64) {
        const float z = zout[i];
        z_fine[ray * nt + i] = z;
        float* p = pts + 3 * (ray * nt + i);
        p[0] = mul_add_rn(dx, z, ox);
        p[1] = mul_add_rn(dy, z, oy);
        p[2] = mul_add_rn(dz, z, oz);
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_sample_pdf(const float* bins, const float* weights, int64_t n_rays, int n_bins, const float* u,
                                  int u_is_row, int n_samples, float* samples, void* stream) {
    NF_REQUIRE(n_rays >= 0, "n_rays is negative");
    NF_REQUIRE(n_bins >= 2 && n_bins <= kMaxBins, "n_bins must be in [2, 256]");
    NF_REQUIRE(n_samples >= 1, "n_samples must be positive");
    if (n_rays == 0) return NERFAIL_OK;
    NF_REQUIRE(bins != nullptr && weights != nullptr && u != nullptr && samples != nullptr, "NULL pointer");
    sample_pdf_kernel<<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        bins, weights, n_rays, n_bins, u, u_is_row, n_samples, samples);
    NF_LAUNCHED("sample_pdf_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_sample_fine(const float* rays, int64_t n_rays, const float* z_coarse, const float* weights,
                                   int n_coarse, const float* u, int u_is_row, int n_fine, float* z_samples,
                                   float* z_fine, float* pts, float* z_std, void* stream) {
    NF_REQUIRE(n_rays >= 0, "n_rays is negative");
    NF_REQUIRE(n_coarse >= 3 && n_coarse - 1 <= kMaxBins, "n_coarse must be in [3, 257]");
    NF_REQUIRE(n_fine >= 1 && n_coarse + n_fine <= kMaxMerged - 4, "n_coarse + n_fine too large (max 508)");
    if (n_rays == 0) return NERFAIL_OK;
    NF_REQUIRE(rays != nullptr && z_coarse != nullptr && weights != nullptr && u != nullptr, "NULL input pointer");
    NF_REQUIRE(z_fine != nullptr && pts != nullptr && z_std != nullptr, "NULL output pointer");
    sample_fine_kernel<<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        rays, n_rays, z_coarse, weights, n_coarse, u, u_is_row, n_fine, z_samples, z_fine, pts, z_std);
    NF_LAUNCHED("sample_fine_kernel");
    return NERFAIL_OK;
}
