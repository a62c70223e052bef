// K6 hierarchical sampling: sample_pdf (run_nerf_helpers.py:200-243) and the fused fine-sample
// step of render_rays (run_nerf.py:392-397 merge/sort, :412 z_std).
//
// One wavefront per ray; the ray's cdf / bins / merged z live in LDS (per-wave slice, < 2 KB).
//   - pdf normalisation: wave shuffle sum.
//   - cdf: torch's CPU cumsum accumulates in double and rounds each prefix to float; the wave scan
//     here runs in double too, so prefixes agree to the last float bit for equal pdf inputs.
//   - searchsorted(right=True): 6-step binary search in LDS.
//   - sort(cat(z_coarse, z_samples)): a binary-search merge when both halves are already ascending (every
//     perturb = 0 render), else a rank sort (each element counts smaller elements; ties by position);
//     values only, so neither the algorithm nor the tie order can change the output.
#include "common.h"

namespace nerfail {

constexpr int kMaxBins = 256;     // cdf entries per ray (n_coarse - 1 <= 255)
constexpr int kMaxMerged = 512;   // n_coarse + n_fine

// #{i < n : a[i] < v} (STRICT) or #{a[i] <= v} for ascending a[0..n-1]; top = a power of two >= (n + 1) / 2.
template <bool STRICT>
__device__ __forceinline__ int count_sorted(const float* a, int n, int top, float v) {
    int pos = 0;
    for (int step = top; step > 0; step >>= 1) {
        const int p = pos + step;
        const float o = a[min(p, n) - 1];
        if (p <= n && (STRICT ? o < v : o <= v)) pos = p;
    }
    return pos;
}

// Inverse-CDF sample for one u (RH:227-241). cdf[0..nb-1] (cdf[0] = 0), bins[0..nb-1] in LDS.
__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int nb, float u) {
    // inds = number of cdf entries <= u (searchsorted right=True); cdf ascends, so a branch-free descent over
    // power-of-two steps counts them (every lane runs the same 8 steps: no exec-mask loop)
    const int lo = count_sorted<false>(cdf, nb, 1 << (31 - __clz(nb)), u);
    const int below = max(0, lo - 1);
    const int above = min(nb - 1, lo);
    const float cb = cdf[below], ca = cdf[above];
    const float bb = bins[below], ba = bins[above];
    float denom = __fsub_rn(ca, cb);
    if (denom < 1e-5f) denom = 1.0f;
    const float t = __fdiv_rn(__fsub_rn(u, cb), denom);
    return __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));
}

// The nb-1 <= 255 weights of a ray, lane-strided in registers (entry c of lane l = weight 64 c + l): loaded once, at
// the top of the kernel, so that the HBM round trip overlaps the other loads instead of sitting between two phases.
constexpr int kWRegs = (kMaxBins - 1 + 63) / 64;
__device__ __forceinline__ void load_weights(float (&wr)[kWRegs], const float* __restrict__ w, int nw, int lane) {
#pragma unroll
    for (int c = 0; c < kWRegs; ++c) wr[c] = (64 * c + lane < nw) ? w[64 * c + lane] : 0.0f;
}

// Builds cdf (nb entries) in LDS from the nb-1 weights in wr; all 64 lanes participate.
__device__ __forceinline__ void build_cdf(float* cdf, int nb, int lane, const float (&wr)[kWRegs]) {
    const int nw = nb - 1;
    // pass 1: sum of (w + 1e-5)
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < kWRegs; ++c)
        if (64 * c + lane < nw) part += __fadd_rn(wr[c], 1e-5f);
    const float total = wave_sum(part);
    // pass 2: running prefix in double across 64-wide chunks
    double carry = 0.0;
#pragma unroll
    for (int c = 0; c < kWRegs; ++c) {
        if (64 * c >= nw) break;                     // wave-uniform
        const int i = 64 * c + lane;
        const float pdf = (i < nw) ? __fdiv_rn(__fadd_rn(wr[c], 1e-5f), total) : 0.0f;
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
    float wr[kWRegs];
    load_weights(wr, weights + ray * (nb - 1), nb - 1, lane);
    for (int i = lane; i < nb; i += 64) bn[i] = bins[ray * nb + i];
    build_cdf(cdf, nb, lane, wr);
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
    __shared__ float s_cb[4][2 * kMaxBins];                                // cdf | bins
    __shared__ __attribute__((aligned(16))) float s_in[4][kMaxMerged];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;
    float* cdf = s_cb[wv];
    float* bn = s_cb[wv] + kMaxBins;
    float* zin = s_in[wv];
    const int nb = nc - 1;            // z_vals_mid has nc-1 entries; weights[...,1:-1] has nc-2
    const int nt = nc + nf;
    const float* zc = z_coarse + ray * nc;
    float wr[kWRegs];
    load_weights(wr, weights + ray * nc + 1, nb - 1, lane);                                       // weights[...,1:-1]
    const float* rr = rays + NERFAIL_RAY_FLOATS * ray;
    const float ox = rr[0], oy = rr[1], oz = rr[2], dx = rr[3], dy = rr[4], dz = rr[5];
    const float u0 = (lane < nf) ? (u_is_row ? u[lane] : u[ray * nf + lane]) : 0.0f;             // first sweep's draws

    for (int i = lane; i < nc; i += 64) zin[i] = zc[i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int i = lane; i < nb; i += 64) bn[i] = __fmul_rn(0.5f, __fadd_rn(zin[i + 1], zin[i]));   // RN:392
    build_cdf(cdf, nb, lane, wr);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    float ssum = 0.f;
    for (int k = lane; k < nf; k += 64) {
        const float uk = (k == lane) ? u0 : (u_is_row ? u[k] : u[ray * nf + k]);
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

    // sort(cat(z_coarse, z_samples)) (RN:397) - values only, so every correct sort writes the same row.
    // Both halves are normally already ascending (z_vals always; z_samples whenever u is: perturb = 0 uses
    // linspace): then an element's place in the merged row is its own index plus a binary search in the
    // OTHER half (~20 dependent LDS reads per lane instead of the 3 x 48 float4 sweeps of the rank sort below,
    // which made this kernel VALU-bound: 3.1 ms per 640 000-ray view against 0.3 ms of HBM time).
    int ordered = 1;
    for (int i = lane; i < nt - 1; i += 64)
        if (i != nc - 1) ordered &= (int)(zin[i] <= zin[i + 1]);              // false for NaN: those rows take the rank sort
    // The row is written straight from the ranks: 4-byte stores scattered inside the ray's own 768 + 2304 contiguous
    // bytes (L2 combines the lines). Measured at 640 000 rays: 0.97 ms; staging the sorted row in LDS for 16-byte
    // stores: 1.09 ms (the kernel is instruction-issue bound, ~900 wave instructions per ray, not store bound).
    float* zrow = z_fine + ray * nt;
    float* prow = pts != nullptr ? pts + 3 * ray * nt : nullptr;     // (NULL: the MLP kernel forms the points itself)
    auto emit = [&](int rank, float z) {
        zrow[rank] = z;
        if (prow != nullptr) {
            prow[3 * rank + 0] = mul_add_rn(dx, z, ox);           // pts = o + d z (RN:399)
            prow[3 * rank + 1] = mul_add_rn(dy, z, oy);
            prow[3 * rank + 2] = mul_add_rn(dz, z, oz);
        }
    };
    if (__all(ordered)) {                                                     // wave-uniform
        const int top = 1 << (31 - __clz(max(nc, nf)));
        for (int e = lane; e < nt; e += 64) {
            const float v = zin[e];
            const bool is_c = e < nc;
            // coarse i: i + #{samples < v};  sample j: j + #{coarse <= v}  (a stable merge, coarse first on ties)
            const float* other = is_c ? zin + nc : zin;
            const int n = is_c ? nf : nc;
            int lo = 0;
            for (int step = top; step > 0; step >>= 1) {
                const int p = lo + step;
                const float o = other[min(p, n) - 1];
                if (p <= n && (is_c ? (o < v) : (o <= v))) lo = p;
            }
            emit((is_c ? e : e - nc) + lo, v);
        }
        return;
    }
    // general case (random u: training with perturb = 1): rank(e) = #{j : z_j < z_e or (z_j == z_e and j < e)}
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
        emit(rank, v);
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
    NF_REQUIRE(z_fine != nullptr && z_std != nullptr, "NULL output pointer");
    sample_fine_kernel<<<dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        rays, n_rays, z_coarse, weights, n_coarse, u, u_is_row, n_fine, z_samples, z_fine, pts, z_std);
    NF_LAUNCHED("sample_fine_kernel");
    return NERFAIL_OK;
}
