// K11 deterministic form: the gather backward (autograd of model/GaussNet.py:63-83) as a gather-REDUCE
// over an inverted index instead of a scatter with float atomics.
//
// The 8-NN index map of a view is static (it is built once by create_index_and_dist and reused by every
// attack epoch), so its inverse - for each row j of the perturbation table, the list of (pixel, k) that
// gather from it - is built once (radix sort of (destination, contribution id) pairs, rocPRIM via hipCUB)
// and reused. The backward is then
//     pass 1 (per pixel, streaming):  g[p] = dL/dx[p]  (chain through alpha / epsilon clip / where / clip)
//     pass 2 (per destination row):   grad_s[j] = sum_{c in row j} w_c * g[pixel_c]     in a FIXED order
// => bitwise reproducible, no atomics (the MI355X float-atomic rate for 16-byte scattered segments is
// ~0.08 TB/s; the gathers here are served by L2 / Infinity Cache).
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace nerfail {

__global__ __launch_bounds__(256) void csr_keys_kernel(const float* __restrict__ wi, long Ns, long B, long P,
                                                       unsigned* __restrict__ keys, int* __restrict__ vals) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;     // contribution id = (b*P + p)*8 + k
    if (g >= B * P * 8) return;
    const long bp = g >> 3;
    const int k = (int)(g & 7);
    const long b = bp / P, p = bp - b * P;
    long j = (long)wi[((b * 2 + 1) * P + p) * 8 + k];
    j = j < 0 ? 0 : (j >= Ns ? Ns - 1 : j);
    keys[g] = (unsigned)j;          // one list per destination row over the WHOLE batch (ids ascending inside: b, p, k)
    vals[g] = (int)g;
}

// row_ptr[r] = first sorted position whose key >= r (r in [0, Ns]); weights gathered into sorted order
__global__ __launch_bounds__(256) void csr_rows_kernel(const unsigned* __restrict__ keys_sorted, long n, long rows,
                                                       int* __restrict__ row_ptr) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)keys_sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    row_ptr[r] = (int)lo;
}

__global__ __launch_bounds__(256) void csr_weights_kernel(const float* __restrict__ wi, const int* __restrict__ contrib,
                                                          long n, long P, float* __restrict__ w_sorted) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const long g = contrib[c];
    const long bp = g >> 3;
    const long b = bp / P, p = bp - b * P;
    w_sorted[c] = wi[((b * 2 + 0) * P + p) * 8 + (g & 7)];
}

// pass 1: effective dL/dx per pixel (same chain as gauss_bwd_kernel in gauss.hip)
__global__ __launch_bounds__(256) void gauss_pixel_grad_kernel(const float4* __restrict__ ori, const float4* __restrict__ x_saved,
                                                               const float4* __restrict__ grad_x,
                                                               const float4* __restrict__ grad_x_rgba, long n, float epsilon,
                                                               float4* __restrict__ g_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float4 x = x_saved[g];
    const float4 o = ori[g];
    float4 gx = (grad_x != nullptr) ? grad_x[g] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (grad_x_rgba != nullptr && o.w > 0.f) {
        const float4 gr = grad_x_rgba[g];
        const float alpha = x.w / 255.0f;
        const float xc[3] = {x.x, x.y, x.z}, oc[3] = {o.x, o.y, o.z}, grc[3] = {gr.x, gr.y, gr.z};
        float gxc[3] = {0.f, 0.f, 0.f};
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = xc[c] * alpha;
            bool pass = true;
            if (epsilon >= 0.f) {
                pass = (d >= -epsilon) && (d <= epsilon);
                d = fminf(fmaxf(d, -epsilon), epsilon);
            }
            const float pre = oc[c] + d;
            pass = pass && (pre >= 0.f) && (pre <= 255.f);
            const float gd = pass ? grc[c] : 0.f;
            gxc[c] = gd * alpha;
            ga += gd * xc[c];
        }
        gx.x += gxc[0]; gx.y += gxc[1]; gx.z += gxc[2];
        gx.w += ga / 255.0f;
    }
    g_out[g] = gx;
}

// pass 2: one thread per destination row, contributions of the whole batch in ascending id order (fixed order =>
// bitwise reproducible).
//   * The 64 rows of a wave are one CONTIGUOUS range of the index arrays (~21 entries per row): the wave first copies
//     that range into LDS with coalesced loads (lane i takes entries i, i+64, ...). Walking the rows straight from
//     global memory instead makes every lane stream its own 84-byte-strided slice: ~43 cache lines per wave load, L1
//     thrash, ~30x read amplification.
//   * The trip count is WAVE-UNIFORM (max row length of the wave) and all loads are unconditional (clamped index,
//     weight 0 past the row's end), so the 4x unrolled loop keeps the 16-byte gathers of 4 contributions in flight.
// Gathers in flight per lane and loop iteration. The row walk is latency bound (16-byte gathers out of an 80 MB
// array): 4 -> 1.74 ms per 8-view iteration, 8 -> 1.58, 16 -> 1.42, 32 (a whole typical row at once) -> 1.33.
#ifndef NF_ROW_UNROLL
#define NF_ROW_UNROLL 32
#endif
#ifndef NF_ROW_ABLATE
#define NF_ROW_ABLATE 0     // timing experiments: 1 no gathers, 2 no output store, 3 no LDS staging (direct loads)
#endif
#ifndef NF_ROW_CAP
#define NF_ROW_CAP 2048
#endif
constexpr int kRowCap = NF_ROW_CAP;    // index entries staged per wave (8 B each in LDS); longer ranges take the direct path

constexpr int kLongRow = 256;     // rows longer than this are reduced by the whole wave

// One long row, all 64 lanes: lane l takes entries l, l+64, ... (coalesced index / weight loads, 64 gathers in flight),
// each chunk of 64 products is summed by a fixed butterfly, chunks are added in order: deterministic, and the same
// order in the single- and the multi-RHS kernel. Every lane returns the total.
template <int C>
__device__ __forceinline__ void long_row(const int* __restrict__ contrib, const float* __restrict__ w_sorted,
                                         const float4* __restrict__ g_pix, int c0, int len, int lane, float4 (&tot)[C]) {
#pragma unroll
    for (int c = 0; c < C; ++c) tot[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < len; k += 64) {
        const int i = k + lane;
        const bool ok = i < len;
        const int ci = c0 + (ok ? i : 0);
        const int id = contrib[ci];
        const float w = ok ? w_sorted[ci] : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 g = g_pix[(long)(id >> 3) * C + c];
            float4 p = ok ? make_float4(w * g.x, w * g.y, w * g.z, w * g.w) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                p.x += __shfl_xor(p.x, o, 64); p.y += __shfl_xor(p.y, o, 64);
                p.z += __shfl_xor(p.z, o, 64); p.w += __shfl_xor(p.w, o, 64);
            }
            tot[c].x += p.x; tot[c].y += p.y; tot[c].z += p.z; tot[c].w += p.w;
        }
    }
}

// Copies the wave's contiguous index / weight range into LDS. 8 coalesced load pairs are issued before the first is
// consumed (clamped addresses, so every load is unconditional): with one pair per loop iteration the ~21 iterations of
// a wave were a chain of exposed memory round trips - most of the kernel's time.
__device__ __forceinline__ void stage_rows(int* __restrict__ sid, float* __restrict__ sw, const int* __restrict__ contrib,
                                           const float* __restrict__ w_sorted, int n, int lane) {
    constexpr int U = 8;
    for (int i0 = 0; i0 < n; i0 += 64 * U) {
        int id[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 64 + lane;
            const int ic = i < n ? i : n - 1;
            id[u] = contrib[ic];
            w[u] = w_sorted[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 64 + lane;
            if (i < n) { sid[i] = id[u]; sw[i] = w[u]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's LDS writes before its reads
}

__global__ __launch_bounds__(256) void gauss_row_reduce_kernel(const int* __restrict__ row_ptr, const int* __restrict__ contrib,
                                                               const float* __restrict__ w_sorted,
                                                               const float4* __restrict__ g_pix, long Ns,
                                                               int accumulate, float4* __restrict__ grad_spatial) {
    __shared__ int s_id[4][kRowCap];
    __shared__ float s_w[4][kRowCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j0 = j - lane;                      // first row of this wave
    if (j0 >= Ns) return;                          // whole wave out of range
    const unsigned long long nf_t0 = (NF_ROW_ABLATE == 9) ? wall_clock64() : 0;
    const long jc = j < Ns ? j : Ns - 1;
    const int c0 = row_ptr[jc];
    const int len_raw = (j < Ns) ? row_ptr[jc + 1] - c0 : 0;
    // A row far longer than the typical ~21 entries (a point that is the neighbour of thousands of pixels) would keep
    // its one lane - and so the whole kernel - busy long after everything else has finished: such rows are left out of
    // the lane-per-row walk and reduced afterwards by all 64 lanes together (long_row()).
    const bool is_long = len_raw > kLongRow;
    const int len = is_long ? 0 : len_raw;
    const int base = __shfl(c0, 0, 64);
    const long jl = (j0 + 64 < Ns) ? j0 + 64 : Ns;
    const int n = row_ptr[jl] - base;              // entries of the wave's 64 rows (wave-uniform)
    int maxlen = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long nf_t1 = (NF_ROW_ABLATE == 9) ? wall_clock64() + (unsigned long long)(maxlen & 0) : 0;
    const bool staged = n <= kRowCap && NF_ROW_ABLATE != 3;
    if (staged) stage_rows(s_id[wv], s_w[wv], contrib + base, w_sorted + base, n, lane);
    const unsigned long long nf_t2 = (NF_ROW_ABLATE == 9) ? wall_clock64() + (unsigned long long)(s_id[wv][0] & 0) : 0;
    const int r0 = c0 - base;
    constexpr int U = NF_ROW_UNROLL;               // gathers in flight per lane and iteration
    for (int k = 0; k < maxlen; k += U) {
        int id[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = k + u < len;
            const int c = ok ? r0 + k + u : 0;
            if (staged) { id[u] = s_id[wv][c]; w[u] = s_w[wv][c]; }
            else { id[u] = contrib[base + c]; w[u] = w_sorted[base + c]; }
            if (!ok) w[u] = 0.f;
        }
        float4 g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) g[u] = (NF_ROW_ABLATE == 1) ? make_float4(w[u], 1.f, 2.f, (float)id[u]) : g_pix[id[u] >> 3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u < len) {                     // keeps 0 * inf / NaN of a foreign row out of the sum
                acc.x += w[u] * g[u].x; acc.y += w[u] * g[u].y; acc.z += w[u] * g[u].z; acc.w += w[u] * g[u].w;
            }
        }
    }
    for (unsigned long long lm = __ballot(is_long); lm != 0; lm &= lm - 1) {     // wave-uniform: the long rows of this wave
        const int src = __ffsll((long long)lm) - 1;
        float4 tot[1];
        long_row<1>(contrib, w_sorted, g_pix, __shfl(c0, src, 64), __shfl(len_raw, src, 64), lane, tot);
        if (lane == src) acc = tot[0];
    }
    if (NF_ROW_ABLATE == 9) {     // phase probe: ticks (10 ns) of header / staging / row walk, written instead of the result
        const unsigned long long nf_t3 = wall_clock64() + (unsigned long long)(__float_as_uint(acc.x) & 0u);
        if (j < Ns) grad_spatial[j] = make_float4((float)(nf_t1 - nf_t0), (float)(nf_t2 - nf_t1), (float)(nf_t3 - nf_t2), (float)n);
        return;
    }
    if (j >= Ns || (NF_ROW_ABLATE == 2 && acc.x != 123.456f)) return;
    if (accumulate) {
        const float4 old = grad_spatial[j];
        acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    grad_spatial[j] = acc;
}

// ---- multi-RHS form (DeepFool: the gradients of all class logits of one iteration, deepfool.py:66-96). The index
// arrays are walked ONCE for C right-hand sides; the per-pixel gradients are stored [pixel][C] so that the C float4 a
// contribution needs are one contiguous run (C = 8: one 128-byte line) instead of C scattered 16-byte gathers.
__global__ __launch_bounds__(256) void gauss_pixel_grad_multi_kernel(const float4* __restrict__ ori, const float4* __restrict__ x_saved,
                                                                     const float4* __restrict__ grad_x_rgba, long n, int C,
                                                                     float epsilon, float4* __restrict__ g_out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (pixel, rhs) pairs, rhs fastest
    if (g >= n * C) return;
    const long pix = g / C;
    const int c_ = (int)(g - pix * C);
    const float4 x = x_saved[pix];
    const float4 o = ori[pix];
    float4 gx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.w > 0.f) {
        const float4 gr = grad_x_rgba[(long)c_ * n + pix];
        const float alpha = x.w / 255.0f;
        const float xc[3] = {x.x, x.y, x.z}, oc[3] = {o.x, o.y, o.z}, grc[3] = {gr.x, gr.y, gr.z};
        float gxc[3] = {0.f, 0.f, 0.f};
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = xc[c] * alpha;
            bool pass = true;
            if (epsilon >= 0.f) {
                pass = (d >= -epsilon) && (d <= epsilon);
                d = fminf(fmaxf(d, -epsilon), epsilon);
            }
            const float pre = oc[c] + d;
            pass = pass && (pre >= 0.f) && (pre <= 255.f);
            const float gd = pass ? grc[c] : 0.f;
            gxc[c] = gd * alpha;
            ga += gd * xc[c];
        }
        gx.x += gxc[0]; gx.y += gxc[1]; gx.z += gxc[2];
        gx.w += ga / 255.0f;
    }
    g_out[g] = gx;
}

template <int C>
__global__ __launch_bounds__(256) void gauss_row_reduce_multi_kernel(const int* __restrict__ row_ptr, const int* __restrict__ contrib,
                                                                     const float* __restrict__ w_sorted,
                                                                     const float4* __restrict__ g_pix, long Ns,
                                                                     float4* __restrict__ grad_spatial) {
    __shared__ int s_id[4][kRowCap];
    __shared__ float s_w[4][kRowCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long j0 = j - lane;
    if (j0 >= Ns) return;
    const long jc = j < Ns ? j : Ns - 1;
    const int c0 = row_ptr[jc];
    const int len_raw = (j < Ns) ? row_ptr[jc + 1] - c0 : 0;
    const bool is_long = len_raw > kLongRow;       // see gauss_row_reduce_kernel
    const int len = is_long ? 0 : len_raw;
    const int base = __shfl(c0, 0, 64);
    const long jl = (j0 + 64 < Ns) ? j0 + 64 : Ns;
    const int n = row_ptr[jl] - base;
    int maxlen = len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    float4 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool staged = n <= kRowCap;
    if (staged) stage_rows(s_id[wv], s_w[wv], contrib + base, w_sorted + base, n, lane);
    const int r0 = c0 - base;
    constexpr int U = 4;
    for (int k = 0; k < maxlen; k += U) {          // same contribution order per row as the single-RHS kernel => same bits
        int id[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = k + u < len;
            const int c = ok ? r0 + k + u : 0;
            if (staged) { id[u] = s_id[wv][c]; w[u] = s_w[wv][c]; }
            else { id[u] = contrib[base + c]; w[u] = w_sorted[base + c]; }
            if (!ok) w[u] = 0.f;
        }
        float4 g[U][C];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) g[u][c] = g_pix[(long)(id[u] >> 3) * C + c];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + u < len) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    acc[c].x += w[u] * g[u][c].x; acc[c].y += w[u] * g[u][c].y;
                    acc[c].z += w[u] * g[u][c].z; acc[c].w += w[u] * g[u][c].w;
                }
            }
        }
    }
    for (unsigned long long lm = __ballot(is_long); lm != 0; lm &= lm - 1) {
        const int src = __ffsll((long long)lm) - 1;
        float4 tot[C];
        long_row<C>(contrib, w_sorted, g_pix, __shfl(c0, src, 64), __shfl(len_raw, src, 64), lane, tot);
        if (lane == src) {
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = tot[c];
        }
    }
    if (j >= Ns) return;
#pragma unroll
    for (int c = 0; c < C; ++c) grad_spatial[(long)c * Ns + j] = acc[c];
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t cub_temp_bytes(long n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (int)n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}

}  // namespace nerfail

using namespace nerfail;

extern "C" size_t nerfail_gauss_csr_workspace_bytes(int64_t Ns, int64_t B, int64_t P) {
    if (Ns <= 0 || B <= 0 || P <= 0) return 0;
    const long n = B * P * 8;
    if (n >= (1L << 31) || Ns >= (1L << 31) - 1) return 0;
    return 3 * align256((size_t)n * 4) + align256(cub_temp_bytes(n));
}

extern "C" int nerfail_gauss_csr_build(const float* weight_and_index, int64_t Ns, int64_t B, int64_t P, int32_t* row_ptr,
                                       int32_t* contrib, float* w_sorted, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    const long n = B * P * 8;
    NF_REQUIRE(n < (1L << 31) && Ns < (1L << 31) - 1, "batch too large for 32-bit CSR ids");
    NF_REQUIRE(weight_and_index && row_ptr && contrib && w_sorted && workspace, "NULL pointer");
    const size_t need = nerfail_gauss_csr_workspace_bytes(Ns, B, P);
    NF_REQUIRE(workspace_bytes >= need, "workspace too small (nerfail_gauss_csr_workspace_bytes)");
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    const size_t seg = align256((size_t)n * 4);
    unsigned* keys_in = (unsigned*)ws;
    unsigned* keys_out = (unsigned*)(ws + seg);
    int* vals_in = (int*)(ws + 2 * seg);
    void* temp = ws + 3 * seg;
    size_t temp_bytes = workspace_bytes - 3 * seg;
    csr_keys_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(weight_and_index, Ns, B, P, keys_in, vals_in);
    NF_LAUNCHED("csr_keys_kernel");
    int bits = 1;
    while ((1L << bits) < Ns + 1) ++bits;         // sort only the significant key bits (stable LSD radix)
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, contrib, (int)n, 0, bits, s);
    if (e != hipSuccess) return hip_fail(e, "hipcub::DeviceRadixSort::SortPairs");
    const long rows = Ns;
    csr_rows_kernel<<<dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, s>>>(keys_out, n, rows, row_ptr);
    NF_LAUNCHED("csr_rows_kernel");
    csr_weights_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(weight_and_index, contrib, n, P, w_sorted);
    NF_LAUNCHED("csr_weights_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd_csr(const float* ori_img, const float* x, const float* grad_x, const float* grad_x_rgba,
                                     const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted, int64_t Ns,
                                     int64_t B, int64_t P, float epsilon, float* pixel_grad_scratch, int accumulate,
                                     float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    NF_REQUIRE(ori_img && x && row_ptr && contrib && w_sorted && pixel_grad_scratch && grad_spatial, "NULL pointer");
    hipStream_t s = as_stream(stream);
    const long n = B * P;
    gauss_pixel_grad_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x, (const float4*)grad_x_rgba, n, epsilon,
        (float4*)pixel_grad_scratch);
    NF_LAUNCHED("gauss_pixel_grad_kernel");
    gauss_row_reduce_kernel<<<dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s>>>(
        row_ptr, contrib, w_sorted, (const float4*)pixel_grad_scratch, Ns, accumulate, (float4*)grad_spatial);
    NF_LAUNCHED("gauss_row_reduce_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd_csr_multi(const float* ori_img, const float* x, const float* grad_x_rgba, int n_rhs,
                                           const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted, int64_t Ns,
                                           int64_t B, int64_t P, float epsilon, float* pixel_grad_scratch,
                                           float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && B > 0 && P > 0, "bad sizes");
    NF_REQUIRE(n_rhs >= 1 && n_rhs <= 8, "n_rhs must be in 1..8");
    NF_REQUIRE(ori_img && x && grad_x_rgba && row_ptr && contrib && w_sorted && pixel_grad_scratch && grad_spatial, "NULL pointer");
    hipStream_t s = as_stream(stream);
    const long n = B * P;
    const long total = n * n_rhs;
    gauss_pixel_grad_multi_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        (const float4*)ori_img, (const float4*)x, (const float4*)grad_x_rgba, n, n_rhs, epsilon, (float4*)pixel_grad_scratch);
    NF_LAUNCHED("gauss_pixel_grad_multi_kernel");
    const dim3 grid((unsigned)((Ns + 255) / 256)), block(256);
#define NF_ROWS(C) gauss_row_reduce_multi_kernel<C><<<grid, block, 0, s>>>(row_ptr, contrib, w_sorted, (const float4*)pixel_grad_scratch, Ns, (float4*)grad_spatial)
    switch (n_rhs) {
        case 1: NF_ROWS(1); break;
        case 2: NF_ROWS(2); break;
        case 3: NF_ROWS(3); break;
        case 4: NF_ROWS(4); break;
        case 5: NF_ROWS(5); break;
        case 6: NF_ROWS(6); break;
        case 7: NF_ROWS(7); break;
        default: NF_ROWS(8); break;
    }
#undef NF_ROWS
    NF_LAUNCHED("gauss_row_reduce_multi_kernel");
    return NERFAIL_OK;
}
