// K13: fused multi-tensor Adam step of the NeRF training loop (run_nerf.py:207, :792).
// torch runs ~10 foreach kernels over 48 tensors per step (0.5 ms of a 5 ms training step); this is one launch that
// reads and writes each of p, m, v once (28.6 MB for the two D=8 W=256 networks). Bound: HBM, ~10 us.
#include "common.h"

namespace nerfail {

constexpr int kAdamMaxTensors = 48;
struct AdamTable {
    int n;
    float w1, b2, w2, eps;                 // 1-beta1, beta2, 1-beta2 (rounded from double like torch's scalar arguments)
    nerfail_adam_tensor t[kAdamMaxTensors];
};

__global__ __launch_bounds__(256) void adam_step_kernel(AdamTable tab) {
    const nerfail_adam_tensor& t = tab.t[blockIdx.y];
    const float nss = -t.step_size, bc = t.bias_correction2_sqrt;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t.numel; i += (long)gridDim.x * blockDim.x) {
        const float g = t.grad[i];
        float m = t.exp_avg[i], v = t.exp_avg_sq[i];
        m = fmaf(tab.w1, g - m, m);                                // exp_avg.lerp_(grad, 1 - beta1)
        v = fmaf(tab.w2 * g, g, v * tab.b2);                       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v) / bc + tab.eps;               // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        t.param[i] = t.param[i] + (nss * m) / denom;               // param.addcdiv_(exp_avg, denom, value=-step_size)
        t.exp_avg[i] = m;
        t.exp_avg_sq[i] = v;
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_adam_step(const nerfail_adam_tensor* tensors, int n_tensors, double beta1, double beta2, double eps,
                                 void* stream) {
    NF_REQUIRE(n_tensors >= 0, "n_tensors is negative");
    if (n_tensors == 0) return NERFAIL_OK;
    NF_REQUIRE(tensors != nullptr, "NULL pointer");
    NF_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., "invalid beta / eps");
    for (int i0 = 0; i0 < n_tensors; i0 += kAdamMaxTensors) {
        AdamTable tab;
        tab.n = n_tensors - i0 < kAdamMaxTensors ? n_tensors - i0 : kAdamMaxTensors;
        tab.w1 = (float)(1.0 - beta1); tab.b2 = (float)beta2; tab.w2 = (float)(1.0 - beta2); tab.eps = (float)eps;
        int64_t biggest = 0;
        for (int i = 0; i < tab.n; ++i) {
            const nerfail_adam_tensor& t = tensors[i0 + i];
            NF_REQUIRE(t.numel >= 0, "negative numel");
            NF_REQUIRE(t.numel == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), "NULL tensor pointer");
            NF_REQUIRE(t.bias_correction2_sqrt > 0.f, "bias_correction2_sqrt must be positive (step >= 1)");
            tab.t[i] = t;
            biggest = t.numel > biggest ? t.numel : biggest;
        }
        if (biggest == 0) continue;
        int64_t bx = (biggest + 1023) / 1024;                      // ~4 elements per thread for the largest tensor
        bx = bx > 256 ? 256 : (bx < 1 ? 1 : bx);
        adam_step_kernel<<<dim3((unsigned)bx, (unsigned)tab.n), dim3(256), 0, as_stream(stream)>>>(tab);
        NF_LAUNCHED("adam_step_kernel");
    }
    return NERFAIL_OK;
}

// ---- img2mse (run_nerf_helpers.py:9: mean((x - y)^2)) of the training loss RN:781-789, fused with its own gradient.
// torch runs 3 kernels forward and ~6 backward per call (sub, pow, mean, expand, copy, mul ...), each ~5 us of launch +
// boundary on a 1024-ray batch: two calls per training step were 1 % of the step. One launch: the mean by a fixed-order
// tree (one workgroup: the batch is 3072 values; bitwise reproducible), and d mean / d x = 2 (x - y) / n written alongside
// so that the backward is a single multiply by the upstream scalar.
namespace nerfail {
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                   float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ float part[16];
    const float inv_n = 1.0f / (float)n;
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float d = x[i] - y[i];
        s = fmaf(d, d, s);
        if (dx != nullptr) dx[i] = 2.0f * d * inv_n;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) loss[0] = v * inv_n;
    }
}
}  // namespace nerfail

extern "C" int nerfail_mse(const float* x, const float* y, int64_t n, float* loss, float* dx, void* stream) {
    NF_REQUIRE(n > 0, "n must be positive");
    NF_REQUIRE(x && y && loss, "NULL pointer");
    nerfail::mse_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(x, y, (long)n, loss, dx);
    NF_LAUNCHED("mse_kernel");
    return NERFAIL_OK;
}
