// K3 standalone: Embedder.embed (run_nerf_helpers.py:15-50). One thread per (point, band);
// the fused MLP kernel computes the same values in registers and never calls this.
#include "common.h"

namespace nerfail {

__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ x, long M, int L, float* __restrict__ out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M * (L + 1)) return;
    const long m = g / (L + 1);
    const int band = (int)(g - m * (L + 1));     // 0 = identity, 1..L = frequency band band-1
    const int C = 3 + 6 * L;
    const float v[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
    float* o = out + m * C;
    if (band == 0) {
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        return;
    }
    // freq_bands = 2**linspace(0, L-1, L): exact powers of two; same evaluation as the fused kernels (SinCosBands)
    float* ob = o + 3 + 6 * (band - 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float sn, cs;
        SinCosBands(v[d]).band(band - 1, sn, cs);
        ob[d] = sn;
        ob[3 + d] = cs;
    }
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_embed(const float* x, int64_t M, int multires, float* out, void* stream) {
    NF_REQUIRE(M >= 0, "M is negative");
    NF_REQUIRE(multires >= 0 && multires <= 24, "multires must be in [0, 24]");
    if (M == 0) return NERFAIL_OK;
    NF_REQUIRE(x != nullptr && out != nullptr, "NULL pointer");
    const long n = M * (multires + 1);
    embed_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(x, M, multires, out);
    NF_LAUNCHED("embed_kernel");
    return NERFAIL_OK;
}
