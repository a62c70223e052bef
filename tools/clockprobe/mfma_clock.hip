// Measures what the MFMA pipes actually deliver on this MI355X under sustained load: a bare register-resident MFMA
// loop (one wave per SIMD on every CU, like the MLP kernels), timed with HIP events, with the shader clock read from
// s_memtime (clock64) against the constant 100 MHz wall clock. Gives the DVFS-adjusted ceilings that DESIGN.md quotes
// next to the nominal peaks. Build + run:  hipcc -O3 --offload-arch=gfx950 tools/clockprobe/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int KIND, int RANDOM>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* out, unsigned long long* clk) {
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // operands: RANDOM == 0: constants (low toggle rate, the optimistic case); 1: per-lane pseudo-random values, a
    // different register pair for each of the 8 MFMAs of the loop body (data-dependent power, the realistic case)
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f; };
    float av[8], bv[8];
    h8 hav[8], hbv[8];
    b8 bav[8], bbv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        av[t] = RANDOM ? rnd() : 1.0f + threadIdx.x * 1e-6f;
        bv[t] = RANDOM ? rnd() : 0.5f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = RANDOM ? rnd() : 0.01f * i, y = RANDOM ? rnd() : 0.5f;
            hav[t][i] = (_Float16)x; hbv[t][i] = (_Float16)y; bav[t][i] = (__bf16)x; bbv[t][i] = (__bf16)y;
        }
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[t], 0, 0, 0);
            if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hav[t], hbv[t], acc[t], 0, 0, 0);
            if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bav[t], bbv[t], acc[t], 0, 0, 0);
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND, int RANDOM>
static void run(const char* name, double flop_per_mfma, int cus, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)cus * 256 * 4);
    hipMalloc(&clk, (size_t)cus * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, RANDOM><<<cus, 256>>>(iters / 10, out, clk);          // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, RANDOM><<<cus, 256>>>(iters, out, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * cus);
    hipMemcpy(h.data(), clk, (size_t)cus * 16, hipMemcpyDeviceToHost);
    double ghz = 0.;
    for (int i = 0; i < cus; ++i) ghz += (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);     // wall clock = 100 MHz = 10 ns
    ghz /= cus;
    const double mfmas = (double)cus * 4 * 8 * iters;
    printf("%-28s %8.3f ms  %8.1f TFLOP/s  shader clock %.3f GHz  cycles per MFMA per SIMD %.1f\n", name, ms,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, ghz, (double)h[0] / (8.0 * iters));
    hipFree(out); hipFree(clk);
}

int main() {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    printf("CUs %d\n", cus);
    run<0, 0>("f32_32x32x2  constants", 2.0 * 32 * 32 * 2, cus, 400000);
    run<0, 1>("f32_32x32x2  random", 2.0 * 32 * 32 * 2, cus, 400000);
    run<1, 0>("f16_32x32x16 constants", 2.0 * 32 * 32 * 16, cus, 800000);
    run<1, 1>("f16_32x32x16 random", 2.0 * 32 * 32 * 16, cus, 800000);
    run<2, 0>("bf16_32x32x16 constants", 2.0 * 32 * 32 * 16, cus, 800000);
    run<2, 1>("bf16_32x32x16 random", 2.0 * 32 * 32 * 16, cus, 800000);
    return 0;
}
