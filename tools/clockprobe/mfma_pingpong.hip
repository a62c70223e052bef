// Round 5: why does the bare MFMA stream of nerf_mlp_fwd_lds_kernel run at ~67 cycles per v_mfma_f32_32x32x2_f32 when every
// pattern of mfma_patterns.hip runs at 64.0? The ingredient those patterns lack: TWO 8-tile accumulator arrays that swap roles
// from layer to layer (the output of one layer is the B operand of the next), so both live in AGPRs and a B operand is an
// accumulator register. One wave per SIMD on every CU; a "layer" = 32 quads x 2 steps x 16 MFMAs (8 out tiles x 128 k-steps).
//   MODE 0: B = AGPR of the input array passed directly as srcB        (what hipcc emits for the kernel without ReLU)
//   MODE 1: B = v_max_i32(0, accvgpr_read(in)) prepared one quad ahead (the kernel's lazy ReLU)
//   MODE 2: like 1, but the input array is copied to VGPRs tile by tile (16 accvgpr_read per 4 quads, one tile ahead)
//   MODE 3: B = fixed VGPRs (no dependence on the input array at all): the ceiling of this loop structure
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/clockprobe/mfma_pingpong.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu_bits(float x) { const int i = __float_as_int(x); return __int_as_float(i > 0 ? i : 0); }

template <int MODE>
__device__ __forceinline__ void layer(const f32x16 (&in)[8], f32x16 (&out)[8], const f32x4 (&A)[4]) {
    float bq[33][4];
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[0][e] = relu_bits(in[0][e]);
    }
    float tilev[16];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (MODE == 2 && (q & 3) == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tilev[r] = relu_bits(in[q >> 2][r]);
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float b;
                    if (MODE == 0) b = in[q >> 2][4 * (q & 3) + e];
                    else if (MODE == 1) b = bq[q][e];
                    else if (MODE == 2) b = tilev[4 * (q & 3) + e];
                    else b = A[e][1];
                    out[sp * 4 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t][e], b, out[sp * 4 + t], 0, 0, 0);
                }
                if (t == 0 && sp == 1 && MODE == 1 && q + 1 < 32) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bq[q + 1][e] = relu_bits(in[(q + 1) >> 2][4 * ((q + 1) & 3) + e]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    f32x16 P[8], Q[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { P[t][r] = (float)(lane + r + t) * 1e-3f - 0.02f; Q[t][r] = 0.f; }
    f32x4 A[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) A[t] = (f32x4){1e-3f * (lane + t), 2e-3f, -1e-3f, 5e-4f};
    const unsigned long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        layer<MODE>(P, Q, A);
        layer<MODE>(Q, P, A);
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += P[t][0] + P[t][7] + Q[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE>
static void run(const char* name, int cus, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)cus * 256 * 4);
    hipMalloc(&clk, (size_t)cus * 8);
    for (int rep = 0; rep < 2; ++rep) probe<MODE><<<cus, 256>>>(iters, out, clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(cus);
    hipMemcpy(h.data(), clk, cus * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-72s %6.2f cycles per MFMA (median over %d workgroups)\n", name, (double)h[cus / 2] / ((double)iters * 2048.0), cus);
    hipFree(out); hipFree(clk);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 100;
    run<3>("3 ping-pong arrays, B = fixed VGPRs", cus, iters);
    run<0>("0 ping-pong arrays, B = accumulator register of the input array (direct)", cus, iters);
    run<1>("1 ping-pong arrays, B = relu(accvgpr_read) one quad ahead (the kernel)", cus, iters);
    run<2>("2 ping-pong arrays, input tile copied to VGPRs once per 4 quads", cus, iters);
    return 0;
}
