// Shader cycles per v_mfma_f32_32x32x2_f32 in the instruction patterns the LDS-streaming forward kernel is made of (one
// wave per SIMD on every CU). Answers "which ingredient keeps a 16-MFMA step from running at 64 cycles per MFMA".
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/clockprobe/mfma_patterns.hip -o /tmp/mfma_patterns && /tmp/mfma_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;

// PAT: 0 e-major over 4 accumulators (reuse distance 4), operands fixed
//      1 t-major: 4 dependent MFMAs per accumulator back to back
//      2 = 0 + A operands read from LDS one step ahead (4 ds_read_b128 behind the first 4 MFMAs)
//      3 = 2 + B operands produced by v_max_i32 from accumulator registers of a second array (lazy ReLU)
//      4 = 3 with t-major first tile (the kernel's step as shipped: 4 dependent, then 12 interleaved over 3 tiles)
//      5 = 0 over 8 accumulators of one array while reading B from a second 8-tile array (register pressure: AGPR use)
// UNR: copies of the 512-MFMA body in the loop (code size = UNR x ~5 KB): does the stream outgrow the instruction cache?
template <int PAT, int UNR = 1>
__global__ __launch_bounds__(256, 1) void probe(int iters, float* out, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float smem[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc[8], in[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; in[t][r] = (float)(lane + r + t) * 1e-3f - 0.02f; }
    f32x4 fr[4], cur[4];
    const float* rl = smem + lane * 4;
    int rd = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) fr[t] = *(lds_cf4*)(rl + t * 256);
    float bfix[4] = {0.5f, 0.25f, 0.125f, 0.75f};
    float tb[16];
    float tbn[16] = {};
    float bn[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; it += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
#pragma unroll
        for (int qq = 0; qq < 32; ++qq) {        // 32 steps of 16 MFMAs (all register indices static)
            const int q = qq >> 2, kq = qq & 3;
            float b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) cur[t] = (PAT >= 2 && PAT <= 4) ? fr[t] : (f32x4){1.f, 2.f, 3.f, 4.f};
            rd = (rd + 4) & 31;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (PAT == 11) b[e] = tb[4 * kq + e];
                else if (PAT == 10) b[e] = bn[e];
                else if (PAT == 6) { float x = in[q][4 * kq + e]; asm volatile("" : "+a"(x)); in[q][4 * kq + e] = x; const int i = __float_as_int(x); b[e] = __int_as_float(i > 0 ? i : 0); }
                else if (PAT == 8) { float x = in[q][4 * kq + e]; asm volatile("" : "+a"(x)); in[q][4 * kq + e] = x; b[e] = x; }
                else if (PAT == 9) {
                    if (kq == 0 && e == 0) {
                        f32x4 w4[4];
#pragma unroll
                        for (int r = 0; r < 16; ++r) { float x = in[q][r]; asm volatile("" : "+a"(x)); w4[r >> 2][r & 3] = x; }
                        float* tp = smem + 8192 + lane * 4;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) *(__attribute__((address_space(3))) f32x4*)(tp + r4 * 256) = w4[r4];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const f32x4 v = *(lds_cf4*)(tp + r4 * 256);
#pragma unroll
                            for (int c = 0; c < 4; ++c) { const int i = __float_as_int(v[c]); tb[4 * r4 + c] = __int_as_float(i > 0 ? i : 0); }
                        }
                    }
                    b[e] = tb[4 * kq + e];
                }
                else if (PAT == 7) {
                    if (kq == 0 && e == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { float x = in[q][r]; asm volatile("" : "+a"(x)); const int i = __float_as_int(x); tb[r] = __int_as_float(i > 0 ? i : 0); }
                    }
                    b[e] = tb[4 * kq + e];
                }
                else if (PAT == 3 || PAT == 4) { const int i = __float_as_int(in[q][4 * kq + e]); b[e] = __int_as_float(i > 0 ? i : 0); }
                else if (PAT == 5) b[e] = in[q][4 * kq + e];
                else b[e] = bfix[e];
            }
            if (PAT == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][e], b[e], acc[t], 0, 0, 0);
            } else if (PAT == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[0][e], b[e], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) fr[t] = *(lds_cf4*)(rl + (rd + t) * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 1; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][e], b[e], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int tt = (PAT == 5) ? ((q & 1) * 4 + t) : t;
                        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t][e], b[e], acc[tt], 0, 0, 0);
                        if (PAT == 11 && e == 1 && t == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            float* tp = smem + 8192 + lane * 4;
                            const int qn = (q + 1) & 7;
                            if (kq == 0) {
                                f32x4 w4[4];
#pragma unroll
                                for (int r = 0; r < 16; ++r) { float x = in[qn][r]; asm volatile("" : "+a"(x)); w4[r >> 2][r & 3] = x; }
#pragma unroll
                                for (int r4 = 0; r4 < 4; ++r4) *(__attribute__((address_space(3))) f32x4*)(tp + r4 * 256) = w4[r4];
                            }
                            if (kq == 2) {
#pragma unroll
                                for (int r4 = 0; r4 < 4; ++r4) {
                                    const f32x4 v = *(lds_cf4*)(tp + r4 * 256);
#pragma unroll
                                    for (int c = 0; c < 4; ++c) { const int i = __float_as_int(v[c]); tbn[4 * r4 + c] = __int_as_float(i > 0 ? i : 0); }
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (PAT == 10 && e == 1 && t == 0) {           // operands of the NEXT step
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                const int qn = ((qq + 1) & 31) >> 2, kn = (qq + 1) & 3;
                                float x = in[qn][4 * kn + e2]; asm volatile("" : "+a"(x)); in[qn][4 * kn + e2] = x;
                                const int i = __float_as_int(x); bn[e2] = __int_as_float(i > 0 ? i : 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (e == 0 && (PAT == 2 || PAT == 3)) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) fr[t] = *(lds_cf4*)(rl + (rd + t) * 256);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (PAT == 11 && kq == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[r] = tbn[r];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7] + in[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + fr[0][0];
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int PAT, int UNR = 1>
static void run(const char* name, int cus, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)cus * 256 * 4);
    hipMalloc(&clk, (size_t)cus * 8);
    for (int rep = 0; rep < 2; ++rep) probe<PAT, UNR><<<cus, 256>>>(iters, out, clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(cus);
    hipMemcpy(h.data(), clk, cus * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %6.2f cycles per MFMA (median over %d workgroups)\n", name, (double)h[cus / 2] / ((double)iters * 512.0), cus);
    hipFree(out); hipFree(clk);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 480;
    run<0>("0 e-major over 4 accumulators, fixed operands", cus, iters);
    run<1>("1 t-major: 4 dependent MFMAs back to back", cus, iters);
    run<2>("2 e-major + A from LDS one step ahead", cus, iters);
    run<3>("3 = 2 + lazy-ReLU B operands (v_max_i32 of accumulator regs)", cus, iters);
    run<4>("4 = 3 with a dependent first tile (the kernel's step)", cus, iters);
    run<5>("5 e-major, 8 accumulators, B from a second accumulator array", cus, iters);
    run<6>("6 = 3, B array pinned in AGPRs: 4 x (accvgpr_read + v_max) per step", cus, iters);
    run<7>("7 = 6, batched: 16 x (accvgpr_read + v_max) every 4th step", cus, iters);
    run<9>("9 = 7 via LDS: ds_write_b128 from AGPRs, ds_read_b128 to VGPRs, v_max", cus, iters);
    run<10>("10 = 6 with the operands prepared one step ahead", cus, iters);
    run<11>("11 = 9 pipelined a tile ahead (LDS transit of accumulator tiles)", cus, iters);
    run<8>("8 = 6 without v_max (B passes through a VGPR copy only)", cus, iters);
    run<2, 4>("2 with the body unrolled x4   (~20 KB of code)", cus, iters);
    run<2, 8>("2 with the body unrolled x8   (~40 KB)", cus, iters);
    run<2, 12>("2 with the body unrolled x12  (~60 KB)", cus, iters);
    run<2, 16>("2 with the body unrolled x16  (~80 KB)", cus, iters);
    run<2, 24>("2 with the body unrolled x24  (~120 KB)", cus, iters);
    return 0;
}
