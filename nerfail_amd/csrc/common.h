// Shared helpers for libnerfail_hip.so (gfx950 only). Error reporting follows include/nerfail_hip.h:
// every entry point returns 0 or a NERFAIL_E* code and leaves a thread-local message behind.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <math.h>

#include "../../include/nerfail_hip.h"

namespace nerfail {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
// Launch trace (debug aid, env NERFAIL_TRACE): 1 = write every launched kernel's name to stderr, unbuffered;
// 2 = also hipDeviceSynchronize() after the launch and write " ok" - the last name without "ok" is the kernel that faulted.
extern int g_trace;
int trace_launch(const char* name);

#define NF_REQUIRE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            ::nerfail::set_error("%s: %s", __func__, msg);      \
            return NERFAIL_EINVAL;                              \
        }                                                       \
    } while (0)

// After a <<<>>> launch: surface launch-configuration errors as NERFAIL_EHIP.
#define NF_LAUNCHED(name)                                       \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return ::nerfail::hip_fail(e__, name); \
        if (::nerfail::g_trace) {                               \
            int t__ = ::nerfail::trace_launch(name);            \
            if (t__) return t__;                                \
        }                                                       \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

// ---- wave-level primitives (64 lanes) -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ double wave_scan_add_f64(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- the same on DPP moves (row_shr 1 2 4 8 inside each row of 16 lanes, then row_bcast:15 / :31 hand the last lane of a row
// to the rows behind): no trip through the LDS crossbar (__shfl_* is ds_bpermute / ds_swizzle on gfx9), one VALU instruction
// per step. The inclusive result of lane 63 is the wave total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_f(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_scan_add_dpp(float v) {          // inclusive sum scan
    v += dpp_move_f<0x111, 0xF>(0.f, v);
    v += dpp_move_f<0x112, 0xF>(0.f, v);
    v += dpp_move_f<0x114, 0xF>(0.f, v);
    v += dpp_move_f<0x118, 0xF>(0.f, v);
    v += dpp_move_f<0x142, 0xA>(0.f, v);
    v += dpp_move_f<0x143, 0xC>(0.f, v);
    return v;
}
__device__ __forceinline__ float wave_scan_mul_dpp(float v) {          // inclusive product scan
    v *= dpp_move_f<0x111, 0xF>(1.f, v);
    v *= dpp_move_f<0x112, 0xF>(1.f, v);
    v *= dpp_move_f<0x114, 0xF>(1.f, v);
    v *= dpp_move_f<0x118, 0xF>(1.f, v);
    v *= dpp_move_f<0x142, 0xA>(1.f, v);
    v *= dpp_move_f<0x143, 0xC>(1.f, v);
    return v;
}
__device__ __forceinline__ float wave_total_dpp(float v) {             // sum of the 64 lanes, wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_add_dpp(v)), 63));
}

// Correctly rounded float32 sqrt. NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the HIP headers
// map that to __ocml_native_sqrt_f32 (approximate). sqrtf is IEEE-rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt, like the `/` behind __fdiv_rn.
__device__ __forceinline__ float sqrt_rn(float x) { return sqrtf(x); }

// sin / cos of x * 2^f for the positional-encoding bands f = 0 .. L-1 (run_nerf_helpers.py:36-41).
// The product x * 2^f is exact in float32, so the reference evaluates sin / cos of exactly that real number (with a
// ~1 ulp libm). All bands share one argument reduction: u0 = x * (2/pi) in double (relative error 2^-53), band f is
// u = u0 * 2^f (exact), q = rint(u), w = u - q in [-0.5, 0.5] quadrants, phi = w * pi/2 in [-pi/4, pi/4], then the
// classic single-precision minimax polynomials (Cephes sinf / cosf kernels, < 1 ulp on that interval) and a quadrant
// rotation. ~25 instructions per (sin, cos) pair instead of a full-range sincosf with Payne-Hanek fallback per call.
struct SinCosBands {
    double u0;
    __device__ __forceinline__ explicit SinCosBands(float x) : u0((double)x * 0.63661977236758134308) {}
    __device__ __forceinline__ void band(int f, float& sn, float& cs) const {
        const double u = u0 * (double)(1 << f);
        const double q = __builtin_rint(u);
        const float phi = (float)(u - q) * 1.57079632679489661923f;
        const float z = phi * phi;
        const float s = phi + phi * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
        const float c = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
        const int k = (int)q & 3;                 // two's complement & 3 == q mod 4 also for negative q
        const float ss = (k & 1) ? c : s, cc = (k & 1) ? s : c;
        sn = (k & 2) ? -ss : ss;
        cs = ((k + 1) & 2) ? -cc : cc;
    }
    // h ? cos : sin of band f - the ONE value a lane of the MLP kernels needs (lane half 0 feeds the sine channel of a k-step,
    // half 1 the cosine channel). cos t = sin(t + pi/2): the quadrant is advanced by h and a single select + sign flip remain
    // (7 vector instructions instead of 14 behind the two polynomials; round 5: vector instructions are never hidden beside
    // the f32 MFMA). The same polynomial values, the same sign: bit for bit band()'s h ? cs : sn.
    __device__ __forceinline__ float band_sel(int f, int h) const {
        const double u = u0 * (double)(1 << f);
        const double q = __builtin_rint(u);
        const float phi = (float)(u - q) * 1.57079632679489661923f;
        const float z = phi * phi;
        const float s = phi + phi * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
        const float c = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
        const int k = (int)q + h;
        const float v = (k & 1) ? c : s;
        return __uint_as_float(__float_as_uint(v) ^ ((unsigned)(k & 2) << 30));
    }
};

// pts = o + d*z with the reference's rounding (multiply, then add; no FMA): RN:381
__device__ __forceinline__ float mul_add_rn(float a, float b, float c) { return __fadd_rn(__fmul_rn(a, b), c); }

}  // namespace nerfail
