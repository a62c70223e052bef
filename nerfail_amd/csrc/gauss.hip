// K9-K12: the differentiable pixel <-> 3-D-point map of the NeRFail attacks.
//   K9  gauss_weight   create_gauss_w.forward           model/GaussNet.py:169-186
//   K10 gauss_fwd      gauss_net.forward (hot part)      model/GaussNet.py:53-119
//   K11 gauss_bwd      autograd of the above             (gather backward = scatter-add)
//   K12 igsm_step      NeRFail-S sign step               attack_NeRFail_S.py:352-392
//
// All four are HBM-bound streaming kernels over pixels (one thread per pixel, 32 B of weights + 32 B of
// indices per pixel read as two float4 pairs); K10 adds 8 random 16-B row gathers from the 30.7 MB
// perturbation table (Infinity-Cache resident), K11 8 x 4 float atomics per pixel into it.
#include "common.h"

namespace nerfail {

// ------------------------------------------------------------------------------------------ K9
__global__ __launch_bounds__(256) void gauss_weight_kernel(const float* __restrict__ dai, long B, long P, float c,
                                                           float* __restrict__ out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * P) return;
    const long b = g / P, p = g - b * P;
    const float4* dist4 = reinterpret_cast<const float4*>(dai + ((b * 2 + 0) * P + p) * 8);
    const float4* idx4 = reinterpret_cast<const float4*>(dai + ((b * 2 + 1) * P + p) * 8);
    float4* w4 = reinterpret_cast<float4*>(out + ((b * 2 + 0) * P + p) * 8);
    float4* i4 = reinterpret_cast<float4*>(out + ((b * 2 + 1) * P + p) * 8);
    const float4 da = dist4[0], db = dist4[1];
    const float d[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
    float gk[8];
    float ds = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float q = __fdiv_rn(d[k], c);
        gk[k] = expf(-__fdiv_rn(__fmul_rn(q, q), 2.0f));     // exp(-(d/c)^2 / 2)
        ds = __fadd_rn(ds, gk[k]);
    }
    const float dq = __fadd_rn(ds, 0.001f);
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = (ds > 0.f) ? __fdiv_rn(gk[k], dq) : 0.f;
    w4[0] = make_float4(w[0], w[1], w[2], w[3]);
    w4[1] = make_float4(w[4], w[5], w[6], w[7]);
    i4[0] = idx4[0];
    i4[1] = idx4[1];
}

// ------------------------------------------------------------------------------------------ K10
// block-level min/max of x_rgb*alpha folded into eps_minmax with float atomics (ordered-int trick
// is unnecessary: atomicMin/atomicMax on float exist for HIP via CAS-free int reinterpretation only
// for non-negative values, so use the sign-aware integer form).
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// x -> (x, x_rgba) of one pixel and the epsilon bookkeeping (GN:85-119)
// `mask`: bit c set <=> d x_rgba_c / d (x_c * alpha) = 1, i.e. the pixel is opaque and channel c is inside both clips - all the
// backward needs besides alpha when only the rgb gradient is wanted (same comparisons as gauss_pixel_grad_kernel).
__device__ __forceinline__ void finish_pixel(const float4 x, const float4 o, const float epsilon, float& emin, float& emax,
                                             float4& x_rgba, float& alpha_out, unsigned& mask) {
    const float alpha = __fdiv_rn(x.w, 255.0f);
    float dlt[3] = {__fmul_rn(x.x, alpha), __fmul_rn(x.y, alpha), __fmul_rn(x.z, alpha)};
    if (alpha > 0.f) {                // GN:89-103 bookkeeping uses where(alpha>0, x, 0) * alpha
#pragma unroll
        for (int c = 0; c < 3; ++c) { emin = fminf(emin, dlt[c]); emax = fmaxf(emax, dlt[c]); }
    }
    float rgb[3];
    const float oc[3] = {o.x, o.y, o.z};
    mask = 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float d = dlt[c];
        bool pass = o.w > 0.f;
        if (epsilon >= 0.f) {
            pass = pass && (d >= -epsilon) && (d <= epsilon);     // clip backward passes inside [min, max] inclusive
            d = fminf(fmaxf(d, -epsilon), epsilon);
        }
        const float pre = __fadd_rn(oc[c], d);
        pass = pass && (pre >= 0.f) && (pre <= 255.f);
        mask |= pass ? (1u << c) : 0u;
        const float v = (o.w > 0.f) ? pre : 0.f;
        rgb[c] = fminf(fmaxf(v, 0.f), 255.f);
    }
    alpha_out = alpha;
    x_rgba = make_float4(rgb[0], rgb[1], rgb[2], fminf(fmaxf(o.w, 0.f), 255.f));
}

__device__ __forceinline__ void fold_minmax(float emin, float emax, float* __restrict__ eps_minmax) {
    if (eps_minmax == nullptr) return;
    emin = wave_min(emin);
    emax = wave_max(emax);
    if ((threadIdx.x & 63) == 0) {
        // Only a wave that can still move the running value issues the atomic: 80 000 waves hammering two addresses
        // took 0.76 ms of a 0.92 ms launch. The read may be stale, but the values only ever move outwards, so a
        // stale one can cause a superfluous atomic, never a missed one; min / max do not depend on the order.
        // (device-scope atomic loads: a plain load is served by this XCD's L2, which the other XCDs' atomics never update)
        const float cur_min = __hip_atomic_load(eps_minmax + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float cur_max = __hip_atomic_load(eps_minmax + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (emin < 0.f && !(emin >= cur_min)) atomic_min_f32(eps_minmax + 0, emin);
        if (emax > 0.f && !(emax <= cur_max)) atomic_max_f32(eps_minmax + 1, emax);
    }
}

// Views are addressed through a pointer table (kernel argument): a batch tensor [B,2,P,8] is B consecutive entries, the
// device-resident maps of the attack loop (one tensor per view, kept by view id) are whatever the cache holds. One
// workgroup never straddles two views, so its table entry is a scalar load. ORI_U8: ori_img as the uint8 BGRA the
// reference's dataset reads (cv2.imread, MyDataset.py:200) - 4 bytes per pixel instead of 16.
// Outputs besides x_rgba are optional: x (GN:83; the first element of gauss_net's return tuple) and, for the
// rgb-gradient-only backward of the NeRFail-S step, alpha = x_3 / 255 plus the 3-bit pass mask (5 bytes per pixel
// instead of re-reading x and ori: 32).
#ifndef NF_K10_EARLY_INDEX
#define NF_K10_EARLY_INDEX 0    // measured round 5: 0.113 ms instead of 0.090 (the extra 32 B per background pixel cost more than the saved round trip)
#endif
constexpr int kFwdViews = 16;
struct FwdViews {
    const float* wi[kFwdViews];
    const void* ori[kFwdViews];
    int nv;
};

template <bool ORI_U8>
__global__ __launch_bounds__(256) void gauss_fwd_views_kernel(const float4* __restrict__ spatial, long Ns, FwdViews tab, long P,
                                                              int bpv, float epsilon, float4* __restrict__ x_out,
                                                              float4* __restrict__ x_rgba, float* __restrict__ aux_alpha,
                                                              unsigned char* __restrict__ aux_mask, float* __restrict__ eps_minmax) {
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 picks the XCD, each with its own L2). Neighbouring
    // pixels gather neighbouring rows of the table, so every XCD gets ONE contiguous eighth of the pixel range instead of
    // every eighth workgroup: a row fetched into an L2 serves the whole neighbourhood from there.
    const long per = gridDim.x >> 3;                                           // (the grid is a multiple of 8 workgroups)
    const long vb = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);          // virtual block: [xcd][position in its eighth]
    const int v = (int)(vb / bpv);                                             // wave-uniform
    const long p = (vb - (long)v * bpv) * blockDim.x + threadIdx.x;
    float emin = 0.f, emax = 0.f;   // the running values start at 0 (GN:27-28), so 0 is neutral
    if (v < tab.nv && p < P) {
        const float* __restrict__ wi = tab.wi[v];
        const long g = (long)v * P + p;
        const float4* w4 = reinterpret_cast<const float4*>(wi + p * 8);
        const float4* i4 = reinterpret_cast<const float4*>(wi + (P + p) * 8);
        const float4 wa = w4[0], wb = w4[1];
        const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        float4 o;
        if (ORI_U8) {
            const uchar4 u = reinterpret_cast<const uchar4*>(tab.ori[v])[p];
            o = make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
        } else {
            o = reinterpret_cast<const float4*>(tab.ori[v])[p];
        }
        // a background pixel (60 % of a real view): all 8 weights are 0 (GN:181) - neither its indices nor any row is read
        const bool any = (wa.x != 0.f) | (wa.y != 0.f) | (wa.z != 0.f) | (wa.w != 0.f) | (wb.x != 0.f) | (wb.y != 0.f) | (wb.z != 0.f) | (wb.w != 0.f);
        float4 rows[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) rows[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#if NF_K10_EARLY_INDEX
        // Round 5: the index words are requested TOGETHER with the weights (both streams are contiguous per pixel), not behind
        // the "any weight non-zero" test: the kernel is bound by its chain of dependent round trips (weights -> indices ->
        // rows), not by bytes; a background pixel now reads 32 bytes it does not use and every foreground pixel is one
        // memory latency shorter.
        float4 ia = i4[0], ib = i4[1];
        asm volatile("" : "+v"(ia.x), "+v"(ia.y), "+v"(ia.z), "+v"(ia.w), "+v"(ib.x), "+v"(ib.y), "+v"(ib.z), "+v"(ib.w));   // (not sunk below the branch)
        if (any) {
#else
        if (any) {
            const float4 ia = i4[0], ib = i4[1];
#endif
            const float fi[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
            const int last = (int)Ns - 1;     // (the launcher checks Ns < 2^31)
#pragma unroll
            for (int k = 0; k < 8; ++k) {     // issue all 8 gathers before using any
                // .type(torch.long): truncation (GN:62). Through int32 (round 4; the 64-bit conversion cost ~10 vector instructions
                // per index, 80 of this kernel's 286 per wave). The value is clamped to [0, Ns-1] AS A FLOAT first (v_med3_f32;
                // NaN -> 0), so the conversion itself is always in range - an out-of-range float -> int conversion is undefined in
                // C++ and the in-bounds guarantee of the gather must not rest on it. Same row as before for every float.
                int j = (int)__builtin_amdgcn_fmed3f(fi[k], 0.f, (float)last);
                j = j > last ? last : j;      // ((float)last may round up)
                rows[k] = spatial[j];         // unconditional: a per-gather "skip if w == 0" branch serialises the 8 loads
            }
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 xlo = {0.f, 0.f}, xhi = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {     // x = sum_k s[idx_k] * w_k, sequential in k (GN:81-83); multiply, then add (no
            const f32x2 w2 = {w[k], w[k]};    // contraction: -ffp-contract=off), two channels per v_pk_mul_f32 / v_pk_add_f32
            xlo = xlo + (f32x2){rows[k].x, rows[k].y} * w2;
            xhi = xhi + (f32x2){rows[k].z, rows[k].w} * w2;
        }
        const float4 x = make_float4(xlo.x, xlo.y, xhi.x, xhi.y);
        float alpha;
        unsigned mask;
        float4 xr;
        finish_pixel(x, o, epsilon, emin, emax, xr, alpha, mask);
        x_rgba[g] = xr;
        if (x_out != nullptr) x_out[g] = x;
        if (aux_alpha != nullptr) { aux_alpha[g] = alpha; aux_mask[g] = (unsigned char)mask; }
    }
    fold_minmax(emin, emax, eps_minmax);
}

// ------------------------------------------------------------------------------------------ K11
__global__ __launch_bounds__(256) void gauss_bwd_kernel(const float* __restrict__ wi, const float4* __restrict__ ori,
                                                        const float4* __restrict__ x_saved,
                                                        const float4* __restrict__ grad_x,
                                                        const float4* __restrict__ grad_x_rgba, long Ns, long B, long P,
                                                        float epsilon, float* __restrict__ grad_spatial) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * P) return;
    const long b = g / P, p = g - b * P;
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
            if (epsilon >= 0.f) {         // clip backward passes inside [min, max] inclusive
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
    if (gx.x == 0.f && gx.y == 0.f && gx.z == 0.f && gx.w == 0.f) return;   // background pixels: nothing to scatter
    const float4* w4 = reinterpret_cast<const float4*>(wi + ((b * 2 + 0) * P + p) * 8);
    const float4* i4 = reinterpret_cast<const float4*>(wi + ((b * 2 + 1) * P + p) * 8);
    const float4 wa = w4[0], wb = w4[1], ia = i4[0], ib = i4[1];
    const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    const float fi[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (w[k] == 0.f) continue;
        long j = (long)fi[k];
        j = j < 0 ? 0 : (j >= Ns ? Ns - 1 : j);
        float* dst = grad_spatial + 4 * j;
        atomicAdd(dst + 0, w[k] * gx.x);
        atomicAdd(dst + 1, w[k] * gx.y);
        atomicAdd(dst + 2, w[k] * gx.z);
        atomicAdd(dst + 3, w[k] * gx.w);
    }
}

// ------------------------------------------------------------------------------------------ K12
__global__ __launch_bounds__(256) void igsm_step_kernel(const float4* __restrict__ s, const float4* __restrict__ grad,
                                                        const float4* __restrict__ s_init, long n, float a, float epsilon,
                                                        int targeted, float4* __restrict__ out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float4 v = s[g], gr = grad[g], in = s_init[g];
    const float sv[3] = {v.x, v.y, v.z}, gv[3] = {gr.x, gr.y, gr.z}, iv[3] = {in.x, in.y, in.z};
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sg = (gv[c] > 0.f) ? 1.f : ((gv[c] < 0.f) ? -1.f : 0.f);   // torch.sign (sign(0) = 0)
        const float st = __fmul_rn(a, sg);
        float q = targeted ? __fsub_rn(sv[c], st) : __fadd_rn(sv[c], st);
        q = (v.w > 0.f) ? q : 0.f;
        q = fmaxf(q, __fsub_rn(iv[c], epsilon));
        q = fminf(q, __fadd_rn(iv[c], epsilon));
        r[c] = q;
    }
    out[g] = make_float4(r[0], r[1], r[2], v.w);
}

// the same with the gradient as [n,3] (rgb only: AS:357-392 never reads the alpha channel's gradient)
__global__ __launch_bounds__(256) void igsm_step_rgb_kernel(const float4* __restrict__ s, const float* __restrict__ grad3,
                                                            const float4* __restrict__ s_init, long n, float a, float epsilon,
                                                            int targeted, float4* __restrict__ out) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float4 v = s[g], in = s_init[g];
    const float sv[3] = {v.x, v.y, v.z}, gv[3] = {grad3[3 * g], grad3[3 * g + 1], grad3[3 * g + 2]}, iv[3] = {in.x, in.y, in.z};
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float sg = (gv[c] > 0.f) ? 1.f : ((gv[c] < 0.f) ? -1.f : 0.f);
        const float st = __fmul_rn(a, sg);
        float q = targeted ? __fsub_rn(sv[c], st) : __fadd_rn(sv[c], st);
        q = (v.w > 0.f) ? q : 0.f;
        q = fmaxf(q, __fsub_rn(iv[c], epsilon));
        q = fminf(q, __fadd_rn(iv[c], epsilon));
        r[c] = q;
    }
    out[g] = make_float4(r[0], r[1], r[2], v.w);
}

static int launch_fwd_views(const float4* spatial, long Ns, const FwdViews& tab, long P, bool ori_u8, float epsilon, float4* x,
                            float4* x_rgba, float* aux_alpha, unsigned char* aux_mask, float* eps_minmax, hipStream_t s) {
    if (Ns >= (1L << 31)) { set_error("nerfail_gauss_fwd: the perturbation table must have fewer than 2^31 rows"); return NERFAIL_EINVAL; }
    const int bpv = (int)((P + 255) / 256);
    const unsigned blocks = (unsigned)((((long)tab.nv * bpv + 7) / 8) * 8);
    if (ori_u8) gauss_fwd_views_kernel<true><<<dim3(blocks), dim3(256), 0, s>>>(spatial, Ns, tab, P, bpv, epsilon, x, x_rgba, aux_alpha, aux_mask, eps_minmax);
    else gauss_fwd_views_kernel<false><<<dim3(blocks), dim3(256), 0, s>>>(spatial, Ns, tab, P, bpv, epsilon, x, x_rgba, aux_alpha, aux_mask, eps_minmax);
    NF_LAUNCHED("gauss_fwd_views_kernel");
    return NERFAIL_OK;
}

// ------------------------------------------------------------------------------------------ gauss_get_img (GN:309-319)
// x_rgb = ori_rgb + r_rgb * (r_a / 255) where ori_a > 0 else 0, x_a = ori_a: the composite of K10 WITHOUT the epsilon clip and
// WITHOUT the [0, 255] clip (gauss_get_img has neither), on an already gathered r. Streaming, 48 B per pixel.
__global__ __launch_bounds__(256) void gauss_compose_kernel(const float4* __restrict__ ori, const float4* __restrict__ r, long n,
                                                            float4* __restrict__ x_rgba) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float4 o = ori[g], v = r[g];
    const float alpha = __fdiv_rn(v.w, 255.0f);
    const bool opaque = o.w > 0.f;
    x_rgba[g] = make_float4(opaque ? __fadd_rn(o.x, __fmul_rn(v.x, alpha)) : 0.f, opaque ? __fadd_rn(o.y, __fmul_rn(v.y, alpha)) : 0.f,
                            opaque ? __fadd_rn(o.z, __fmul_rn(v.z, alpha)) : 0.f, o.w);
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_gauss_compose(const float* ori_img, const float* r, int64_t n_pixels, float* x_rgba, void* stream) {
    NF_REQUIRE(n_pixels >= 0, "n_pixels is negative");
    if (n_pixels == 0) return NERFAIL_OK;
    NF_REQUIRE(ori_img != nullptr && r != nullptr && x_rgba != nullptr, "NULL pointer");
    gauss_compose_kernel<<<dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        (const float4*)ori_img, (const float4*)r, n_pixels, (float4*)x_rgba);
    NF_LAUNCHED("gauss_compose_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_weight(const float* dist_and_index, int64_t B, int64_t P, float c, float* out, void* stream) {
    NF_REQUIRE(B >= 0 && P >= 0, "negative size");
    NF_REQUIRE(c > 0.f, "c must be positive");
    if (B * P == 0) return NERFAIL_OK;
    NF_REQUIRE(dist_and_index != nullptr && out != nullptr, "NULL pointer");
    gauss_weight_kernel<<<dim3((unsigned)((B * P + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(dist_and_index, B, P, c, out);
    NF_LAUNCHED("gauss_weight_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_fwd(const float* spatial, int64_t Ns, const float* weight_and_index, const float* ori_img,
                                 int64_t B, int64_t P, float epsilon, float* x, float* x_rgba, float* eps_minmax,
                                 void* stream) {
    NF_REQUIRE(Ns > 0 && B >= 0 && P >= 0, "bad sizes");
    if (B * P == 0) return NERFAIL_OK;
    NF_REQUIRE(spatial != nullptr && weight_and_index != nullptr && ori_img != nullptr, "NULL input pointer");
    NF_REQUIRE(x != nullptr && x_rgba != nullptr, "NULL output pointer");
    for (int64_t b0 = 0; b0 < B; b0 += kFwdViews) {
        FwdViews tab;
        tab.nv = (int)(B - b0 < kFwdViews ? B - b0 : kFwdViews);
        for (int i = 0; i < kFwdViews; ++i) {
            const int64_t b = b0 + (i < tab.nv ? i : 0);
            tab.wi[i] = weight_and_index + b * 2 * P * 8;
            tab.ori[i] = ori_img + b * P * 4;
        }
        const int rc = launch_fwd_views((const float4*)spatial, Ns, tab, P, false, epsilon, (float4*)x + b0 * P, (float4*)x_rgba + b0 * P,
                                        nullptr, nullptr, eps_minmax, as_stream(stream));
        if (rc != NERFAIL_OK) return rc;
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_fwd_views(const float* spatial, int64_t Ns, const nerfail_view_fwd* views, int n_views, int64_t P,
                                       int ori_is_u8, float epsilon, float* x, float* x_rgba, float* aux_alpha,
                                       unsigned char* aux_mask, float* eps_minmax, void* stream) {
    NF_REQUIRE(Ns > 0 && n_views >= 0 && P >= 0, "bad sizes");
    if ((int64_t)n_views * P == 0) return NERFAIL_OK;
    NF_REQUIRE(spatial != nullptr && views != nullptr && x_rgba != nullptr, "NULL pointer");
    NF_REQUIRE((aux_alpha == nullptr) == (aux_mask == nullptr), "aux_alpha and aux_mask come together");
    for (int v = 0; v < n_views; ++v) NF_REQUIRE(views[v].weight_and_index != nullptr && views[v].ori_img != nullptr, "a view has a NULL map or image");
    for (int v0 = 0; v0 < n_views; v0 += kFwdViews) {
        FwdViews tab;
        tab.nv = n_views - v0 < kFwdViews ? n_views - v0 : kFwdViews;
        for (int i = 0; i < kFwdViews; ++i) {
            const nerfail_view_fwd& vw = views[v0 + (i < tab.nv ? i : 0)];
            tab.wi[i] = vw.weight_and_index;
            tab.ori[i] = vw.ori_img;
        }
        const int rc = launch_fwd_views((const float4*)spatial, Ns, tab, P, ori_is_u8 != 0, epsilon,
                                        x ? (float4*)x + (int64_t)v0 * P : nullptr, (float4*)x_rgba + (int64_t)v0 * P,
                                        aux_alpha ? aux_alpha + (int64_t)v0 * P : nullptr, aux_mask ? aux_mask + (int64_t)v0 * P : nullptr,
                                        eps_minmax, as_stream(stream));
        if (rc != NERFAIL_OK) return rc;
    }
    return NERFAIL_OK;
}

extern "C" int nerfail_gauss_bwd(const float* weight_and_index, const float* ori_img, const float* x, const float* grad_x,
                                 const float* grad_x_rgba, int64_t Ns, int64_t B, int64_t P, float epsilon,
                                 float* grad_spatial, void* stream) {
    NF_REQUIRE(Ns > 0 && B >= 0 && P >= 0, "bad sizes");
    if (B * P == 0) return NERFAIL_OK;
    NF_REQUIRE(weight_and_index != nullptr && ori_img != nullptr && x != nullptr, "NULL input pointer");
    NF_REQUIRE(grad_spatial != nullptr, "grad_spatial is NULL");
    if (grad_x == nullptr && grad_x_rgba == nullptr) return NERFAIL_OK;
    gauss_bwd_kernel<<<dim3((unsigned)((B * P + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        weight_and_index, (const float4*)ori_img, (const float4*)x, (const float4*)grad_x, (const float4*)grad_x_rgba, Ns,
        B, P, epsilon, grad_spatial);
    NF_LAUNCHED("gauss_bwd_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_igsm_step(const float* spatial, const float* grad, const float* spatial_init, int64_t n, float a,
                                 float epsilon, int targeted, float* out, void* stream) {
    NF_REQUIRE(n >= 0, "n is negative");
    if (n == 0) return NERFAIL_OK;
    NF_REQUIRE(spatial != nullptr && grad != nullptr && spatial_init != nullptr && out != nullptr, "NULL pointer");
    igsm_step_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        (const float4*)spatial, (const float4*)grad, (const float4*)spatial_init, n, a, epsilon, targeted, (float4*)out);
    NF_LAUNCHED("igsm_step_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_igsm_step_rgb(const float* spatial, const float* grad_rgb, const float* spatial_init, int64_t n, float a,
                                     float epsilon, int targeted, float* out, void* stream) {
    NF_REQUIRE(n >= 0, "n is negative");
    if (n == 0) return NERFAIL_OK;
    NF_REQUIRE(spatial != nullptr && grad_rgb != nullptr && spatial_init != nullptr && out != nullptr, "NULL pointer");
    igsm_step_rgb_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        (const float4*)spatial, grad_rgb, (const float4*)spatial_init, n, a, epsilon, targeted, (float4*)out);
    NF_LAUNCHED("igsm_step_rgb_kernel");
    return NERFAIL_OK;
}
