// K1 ray generation + packing, K2 coarse sampling.
// Replaces get_rays (run_nerf_helpers.py:157-166), the viewdir normalise / pack of render()
// (run_nerf.py:102-123) and the coarse-sample block of render_rays (run_nerf.py:357-381).
// One thread per ray (or per sample); pure streaming stores, HBM-bound and tiny next to the MLP.
#include "common.h"

namespace nerfail {

struct Cam {
    float fx, fy, cx, cy;
    float c2w[12];
};

// dirs = ((i-cx)/fx, -(j-cy)/fy, -1); rays_d[a] = (dirs0*R[a][0] + dirs1*R[a][1]) + dirs2*R[a][2]
// (torch.sum over 3 products, no FMA), rays_o = c2w[:,3].
__device__ __forceinline__ void pixel_ray(const Cam& c, int col, int row, float* o, float* d) {
    const float d0 = __fdiv_rn(__fsub_rn((float)col, c.cx), c.fx);
    const float d1 = -__fdiv_rn(__fsub_rn((float)row, c.cy), c.fy);
    const float d2 = -1.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p0 = __fmul_rn(d0, c.c2w[4 * a + 0]);
        const float p1 = __fmul_rn(d1, c.c2w[4 * a + 1]);
        const float p2 = __fmul_rn(d2, c.c2w[4 * a + 2]);
        d[a] = __fadd_rn(__fadd_rn(p0, p1), p2);
        o[a] = c.c2w[4 * a + 3];
    }
}

__device__ __forceinline__ void viewdir_of(const float* d, float* v) {
    const float n = sqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = __fdiv_rn(d[a], n);
}

__global__ void get_rays_kernel(Cam c, int H, int W, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)H * W) return;
    float o[3], d[3];
    pixel_ray(c, (int)(p % W), (int)(p / W), o, d);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        rays_o[3 * p + a] = o[a];
        rays_d[3 * p + a] = d[a];
    }
}

__device__ __forceinline__ void store_ray(float* __restrict__ r, const float* o, const float* d, float near_, float far_) {
    float v[3];
    viewdir_of(d, v);
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
    r[3] = d[0]; r[4] = d[1]; r[5] = d[2];
    r[6] = near_; r[7] = far_;
    r[8] = v[0]; r[9] = v[1]; r[10] = v[2];
}

__global__ void pack_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, long n,
                                 float near_, float far_, float* __restrict__ rays) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float o[3] = {rays_o[3 * p], rays_o[3 * p + 1], rays_o[3 * p + 2]};
    float d[3] = {rays_d[3 * p], rays_d[3 * p + 1], rays_d[3 * p + 2]};
    store_ray(rays + NERFAIL_RAY_FLOATS * p, o, d, near_, far_);
}

__global__ void ray_gen_kernel(Cam c, int W, long pix_begin, long pix_count, float near_, float far_,
                               float* __restrict__ rays) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= pix_count) return;
    const long p = pix_begin + q;
    float o[3], d[3];
    pixel_ray(c, (int)(p % W), (int)(p / W), o, d);
    store_ray(rays + NERFAIL_RAY_FLOATS * q, o, d, near_, far_);
}

// z = near*(1-t) + far*t (or the lindisp form), optional stratified jitter, pts = o + d*z.
__device__ __forceinline__ float coarse_z(float near_, float far_, float t, int lindisp) {
    if (!lindisp) return __fadd_rn(__fmul_rn(near_, __fsub_rn(1.0f, t)), __fmul_rn(far_, t));
    const float a = __fmul_rn(__fdiv_rn(1.0f, near_), __fsub_rn(1.0f, t));
    const float b = __fmul_rn(__fdiv_rn(1.0f, far_), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));
}

__global__ void sample_coarse_kernel(const float* __restrict__ rays, long n_rays, const float* __restrict__ t_vals,
                                     int N, const float* __restrict__ t_rand, int lindisp,
                                     float* __restrict__ z_vals, float* __restrict__ pts) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_rays * N) return;
    const long r = g / N;
    const int i = (int)(g - r * N);
    const float* ray = rays + NERFAIL_RAY_FLOATS * r;
    const float near_ = ray[6], far_ = ray[7];
    float z = coarse_z(near_, far_, t_vals[i], lindisp);
    if (t_rand != nullptr) {   // RN:365-379
        const float zm = (i > 0) ? coarse_z(near_, far_, t_vals[i - 1], lindisp) : z;
        const float zp = (i < N - 1) ? coarse_z(near_, far_, t_vals[i + 1], lindisp) : z;
        const float upper = (i < N - 1) ? __fmul_rn(0.5f, __fadd_rn(zp, z)) : z;
        const float lower = (i > 0) ? __fmul_rn(0.5f, __fadd_rn(z, zm)) : z;
        z = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[g]));
    }
    z_vals[g] = z;
    if (pts != nullptr) {        // (NULL: the MLP kernel forms the points itself - nerfail_mlp_fwd_rays)
#pragma unroll
        for (int a = 0; a < 3; ++a) pts[3 * g + a] = mul_add_rn(ray[3 + a], z, ray[a]);
    }
}

static inline bool fill_cam(Cam& c, const float* K4, const float* c2w) {
    if (K4 == nullptr || c2w == nullptr) return false;
    c.fx = K4[0]; c.fy = K4[1]; c.cx = K4[2]; c.cy = K4[3];
    for (int i = 0; i < 12; ++i) c.c2w[i] = c2w[i];
    return true;
}

}  // namespace nerfail

using namespace nerfail;

extern "C" int nerfail_get_rays(int H, int W, const float* K4_host, const float* c2w_host, float* rays_o,
                                float* rays_d, void* stream) {
    NF_REQUIRE(H > 0 && W > 0, "H and W must be positive");
    NF_REQUIRE(rays_o != nullptr && rays_d != nullptr, "rays_o / rays_d is NULL");
    Cam c;
    NF_REQUIRE(fill_cam(c, K4_host, c2w_host), "K4_host / c2w_host is NULL");
    const long n = (long)H * W;
    get_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(c, H, W, rays_o, rays_d);
    NF_LAUNCHED("get_rays_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_pack_rays(const float* rays_o, const float* rays_d, int64_t n, float near_, float far_,
                                 float* rays, void* stream) {
    NF_REQUIRE(n >= 0, "n is negative");
    if (n == 0) return NERFAIL_OK;
    NF_REQUIRE(rays_o != nullptr && rays_d != nullptr && rays != nullptr, "NULL pointer");
    pack_rays_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(rays_o, rays_d, n, near_, far_, rays);
    NF_LAUNCHED("pack_rays_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_ray_gen(int H, int W, const float* K4_host, const float* c2w_host, float near_, float far_,
                               int64_t pix_begin, int64_t pix_count, float* rays, void* stream) {
    NF_REQUIRE(H > 0 && W > 0, "H and W must be positive");
    NF_REQUIRE(pix_begin >= 0 && pix_count >= 0 && pix_begin + pix_count <= (int64_t)H * W, "pixel range outside the image");
    if (pix_count == 0) return NERFAIL_OK;
    NF_REQUIRE(rays != nullptr, "rays is NULL");
    Cam c;
    NF_REQUIRE(fill_cam(c, K4_host, c2w_host), "K4_host / c2w_host is NULL");
    ray_gen_kernel<<<dim3((unsigned)((pix_count + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        c, W, pix_begin, pix_count, near_, far_, rays);
    NF_LAUNCHED("ray_gen_kernel");
    return NERFAIL_OK;
}

extern "C" int nerfail_sample_coarse(const float* rays, int64_t n_rays, const float* t_vals, int n_samples,
                                     const float* t_rand, int lindisp, float* z_vals, float* pts, void* stream) {
    NF_REQUIRE(n_rays >= 0 && n_samples > 0, "bad n_rays / n_samples");
    if (n_rays == 0) return NERFAIL_OK;
    NF_REQUIRE(rays != nullptr && t_vals != nullptr && z_vals != nullptr, "NULL pointer");
    const long n = n_rays * n_samples;
    sample_coarse_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(
        rays, n_rays, t_vals, n_samples, t_rand, lindisp, z_vals, pts);
    NF_LAUNCHED("sample_coarse_kernel");
    return NERFAIL_OK;
}
