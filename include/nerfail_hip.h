/*
 * nerfail_hip.h - C ABI of libnerfail_hip.so: the MI355X (gfx950) implementation of the NeRFail
 * render-and-attack hot path.
 *
 * The reference (jiang-wenxiang/NeRFail) is pure Python/PyTorch and has no FFI of its own; its
 * boundary is a set of Python call signatures. Each entry point below names the reference
 * function (file:line, relative to the reference root) whose arithmetic it replaces; the Python
 * mirror of those signatures lives in nerfail_amd/ and binds this library with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 *   RN = Create_spatial_point_set/nerf_pytorch/run_nerf.py      RH = .../run_nerf_helpers.py
 *   NC = Create_spatial_point_set/nerf_to_coord.py              CI = .../create_index_and_dist.py
 *   GN = model/GaussNet.py    AS = attack_NeRFail_S.py          DW = tools/dist_to_weight.py
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer to float32 unless the parameter name ends in `_host`.
 *     Buffers are dense, row-major, 16-byte aligned (hipMalloc / torch allocations are).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Calls only enqueue
 *     work; they never synchronise and never allocate.
 *   - Inputs are borrowed and never written; outputs are fully overwritten unless stated.
 *   - Return value: 0 on success, otherwise a NERFAIL_E* code; nerfail_last_error() then returns a
 *     thread-local message (argument name or HIP error string).
 *   - All arithmetic is IEEE float32 (the MLP uses the exact-f32 MFMA v_mfma_f32_32x32x2_f32).
 */
#ifndef NERFAIL_HIP_H
#define NERFAIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFAIL_ABI_VERSION 7

#define NERFAIL_OK 0
#define NERFAIL_EINVAL 1   /* bad argument (null pointer, size, unsupported shape) */
#define NERFAIL_EHIP 2     /* a HIP runtime call or kernel launch failed           */
#define NERFAIL_ENODEV 3   /* no gfx950 device visible                             */

#define NERFAIL_RAY_FLOATS 11      /* o(3) d(3) near far viewdir(3): RN:116-123      */
#define NERFAIL_KNN 8              /* top_number, CI:46                              */
#define NERFAIL_MAX_DEPTH 16

int nerfail_abi_version(void);
const char* nerfail_last_error(void);
/* name of the device the library will launch on ("gfx950..." expected); NERFAIL_ENODEV if none */
int nerfail_device_name(char* buf_host, size_t buf_len);

/* ------------------------------------------------------------------ rays (K1) ------------- */

/* get_rays, RH:157-166. K4_host = {fx, fy, cx, cy} (K[0][0], K[1][1], K[0][2], K[1][2]),
 * c2w_host = 12 floats, row-major [3,4]. Writes rays_o, rays_d as [H,W,3]. */
int nerfail_get_rays(int H, int W, const float* K4_host, const float* c2w_host,
                     float* rays_o, float* rays_d, void* stream);

/* Ray packing of render(), RN:102-123 (use_viewdirs=True, ndc=False): viewdirs = d/|d|;
 * rays[n, 11] = o, d, near, far, viewdirs. */
int nerfail_pack_rays(const float* rays_o, const float* rays_d, int64_t n, float near_, float far_,
                      float* rays, void* stream);

/* get_rays + pack for the pixel range [pix_begin, pix_begin+pix_count) of an H x W image
 * (row-major pixel index = row*W + col): the per-rank unit when a view is sharded. */
int nerfail_ray_gen(int H, int W, const float* K4_host, const float* c2w_host, float near_, float far_,
                    int64_t pix_begin, int64_t pix_count, float* rays, void* stream);

/* ------------------------------------------------------------------ sampling (K2, K6) ----- */

/* Coarse samples of render_rays, RN:357-381. t_vals[N] = torch.linspace(0,1,N) (host computes it so
 * the bits are the reference's); t_rand[R,N] = stratified draws in [0,1) or NULL (perturb == 0).
 * Writes z_vals[R,N] and pts[R,N,3] = o + d*z. */
int nerfail_sample_coarse(const float* rays, int64_t n_rays, const float* t_vals, int n_samples,
                          const float* t_rand, int lindisp, float* z_vals, float* pts, void* stream);

/* sample_pdf, RH:200-243: bins[R,nb], weights[R,nb-1] -> samples[R,n]. u[R,n] explicit uniform draws
 * (u_is_row != 0: one row u[n] shared by all rays, e.g. torch.linspace(0,1,n) for det=True). */
int nerfail_sample_pdf(const float* bins, const float* weights, int64_t n_rays, int n_bins,
                       const float* u, int u_is_row, int n_samples, float* samples, void* stream);

/* Hierarchical step of render_rays, RN:392-397 + RN:412, fused: z_mid, sample_pdf on weights[...,1:-1],
 * sort(cat(z_vals, z_samples)), pts = o + d*z, z_std = std(z_samples, unbiased=False).
 * z_coarse[R,Nc], weights[R,Nc] -> z_samples[R,Nf] (may be NULL), z_fine[R,Nc+Nf], pts[R,Nc+Nf,3], z_std[R]. */
int nerfail_sample_fine(const float* rays, int64_t n_rays, const float* z_coarse, const float* weights,
                        int n_coarse, const float* u, int u_is_row, int n_fine,
                        float* z_samples, float* z_fine, float* pts, float* z_std, void* stream);

/* ------------------------------------------------------------------ MLP (K3 + K4) --------- */

/* Embedder.embed, RH:15-50 (log-sampled bands, include_input): x[M,3] -> out[M, 3 + 6*multires] =
 * [x, sin(x*2^0), cos(x*2^0), ..., sin(x*2^(L-1)), cos(x*2^(L-1))]. Standalone form of the encoding;
 * the render path never calls it (nerfail_mlp_fwd computes the encoding in registers). */
int nerfail_embed(const float* x, int64_t M, int multires, float* out, void* stream);

/* nn.Linear tensors of one NeRF (RH:83-98), weight layout [out, in] row-major as in the state_dict. */
typedef struct nerfail_mlp_params {
    int32_t D, W;                 /* netdepth, netwidth: (8,256) or (4,64); W % 32 == 0, W <= 256 */
    int32_t input_ch;             /* 63 (multires 10)                                              */
    int32_t input_ch_views;       /* 27 (multires_views 4)                                         */
    int32_t skip;                 /* layer index after which [input_pts, h] is concatenated (4), -1 = none */
    int32_t reserved;
    const float* pts_w[NERFAIL_MAX_DEPTH];   /* pts_linears.i.weight */
    const float* pts_b[NERFAIL_MAX_DEPTH];   /* pts_linears.i.bias   */
    const float* views_w;  const float* views_b;      /* views_linears.0  [W/2, W+27] */
    const float* feature_w; const float* feature_b;   /* feature_linear   [W, W]      */
    const float* alpha_w;  const float* alpha_b;      /* alpha_linear     [1, W]      */
    const float* rgb_w;    const float* rgb_b;        /* rgb_linear       [3, W/2]    */
} nerfail_mlp_params;

/* Number of floats of the MFMA-fragment-ordered weight image for a (D, W, skip) network; 0 if unsupported. */
size_t nerfail_mlp_packed_floats(int D, int W, int skip);
/* Re-pack the nn.Linear tensors into that image (device to device; call again after weights change). */
int nerfail_mlp_pack(const nerfail_mlp_params* params_host, float* packed, void* stream);

/* run_network, RN:37-51 (+ Embedder RH:15-50, NeRF.forward RH:100-123), fused: positional encoding of
 * pts (L=10) and of the per-ray viewdir (L=4) is computed in registers and never written to memory.
 * pts[M,3]; viewdirs[n_rays,3] with sample m using row m / samples_per_ray; raw[M,4] = rgb(3), sigma(1). */
int nerfail_mlp_fwd(const float* packed, int D, int W, int skip, const float* pts, const float* viewdirs,
                    int64_t M, int samples_per_ray, float* raw, void* stream);

/* Which kernel serves nerfail_mlp_fwd / nerfail_mlp_fwd_embedded: 0 = automatic (default: the LDS-streaming kernel for
 * even depths <= 8, else the register-streamed one), 1 = register-streamed (mlp.hip), 2 = LDS-streaming (mlp_lds.hip;
 * shapes it does not cover return NERFAIL_EINVAL). Both compute the same f32 FMA chains in the same order; the switch
 * exists for A/B timing and for the parity test of one against the other. Process-wide; returns the previous value. */
int nerfail_mlp_fwd_select(int which);

/* Which kernel serves nerfail_mlp_bwd_data / nerfail_mlp_bwd_data2: 0 = automatic (default: the LDS-ring kernel for W = 256 and
 * even depths <= 8, else the register-streamed one), 1 = register-streamed (mlp_bwd.hip), 2 = LDS ring (mlp_lds.hip; shapes it
 * does not cover return NERFAIL_EINVAL). Both store the same bits; the switch exists for A/B timing and for the parity test of
 * one against the other. Initial value from NERFAIL_BWD_KERNEL=reg|lds (read once). Process-wide; returns the previous value. */
int nerfail_mlp_bwd_select(int which);

/* NeRF.forward on an already embedded batch x[M, 63+27] (RH:100-123 as a standalone call). */
int nerfail_mlp_fwd_embedded(const float* packed, int D, int W, int skip, const float* x, int64_t M,
                             float* raw, void* stream);

/* ---- split-precision ("f16x3") forward: same contract as nerfail_mlp_fwd, fp32-equivalent results --------
 * Every product a*w is evaluated as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the fp16 matrix cores with fp32
 * accumulation (a_hi = fp16(a), a_lo = fp16(a - a_hi); weights pre-scaled by 2^10 inside the image). Opt-in: the
 * default path is the exact-f32 kernel. `image`: nerfail_mlp_f16_image_bytes() bytes, filled by
 * nerfail_mlp_pack_f16 from the nn.Linear tensors; `packed` is the f32 image of nerfail_mlp_pack (biases, heads). */
size_t nerfail_mlp_f16_image_bytes(int D, int W, int skip);
int nerfail_mlp_pack_f16(const nerfail_mlp_params* params_host, void* image, void* stream);
int nerfail_mlp_fwd_f16(const float* packed, const void* image, int D, int W, int skip, const float* pts,
                        const float* viewdirs, int64_t M, int samples_per_ray, float* raw, void* stream);
/* Same, additionally saving the fp32 activation tiles exactly like nerfail_mlp_fwd_train (the backward kernels
 * consume them unchanged). */
int nerfail_mlp_fwd_f16_train(const float* packed, const void* image, int D, int W, int skip, const float* pts,
                              const float* viewdirs, int64_t M, int samples_per_ray, float* raw, float* acts,
                              void* stream);

/* Split-precision backward-data: same contract as nerfail_mlp_bwd_data with the transposed fp16 image. */
size_t nerfail_mlp_f16_image_T_bytes(int D, int W, int skip);
int nerfail_mlp_pack_f16_T(const nerfail_mlp_params* params_host, void* imageT, void* stream);
int nerfail_mlp_bwd_data_f16(const float* packed, const void* imageT, int D, int W, int skip, const float* d_raw,
                             const float* acts, int64_t M, float* dz, void* stream);

/* ---- training step (RN:776-801): forward that saves activations, backward-data, weight gradients -------- */

/* Floats of the activation / gradient scratch of one pass over M samples (0 = unsupported shape). */
size_t nerfail_mlp_train_acts_floats(int D, int W, int64_t M);
size_t nerfail_mlp_train_dz_floats(int D, int W, int64_t M);
/* nerfail_mlp_fwd that additionally writes every layer's activation to `acts` (register-fragment layout). */
int nerfail_mlp_fwd_train(const float* packed, int D, int W, int skip, const float* pts, const float* viewdirs,
                          int64_t M, int samples_per_ray, float* raw, float* acts, void* stream);
/* The north-star form of nerfail_mlp_fwd / nerfail_mlp_fwd_train: the sample points are formed INSIDE the kernel from the
 * packed rays [n_rays,11] and the depths z_vals [n_rays, samples_per_ray] - pts = o + d * z with the reference's rounding
 * (RN:381, :399) - so the [M,3] point tensor is never written or read (nerfail_sample_coarse / nerfail_sample_fine accept
 * pts = NULL, nerfail_composite forms pts_max from the ray when pts is NULL). acts: NULL, or the training forward's
 * activation buffer (nerfail_mlp_train_acts_floats). Same bits as the pts form. */
int nerfail_mlp_fwd_rays(const float* packed, int D, int W, int skip, const float* rays, const float* z_vals,
                         int64_t n_rays, int samples_per_ray, float* raw, float* acts, void* stream);
/* Transposed weight image for the backward-data pass (re-pack after every optimizer step). */
size_t nerfail_mlp_packed_T_floats(int D, int W, int skip);
int nerfail_mlp_pack_T(const nerfail_mlp_params* params_host, float* packedT, void* stream);
/* nerfail_mlp_pack + nerfail_mlp_pack_T in ONE launch (the training loop re-packs both after every optimizer step). */
int nerfail_mlp_pack_train(const nerfail_mlp_params* params_host, float* packed, float* packedT, void* stream);
/* d_raw[M,4] -> gradient w.r.t. every layer's pre-activation (`dz`, fragment layout). */
int nerfail_mlp_bwd_data(const float* packed, const float* packedT, int D, int W, int skip, const float* d_raw,
                         const float* acts, int64_t M, float* dz, void* stream);
/* The same for TWO networks of one architecture in ONE launch (the coarse and the fine network of a training step:
 * independent, RN:394): d_raw / acts / dz hold the M0 samples (tiles) of network 0 followed by the M1 of network 1;
 * M0 must be a multiple of 32 when M1 > 0. */
int nerfail_mlp_bwd_data2(const float* packed0, const float* packedT0, int64_t M0, const float* packed1,
                          const float* packedT1, int64_t M1, int D, int W, int skip, const float* d_raw,
                          const float* acts, float* dz, void* stream);
/* Parameter gradients of ONE or TWO networks of the same architecture in one launch (RN:791; the coarse and the
 * fine network of a training step are independent because RN:394 detaches z_samples). `acts` / `dz` hold the tiles
 * of network 0 (M0 samples) followed by those of network 1 (M1 samples; M1 = 0: one network, then M0 need not be a
 * multiple of 32). gradsN hold DEVICE pointers shaped like the nn.Linear tensors (same struct as the weights).
 * flags: NERFAIL_DW_ACCUMULATE adds to the gradient tensors (+=), otherwise they are overwritten;
 *        NERFAIL_DW_BF16X3 converts the operands in registers to bf16 hi/lo pairs, 3 bf16 MFMAs per product block
 *        (fp32 accumulation, fp32 exponent range, ~1.5e-5 relative per product: a gradient-grade opt-in mode).
 * W = 256: LDS-staged kernel, NO float atomics - every workgroup writes its partial block to `scratch`
 * (nerfail_mlp_bwd_weights_scratch_bytes) and a second kernel adds the partials in workgroup order, so the result is
 * bitwise reproducible. Other widths (and NERFAIL_DW_KERNEL=reg): register-fed kernel with float atomics, no scratch. */
#define NERFAIL_DW_BF16X3 1
#define NERFAIL_DW_ACCUMULATE 2
size_t nerfail_mlp_bwd_weights_scratch_bytes(int D, int W, int skip, int64_t M0, int64_t M1, int flags);
int nerfail_mlp_bwd_weights(int D, int W, int skip, const float* acts, const float* dz, int64_t M0,
                            const nerfail_mlp_params* grads0, int64_t M1, const nerfail_mlp_params* grads1, int flags,
                            void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------ compositing (K5, K7) -- */

/* raw2outputs, RN:262-305, one wavefront per ray with a wave-level exclusive product scan.
 * raw[R,N,4], z_vals[R,N], rays (packed [R,11]; d is read from it), noise[R,N] already scaled by
 * raw_noise_std or NULL. Outputs rgb_map[R,3], disp_map[R], acc_map[R], weights[R,N], depth_map[R];
 * if pts_max[R,3] is non-NULL also NC:418-423 (first argmax of weights -> point): gathered from pts[R,N,3], or - pts
 * NULL - formed from the ray and its depth (o + d * z, the same bits). */
int nerfail_composite(const float* raw, const float* z_vals, const float* rays, const float* noise,
                      int64_t n_rays, int n_samples, int white_bkgd,
                      float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                      const float* pts, float* pts_max, void* stream);

/* Which kernel serves nerfail_composite: 0 = automatic (default: two rays per wave when n_samples is a multiple of 32, else
 * one ray per wave), 1 = one ray per wave everywhere. The two forms add a ray's sums in different orders (last bits). The
 * initial value is read ONCE from NERFAIL_COMPOSITE_KERNEL ("1"); the switch exists for A/B timing and for the parity test of
 * one form against the other. Process-wide; returns the previous value. */
int nerfail_composite_select(int which);

/* Backward of raw2outputs (autograd of RN:262-305, what loss.backward() at RN:791 needs): given the upstream
 * gradients of rgb_map[R,3], disp_map[R], acc_map[R], depth_map[R], weights[R,N] (each may be NULL = zero)
 * writes d_raw[R,N,4]. z_vals / rays_d receive no gradient (z_samples is detached, RN:394). */
int nerfail_composite_bwd(const float* raw, const float* z_vals, const float* rays, const float* noise,
                          int64_t n_rays, int n_samples, int white_bkgd, const float* g_rgb_map,
                          const float* g_disp_map, const float* g_acc_map, const float* g_depth_map,
                          const float* g_weights, float* d_raw, void* stream);

/* ------------------------------------------------------------------ 8-NN build (K8) ------- */

/* create_index_and_dist core, CI:126-145, exact: for each query the 8 smallest keys
 * (d2, index) with d2 = ((dx*dx + dy*dy) + dz*dz) in float32 without FMA; dist = sqrt(d2).
 * queries[Nq,3], points[M,3] (M < 2^24 so indices are exact in float32, as on disk CI:148-163).
 * dist[Nq,8] ascending; idx_f32[Nq,8] (may be NULL) and idx_i32[Nq,8] (may be NULL). */
int nerfail_knn8(const float* queries, int64_t n_queries, const float* points, int64_t n_points,
                 float* dist, float* idx_f32, int32_t* idx_i32, void* stream);

/* Same result, bit for bit, through a uniform grid (cell sort + shell expansion with a conservative termination
 * bound): ~1000x less work than the brute-force scan at 1.92 M points. workspace: scratch of
 * nerfail_knn8_grid_workspace_bytes(n_points) bytes (0 = unsupported size). */
size_t nerfail_knn8_grid_workspace_bytes(int64_t n_points);
int nerfail_knn8_grid(const float* queries, int64_t n_queries, const float* points, int64_t n_points,
                      float* dist, float* idx_f32, int32_t* idx_i32, void* workspace, size_t workspace_bytes,
                      void* stream);
/* The two halves of nerfail_knn8_grid: the grid of a point set is built ONCE into `workspace` (CI:57-61 stacks the base
 * views of a scene once) and searched for every view of the scene (CI:110-163: 400 of them) - same results, the build's
 * ~1 ms per view saved. The workspace must not be modified between build and search. */
int nerfail_knn8_grid_build(const float* points, int64_t n_points, void* workspace, size_t workspace_bytes, void* stream);
int nerfail_knn8_grid_search(const float* queries, int64_t n_queries, int64_t n_points, float* dist, float* idx_f32,
                             int32_t* idx_i32, const void* workspace, size_t workspace_bytes, void* stream);
/* The same search for the [height, width, 3] point image of ONE VIEW (pts_max, NC:418-423 -> CI:126): results in the same
 * row-major [height*width, 8] layout, but a wave searches an 8 x 8 pixel tile instead of 64 consecutive pixels of a row -
 * neighbouring pixels' points lie within a fraction of a grid cell, the 64 searches share their candidates. */
/* (all grid searches: dist / idx_f32 / idx_i32 must be 16-byte aligned - a query's eight results leave as two 16-byte stores) */
int nerfail_knn8_grid_search_view(const float* queries, int height, int width, int64_t n_points, float* dist, float* idx_f32,
                                  int32_t* idx_i32, const void* workspace, size_t workspace_bytes, void* stream);
/* Work counters of the grid search (measurement aid, off by default): while `stats` (two device uint64, zeroed by the caller)
 * is set, every search adds [0] the distances it computed (summed over queries) and [1] the queries answered by the
 * wave-cooperative search (all of a coherent wave's; the leftovers of the per-lane shell walk otherwise). NULL switches the
 * counters off again. Process-wide, not stream-ordered with other threads' searches. */
int nerfail_knn8_grid_stats(unsigned long long* stats);

/* ------------------------------------------------------------------ gauss path (K9-K12) --- */

/* create_gauss_w.forward, GN:169-186 (driver DW:82-97): dist_and_index[B,2,P,8] -> out[B,2,P,8]
 * (weight, index) with g = exp(-(d/c)^2/2), w = g/(sum g + 0.001) if sum g > 0 else 0. P = H*W. */
int nerfail_gauss_weight(const float* dist_and_index, int64_t B, int64_t P, float c, float* out, void* stream);

/* gauss_get_img.forward's hot part, GN:309-319: x_rgba[n,4] from an already gathered r[n,4] (gauss_get_r's output, GN:224-268)
 * and ori_img[n,4] (float BGRA): rgb = ori_rgb + r_rgb * (r_a/255) where ori_a > 0 else 0, a = ori_a. Unlike gauss_net's
 * composite (nerfail_gauss_fwd) there is neither an epsilon clip nor a [0,255] clip. x_rgba may alias nothing. */
int nerfail_gauss_compose(const float* ori_img, const float* r, int64_t n_pixels, float* x_rgba, void* stream);

/* gauss_net.forward hot part, GN:53-119. spatial[Ns,4] (BGRA, 0..255), weight_and_index[B,2,P,8],
 * ori_img[B,P,4] float. epsilon < 0 means None (no clip). Writes x[B,P,4], x_rgba[B,P,4];
 * eps_minmax (2 floats, may be NULL) is UPDATED with the running min / max of x_rgb*alpha (GN:89-103):
 * eps_minmax[0] = min(old, new_min), eps_minmax[1] = max(old, new_max). */
int nerfail_gauss_fwd(const float* spatial, int64_t Ns, const float* weight_and_index, const float* ori_img,
                      int64_t B, int64_t P, float epsilon, float* x, float* x_rgba, float* eps_minmax,
                      void* stream);

/* The same over views that are addressed ONE BY ONE (host table of n_views entries; all pointers device memory): the
 * device-resident maps of the attack loop, kept per view id instead of re-uploaded per iteration (MyDataset.py:199-204
 * hands out a freshly loaded 41 MB map per view and step). ori_is_u8: ori_img entries are uint8 [P,4] BGRA as cv2.imread
 * delivers them (MyDataset.py:200), else float32 [P,4]. x may be NULL (GN:83's tensor is not needed by the attack step);
 * aux_alpha [n_views*P] / aux_mask [n_views*P] (both or neither) receive alpha = x_3 / 255 and the 3-bit mask "channel c
 * passes its gradient" - all nerfail_gauss_bwd_views_rgb needs. Outputs are indexed [view][pixel] like a batch tensor. */
typedef struct nerfail_view_fwd {
    const float* weight_and_index;  /* [2,P,8] float32: weights, then indices (as float) */
    const void* ori_img;            /* [P,4] uint8 or float32 */
} nerfail_view_fwd;
int nerfail_gauss_fwd_views(const float* spatial, int64_t Ns, const nerfail_view_fwd* views, int n_views, int64_t P,
                            int ori_is_u8, float epsilon, float* x, float* x_rgba, float* aux_alpha, unsigned char* aux_mask,
                            float* eps_minmax, void* stream);

/* Backward of the above (autograd of GN:63-119): grad_spatial[Ns,4] += d/ds( sum(x*grad_x) +
 * sum(x_rgba*grad_x_rgba) ). grad_x / grad_x_rgba may be NULL (treated as zero). x is the forward's
 * saved output. grad_spatial is ACCUMULATED into (zero it first for a fresh gradient) with float atomics. */
int nerfail_gauss_bwd(const float* weight_and_index, const float* ori_img, const float* x,
                      const float* grad_x, const float* grad_x_rgba, int64_t Ns, int64_t B, int64_t P,
                      float epsilon, float* grad_spatial, void* stream);

/* Deterministic form of the same backward. The index map of a view is static, so its inverse (for each
 * row of the perturbation table, the contributions (view b, pixel p, neighbour k) that gather from it, in
 * ascending (b*P+p)*8+k order) is built once and reused by every attack epoch:
 *   row_ptr[Ns+1] int32, contrib[B*P*8] int32 (contribution id), w_sorted[B*P*8] (its weight).
 * workspace: nerfail_gauss_csr_workspace_bytes() bytes of scratch (0 = sizes unsupported). */
/* Content fingerprint of n_items equally sized device buffers laid out back to back (32-bit words): out[2i] = sum of the
 * words, out[2i+1] = sum of word * (1 + position mod 65521), modulo 2^64. Host code keys per-view inverted indices on it
 * when a view's map arrives as an anonymous tensor (MyDataset.py:199-204 + DataLoader: a fresh tensor per iteration). */
int nerfail_fingerprint(const void* data, int64_t words_per_item, int64_t n_items, uint64_t* out, void* stream);
size_t nerfail_gauss_csr_workspace_bytes(int64_t Ns, int64_t B, int64_t P);
/* row_of [B*P*8] int32: the destination row of every entry of the row-sorted list (contrib / w_sorted are in that order);
 * entries of weight 0 are sorted behind row_ptr[Ns] and never visited. */
int nerfail_gauss_csr_build(const float* weight_and_index, int64_t Ns, int64_t B, int64_t P, int32_t* row_ptr,
                            int32_t* contrib, float* w_sorted, int32_t* row_of, void* workspace, size_t workspace_bytes,
                            void* stream);
/* floats of scratch the two backward calls below need (per-pixel gradients + partial-row records); 0 = bad arguments */
size_t nerfail_gauss_bwd_scratch_floats(int64_t B, int64_t P, int n_rhs);
/* grad_spatial[Ns,4] = (accumulate ? grad_spatial : 0) + segmented reduction of w_e * (per-pixel gradient) over the
 * row-sorted entries: no atomics, fixed summation order, bitwise reproducible. */
int nerfail_gauss_bwd_csr(const float* ori_img, const float* x, const float* grad_x, const float* grad_x_rgba,
                          const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted, const int32_t* row_of,
                          int64_t Ns, int64_t B, int64_t P, float epsilon, float* scratch, int accumulate,
                          float* grad_spatial, void* stream);

/* The compact form of ONE view's index, from the arrays of nerfail_gauss_csr_build with B = 1:
 *   pos[Ns]            ordinal of row j among the view's non-empty rows, -1 for an empty row;
 *   packed[e]          pixel * 2 + (1 if entry e starts a row), for the first min(entry_capacity, row_ptr[Ns]) entries;
 *   chunk_ord[c]       ordinal of the row of entry 512 c (one int per 512 entries: nerfail_gauss_view_chunks(n) ints);
 *   n_rows[0] (device) number of non-empty rows.
 * With w_sorted that is everything the backward reads: 8 bytes per entry + 4 Ns. */
size_t nerfail_gauss_view_pack_workspace_bytes(int64_t Ns);
int64_t nerfail_gauss_view_chunks(int64_t n_entries);
int nerfail_gauss_view_pack(const int32_t* row_ptr, const int32_t* row_of, const int32_t* contrib, int64_t Ns,
                            int64_t entry_capacity, int32_t* pos, int32_t* packed, int32_t* chunk_ord, int32_t* n_rows,
                            void* workspace, size_t workspace_bytes, void* stream);

/* One view's inverted index in compact form (nerfail_gauss_view_pack). All pointers are device memory; the counts are
 * host values (read back once when the index is built). `packed` and `w_sorted` are read 16 bytes per lane at multiples of 8
 * entries: keep them 16-byte aligned (any allocator does; a slice starting at an odd entry still works, slower). */
typedef struct nerfail_view_index {
    const int32_t* packed;     /* [n_entries] pixel * 2 + row-start flag, entries sorted by destination row             */
    const float* w_sorted;     /* [n_entries] the entry's Gaussian weight                                               */
    const int32_t* chunk_ord;  /* [nerfail_gauss_view_chunks(n_entries)] row ordinal of every 512th entry               */
    const int32_t* pos;        /* [Ns]        ordinal of a row, -1 if the view has no entry for it                      */
    int64_t n_entries;         /* entries with non-zero weight (= row_ptr[Ns] of the csr build)                         */
    int64_t n_rows;            /* non-empty rows                                                                        */
} nerfail_view_index;
/* The backward of a BATCH of views through their per-view indices (host table of n_views structs): per-pixel gradients of
 * the whole batch in one pass; every view's entries reduced to that view's own row sums (all views in one launch, no
 * shared destination); the views' sums added row by row IN VIEW ORDER. No atomics, every order fixed: bitwise
 * reproducible. ori_img / x / grad_* are [n_views*P, 4]; grad_spatial [Ns,4] is overwritten. A view's map is static, so
 * its index is built once whatever batches it later appears in (the reference's DataLoader shuffles, AS:222-231).
 * scratch: nerfail_gauss_bwd_views_scratch_floats(views, n_views, P, 1) floats (0 = bad arguments). */
size_t nerfail_gauss_bwd_views_scratch_floats(const nerfail_view_index* views, int n_views, int64_t P, int n_rhs);
int nerfail_gauss_bwd_views(const float* ori_img, const float* x, const float* grad_x, const float* grad_x_rgba,
                            const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float epsilon,
                            float* scratch, float* grad_spatial, void* stream);
/* The rgb-gradient-only form for the NeRFail-S step (AS:357-392 never reads the alpha channel's gradient): upstream
 * gradient w.r.t. x_rgba only, per-pixel chain from nerfail_gauss_fwd_views' aux outputs (5 bytes per pixel instead of
 * x and ori), result as grad_rgb [Ns,3] - the buffer the perturbation-gradient all-reduce moves. Same sums, same order
 * as nerfail_gauss_bwd_views: the three channels are bitwise equal to its rgb channels. Same scratch size. */
int nerfail_gauss_bwd_views_rgb(const float* aux_alpha, const unsigned char* aux_mask, const float* grad_x_rgba,
                                const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float* scratch,
                                float* grad_rgb, void* stream);
/* ABI 7 (round 6). The same backward with the NeRFail-S sign step (AS:352-392 = nerfail_igsm_step_rgb) as the epilogue of its last
 * launch: spatial_out[j] = step(spatial[j], d CE / d spatial_rgb[j], spatial_init[j]) - for a 1-rank run, where the gradient is
 * not all-reduced, it is then neither written nor read back (23 MB each way) and one launch goes. grad_rgb: NULL, or [Ns,3] to
 * receive the gradient as well (required when n_views > 16: the running sum between launches). Same sums in the same order as
 * nerfail_gauss_bwd_views_rgb + nerfail_igsm_step_rgb: bit-identical spatial_out. spatial_out must not alias an input. */
int nerfail_gauss_bwd_views_rgb_step(const float* aux_alpha, const unsigned char* aux_mask, const float* grad_x_rgba,
                                     const nerfail_view_index* views, int n_views, int64_t Ns, int64_t P, float* scratch,
                                     float* grad_rgb, const float* spatial, const float* spatial_init, float a, float epsilon,
                                     int targeted, float* spatial_out, void* stream);
/* ONE view, n_rhs (1..8) upstream gradients at once - the class-logit gradients of one DeepFool iteration (deepfool.py:
 * 66-96 takes them one autograd.grad call at a time). grad_x_rgba: [n_rhs][P,4]; grad_spatial: [n_rhs][Ns,4],
 * overwritten. Sums run in the same order as nerfail_gauss_bwd_views, so each right-hand side gets bitwise the result of
 * a single call. scratch: nerfail_gauss_bwd_views_scratch_floats(view, 1, P, n_rhs) floats. */
int nerfail_gauss_bwd_view_multi(const float* ori_img, const float* x, const float* grad_x_rgba, int n_rhs,
                                 const nerfail_view_index* view, int64_t Ns, int64_t P, float epsilon, float* scratch,
                                 float* grad_spatial, void* stream);

/* The same backward for n_rhs (1..8) upstream gradients at once - the class-logit gradients of one DeepFool iteration
 * (deepfool.py:66-96 takes them one autograd.grad call at a time). grad_x_rgba: [n_rhs][B*P,4]; grad_spatial:
 * [n_rhs][Ns,4], overwritten. Sums run in the same order as nerfail_gauss_bwd_csr, so each right-hand side gets bitwise
 * the result of a single call. */
int nerfail_gauss_bwd_csr_multi(const float* ori_img, const float* x, const float* grad_x_rgba, int n_rhs,
                                const int32_t* row_ptr, const int32_t* contrib, const float* w_sorted, const int32_t* row_of,
                                int64_t Ns, int64_t B, int64_t P, float epsilon, float* scratch, float* grad_spatial,
                                void* stream);

/* K14 - DeepFool step arithmetic over the perturbation table (deepfool.py:76-102). grads = [n_rhs][n,4] as written by
 * nerfail_gauss_bwd_csr_multi, slice 0 = original class.
 *   norms:  norms2[k-1] = ||grads[k] - grads[0]||^2 (torch.norm(grad_prime)**2), k = 1..n_rhs-1; two-stage reduction with a
 *           fixed tree (bitwise reproducible); scratch: nerfail_deepfool_norms_scratch_bytes() bytes.
 *   apply:  rot += scale[0] * (grads[best[0]] - grads[0]) (skipped when scale[0] == 0);
 *           spatial_out = clamp(spatial_init + overshoot * rot, -255, 255), alpha channel copied from spatial_init.
 *           best (int32, 1..n_rhs-1) and scale are DEVICE scalars: no host round trip between choosing the class and
 *           applying it. rot is updated in place; spatial_out may alias nothing else. */
size_t nerfail_deepfool_norms_scratch_bytes(int n_rhs, int64_t n);
int nerfail_deepfool_norms(const float* grads, int n_rhs, int64_t n, void* scratch, size_t scratch_bytes, float* norms2,
                           void* stream);
int nerfail_deepfool_apply(const float* grads, int n_rhs, int64_t n, const int32_t* best, const float* scale,
                           float overshoot, const float* spatial_init, float* rot, float* spatial_out, void* stream);

/* NeRFail-S sign step, AS:352-392: rgb <- rgb -/+ a*sign(grad) where alpha > 0 else 0, clamped to
 * init +- epsilon; alpha channel copied. spatial/grad/spatial_init/out are [n,4]; out may alias spatial. */
int nerfail_igsm_step(const float* spatial, const float* grad, const float* spatial_init, int64_t n,
                      float a, float epsilon, int targeted, float* out, void* stream);
/* The same with the gradient as [n,3] (rgb only, nerfail_gauss_bwd_views_rgb's output). */
int nerfail_igsm_step_rgb(const float* spatial, const float* grad_rgb, const float* spatial_init, int64_t n, float a,
                          float epsilon, int targeted, float* out, void* stream);

/* K13 - one optimizer step of the NeRF training loop for ALL parameter tensors in one launch. Replaces
 * optimizer.step() of torch.optim.Adam(params, lr, betas=(0.9, 0.999)) (run_nerf.py:207, :792): no weight decay, no
 * amsgrad. Same fp32 operation order as torch's CPU kernels (bit-exact but for rare 1-ulp differences, fixture g13):
 *   m <- fma(1-beta1, g - m, m);  v <- fma((1-beta2) g, g, beta2 v);
 *   p <- p + (-step_size m) / (sqrt(v) / bias_correction2_sqrt + eps)
 * step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) are computed by the caller in double
 * (per tensor: torch keeps one step counter per parameter). `tensors` is a HOST array, copied at enqueue. */
typedef struct nerfail_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    float step_size;
    float bias_correction2_sqrt;
} nerfail_adam_tensor;
int nerfail_adam_step(const nerfail_adam_tensor* tensors, int n_tensors, double beta1, double beta2, double eps,
                      void* stream);

/* img2mse of the training loss (RH:9, RN:781-789): loss[0] = mean((x - y)^2) over n values by a fixed-order tree, and - when
 * dx is not NULL - dx[i] = 2 (x[i] - y[i]) / n, the gradient of that mean (one launch instead of ~9 torch kernels). */
int nerfail_mse(const float* x, const float* y, int64_t n, float* loss, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFAIL_HIP_H */
