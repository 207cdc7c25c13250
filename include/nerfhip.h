/* nerfhip.h — C ABI of libnerfhip.so: the MI355X (gfx950) NeRF volume-rendering hot path.
 *
 * Drop-in boundary for kwea123/nerf_pl's `models/nerf.py` + `models/rendering.py`.
 * The reference has no FFI of its own for this path except the `torchsearchsorted`
 * extension (rendering.py:2,42); every other entry point below replaces a *sequence of
 * ATen launches* inside one reference Python function, cited per function
 * (paths relative to the reference root, line numbers as in SURVEY.md).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  All pointers are DEVICE pointers
 *     (HBM) unless the name ends in `_host`.  All tensors are dense row-major fp32 unless noted.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library allocates
 *     nothing and keeps no mutable global state => every entry point is hipGraph-capturable.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing synchronises.
 *   - return value: 0 on success, a positive hipError_t, or a negative NERFHIP_E* code.
 *     Nothing throws across the ABI.
 */
#ifndef NERFHIP_H
#define NERFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFHIP_ABI_VERSION 3     /* 2: nerfhip_mlp_pack_weights_bwd takes the biases (fold block of the W^T image); 3 (additive):
                                   * nerfhip_render_args.regen_enc, nerfhip_mlp_bwd_multi_rays                                   */

#define NERFHIP_E_BADARG (-1)  /* null pointer / non-positive size / unsupported shape */
#define NERFHIP_E_UNSUPPORTED (-2)
#define NERFHIP_E_ALIGN (-3)   /* pointer not aligned as documented */

/* MLP arithmetic type (`dtype` arguments). */
#define NERFHIP_F32 0  /* v_mfma_f32_32x32x2_f32, exact fp32 (parity configuration)      */
#define NERFHIP_BF16 1 /* v_mfma_f32_32x32x16_bf16, fp32 accumulate (roofline config)    */
#define NERFHIP_BF16_F8 2 /* as NERFHIP_BF16 (forward, dX chain: bf16 MFMA); the tensors saved for the weight-gradient
                           * GEMM (activations X, dY) are stored as block-scaled OCP e4m3 (one e8m0 scale per 32 points x
                           * 32 features) and consumed by v_mfma_scale_f32_32x32x64_f8f6f4: half the backward's HBM bytes.
                           * Inference entry points treat it as NERFHIP_BF16.                                          */

typedef void* nerfhip_stream_t;

int nerfhip_abi_version(void);
const char* nerfhip_error_string(int code);

/* ---- a2. Embedding.forward  (models/nerf.py:21-38) -------------------------------------
 * out[i, :] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(F-1) x), cos(2^(F-1) x)]
 * x (n,C) -> out (n, C*(2F+1)).  1 <= C <= 8, 0 <= F <= 16.                               */
int nerfhip_posenc(const float* x, float* out, int64_t n, int C, int n_freqs, nerfhip_stream_t stream);
/* backward of the above w.r.t. x: gx (n,C) = d<gout,out>/dx.                              */
int nerfhip_posenc_bwd(const float* x, const float* gout, float* gx, int64_t n, int C, int n_freqs,
                       nerfhip_stream_t stream);

/* The same with explicit frequency bands (bands: n_freqs DEVICE floats; NULL = the 2^k above): the reference's
 * `Embedding(..., logscale=False)` = torch.linspace(1, 2^(F-1), F) (nerf.py:16-19), passed as the module built them.   */
int nerfhip_posenc_bands(const float* x, const float* bands, float* out, int64_t n, int C, int n_freqs,
                         nerfhip_stream_t stream);
int nerfhip_posenc_bands_bwd(const float* x, const float* bands, const float* gout, float* gx, int64_t n, int C, int n_freqs,
                             nerfhip_stream_t stream);

/* ---- a5. coarse depth sampling  (models/rendering.py:183-204) --------------------------
 * rays (B,8)=[o d near far]; z (B,S): linear in depth or disparity; when perturb>0,
 * stratified jitter with `perturb_rand` (B,S) ~ U[0,1) (the caller's torch.rand draw).     */
int nerfhip_sample_coarse_z(const float* rays, const float* perturb_rand, float* z, int64_t B, int S,
                            int use_disp, float perturb, nerfhip_stream_t stream);

/* ---- a9. torchsearchsorted.searchsorted(a, v, side='right')  (rendering.py:2,42) --------
 * idx[r,k] = first j in [0,M] with a[r,j] > v[r,k]  (numpy side='right'); a (B,M), v (B,K),
 * idx int64 (B,K).  This is the one native extension the reference depends on.             */
int nerfhip_searchsorted_right(const float* a, const float* v, int64_t* idx, int64_t B, int M, int K,
                               nerfhip_stream_t stream);
/* side='left' (the extension's default; unused by the reference): first j with a[r,j] >= v[r,k]. */
int nerfhip_searchsorted_left(const float* a, const float* v, int64_t* idx, int64_t B, int M, int K,
                              nerfhip_stream_t stream);

/* ---- a8. sample_pdf  (models/rendering.py:14-55) ----------------------------------------
 * bins (B,M+1) with row stride `bins_stride`, weights (B,M) with row stride `w_stride`
 * (so that the reference's `weights_coarse[:, 1:-1]` view needs no copy).
 * u: NULL => deterministic linspace(0,1,K) (`det=True`); else (K) if u_stride==0 or (B,K).
 * samples (B,K).                                                                           */
int nerfhip_sample_pdf(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                       const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K, float eps,
                       nerfhip_stream_t stream);

/* Rounding of the pdf normaliser `torch.sum(weights, -1)` (rendering.py:30), on whose last bit the searchsorted indices of
 * rendering.py:42 have knife edges (u == 1.0, cdf ties):
 *   NERFHIP_ROW_TOTAL_EXACT  the correctly rounded fp32 sum (host-independent).  What the plain entry points without a row_total
 *                            argument — nerfhip_sample_pdf, nerfhip_fine_z — use.
 *   NERFHIP_ROW_TOTAL_ATEN   the fp32 additions in the order of ATen's CPU sum kernel (torch 2.x, every x86 capability:
 *                            8-float vectors, 4 interleaved accumulators, scalar tail) = the reference's own bits on CPU;
 *                            reproduces the (cdf, u) -> inds triples recorded at the reference's call site on every element.
 *                            What the Python operators (nerf_pl_amd.ops: sample_pdf_u, fine_z, render_rays, the fused training
 *                            step) pass BY DEFAULT through the entry points that take row_total (ops.set_row_total("exact") /
 *                            NERFHIP_ROW_TOTAL=exact selects the other): over the minted reference training runs the HIP fp32
 *                            path then follows the reference's PSNR@step with half the seed-to-seed scatter (oracle and parity). */
#define NERFHIP_ROW_TOTAL_EXACT 0
#define NERFHIP_ROW_TOTAL_ATEN 1

/* Same, additionally exporting what the fused kernel computed on the way (either may be NULL): cdf_out (B,M+1) = the
 * cdf of rendering.py:31-33 and inds_out (B,K) int64 = searchsorted(cdf, u, side='right') of rendering.py:42 — the
 * bit-exact contract of the path, checkable against the indices recorded at the reference's own call site.
 * row_total: NERFHIP_ROW_TOTAL_*.                                                                                   */
int nerfhip_sample_pdf_ex(const float* bins, int64_t bins_stride, const float* weights, int64_t w_stride,
                          const float* u, int64_t u_stride, float* samples, int64_t B, int M, int K, float eps,
                          float* cdf_out, int64_t* inds_out, int row_total, nerfhip_stream_t stream);

/* ---- a10. fine-pass depth assembly  (models/rendering.py:223-229) -----------------------
 * z_mid = midpoints(z_coarse); z_new = sample_pdf(z_mid, w_coarse[:,1:-1], N_i, u);
 * z_fine = sort(cat(z_coarse, z_new)).  One launch, one wave per ray.
 * z_coarse,w_coarse (B,S_c); u as above; z_fine (B,S_c+N_i); z_new (B,N_i) optional (NULL ok). */
int nerfhip_fine_z(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride,
                   float* z_fine, float* z_new, int64_t B, int S_c, int N_i, float eps,
                   nerfhip_stream_t stream);

/* Same with the optional exports and the row_total choice of nerfhip_sample_pdf_ex: cdf_out (B,S_c-1), inds_out (B,N_i) int64. */
int nerfhip_fine_z_ex(const float* z_coarse, const float* w_coarse, const float* u, int64_t u_stride, float* z_fine,
                      float* z_new, int64_t B, int S_c, int N_i, float eps, float* cdf_out, int64_t* inds_out,
                      int row_total, nerfhip_stream_t stream);

/* ---- a7. alpha compositing  (models/rendering.py:143-172) -------------------------------
 * raw: (B,S,4)=[r g b sigma] when raw_ch==4, or (B,S) sigma only when raw_ch==1 (weights_only).
 * noise: (B,S) standard-normal draws or NULL; sigma_eff = relu(sigma + noise*noise_std).
 * weights (B,S) always written; opacity (B) always; rgb (B,3) and depth (B) when raw_ch==4.  */
int nerfhip_composite_fwd(const float* raw, int raw_ch, const float* z, const float* rays,
                          const float* noise, float noise_std, int white_back, float* weights, float* rgb,
                          float* depth, float* opacity, int64_t B, int S, nerfhip_stream_t stream);
/* backward: given g_rgb (B,3) [, g_depth (B), g_opacity (B); NULL = 0] produce g_raw (B,S,raw_ch).
 * Recomputes alpha/T from raw,z,noise (nothing else is saved by forward).                   */
int nerfhip_composite_bwd(const float* raw, int raw_ch, const float* z, const float* rays,
                          const float* noise, float noise_std, int white_back, const float* g_rgb,
                          const float* g_depth, const float* g_opacity, const float* g_weights,
                          float* g_raw, int64_t B, int S, nerfhip_stream_t stream);

/* Training fast path (train.py:103-117 with losses.py:9-14): composite_fwd (raw_ch 4) + d MSE / d rgb of every ray against
 * `target` (B,3), i.e. g_rgb = (rgb - target) * grad_scale with grad_scale = 2 / (3 B) for one image's mean-squared error, +
 * composite_bwd for that g_rgb, in ONE launch.  Writes weights (B,S; NULL ok), rgb (B,3), depth (B), opacity (B) and
 * g_raw (B,S,4) = d loss / d raw — bit-identical to composite_fwd -> mse_psnr -> composite_bwd.                          */
int nerfhip_composite_train(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                            int white_back, const float* target, float grad_scale, float* weights, float* rgb, float* depth,
                            float* opacity, float* g_raw, int64_t B, int S, nerfhip_stream_t stream);

/* The coarse pass of a training step (rendering.py:143-172 + :223-229): nerfhip_composite_train and, for the same ray in the same
 * wave, nerfhip_fine_z_ex on its weights (u / u_stride / N_i / eps / z_fine / row_total as there) — the weights travel from the quadrature to the
 * inverse-CDF sampling through LDS (`weights` may be NULL).  Bit-identical to the two launches.                            */
int nerfhip_composite_train_fine_z(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                                   int white_back, const float* target, float grad_scale, float* weights, float* rgb, float* depth,
                                   float* opacity, float* g_raw, int64_t B, int S, const float* u, int64_t u_stride, int N_i,
                                   float eps, float* z_fine, int row_total, nerfhip_stream_t stream);
/* The last pass of a training step: nerfhip_composite_train + the loss values of nerfhip_mse_psnr (losses.py:9-14,
 * metrics.py:4-13) — out3 = [loss, psnr, mse] over this pass's rgb as the fine image and rgb_coarse (B,3; NULL when this pass is
 * the only one) as the coarse image, reduced by the last workgroup to finish in nerfhip_mse_psnr's own order (same bits).
 * ticket: one zero-initialised DEVICE word owned by the caller; the launch leaves it at zero.                              */
int nerfhip_composite_train_loss(const float* raw, const float* z, const float* rays, const float* noise, float noise_std,
                                 int white_back, const float* target, float grad_scale, float* weights, float* rgb, float* depth,
                                 float* opacity, float* g_raw, int64_t B, int S, const float* rgb_coarse, float* out3,
                                 uint32_t* ticket, nerfhip_stream_t stream);

/* ---- a3/a4. NeRF MLP  (models/nerf.py:42-124; D=8 W=256 skips=[4] in 63/27) -------------
 * Parameters are repacked once per weight update into the MFMA A-fragment stream the
 * kernel consumes (layout: DESIGN.md §3).  weights_host[i]/biases_host[i] are HOST arrays of
 * 12 DEVICE pointers in state_dict order: xyz_encoding_1..8, xyz_encoding_final,
 * dir_encoding, sigma, rgb (each weight (out,in) row-major fp32).                           */
size_t nerfhip_mlp_packed_bytes(int dtype);
int nerfhip_mlp_pack_weights(const float* const* weights_host, const float* const* biases_host, void* packed,
                             int dtype, nerfhip_stream_t stream);

/* Training: bytes of the saved-activation buffer the forward fills when `save_acts` != NULL
 * (every B-operand slab of every layer, in register/fragment order; 4.94 KB/point in bf16,
 * 9.9 KB/point in fp32).  The backward entry points consume it.               */
size_t nerfhip_mlp_act_bytes(int64_t n_points, int dtype);

/* NeRF.forward(x, sigma_only) on pre-embedded inputs (nerf.py:83-124):
 * x (n, 90 | 63) with row stride x_stride floats -> out (n,4)=[rgb sigma] | (n,1).
 * save_acts: NULL for inference; else a nerfhip_mlp_act_bytes(n,dtype) buffer (sigma_only must be 0). */
int nerfhip_mlp_fwd_embedded(const float* x, int64_t x_stride, int64_t n, const void* packed, float* out,
                             int sigma_only, int dtype, void* save_acts, nerfhip_stream_t stream);

/* Fused `inference` MLP loop (rendering.py:115-141 + 206-207 + nerf.py:21-38): points are
 * generated in-register as o + d*z, encoded (10 / 4 frequencies) and pushed through the MLP;
 * no (n,63)/(n,90) tensor and no repeat_interleave'd dir embedding ever exists in HBM.
 * rays (B,8), z (B,S) -> out (B,S,4) | (B,S) when sigma_only.                               */
int nerfhip_mlp_fwd_rays(const float* rays, const float* z, int64_t B, int S, const void* packed, float* out,
                         int sigma_only, int dtype, void* save_acts, nerfhip_stream_t stream);

/* The same for the COARSE pass with its depths formed in the kernel's prologue (rendering.py:183-204 = nerfhip_sample_coarse_z,
 * then :206-207): z (B,S) is an OUTPUT here — every point's depth is computed from its ray's bounds (and perturb_rand (B,S), the
 * caller's U[0,1) draw, when perturb > 0), used, and written for the compositing that follows.  Same bits as the two launches. */
int nerfhip_mlp_fwd_rays_coarse(const float* rays, const float* perturb_rand, float* z, int64_t B, int S, int use_disp,
                                float perturb, const void* packed, float* out, int sigma_only, int dtype, void* save_acts,
                                nerfhip_stream_t stream);

/* ---- backward of the MLP (autograd mirror of nerf.py:100-124 under train.py:103-117) -----------
 * Two hand-written phases: the register-resident chain in reverse with W^T streamed
 * (writes dL/d(pre-activation) slabs into `dys`), then dW = dY^T X with points as the MFMA K
 * dimension (split-K partial slabs in `dw_workspace`, reduced and un-permuted into the gradients).
 *   g_out (n,4)  dL/d[rgb sigma];  out (n,4) the forward's output (for sigmoid');
 *   packed_bwd   nerfhip_mlp_pack_weights_bwd() image of the CURRENT parameters: the W^T fragment stream of the chain, then an fp32
 *                snapshot of xyz_encoding_final's weight + bias and of dir_encoding's first 256 weight columns.  xyz_encoding_final
 *                has no activation (nerf.py:70,116): the forward does not save its output, the chain does not store its output
 *                gradient, the dW launch forms G = dY_dir^T h8 and a small fp32 launch (mlp_bwd_fold_kernel) finishes
 *                dW_dir[:, :256] = G W_f^T + db_dir b_f^T, dW_final = W_dir[:, :256]^T G, db_final = W_dir[:, :256]^T db_dir —
 *                the same sums, re-associated (csrc/mlp_layout.h kDwJobs);
 *   acts         buffer the forward filled (save_acts);  dys  nerfhip_mlp_dy_bytes() scratch;
 *   dw_workspace nerfhip_mlp_dw_workspace_bytes() scratch;
 *   grad_w_host / grad_b_host: HOST arrays of 12 DEVICE pointers (state_dict order, (out,in) fp32);
 *   accumulate != 0 adds into them, else overwrites.  No gradient flows to rays / z / x
 *   (the reference's sampled depths are detached, rendering.py:226; rays carry no grad).            */
size_t nerfhip_mlp_packed_bwd_bytes(int dtype);
int nerfhip_mlp_pack_weights_bwd(const float* const* weights_host, const float* const* biases_host, void* packed_bwd, int dtype,
                                 nerfhip_stream_t stream);
/* both images (nerfhip_mlp_pack_weights + nerfhip_mlp_pack_weights_bwd) in ONE launch: what a training step needs
 * of a model whose weights do not change between its forward and its backward                                   */
int nerfhip_mlp_pack_weights_train(const float* const* weights_host, const float* const* biases_host, void* packed,
                                   void* packed_bwd, int dtype, nerfhip_stream_t stream);
/* ... of `n_models` (<= 4) models in ONE launch: weights_host / biases_host hold n_models x 12 DEVICE pointers (model-major),
 * packed_host / packed_bwd_host one DEVICE buffer per model.  A training step packs its coarse and its fine network here. */
int nerfhip_mlp_pack_weights_train_multi(const float* const* weights_host, const float* const* biases_host,
                                         void* const* packed_host, void* const* packed_bwd_host, int n_models, int dtype,
                                         nerfhip_stream_t stream);
size_t nerfhip_mlp_dy_bytes(int64_t n_points, int dtype);
int nerfhip_mlp_dw_splits(int64_t n_points, int dtype);   /* total (job, point-split) workgroups = partial slabs */
size_t nerfhip_mlp_dw_workspace_bytes(int64_t n_points, int dtype);
int nerfhip_mlp_bwd(const float* g_out, const float* out, int64_t n, const void* packed_bwd, const void* acts,
                    void* dys, void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host,
                    int accumulate, int dtype, nerfhip_stream_t stream);

/* The same with a phase mask (measurement: bench.py times the three kernels of the backward separately with HIP events):
 * bit 0 = backward chain (writes dys), bit 1 = weight-gradient GEMM (reads acts + dys, writes dw_workspace),
 * bit 2 = reduce (dw_workspace -> gradients).  nerfhip_mlp_bwd == phases 7.                                          */
int nerfhip_mlp_bwd_phases(const float* g_out, const float* out, int64_t n, const void* packed_bwd, const void* acts,
                           void* dys, void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host,
                           int accumulate, int dtype, int phases, nerfhip_stream_t stream);

/* The backward of SEVERAL models (<= 2: a training step's fine and coarse network) with ONE weight-gradient launch and ONE
 * reduce launch for all of them: per model its chain kernel, then one dW GEMM whose workgroups are shared out over the 12 jobs
 * of every model in proportion to the models' points (equal ring iterations per workgroup), then one reduce.  All `_host`
 * arguments are HOST arrays with one entry per model (grad_w_host / grad_b_host: n_models x 12 DEVICE pointers, model-major);
 * dw_workspace holds nerfhip_mlp_dw_workspace_bytes_multi() bytes.  `adam` (NULL ok; requires accumulate == 0): apply the Adam
 * update of nerfhip_adam_step to every model's flat parameter buffer inside the reduce kernel — each model's parameters,
 * exp_avg, exp_avg_sq are flat fp32 buffers laid out like its flat gradient buffer grad_flat (the 24 gradient tensors
 * contiguous: w0..w11, b0..b11) — for single-GPU steps, where no all-reduce sits between gradients and update.
 * g_scale (NULL = 1): a DEVICE scalar every g_out is multiplied by inside the chain kernels — the upstream d L / d loss of an
 * autograd backward, applied without a scaling launch.                                                                   */
typedef struct nerfhip_adam_fused {
    int n_models;
    float* param[2];
    float* exp_avg[2];
    float* exp_avg_sq[2];
    const float* grad_flat[2];
    float* state;                 /* {step count, arrival ticket} as for nerfhip_adam_step */
    float lr, beta1, beta2, eps, weight_decay;
} nerfhip_adam_fused;
size_t nerfhip_mlp_dw_workspace_bytes_multi(const int64_t* n_points_host, int n_models, int dtype);
/* The split plan of that launch (host logic only): splits_out[12 m + j] = workgroups of weight-gradient job j (the 12 parameter
 * tensors in state_dict order, reference nerf.py:42-81) of model m; stage_kib_out (NULL ok) = KiB one ring iteration of the job
 * reads.  bf16: workgroups are shared out so that iterations x (fixed cost + cost per KiB) is equal across jobs and models.
 * Returns the number of workgroups (= partial slabs), < 0 on bad arguments.                                                      */
int nerfhip_mlp_dw_plan(const int64_t* n_points_host, int n_models, int dtype, int* splits_out, int* stage_kib_out);
int nerfhip_mlp_bwd_multi(int n_models, const float* const* g_out_host, const float* const* out_host, const int64_t* n_host,
                          const void* const* packed_bwd_host, const void* const* acts_host, void* const* dys_host,
                          void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host, int accumulate, int dtype,
                          int phases, const float* g_scale, const nerfhip_adam_fused* adam, nerfhip_stream_t stream);

/* nerfhip_mlp_bwd_multi for models evaluated ON RAYS (nerfhip_render_train_fwd with regen_enc = 1; NERFHIP_BF16): the
 * weight-gradient launch does not read the positional encodings embedding_xyz(o + d z) / embedding_dir(d) (nerf.py:21-38,
 * rendering.py:186,206-207) from the saved activations — they are the X operand of xyz_encoding_1, of the skip layer and of
 * dir_encoding: 10 of the 282 KiB a 32-point tile is read for — but forms them again from 4 B of depth per point with the forward's
 * own arithmetic (bit-identical operands, so bit-identical gradients).  enc (NULL = nerfhip_mlp_bwd_multi): per model the rays
 * (B,8), the depths z (B,S) the forward evaluated the model at, and S; needs S % 32 == 0 and n = B S a multiple of 256.
 * Measured on MI355X (round 6, same box, alternating): the forward gains 8-10 us of the 1024 x (64+128) step, this launch loses
 * 12-25 — its ring iterations are paced by their instructions, not their bytes — so the Python step leaves it off by default. */
typedef struct nerfhip_enc_source {
    const float* rays[2];
    const float* z[2];
    int S[2];
} nerfhip_enc_source;
int nerfhip_mlp_bwd_multi_rays(int n_models, const float* const* g_out_host, const float* const* out_host, const int64_t* n_host,
                               const void* const* packed_bwd_host, const void* const* acts_host, void* const* dys_host,
                               void* dw_workspace, float* const* grad_w_host, float* const* grad_b_host, int accumulate, int dtype,
                               int phases, const float* g_scale, const nerfhip_adam_fused* adam, const nerfhip_enc_source* enc,
                               nerfhip_stream_t stream);

/* d loss / d x of NeRF.forward on pre-embedded inputs (nerf.py:100-124 is differentiable w.r.t. x): from the dY slabs a
 * nerfhip_mlp_bwd call left in `dys`,  gx[:, 0:63] = W_1^T dY_1 + W_5[:, :63]^T dY_5,  gx[:, 63:90] = W_dir[:, 256:]^T dY_dir.
 * w_xyz1 / w_xyz5 / w_dir: the (out,in) fp32 weights of xyz_encoding_1, xyz_encoding_5, dir_encoding; gx (n, >= 90) with
 * row stride gx_stride floats.  dtype NERFHIP_F32 | NERFHIP_BF16 (NERFHIP_BF16_F8 stores dY too coarsely: unsupported).   */
int nerfhip_mlp_dx_embedded(const void* dys, int64_t n, const float* w_xyz1, const float* w_xyz5, const float* w_dir,
                            float* gx, int64_t gx_stride, int dtype, nerfhip_stream_t stream);

/* ---- a3/a4 for a NON-default NeRF(D, W, in_channels_xyz, in_channels_dir, skips)  (models/nerf.py:42-124) ----
 * The fused kernels above are built for the default shape.  Any other shape runs layer by layer through one MFMA GEMM
 * kernel: activations (n, features) fp32 row-major with a row stride (ld*, in floats; unit column stride), weights (out, in)
 * fp32 row-major with row stride ldw (so a column block of a weight — the two halves of a skip / direction concat,
 * nerf.py:108-109,118 — is addressed by pointer offset + ldw).  dtype: NERFHIP_F32 exact fp32 MFMA; NERFHIP_BF16 and
 * NERFHIP_BF16_F8: operands rounded to bf16, fp32 accumulation.
 *   fwd         y[n, n_out] = act( x[n, n_in] . w[n_out, n_in]^T (+ y when accumulate) (+ bias when non-NULL) )
 *   bwd_input   gx[n, n_in] (+)= (gy * act'(y))[n, n_out] . w[n_out, n_in]        act' from the layer OUTPUT y (ReLU: y > 0;
 *                                                                                 Sigmoid: y (1 - y)); y unused for ACT_NONE
 *   bwd_weight  gw[n_out, n_in] (+)= (gy * act'(y))^T . x;  gb[n_out] (+)= column sums (gb NULL: skipped).  Split over the
 *               points into `workspace` (nerfhip_linear_bwd_weight_workspace_bytes) and reduced in a fixed order.          */
#define NERFHIP_ACT_NONE 0
#define NERFHIP_ACT_RELU 1
#define NERFHIP_ACT_SIGMOID 2
int nerfhip_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                       int64_t n, int n_in, int n_out, int act, int accumulate, int dtype, nerfhip_stream_t stream);
int nerfhip_linear_bwd_input(const float* gy, int64_t ldgy, const float* y, int64_t ldy, int act, const float* w, int64_t ldw,
                             float* gx, int64_t ldgx, int64_t n, int n_in, int n_out, int accumulate, int dtype,
                             nerfhip_stream_t stream);
size_t nerfhip_linear_bwd_weight_workspace_bytes(int64_t n, int n_in, int n_out);
int nerfhip_linear_bwd_weight(const float* gy, int64_t ldgy, const float* y, int64_t ldy, int act, const float* x, int64_t ldx,
                              float* gw, int64_t ldgw, float* gb, void* workspace, int64_t n, int n_in, int n_out,
                              int accumulate, int dtype, nerfhip_stream_t stream);

/* ---- N2. MSELoss.forward + psnr + backward seed  (losses.py:9-14, metrics.py:4-13, train.py:103-117) ----
 * rgb_coarse, rgb_fine (NULL when N_importance == 0), target: n = 3*rays floats each.
 * out3 = [loss, psnr of the fine (else coarse) image, its mse];  g_coarse / g_fine (NULL ok) receive
 * d loss / d rgb = 2 (rgb - target) / n.  One launch, deterministic reduction order.                        */
int nerfhip_mse_psnr(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n, float* out3,
                     float* g_coarse, float* g_fine, nerfhip_stream_t stream);

/* ---- N2. Adam on flat parameter storage  (utils/__init__.py:10-30 -> torch.optim.Adam(lr, eps=1e-8, weight_decay)) ----
 * One launch over `n_tensors` (<= 8) flat fp32 tensors: params/grads/exp_avg/exp_avg_sq are HOST arrays of DEVICE
 * pointers, numel_host their element counts.  `state` is a 2-float DEVICE buffer {step count, arrival ticket},
 * zero-initialised by the caller; the kernel uses t = state[0] + 1 for the bias corrections and advances state[0]
 * itself (device-resident step => hipGraph replays keep counting).  Non-amsgrad, L2 weight decay (g += wd * p).      */
int nerfhip_adam_step(float* const* params_host, const float* const* grads_host, float* const* exp_avg_host,
                      float* const* exp_avg_sq_host, const int64_t* numel_host, int n_tensors, float* state, float lr,
                      float beta1, float beta2, float eps, float weight_decay, nerfhip_stream_t stream);

/* ---- N1. ray generation  (datasets/ray_utils.py:5-94; consumers blender.py:37-69, llff.py:236-253) --------
 * get_ray_directions: dirs (H,W,3) = ((i-W/2)/focal, -(j-H/2)/focal, -1), i = column, j = row.
 * get_rays: rays_d = normalise(directions @ c2w[:, :3].T), rays_o = c2w[:, 3]; c2w (3,4) row-major DEVICE array.
 * get_ndc_rays: shift to the near plane and project (forward-facing LLFF scenes).  `focal` is a double: the
 * reference forms -1/(W/(2 focal)) in Python double precision before it meets the fp32 tensors.                */
int nerfhip_ray_directions(float* dirs, int H, int W, double focal, nerfhip_stream_t stream);
int nerfhip_get_rays(const float* directions, const float* c2w, float* rays_o, float* rays_d, int64_t n,
                     nerfhip_stream_t stream);
int nerfhip_ndc_rays(int H, int W, double focal, float near, const float* rays_o, const float* rays_d, float* out_o,
                     float* out_d, int64_t n, nerfhip_stream_t stream);
/* All of the above fused, for a batch of pixels: rays (n,8) = [o d near far] of global pixel ids
 * `pixel_ids[r]` = image*H*W + row*W + col (or first_pixel + r when pixel_ids is NULL) under poses
 * c2w (n_images,3,4); use_ndc applies get_ndc_rays with the near plane at ndc_near_plane (llff.py uses 1.0). */
int nerfhip_gen_rays(const float* c2w, const int64_t* pixel_ids, int64_t first_pixel, int64_t n, int H, int W,
                     double focal, float near, float far, int use_ndc, float ndc_near_plane, float* rays,
                     nerfhip_stream_t stream);
/* A training batch in one launch (the reference's Dataset.__getitem__ + DataLoader collate, blender.py:81-84 /
 * train.py:89-94): rays as nerfhip_gen_rays for the n pixel ids, and rgbs (n,3) = rgbs_all[pixel_ids] gathered from the
 * device-resident pixel colours rgbs_all (n_images*H*W, 3).                                                      */
int nerfhip_sample_batch(const float* c2w, const int64_t* pixel_ids, const float* rgbs_all, int64_t n, int H, int W,
                         double focal, float near, float far, int use_ndc, float ndc_near_plane, float* rays, float* rgbs,
                         nerfhip_stream_t stream);

/* ---- the random draws of a training step, as torch's generator would make them  (rendering.py:203,152,39,152; the
 * DataLoader's batch of train.py:89-94) ----
 * ONE launch producing up to 6 tensors from the Philox4x32-10 stream that torch.rand / torch.randn / torch.randint on the GPU
 * consume: draw i is what the i-th of those torch calls would return for a generator at (seed, offset), bit for bit, and
 * *increment_host is what the calls together advance the generator's offset by (the caller then sets the generator to
 * offset + increment: same seed => same training run, whichever path draws).  max_blocks = multiProcessorCount *
 * (maxThreadsPerMultiProcessor / 256) of the device (ATen's launch cap; 2048 on MI355X).
 *   kind UNIFORM: out = numel floats of torch.rand; NORMAL: torch.randn; RANDINT: numel int64 of torch.randint(0, range)
 *   (range <= 2^62).  A draw with out == NULL (and no batch) is one nobody reads — the reference always draws its noise tensors,
 *   also when noise_std == 0 (rendering.py:152): the stream moves past it exactly as the torch call would, nothing is computed.
 * batch_host (NULL ok): draws_host[0] must then be RANDINT over the store's pixel ids; the drawn ids are turned into the
 * training batch in the same launch — rays (numel,8) and rgbs (numel,3) exactly as nerfhip_sample_batch — and draws_host[0].out
 * may be NULL (the ids themselves are then not stored).
 * state (NULL ok): a DEVICE buffer of 4 x uint64 {seed, offset, arrival ticket (0), -}.  When given, seed / offset are read
 * from it instead of the arguments and its offset is advanced by the increment at the end of the launch, so that hipGraph
 * replays of one captured call keep walking the stream.                                                                  */
#define NERFHIP_DRAW_UNIFORM 0
#define NERFHIP_DRAW_NORMAL 1
#define NERFHIP_DRAW_RANDINT 2
typedef struct nerfhip_draw {
    int kind;
    int64_t numel;
    void* out;
    uint64_t range;
} nerfhip_draw;
typedef struct nerfhip_ray_batch {
    const float* c2w;       /* (n_images,3,4) poses */
    const float* rgbs_all;  /* (n_images*H*W,3) pixel colours, or NULL (then rgbs NULL too) */
    float* rays;            /* (numel,8) out, 16-byte aligned */
    float* rgbs;            /* (numel,3) out */
    int H, W;
    double focal;
    float near, far;
    int use_ndc;
    float ndc_near_plane;
} nerfhip_ray_batch;
uint64_t nerfhip_torch_draw_increment(int64_t numel, int max_blocks);
int nerfhip_torch_draws(const nerfhip_draw* draws_host, int n_draws, const nerfhip_ray_batch* batch_host, uint64_t seed,
                        uint64_t offset, uint64_t* state, int max_blocks, uint64_t* increment_host, nerfhip_stream_t stream);

/* The prologue of a training step in ONE launch: nerfhip_torch_draws (same arguments: the batch, its rays, the step's draws) and
 * nerfhip_mlp_pack_weights_train_multi (same arguments: both images of n_models models) — the two jobs of the step that depend
 * only on the generator state and on the parameters.  Same device code as the two launches, same results.                     */
int nerfhip_train_prologue(const nerfhip_draw* draws_host, int n_draws, const nerfhip_ray_batch* batch_host, uint64_t seed,
                           uint64_t offset, uint64_t* state, int max_blocks, uint64_t* increment_host,
                           const float* const* weights_host, const float* const* biases_host, void* const* packed_host,
                           void* const* packed_bwd_host, int n_models, int dtype, nerfhip_stream_t stream);

/* ---- render_rays in ONE launch  (models/rendering.py:58-244; SURVEY 8b "_render_fwd fused: 32 B in + 40 B out per ray") ----
 * The whole pipeline of one `render_rays` call — coarse depths (:183-204), coarse MLP over o + d z (:206-217), compositing
 * (:143-172), sample_pdf + sort (:223-229), fine MLP, compositing — by workgroups that own 4 whole rays each: the MLP runs as
 * sub-passes of nerfhip_mlp_fwd_rays' own network code, the rays' compositing and fine depths by the same workgroup in between
 * (same device code, hence the same bits, as nerfhip_mlp_fwd_rays_coarse -> nerfhip_composite_fwd -> nerfhip_fine_z_ex ->
 * nerfhip_mlp_fwd_rays -> nerfhip_composite_fwd).  The coarse pass always evaluates the full network (`test_time`'s
 * sigma-only shortcut changes no value: pass rgb_coarse = depth_coarse = NULL to drop what the reference does not return).
 * The per-point intermediates z_* / raw_* are caller-owned buffers (outputs for whoever wants them; they stay in L2 between the
 * sub-passes that write and read them).
 * Shapes the kernels take (nerfhip_render_supported): B % 4 == 0, 4 S_c and 4 (S_c + N_i) multiples of the points per
 * sub-pass (256 for bf16, 128 for fp32), S_c >= 3; N_i == 0 renders the coarse pass only.  z_* / raw_* / g_raw_* must be
 * 128-byte aligned (a group's slice of each is then whole cache lines), packed_* 16-byte aligned.                            */
typedef struct nerfhip_render_args {
    const float* rays;          /* (B,8) */
    int64_t B;
    int S_c, N_i;               /* N_samples, N_importance */
    const void* packed_coarse;  /* nerfhip_mlp_pack_weights images of the two models (fine unused when N_i == 0) */
    const void* packed_fine;
    float* z_coarse;            /* (B,S_c)        out */
    float* raw_coarse;          /* (B,S_c,4)      out: [r g b sigma] per coarse point */
    float* z_fine;              /* (B,S_c+N_i)    out */
    float* raw_fine;            /* (B,S_c+N_i,4)  out */
    void* save_coarse;          /* training: nerfhip_mlp_act_bytes(B S_c) / (B (S_c+N_i)) bytes of saved activations */
    void* save_fine;
    const float* perturb_rand;  /* (B,S_c) uniforms when perturb > 0, else NULL                    rendering.py:203 */
    float perturb;
    int use_disp;
    const float* noise_coarse;  /* (B,S_c) / (B,S_c+N_i) standard-normal draws, read when noise_std != 0      :152 */
    const float* noise_fine;
    float noise_std;
    int white_back;
    const float* u;             /* (B,N_i) uniforms (row stride u_stride), or NULL = deterministic linspace    :36-39 */
    int64_t u_stride;
    float eps;                  /* sample_pdf's eps (1e-5) */
    int row_total;              /* NERFHIP_ROW_TOTAL_* */
    float* rgb_coarse;          /* (B,3)  NULL ok */
    float* depth_coarse;        /* (B)    NULL ok */
    float* opacity_coarse;      /* (B) */
    float* rgb_fine;            /* (B,3) (B) (B); unused when N_i == 0 */
    float* depth_fine;
    float* opacity_fine;
    /* training forward only */
    const float* target;        /* (B,3) */
    float grad_scale;           /* 2 / (3 B) */
    float* g_raw_coarse;        /* (B,S_c,4)      out: d loss / d raw */
    float* g_raw_fine;          /* (B,S_c+N_i,4)  out */
    float* out3;                /* [loss, psnr, mse] */
    uint32_t* ticket;           /* one zero-initialised device word owned by the caller; left at zero */
    int regen_enc;              /* nerfhip_render_train_fwd, NERFHIP_BF16 only (ignored otherwise): 1 = the input-encoding slabs (6 of a
                                 * tile block's 151 KiB) are NOT saved; the backward must then be nerfhip_mlp_bwd_multi_rays with
                                 * these rays and z_coarse / z_fine, which forms them again                                     */
} nerfhip_render_args;
/* 1 when the single-launch kernels take this shape and arithmetic (dtype NERFHIP_F32 / NERFHIP_BF16 / NERFHIP_BF16_F8), else 0 */
int nerfhip_render_supported(int64_t B, int S_c, int N_i, int dtype);
/* inference: rgb / depth / opacity of both passes (the training-only fields are ignored) */
int nerfhip_render_fwd(const nerfhip_render_args* args_host, int dtype, nerfhip_stream_t stream);
/* inference under test_time (rendering.py:209-213; what eval.py:69-79 asks for): the coarse network stops at its density head
 * (nerf.py:112-114 sigma_only), so raw_coarse is (B,S_c) — sigma alone — and the coarse pass leaves opacity_coarse only
 * (rgb_coarse / depth_coarse are ignored); N_i > 0.  Bit-identical to nerfhip_mlp_fwd_rays_coarse(sigma_only) ->
 * nerfhip_composite_fwd -> nerfhip_fine_z -> nerfhip_mlp_fwd_rays -> nerfhip_composite_fwd.                                    */
int nerfhip_render_test_fwd(const nerfhip_render_args* args_host, int dtype, nerfhip_stream_t stream);
/* The forward of a training step (train.py:103-117 up to the loss) in ONE launch: the above with the activations saved for
 * nerfhip_mlp_bwd_multi, and per pass what nerfhip_composite_train_fine_z / nerfhip_composite_train_loss append to the
 * quadrature — d MSE / d rgb, the compositing backward (g_raw_*), loss / PSNR / MSE (out3, reduced by the last workgroup to
 * finish in nerfhip_mse_psnr's own order).  Bit-identical to those launches.                                                 */
int nerfhip_render_train_fwd(const nerfhip_render_args* args_host, int dtype, nerfhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFHIP_H */
