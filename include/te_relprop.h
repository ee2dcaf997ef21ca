/*
 * te_relprop.h -- C ABI of libte_relprop.so: the MI355X (gfx950) relevance-propagation hot path of
 * hila-chefer/Transformer-Explainability ("transformer_attribution": LRP relprop rules ->
 * gradient x attention head-mean -> rollout).
 *
 * Drop-in boundary.  The reference is pure Python; every entry point below replaces the body of one
 * `relprop` method (or one tail function) that the reference expresses as torch.autograd.grad on a
 * re-built micro-graph.  The reference-side binding is a ctypes stub; the
 * shipped host side is transformer-explainability_amd/ (same class / method names as the reference).
 *
 * Conventions
 *   - all tensors are fp32, row-major, device (HBM) pointers; shapes are int64 element counts,
 *     strides are in ELEMENTS; the innermost dimension is always contiguous.
 *   - BATCH SEMANTICS: the reference is batch-1 only (ViT_explanation_generator.py:31-32); a batch
 *     of B samples here means B independent batch-1 problems -- every "whole tensor" reduction of
 *     the reference (Add.relprop sums, modules/layers_ours.py:109-116) is taken per sample.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises,
 *     allocates, frees, or keeps a pointer after it returns; the library has no global state.
 *   - scratch memory is caller-provided: query te_<op>_workspace_bytes(...) and pass a device
 *     buffer of at least that size (16-byte aligned).  Ops without a query need none.
 *   - return value: 0 = TE_OK; negative = TE_ERR_* below; positive = a hipError_t from the launch.
 *   - `variant`: low byte TE_VARIANT_OURS (modules/layers_ours.py) or TE_VARIANT_LRP
 *     (modules/layers_lrp.py); OR-in TE_IMPL_SIMPLE to force the simple (non-MFMA) device kernels,
 *     which tests use as an on-device cross-check of the tiled kernels.
 *   - safe_divide(a,b) = a / (b + 1e-9 [==0 -> 1e-9]) * (b != 0)     modules/layers_ours.py:10-13
 */
#ifndef TE_RELPROP_H
#define TE_RELPROP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* te_stream_t; /* hipStream_t */

enum {
  TE_OK = 0,
  TE_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, bad variant */
  TE_ERR_WORKSPACE = -2,   /* ws == NULL or ws_bytes too small */
  TE_ERR_UNSUPPORTED = -3, /* shape outside what the kernels implement */
  TE_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
};

enum { TE_VARIANT_OURS = 0, TE_VARIANT_LRP = 1, TE_IMPL_SIMPLE = 0x100 };

enum { TE_ROLLOUT_NORMALISE = 1, TE_ROLLOUT_CLS_FIXUP = 2, TE_ROLLOUT_ROW0 = 4 };

/* library version (major*10000 + minor*100 + patch) and status strings */
int te_version(void);
const char* te_status_string(int status);
/* 0 if device 0..n-1 contains a gfx950 agent usable by this library, TE_ERR_NO_DEVICE otherwise */
int te_device_check(void);
/* 0 for the shipped library.  Bit 0: built with -DTE_X6_STUDY (measurement builds: TE_X6_STAGES_3 / TE_X6_KSPLIT schedules
 * and the main-loop ablations are compiled in; the shipped library's x6 entry points answer TE_ERR_UNSUPPORTED to those
 * flags).  Bit 1: built with -DTE_STUDY (getenv switches and study variants of the attention, fp32-MFMA and GELU-plane
 * kernels compiled in). */
int te_x6_study_build(void);
/* Provenance of this binary: "<16 hex>-<8 hex>" = sha256 over csrc/{*.hip,*.h} + include/{*.h} (names and contents,
 * sorted) and over the compiler flags, baked in by build.py.  The Python loader recomputes the first half from the tree it
 * was imported from and refuses a library built from other sources; bench.py and smoke() print it. */
const char* te_build_id(void);

/* ---- a3  Linear.relprop -------------------------------------------------------------------
 * replaces modules/layers_ours.py:207-230 (ours) and modules/layers_lrp.py:188-211 (lrp), and the
 * BERT copies BERT_explainability/modules/layers_ours.py:219-242.
 *   R [T,out_f], X [T,in_f], W [out_f,in_f] -> out [T,in_f]          (T = B*N rows, bias unused)
 *   ours: Z = X+ W+^T + X- W-^T ; S = sd(R,Z) ; act = X+ .(S W+) + X- .(S W-)
 *   lrp : S1 = sd(R, X+ W+^T) ; S2 = sd(R, X- W-^T) ; act = X+ .(S1 W+) + X- .(S2 W-)
 *   out = alpha*act - (alpha-1)*inh, inh = the same with W+ and W- exchanged (skipped at alpha==1).
 */
size_t te_linear_relprop_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f, int variant);
int te_linear_relprop_f32(const float* R, const float* X, const float* W, float* out,
                          int64_t T, int64_t in_f, int64_t out_f, float alpha, int variant,
                          void* ws, size_t ws_bytes, te_stream_t stream);

/* The two kernels of te_linear_relprop_f32 (variant ours, alpha = 1, in_f and out_f multiples of 4,
 * 16-byte aligned pointers) as separate launches:
 *   zpass: S [T,out_f] = sd(R, X+ W+^T + X- W-^T)      (layers_ours.py:216-219)
 *   cpass: out [T,in_f] = X+ .(S W+) + X- .(S W-)       (layers_ours.py:220-223,225)
 * Callers that want to time or overlap a single kernel launch use these; results are identical to
 * the composed call. */
int te_linear_zpass_f32(const float* R, const float* X, const float* W, float* S,
                        int64_t T, int64_t in_f, int64_t out_f, te_stream_t stream);
int te_linear_cpass_f32(const float* S, const float* X, const float* W, float* out,
                        int64_t T, int64_t in_f, int64_t out_f, te_stream_t stream);

/* Linear.relprop (variant ours, alpha = 1) using the FORWARD OUTPUT the rule module already caches
 * (forward_hook: `self.Y = output`, modules/layers_ours.py:16-27):  Y [T,out_f] = X W^T + bias  (bias may be
 * NULL).  Same-sign products sum to Z = X+ W+^T + X- W-^T and opposite-sign ones to X W^T - Z, while |X| |W|^T is
 * their difference, so   Z = ((Y - bias) + |X| |W|^T) / 2   needs ONE product instead of two (25 % fewer FLOPs per
 * Linear.relprop).  Elements where the halves cancel (Z < 2^-7 |X||W|^T) are recomputed as the plain positive-part
 * sum, so zero rows and near-zero Z behave as in layers_ours.py:216-219.  in_f, out_f multiples of 4 and 16-byte
 * aligned pointers (else TE_ERR_UNSUPPORTED: use te_linear_relprop_f32).  Workspace: te_linear_relprop_workspace_bytes
 * (T, in_f, out_f, TE_VARIANT_OURS).
 *   zpass_fwd: S [T,out_f] = sd(R, Z)          relprop_fwd: zpass_fwd, then te_linear_cpass_f32 */
int te_linear_zpass_fwd_f32(const float* R, const float* X, const float* W, const float* Y, const float* bias,
                            float* S, int64_t T, int64_t in_f, int64_t out_f, te_stream_t stream);
int te_linear_relprop_fwd_f32(const float* R, const float* X, const float* W, const float* Y, const float* bias,
                              float* out, int64_t T, int64_t in_f, int64_t out_f,
                              void* ws, size_t ws_bytes, te_stream_t stream);

/* The same with a DEFERRED per-sample factor on the relevance operand: row t of R enters the rule as
 * R[t,:] * r_scale[(t / rows_per_sample) * r_scale_stride] (the fp32 product Add.relprop's rescale would have stored,
 * layers_ours.py:117-118) -- the consumer side of te_add_relprop_deferred_f32.  r_scale == NULL: plain R. */
int te_linear_zpass_fwd_scaled_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                   int64_t rows_per_sample, const float* X, const float* W, const float* Y,
                                   const float* bias, float* S, int64_t T, int64_t in_f, int64_t out_f,
                                   te_stream_t stream);
int te_linear_relprop_fwd_scaled_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                     int64_t rows_per_sample, const float* X, const float* W, const float* Y,
                                     const float* bias, float* out, int64_t T, int64_t in_f, int64_t out_f,
                                     void* ws, size_t ws_bytes, te_stream_t stream);

/* ---- a4  einsum / MatMul relprop (RelPropSimple) ---------------------------------------------
 * replaces modules/layers_ours.py:48-60,122-127 and BERT_explainability/modules/layers_ours.py:89-91
 * for the two products of self-attention.  Element (b,h,n,d) of a strided operand T lives at
 * T + b*sb + h*sh + n*sn + d, so q/k/v can be read in place from the fused qkv activation
 * [B,N,3*H*D] (ViT_LRP.py:135) or from separate [B,N,H*D] tensors (BERT.py:330-336), and the
 * outputs can be written straight into the concatenated 'b n (qkv h d)' layout (ViT_LRP.py:175).
 * attn / cam_attn / R_nn are contiguous [B,H,N,N].  Both outputs are multiplied by out_scale
 * (callers pass 0.5: ViT_LRP.py:161-162,172-173; BERT.py:373-374,392-393).
 *
 * AV  (einsum 'bhij,bhjd->bhid'; BERT MatMul([probs, V])):
 *   Z = attn v ; S = sd(R,Z) ; cam_attn = attn .(S v^T) ; cam_v = v .(attn^T S)
 * QK  (einsum 'bhid,bhjd->bhij'; BERT MatMul([Q, K^T])), Z uses the UNSCALED q k^T:
 *   Z = q k^T ; S = sd(R_nn,Z) ; cam_q = q .(S k) ; cam_k = k .(S^T q)
 */
size_t te_matmul_relprop_av_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t D);
int te_matmul_relprop_av_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                             const float* attn,
                             const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                             float* cam_attn,
                             float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                             int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                             void* ws, size_t ws_bytes, te_stream_t stream);
size_t te_matmul_relprop_qk_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t D);
/* The same two rules with Z supplied by the caller: Z is the FORWARD output of the product whose rule is evaluated
 * (einsum / MatMul modules cache it as self.Y, forward_hook layers_ours.py:16-27; contiguous [B,H,N,D] for AV,
 * [B,H,N,N] = unscaled q k^T for QK).  The reference's autograd re-evaluates that product and gets the same bits, so
 * the rule needs two products instead of three and S matches the forward pass exactly.  Z == NULL computes it. */
int te_matmul_relprop_av_fwd_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                                 const float* attn,
                                 const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                 const float* Z,
                                 float* cam_attn,
                                 float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                                 int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                                 void* ws, size_t ws_bytes, te_stream_t stream);
/* te_matmul_relprop_av_fwd_f32 with a STRIDED Z (element (b,h,n,d) at Z + b*z_sb + h*z_sh + n*z_sn + d): the forward
 * product attn v usually lives only as the 'b n (h d)' activation that feeds the output projection (ViT_LRP.py:148);
 * z_sb = N*C, z_sh = D, z_sn = C reads it in place.  One-pass kernels only (head dim 64): TE_ERR_UNSUPPORTED otherwise
 * (callers then pass a contiguous copy to te_matmul_relprop_av_fwd_f32). */
int te_matmul_relprop_av_fwdz_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                                  const float* attn,
                                  const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                  const float* Z, int64_t z_sb, int64_t z_sh, int64_t z_sn,
                                  float* cam_attn,
                                  float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                                  int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                                  void* ws, size_t ws_bytes, te_stream_t stream);
int te_matmul_relprop_qk_fwd_f32(const float* R_nn,
                                 const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                 const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                 const float* Z,
                                 float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
                                 float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                                 int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                                 void* ws, size_t ws_bytes, te_stream_t stream);
/* te_matmul_relprop_qk_fwd_f32 whose relevance operand carries a DEFERRED per-sample factor: sample b's R_nn enters as
 * R_nn[b] * r_scale[b * r_scale_stride] -- the consumer side of te_add_bcast_relprop_deferred_f32 (BERT.py:386-393).
 * r_scale == NULL: plain.  One-pass kernels only (head dim 64): TE_ERR_UNSUPPORTED otherwise. */
int te_matmul_relprop_qk_fwd_scaled_f32(const float* R_nn, const float* r_scale, int64_t r_scale_stride,
                                        const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                        const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                        const float* Z,
                                        float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
                                        float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                                        int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                                        void* ws, size_t ws_bytes, te_stream_t stream);
int te_matmul_relprop_qk_f32(const float* R_nn,
                             const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                             const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                             float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
                             float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                             int64_t B, int64_t H, int64_t N, int64_t D, float out_scale, int variant,
                             void* ws, size_t ws_bytes, te_stream_t stream);

/* ---- a5  Add.relprop ---------------------------------------------------------------------------
 * replaces modules/layers_ours.py:97-120 (ours: three per-sample sums + rescale) and
 * modules/layers_lrp.py:98-100 (lrp: plain RelPropSimple).  R, X0, out0, out1 are [B,n]; X1 is [B,n]
 * (x1_batch_stride = n) or shared by all samples (x1_batch_stride = 0, e.g. pos_embed).
 *   S = sd(R, X0+X1) ; a = X0.S ; b = X1.S ;
 *   ours: a *= sd(sd(|Sa|,|Sa|+|Sb|)*SR, Sa) ; b *= sd(sd(|Sb|,|Sa|+|Sb|)*SR, Sb)   (per sample)
 */
size_t te_add_relprop_workspace_bytes(int64_t B, int64_t n);
int te_add_relprop_f32(const float* R, const float* X0, const float* X1, float* out0, float* out1,
                       int64_t B, int64_t n, int64_t x1_batch_stride, int variant,
                       void* ws, size_t ws_bytes, te_stream_t stream);

/* Add.relprop, variant ours, with the per-sample rescale DEFERRED to the consumers: ONE streaming pass (the rule's
 * algorithmic 5 n floats per sample) writes a = X0.S and b = X1.S unscaled and fac [B,2] = {fa, fb} per sample, where
 * te_add_relprop_f32 would have stored out0 = a*fa, out1 = b*fb (layers_ours.py:117-118).  The consumers take the
 * factor with the operand and form the identical product in registers: te_clone_relprop_scaled_f32,
 * te_linear_relprop_fwd_scaled_f32.  Workspace: te_add_relprop_deferred_workspace_bytes. */
size_t te_add_relprop_deferred_workspace_bytes(int64_t B, int64_t n);
int te_add_relprop_deferred_f32(const float* R, const float* X0, const float* X1, float* a, float* b, float* fac,
                                int64_t B, int64_t n, int64_t x1_batch_stride,
                                void* ws, size_t ws_bytes, te_stream_t stream);

/* Broadcast-operand Add of BERT self-attention (BERT.py:342,386-388): X0 = scaled scores
 * [B,H,N,N], X1 = extended mask [B,1,1,N] given as mask [B,N].  out0 [B,H,N,N] (relevance of the
 * scores); out1 [B,N] (relevance of the mask, discarded by the reference) may be NULL.
 *   S = sd(R, X0 + mask_j) ; a = X0.S ; C1_j = sum_{h,i} S ; b_j = mask_j C1_j ; rescale as above.
 */
size_t te_add_bcast_relprop_workspace_bytes(int64_t B, int64_t H, int64_t N);
int te_add_bcast_relprop_f32(const float* R, const float* X0, const float* mask, float* out0,
                             float* out1, int64_t B, int64_t H, int64_t N, int variant,
                             void* ws, size_t ws_bytes, te_stream_t stream);

/* The same rule (variant ours) with the rescale deferred: ONE pass over R and X0 writes a = X0.S unscaled and
 * fac [B,2] = {fa, fb}; out1 [B,N] (optional) is written scaled.  a * fa is bitwise te_add_bcast_relprop_f32's out0;
 * the consumer is te_matmul_relprop_qk_fwd_scaled_f32.  Workspace: te_add_bcast_relprop_workspace_bytes. */
int te_add_bcast_relprop_deferred_f32(const float* R, const float* X0, const float* mask, float* a, float* out1,
                                      float* fac, int64_t B, int64_t H, int64_t N,
                                      void* ws, size_t ws_bytes, te_stream_t stream);

/* ---- a6  Clone.relprop --------------------------------------------------------------------------
 * replaces modules/layers_ours.py:151-169.  out = X .(sd(R0,X) + sd(R1,X) [+ sd(R2,X)]); R2 may be
 * NULL (num = 2: ViT_LRP.py:194-196, BERT.py:247,527) or not (num = 3: BERT.py:407). n elements. */
int te_clone_relprop_f32(const float* R0, const float* R1, const float* R2, const float* X,
                         float* out, int64_t n, te_stream_t stream);

/* Clone.relprop on [B,n] operands whose relevance inputs carry deferred per-sample factors: R_i enters as
 * R_i[b,:] * s_i[b * s_i_stride]; s_i == NULL means no factor (R2 == NULL: two aliases). */
int te_clone_relprop_scaled_f32(const float* R0, const float* s0, int64_t s0_stride,
                                const float* R1, const float* s1, int64_t s1_stride,
                                const float* R2, const float* s2, int64_t s2_stride,
                                const float* X, float* out, int64_t B, int64_t n, te_stream_t stream);

/* ---- a7  IndexSelect.relprop ----------------------------------------------------------------------
 * replaces modules/layers_ours.py:129-147 for dim=1, one index (ViT_LRP.py:319,329; BERT.py:170,189).
 * R [B,C], X [B,N,C] -> out [B,N,C]: row `index` = X_row . sd(R, X_row), all other rows zero. */
int te_index_select_relprop_f32(const float* R, const float* X, float* out,
                                int64_t B, int64_t N, int64_t C, int64_t index, te_stream_t stream);

/* ---- a10  gradient x relevance, clamp, head mean ------------------------------------------------
 * replaces ViT_LRP.py:359-366 and ExplanationGenerator.py:49-56 (per sample):
 * grad, cam [B,H,N,N] -> out [B,N,N] = (sum_h max(grad*cam, 0)) / H. */
int te_gradcam_headmean_f32(const float* grad, const float* cam, float* out,
                            int64_t B, int64_t H, int64_t N, te_stream_t stream);

/* ---- a11  rollout ---------------------------------------------------------------------------------
 * replaces compute_rollout_attention, ViT_LRP.py:38-49 (flags = 0) and ExplanationGenerator.py:7-18
 * (TE_ROLLOUT_NORMALISE), plus the CLS fix-up ExplanationGenerator.py:58 (TE_ROLLOUT_CLS_FIXUP:
 * joint[b,0,0] = min_j joint[b,0,j]).  cams [L,B,N,N] -> joint [B,N,N]:
 *   M_l = cams_l + I (/ rowsum) ; J = M_start ; J = M_i J for i = start+1 .. L-1.
 * The (N x N)(N x N) products run on fp32 MFMA tiles; OR TE_IMPL_SIMPLE into `flags` for the plain fmaf kernel.
 * TE_ROLLOUT_ROW0: only row 0 of the joint matrix is produced -- what both generators consume (ViT_LRP.py:369
 * `rollout[:, 0, 1:]`, ExplanationGenerator.py:58-59 `rollout[:, 0]`) -- as the vector chain r = e_0^T M_{L-1},
 * r = r M_i (i = L-2 .. start): every layer matrix is read once (HBM-bound) instead of L-1-start N^3 products.
 * `joint` is then [B,N]; the workspace query is te_rollout_row0_workspace_bytes; N <= 1024. */
size_t te_rollout_workspace_bytes(int64_t L, int64_t B, int64_t N);
size_t te_rollout_row0_workspace_bytes(int64_t B, int64_t N);
int te_rollout_f32(const float* cams, int64_t L, int64_t start_layer, int64_t B, int64_t N,
                   int flags, float* joint, void* ws, size_t ws_bytes, te_stream_t stream);

/* te_linear_relprop_fwd_scaled_f32 (same rule: modules/layers_ours.py:207-230, variant ours, alpha = 1, Z from the
 * forward output, optional per-sample factor on R) with its three products on bf16 MFMAs at fp32 accuracy: every fp32
 * operand is used as the exact sum of three bf16 parts and the six partial products above 2^-24 are accumulated in fp32
 * (csrc/te_linear_x6.hip).  in_f, out_f >= 128, out_f % 128 == 0, in_f % 64 == 0.
 *
 * Operand planes ("P3"): an fp32 [rows, K] operand as bf16 planes in MFMA-fragment order
 * [ceil(rows / 32)][K / 16][3 planes][k-half 2][row 32][8 bf16]; te_linear_x6_planes_bytes gives the size.
 *   te_linear_x6_prepare_weights_f32   W [out_f, in_f] -> the weight-side planes of both passes (|W| as P3, then
 *                                      max(W,0)^T / min(W,0)^T interleaved per 32-row block); a caller evaluates this
 *                                      ONCE per weight version and passes the result as w_planes
 *   te_linear_x6_split_abs_f32         X [rows, K] -> P3 planes of |X| (what a producer of X may emit itself)
 *   te_linear_relprop_x6_f32           x_planes = NULL: |X| is split into the workspace first.
 * Workspace = |X| planes, S planes (the Z-pass writes S only in plane form), 64 MiB of accumulator hand-over between
 * workgroups that share a tile (sequential stream-K: one k-ordered chain per output wherever a tile is cut), flags.
 * flags: TE_X6_TILE_AUTO (per pass: 256 weight rows per tile / one 512-thread workgroup per CU where that gives every CU
 * a tile, else 128 rows / two 256-thread workgroups per CU; the result does not depend on the tile geometry, bit for
 * bit), TE_X6_TILE_128 / TE_X6_TILE_256 pin it; shifted left by TE_X6_TILE_Z_SHIFT / TE_X6_TILE_C_SHIFT they pin one pass.
 * TE_X6_TILE_128x128: 128 x 128 tiles, three 256-thread workgroups per CU (launches with few weight rows).
 * TE_X6_STAGES_3: three LDS stages instead of two in the 256-row geometry (measurement; same results).  STUDY BUILDS ONLY
 * (-DTE_X6_STUDY, te_x6_study_build() & 1): the shipped library does not contain the instantiation and answers
 * TE_ERR_UNSUPPORTED -- likewise TE_X6_KSPLIT.
 *
 * Failure is loud.  A workgroup that continues a tile another workgroup started waits for that one's accumulators for at
 * most 250 ms.  If the wait expires it ORs 1 into *status -- a caller-owned, caller-zeroed device word that is NEVER
 * cleared by the library (sticky across calls: read it once where the caller synchronises anyway) -- sets the error word
 * te_linear_relprop_x6_check reads, and continues from NaN accumulators: every output of that tile is NaN (te_gemm_x6_f32,
 * C-pass) or recomputed by the rule's exact fallback (Z-pass).  Once *status is non-zero, later waits give up at once.
 * status = NULL: the per-call error word only.  TE_X6_TEST_DROP_HANDOVER (tests): publishers keep their flags down, so
 * every such wait expires; TE_X6_TEST_SMALL_GRID (tests): sixteen workgroups, i.e. cut tiles on small shapes.
 * te_linear_relprop_x6_check (synchronises) returns 1 if a bounded hand-over wait of the last call expired. */
#define TE_X6_TILE_AUTO 0
#define TE_X6_TILE_128 1
#define TE_X6_TILE_256 2
#define TE_X6_TILE_128x128 3
#define TE_X6_STAGES_3 0x100
#define TE_X6_TEST_DROP_HANDOVER 0x200
#define TE_X6_KSPLIT 0x8000            /* study, off by default: products with K >= 1536 and <= 768 weight rows as two K segments per
                                          output (two k-ordered chains, summed once).  It changes the bits: set it for every
                                          launch of a process or for none.  Measured: no gain in the step */
#define TE_X6_WHOLE_TILES 0x10000       /* ranges cut at tile boundaries only, whatever the fill of the last round: for callers that
                                          keep several streams busy (the idle CUs of a last round are used by their other kernels,
                                          and no tile is handed over between workgroups).  Same results, bit for bit */
#define TE_X6_TEST_SMALL_GRID 0x4000   /* tests: a persistent grid of 16 workgroups, so that small shapes get stream-K cuts */
#define TE_X6_TILE_Z_SHIFT 10
#define TE_X6_TILE_C_SHIFT 12
/* optional phase mask (measurement: one phase per call on the same workspace, in this order); 0 = the whole rule */
#define TE_X6_PHASE_SPLIT 4    /* clear the flags, |X| -> planes (unless x_planes is given) */
#define TE_X6_PHASE_Z 8        /* S planes */
#define TE_X6_PHASE_C 16       /* out */
#define TE_X6_PHASE_MASK 28
int te_linear_relprop_x6_supported(int64_t T, int64_t in_f, int64_t out_f);
size_t te_linear_relprop_x6_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f);
size_t te_linear_x6_weight_planes_bytes(int64_t in_f, int64_t out_f);
size_t te_linear_x6_planes_bytes(int64_t rows, int64_t K);
int te_linear_x6_prepare_weights_f32(const float* W, int64_t in_f, int64_t out_f, void* planes, size_t planes_bytes,
                                     te_stream_t stream);
int te_linear_x6_split_abs_f32(const float* X, int64_t rows, int64_t K, void* planes, size_t planes_bytes,
                               te_stream_t stream);
int te_linear_relprop_x6_f32(const float* R, const float* r_scale, int64_t r_scale_stride, int64_t rows_per_sample,
                             const float* X, const float* W, const void* w_planes, const void* x_planes,
                             const float* Y, const float* bias, float* out,
                             int64_t T, int64_t in_f, int64_t out_f, int flags, unsigned* status, void* ws,
                             size_t ws_bytes, te_stream_t stream);
int te_linear_relprop_x6_check(const void* ws, int64_t T, int64_t in_f, int64_t out_f, te_stream_t stream);

/* The same rule for EVERY variant and alpha on the x6 kernels (csrc/te_linear_x6.hip): variant TE_VARIANT_LRP =
 * modules/layers_lrp.py:188-211 (S1 = sd(R, X+ W+^T), S2 = sd(R, X- W-^T), separate denominators), and the inhibitor half
 * alpha * f(pw, nw, px, nx) - beta * f(nw, pw, px, nx), beta = alpha - 1, of both variants (layers_ours.py:225-228).
 *   variant ours: w_planes as above, Y (the forward output) required; x_abs_planes optional (the |X| planes).
 *   variant lrp : w_planes_lrp from te_linear_x6_prepare_weights_lrp_f32 (P3 planes of max(W,0), min(W,0) and of their
 *                 transposes); in_f % 128 == 0; Y, bias, w_planes unused (may be NULL).
 * flags: TE_X6_TILE_* | TE_X6_STAGES_3 | test hooks; status as te_linear_relprop_x6_f32. */
int te_linear_relprop_x6_general_supported(int64_t T, int64_t in_f, int64_t out_f, int variant);
size_t te_linear_x6_weight_planes_lrp_bytes(int64_t in_f, int64_t out_f);
int te_linear_x6_prepare_weights_lrp_f32(const float* W, int64_t in_f, int64_t out_f, void* planes, size_t planes_bytes,
                                         te_stream_t stream);
size_t te_linear_relprop_x6_general_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f, int variant);
int te_linear_relprop_x6_general_f32(const float* R, const float* r_scale, int64_t r_scale_stride, int64_t rows_per_sample,
                                     const float* X, const float* W, const void* w_planes, const void* w_planes_lrp,
                                     const void* x_abs_planes, const float* Y, const float* bias, float* out,
                                     int64_t T, int64_t in_f, int64_t out_f, float alpha, int variant, int flags,
                                     unsigned* status, void* ws, size_t ws_bytes, te_stream_t stream);

/* The plain fp32 product on the same kernels (the forward output and the input gradient of a Linear layer,
 * modules/layers_ours.py:207 = nn.Linear): out [T, M] = X [T, K] . W^T + bias [M] with W given as signed P3 planes of an
 * [M, K] matrix.  te_linear_x6_split_matrix_f32 builds such planes from a row-major [rows, K] matrix (transposed = 0) or
 * from its transpose stored as [K, rows] (transposed = 1: the planes of W^T for d_x = d_y W).  M % 128 == 0, K % 16 == 0.
 * x_planes = NULL: X is split into the workspace first.  flags: TE_X6_TILE_* | TE_X6_STAGES_3 | TE_X6_TEST_DROP_HANDOVER | TE_X6_TEST_SMALL_GRID;
 * status: as te_linear_relprop_x6_f32. */
int te_gemm_x6_supported(int64_t T, int64_t K, int64_t M);
size_t te_gemm_x6_workspace_bytes(int64_t T, int64_t K, int64_t M);
int te_linear_x6_split_matrix_f32(const float* A, int64_t rows, int64_t K, int transposed, void* planes,
                                  size_t planes_bytes, te_stream_t stream);
int te_gemm_x6_f32(const float* X, const void* x_planes, const void* w_planes, const float* bias, float* out,
                   int64_t T, int64_t K, int64_t M, int flags, unsigned* status, void* ws, size_t ws_bytes,
                   te_stream_t stream);
/* One pass over a row-major A [rows, K]: its signed planes (the x_planes of te_gemm_x6_f32) AND the planes of |A| (the
 * x_planes of te_linear_relprop_x6_f32 for the same layer input: layers_ours.py:215 clamps / pairs X by sign, the rule's Z
 * is |X| |W|^T).  Both buffers te_linear_x6_planes_bytes(rows, K) bytes; planes_bytes = the size of each. */
int te_linear_x6_split_dual_f32(const float* A, int64_t rows, int64_t K, void* planes, void* planes_abs,
                                size_t planes_bytes, te_stream_t stream);
/* nn.GELU between the two Linear layers of an Mlp block (modules/layers_ours.py:70; ViT_LRP.py:57-69: fc1 -> act -> fc2;
 * BERT.py BertIntermediate) as a producer of operand planes -- the split passes of its two neighbours disappear (round 5):
 *   te_gelu_backward_x6_planes_f32  planes of d_h = d_a . gelu'(h), [rows, K] row-major inputs: the x_planes of the
 *                                   te_gemm_x6_f32 that forms fc1's input gradient.  Bit for bit the planes
 *                                   te_linear_x6_split_matrix_f32 builds from te_gelu_backward_f32's output, which is never
 *                                   written.
 *   te_gelu_forward_x6_planes_f32   y = gelu(x) as fp32 (te_gelu_forward_f32's bits) AND the two plane sets
 *                                   te_linear_x6_split_dual_f32 builds from y (fc2's forward product, fc2's rule).
 * K % 16 == 0; planes / planes_abs: te_linear_x6_planes_bytes(rows, K) bytes each = planes_bytes. */
int te_gelu_backward_x6_planes_f32(const float* dy, const float* x, int64_t rows, int64_t K, void* planes,
                                   size_t planes_bytes, te_stream_t stream);
int te_gelu_forward_x6_planes_f32(const float* x, float* y, int64_t rows, int64_t K, void* planes, void* planes_abs,
                                  size_t planes_bytes, te_stream_t stream);

/* ---- producers of the cached tensors ----------------------------------------------------
 * The attention block of baselines/ViT/ViT_LRP.py:132-152 (and its gradient, the tensor save_attn_gradients receives,
 * :144-145) on the fused qkv activation [B,N,3*H*D] ('b n (qkv h d)'), head dim 64, N <= 224 (k and v of a head stay
 * in LDS; te_attention_forward_supported says whether a shape qualifies -- callers keep stock PyTorch otherwise):
 *   forward : z_qk [B,H,N,N] = q k^T (unscaled: the QK rule's Z) ; attn [B,H,N,N] = softmax(z_qk * scale) ;
 *             out [B,N,H*D] = attn v in the 'b n (h d)' layout the projection consumes (= the AV rule's Z, strided)
 *   backward: d_attn [B,H,N,N] = d_out v^T (the attention gradient of the explanation) ; d_qkv [B,N,3*H*D]:
 *             d_v = attn^T d_out always ; need_qk != 0 additionally d_s = attn .(d_attn - rowsum(d_attn . attn)) * scale,
 *             d_q = d_s k, d_k = d_s^T q (need_qk == 0: the q / k thirds of d_qkv are left untouched -- the lowest
 *             block whose attention gradient is wanted has no consumer for them). */
int te_attention_forward_supported(int64_t N, int64_t D);
int te_attention_forward_f32(const float* qkv, float* z_qk, float* attn, float* out,
                             int64_t B, int64_t H, int64_t N, int64_t D, float scale, te_stream_t stream);
/* te_attention_forward_f32 that also writes the operand planes of `out` for the projection layer's x6 kernels (round 6): out_planes =
 * the signed planes of out [B N, H D], out_abs_planes (optional) = the planes of |out|, each te_linear_x6_planes_bytes(B N, H D)
 * bytes, bit for bit what te_linear_x6_split_dual_f32 writes from the fp32 tensor (te_gemm_x6_f32's x_planes / the rule's
 * x_abs_planes).  N <= 224. */
int te_attention_forward_planes_f32(const float* qkv, float* z_qk, float* attn, float* out, void* out_planes,
                                    void* out_abs_planes, size_t planes_bytes, int64_t B, int64_t H, int64_t N, int64_t D,
                                    float scale, te_stream_t stream);
int te_attention_backward_f32(const float* d_out, const float* qkv, const float* attn, float* d_attn, float* d_qkv,
                              int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk, te_stream_t stream);
/* The same with the block's forward output `out` [B,N,H*D] (what te_attention_forward_f32 wrote) at hand (round 6): the row
 * sums of the softmax backward are then taken as sum_d d_out[i][d] out[i][d] (= sum_j attn[i][j] d_attn[i][j]: out = attn v,
 * d_attn = d_out v^T) instead of from a pass over the two N x N tensors.  Results agree with te_attention_backward_f32 to
 * fp32 rounding (another summation of the same quantity), not bit for bit. */
int te_attention_backward_out_f32(const float* d_out, const float* out, const float* qkv, const float* attn, float* d_attn,
                                  float* d_qkv, int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk,
                                  te_stream_t stream);

/* The same producers for the shapes the one-workgroup-per-head kernels cannot hold (csrc/te_attn_long.hip): head dim 64,
 * N <= 640 (ViT-L/16 at 384^2: 577, baselines/ViT/ViT_LRP.py:419-425; BERT: 512), q / k / v / out / gradients as
 * [B,H,N,64] views with element strides (sb, sh, sn) -- the fused 'b n (qkv h d)' activation of ViT or the three separate
 * 'b n (h d)' activations of BERT (BERT_explainability/modules/BERT/BERT.py:307-365).
 *   forward : z_qk (optional) = q k^T unscaled ; x_scaled (optional) = z_qk * scale, WITHOUT the mask: the first operand of
 *             the Add module, BERT.py:339-342 (Add.relprop divides by x_scaled + mask: the mask must not be in it twice) ;
 *             x = x_scaled + mask[b, key] (mask [B,N] additive, NULL = none) ; attn = softmax(x) ; out = attn v
 *   backward: d_attn = d_out v^T ; d_v = attn^T d_out ; need_qk: d_q = d_s k, d_k = d_s^T q with
 *             d_s = ((d_attn - rowsum(d_attn . attn)) . attn) * scale.  Workspace: B*H*N floats. */
int te_attention_strided_supported(int64_t N, int64_t D);
size_t te_attention_backward_strided_workspace_bytes(int64_t B, int64_t H, int64_t N);
int te_attention_forward_strided_f32(const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                     const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                     const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                     const float* mask, float* z_qk, float* x_scaled, float* attn,
                                     float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                                     int64_t B, int64_t H, int64_t N, int64_t D, float scale, te_stream_t stream);
int te_attention_backward_strided_f32(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                                      const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                      const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                      const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                      const float* attn, float* d_attn,
                                      float* d_q, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                                      float* d_k, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                                      float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                                      int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk,
                                      void* ws, size_t ws_bytes, te_stream_t stream);
/* te_attention_backward_strided_f32 with the block's forward output `out` ([B,H,N,64] view, the tensor
 * te_attention_forward_strided_f32 wrote) as one more operand: the softmax backward's row sums sum_j d_attn . attn are taken
 * as d_out . out (out = attn v, d_attn = d_out v^T), which lets the row side run in ONE walk over the keys on bf16 MFMAs
 * (csrc/te_attn_bwd6l.hip, 64 < N <= 640; other lengths: the kernels of te_attention_backward_strided_f32).  Replaces the same
 * autograd nodes (ViT_LRP.py:144-145, BERT.py:349-350). */
int te_attention_backward_strided_out_f32(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                                          const float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                                          const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                          const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                          const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                          const float* attn, float* d_attn,
                                          float* d_q, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                                          float* d_k, int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                                          float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                                          int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk,
                                          void* ws, size_t ws_bytes, te_stream_t stream);

/* The LayerNorm and GELU layers around the Linear rules (modules/layers_ours.py:70-77; ViT_LRP.py:57,184,187,266;
 * BERT.py:18,52,416,463): their relprop rules are the identity, the path needs their forward values (the X / Y the
 * neighbouring rules cache) and their input gradient on the way to the attention maps.
 *   te_layernorm_forward_f32 : x [T,C] -> y = (x - mean) * rstd * weight + bias (bias may be NULL); mean, rstd [T] are
 *                              kept for the backward.  C a multiple of 4, <= 2048 (te_layernorm_supported).
 *   te_layernorm_backward_f32: dx = rstd * (a - mean_C(a) - xhat * mean_C(a * xhat)) + add,  a = dy * weight,
 *                              xhat = (x - mean) * rstd; `add` (NULL = none) is the gradient of the residual branch
 *                              that bypasses the LayerNorm (ViT_LRP.py:203-205: clone -> norm -> ... -> add).
 *   te_gelu_forward_f32      : y = 0.5 x (1 + erf(x / sqrt 2)), n a multiple of 4
 *   te_gelu_backward_f32     : dx = dy (0.5 (1 + erf(x / sqrt 2)) + x exp(-x^2 / 2) / sqrt(2 pi)) */
int te_layernorm_supported(int64_t C);
int te_layernorm_forward_f32(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                             int64_t T, int64_t C, float eps, te_stream_t stream);
int te_layernorm_backward_f32(const float* dy, const float* x, const float* weight, const float* mean, const float* rstd,
                              const float* add, float* dx, int64_t T, int64_t C, te_stream_t stream);
int te_gelu_forward_f32(const float* x, float* y, int64_t n, te_stream_t stream);
int te_gelu_backward_f32(const float* dy, const float* x, float* dx, int64_t n, te_stream_t stream);

/* ---- consumer of a relevance map ---------------------------------------------------
 * replaces baselines/ViT/imagenet_seg_eval.py:214-222 and generate_visualizations.py:99-100 (per map):
 * maps [B,g,g] -> heat [B, g*scale, g*scale] = F.interpolate(scale_factor=scale, mode='bilinear') of each map,
 * then (normalise != 0) (heat - min) / (max - min) per map; fg_mask (optional, NULL to skip) = heat > mean(heat)
 * as 0/1 floats (Res.gt(Res.mean())). */
int te_heatmap_f32(const float* maps, float* heat, float* fg_mask, int64_t B, int64_t g, int64_t scale,
                   int normalise, te_stream_t stream);

/* ---- Conv2d.relprop, z^B rule of the patch embedding (method="full") ------------------
 * replaces modules/layers_ours.py:242-256 (= modules/layers_lrp.py:223-237), the `X.shape[1] == 3` branch, for a
 * convolution with stride == kernel == p and no padding (baselines/ViT/ViT_LRP.py:215-242, PatchEmbed):
 *   Za = conv(X,W) - conv(L,W+) - conv(H,W-) + 1e-9 ; S = R / Za ; out = X convT(S,W) - L convT(S,W+) - H convT(S,W-)
 * with L / H = per-sample pixel min / max.  R: relevance of the conv output in TOKEN-MAJOR layout [B, P, E]
 * (P = (H/p)(W/p); what PatchEmbed.relprop holds before its transpose), consecutive samples r_bs floats apart
 * (r_bs >= P*E; (P+1)*E when R is cam[:, 1:] of a [B, P+1, E] tensor).  X [B,C,H,W]; W [E,C,p,p]; Y [B,E,H/p,W/p] =
 * the layer's forward output (conv(X,W) is taken as Y - bias; bias may be NULL); out [B,C,H,W].
 * flags: TE_IMPL_SIMPLE selects the one-thread-per-element C-pass. */
size_t te_conv2d_zb_relprop_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int64_t E, int64_t p);
int te_conv2d_zb_relprop_f32(const float* R, int64_t r_bs, const float* X, const float* W, const float* Y,
                             const float* bias, float* out, int64_t B, int64_t C, int64_t H, int64_t W_, int64_t E,
                             int64_t p, int flags, void* ws, size_t ws_bytes, te_stream_t stream);

/* ---- perturbation-test input builder ---------------------------------------------------
 * replaces baselines/ViT/pertubation_eval_from_hdf5.py:88-101 for ALL perturbation steps of a batch in two launches:
 *   for each step s:  idx = topk(vis[b], ks[s]) ; out[s][b][c][idx] = 0 for every channel c, data elsewhere ;
 *                     out = (out - mean[c]) / std[c]
 * vis [B,HW] relevance per pixel (negate it first for the script's --neg mode); data [B,C,HW] pixels; ks: HOST array
 * of n_steps pixel counts (int(base_size * step)), clamped to [0, HW]; mean / std: HOST arrays [C] (NULL = 0 / 1);
 * out [n_steps,B,C,HW].  Equal relevance values are removed in ascending pixel-index order (torch.topk leaves the
 * order among ties unspecified).  NaN relevance counts as largest, as in torch.topk. */
#define TE_PERTURB_MAX_STEPS 16
#define TE_PERTURB_MAX_CHANNELS 4
size_t te_perturb_workspace_bytes(int64_t B, int64_t n_steps);
int te_perturb_f32(const float* vis, const float* data, float* out, int64_t B, int64_t C, int64_t HW,
                   const int64_t* ks, int64_t n_steps, const float* mean, const float* std_, void* ws,
                   size_t ws_bytes, te_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TE_RELPROP_H */
