// te_attn.hip -- einsum / MatMul relprop of self-attention (modules/layers_ours.py:48-60,122-127;
// BERT_explainability/modules/layers_ours.py:89-91) for gfx950.
//
//   AV rule:  Z = attn v ; S = sd(R, Z) ; cam_attn = attn .(S v^T) ; cam_v = v .(attn^T S)
//   QK rule:  Z = q k^T  ; S = sd(R, Z) ; cam_q = q .(S k)         ; cam_k = k .(S^T q)
//
// q / k / v / R(AV) / cam_q / cam_k / cam_v are strided [B,H,N,D] views (element (b,h,n,d) at
// base + b*sb + h*sh + n*sn + d) so the fused qkv activation and the 'b n (qkv h d)' relevance layout
// are read and written in place -- the reference's four einops rearrange copies per block
// (ViT_LRP.py:135,148,157,175) do not exist here.
//
// This file holds the SIMPLE kernels (one thread per output element, k-ordered fmaf chains); the
// LDS-tiled MFMA versions live in te_attn_mfma.hip and are selected by the API below unless
// TE_IMPL_SIMPLE is set or the head dim is not 64.
#include "te_common.h"

namespace te_attn_mfma {
// implemented in te_attn_mfma.hip; return false if the shape is not supported by the tiled kernels
bool av_supported(int64_t N, int64_t D);
int av_launch(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn, const float* attn,
              const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, const float* Z, int64_t z_sb, int64_t z_sh,
              int64_t z_sn, float* cam_attn, float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn, int64_t B,
              int64_t H, int64_t N, int64_t D, float scale, float* ws, hipStream_t stream);
bool qk_supported(int64_t N, int64_t D);
int qk_launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
              const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb,
              int64_t cq_sh, int64_t cq_sn, float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
              int64_t B, int64_t H, int64_t N, int64_t D, float scale, float* ws, const float* r_scale,
              int64_t r_scale_stride, hipStream_t stream);
}  // namespace te_attn_mfma

namespace {

constexpr int kThreads = 256;

struct Strided {  // [B,H,N,D] view, D contiguous
  int64_t sb, sh, sn;
  __device__ __forceinline__ int64_t at(int64_t b, int64_t h, int64_t n) const { return b * sb + h * sh + n * sn; }
};

// S[b,h,i,d] = sd(R[b,h,i,d], sum_j attn[b,h,i,j] v[b,h,j,d])             (contiguous workspace)
__global__ __launch_bounds__(kThreads) void av_s_simple(
    const float* __restrict__ R, Strided rs, const float* __restrict__ attn, const float* __restrict__ v,
    Strided vs, float* __restrict__ S, int64_t B, int64_t H, int64_t N, int64_t D) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * D) return;
  const int64_t d = idx % D, i = (idx / D) % N, h = (idx / (D * N)) % H, b = idx / (D * N * H);
  const float* arow = attn + ((b * H + h) * N + i) * N;
  float z = 0.0f;
  for (int64_t j = 0; j < N; ++j) z = fmaf(arow[j], v[vs.at(b, h, j) + d], z);
  S[idx] = te_sd(R[rs.at(b, h, i) + d], z);
}

// the same with Z = the cached forward product attn v, contiguous [B,H,N,D]
__global__ __launch_bounds__(kThreads) void av_s_from_z_simple(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, float* __restrict__ S, int64_t B, int64_t H,
    int64_t N, int64_t D) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * D) return;
  const int64_t d = idx % D, i = (idx / D) % N, h = (idx / (D * N)) % H, b = idx / (D * N * H);
  S[idx] = te_sd(R[rs.at(b, h, i) + d], Z[idx]);
}
__global__ __launch_bounds__(kThreads) void qk_s_from_z_simple(const float* __restrict__ R, const float* __restrict__ Z,
                                                               float* __restrict__ S, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx < n) S[idx] = te_sd(R[idx], Z[idx]);
}

// cam_attn[b,h,i,j] = attn[b,h,i,j] * (sum_d S[b,h,i,d] v[b,h,j,d]) * scale
__global__ __launch_bounds__(kThreads) void av_cam_attn_simple(
    const float* __restrict__ S, const float* __restrict__ attn, const float* __restrict__ v, Strided vs,
    float* __restrict__ cam, int64_t B, int64_t H, int64_t N, int64_t D, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * N) return;
  const int64_t j = idx % N, i = (idx / N) % N, h = (idx / (N * N)) % H, b = idx / (N * N * H);
  const float* srow = S + ((b * H + h) * N + i) * D;
  const float* vrow = v + vs.at(b, h, j);
  float g = 0.0f;
  for (int64_t d = 0; d < D; ++d) g = fmaf(srow[d], vrow[d], g);
  cam[idx] = (attn[idx] * g) * scale;
}

// cam_v[b,h,j,d] = v[b,h,j,d] * (sum_i attn[b,h,i,j] S[b,h,i,d]) * scale
__global__ __launch_bounds__(kThreads) void av_cam_v_simple(
    const float* __restrict__ S, const float* __restrict__ attn, const float* __restrict__ v, Strided vs,
    float* __restrict__ cam_v, Strided cs, int64_t B, int64_t H, int64_t N, int64_t D, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * D) return;
  const int64_t d = idx % D, j = (idx / D) % N, h = (idx / (D * N)) % H, b = idx / (D * N * H);
  const float* a = attn + (b * H + h) * N * N + j;
  const float* s = S + (b * H + h) * N * D + d;
  float u = 0.0f;
  for (int64_t i = 0; i < N; ++i) u = fmaf(a[i * N], s[i * D], u);
  cam_v[cs.at(b, h, j) + d] = (v[vs.at(b, h, j) + d] * u) * scale;
}

// S[b,h,i,j] = sd(R[b,h,i,j], sum_d q[b,h,i,d] k[b,h,j,d])
__global__ __launch_bounds__(kThreads) void qk_s_simple(
    const float* __restrict__ R, const float* __restrict__ q, Strided qs, const float* __restrict__ k,
    Strided ks, float* __restrict__ S, int64_t B, int64_t H, int64_t N, int64_t D) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * N) return;
  const int64_t j = idx % N, i = (idx / N) % N, h = (idx / (N * N)) % H, b = idx / (N * N * H);
  const float* qrow = q + qs.at(b, h, i);
  const float* krow = k + ks.at(b, h, j);
  float z = 0.0f;
  for (int64_t d = 0; d < D; ++d) z = fmaf(qrow[d], krow[d], z);
  S[idx] = te_sd(R[idx], z);
}

// cam_q[b,h,i,d] = q[b,h,i,d] * (sum_j S[b,h,i,j] k[b,h,j,d]) * scale
__global__ __launch_bounds__(kThreads) void qk_cam_q_simple(
    const float* __restrict__ S, const float* __restrict__ q, Strided qs, const float* __restrict__ k,
    Strided ks, float* __restrict__ cam_q, Strided cs, int64_t B, int64_t H, int64_t N, int64_t D, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * D) return;
  const int64_t d = idx % D, i = (idx / D) % N, h = (idx / (D * N)) % H, b = idx / (D * N * H);
  const float* srow = S + ((b * H + h) * N + i) * N;
  float u = 0.0f;
  for (int64_t j = 0; j < N; ++j) u = fmaf(srow[j], k[ks.at(b, h, j) + d], u);
  cam_q[cs.at(b, h, i) + d] = (q[qs.at(b, h, i) + d] * u) * scale;
}

// cam_k[b,h,j,d] = k[b,h,j,d] * (sum_i S[b,h,i,j] q[b,h,i,d]) * scale
__global__ __launch_bounds__(kThreads) void qk_cam_k_simple(
    const float* __restrict__ S, const float* __restrict__ q, Strided qs, const float* __restrict__ k,
    Strided ks, float* __restrict__ cam_k, Strided cs, int64_t B, int64_t H, int64_t N, int64_t D, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * H * N * D) return;
  const int64_t d = idx % D, j = (idx / D) % N, h = (idx / (D * N)) % H, b = idx / (D * N * H);
  const float* s = S + (b * H + h) * N * N + j;
  float u = 0.0f;
  for (int64_t i = 0; i < N; ++i) u = fmaf(s[i * N], q[qs.at(b, h, i) + d], u);
  cam_k[cs.at(b, h, j) + d] = (k[ks.at(b, h, j) + d] * u) * scale;
}

inline bool strides_ok(int64_t sb, int64_t sh, int64_t sn) { return sb >= 0 && sh >= 0 && sn > 0; }

}  // namespace

// AV workspace: S [B,H,N,D] + Z [B,H,N,D] (Z only used when the caller has no cached forward product)
extern "C" size_t te_matmul_relprop_av_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t D) {
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0) return 0;
  return te_align_up((size_t)2 * B * H * N * D * sizeof(float), 256);
}

extern "C" int te_matmul_relprop_av_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                                        const float* attn, const float* v, int64_t v_sb, int64_t v_sh,
                                        int64_t v_sn, float* cam_attn, float* cam_v, int64_t cv_sb,
                                        int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H, int64_t N,
                                        int64_t D, float out_scale, int variant, void* ws, size_t ws_bytes,
                                        te_stream_t stream_) {
  return te_matmul_relprop_av_fwd_f32(R, r_sb, r_sh, r_sn, attn, v, v_sb, v_sh, v_sn, nullptr, cam_attn, cam_v, cv_sb,
                                      cv_sh, cv_sn, B, H, N, D, out_scale, variant, ws, ws_bytes, stream_);
}

extern "C" int te_matmul_relprop_av_fwd_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                                            const float* attn, const float* v, int64_t v_sb, int64_t v_sh,
                                            int64_t v_sn, const float* Z, float* cam_attn, float* cam_v,
                                            int64_t cv_sb, int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H,
                                            int64_t N, int64_t D, float out_scale, int variant, void* ws,
                                            size_t ws_bytes, te_stream_t stream_) {
  return te_matmul_relprop_av_fwdz_f32(R, r_sb, r_sh, r_sn, attn, v, v_sb, v_sh, v_sn, Z, H * N * D, N * D, D, cam_attn,
                                       cam_v, cv_sb, cv_sh, cv_sn, B, H, N, D, out_scale, variant, ws, ws_bytes, stream_);
}

extern "C" int te_matmul_relprop_av_fwdz_f32(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn,
                                             const float* attn, const float* v, int64_t v_sb, int64_t v_sh,
                                             int64_t v_sn, const float* Z, int64_t z_sb, int64_t z_sh, int64_t z_sn,
                                             float* cam_attn, float* cam_v, int64_t cv_sb, int64_t cv_sh,
                                             int64_t cv_sn, int64_t B, int64_t H, int64_t N, int64_t D,
                                             float out_scale, int variant, void* ws, size_t ws_bytes,
                                             te_stream_t stream_) {
  if (!R || !attn || !v || !cam_attn || !cam_v || B <= 0 || H <= 0 || N <= 0 || D <= 0)
    return TE_ERR_INVALID_ARG;
  const bool z_contig = !Z || (z_sn == D && z_sh == N * D && z_sb == H * N * D);
  if (!strides_ok(r_sb, r_sh, r_sn) || !strides_ok(v_sb, v_sh, v_sn) || !strides_ok(cv_sb, cv_sh, cv_sn))
    return TE_ERR_INVALID_ARG;
  if (!ws || ws_bytes < te_matmul_relprop_av_workspace_bytes(B, H, N, D)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* S = (float*)ws;
  if (!(variant & TE_IMPL_SIMPLE) && te_attn_mfma::av_supported(N, D)) {
    int rc = te_attn_mfma::av_launch(R, r_sb, r_sh, r_sn, attn, v, v_sb, v_sh, v_sn, Z, z_sb, z_sh, z_sn, cam_attn,
                                     cam_v, cv_sb, cv_sh, cv_sn, B, H, N, D, out_scale, S, stream);
    if (rc != TE_OK) return rc;
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
  if (!z_contig) return TE_ERR_UNSUPPORTED;      // the simple kernels read Z as contiguous [B,H,N,D]
  const Strided rs{r_sb, r_sh, r_sn}, vs{v_sb, v_sh, v_sn}, cs{cv_sb, cv_sh, cv_sn};
  const int64_t nd = B * H * N * D, nn = B * H * N * N;
  dim3 blk(kThreads);
  if (Z) av_s_from_z_simple<<<dim3((unsigned)te_ceil_div(nd, kThreads)), blk, 0, stream>>>(R, rs, Z, S, B, H, N, D);
  else av_s_simple<<<dim3((unsigned)te_ceil_div(nd, kThreads)), blk, 0, stream>>>(R, rs, attn, v, vs, S, B, H, N, D);
  av_cam_attn_simple<<<dim3((unsigned)te_ceil_div(nn, kThreads)), blk, 0, stream>>>(S, attn, v, vs, cam_attn, B,
                                                                                   H, N, D, out_scale);
  av_cam_v_simple<<<dim3((unsigned)te_ceil_div(nd, kThreads)), blk, 0, stream>>>(S, attn, v, vs, cam_v, cs, B, H,
                                                                                N, D, out_scale);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// QK workspace: S [B,H,N,N] + Z [B,H,N,N] (Z only used when the caller has no cached forward product)
extern "C" size_t te_matmul_relprop_qk_workspace_bytes(int64_t B, int64_t H, int64_t N, int64_t D) {
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0) return 0;
  return te_align_up((size_t)2 * B * H * N * N * sizeof(float), 256);
}

extern "C" int te_matmul_relprop_qk_f32(const float* R_nn, const float* q, int64_t q_sb, int64_t q_sh,
                                        int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                        float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, float* cam_k,
                                        int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H,
                                        int64_t N, int64_t D, float out_scale, int variant, void* ws,
                                        size_t ws_bytes, te_stream_t stream_) {
  return te_matmul_relprop_qk_fwd_f32(R_nn, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, nullptr, cam_q, cq_sb, cq_sh,
                                      cq_sn, cam_k, ck_sb, ck_sh, ck_sn, B, H, N, D, out_scale, variant, ws, ws_bytes,
                                      stream_);
}

extern "C" int te_matmul_relprop_qk_fwd_f32(const float* R_nn, const float* q, int64_t q_sb, int64_t q_sh,
                                            int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh,
                                            int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb,
                                            int64_t cq_sh, int64_t cq_sn, float* cam_k, int64_t ck_sb,
                                            int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N,
                                            int64_t D, float out_scale, int variant, void* ws, size_t ws_bytes,
                                            te_stream_t stream_) {
  return te_matmul_relprop_qk_fwd_scaled_f32(R_nn, nullptr, 0, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, Z, cam_q, cq_sb,
                                             cq_sh, cq_sn, cam_k, ck_sb, ck_sh, ck_sn, B, H, N, D, out_scale, variant, ws,
                                             ws_bytes, stream_);
}

extern "C" int te_matmul_relprop_qk_fwd_scaled_f32(const float* R_nn, const float* r_scale, int64_t r_scale_stride,
                                                   const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                                   const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                                   const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh,
                                                   int64_t cq_sn, float* cam_k, int64_t ck_sb, int64_t ck_sh,
                                                   int64_t ck_sn, int64_t B, int64_t H, int64_t N, int64_t D,
                                                   float out_scale, int variant, void* ws, size_t ws_bytes,
                                                   te_stream_t stream_) {
  if (!R_nn || !q || !k || !cam_q || !cam_k || B <= 0 || H <= 0 || N <= 0 || D <= 0) return TE_ERR_INVALID_ARG;
  if (!strides_ok(q_sb, q_sh, q_sn) || !strides_ok(k_sb, k_sh, k_sn) || !strides_ok(cq_sb, cq_sh, cq_sn) ||
      !strides_ok(ck_sb, ck_sh, ck_sn))
    return TE_ERR_INVALID_ARG;
  if (!ws || ws_bytes < te_matmul_relprop_qk_workspace_bytes(B, H, N, D)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* S = (float*)ws;
  if (!(variant & TE_IMPL_SIMPLE) && te_attn_mfma::qk_supported(N, D)) {
    int rc = te_attn_mfma::qk_launch(R_nn, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, Z, cam_q, cq_sb, cq_sh, cq_sn,
                                     cam_k, ck_sb, ck_sh, ck_sn, B, H, N, D, out_scale, S, r_scale, r_scale_stride,
                                     stream);
    if (rc != TE_OK) return rc;
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
  if (r_scale) return TE_ERR_UNSUPPORTED;        // the simple kernels take a plain relevance operand
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
  const int64_t nd = B * H * N * D, nn = B * H * N * N;
  dim3 blk(kThreads);
  if (Z) qk_s_from_z_simple<<<dim3((unsigned)te_ceil_div(nn, kThreads)), blk, 0, stream>>>(R_nn, Z, S, nn);
  else qk_s_simple<<<dim3((unsigned)te_ceil_div(nn, kThreads)), blk, 0, stream>>>(R_nn, q, qs, k, ks, S, B, H, N, D);
  qk_cam_q_simple<<<dim3((unsigned)te_ceil_div(nd, kThreads)), blk, 0, stream>>>(S, q, qs, k, ks, cam_q, cqs, B, H,
                                                                                N, D, out_scale);
  qk_cam_k_simple<<<dim3((unsigned)te_ceil_div(nd, kThreads)), blk, 0, stream>>>(S, q, qs, k, ks, cam_k, cks, B, H,
                                                                                N, D, out_scale);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
