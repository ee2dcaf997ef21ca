// te_attn_mfma.hip -- LDS-tiled fp32-MFMA kernels for the attention einsum / MatMul relprop rules
// (modules/layers_ours.py:48-60,122-127) on gfx950, head dim D = 64.
//
//   AV rule:  Z = attn v ; S = sd(R, Z) ; cam_attn = attn .(S v^T) ; cam_v = v .(attn^T S)
//   QK rule:  Z = q k^T  ; S = sd(R, Z) ; cam_q = q .(S k)         ; cam_k = k .(S^T q)
//
// Three kernels, each a 256-thread block (4 waves as 2 x 2) owning a 64-wide tile of one (b,h) problem
// and marching over the token dimension in 64-deep chunks staged through LDS:
//
//   av_row   (b,h, 64 query rows):  Z tile (K = N) -> S tile kept in LDS and written to the workspace
//                                   -> per key chunk G = S v^T (K = 64) -> cam_attn = attn . G
//   qk_row   (b,h, 64 query rows):  per key chunk: Z = q k^T (K = 64) -> S chunk = sd(R, Z) into LDS and
//                                   the workspace -> cam_q accumulator += S_chunk k_chunk (same k tile)
//   col      (b,h, 64 key columns): out = X .(M^T Y) with K = N; used for cam_v (M = attn, Y = S, X = v)
//                                   and cam_k (M = S of the QK rule, Y = q, X = k)
//
// Every product is v_mfma_f32_32x32x2_f32 (exact f32 fma chain).  LDS tiles are [64][65] floats: the odd
// leading dimension makes both fragment access patterns conflict-free for ds_read_b32 -- row-operand
// reads (lane -> row, stride 65) and column-operand reads (lane -> consecutive column) -- so one staged
// tile can serve as the A operand of one product and the B operand of the next (k in qk_row, v in
// av_row) without a transposed copy.  Strided [B,H,N,D] operands are read in place (fused qkv layout).
// Blocks of one (b,h) are blockIdx = tile * BH + bh apart: with BH a multiple of 8 they land on one XCD
// and share that L2's copy of k / v / S.
#include "te_common.h"

namespace te_attn_mfma {

namespace {

constexpr int TS = 64;        // tile side
constexpr int LD = TS + 1;    // odd leading dim: conflict-free ds_read_b32 in both operand roles
constexpr int kThreads = 256;

struct Strided {  // [B,H,N,D] view, D contiguous
  int64_t sb, sh, sn;
};

#define TE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Copy src[r * row_stride + c] (r < rows_valid, c < cols_valid) into lds[r * LD + c], zero-filling the
// rest of the 64 x 64 tile.  One wave-instruction covers one 256-byte row segment.
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ src,
                                           int64_t row_stride, int rows_valid, int cols_valid) {
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + 4 * i;
    v[i] = (r < rows_valid && c < cols_valid) ? src[(int64_t)r * row_stride + c] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) lds[(r0 + 4 * i) * LD + c] = v[i];
}

// acc(32 x 32 block (wm, wn) of a 64 x 64 tile) += A B over `ksteps` k-pairs.
//   A_KM = false: A stored [m][k];  true: A stored [k][m]
//   B_NK = false: B stored [k][n];  true: B stored [n][k]
template <bool A_KM, bool B_NK>
__device__ __forceinline__ void mma_tile(f32x16& acc, const float* __restrict__ At,
                                         const float* __restrict__ Bt, int wm, int wn, int lr, int kh,
                                         int ksteps) {
  const float* ap = A_KM ? (At + kh * LD + wm * 32 + lr) : (At + (wm * 32 + lr) * LD + kh);
  const float* bp = B_NK ? (Bt + (wn * 32 + lr) * LD + kh) : (Bt + kh * LD + wn * 32 + lr);
  constexpr int a_step = A_KM ? 2 * LD : 2;
  constexpr int b_step = B_NK ? 2 : 2 * LD;
#pragma unroll 8
  for (int s = 0; s < ksteps; ++s) acc = TE_MFMA(ap[s * a_step], bp[s * b_step], acc);
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.0f;
}

// row of accumulator element e inside the wave's 32 x 32 block (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int e, int kh) { return (e & 3) + 8 * (e >> 2) + 4 * kh; }

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void av_row_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ attn, const float* __restrict__ v,
    Strided vs, float* __restrict__ cam_attn, float* __restrict__ Sws, int H, int N, int BH, float scale) {
  __shared__ float At[TS * LD];
  __shared__ float Vt[TS * LD];
  __shared__ float St[TS * LD];
  const int bh = blockIdx.x % BH, rt = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, rows_valid = min(TS, N - row0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool row_active = (wm * 32) < rows_valid;
  const float* attn_bh = attn + (int64_t)bh * N * N;
  const float* v_bh = v + (int64_t)b * vs.sb + (int64_t)h * vs.sh;
  const float* r_bh = R + (int64_t)b * rs.sb + (int64_t)h * rs.sh;
  const int nch = (N + TS - 1) / TS;

  f32x16 acc;
  zero(acc);
  for (int c = 0; c < nch; ++c) {
    const int kc = min(TS, N - c * TS);
    stage_tile(At, attn_bh + (int64_t)row0 * N + c * TS, N, rows_valid, kc);
    stage_tile(Vt, v_bh + (int64_t)(c * TS) * vs.sn, vs.sn, kc, TS);
    __syncthreads();
    if (row_active) mma_tile<false, false>(acc, At, Vt, wm, wn, lr, kh, (kc + 1) >> 1);
    __syncthreads();
  }
  // S = sd(R, Z): into LDS (A operand of the next product) and the workspace (cam_v kernel)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = wm * 32 + acc_row(e, kh), d = wn * 32 + lr;
    const int gi = row0 + row;
    float s = 0.0f;
    if (gi < N) {
      s = te_sd(r_bh[(int64_t)gi * rs.sn + d], acc[e]);
      Sws[((int64_t)bh * N + gi) * TS + d] = s;
    }
    St[row * LD + d] = s;
  }
  __syncthreads();
  // cam_attn = attn . (S v^T) * scale, one 64-key chunk at a time
  for (int c = 0; c < nch; ++c) {
    const int kc = min(TS, N - c * TS);
    stage_tile(Vt, v_bh + (int64_t)(c * TS) * vs.sn, vs.sn, kc, TS);   // [n = key][k = d]
    __syncthreads();
    if (row_active && (wn * 32) < kc) {
      f32x16 g;
      zero(g);
      mma_tile<false, true>(g, St, Vt, wm, wn, lr, kh, TS / 2);
      const int gj = c * TS + wn * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int gi = row0 + wm * 32 + acc_row(e, kh);
        if (gi < N && gj < N) {
          const int64_t off = (int64_t)bh * N * N + (int64_t)gi * N + gj;
          cam_attn[off] = (attn[off] * g[e]) * scale;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// out[j,d] = X[j,d] * (sum_i M[i,j] Y[i,d]) * scale       (M contiguous [BH,N,N]; X, Y, out strided)
__global__ __launch_bounds__(kThreads) void col_kernel(
    const float* __restrict__ M, const float* __restrict__ Y, Strided ys, const float* __restrict__ X,
    Strided xs, float* __restrict__ out, Strided os, int H, int N, int BH, float scale) {
  __shared__ float At[TS * LD];
  __shared__ float Bt[TS * LD];
  const int bh = blockIdx.x % BH, ct = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int col0 = ct * TS, cols_valid = min(TS, N - col0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool active = (wm * 32) < cols_valid;
  const float* m_bh = M + (int64_t)bh * N * N;
  const float* y_bh = Y + (int64_t)b * ys.sb + (int64_t)h * ys.sh;
  const int nch = (N + TS - 1) / TS;
  f32x16 acc;
  zero(acc);
  for (int c = 0; c < nch; ++c) {
    const int kc = min(TS, N - c * TS);
    stage_tile(At, m_bh + (int64_t)(c * TS) * N + col0, N, kc, cols_valid);   // [k = i][m = j]
    stage_tile(Bt, y_bh + (int64_t)(c * TS) * ys.sn, ys.sn, kc, TS);          // [k = i][n = d]
    __syncthreads();
    if (active) mma_tile<true, false>(acc, At, Bt, wm, wn, lr, kh, (kc + 1) >> 1);
    __syncthreads();
  }
  const float* x_bh = X + (int64_t)b * xs.sb + (int64_t)h * xs.sh;
  float* o_bh = out + (int64_t)b * os.sb + (int64_t)h * os.sh;
  const int d = wn * 32 + lr;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = col0 + wm * 32 + acc_row(e, kh);
    if (j < N) o_bh[(int64_t)j * os.sn + d] = (x_bh[(int64_t)j * xs.sn + d] * acc[e]) * scale;
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void qk_row_kernel(
    const float* __restrict__ Rnn, const float* __restrict__ q, Strided qs, const float* __restrict__ k,
    Strided ks, float* __restrict__ cam_q, Strided cs, float* __restrict__ Sws, int H, int N, int BH,
    float scale) {
  __shared__ float Qt[TS * LD];
  __shared__ float Kt[TS * LD];
  __shared__ float St[TS * LD];
  const int bh = blockIdx.x % BH, rt = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, rows_valid = min(TS, N - row0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool row_active = (wm * 32) < rows_valid;
  const float* q_bh = q + (int64_t)b * qs.sb + (int64_t)h * qs.sh;
  const float* k_bh = k + (int64_t)b * ks.sb + (int64_t)h * ks.sh;
  const int nch = (N + TS - 1) / TS;

  stage_tile(Qt, q_bh + (int64_t)row0 * qs.sn, qs.sn, rows_valid, TS);   // [m = i][k = d]
  f32x16 accq;
  zero(accq);
  for (int c = 0; c < nch; ++c) {
    const int kc = min(TS, N - c * TS);
    stage_tile(Kt, k_bh + (int64_t)(c * TS) * ks.sn, ks.sn, kc, TS);      // [j][d]
    __syncthreads();
    // Z chunk = q k^T (K = 64), S chunk = sd(R, Z) -> LDS [m = i][k = j] and workspace
    f32x16 z;
    zero(z);
    if (row_active && (wn * 32) < kc) mma_tile<false, true>(z, Qt, Kt, wm, wn, lr, kh, TS / 2);
    const int gj = c * TS + wn * 32 + lr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + acc_row(e, kh);
      const int gi = row0 + row;
      float s = 0.0f;
      if (gi < N && gj < N) {
        const int64_t off = (int64_t)bh * N * N + (int64_t)gi * N + gj;
        s = te_sd(Rnn[off], z[e]);
        Sws[off] = s;
      }
      St[row * LD + wn * 32 + lr] = s;
    }
    __syncthreads();
    // cam_q accumulator += S_chunk k_chunk  (K = keys of this chunk; the same k tile, now [k = j][n = d])
    if (row_active) mma_tile<false, false>(accq, St, Kt, wm, wn, lr, kh, (kc + 1) >> 1);
    __syncthreads();
  }
  float* o_bh = cam_q + (int64_t)b * cs.sb + (int64_t)h * cs.sh;
  const int d = wn * 32 + lr;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int gi = row0 + wm * 32 + acc_row(e, kh);
    if (gi < N) o_bh[(int64_t)gi * cs.sn + d] = (q_bh[(int64_t)gi * qs.sn + d] * accq[e]) * scale;
  }
}

}  // namespace

bool av_supported(int64_t N, int64_t D) { return D == TS && N >= 1 && N <= (1 << 20); }
bool qk_supported(int64_t N, int64_t D) { return D == TS && N >= 1 && N <= (1 << 20); }

int av_launch(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn, const float* attn, const float* v,
              int64_t v_sb, int64_t v_sh, int64_t v_sn, float* cam_attn, float* cam_v, int64_t cv_sb,
              int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H, int64_t N, int64_t D, float scale, float* wsS,
              hipStream_t stream) {
  if (D != TS) return TE_ERR_UNSUPPORTED;
  const int BH = (int)(B * H);
  const int nt = (int)((N + TS - 1) / TS);
  const Strided rs{r_sb, r_sh, r_sn}, vs{v_sb, v_sh, v_sn}, cs{cv_sb, cv_sh, cv_sn};
  const Strided ss{H * N * (int64_t)TS, N * (int64_t)TS, (int64_t)TS};   // workspace S [B,H,N,64]
  av_row_kernel<<<dim3((unsigned)(BH * nt)), dim3(kThreads), 0, stream>>>(R, rs, attn, v, vs, cam_attn, wsS, (int)H,
                                                                          (int)N, BH, scale);
  col_kernel<<<dim3((unsigned)(BH * nt)), dim3(kThreads), 0, stream>>>(attn, wsS, ss, v, vs, cam_v, cs, (int)H,
                                                                       (int)N, BH, scale);
  return TE_OK;
}

int qk_launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k,
              int64_t k_sb, int64_t k_sh, int64_t k_sn, float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
              float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N, int64_t D,
              float scale, float* wsS, hipStream_t stream) {
  if (D != TS) return TE_ERR_UNSUPPORTED;
  const int BH = (int)(B * H);
  const int nt = (int)((N + TS - 1) / TS);
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
  qk_row_kernel<<<dim3((unsigned)(BH * nt)), dim3(kThreads), 0, stream>>>(Rnn, q, qs, k, ks, cam_q, cqs, wsS, (int)H,
                                                                          (int)N, BH, scale);
  col_kernel<<<dim3((unsigned)(BH * nt)), dim3(kThreads), 0, stream>>>(wsS, q, qs, k, ks, cam_k, cks, (int)H, (int)N,
                                                                       BH, scale);
  return TE_OK;
}

}  // namespace te_attn_mfma
