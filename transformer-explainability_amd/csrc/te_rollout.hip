// te_rollout.hip -- compute_rollout_attention (ViT_LRP.py:38-49 un-normalised; ExplanationGenerator.py:7-18
// row-normalised) and the BERT CLS fix-up (ExplanationGenerator.py:58) for gfx950.
//
//   M_l = cams_l + I  [/ rowsum(M_l)] ;  J = M_start ;  J = M_i J  (i = start+1 .. L-1)
//
// prep kernel: one wave per row builds M_l in the workspace (adds the identity, optional row
// normalisation with the row sum taken in index order like torch.sum over the last dim).
// chain kernel: batched (N x N)(N x N) fp32 product on v_mfma_f32_32x32x2_f32, 64 x 64 output tile per block
// (rollout_bmm_mfma_kernel in te_attn_mfma.hip, the tile machinery of the attention rules).  The plain kernel below
// (LDS-staged 16-deep K slices, 4 x 4 register micro-tile per thread, k-ordered fmaf chains) is its cross-check,
// selected by TE_IMPL_SIMPLE in `flags`.
//
// TE_ROLLOUT_ROW0: both generators read ROW 0 of the joint matrix only (ViT_LRP.py:369 `[:, 0, 1:]`,
// ExplanationGenerator.py:58-59 `rollout[:, 0]`), and row 0 of M_{L-1} ... M_s is the vector chain
//   r = e_0^T M_{L-1} ;  r = r M_i  (i = L-2 .. s)
// -- (L - s) vector x matrix steps that read every layer's N x N matrix once (HBM-bound, (L-s) N^2 4 B per sample)
// instead of (L-1-s) full N^3 products.  rollout_row_step_kernel: one block per (sample, 16-row slab of M_i); each
// wave takes rows k of the slab, builds (A[k,:] + e_k) [/ rowsum] on the fly from the head-mean matrix (no M copy),
// and accumulates r[k] * row into per-lane registers; the four waves are folded through LDS in a fixed order and
// the slab's partial vector goes to the workspace, which the NEXT step folds (again in slab order) into its r.
// Slab size and fold order depend on N only, so a batch equals its samples run one by one, bit for bit.
#include "te_common.h"

namespace te_attn_mfma {
int rollout_bmm_launch(const float* A, const float* Bm, float* C, int64_t B, int64_t N, hipStream_t stream);
}

namespace {

constexpr int kThreads = 256;
constexpr int TS = 64, KS = 16;

__global__ __launch_bounds__(kThreads) void rollout_prep_kernel(const float* __restrict__ cams,
                                                                float* __restrict__ M, int64_t rows, int64_t N,
                                                                int normalise) {
  // rows = L*B*N matrix rows; one wave per row
  const int64_t row = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int64_t i = row % N;
  const float* src = cams + row * N;
  float* dst = M + row * N;
  float inv_den = 1.0f;
  float den = 0.0f;
  if (normalise) {
    double s = 0.0;
    for (int64_t j = lane; j < N; j += 64) s += (double)(src[j] + (j == i ? 1.0f : 0.0f));
    s = te_wave_sum(s);
    s = __shfl(s, 0, 64);
    den = (float)s;
  }
  (void)inv_den;
  for (int64_t j = lane; j < N; j += 64) {
    float v = src[j] + (j == i ? 1.0f : 0.0f);
    if (normalise) v = v / den;
    dst[j] = v;
  }
}

// C[b] = A[b] * Bm[b], all N x N row-major
__global__ __launch_bounds__(kThreads) void rollout_bmm_kernel(const float* __restrict__ A,
                                                               const float* __restrict__ Bm, float* __restrict__ C,
                                                               int64_t N) {
  __shared__ float As[KS][TS + 1];
  __shared__ float Bs[KS][TS + 1];
  const int64_t b = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.y * TS, c0 = (int64_t)blockIdx.x * TS;
  const float* a = A + b * N * N;
  const float* bm = Bm + b * N * N;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
  for (int64_t k0 = 0; k0 < N; k0 += KS) {
    // A tile [64 rows][16 k] -> As[k][row];  B tile [16 k][64 cols] -> Bs[k][col]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = threadIdx.x + t * kThreads;  // 0..1023
      const int ar = idx >> 4, ak = idx & 15;
      const int64_t gr = r0 + ar, gk = k0 + ak;
      As[ak][ar] = (gr < N && gk < N) ? a[gr * N + gk] : 0.0f;
      const int bk = idx >> 6, bc = idx & 63;
      const int64_t gk2 = k0 + bk, gc = c0 + bc;
      Bs[bk][bc] = (gk2 < N && gc < N) ? bm[gk2 * N + gc] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gr = r0 + ty * 4 + i, gc = c0 + tx * 4 + j;
      if (gr < N && gc < N) C[b * N * N + gr * N + gc] = acc[i][j];
    }
}

__global__ __launch_bounds__(kThreads) void rollout_copy_kernel(const float* __restrict__ src,
                                                                float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// joint[b,0,0] = min_j joint[b,0,j]     (ExplanationGenerator.py:58)
__global__ __launch_bounds__(64) void rollout_cls_fixup_kernel(float* __restrict__ joint, int64_t N) {
  float* row = joint + (int64_t)blockIdx.x * N * N;
  float m = INFINITY;
  for (int64_t j = threadIdx.x; j < N; j += 64) m = fminf(m, row[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fminf(m, __shfl_down(m, off, 64));
  if (threadIdx.x == 0) row[0] = m;
}


// ---- row-0 chain ------------------------------------------------------------------------------------------------
constexpr int kRowSlab = 16;      // rows of M_i per block
constexpr int kRowMaxJ = 16;      // most columns per lane: N <= 64 * 16 = 1024

// partial_out[b, slab, :] = sum_{k in slab} r[k] * M[k, :],  M = (A + I) [/ rowsum(A + I)]
//   r[k] = sum over the previous step's slabs of partial_in[b, slab', k]  (fixed order), or e_0 when FIRST
template <bool FIRST, int MJ>     // MJ columns per lane: N <= 64 * MJ
__global__ __launch_bounds__(kThreads) void rollout_row_step_kernel(
    const float* __restrict__ A, const float* __restrict__ partial_in, float* __restrict__ partial_out, int N,
    int nslab_in, int normalise) {
  __shared__ float r_s[kRowSlab];
  __shared__ float fold[kThreads / 64 - 1][64 * MJ];
  constexpr int RPW = kRowSlab / (kThreads / 64);      // rows per wave
  const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k0 = slab * kRowSlab, k1 = min(N, k0 + kRowSlab);
  const float* a_b = A + (int64_t)b * N * N;
  // this wave's rows of M (k = k0 + wave + 4 u), requested before anything waits: e_0^T M needs row 0 only
  const int kend = FIRST ? min(k1, 1) : k1;
  float v[RPW][MJ];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int k = k0 + wave + u * (kThreads / 64);
#pragma unroll
    for (int m = 0; m < MJ; ++m) {
      const int j = lane + 64 * m;
      v[u][m] = (k < kend && j < N) ? a_b[(int64_t)k * N + j] : 0.0f;
    }
  }
  if (threadIdx.x < kRowSlab) {
    const int k = k0 + threadIdx.x;
    float r = 0.0f;
    if (k < N) {
      if constexpr (FIRST) {
        r = (k == 0) ? 1.0f : 0.0f;
      } else {
        // slab order, loads issued 16 at a time (a plain loop waits for every load before the next add)
        const float* p = partial_in + (int64_t)b * nslab_in * N + k;
        for (int s0 = 0; s0 < nslab_in; s0 += 16) {
          float t[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) t[u] = (s0 + u < nslab_in) ? p[(int64_t)(s0 + u) * N] : 0.0f;
#pragma unroll
          for (int u = 0; u < 16; ++u) r = r + t[u];
        }
      }
    }
    r_s[threadIdx.x] = r;
  }
  __syncthreads();
  float acc[MJ];
#pragma unroll
  for (int m = 0; m < MJ; ++m) acc[m] = 0.0f;
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int k = k0 + wave + u * (kThreads / 64);
    if (k < kend) {                      // wave-uniform
      const float rk = r_s[k - k0];
#pragma unroll
      for (int m = 0; m < MJ; ++m) {
        const int j = lane + 64 * m;
        if (j < N) v[u][m] = v[u][m] + (j == k ? 1.0f : 0.0f);
      }
      if (normalise) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < MJ; ++m)
          if (lane + 64 * m < N) s += (double)v[u][m];
        s = te_wave_sum(s);
        const float den = (float)__shfl(s, 0, 64);
#pragma unroll
        for (int m = 0; m < MJ; ++m) v[u][m] = v[u][m] / den;
      }
#pragma unroll
      for (int m = 0; m < MJ; ++m) acc[m] = fmaf(rk, v[u][m], acc[m]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int m = 0; m < MJ; ++m) fold[wave - 1][m * 64 + lane] = acc[m];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = partial_out + ((int64_t)b * nslab + slab) * N;
#pragma unroll
    for (int m = 0; m < MJ; ++m) {
      const int j = lane + 64 * m;
      float t = acc[m];
#pragma unroll
      for (int w = 0; w < kThreads / 64 - 1; ++w) t = t + fold[w][m * 64 + lane];
      if (j < N) out[j] = t;
    }
  }
}

// joint_row[b, :] = sum over slabs of partial[b, slab, :]; optional CLS fix-up joint_row[b,0] = min_j joint_row[b,j]
__global__ __launch_bounds__(kThreads) void rollout_row_finish_kernel(const float* __restrict__ partial,
                                                                      float* __restrict__ out, int N, int nslab,
                                                                      int cls_fixup) {
  __shared__ float red[kThreads / 64];
  const int b = blockIdx.x;
  const float* p = partial + (int64_t)b * nslab * N;
  float mn = INFINITY;
  for (int j = threadIdx.x; j < N; j += kThreads) {
    float r = 0.0f;
    for (int s0 = 0; s0 < nslab; s0 += 16) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = (s0 + u < nslab) ? p[(int64_t)(s0 + u) * N + j] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u) r = r + t[u];
    }
    out[(int64_t)b * N + j] = r;
    mn = fminf(mn, r);
  }
  if (!cls_fixup) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mn = fminf(mn, __shfl_down(mn, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int w = 1; w < kThreads / 64; ++w) m = fminf(m, red[w]);
    out[(int64_t)b * N] = m;
  }
}
}  // namespace

// workspace: M [L,B,N,N] (identity added / normalised copies) + one ping-pong joint [B,N,N]
extern "C" size_t te_rollout_workspace_bytes(int64_t L, int64_t B, int64_t N) {
  if (L <= 0 || B <= 0 || N <= 0) return 0;
  const size_t m = te_align_up((size_t)L * B * N * N * sizeof(float), 256);
  const size_t j = te_align_up((size_t)B * N * N * sizeof(float), 256);
  return m + j;
}

// row-0 chain: two ping-pong slab-partial buffers [B, ceil(N/16), N]
extern "C" size_t te_rollout_row0_workspace_bytes(int64_t B, int64_t N) {
  if (B <= 0 || N <= 0) return 0;
  return 2 * te_align_up((size_t)B * te_ceil_div(N, kRowSlab) * N * sizeof(float), 256);
}

extern "C" int te_rollout_f32(const float* cams, int64_t L, int64_t start_layer, int64_t B, int64_t N,
                              int flags, float* joint, void* ws, size_t ws_bytes, te_stream_t stream_) {
  if (!cams || !joint || L <= 0 || B <= 0 || N <= 0 || start_layer < 0 || start_layer >= L)
    return TE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t mat = B * N * N;
  if (flags & TE_ROLLOUT_ROW0) {
    // joint is [B,N]: row 0 of the chain product only
    if (N > 64 * kRowMaxJ || B > 65535) return TE_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < te_rollout_row0_workspace_bytes(B, N)) return TE_ERR_WORKSPACE;
    const int nslab = (int)te_ceil_div(N, kRowSlab);
    float* pp[2] = {(float*)ws, (float*)((char*)ws + te_align_up((size_t)B * nslab * N * sizeof(float), 256))};
    const int norm = (flags & TE_ROLLOUT_NORMALISE) ? 1 : 0;
    const dim3 grid((unsigned)nslab, (unsigned)B), blk(kThreads);
    int cur = 0;
    // e_0^T M_{L-1}: only slab 0 has a non-zero row, the others write zeros (uniform fold in the next step)
#define TE_ROW_STEPS(MJ_)                                                                                           \
  do {                                                                                                             \
    rollout_row_step_kernel<true, MJ_><<<grid, blk, 0, stream>>>(cams + (L - 1) * mat, nullptr, pp[cur], (int)N, 0, \
                                                                 norm);                                            \
    for (int64_t i = L - 2; i >= start_layer; --i) {                                                               \
      rollout_row_step_kernel<false, MJ_><<<grid, blk, 0, stream>>>(cams + i * mat, pp[cur], pp[cur ^ 1], (int)N,   \
                                                                    nslab, norm);                                  \
      cur ^= 1;                                                                                                    \
    }                                                                                                              \
  } while (0)
    if (N <= 256) TE_ROW_STEPS(4);
    else if (N <= 640) TE_ROW_STEPS(10);
    else TE_ROW_STEPS(16);
#undef TE_ROW_STEPS
    rollout_row_finish_kernel<<<dim3((unsigned)B), blk, 0, stream>>>(pp[cur], joint, (int)N, nslab,
                                                                    (flags & TE_ROLLOUT_CLS_FIXUP) ? 1 : 0);
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
  if (!ws || ws_bytes < te_rollout_workspace_bytes(L, B, N)) return TE_ERR_WORKSPACE;
  float* M = (float*)ws;
  float* tmp = (float*)((char*)ws + te_align_up((size_t)L * mat * sizeof(float), 256));
  // only layers start..L-1 are consumed
  const int64_t rows = (L - start_layer) * B * N;
  rollout_prep_kernel<<<dim3((unsigned)te_ceil_div(rows, kThreads / 64)), dim3(kThreads), 0, stream>>>(
      cams + start_layer * mat, M + start_layer * mat, rows, N, (flags & TE_ROLLOUT_NORMALISE) ? 1 : 0);
  const int64_t steps = L - 1 - start_layer;
  // ping-pong so that the last product lands in `joint`
  const float* cur = M + start_layer * mat;
  dim3 grid((unsigned)te_ceil_div(N, TS), (unsigned)te_ceil_div(N, TS), (unsigned)B), blk(kThreads);
  if (steps == 0) {
    int64_t blocks = te_ceil_div(mat, (int64_t)kThreads * 4);
    if (blocks > 4096) blocks = 4096;
    rollout_copy_kernel<<<dim3((unsigned)blocks), blk, 0, stream>>>(cur, joint, mat);
  } else {
    for (int64_t s = 0; s < steps; ++s) {
      const int64_t i = start_layer + 1 + s;
      float* dst = ((steps - 1 - s) % 2 == 0) ? joint : tmp;
      if ((flags & TE_IMPL_SIMPLE) || te_attn_mfma::rollout_bmm_launch(M + i * mat, cur, dst, B, N, stream) != TE_OK)
        rollout_bmm_kernel<<<grid, blk, 0, stream>>>(M + i * mat, cur, dst, N);
      cur = dst;
    }
  }
  if (flags & TE_ROLLOUT_CLS_FIXUP)
    rollout_cls_fixup_kernel<<<dim3((unsigned)B), dim3(64), 0, stream>>>(joint, N);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
