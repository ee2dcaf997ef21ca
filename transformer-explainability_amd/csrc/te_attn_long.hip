// te_attn_long.hip -- attention producers for the sequence lengths the one-workgroup-per-head kernels of
// te_attn_rules.hip cannot hold (they keep k AND v of a head in LDS: N <= 224): ViT-L/16 at 384^2 (N = 577,
// baselines/ViT/ViT_LRP.py:132-152,419-425) and BERT (N = 512; separate q / k / v activations, scores / sqrt(D), additive
// mask, BERT_explainability/modules/BERT/BERT.py:307-365).  Head dim 64, N <= 640, any [B,H,N,64] strides.
//
//   forward   one workgroup per (b, h, 32 query rows): the [32, N] score tile lives in LDS --
//                z = q k^T (unscaled: the QK rule's Z)        keys staged 256 at a time, one 32x32 block per wave
//                x = z * scale (+ mask)                       optional output (BERT's Add.X[0])
//                attn = softmax(x)                            16 lanes per row, whole rows in registers
//                out = attn v                                 v staged transposed, one 16x16 block of [32, 64] per wave
//   backward  rows  (b, h, 32 query rows): d_attn = d_out v^T (the tensor save_attn_gradients receives), per-row
//                dot = sum_j d_attn . attn, d_s = ((d_attn - dot) . attn) * scale, d_q = d_s k
//             cols  one wave per (b, h, 32 keys): d_v = attn^T d_out, d_k = d_s^T q over all N query rows, operands
//                straight from global memory (the key block of a row is one 128-B line), accumulators in registers
//
// Round 6: for 64 < N <= 640 the entry points below dispatch to te_attn_fwd6l.hip (forward) and te_attn_bwd6l.hip (backward: the
// column side always, the row side when the caller hands over the block's forward output or needs no d_q / d_k) -- bf16 MFMAs with
// split operands, 1.4-2.2 x these kernels, which remain for N <= 64, for the backward without `out`, and as the comparison path of
// the A/B scripts (TE_ATTN_FWD_LONG / TE_ATTN_BWD_LONG / TE_ATTN_BWD_COLS = old in measurement builds).
//
// fp32 MFMAs (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): exact k-ordered fma chains; every reduction has a fixed order that
// depends on N only, so a batch equals its samples run one by one, bit for bit.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "te_common.h"

namespace te_attn_fwd6l {      // te_attn_fwd6l.hip: row-block owners on bf16 MFMAs, two walks over the keys (round 6) -- the default forward, 64 < N <= 640
bool supported(int64_t B, int64_t H, int64_t N, int64_t D);
int launch(const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn,
           const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn, const float* mask, float* z_qk, float* x_scaled, float* attn,
           float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, int64_t B, int64_t H, int64_t N, float scale, hipStream_t stream);
}  // namespace te_attn_fwd6l

namespace te_attn_bwd6l {      // te_attn_bwd6l.hip: the row side of the backward pass in the same structure (round 6), 64 < N <= 640
bool supported(int64_t B, int64_t H, int64_t N, int64_t D);
int launch_rows(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn, const float* out, int64_t o_sb, int64_t o_sh,
                int64_t o_sn, const float* k, int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb, int64_t v_sh,
                int64_t v_sn, const float* attn, float* d_attn, float* rowdot, float* d_q, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                int64_t B, int64_t H, int64_t N, float scale, int need_qk, hipStream_t stream);
int launch_cols(const float* attn, const float* d_attn, const float* rowdot, const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn, float* d_k,
                int64_t dk_sb, int64_t dk_sh, int64_t dk_sn, int64_t B, int64_t H, int64_t N, float scale, int need_qk, hipStream_t stream);
}  // namespace te_attn_bwd6l

namespace {

constexpr int TI = 32;          // query rows (or keys) per workgroup
constexpr int kT = 512;         // threads
constexpr int NMAX = 640;
constexpr int QLD = 66;         // row stride of [rows][64] MFMA operands read 4 bytes per lane: (2 r) mod 64 distinct banks
constexpr int VLD = 80;         // row stride of the [192][64] key-side operand of the 16x16x4 products: a lane group reads
                                // 4 keys x 16 consecutive features, (16 key + d) mod 64 distinct banks
constexpr int KG = 256;         // keys staged per pass of the score products
constexpr int KV = 192;         // keys staged per pass of the row products (P v, d_s k)

struct Strided {  // [B,H,N,64] view, last dim contiguous
  int64_t sb, sh, sn;
};

#define TE_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define TE_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int crow(int e, int kh) { return (e & 3) + 8 * (e >> 2) + 4 * kh; }
__device__ __forceinline__ int ldw_of(int N) { return ((N + 31) & ~31) + 4; }

// A group of <= 256 rows x 64 floats of a strided operand travels global -> registers (fetch_rows: 8 float4 per thread, the
// request of group g + 1 is issued before the products of group g) -> LDS [rows][LD] (commit_rows; LD = QLD for the score
// products' operands, VLD for the row products').  Rows past `valid` are zero.  Slot idx = t + 512 r <-> (row idx >> 4,
// float4 chunk idx & 15): 16 lanes read one 256-B row.
struct Group {
  f32x4 v[8];
};
template <int ROWS>
__device__ __forceinline__ void fetch_rows(Group& g, const float* __restrict__ src, int64_t sn, int valid) {
#pragma unroll
  for (int r = 0; r < ROWS / 32; ++r) {
    const int idx = threadIdx.x + r * kT;
    const int row = idx >> 4, c4 = idx & 15;
    g.v[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < valid) g.v[r] = *reinterpret_cast<const f32x4_u*>(src + (int64_t)row * sn + (c4 << 2));
  }
}
template <int ROWS, int LD>
__device__ __forceinline__ void commit_rows(float* __restrict__ dst, const Group& g) {
#pragma unroll
  for (int r = 0; r < ROWS / 32; ++r) {
    const int idx = threadIdx.x + r * kT;
    const int row = idx >> 4, c4 = idx & 15;
    if constexpr (LD % 4 == 0) {
      *reinterpret_cast<f32x4*>(dst + row * LD + (c4 << 2)) = g.v[r];
    } else {
      *reinterpret_cast<f32x2*>(dst + row * LD + (c4 << 2)) = f32x2{g.v[r][0], g.v[r][1]};
      *reinterpret_cast<f32x2*>(dst + row * LD + (c4 << 2) + 2) = f32x2{g.v[r][2], g.v[r][3]};
    }
  }
}
// the 32-row query-side tile (no prefetch needed: once per workgroup)
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const float* __restrict__ src, int64_t sn, int valid) {
  const int r = threadIdx.x >> 4, c4 = threadIdx.x & 15;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < valid) v = *reinterpret_cast<const f32x4_u*>(src + (int64_t)r * sn + (c4 << 2));
  *reinterpret_cast<f32x2*>(dst + r * QLD + (c4 << 2)) = f32x2{v[0], v[1]};
  *reinterpret_cast<f32x2*>(dst + r * QLD + (c4 << 2) + 2) = f32x2{v[2], v[3]};
}

// rowbuf[32][ldw] (+)= A[32][64] . Bkeys[keys][64]^T for the 32-key block `jb` of the staged group (one wave)
__device__ __forceinline__ void score_block(float* __restrict__ rowbuf, int ldw, int col0, const float* __restrict__ At,
                                            const float* __restrict__ Kt, int jb, int lane) {
  const int lr = lane & 31, kh = lane >> 5;
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.0f;
  const float* ap = At + lr * QLD + kh;
  const float* bp = Kt + (jb * 32 + lr) * QLD + kh;
#pragma unroll
  for (int s = 0; s < 32; ++s) z = TE_MFMA32(ap[2 * s], bp[2 * s], z);
#pragma unroll
  for (int e = 0; e < 16; ++e) rowbuf[crow(e, kh) * ldw + col0 + jb * 32 + lr] = z[e];
}

// acc (16x16 block (ib, db) of a [32, 64] output) += rowbuf[32][col0 .. col0 + 4 nk4) . V  (V = [keys][VLD] of the group)
__device__ __forceinline__ f32x4 row_product(f32x4 acc, const float* __restrict__ rowbuf, int ldw, int col0,
                                             const float* __restrict__ Vt, int nk4, int ib, int db, int lane) {
  const int l15 = lane & 15, kq = lane >> 4;
  const float* ap = rowbuf + (ib * 16 + l15) * ldw + col0 + kq;
  const float* bp = Vt + kq * VLD + db * 16 + l15;
  for (int s = 0; s < nk4; ++s) acc = TE_MFMA16(ap[4 * s], bp[4 * s * VLD], acc);
  return acc;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void attn_fwd_rows_kernel(const float* __restrict__ q, Strided qs,
                                                           const float* __restrict__ k, Strided ks,
                                                           const float* __restrict__ v, Strided vs,
                                                           const float* __restrict__ mask, float* __restrict__ zqk,
                                                           float* __restrict__ xsc, float* __restrict__ attn,
                                                           float* __restrict__ out, Strided os, int H, int N, int ntile,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldw = ldw_of(N);
  float* rowbuf = smem;                       // [TI][ldw]
  float* Qt = rowbuf + TI * ldw;              // [TI][QLD]
  float* St = Qt + TI * QLD;                  // [KG][QLD] keys, then [KV][VLD] values
  const int bh = blockIdx.x / ntile, it = blockIdx.x - bh * ntile, b = bh / H, h = bh - b * H;
  const int i0 = it * TI;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* q_bh = q + b * qs.sb + h * qs.sh;
  const float* k_bh = k + b * ks.sb + h * ks.sh;
  const float* v_bh = v + b * vs.sb + h * vs.sh;
  const int ng = (N + KG - 1) / KG;
  const int nv = (N + KV - 1) / KV;
  Group grp;
  fetch_rows<KG>(grp, k_bh, ks.sn, min(KG, N));
  stage_tile(Qt, q_bh + (int64_t)i0 * qs.sn, qs.sn, N - i0);
  for (int g = 0; g < ng; ++g) {
    const int j0 = g * KG, nk = min(KG, N - j0), nk32 = (nk + 31) & ~31;
    __syncthreads();
    commit_rows<KG, QLD>(St, grp);
    __syncthreads();
    // the next group's keys -- or the first group's values -- are in flight during the products / the softmax
    if (g + 1 < ng) fetch_rows<KG>(grp, k_bh + (int64_t)(j0 + KG) * ks.sn, ks.sn, min(KG, N - j0 - KG));
    else fetch_rows<KV>(grp, v_bh, vs.sn, min(KV, N));
    if (wave * 32 < nk32) score_block(rowbuf, ldw, j0, Qt, St, wave, lane);
  }
  __syncthreads();
  {
    // row softmax: 16 lanes per row, float4 chunks c = sub + 16 m of the row
    const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int i = i0 + row;
    const bool row_ok = i < N;
    const int nch = (ldw - 4) >> 2;                       // float4 chunks of the padded row (multiple of 8)
    float* rb = rowbuf + row * ldw;
    const int64_t rowoff = ((int64_t)bh * N + (row_ok ? i : 0)) * N;
    const float* mrow = mask ? mask + (int64_t)b * N : nullptr;
    constexpr int MCH = NMAX / 64;                        // chunks per lane
    f32x4 x[MCH];
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < MCH; ++m) {
      const int c = sub + 16 * m;
      x[m] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (c < nch) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(rb + (c << 2));
        const int j = c << 2;
        f32x4 xs, xu;         // softmax input (scaled + mask); the scaled scores alone = Add.X[0] (BERT.py:339-342)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = z[e] * scale;                                          // 'dots * self.scale' / 'scores / sqrt(D)'
          xu[e] = t;
          if (mrow && j + e < N) t = t + mrow[j + e];                      // BERT.py:341-342
          xs[e] = t;
        }
        if (row_ok) {
          if (j + 3 < N) {
            if (zqk) *reinterpret_cast<f32x4_u*>(zqk + rowoff + j) = z;
            if (xsc) *reinterpret_cast<f32x4_u*>(xsc + rowoff + j) = xu;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < N) {
                if (zqk) zqk[rowoff + j + e] = z[e];
                if (xsc) xsc[rowoff + j + e] = xu[e];
              }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[m][e] = (j + e < N) ? xs[e] : -INFINITY;
          mx = fmaxf(mx, x[m][e]);
        }
      }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.0f;
#pragma unroll
    for (int m = 0; m < MCH; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[m][e] = expf(x[m][e] - mx);                    // exp(-inf) = 0 for padding
        sum = sum + x[m][e];
      }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) sum = sum + __shfl_xor(sum, off, 64);
#pragma unroll
    for (int m = 0; m < MCH; ++m) {
      const int c = sub + 16 * m;
      if (c < nch) {
        const int j = c << 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[m][e] = x[m][e] / sum;
        *reinterpret_cast<f32x4*>(rb + j) = x[m];
        if (row_ok) {
          if (j + 3 < N) {
            *reinterpret_cast<f32x4_u*>(attn + rowoff + j) = x[m];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < N) attn[rowoff + j + e] = x[m][e];
          }
        }
      }
    }
  }
  // out tile [32, 64] = P v : 16x16 block (ib, db) per wave
  const int ib = wave >> 2, db = wave & 3;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < nv; ++g) {
    const int j0 = g * KV, nk = min(KV, N - j0), nk32 = (nk + 31) & ~31;
    __syncthreads();
    commit_rows<KV, VLD>(St, grp);
    __syncthreads();
    if (g + 1 < nv) fetch_rows<KV>(grp, v_bh + (int64_t)(j0 + KV) * vs.sn, vs.sn, min(KV, N - j0 - KV));
    o = row_product(o, rowbuf, ldw, j0, St, nk32 >> 2, ib, db, lane);
  }
  float* o_bh = out + b * os.sb + h * os.sh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + ib * 16 + (lane >> 4) * 4 + r;
    if (i < N) o_bh[(int64_t)i * os.sn + db * 16 + (lane & 15)] = o[r];
  }
}

// ------------------------------------------------------------------------------------------------
// backward, row side: d_attn = d_out v^T ; rowdot ; d_q = (((d_attn - rowdot) . attn) * scale) k
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void attn_bwd_rows_kernel(const float* __restrict__ dout, Strided dos,
                                                           const float* __restrict__ k, Strided ks,
                                                           const float* __restrict__ v, Strided vs,
                                                           const float* __restrict__ attn, float* __restrict__ dattn,
                                                           float* __restrict__ rowdot, float* __restrict__ dq, Strided dqs,
                                                           int H, int N, int ntile, float scale, int need_qk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldw = ldw_of(N);
  float* rowbuf = smem;
  float* Qt = rowbuf + TI * ldw;
  float* St = Qt + TI * QLD;
  const int bh = blockIdx.x / ntile, it = blockIdx.x - bh * ntile, b = bh / H, h = bh - b * H;
  const int i0 = it * TI;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* k_bh = k + b * ks.sb + h * ks.sh;
  const float* v_bh = v + b * vs.sb + h * vs.sh;
  const int ng = (N + KG - 1) / KG;
  const int nv = (N + KV - 1) / KV;
  Group grp;
  fetch_rows<KG>(grp, v_bh, vs.sn, min(KG, N));
  stage_tile(Qt, dout + b * dos.sb + h * dos.sh + (int64_t)i0 * dos.sn, dos.sn, N - i0);
  for (int g = 0; g < ng; ++g) {
    const int j0 = g * KG, nk = min(KG, N - j0), nk32 = (nk + 31) & ~31;
    __syncthreads();
    commit_rows<KG, QLD>(St, grp);
    __syncthreads();
    if (g + 1 < ng) fetch_rows<KG>(grp, v_bh + (int64_t)(j0 + KG) * vs.sn, vs.sn, min(KG, N - j0 - KG));
    else if (need_qk) fetch_rows<KV>(grp, k_bh, ks.sn, min(KV, N));
    if (wave * 32 < nk32) score_block(rowbuf, ldw, j0, Qt, St, wave, lane);
  }
  __syncthreads();
  {
    const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int i = i0 + row;
    const bool row_ok = i < N;
    const int nch = (ldw - 4) >> 2;
    float* rb = rowbuf + row * ldw;
    const int64_t rowoff = ((int64_t)bh * N + (row_ok ? i : 0)) * N;
    constexpr int MCH = NMAX / 64;
    f32x4 da[MCH], pa[MCH];
    float dot = 0.0f;
#pragma unroll
    for (int m = 0; m < MCH; ++m) {
      const int c = sub + 16 * m;
      da[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      pa[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) {
        const int j = c << 2;
        da[m] = *reinterpret_cast<const f32x4*>(rb + j);
        if (row_ok) {
          if (j + 3 < N) {
            *reinterpret_cast<f32x4_u*>(dattn + rowoff + j) = da[m];
            if (need_qk) pa[m] = *reinterpret_cast<const f32x4_u*>(attn + rowoff + j);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j + e < N) {
                dattn[rowoff + j + e] = da[m][e];
                if (need_qk) pa[m][e] = attn[rowoff + j + e];
              }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dot = dot + da[m][e] * pa[m][e];       // padding: pa = 0
      }
    }
    if (need_qk) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) dot = dot + __shfl_xor(dot, off, 64);
      if (sub == 0 && row_ok) rowdot[(int64_t)bh * N + i] = dot;
#pragma unroll
      for (int m = 0; m < MCH; ++m) {
        const int c = sub + 16 * m;
        if (c < nch) {
          f32x4 ds;
#pragma unroll
          for (int e = 0; e < 4; ++e) ds[e] = ((da[m][e] - dot) * pa[m][e]) * scale;      // softmax backward, then '* scale'
          *reinterpret_cast<f32x4*>(rb + (c << 2)) = ds;
        }
      }
    }
  }
  if (!need_qk) return;
  const int ib = wave >> 2, db = wave & 3;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < nv; ++g) {
    const int j0 = g * KV, nk = min(KV, N - j0), nk32 = (nk + 31) & ~31;
    __syncthreads();
    commit_rows<KV, VLD>(St, grp);
    __syncthreads();
    if (g + 1 < nv) fetch_rows<KV>(grp, k_bh + (int64_t)(j0 + KV) * ks.sn, ks.sn, min(KV, N - j0 - KV));
    o = row_product(o, rowbuf, ldw, j0, St, nk32 >> 2, ib, db, lane);
  }
  float* o_bh = dq + b * dqs.sb + h * dqs.sh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + ib * 16 + (lane >> 4) * 4 + r;
    if (i < N) o_bh[(int64_t)i * dqs.sn + db * 16 + (lane & 15)] = o[r];
  }
}

// ------------------------------------------------------------------------------------------------
// backward, column side: d_v[32 keys, 64] = attn^T d_out ; d_k = d_s^T q, d_s recomputed from attn, d_attn, rowdot.
// One WAVE per (b, h, 32 keys): it walks all N query rows (two per MFMA step) with its operands straight from global
// memory -- the key block of a row of attn / d_attn is one 128-B line, d_out / q rows are shared by every wave of the
// head through L2 -- and keeps the four 32x32 accumulators in registers: no LDS, no barrier, no cross-wave sum; the
// latency of the loads is covered by the other waves of the SIMD (4-5 resident).
// ------------------------------------------------------------------------------------------------
constexpr int kTC = 256;        // threads of the column-side kernel: four independent waves
__global__ __launch_bounds__(kTC) void attn_bwd_cols_kernel(const float* __restrict__ attn, const float* __restrict__ dattn,
                                                            const float* __restrict__ rowdot,
                                                            const float* __restrict__ dout, Strided dos,
                                                            const float* __restrict__ q, Strided qs, float* __restrict__ dv,
                                                            Strided dvs, float* __restrict__ dk, Strided dks, int H, int N,
                                                            int nwg, float scale, int need_qk) {
  const int bh = blockIdx.x / nwg, jt = (blockIdx.x - bh * nwg) * 4 + (threadIdx.x >> 6), b = bh / H, h = bh - b * H;
  const int j0 = jt * TI;
  if (j0 >= N) return;
  const int lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const float* a_bh = attn + (int64_t)bh * N * N + j0 + lr;
  const float* g_bh = dattn + (int64_t)bh * N * N + j0 + lr;
  const float* rd = rowdot + (int64_t)bh * N;
  const float* do_bh = dout + b * dos.sb + h * dos.sh + lr;
  const float* q_bh = q + b * qs.sb + h * qs.sh + lr;
  f32x16 av[2], ak[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      av[d][e] = 0.0f;
      ak[d][e] = 0.0f;
    }
  const bool key_ok = j0 + lr < N;
  const int npair = (N + 1) >> 1;
  // U steps of operands are requested before the first of their MFMAs: the loop is bound by the latency of its loads
  // otherwise (measured: 29 % of the MFMA rate with two steps in flight)
  constexpr int U = 6;
  for (int p0 = 0; p0 < npair; p0 += U) {
    float pa[U], b0[U], b1[U], ga[U], rr[U], q0[U], q1[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = 2 * (p0 + u) + kh;
      const bool row_ok = i < N;
      ok[u] = row_ok && key_ok;
      const int64_t ic = row_ok ? i : 0;
      pa[u] = a_bh[ic * N];
      b0[u] = do_bh[ic * dos.sn];
      b1[u] = do_bh[ic * dos.sn + 32];
      if (need_qk) {
        ga[u] = g_bh[ic * N];
        rr[u] = rd[ic];
        q0[u] = q_bh[ic * qs.sn];
        q1[u] = q_bh[ic * qs.sn + 32];
      }
      if (!row_ok) b0[u] = b1[u] = 0.0f;
      if (!row_ok && need_qk) q0[u] = q1[u] = 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pv = ok[u] ? pa[u] : 0.0f;
      av[0] = TE_MFMA32(pv, b0[u], av[0]);
      av[1] = TE_MFMA32(pv, b1[u], av[1]);
      if (need_qk) {
        const float ds = ok[u] ? ((ga[u] - rr[u]) * pv) * scale : 0.0f;
        ak[0] = TE_MFMA32(ds, q0[u], ak[0]);
        ak[1] = TE_MFMA32(ds, q1[u], ak[1]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = j0 + crow(e, kh), col = d * 32 + lr;
      if (key < N) {
        dv[b * dvs.sb + h * dvs.sh + (int64_t)key * dvs.sn + col] = av[d][e];
        if (need_qk) dk[b * dks.sb + h * dks.sh + (int64_t)key * dks.sn + col] = ak[d][e];
      }
    }
}

inline size_t lds_rows(int64_t N) {
  const int ldw = ((int)((N + 31) & ~31)) + 4;
  const size_t stage = (size_t)std::max(KG * QLD, KV * VLD);
  return ((size_t)TI * ldw + (size_t)TI * QLD + stage) * sizeof(float);
}

template <typename K>
inline void allow_lds(K kern, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline bool strides_ok(int64_t sb, int64_t sh, int64_t sn) { return sb >= 0 && sh >= 0 && sn >= 64 && (sn % 4) == 0 && (sh % 4) == 0 && (sb % 4) == 0; }

}  // namespace

extern "C" int te_attention_strided_supported(int64_t N, int64_t D) { return (D == 64 && N >= 1 && N <= NMAX) ? 1 : 0; }

extern "C" size_t te_attention_backward_strided_workspace_bytes(int64_t B, int64_t H, int64_t N) {
  return te_align_up((size_t)(B * H * N) * sizeof(float), 256);
}

extern "C" int te_attention_forward_strided_f32(const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k,
                                                int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb,
                                                int64_t v_sh, int64_t v_sn, const float* mask, float* z_qk,
                                                float* x_scaled, float* attn, float* out, int64_t o_sb, int64_t o_sh,
                                                int64_t o_sn, int64_t B, int64_t H, int64_t N, int64_t D, float scale,
                                                te_stream_t stream_) {
  if (!q || !k || !v || !attn || !out || B <= 0 || H <= 0 || N <= 0) return TE_ERR_INVALID_ARG;
  const int64_t ntile = te_ceil_div(N, TI);
  if (!te_attention_strided_supported(N, D) || B * H * ntile > 0x7fffffff) return TE_ERR_UNSUPPORTED;
  if (!strides_ok(q_sb, q_sh, q_sn) || !strides_ok(k_sb, k_sh, k_sn) || !strides_ok(v_sb, v_sh, v_sn) ||
      !strides_ok(o_sb, o_sh, o_sn))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
#ifdef TE_STUDY      // TE_ATTN_FWD_LONG=old selects the round-3 kernel in measurement builds for same-box A/B runs
  static const bool old_fwd = [] { const char* e = getenv("TE_ATTN_FWD_LONG"); return e && !strcmp(e, "old"); }();
#else
  constexpr bool old_fwd = false;
#endif
  if (!old_fwd && te_attn_fwd6l::supported(B, H, N, D)) {
    const int rc = te_attn_fwd6l::launch(q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, v, v_sb, v_sh, v_sn, mask, z_qk, x_scaled, attn,
                                         out, o_sb, o_sh, o_sn, B, H, N, scale, stream);
    if (rc != TE_OK) return rc;
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
  allow_lds(attn_fwd_rows_kernel, lds_rows(NMAX));
  attn_fwd_rows_kernel<<<dim3((unsigned)(B * H * ntile)), dim3(kT), lds_rows(N), stream>>>(
      q, Strided{q_sb, q_sh, q_sn}, k, Strided{k_sb, k_sh, k_sn}, v, Strided{v_sb, v_sh, v_sn}, mask, z_qk, x_scaled, attn,
      out, Strided{o_sb, o_sh, o_sn}, (int)H, (int)N, (int)ntile, scale);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// out (optional): the forward output attention_forward returned for these inputs ([B,H,N,64] view).  With it -- or with
// need_qk = 0, where no row sum is needed -- the row side runs on te_attn_bwd6l.hip (row sums from d_out . out); without it
// on attn_bwd_rows_kernel (row sums from d_attn . attn).
static int backward_strided(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn, const float* out, int64_t o_sb,
                            int64_t o_sh, int64_t o_sn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k,
                            int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb, int64_t v_sh, int64_t v_sn,
                            const float* attn, float* d_attn, float* d_q, int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, float* d_k,
                            int64_t dk_sb, int64_t dk_sh, int64_t dk_sn, float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                            int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk, void* ws, size_t ws_bytes,
                            te_stream_t stream_) {
  if (!d_out || !k || !v || !attn || !d_attn || !d_v || B <= 0 || H <= 0 || N <= 0) return TE_ERR_INVALID_ARG;
  if (need_qk && (!q || !d_q || !d_k)) return TE_ERR_INVALID_ARG;
  const int64_t ntile = te_ceil_div(N, TI);
  if (!te_attention_strided_supported(N, D) || B * H * ntile > 0x7fffffff) return TE_ERR_UNSUPPORTED;
  if (!strides_ok(do_sb, do_sh, do_sn) || !strides_ok(k_sb, k_sh, k_sn) || !strides_ok(v_sb, v_sh, v_sn) ||
      !strides_ok(dv_sb, dv_sh, dv_sn) || (out && !strides_ok(o_sb, o_sh, o_sn)))
    return TE_ERR_UNSUPPORTED;
  if (need_qk && (!strides_ok(q_sb, q_sh, q_sn) || !strides_ok(dq_sb, dq_sh, dq_sn) || !strides_ok(dk_sb, dk_sh, dk_sn)))
    return TE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < te_attention_backward_strided_workspace_bytes(B, H, N)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* rowdot = (float*)ws;
#ifdef TE_STUDY      // TE_ATTN_BWD_LONG=old selects the round-3 row kernel in measurement builds for same-box A/B runs
  static const bool old_rows = [] { const char* e = getenv("TE_ATTN_BWD_LONG"); return e && !strcmp(e, "old"); }();
#else
  constexpr bool old_rows = false;
#endif
  if (!old_rows && (out || !need_qk) && te_attn_bwd6l::supported(B, H, N, D)) {
    const int rc = te_attn_bwd6l::launch_rows(d_out, do_sb, do_sh, do_sn, out, o_sb, o_sh, o_sn, k, k_sb, k_sh, k_sn, v, v_sb, v_sh, v_sn,
                                              attn, d_attn, rowdot, d_q, dq_sb, dq_sh, dq_sn, B, H, N, scale, need_qk ? 1 : 0, stream);
    if (rc != TE_OK) return rc;
  } else {
    allow_lds(attn_bwd_rows_kernel, lds_rows(NMAX));
    attn_bwd_rows_kernel<<<dim3((unsigned)(B * H * ntile)), dim3(kT), lds_rows(N), stream>>>(
        d_out, Strided{do_sb, do_sh, do_sn}, k, Strided{k_sb, k_sh, k_sn}, v, Strided{v_sb, v_sh, v_sn}, attn, d_attn, rowdot,
        d_q, Strided{dq_sb, dq_sh, dq_sn}, (int)H, (int)N, (int)ntile, scale, need_qk ? 1 : 0);
  }
#ifdef TE_STUDY      // TE_ATTN_BWD_COLS=old selects the round-3 column kernel in measurement builds
  static const bool old_cols = [] { const char* e = getenv("TE_ATTN_BWD_COLS"); return e && !strcmp(e, "old"); }();
#else
  constexpr bool old_cols = false;
#endif
  if (!old_cols && te_attn_bwd6l::supported(B, H, N, D)) {
    const int rc = te_attn_bwd6l::launch_cols(attn, d_attn, rowdot, d_out, do_sb, do_sh, do_sn, q, q_sb, q_sh, q_sn, d_v, dv_sb, dv_sh, dv_sn,
                                              d_k, dk_sb, dk_sh, dk_sn, B, H, N, scale, need_qk ? 1 : 0, stream);
    if (rc != TE_OK) return rc;
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
  const int64_t nwg = te_ceil_div(ntile, 4);       // four key blocks (waves) per workgroup
  attn_bwd_cols_kernel<<<dim3((unsigned)(B * H * nwg)), dim3(kTC), 0, stream>>>(
      attn, d_attn, rowdot, d_out, Strided{do_sb, do_sh, do_sn}, q, Strided{q_sb, q_sh, q_sn}, d_v,
      Strided{dv_sb, dv_sh, dv_sn}, d_k, Strided{dk_sb, dk_sh, dk_sn}, (int)H, (int)N, (int)nwg, scale, need_qk ? 1 : 0);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_attention_backward_strided_f32(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                                                 const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k,
                                                 int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb,
                                                 int64_t v_sh, int64_t v_sn, const float* attn, float* d_attn, float* d_q,
                                                 int64_t dq_sb, int64_t dq_sh, int64_t dq_sn, float* d_k, int64_t dk_sb,
                                                 int64_t dk_sh, int64_t dk_sn, float* d_v, int64_t dv_sb, int64_t dv_sh,
                                                 int64_t dv_sn, int64_t B, int64_t H, int64_t N, int64_t D, float scale,
                                                 int need_qk, void* ws, size_t ws_bytes, te_stream_t stream_) {
  return backward_strided(d_out, do_sb, do_sh, do_sn, nullptr, 0, 0, 0, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, v, v_sb, v_sh, v_sn,
                          attn, d_attn, d_q, dq_sb, dq_sh, dq_sn, d_k, dk_sb, dk_sh, dk_sn, d_v, dv_sb, dv_sh, dv_sn, B, H, N, D, scale,
                          need_qk, ws, ws_bytes, stream_);
}

extern "C" int te_attention_backward_strided_out_f32(const float* d_out, int64_t do_sb, int64_t do_sh, int64_t do_sn,
                                                     const float* out, int64_t o_sb, int64_t o_sh, int64_t o_sn, const float* q,
                                                     int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb,
                                                     int64_t k_sh, int64_t k_sn, const float* v, int64_t v_sb, int64_t v_sh,
                                                     int64_t v_sn, const float* attn, float* d_attn, float* d_q, int64_t dq_sb,
                                                     int64_t dq_sh, int64_t dq_sn, float* d_k, int64_t dk_sb, int64_t dk_sh,
                                                     int64_t dk_sn, float* d_v, int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                                                     int64_t B, int64_t H, int64_t N, int64_t D, float scale, int need_qk, void* ws,
                                                     size_t ws_bytes, te_stream_t stream_) {
  if (!out) return TE_ERR_INVALID_ARG;
  return backward_strided(d_out, do_sb, do_sh, do_sn, out, o_sb, o_sh, o_sn, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, v, v_sb, v_sh,
                          v_sn, attn, d_attn, d_q, dq_sb, dq_sh, dq_sn, d_k, dk_sb, dk_sh, dk_sn, d_v, dv_sb, dv_sh, dv_sn, B, H, N, D,
                          scale, need_qk, ws, ws_bytes, stream_);
}
