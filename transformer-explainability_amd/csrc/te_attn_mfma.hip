// te_attn_mfma.hip -- LDS-tiled fp32-MFMA kernels for the attention einsum / MatMul relprop rules
// (modules/layers_ours.py:48-60,122-127) on gfx950, head dim D = 64.
//
//   AV rule:  Z = attn v ; S = sd(R, Z) ; cam_attn = attn .(S v^T) ; cam_v = v .(attn^T S)
//   QK rule:  Z = q k^T  ; S = sd(R, Z) ; cam_q = q .(S k)         ; cam_k = k .(S^T q)
//
// Z is the FORWARD output of the very product whose rule is evaluated (the einsum / MatMul module caches it as
// self.Y, forward_hook layers_ours.py:16-27; the reference's autograd re-evaluates the same einsum and gets the same
// bits), so the rule kernels take it as an input: each rule is then two products, not three, and S agrees with the
// forward pass to the last bit of Z.  Callers without a cached Z get it from z_av_kernel / z_qk_kernel first.
//
// Kernels (256 threads = 4 waves as 2 x 2, one 64 x 64 output tile per block per product, v_mfma_f32_32x32x2_f32):
//   av_row  (b,h, 64 query rows):  S = sd(R, Z) -> LDS + workspace; per 64-key chunk G = S v^T -> cam_attn = attn . G
//   qk_row  (b,h, 64 query rows):  per 64-key chunk S = sd(R_nn, Z) -> LDS + workspace; cam_q += S_chunk k_chunk
//   col     (b,h, 64 key columns): out = X .(M^T Y) over all query rows; cam_v (M = attn, Y = S, X = v) and
//                                  cam_k (M = S of the QK rule, Y = q, X = k)
//
// Every tile is [64][64] floats in LDS with the 16-B chunks of a row XOR-swizzled by (row & 15): one image serves as a
// K-contiguous operand (ds_read_b128: four consecutive k per lane, conflict-free over the 16-lane groups) and as a
// k-major operand (ds_read_b32 of [k][x], conflict-free over 32 consecutive x).  Global traffic moves as 16-B
// accesses that only assume dword alignment (attention rows are N = 197 floats long); the next chunk's tiles are in
// flight in registers while the current chunk's MFMAs and epilogue run.  Strided [B,H,N,D] operands are read in
// place (fused qkv layout).  Blocks of one (b,h) are blockIdx = tile * BH + bh apart: with BH a multiple of 8 they
// land on one XCD and share that L2's copy of k / v / S.
#include "te_common.h"

namespace te_attn_rules {   // te_attn_rules.hip: the one-pass rule kernels (default)
bool enabled();
bool supported(int64_t B, int64_t H, int64_t N, int64_t D);
int av_launch(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn, const float* attn, const float* v, int64_t v_sb,
              int64_t v_sh, int64_t v_sn, const float* Z, int64_t z_sb, int64_t z_sh, int64_t z_sn, float* cam_attn,
              float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H, int64_t N, float scale,
              hipStream_t stream);
int qk_launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k, int64_t k_sb,
              int64_t k_sh, int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh, int64_t cq_sn,
              float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N, float scale,
              float* qpart, const float* r_scale, int64_t r_scale_stride, hipStream_t stream);
}  // namespace te_attn_rules

namespace te_attn_mfma {

namespace {

constexpr int TS = 64;        // tile side
constexpr int kThreads = 256;

struct Strided {  // [B,H,N,D] view, D contiguous
  int64_t sb, sh, sn;
};

// 16-byte access that only promises 4-byte alignment (legal for gfx950 global loads / stores)
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

#define TE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int swz(int row, int chunk) { return row * TS + ((chunk ^ (row & 15)) << 2); }
// row of accumulator element e inside the wave's 32 x 32 block (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int crow(int e, int kh) { return (e & 3) + 8 * (e >> 2) + 4 * kh; }

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
  for (int e = 0; e < 16; ++e) a[e] = 0.0f;
}

// guarded 4-wide access at a dword-aligned address: elements [c, c+4) of a row with `cols_valid` valid columns
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int c, int cols_valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (c + 3 < cols_valid) {
    v = *reinterpret_cast<const f32x4_u*>(p + c);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < cols_valid) v[e] = p[c + e];
  }
  return v;
}
__device__ __forceinline__ void store4(float* __restrict__ p, int c, int cols_valid, f32x4 v) {
  if (c + 3 < cols_valid) {
    *reinterpret_cast<f32x4_u*>(p + c) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < cols_valid) p[c + e] = v[e];
  }
}

// One 64 x 64 tile in flight: thread t holds elements (row = idx >> 4, cols 4 * (idx & 15) ..+3), idx = t + 256 i.
struct TileRegs {
  f32x4 v[4];
};
__device__ __forceinline__ void load_tile(TileRegs& t, const float* __restrict__ src, int64_t ld, int rows_valid,
                                          int cols_valid) {
  if (rows_valid == TS && cols_valid == TS) {     // full tile (block-uniform): four unguarded loads issued together
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      t.v[i] = *reinterpret_cast<const f32x4_u*>(src + (int64_t)(idx >> 4) * ld + ((idx & 15) << 2));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int row = idx >> 4, c = (idx & 15) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < rows_valid) v = load4(src + (int64_t)row * ld, c, cols_valid);
    t.v[i] = v;
  }
}
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const TileRegs& t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    *reinterpret_cast<f32x4*>(lds + swz(idx >> 4, idx & 15)) = t.v[i];
  }
}

// operand fragments.  K-contiguous tile: `row` is the operand index (m or n); returns k = kg*8 + kh*4 + {0..3}.
__device__ __forceinline__ f32x4 frag_kc(const float* __restrict__ T, int row, int kg, int kh) {
  return *reinterpret_cast<const f32x4*>(T + swz(row, kg * 2 + kh));
}
// k-major tile [k][x]
__device__ __forceinline__ float frag_km(const float* __restrict__ T, int k, int x) {
  return T[swz(k, x >> 2) + (x & 3)];
}

// acc(32 x 32 block (wm, wn)) += A B over K = 64;  A, B each K-contiguous (kc) or k-major (km) 64 x 64 LDS tiles
template <bool A_KM, bool B_KM>
__device__ __forceinline__ void mma64(f32x16& acc, const float* __restrict__ At, const float* __restrict__ Bt,
                                      int wm, int wn, int lr, int kh) {
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) {
    f32x4 a, b;
    if constexpr (!A_KM) a = frag_kc(At, wm * 32 + lr, kg, kh);
    if constexpr (!B_KM) b = frag_kc(Bt, wn * 32 + lr, kg, kh);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kg * 8 + kh * 4 + j;
      const float av = A_KM ? frag_km(At, k, wm * 32 + lr) : a[j];
      const float bv = B_KM ? frag_km(Bt, k, wn * 32 + lr) : b[j];
      acc = TE_MFMA(av, bv, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// AV rule, query-row side.  R strided [B,H,N,64]; Z, Sws contiguous [B*H,N,64]; attn, cam_attn contiguous [B*H,N,N]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void av_row_kernel(
    const float* __restrict__ R, Strided rs, const float* __restrict__ Z, const float* __restrict__ attn,
    const float* __restrict__ v, Strided vs, float* __restrict__ cam_attn, float* __restrict__ Sws, int H, int N, int BH,
    float scale) {
  __shared__ __attribute__((aligned(16))) float St[TS * TS];
  __shared__ __attribute__((aligned(16))) float Vt[TS * TS];
  const int bh = blockIdx.x % BH, rt = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, rows_valid = min(TS, N - row0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool row_active = (wm * 32) < rows_valid;
  const float* v_bh = v + (int64_t)b * vs.sb + (int64_t)h * vs.sh;
  const float* r_bh = R + (int64_t)b * rs.sb + (int64_t)h * rs.sh;
  const int nch = (N + TS - 1) / TS;

  TileRegs tv;
  load_tile(tv, v_bh, vs.sn, min(TS, N), TS);
  // S = sd(R, Z) for this stripe: into LDS (A operand of S v^T) and the workspace (cam_v kernel)
  if (rows_valid == TS) {            // full stripe: the eight loads go out before the first division
    f32x4 r[4], z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int row = idx >> 4, c = (idx & 15) << 2;
      r[i] = *reinterpret_cast<const f32x4_u*>(r_bh + (int64_t)(row0 + row) * rs.sn + c);
      z[i] = *reinterpret_cast<const f32x4_u*>(Z + ((int64_t)bh * N + row0 + row) * TS + c);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int row = idx >> 4, c = (idx & 15) << 2;
      f32x4 s;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = te_sd(r[i][e], z[i][e]);
      *reinterpret_cast<f32x4_u*>(Sws + ((int64_t)bh * N + row0 + row) * TS + c) = s;
      *reinterpret_cast<f32x4*>(St + swz(row, idx & 15)) = s;
    }
  } else
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int row = idx >> 4, c = (idx & 15) << 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (row < rows_valid) {
      const int64_t zoff = ((int64_t)bh * N + row0 + row) * TS + c;
      const f32x4 r = *reinterpret_cast<const f32x4_u*>(r_bh + (int64_t)(row0 + row) * rs.sn + c);
      const f32x4 z = *reinterpret_cast<const f32x4_u*>(Z + zoff);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = te_sd(r[e], z[e]);
      *reinterpret_cast<f32x4_u*>(Sws + zoff) = s;
    }
    *reinterpret_cast<f32x4*>(St + swz(row, idx & 15)) = s;
  }
  // cam_attn = attn . (S v^T) * scale, one 64-key chunk at a time
  for (int c = 0; c < nch; ++c) {
    const int kc = min(TS, N - c * TS);
    __syncthreads();                 // the previous chunk's reads of Vt are done
    store_tile(Vt, tv);              // rows = keys (n), K = d
    __syncthreads();
    if (c + 1 < nch) load_tile(tv, v_bh + (int64_t)(c + 1) * TS * vs.sn, vs.sn, min(TS, N - (c + 1) * TS), TS);
    if (row_active && (wn * 32) < kc) {
      f32x16 g;
      zero(g);
      mma64<false, false>(g, St, Vt, wm, wn, lr, kh);
      const int gj = c * TS + wn * 32 + lr;
      if (rows_valid == TS && kc == TS) {
        // full tile: all 16 attention values of the block requested before the first is used (the guarded form below
        // issues every load behind its own bounds test)
        const int64_t base = ((int64_t)bh * N + row0 + wm * 32) * N + gj;
        float av[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) av[e] = attn[base + (int64_t)crow(e, kh) * N];
#pragma unroll
        for (int e = 0; e < 16; ++e) cam_attn[base + (int64_t)crow(e, kh) * N] = (av[e] * g[e]) * scale;
      } else if (gj < N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int gi = row0 + wm * 32 + crow(e, kh);
          if (gi < N) {
            const int64_t off = ((int64_t)bh * N + gi) * N + gj;
            cam_attn[off] = (attn[off] * g[e]) * scale;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// out[j,d] = X[j,d] * (sum_i M[i,j] Y[i,d]) * scale       (M contiguous [BH,N,N]; X, Y, out strided)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void col_kernel(
    const float* __restrict__ M, const float* __restrict__ Y, Strided ys, const float* __restrict__ X,
    Strided xs, float* __restrict__ out, Strided os, int H, int N, int BH, float scale) {
  __shared__ __attribute__((aligned(16))) float Mt[TS * TS];
  __shared__ __attribute__((aligned(16))) float Yt[TS * TS];
  const int bh = blockIdx.x % BH, ct = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int col0 = ct * TS, cols_valid = min(TS, N - col0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool active = (wm * 32) < cols_valid;
  const float* m_bh = M + (int64_t)bh * N * N + col0;
  const float* y_bh = Y + (int64_t)b * ys.sb + (int64_t)h * ys.sh;
  const int nch = (N + TS - 1) / TS;
  f32x16 acc;
  zero(acc);
  TileRegs tm, ty;
  load_tile(tm, m_bh, N, min(TS, N), cols_valid);      // [k = i][m = j]
  load_tile(ty, y_bh, ys.sn, min(TS, N), TS);          // [k = i][n = d]
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
    store_tile(Mt, tm);
    store_tile(Yt, ty);
    __syncthreads();
    if (c + 1 < nch) {
      const int rv = min(TS, N - (c + 1) * TS);
      load_tile(tm, m_bh + (int64_t)(c + 1) * TS * N, N, rv, cols_valid);
      load_tile(ty, y_bh + (int64_t)(c + 1) * TS * ys.sn, ys.sn, rv, TS);
    }
    if (active) mma64<true, true>(acc, Mt, Yt, wm, wn, lr, kh);
  }
  const float* x_bh = X + (int64_t)b * xs.sb + (int64_t)h * xs.sh;
  float* o_bh = out + (int64_t)b * os.sb + (int64_t)h * os.sh;
  const int d = wn * 32 + lr;
  if (cols_valid == TS) {            // full tile: the 16 X values are requested together
    float xv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) xv[e] = x_bh[(int64_t)(col0 + wm * 32 + crow(e, kh)) * xs.sn + d];
#pragma unroll
    for (int e = 0; e < 16; ++e) o_bh[(int64_t)(col0 + wm * 32 + crow(e, kh)) * os.sn + d] = (xv[e] * acc[e]) * scale;
    return;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int j = col0 + wm * 32 + crow(e, kh);
    if (j < N) o_bh[(int64_t)j * os.sn + d] = (x_bh[(int64_t)j * xs.sn + d] * acc[e]) * scale;
  }
}

// ------------------------------------------------------------------------------------------------
// QK rule, query-row side.  Rnn, Z, Sws contiguous [B*H,N,N]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void qk_row_kernel(
    const float* __restrict__ Rnn, const float* __restrict__ Z, const float* __restrict__ q, Strided qs,
    const float* __restrict__ k, Strided ks, float* __restrict__ cam_q, Strided cs, float* __restrict__ Sws, int H,
    int N, int BH, float scale) {
  __shared__ __attribute__((aligned(16))) float St[TS * TS];
  __shared__ __attribute__((aligned(16))) float Kt[TS * TS];
  const int bh = blockIdx.x % BH, rt = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, rows_valid = min(TS, N - row0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const bool row_active = (wm * 32) < rows_valid;
  const float* q_bh = q + (int64_t)b * qs.sb + (int64_t)h * qs.sh;
  const float* k_bh = k + (int64_t)b * ks.sb + (int64_t)h * ks.sh;
  const int64_t nn0 = ((int64_t)bh * N + row0) * N;     // first element of this stripe in the [N,N] tensors
  const int nch = (N + TS - 1) / TS;

  // S chunk (rows = this stripe, cols = keys of chunk c) = sd(Rnn, Z): computed in the load layout, kept in
  // registers until the LDS tile is free, and written to the workspace for the cam_k kernel
  auto s_chunk = [&](TileRegs& ts, int c) __attribute__((always_inline)) {
    const int kc = min(TS, N - c * TS);
    if (rows_valid == TS && kc == TS) {            // full tile: the eight loads go out before the first division
      f32x4 r[4], z[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        const int64_t off = nn0 + (int64_t)(idx >> 4) * N + c * TS + ((idx & 15) << 2);
        r[i] = *reinterpret_cast<const f32x4_u*>(Rnn + off);
        z[i] = *reinterpret_cast<const f32x4_u*>(Z + off);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * kThreads;
        const int64_t off = nn0 + (int64_t)(idx >> 4) * N + c * TS + ((idx & 15) << 2);
        f32x4 sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = te_sd(r[i][e], z[i][e]);
        *reinterpret_cast<f32x4_u*>(Sws + off) = sv;
        ts.v[i] = sv;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int row = idx >> 4, cc = (idx & 15) << 2;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      if (row < rows_valid) {
        const int64_t off = nn0 + (int64_t)row * N + c * TS;
        const f32x4 r = load4(Rnn + off, cc, kc), z = load4(Z + off, cc, kc);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = (cc + e < kc) ? te_sd(r[e], z[e]) : 0.0f;
        store4(Sws + off, cc, kc, s);
      }
      ts.v[i] = s;
    }
  };

  f32x16 accq;
  zero(accq);
  TileRegs tk, ts;
  load_tile(tk, k_bh, ks.sn, min(TS, N), TS);
  s_chunk(ts, 0);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
    store_tile(St, ts);              // rows = queries (m), K = keys of this chunk
    store_tile(Kt, tk);              // [k = key][n = d]
    __syncthreads();
    if (c + 1 < nch) {
      load_tile(tk, k_bh + (int64_t)(c + 1) * TS * ks.sn, ks.sn, min(TS, N - (c + 1) * TS), TS);
      s_chunk(ts, c + 1);
    }
    if (row_active) mma64<false, true>(accq, St, Kt, wm, wn, lr, kh);
  }
  float* o_bh = cam_q + (int64_t)b * cs.sb + (int64_t)h * cs.sh;
  const int d = wn * 32 + lr;
  if (rows_valid == TS) {            // full stripe: the 16 q values are requested together
    float qv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) qv[e] = q_bh[(int64_t)(row0 + wm * 32 + crow(e, kh)) * qs.sn + d];
#pragma unroll
    for (int e = 0; e < 16; ++e) o_bh[(int64_t)(row0 + wm * 32 + crow(e, kh)) * cs.sn + d] = (qv[e] * accq[e]) * scale;
    return;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int gi = row0 + wm * 32 + crow(e, kh);
    if (gi < N) o_bh[(int64_t)gi * cs.sn + d] = (q_bh[(int64_t)gi * qs.sn + d] * accq[e]) * scale;
  }
}

// ------------------------------------------------------------------------------------------------
// Z for callers that did not keep the forward products:  Zav [BH,N,64] = attn v ;  Zqk [BH,N,N] = q k^T
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void z_av_kernel(const float* __restrict__ attn, const float* __restrict__ v,
                                                        Strided vs, float* __restrict__ Zout, int H, int N, int BH) {
  __shared__ __attribute__((aligned(16))) float At[TS * TS];
  __shared__ __attribute__((aligned(16))) float Vt[TS * TS];
  const int bh = blockIdx.x % BH, rt = blockIdx.x / BH;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, rows_valid = min(TS, N - row0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const float* a_bh = attn + ((int64_t)bh * N + row0) * N;
  const float* v_bh = v + (int64_t)b * vs.sb + (int64_t)h * vs.sh;
  const int nch = (N + TS - 1) / TS;
  f32x16 acc;
  zero(acc);
  TileRegs ta, tv;
  load_tile(ta, a_bh, N, rows_valid, min(TS, N));
  load_tile(tv, v_bh, vs.sn, min(TS, N), TS);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
    store_tile(At, ta);              // rows = queries, K = keys of this chunk
    store_tile(Vt, tv);              // [k = key][n = d]
    __syncthreads();
    if (c + 1 < nch) {
      const int kv = min(TS, N - (c + 1) * TS);
      load_tile(ta, a_bh + (c + 1) * TS, N, rows_valid, kv);
      load_tile(tv, v_bh + (int64_t)(c + 1) * TS * vs.sn, vs.sn, kv, TS);
    }
    if ((wm * 32) < rows_valid) mma64<false, true>(acc, At, Vt, wm, wn, lr, kh);
  }
  const int d = wn * 32 + lr;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int gi = row0 + wm * 32 + crow(e, kh);
    if (gi < N) Zout[((int64_t)bh * N + gi) * TS + d] = acc[e];
  }
}

__global__ __launch_bounds__(kThreads) void z_qk_kernel(const float* __restrict__ q, Strided qs,
                                                        const float* __restrict__ k, Strided ks,
                                                        float* __restrict__ Zout, int H, int N, int BH, int nt) {
  __shared__ __attribute__((aligned(16))) float Qt[TS * TS];
  __shared__ __attribute__((aligned(16))) float Kt[TS * TS];
  const int bh = blockIdx.x % BH, t = blockIdx.x / BH;
  const int rt = t / nt, ct = t % nt;
  const int b = bh / H, h = bh % H;
  const int row0 = rt * TS, col0 = ct * TS;
  const int rows_valid = min(TS, N - row0), cols_valid = min(TS, N - col0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  TileRegs tq, tk;
  load_tile(tq, q + (int64_t)b * qs.sb + (int64_t)h * qs.sh + (int64_t)row0 * qs.sn, qs.sn, rows_valid, TS);
  load_tile(tk, k + (int64_t)b * ks.sb + (int64_t)h * ks.sh + (int64_t)col0 * ks.sn, ks.sn, cols_valid, TS);
  store_tile(Qt, tq);                // rows = queries (m), K = d
  store_tile(Kt, tk);                // rows = keys (n), K = d
  __syncthreads();
  if ((wm * 32) < rows_valid && (wn * 32) < cols_valid) {
    f32x16 acc;
    zero(acc);
    mma64<false, false>(acc, Qt, Kt, wm, wn, lr, kh);
    const int gj = col0 + wn * 32 + lr;
    if (gj < N) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int gi = row0 + wm * 32 + crow(e, kh);
        if (gi < N) Zout[((int64_t)bh * N + gi) * N + gj] = acc[e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Rollout chain step (te_rollout.hip): C[b] = A[b] Bm[b], all [N,N] row-major, on the same 64 x 64 MFMA tiles --
// A stripe K-contiguous, Bm chunk k-major; N need not be a multiple of anything (dword-aligned 16-B accesses).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void rollout_bmm_mfma_kernel(const float* __restrict__ A,
                                                                    const float* __restrict__ Bm,
                                                                    float* __restrict__ C, int N, int nt) {
  __shared__ __attribute__((aligned(16))) float At[TS * TS];
  __shared__ __attribute__((aligned(16))) float Bt[TS * TS];
  const int b = blockIdx.x / (nt * nt), t = blockIdx.x % (nt * nt);
  const int row0 = (t / nt) * TS, col0 = (t % nt) * TS;
  const int rows_valid = min(TS, N - row0), cols_valid = min(TS, N - col0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
  const float* a_b = A + ((int64_t)b * N + row0) * N;
  const float* b_b = Bm + (int64_t)b * N * N + col0;
  f32x16 acc;
  zero(acc);
  TileRegs ta, tb;
  load_tile(ta, a_b, N, rows_valid, min(TS, N));
  load_tile(tb, b_b, N, min(TS, N), cols_valid);
  for (int c = 0; c < nt; ++c) {
    __syncthreads();
    store_tile(At, ta);              // rows = output rows (m), K = k of this chunk
    store_tile(Bt, tb);              // [k][n]
    __syncthreads();
    if (c + 1 < nt) {
      const int kv = min(TS, N - (c + 1) * TS);
      load_tile(ta, a_b + (c + 1) * TS, N, rows_valid, kv);
      load_tile(tb, b_b + (int64_t)(c + 1) * TS * N, N, kv, cols_valid);
    }
    if ((wm * 32) < rows_valid && (wn * 32) < cols_valid) mma64<false, true>(acc, At, Bt, wm, wn, lr, kh);
  }
  const int gj = col0 + wn * 32 + lr;
  if (gj < N) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int gi = row0 + wm * 32 + crow(e, kh);
      if (gi < N) C[((int64_t)b * N + gi) * N + gj] = acc[e];
    }
  }
}

}  // namespace

int rollout_bmm_launch(const float* A, const float* Bm, float* C, int64_t B, int64_t N, hipStream_t stream) {
  const int64_t nt = (N + TS - 1) / TS;
  if (B * nt * nt > 0x7fffffff || N > (1 << 20)) return TE_ERR_UNSUPPORTED;
  rollout_bmm_mfma_kernel<<<dim3((unsigned)(B * nt * nt)), dim3(kThreads), 0, stream>>>(A, Bm, C, (int)N, (int)nt);
  return TE_OK;
}

bool av_supported(int64_t N, int64_t D) { return D == TS && N >= 1 && N <= (1 << 20); }
bool qk_supported(int64_t N, int64_t D) { return D == TS && N >= 1 && N <= (1 << 20); }

// workspace (floats): S [B*H,N,64] followed by Z [B*H,N,64] (used only when the caller passes Z == NULL)
int av_launch(const float* R, int64_t r_sb, int64_t r_sh, int64_t r_sn, const float* attn, const float* v,
              int64_t v_sb, int64_t v_sh, int64_t v_sn, const float* Z, int64_t z_sb, int64_t z_sh, int64_t z_sn,
              float* cam_attn, float* cam_v, int64_t cv_sb, int64_t cv_sh, int64_t cv_sn, int64_t B, int64_t H,
              int64_t N, int64_t D, float scale, float* ws, hipStream_t stream) {
  if (D != TS) return TE_ERR_UNSUPPORTED;
  const bool z_contig = !Z || (z_sn == TS && z_sh == N * TS && z_sb == H * N * TS);
  const int BH = (int)(B * H);
  const int nt = (int)((N + TS - 1) / TS);
  const Strided rs{r_sb, r_sh, r_sn}, vs{v_sb, v_sh, v_sn}, cs{cv_sb, cv_sh, cv_sn};
  const Strided ss{H * N * (int64_t)TS, N * (int64_t)TS, (int64_t)TS};   // workspace S [B,H,N,64]
  float* wsS = ws;
  const dim3 grid((unsigned)(BH * nt)), blk(kThreads);
  if (!Z) {
    float* wsZ = ws + (size_t)BH * N * TS;
    z_av_kernel<<<grid, blk, 0, stream>>>(attn, v, vs, wsZ, (int)H, (int)N, BH);
    Z = wsZ;
    z_sb = H * N * TS, z_sh = N * TS, z_sn = TS;
  }
  if (te_attn_rules::enabled() && te_attn_rules::supported(B, H, N, D))
    return te_attn_rules::av_launch(R, r_sb, r_sh, r_sn, attn, v, v_sb, v_sh, v_sn, Z, z_sb, z_sh, z_sn, cam_attn, cam_v,
                                    cv_sb, cv_sh, cv_sn, B, H, N, scale, stream);
  if (!z_contig) return TE_ERR_UNSUPPORTED;      // the 64 x 64-tile kernels read Z as contiguous [B*H,N,64]
  av_row_kernel<<<grid, blk, 0, stream>>>(R, rs, Z, attn, v, vs, cam_attn, wsS, (int)H, (int)N, BH, scale);
  col_kernel<<<grid, blk, 0, stream>>>(attn, wsS, ss, v, vs, cam_v, cs, (int)H, (int)N, BH, scale);
  return TE_OK;
}

// workspace (floats): S [B*H,N,N] followed by Z [B*H,N,N] (used only when the caller passes Z == NULL)
int qk_launch(const float* Rnn, const float* q, int64_t q_sb, int64_t q_sh, int64_t q_sn, const float* k,
              int64_t k_sb, int64_t k_sh, int64_t k_sn, const float* Z, float* cam_q, int64_t cq_sb, int64_t cq_sh,
              int64_t cq_sn, float* cam_k, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn, int64_t B, int64_t H, int64_t N,
              int64_t D, float scale, float* ws, const float* r_scale, int64_t r_scale_stride, hipStream_t stream) {
  if (D != TS) return TE_ERR_UNSUPPORTED;
  const int BH = (int)(B * H);
  const int nt = (int)((N + TS - 1) / TS);
  const Strided qs{q_sb, q_sh, q_sn}, ks{k_sb, k_sh, k_sn}, cqs{cq_sb, cq_sh, cq_sn}, cks{ck_sb, ck_sh, ck_sn};
  float* wsS = ws;
  const dim3 grid((unsigned)(BH * nt)), blk(kThreads);
  if (!Z) {
    float* wsZ = ws + (size_t)BH * N * N;
    z_qk_kernel<<<dim3((unsigned)(BH * nt * nt)), blk, 0, stream>>>(q, qs, k, ks, wsZ, (int)H, (int)N, BH, nt);
    Z = wsZ;
  }
  if (te_attn_rules::enabled() && te_attn_rules::supported(B, H, N, D))
    // (the per-group cam_q partials of N > 256 live in the S region of the workspace, which this path never writes:
    //  ngroups * 64 <= N whenever ngroups > 1)
    return te_attn_rules::qk_launch(Rnn, q, q_sb, q_sh, q_sn, k, k_sb, k_sh, k_sn, Z, cam_q, cq_sb, cq_sh, cq_sn, cam_k,
                                    ck_sb, ck_sh, ck_sn, B, H, N, scale, wsS, r_scale, r_scale_stride, stream);
  if (r_scale) return TE_ERR_UNSUPPORTED;        // only the one-pass kernel takes the deferred factor
  qk_row_kernel<<<grid, blk, 0, stream>>>(Rnn, Z, q, qs, k, ks, cam_q, cqs, wsS, (int)H, (int)N, BH, scale);
  col_kernel<<<grid, blk, 0, stream>>>(wsS, q, qs, k, ks, cam_k, cks, (int)H, (int)N, BH, scale);
  return TE_OK;
}

}  // namespace te_attn_mfma
