// te_linear_x6.hip -- Linear.relprop (modules/layers_ours.py:207-230, variant "ours", alpha = 1, Z from the cached
// forward output) with its three GEMM-shaped products on bf16 MFMAs at fp32 accuracy.  OPT-IN this round
// (ops.USE_LINEAR_X6 / TE_LINEAR_X6=1): the default path is the fp32-MFMA kernels of te_linear.hip.
//
// An fp32 number is exactly the sum of three bf16 numbers, a = a0 + a1 + a2 (8 + 8 + 8 significand bits), and a product
// of two bf16 values is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Of the nine partial products of
// a b the six above 2^-24 |a||b| are kept:
//        a b  ~=  a1 b1 + a0 b2 + a2 b0 + a0 b1 + a1 b0 + a0 b0          ("x6": what is dropped is below fp32 rounding)
// Measured: error against fp64 BELOW the fp32 GEMM's own (the products are exact, only the
// accumulation rounds), the ViT-B map moves less than under a K-permutation of the fp32 rule; six bf16 MFMAs sustain
// 2.15x the rate of one fp32 MFMA on this chip.
//
//   split kernels   fp32 [R,K] -> bf16 planes, per row and per 32-k block [3][32] (the 192-B tile row of one K-step is
//                   contiguous in memory and in LDS); op = |x| (Z-pass operands), or max(w,0) / min(w,0) of W^T (the
//                   C-pass's K = out_f operands)
//   zpass_x6        A = |X| |W|^T ; Z = ((Y - b) + A) / 2 (cancellation guard as te_linear.hip) ; S = sd(R f, Z), written
//                   directly as bf16 planes -- the C-pass's A operand never exists in fp32
//   cpass_x6        P+ = S W+, P- = S W- (one pass over S, two accumulator sets) ; out = X+ . P+ + X- . P-
//
// Tiles 128 x 128 (Z-pass) / 128 x 64 (C-pass, two products), 256 threads as 2 x 2 waves, K-step 32 = two K16 slices;
// global -> registers -> LDS staging with the next K-step's loads in flight during the MFMAs; LDS rows padded to 208 B
// (16-B fragments of a lane group land on distinct bank groups).  Row t of every output depends on row t of the inputs
// only and its MFMA chain is k-ordered: a batch equals its samples run one by one, bit for bit.
// Shapes: in_f and out_f multiples of 128 (every Linear of ViT-B/L and BERT-base except the classifier head).
#include "te_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int ROWB = 192;            // bytes of one row of one K-step in memory: 3 planes x 32 bf16
constexpr int LROW = 208;            // ... in LDS: 52 dwords per row -> the 16-B fragments of 16 rows cover all 64 banks
constexpr float kCancelTol = 0.0078125f;      // as te_linear.hip

#define TE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned short bf16_rn(float x) {      // round to nearest even, finite input
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// x = p[0] + p[1] + p[2] exactly (the residual of a round-to-nearest bf16 is representable in fp32)
__device__ __forceinline__ void split3(float x, unsigned short (&p)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    p[q] = bf16_rn(x);
    x = x - bf16_f32(p[q]);
  }
}

enum { OP_ABS = 0, OP_POS = 1, OP_NEG = 2 };
template <int OP>
__device__ __forceinline__ float apply_op(float x) {
  if constexpr (OP == OP_ABS) return __int_as_float(__float_as_int(x) & 0x7fffffff);
  const int b = __float_as_int(x);
  if constexpr (OP == OP_POS) return __int_as_float(b > 0 ? b : 0);
  return __int_as_float(b < 0 ? b : 0);
}

// dst row layout: [K / 32 blocks][3 planes][32 k].  One thread = 8 consecutive k of one row.
// TRANSPOSE: the source is [K][R] (row r of the result is column r of the source): the weight operands of the C-pass.
template <int OP, bool TRANSPOSE>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                    int64_t R, int64_t K) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_row = K >> 3;
  if (idx >= R * per_row) return;
  int64_t row;
  int c8;
  float v[8];
  if constexpr (!TRANSPOSE) {
    row = idx / per_row;
    c8 = (int)(idx - row * per_row);
    const float* s = src + row * K + (int64_t)c8 * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v0[e];
      v[4 + e] = v1[e];
    }
  } else {
    // consecutive threads take consecutive rows r (coalesced reads along a source row)
    c8 = (int)(idx / R);
    row = idx - (int64_t)c8 * R;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[((int64_t)c8 * 8 + e) * R + row];
  }
  unsigned short p[8][3];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3(apply_op<OP>(v[e]), p[e]);
  unsigned short* d = dst + row * (K / 32) * 96 + (int64_t)(c8 >> 2) * 96 + (c8 & 3) * 8;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (unsigned)p[2 * e][q] | ((unsigned)p[2 * e + 1][q] << 16);
    *reinterpret_cast<u32x4*>(d + q * 32) = w;
  }
}

// plain positive-part sum of one output element (the reference's Z), k-ordered: the cancellation fallback
__device__ __noinline__ float exact_z(const float* __restrict__ x, const float* __restrict__ w, int64_t K) {
  float z1 = 0.0f, z2 = 0.0f;
  for (int64_t k = 0; k < K; ++k) {
    const float xv = x[k], wv = w[k];
    z1 = fmaf(fmaxf(xv, 0.0f), fmaxf(wv, 0.0f), z1);
    z2 = fmaf(fminf(xv, 0.0f), fminf(wv, 0.0f), z2);
  }
  return z1 + z2;
}

// six partial products of one K16 slice, smallest first
__device__ __forceinline__ f32x16 mma_x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
  c = TE_MFMA_BF16(a[1], b[1], c);
  c = TE_MFMA_BF16(a[0], b[2], c);
  c = TE_MFMA_BF16(a[2], b[0], c);
  c = TE_MFMA_BF16(a[0], b[1], c);
  c = TE_MFMA_BF16(a[1], b[0], c);
  c = TE_MFMA_BF16(a[0], b[0], c);
  return c;
}

// rows [row0, row0 + ROWS) of a split operand, K-step kt -> registers (rows past `rows` re-read the last row: their
// products are never stored) ; registers -> LDS [ROWS][LROW]
template <int ROWS>
__device__ __forceinline__ void load_planes(u32x4 (&reg)[ROWS * 12 / kThreads], const unsigned short* __restrict__ Ps,
                                            int64_t rows, int64_t rowbytes, int64_t row0, int kt) {
#pragma unroll
  for (int i = 0; i < ROWS * 12 / kThreads; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int row = idx / 12, c = idx - row * 12;
    const int64_t gr = min(row0 + row, rows - 1);
    reg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(Ps) + gr * rowbytes + (int64_t)kt * ROWB + c * 16);
  }
}
template <int ROWS>
__device__ __forceinline__ void store_planes(unsigned char* __restrict__ lds, const u32x4 (&reg)[ROWS * 12 / kThreads]) {
#pragma unroll
  for (int i = 0; i < ROWS * 12 / kThreads; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int row = idx / 12, c = idx - row * 12;
    *reinterpret_cast<u32x4*>(lds + row * LROW + c * 16) = reg[i];
  }
}

struct RowScale {
  const float* s;
  int64_t stride;
  int rps;
};

// ------------------------------------------------------------------------------------------------
// Z-pass: S = sd(R f, ((Y - b) + |X||W|^T) / 2), written as bf16 planes [T][Nn/32][3][32]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void zpass_x6_kernel(
    const unsigned short* __restrict__ Xs, const unsigned short* __restrict__ Ws, const float* __restrict__ X,
    const float* __restrict__ W, const float* __restrict__ R, const float* __restrict__ Y, const float* __restrict__ bias,
    unsigned short* __restrict__ Ss, int64_t T, int K, int Nn, int nbn, RowScale rs) {
  constexpr int BN = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* At = smem;
  unsigned char* Bt = smem + BM * LROW;
  const int tile = blockIdx.x;
  const int64_t row0 = (int64_t)(tile / nbn) * BM;
  const int col0 = (tile % nbn) * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = K / BK;
  const int64_t rowbytes = (int64_t)(K / 32) * ROWB;

  u32x4 ra[6], rb[6];
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
  const unsigned char* ap = At + (wm * 64 + lr) * LROW + kh * 16;
  const unsigned char* bp = Bt + (wn * 64 + lr) * LROW + kh * 16;

  load_planes<BM>(ra, Xs, T, rowbytes, row0, 0);
  load_planes<BN>(rb, Ws, Nn, rowbytes, col0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_planes<BM>(At, ra);
    store_planes<BN>(Bt, rb);
    __syncthreads();
    if (kt + 1 < nk) {
      load_planes<BM>(ra, Xs, T, rowbytes, row0, kt + 1);
      load_planes<BN>(rb, Ws, Nn, rowbytes, col0, kt + 1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[i][q] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LROW + q * 64 + s * 32);
          b[i][q] = *reinterpret_cast<const bf16x8*>(bp + i * 32 * LROW + q * 64 + s * 32);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mma_x6(a[mi], b[ni], acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: 32x32 block layout col = lr, row = (e & 3) + 8 (e >> 2) + 4 kh
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int gc = col0 + wn * 64 + ni * 32 + lr;
      const int64_t gr0 = row0 + wm * 64 + mi * 32 + 4 * kh;
      float rr[16], yy[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = min(gr0 + (e & 3) + 8 * (e >> 2), T - 1);
        rr[e] = R[gr * Nn + gc];
        yy[e] = Y[gr * Nn + gc];
        if (rs.s) rr[e] = rr[e] * rs.s[(gr / rs.rps) * rs.stride];
      }
      const float bb = bias ? bias[gc] : 0.0f;
      unsigned short* sp = Ss + (int64_t)(gc >> 5) * 96 + (gc & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = gr0 + (e & 3) + 8 * (e >> 2);
        if (gr < T) {
          const float a_abs = acc[mi][ni][e];
          float z = 0.5f * ((yy[e] - bb) + a_abs);
          if (!(z > kCancelTol * a_abs)) z = exact_z(X + gr * K, W + (int64_t)gc * K, K);
          unsigned short p[3];
          split3(te_sd(rr[e], z), p);
          unsigned short* d = sp + gr * (int64_t)(Nn / 32) * 96;
          d[0] = p[0];
          d[32] = p[1];
          d[64] = p[2];
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// C-pass: out = X+ . (S W+) + X- . (S W-) ; S planes [T][K/32][3][32] (K = out_f), W+^T / W-^T planes [Nn][K/32][3][32]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void cpass_x6_kernel(
    const unsigned short* __restrict__ Ss, const unsigned short* __restrict__ Wp, const unsigned short* __restrict__ Wn,
    const float* __restrict__ X, float* __restrict__ out, int64_t T, int K, int Nn, int nbn) {
  constexpr int BN = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* At = smem;
  unsigned char* Pt = smem + BM * LROW;
  unsigned char* Nt = Pt + BN * LROW;
  const int tile = blockIdx.x;
  const int64_t row0 = (int64_t)(tile / nbn) * BM;
  const int col0 = (tile % nbn) * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = K / BK;
  const int64_t rowbytes = (int64_t)(K / 32) * ROWB;

  u32x4 ra[6], rp[3], rn[3];
  f32x16 accp[2], accn[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      accp[mi][e] = 0.0f;
      accn[mi][e] = 0.0f;
    }
  const unsigned char* ap = At + (wm * 64 + lr) * LROW + kh * 16;
  const unsigned char* pp = Pt + (wn * 32 + lr) * LROW + kh * 16;
  const unsigned char* np = Nt + (wn * 32 + lr) * LROW + kh * 16;

  load_planes<BM>(ra, Ss, T, rowbytes, row0, 0);
  load_planes<BN>(rp, Wp, Nn, rowbytes, col0, 0);
  load_planes<BN>(rn, Wn, Nn, rowbytes, col0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_planes<BM>(At, ra);
    store_planes<BN>(Pt, rp);
    store_planes<BN>(Nt, rn);
    __syncthreads();
    if (kt + 1 < nk) {
      load_planes<BM>(ra, Ss, T, rowbytes, row0, kt + 1);
      load_planes<BN>(rp, Wp, Nn, rowbytes, col0, kt + 1);
      load_planes<BN>(rn, Wn, Nn, rowbytes, col0, kt + 1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2][3], bpos[3], bneg[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[0][q] = *reinterpret_cast<const bf16x8*>(ap + q * 64 + s * 32);
        a[1][q] = *reinterpret_cast<const bf16x8*>(ap + 32 * LROW + q * 64 + s * 32);
        bpos[q] = *reinterpret_cast<const bf16x8*>(pp + q * 64 + s * 32);
        bneg[q] = *reinterpret_cast<const bf16x8*>(np + q * 64 + s * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        accp[mi] = mma_x6(a[mi], bpos, accp[mi]);
        accn[mi] = mma_x6(a[mi], bneg, accn[mi]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int gc = col0 + wn * 32 + lr;
    const int64_t gr0 = row0 + wm * 64 + mi * 32 + 4 * kh;
    float xv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) xv[e] = X[min(gr0 + (e & 3) + 8 * (e >> 2), T - 1) * Nn + gc];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t gr = gr0 + (e & 3) + 8 * (e >> 2);
      const float xp = fmaxf(xv[e], 0.0f), xn = fminf(xv[e], 0.0f);
      if (gr < T) out[gr * Nn + gc] = 1.0f * (xp * accp[mi][e] + xn * accn[mi][e]);
    }
  }
}

inline size_t planes_bytes(int64_t rows, int64_t K) { return te_align_up((size_t)rows * (size_t)K * 6, 256); }

}  // namespace

extern "C" int te_linear_relprop_x6_supported(int64_t T, int64_t in_f, int64_t out_f) {
  return (T >= 1 && in_f >= 128 && out_f >= 128 && in_f % 128 == 0 && out_f % 128 == 0 && in_f <= (1 << 20) &&
          out_f <= (1 << 20) && te_ceil_div(T, BM) * (out_f / 64 + in_f / 64) < 0x7fffffff) ? 1 : 0;
}

// workspace: |X| planes, |W| planes, W+^T and W-^T planes, S planes
extern "C" size_t te_linear_relprop_x6_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f) {
  if (!te_linear_relprop_x6_supported(T, in_f, out_f)) return 0;
  return planes_bytes(T, in_f) + 3 * planes_bytes(out_f, in_f) + planes_bytes(T, out_f);
}

extern "C" int te_linear_relprop_x6_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                        int64_t rows_per_sample, const float* X, const float* W, const float* Y,
                                        const float* bias, float* out, int64_t T, int64_t in_f, int64_t out_f, void* ws,
                                        size_t ws_bytes, te_stream_t stream_) {
  if (!R || !X || !W || !Y || !out || T <= 0) return TE_ERR_INVALID_ARG;
  if (!te_linear_relprop_x6_supported(T, in_f, out_f)) return TE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < te_linear_relprop_x6_workspace_bytes(T, in_f, out_f) || !te_aligned16(ws)) return TE_ERR_WORKSPACE;
  if (!te_aligned16(X) || !te_aligned16(W)) return TE_ERR_UNSUPPORTED;
  if (r_scale && (rows_per_sample <= 0 || rows_per_sample > 0x7fffffff || T % rows_per_sample)) return TE_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* p = (unsigned char*)ws;
  unsigned short* Xs = (unsigned short*)p;
  p += planes_bytes(T, in_f);
  unsigned short* Was = (unsigned short*)p;
  p += planes_bytes(out_f, in_f);
  unsigned short* Wps = (unsigned short*)p;      // rows = in_f, K = out_f
  p += planes_bytes(out_f, in_f);
  unsigned short* Wns = (unsigned short*)p;
  p += planes_bytes(out_f, in_f);
  unsigned short* Ss = (unsigned short*)p;
  auto blocks = [](int64_t rows, int64_t K) { return dim3((unsigned)te_ceil_div(rows * (K / 8), 256)); };
  split_kernel<OP_ABS, false><<<blocks(T, in_f), dim3(256), 0, stream>>>(X, Xs, T, in_f);
  split_kernel<OP_ABS, false><<<blocks(out_f, in_f), dim3(256), 0, stream>>>(W, Was, out_f, in_f);
  split_kernel<OP_POS, true><<<blocks(in_f, out_f), dim3(256), 0, stream>>>(W, Wps, in_f, out_f);
  split_kernel<OP_NEG, true><<<blocks(in_f, out_f), dim3(256), 0, stream>>>(W, Wns, in_f, out_f);
  const int nbm = (int)te_ceil_div(T, BM);
  {
    const int nbn = (int)(out_f / 128);
    zpass_x6_kernel<<<dim3((unsigned)(nbm * nbn)), dim3(kThreads), (size_t)(BM + 128) * LROW, stream>>>(
        Xs, Was, X, W, R, Y, bias, Ss, T, (int)in_f, (int)out_f, nbn, RowScale{r_scale, r_scale_stride, (int)rows_per_sample});
  }
  {
    const int nbn = (int)(in_f / 64);
    cpass_x6_kernel<<<dim3((unsigned)(nbm * nbn)), dim3(kThreads), (size_t)(BM + 128) * LROW, stream>>>(
        Ss, Wps, Wns, X, out, T, (int)out_f, (int)in_f, nbn);
  }
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
