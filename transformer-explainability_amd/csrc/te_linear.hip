// te_linear.hip -- Linear.relprop (modules/layers_ours.py:207-230, modules/layers_lrp.py:188-211) as two
// fused fp32-MFMA GEMM kernels for gfx950.
//
//   K1 ("Z-pass",  NT GEMM, K = in_f):  Z[t,j] = sum_i X+[t,i] W+[j,i] + X-[t,i] W-[j,i]
//                                       S = safe_divide(R, Z)                       (epilogue)
//   K2 ("C-pass",  NN GEMM, K = out_f): P+[t,i] = sum_j S[t,j] W+[j,i] ; P-[t,i] = sum_j S[t,j] W-[j,i]
//                                       out = X+ . P+ + X- . P-                      (epilogue)
//
// The positive / negative parts are formed in registers right before the MFMA (v_max/v_min on the
// fragment that was just read from LDS), so X, W and S are each staged through LDS exactly once and no
// clamped copy of W or X ever exists in HBM.  v_mfma_f32_32x32x2_f32 is an exact-f32 k-ordered fma
// chain (MI355X guide), i.e. numerically a plain fp32 GEMM.
//
// Tiling: 256 threads = 4 waves as 2(M) x 2(N); block tile 128 x 128 x 32; each wave owns 64 x 64 =
// 2 x 2 MFMA 32x32 accumulators (K1: 64 acc VGPRs; K2: 128, P+ and P- share the A fragments).
// A-type tiles (K-contiguous rows: X, S, W in K1) sit in LDS as [128][36] floats -- the 4-float pad
// makes the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots -- and each lane pulls FOUR
// consecutive k of its row per ds_read_b128: lanes 0-31 supply k = kg*8+j, lanes 32-63 k = kg*8+4+j
// to the j-th of four consecutive MFMAs (any k pairing is legal as long as A and B agree).
// The K2 B tile (W rows = k, contiguous along n) is [32][128] and read with conflict-free ds_read_b32.
// LDS is double-buffered (one __syncthreads per K step, global loads for step k+1 in flight during
// the MFMAs of step k); 2 blocks/CU co-reside so one block's barrier hides under the other's MFMAs.
// blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous run of tiles (n fastest), so
// blocks that share an X/S row panel share an L2.
//
// Odd shapes (in_f or out_f not a multiple of 4, or forced by TE_IMPL_SIMPLE) use plain one-thread-
// per-output kernels; they double as the on-device cross-check for the tiled path.
#include <stdlib.h>

#include "te_common.h"

namespace {

constexpr int BM = 128, BK = 32;
constexpr int LDT = BK + 4;       // padded leading dim of K-contiguous tiles (floats)
constexpr int kThreads = 256;
// The block tile is BM x BN with BN = 128 (each wave 64 x 64 = 2 x 2 MFMA blocks) or BN = 64 (each wave 64 x 32 =
// 2 x 1).  fp32 MFMA is 16x slower than bf16 MFMA, so even the narrow tile leaves LDS / L2 bandwidth idle; what it
// buys is granularity: a 12,608 x 768 output is 594 wide tiles = 2.32 per CU (a third round that is 1/3 full),
// but 1188 narrow tiles = 4.64 per CU.  pick_bn() chooses per launch.

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // bijective remap: XCD x (= bid % 8 by dispatch order) gets a contiguous run of logical tiles
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Stage a [ROWS rows][32 k] tile of a K-contiguous matrix M (ld = K) into registers (ROWS/32 x float4/thread).
template <int ROWS>
__device__ __forceinline__ void load_rows_tile(const float* __restrict__ M, int64_t rows, int64_t K,
                                               int64_t row0, int64_t k0, f32x4 (&reg)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int idx = threadIdx.x + i * kThreads;  // 0..1023
    const int row = idx >> 3, c4 = idx & 7;
    const int64_t gr = row0 + row, gk = k0 + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gr < rows && gk < K) v = *reinterpret_cast<const f32x4*>(M + gr * K + gk);
    reg[i] = v;
  }
}
template <int ROWS>
__device__ __forceinline__ void store_rows_tile(float* __restrict__ lds, const f32x4 (&reg)[ROWS / 32]) {
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int row = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<f32x4*>(lds + row * LDT + c4 * 4) = reg[i];
  }
}
// Stage a [32 k][BN n] tile of a row-major K x N matrix (ld = Nn).
template <int BN>
__device__ __forceinline__ void load_kn_tile(const float* __restrict__ M, int64_t K, int64_t Nn,
                                             int64_t k0, int64_t n0, f32x4 (&reg)[BN / 32]) {
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int kk = idx / (BN / 4), c4 = idx % (BN / 4);
    const int64_t gk = k0 + kk, gn = n0 + c4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gk < K && gn < Nn) v = *reinterpret_cast<const f32x4*>(M + gk * Nn + gn);
    reg[i] = v;
  }
}
template <int BN>
__device__ __forceinline__ void store_kn_tile(float* __restrict__ lds, const f32x4 (&reg)[BN / 32]) {
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int idx = threadIdx.x + i * kThreads;
    const int kk = idx / (BN / 4), c4 = idx % (BN / 4);
    *reinterpret_cast<f32x4*>(lds + kk * BN + c4 * 4) = reg[i];
  }
}

#define TE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------------
// K1: S = sd(R, X+ W+^T + X- W-^T)        SWAP exchanges W+ / W- (inhibitor term, beta != 0)
//     LRP: S1 = sd(R, X+ W+^T), S2 = sd(R, X- W-^T) kept apart (layers_lrp.py:199-200)
// ------------------------------------------------------------------------------------------------
template <bool LRP, bool SWAP, int BN>
__global__ __launch_bounds__(kThreads, 2) void linear_k1_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ R,
    float* __restrict__ S1, float* __restrict__ S2, int64_t T, int64_t K, int64_t Nn, int nbn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NI = BN / 64;                 // 32-wide MFMA column blocks per wave
  constexpr int WN = BN / 2;                  // columns per wave
  constexpr int A_SZ = BM * LDT;
  constexpr int STAGE = (BM + BN) * LDT;      // floats per pipeline stage: [A tile | B tile]

  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)(tile / nbn) * BM;
  const int64_t col0 = (int64_t)(tile % nbn) * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, kh = lane >> 5;

  constexpr int NACC = LRP ? 2 : 1;
  f32x16 acc[NACC][2][NI];
#pragma unroll
  for (int s = 0; s < NACC; ++s)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][mi][ni][e] = 0.0f;

  f32x4 ra[BM / 32], rb[BN / 32];
  const int nk = (int)((K + BK - 1) / BK);
  load_rows_tile<BM>(X, T, K, row0, 0, ra);
  load_rows_tile<BN>(W, Nn, K, col0, 0, rb);
  store_rows_tile<BM>(smem, ra);
  store_rows_tile<BN>(smem + A_SZ, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_rows_tile<BM>(X, T, K, row0, (int64_t)(kt + 1) * BK, ra);
      load_rows_tile<BN>(W, Nn, K, col0, (int64_t)(kt + 1) * BK, rb);
    }
    const float* a_base = smem + cur * STAGE + (wm * 64 + lr) * LDT + kh * 4;
    const float* b_base = smem + cur * STAGE + A_SZ + (wn * WN + lr) * LDT + kh * 4;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 a[2], b[NI];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(a_base + mi * 32 * LDT + kg * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(b_base + ni * 32 * LDT + kg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float ap[2], an[2], bp[NI], bn[NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          ap[mi] = fmaxf(a[mi][j], 0.0f);
          an[mi] = fminf(a[mi][j], 0.0f);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float p = fmaxf(b[ni][j], 0.0f), n = fminf(b[ni][j], 0.0f);
          bp[ni] = SWAP ? n : p;   // partner of X+
          bn[ni] = SWAP ? p : n;   // partner of X-
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[0][mi][ni] = TE_MFMA(ap[mi], bp[ni], acc[0][mi][ni]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[NACC - 1][mi][ni] = TE_MFMA(an[mi], bn[ni], acc[NACC - 1][mi][ni]);
      }
    }
    if (kt + 1 < nk) {
      store_rows_tile<BM>(smem + (cur ^ 1) * STAGE, ra);
      store_rows_tile<BN>(smem + (cur ^ 1) * STAGE + A_SZ, rb);
    }
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32 MFMA -- col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t gc = col0 + wn * WN + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = row0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (gr < T && gc < Nn) {
          const float r = R[gr * Nn + gc];
          if constexpr (LRP) {
            S1[gr * Nn + gc] = te_sd(r, acc[0][mi][ni][e]);
            S2[gr * Nn + gc] = te_sd(r, acc[1][mi][ni][e]);
          } else {
            S1[gr * Nn + gc] = te_sd(r, acc[0][mi][ni][e]);
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// K2: out = X+ . (S W+) + X- . (S W-)                       MODE 0 (ours: both, shared S)
//     MODE 1: out  = scale * X+ . (S W(+))   (lrp first half, S = S1)
//     MODE 2: out += scale * X- . (S W(-))   (lrp second half, S = S2)
//     SWAP exchanges W+ / W-.  ACCUM: out = out - scale * (...)   (the beta * inhibitor term)
// ------------------------------------------------------------------------------------------------
template <int MODE, bool SWAP, bool ACCUM, int BN>
__global__ __launch_bounds__(kThreads, 2) void linear_k2_kernel(
    const float* __restrict__ S, const float* __restrict__ W, const float* __restrict__ X,
    float* __restrict__ out, int64_t T, int64_t K, int64_t Nn, int nbn, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NI = BN / 64, WN = BN / 2;
  constexpr int A_SZ = BM * LDT, B_SZ = BK * BN;
  constexpr int STAGE = A_SZ + B_SZ;   // floats per pipeline stage: [A tile | B tile]

  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)(tile / nbn) * BM;
  const int64_t col0 = (int64_t)(tile % nbn) * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, kh = lane >> 5;

  constexpr int NACC = (MODE == 0) ? 2 : 1;
  f32x16 acc[NACC][2][NI];
#pragma unroll
  for (int s = 0; s < NACC; ++s)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][mi][ni][e] = 0.0f;

  f32x4 ra[BM / 32], rb[BN / 32];
  const int nk = (int)((K + BK - 1) / BK);
  load_rows_tile<BM>(S, T, K, row0, 0, ra);
  load_kn_tile<BN>(W, K, Nn, 0, col0, rb);
  store_rows_tile<BM>(smem, ra);
  store_kn_tile<BN>(smem + A_SZ, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_rows_tile<BM>(S, T, K, row0, (int64_t)(kt + 1) * BK, ra);
      load_kn_tile<BN>(W, K, Nn, (int64_t)(kt + 1) * BK, col0, rb);
    }
    const float* a_base = smem + cur * STAGE + (wm * 64 + lr) * LDT + kh * 4;
    const float* b_base = smem + cur * STAGE + A_SZ + (kh * 4) * BN + wn * WN + lr;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      f32x4 a[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(a_base + mi * 32 * LDT + kg * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bp[NI], bn[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float w = b_base[(kg * 8 + j) * BN + ni * 32];
          const float p = fmaxf(w, 0.0f), n = fminf(w, 0.0f);
          bp[ni] = SWAP ? n : p;
          bn[ni] = SWAP ? p : n;
        }
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[0][mi][ni] = TE_MFMA(a[mi][j], bp[ni], acc[0][mi][ni]);
        }
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[NACC - 1][mi][ni] = TE_MFMA(a[mi][j], bn[ni], acc[NACC - 1][mi][ni]);
        }
      }
    }
    if (kt + 1 < nk) {
      store_rows_tile<BM>(smem + (cur ^ 1) * STAGE, ra);
      store_kn_tile<BN>(smem + (cur ^ 1) * STAGE + A_SZ, rb);
    }
    __syncthreads();
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t gc = col0 + wn * WN + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = row0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (gr < T && gc < Nn) {
          const float x = X[gr * Nn + gc];
          const float xp = fmaxf(x, 0.0f), xn = fminf(x, 0.0f);
          float v;
          if constexpr (MODE == 0) v = xp * acc[0][mi][ni][e] + xn * acc[1][mi][ni][e];
          else if constexpr (MODE == 1) v = xp * acc[0][mi][ni][e];
          else v = xn * acc[0][mi][ni][e];
          v = scale * v;
          if constexpr (ACCUM) v = out[gr * Nn + gc] - v;          // alpha*act - beta*inh
          else if constexpr (MODE == 2) v = out[gr * Nn + gc] + v;  // C1 + C2 of the lrp variant
          out[gr * Nn + gc] = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// simple kernels (any shape): one thread per output element, k-ordered fmaf chains
// ------------------------------------------------------------------------------------------------
template <bool LRP, bool SWAP>
__global__ __launch_bounds__(kThreads) void linear_k1_simple(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ R,
    float* __restrict__ S1, float* __restrict__ S2, int64_t T, int64_t K, int64_t Nn) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= T * Nn) return;
  const int64_t t = i / Nn, j = i % Nn;
  float z1 = 0.0f, z2 = 0.0f;
  for (int64_t k = 0; k < K; ++k) {
    const float x = X[t * K + k], w = W[j * K + k];
    const float wp = fmaxf(w, 0.0f), wn = fminf(w, 0.0f);
    z1 = fmaf(fmaxf(x, 0.0f), SWAP ? wn : wp, z1);
    z2 = fmaf(fminf(x, 0.0f), SWAP ? wp : wn, z2);
  }
  const float r = R[i];
  if constexpr (LRP) {
    S1[i] = te_sd(r, z1);
    S2[i] = te_sd(r, z2);
  } else {
    S1[i] = te_sd(r, z1 + z2);
  }
}

template <bool SWAP, bool ACCUM>
__global__ __launch_bounds__(kThreads) void linear_k2_simple(
    const float* __restrict__ S1, const float* __restrict__ S2, const float* __restrict__ W,
    const float* __restrict__ X, float* __restrict__ out, int64_t T, int64_t K, int64_t Nn, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= T * Nn) return;
  const int64_t t = idx / Nn, i = idx % Nn;
  float p1 = 0.0f, p2 = 0.0f;
  for (int64_t j = 0; j < K; ++j) {
    const float w = W[j * Nn + i];
    const float wp = fmaxf(w, 0.0f), wn = fminf(w, 0.0f);
    p1 = fmaf(S1[t * K + j], SWAP ? wn : wp, p1);
    p2 = fmaf(S2[t * K + j], SWAP ? wp : wn, p2);
  }
  const float x = X[idx];
  float v = fmaxf(x, 0.0f) * p1 + fminf(x, 0.0f) * p2;
  v = scale * v;
  if constexpr (ACCUM) v = out[idx] - v;
  out[idx] = v;
}

template <int BN>
constexpr size_t k1_lds() { return (size_t)2 * (BM + BN) * LDT * sizeof(float); }          // 73,728 / 55,296 B
template <int BN>
constexpr size_t k2_lds() { return (size_t)2 * (BM * LDT + BK * BN) * sizeof(float); }      // 69,632 / 53,248 B

template <typename Kern>
inline void allow_lds(Kern kern, size_t bytes) {
  // > 64 KiB of dynamic LDS must be opted into; cheap and idempotent, no device state besides the
  // function attribute.
  if (bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
}

// Tile-width choice per launch.  Work is issued in rounds of one tile per CU-slot; a launch whose tile count is
// just above a multiple of the CU count wastes most of its last round.  Estimate the chip-level efficiency of
// each width as (tiles / CUs) / ceil(tiles / CUs) -- times 0.96 for the narrow tile, whose B operand is re-staged
// twice as often -- and take the better one.  TE_LINEAR_BN = 64 | 128 in the environment pins the choice (tuning).
constexpr int kCUs = 256;
inline int pick_bn(int64_t T, int64_t n_out) {
  static const int pinned = [] {
    const char* e = getenv("TE_LINEAR_BN");
    return e ? atoi(e) : 0;
  }();
  if (pinned == 64 || pinned == 128) return pinned;
  const int64_t nbm = te_ceil_div(T, BM);
  auto eff = [&](int bn) {
    const double per_cu = (double)(nbm * te_ceil_div(n_out, bn)) / kCUs;
    return per_cu / (double)te_ceil_div(nbm * te_ceil_div(n_out, bn), kCUs);
  };
  return (0.96 * eff(64) > eff(128)) ? 64 : 128;
}

template <bool LRP, bool SWAP, int BN>
inline void launch_k1(const float* X, const float* W, const float* R, float* S1, float* S2, int64_t T, int64_t in_f,
                      int64_t out_f, hipStream_t stream) {
  const int nbm = (int)te_ceil_div(T, BM), nbn = (int)te_ceil_div(out_f, BN);
  allow_lds(linear_k1_kernel<LRP, SWAP, BN>, k1_lds<BN>());
  linear_k1_kernel<LRP, SWAP, BN><<<dim3((unsigned)(nbm * nbn)), dim3(kThreads), k1_lds<BN>(), stream>>>(
      X, W, R, S1, S2, T, in_f, out_f, nbn);
}
template <int MODE, bool SWAP, bool ACCUM, int BN>
inline void launch_k2(const float* S, const float* W, const float* X, float* out, int64_t T, int64_t in_f,
                      int64_t out_f, float scale, hipStream_t stream) {
  const int nbm = (int)te_ceil_div(T, BM), nbn = (int)te_ceil_div(in_f, BN);
  allow_lds(linear_k2_kernel<MODE, SWAP, ACCUM, BN>, k2_lds<BN>());
  linear_k2_kernel<MODE, SWAP, ACCUM, BN><<<dim3((unsigned)(nbm * nbn)), dim3(kThreads), k2_lds<BN>(), stream>>>(
      S, W, X, out, T, out_f, in_f, nbn, scale);
}

template <bool SWAP, bool ACCUM>
int run_half(const float* R, const float* X, const float* W, float* out, int64_t T, int64_t in_f,
             int64_t out_f, float scale, bool lrp, bool simple, float* S1, float* S2, hipStream_t stream) {
  if (simple) {
    const int64_t n1 = T * out_f, n2 = T * in_f;
    dim3 g1((unsigned)te_ceil_div(n1, kThreads)), g2((unsigned)te_ceil_div(n2, kThreads)), blk(kThreads);
    if (lrp) {
      linear_k1_simple<true, SWAP><<<g1, blk, 0, stream>>>(X, W, R, S1, S2, T, in_f, out_f);
      linear_k2_simple<SWAP, ACCUM><<<g2, blk, 0, stream>>>(S1, S2, W, X, out, T, out_f, in_f, scale);
    } else {
      linear_k1_simple<false, SWAP><<<g1, blk, 0, stream>>>(X, W, R, S1, S1, T, in_f, out_f);
      linear_k2_simple<SWAP, ACCUM><<<g2, blk, 0, stream>>>(S1, S1, W, X, out, T, out_f, in_f, scale);
    }
    return TE_OK;
  }
  const bool n1 = pick_bn(T, out_f) == 64, n2 = pick_bn(T, in_f) == 64;
  if (lrp) {
    // the lrp variant's K1 holds two accumulator sets (128 registers at BN = 128); out - beta*(C1 + C2) is formed
    // in two accumulate steps
    if (n1) launch_k1<true, SWAP, 64>(X, W, R, S1, S2, T, in_f, out_f, stream);
    else launch_k1<true, SWAP, 128>(X, W, R, S1, S2, T, in_f, out_f, stream);
    if (n2) {
      launch_k2<1, SWAP, ACCUM, 64>(S1, W, X, out, T, in_f, out_f, scale, stream);
      launch_k2<2, SWAP, ACCUM, 64>(S2, W, X, out, T, in_f, out_f, scale, stream);
    } else {
      launch_k2<1, SWAP, ACCUM, 128>(S1, W, X, out, T, in_f, out_f, scale, stream);
      launch_k2<2, SWAP, ACCUM, 128>(S2, W, X, out, T, in_f, out_f, scale, stream);
    }
  } else {
    if (n1) launch_k1<false, SWAP, 64>(X, W, R, S1, S1, T, in_f, out_f, stream);
    else launch_k1<false, SWAP, 128>(X, W, R, S1, S1, T, in_f, out_f, stream);
    if (n2) launch_k2<0, SWAP, ACCUM, 64>(S1, W, X, out, T, in_f, out_f, scale, stream);
    else launch_k2<0, SWAP, ACCUM, 128>(S1, W, X, out, T, in_f, out_f, scale, stream);
  }
  return TE_OK;
}

}  // namespace

// ---- single-pass entry points (variant "ours", alpha = 1): the two kernels of te_linear_relprop_f32
// individually, so that a caller can bracket ONE kernel launch with events (bench.py roofline) or
// interleave other work between the passes.  S is the [T,out_f] scratch of the composed call.
extern "C" int te_linear_zpass_f32(const float* R, const float* X, const float* W, float* S, int64_t T,
                                   int64_t in_f, int64_t out_f, te_stream_t stream_) {
  if (!R || !X || !W || !S || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(R) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(S))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  if (pick_bn(T, out_f) == 64) launch_k1<false, false, 64>(X, W, R, S, S, T, in_f, out_f, stream);
  else launch_k1<false, false, 128>(X, W, R, S, S, T, in_f, out_f, stream);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_linear_cpass_f32(const float* S, const float* X, const float* W, float* out, int64_t T,
                                   int64_t in_f, int64_t out_f, te_stream_t stream_) {
  if (!S || !X || !W || !out || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(S) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(out))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  if (pick_bn(T, in_f) == 64) launch_k2<0, false, false, 64>(S, W, X, out, T, in_f, out_f, 1.0f, stream);
  else launch_k2<0, false, false, 128>(S, W, X, out, T, in_f, out_f, 1.0f, stream);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" size_t te_linear_relprop_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f, int variant) {
  if (T <= 0 || in_f <= 0 || out_f <= 0) return 0;
  const size_t one = te_align_up((size_t)T * (size_t)out_f * sizeof(float), 256);
  return ((variant & 0xff) == TE_VARIANT_LRP) ? 2 * one : one;
}

extern "C" int te_linear_relprop_f32(const float* R, const float* X, const float* W, float* out,
                                     int64_t T, int64_t in_f, int64_t out_f, float alpha, int variant,
                                     void* ws, size_t ws_bytes, te_stream_t stream_) {
  if (!R || !X || !W || !out || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  const int var = variant & 0xff;
  if (var != TE_VARIANT_OURS && var != TE_VARIANT_LRP) return TE_ERR_INVALID_ARG;
  if (!ws || ws_bytes < te_linear_relprop_workspace_bytes(T, in_f, out_f, variant)) return TE_ERR_WORKSPACE;
  if (!te_aligned16(ws)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const bool lrp = (var == TE_VARIANT_LRP);
  const bool tiled_ok = (in_f % 4 == 0) && (out_f % 4 == 0) && te_aligned16(R) && te_aligned16(X) &&
                        te_aligned16(W) && te_aligned16(out);
  const bool simple = (variant & TE_IMPL_SIMPLE) || !tiled_ok;
  float* S1 = (float*)ws;
  float* S2 = lrp ? (float*)((char*)ws + te_align_up((size_t)T * (size_t)out_f * sizeof(float), 256)) : S1;
  const float beta = alpha - 1.0f;
  // out = alpha * act                                   (layers_ours.py:225,228)
  int rc = run_half<false, false>(R, X, W, out, T, in_f, out_f, alpha, lrp, simple, S1, S2, stream);
  if (rc != TE_OK) return rc;
  // out -= beta * inh, inh = f(nw, pw, px, nx)           (layers_ours.py:226,228) -- dead at alpha == 1
  if (beta != 0.0f) {
    rc = run_half<true, true>(R, X, W, out, T, in_f, out_f, beta, lrp, simple, S1, S2, stream);
    if (rc != TE_OK) return rc;
  }
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
