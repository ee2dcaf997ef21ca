// te_linear.hip -- Linear.relprop (modules/layers_ours.py:207-230, modules/layers_lrp.py:188-211) as two
// fused fp32-MFMA GEMM kernels for gfx950.
//
//   K1 ("Z-pass",  NT GEMM, K = in_f):  Z[t,j] = sum_i X+[t,i] W+[j,i] + X-[t,i] W-[j,i]
//                                       S = safe_divide(R, Z)                       (epilogue)
//   K2 ("C-pass",  NN GEMM, K = out_f): P+[t,i] = sum_j S[t,j] W+[j,i] ; P-[t,i] = sum_j S[t,j] W-[j,i]
//                                       out = X+ . P+ + X- . P-                      (epilogue)
//
//   K1f ("Z-pass from the forward output", variant ours, alpha = 1): with Y = X W^T + b cached by the forward pass
//        (forward_hook stores self.Y, layers_ours.py:16-27)   X+W+^T + X-W-^T = ( (Y - b) + |X||W|^T ) / 2,
//        because the same-sign products sum to Z and the opposite-sign ones to XW^T - Z, while |X||W|^T is their
//        difference.  ONE GEMM (|X||W|^T, K = in_f) instead of two; where the two halves cancel (Z < 2^-7 |X||W|^T:
//        nearly every product negative) the element is recomputed as the plain positive-part sum, so exact zeros and
//        tiny Z behave like the reference.  Measured on ViT-B maps: moves the result by 3-8e-7 relative, the same as
//        permuting the summation order of the two-GEMM form (profiles/r01_zpass_from_forward_probe.log).
//
// The positive / negative parts are formed in registers right before the MFMA (one VALU op on the fragment that
// was just read from LDS), so X, W and S are each staged through LDS exactly once and no clamped copy of W or X
// ever exists in HBM.  v_mfma_f32_32x32x2_f32 is an exact-f32 k-ordered fma chain (MI355X guide), i.e.
// numerically a plain fp32 GEMM.
//
// Tiling: 256 threads = 4 waves as 2(M) x 2(N); block tile BM x BN x 32 with (BM, BN) one of 128x128, 128x64,
// 64x64, chosen per launch (pick_tile); each wave owns (BM/2) x (BN/2) = MI x NI accumulators of 32x32.
// K-contiguous tiles (X, S, and W in K1) sit in LDS as [rows][32] floats with an XOR swizzle of the 16-B chunks
// (swz()): conflict-free ds_read_b128 / ds_write_b128 without padding (measured: SQ_LDS_BANK_CONFLICT = 0).  Each
// lane pulls FOUR consecutive k of its row per ds_read_b128: lanes 0-31 supply k = kg*8+j, lanes 32-63 k = kg*8+4+j
// to the j-th of four consecutive MFMAs (any k pairing is legal as long as A and B agree).  The K2 B tile (W rows =
// k, contiguous along n) is [32][BN], read with conflict-free ds_read_b32.  LDS is double-buffered (one
// __syncthreads per K step, global loads for step k+1 in flight during the MFMAs of step k); 2 (64 KB), 3 (48 KB)
// or 5 (32 KB) blocks co-reside per CU so one block's barrier and epilogue hide under the others' MFMAs.
// blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous run of tiles (n fastest), so blocks that
// share an X/S row panel share an L2.
//
// The one-product Z-pass (K1f) has half the MFMAs per K-step and is bound by the latency of its global loads: it runs
// a two-stage REGISTER pipeline (the loads of K-step kt+3 go out right after the barrier that published kt+1) written
// as four straight-line K-steps per loop trip -- hipcc's waitcnt insertion drains every in-flight load at a loop
// back-edge, straight-line code lets it keep the newer stage in flight -- prefers 64x64 tiles (5 blocks per CU), and
// requests all R / Y values of a 32x32 block before the first use in its epilogue.
//
// fp32 MFMA runs at 1/16 of the bf16 rate, so LDS and L2 bandwidth are idle at every tile size (8 B/clk/CU of LDS
// reads against 256); what the tile choice trades is granularity (tiles per CU in the last round) against the
// number of co-resident waves that cover barriers and epilogues.  DESIGN.md section 3 has the ablation study.
//
// Odd shapes (in_f or out_f not a multiple of 4, or forced by TE_IMPL_SIMPLE) use plain one-thread-per-output
// kernels; they double as the on-device cross-check for the tiled path.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "te_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDT = BK;           // K-contiguous tiles are [rows][32] floats = 128-B rows, XOR-swizzled (swz())
constexpr int kThreads = 256;
constexpr int kCUs = 256;

// 16-B chunk c (0..7) of row r lives at chunk c ^ ((r >> 1) & 7).  With 128-B rows two consecutive rows span
// the 256-B bank row, so a ds_read_b128 lane group (16 lanes = 16 different rows, 8 even + 8 odd, same logical
// chunk) lands on 16 distinct 16-B slots iff the 8 rows of one parity get 8 distinct XOR masks -- (r >> 1) & 7
// does that for every lane group of the instruction ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32).
__device__ __forceinline__ int swz(int row, int chunk) { return row * LDT + ((chunk ^ ((row >> 1) & 7)) << 2); }

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
    const int idx = threadIdx.x + i * kThreads;
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
    *reinterpret_cast<f32x4*>(lds + swz(row, c4)) = reg[i];
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

// x+ = max(x, 0), x- = min(x, 0) on the BIT PATTERN: as two's-complement integers, non-negative floats are >= 0 and
// negative floats (sign bit set) are < 0, so v_max_i32(bits, 0) / v_min_i32(bits, 0) are the two clamps -- one VALU op
// each, exact for every finite input (and -0.0 -> x+ = +0, x- = -0).  The float forms (fmaxf / v_med3_f32) cost an
// extra canonicalising v_max_f32 x, x, x per operand under IEEE mode, i.e. 9 instead of 6 VALU ops per 4 MFMAs in
// the Z-pass -- measured +3.4 % on the whole kernel pair: every VALU op in the MFMA stream costs ~3 MFMA-pipe cycles.
// Plain C on purpose: hipcc pads the VALU-write -> MFMA-operand hazard for instructions it emits itself, not for
// inline asm (an asm v_max_f32 here fed stale operands to the MFMAs).
__device__ __forceinline__ float te_pos(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ float te_neg(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b < 0 ? b : 0);
}

__device__ __forceinline__ float te_abs(float x) { return __int_as_float(__float_as_int(x) & 0x7fffffff); }

#define TE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct TileCoord {
  int64_t row0, col0;
};
template <int BM, int BN>
__device__ __forceinline__ TileCoord tile_coord(int v, int ntiles, int nbn) {
  const int tile = xcd_swizzle(v, ntiles);
  return {(int64_t)(tile / nbn) * BM, (int64_t)(tile % nbn) * BN};
}

// Unguarded staging loads for interior tiles (every row / column / k of the tile exists): no predication, no
// branches -- the guarded forms above cost ~70 instructions and 6 taken-or-not branches per K-step.
template <int ROWS>
__device__ __forceinline__ void load_rows_tile_fast(const float* __restrict__ M, int64_t K, int64_t row0, int64_t k0,
                                                    f32x4 (&reg)[ROWS / 32]) {
  const float* p = M + (row0 + (threadIdx.x >> 3)) * K + k0 + ((threadIdx.x & 7) << 2);
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) reg[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)i * 32 * K);
}
template <int BN>
__device__ __forceinline__ void load_kn_tile_fast(const float* __restrict__ M, int64_t Nn, int64_t k0, int64_t n0,
                                                  f32x4 (&reg)[BN / 32]) {
  constexpr int RPI = kThreads / (BN / 4);   // k rows covered per iteration
  const float* p = M + (k0 + threadIdx.x / (BN / 4)) * Nn + n0 + ((threadIdx.x % (BN / 4)) << 2);
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) reg[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)i * RPI * Nn);
}

// Direct-to-LDS staging (global_load_lds_dwordx4: 64 lanes x 16 B land at a wave-uniform LDS base + lane * 16, no VGPR
// round trip, no ds_write): the LDS images above are lane-linear in the thread index (float4 slot idx = t + 256 i sits
// at float offset 4 idx), so the XOR swizzle of the K-contiguous tile moves to the SOURCE address -- lane (row, physical
// chunk pc) fetches logical chunk pc ^ ((row >> 1) & 7) of its row; the eight lanes of a row still read one 128-B line.
__device__ __forceinline__ void glds16(const float* __restrict__ src, float* __restrict__ lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int ROWS>
__device__ __forceinline__ void glds_rows_tile(float* __restrict__ lds, const float* __restrict__ M, int64_t K,
                                               int64_t row0, int64_t k0, int wave_base) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < ROWS / 32; ++i) {
    const int idx = wave_base + lane + i * kThreads;
    const int row = idx >> 3, lc = (idx & 7) ^ ((row >> 1) & 7);
    glds16(M + (row0 + row) * K + k0 + (lc << 2), lds + (wave_base + i * kThreads) * 4);
  }
}
template <int BN>
__device__ __forceinline__ void glds_kn_tile(float* __restrict__ lds, const float* __restrict__ M, int64_t Nn, int64_t k0,
                                             int64_t n0, int wave_base) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int idx = wave_base + lane + i * kThreads;
    const int kk = idx / (BN / 4), c4 = idx % (BN / 4);
    glds16(M + (k0 + kk) * Nn + n0 + (c4 << 2), lds + (wave_base + i * kThreads) * 4);
  }
}

// K-loop schedule shared by both kernels (one K-step = 4 k-groups of 8):
//
//     k-group 0 : MFMAs on fragments read during the PREVIOUS step's last group   | reads of group 1
//                 global loads of the next K-step's tiles are issued
//     k-group 1 : MFMAs                                                            | reads of group 2
//     k-group 2 : MFMAs                                                            | reads of group 3
//                 next tiles: registers -> the other LDS stage ; __syncthreads()
//     k-group 3 : MFMAs                                                            | reads of group 0 of the NEXT stage
//
// so the one barrier of a K-step and the LDS latency of the first fragments sit under a k-group of MFMAs (>= 512
// MFMA-pipe cycles) instead of between two K-steps, and the staging instructions are spread over the step.  After
// the barrier nobody reads the current stage any more (group 3 is already in registers), which is what lets the
// next step overwrite it.

// ------------------------------------------------------------------------------------------------
// K1: S = sd(R, X+ W+^T + X- W-^T)        SWAP exchanges W+ / W- (inhibitor term, beta != 0)
//     ZM_LRP: S1 = sd(R, X+ W+^T), S2 = sd(R, X- W-^T) kept apart (layers_lrp.py:199-200)
//     ZM_FWD: S = sd(R, ((Y - bias) + |X| |W|^T) / 2)  -- one product; Y = forward output, bias may be NULL
// ------------------------------------------------------------------------------------------------
enum { ZM_OURS = 0, ZM_LRP = 1, ZM_FWD = 2 };
constexpr float kCancelTol = 0.0078125f;   // 2^-7: below this share of |X||W|^T the split sum is recomputed exactly

// plain positive-part sum of one output element (the reference's Z), k-ordered
__device__ __noinline__ float exact_z(const float* __restrict__ x, const float* __restrict__ w, int64_t K) {
  float z1 = 0.0f, z2 = 0.0f;
  for (int64_t k = 0; k < K; ++k) {
    const float xv = x[k], wv = w[k];
    z1 = fmaf(fmaxf(xv, 0.0f), fmaxf(wv, 0.0f), z1);
    z2 = fmaf(fminf(xv, 0.0f), fminf(wv, 0.0f), z2);
  }
  return z1 + z2;
}

// Deferred per-sample factor on the relevance operand (te_add_relprop_deferred_f32 hands out b = X1.S unscaled plus
// fac[sample]): row t of R enters the rule as R[t,:] * s[(t / rps) * stride] -- the product the Add's apply pass would
// have stored.  s == NULL: no factor.
struct RowScale {
  const float* s;
  int64_t stride;
  int rps;          // rows per sample
};
// factors of the (up to 32) rows gr0 + crow(e) of one 32x32 accumulator block
__device__ __forceinline__ void row_factors(const RowScale& rs, int64_t gr0, int64_t T, float (&f)[16]) {
  if (rs.rps >= 40) {      // a 36-row span (4*kh + crow < 36) touches at most two samples
    const int64_t b0 = gr0 / rs.rps, edge = (b0 + 1) * rs.rps;
    const float f0 = rs.s[b0 * rs.stride];
    const float f1 = (edge < T) ? rs.s[(b0 + 1) * rs.stride] : f0;
#pragma unroll
    for (int e = 0; e < 16; ++e) f[e] = (gr0 + (e & 3) + 8 * (e >> 2) < edge) ? f0 : f1;
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int64_t gr = min(gr0 + (e & 3) + 8 * (e >> 2), T - 1);
      f[e] = rs.s[(gr / rs.rps) * rs.stride];
    }
  }
}

template <int ZM, bool SWAP, int BM, int BN, int VAR = 0>
__global__ __launch_bounds__(kThreads, 2) void linear_k1_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ R,
    const float* __restrict__ Y, const float* __restrict__ bias,
    float* __restrict__ S1, float* __restrict__ S2, int64_t T, int64_t K, int64_t Nn, int nbn, int ntiles,
    RowScale rs) {
  constexpr bool LRP = (ZM == ZM_LRP), FWD = (ZM == ZM_FWD);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MI = BM / 64, NI = BN / 64;   // 32x32 MFMA blocks per wave
  constexpr int WM = BM / 2, WN = BN / 2;     // rows / columns per wave
  constexpr int A_SZ = BM * LDT;
  constexpr int STAGE = (BM + BN) * LDT;      // floats per pipeline stage: [A tile | B tile]
  static_assert(!(FWD && SWAP), "the forward-output Z-pass has no inhibitor form");

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, kh = lane >> 5;
  const TileCoord tc = tile_coord<BM, BN>(blockIdx.x, ntiles, nbn);
  const int nk = (int)((K + BK - 1) / BK);
  const bool interior = (tc.row0 + BM <= T) && (tc.col0 + BN <= Nn) && (K % BK == 0);   // block-uniform

  constexpr int NACC = LRP ? 2 : 1;
  f32x16 acc[NACC][MI][NI];
#pragma unroll
  for (int s = 0; s < NACC; ++s)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][mi][ni][e] = 0.0f;

  struct Frag {
    f32x4 a[MI], b[NI];
  };
  auto read_frag = [&](Frag& f, int stage, int kg) __attribute__((always_inline)) {
    const float* a_tile = smem + stage * STAGE;
    const float* b_tile = a_tile + A_SZ;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      f.a[mi] = *reinterpret_cast<const f32x4*>(a_tile + swz(wm * WM + mi * 32 + lr, kg * 2 + kh));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      f.b[ni] = *reinterpret_cast<const f32x4*>(b_tile + swz(wn * WN + ni * 32 + lr, kg * 2 + kh));
  };
  auto mma_group = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (FWD) {
        // |X| and |W| were formed when the tiles were staged (abs_regs): a plain GEMM inner loop, no VALU
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[0][mi][ni] = TE_MFMA(f.a[mi][j], f.b[ni][j], acc[0][mi][ni]);
      } else {
        float ap[MI], an[MI], bp[NI], bn[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          ap[mi] = te_pos(f.a[mi][j]);
          an[mi] = te_neg(f.a[mi][j]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float p = te_pos(f.b[ni][j]), n = te_neg(f.b[ni][j]);
          bp[ni] = SWAP ? n : p;   // partner of X+
          bn[ni] = SWAP ? p : n;   // partner of X-
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[0][mi][ni] = TE_MFMA(ap[mi], bp[ni], acc[0][mi][ni]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[NACC - 1][mi][ni] = TE_MFMA(an[mi], bn[ni], acc[NACC - 1][mi][ni]);
      }
    }
  };

  auto k_loop = [&](auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    f32x4 ra[BM / 32], rb[BN / 32];
    auto load_next = [&](int kt) __attribute__((always_inline)) {
      if constexpr (FAST) {
        load_rows_tile_fast<BM>(X, K, tc.row0, (int64_t)kt * BK, ra);
        load_rows_tile_fast<BN>(W, K, tc.col0, (int64_t)kt * BK, rb);
      } else {
        load_rows_tile<BM>(X, T, K, tc.row0, (int64_t)kt * BK, ra);
        load_rows_tile<BN>(W, Nn, K, tc.col0, (int64_t)kt * BK, rb);
      }
    };
    // one-product Z-pass: the operands are |X| and |W|; clearing the sign bits once per staged element (6 VALU ops
    // per thread and K-step, next to the LDS stores) instead of once per fragment use inside the MFMA stream
    auto abs_regs = [&]() __attribute__((always_inline)) {
      if constexpr (FWD) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) ra[i][e] = te_abs(ra[i][e]);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[i][e] = te_abs(rb[i][e]);
      }
    };
    if constexpr (FWD && (VAR == 2)) {
      // prefetch distance 2: two register stages; the loads of K-step kt+2 are issued right after the barrier that
      // published step kt+1 and are consumed (abs + ds_write) at the end of step kt+1 -- two K-steps of MFMAs cover
      // one global round trip instead of three quarters of one
      f32x4 qa[BM / 32], qb[BN / 32];
      auto load_into = [&](int kt, f32x4 (&xa)[BM / 32], f32x4 (&xb)[BN / 32]) __attribute__((always_inline)) {
        if constexpr (FAST) {
          load_rows_tile_fast<BM>(X, K, tc.row0, (int64_t)kt * BK, xa);
          load_rows_tile_fast<BN>(W, K, tc.col0, (int64_t)kt * BK, xb);
        } else {
          load_rows_tile<BM>(X, T, K, tc.row0, (int64_t)kt * BK, xa);
          load_rows_tile<BN>(W, Nn, K, tc.col0, (int64_t)kt * BK, xb);
        }
      };
      auto abs_store = [&](int stage, f32x4 (&xa)[BM / 32], f32x4 (&xb)[BN / 32]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) xa[i][e] = te_abs(xa[i][e]);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) xb[i][e] = te_abs(xb[i][e]);
        store_rows_tile<BM>(smem + stage * STAGE, xa);
        store_rows_tile<BN>(smem + stage * STAGE + A_SZ, xb);
      };
      load_into(0, ra, rb);
      abs_store(0, ra, rb);
      __syncthreads();
      if (nk > 1) load_into(1, ra, rb);       // ra/rb hold odd steps' successors alternately: see body()
      if (nk > 2) load_into(2, qa, qb);
      Frag f0, f1;
      read_frag(f0, 0, 0);
      // body(kt, cur, nxt regs = the set holding step kt+1): compute stage cur, publish kt+1, refill the set with kt+3
      auto body = [&](int kt, int cur, f32x4 (&xa)[BM / 32], f32x4 (&xb)[BN / 32]) __attribute__((always_inline)) {
        const bool more = kt + 1 < nk;
        read_frag(f1, cur, 1);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f0, cur, 2);
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f1, cur, 3);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          abs_store(cur ^ 1, xa, xb);
          __syncthreads();
          if (kt + 3 < nk) load_into(kt + 3, xa, xb);
          read_frag(f0, cur ^ 1, 0);
        }
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
      };
      // steady state: four K-steps of straight-line code per trip, every refill unconditional.  (hipcc's waitcnt
      // insertion loses the age order of in-flight loads across a loop back-edge and then drains ALL of them before
      // the first ds_write -- inside the straight-line stretch it knows that the stage being published is the older
      // one and waits vmcnt(4) only, which is what makes the prefetch distance really 2.)
      auto steady = [&](int kt, int cur, f32x4 (&xa)[BM / 32], f32x4 (&xb)[BN / 32]) __attribute__((always_inline)) {
        read_frag(f1, cur, 1);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f0, cur, 2);
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f1, cur, 3);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        abs_store(cur ^ 1, xa, xb);
        __syncthreads();
        load_into(kt + 3, xa, xb);
        read_frag(f0, cur ^ 1, 0);
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
      };
      int kt = 0;
      for (; kt + 7 <= nk; kt += 4) {          // kt + 3 + 3 < nk for all four steps
        steady(kt, 0, ra, rb);
        steady(kt + 1, 1, qa, qb);
        steady(kt + 2, 0, ra, rb);
        steady(kt + 3, 1, qa, qb);
      }
      for (; kt < nk; kt += 2) {
        body(kt, 0, ra, rb);
        if (kt + 1 < nk) body(kt + 1, 1, qa, qb);
      }
      return;
    }
    load_next(0);
    abs_regs();
    store_rows_tile<BM>(smem, ra);
    store_rows_tile<BN>(smem + A_SZ, rb);
    __syncthreads();
    Frag f0, f1;
    read_frag(f0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      read_frag(f1, cur, 1);
      mma_group(f0);
      if (more) load_next(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f0, cur, 2);
      mma_group(f1);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f1, cur, 3);
      mma_group(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        abs_regs();
        store_rows_tile<BM>(smem + (cur ^ 1) * STAGE, ra);
        store_rows_tile<BN>(smem + (cur ^ 1) * STAGE + A_SZ, rb);
        __syncthreads();
        read_frag(f0, cur ^ 1, 0);
      }
      mma_group(f1);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
  };
  if (interior) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
  if constexpr (FWD && VAR >= 1) {
    if (interior) {
      // all 32 loads of a 32x32 block in flight before the first use (the generic loop below waits per element:
      // its rare exact_z call sits between consecutive loads)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int64_t gc = tc.col0 + wn * WN + ni * 32 + lr;
          const int64_t gr0 = tc.row0 + wm * WM + mi * 32 + 4 * kh;
          float rr[16], yy[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t at = (gr0 + (e & 3) + 8 * (e >> 2)) * Nn + gc;
            rr[e] = R[at];
            yy[e] = Y[at];
          }
          const float bb = bias ? bias[gc] : 0.0f;
          if (rs.s) {
            float fr[16];
            row_factors(rs, gr0, T, fr);
#pragma unroll
            for (int e = 0; e < 16; ++e) rr[e] = rr[e] * fr[e];
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int64_t gr = gr0 + (e & 3) + 8 * (e >> 2);
            const float a_abs = acc[0][mi][ni][e];
            float z = 0.5f * ((yy[e] - bb) + a_abs);
            if (!(z > kCancelTol * a_abs)) z = exact_z(X + gr * K, W + gc * K, K);
            S1[gr * Nn + gc] = te_sd(rr[e], z);
          }
        }
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t gc = tc.col0 + wn * WN + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = tc.row0 + wm * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (gr < T && gc < Nn) {
          float r = R[gr * Nn + gc];
          if (rs.s) r = r * rs.s[(gr / rs.rps) * rs.stride];
          if constexpr (LRP) {
            S1[gr * Nn + gc] = te_sd(r, acc[0][mi][ni][e]);
            S2[gr * Nn + gc] = te_sd(r, acc[1][mi][ni][e]);
          } else if constexpr (FWD) {
            const float a_abs = acc[0][mi][ni][e];                          // |X| |W|^T  >= 0
            const float lin = Y[gr * Nn + gc] - (bias ? bias[gc] : 0.0f);   // X W^T
            float z = 0.5f * (lin + a_abs);
            if (!(z > kCancelTol * a_abs)) z = exact_z(X + gr * K, W + gc * K, K);   // cancellation / all-zero row
            S1[gr * Nn + gc] = te_sd(r, z);
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
//     MODE 3: z^B rule of the patch-embedding convolution (Conv2d.relprop, layers_ours.py:242-256): same two
//             products P = S W+, N = S W-, epilogue out = x (P + N) - l_b P - h_b N with x read from / out written
//             to the NCHW image through the patch geometry (no im2col copy); l_b, h_b = sample b's pixel min / max
// ------------------------------------------------------------------------------------------------
template <int MODE, bool SWAP, bool ACCUM, int BM, int BN, bool GLDS = false>
__global__ __launch_bounds__(kThreads, 2) void linear_k2_kernel(
    const float* __restrict__ S, const float* __restrict__ W, const float* __restrict__ X,
    float* __restrict__ out, int64_t T, int64_t K, int64_t Nn, int nbn, int ntiles, float scale, TeZbGeom zb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MI = BM / 64, NI = BN / 64, WM = BM / 2, WN = BN / 2;
  constexpr int A_SZ = BM * LDT, B_SZ = BK * BN;
  constexpr int STAGE = A_SZ + B_SZ;   // floats per pipeline stage: [A tile | B tile]

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, kh = lane >> 5;
  const TileCoord tc = tile_coord<BM, BN>(blockIdx.x, ntiles, nbn);
  const int nk = (int)((K + BK - 1) / BK);
  const bool interior = (tc.row0 + BM <= T) && (tc.col0 + BN <= Nn) && (K % BK == 0);   // block-uniform

  constexpr bool BOTH = (MODE == 0 || MODE == 3);
  constexpr int NACC = BOTH ? 2 : 1;
  f32x16 acc[NACC][MI][NI];
#pragma unroll
  for (int s = 0; s < NACC; ++s)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][mi][ni][e] = 0.0f;

  struct Frag {
    f32x4 a[MI];
    float b[4][NI];   // W[k][n] for the 4 k of the group
  };
  auto read_frag = [&](Frag& f, int stage, int kg) __attribute__((always_inline)) {
    const float* a_tile = smem + stage * STAGE;
    const float* b_base = a_tile + A_SZ + (kh * 4) * BN + wn * WN + lr;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      f.a[mi] = *reinterpret_cast<const f32x4*>(a_tile + swz(wm * WM + mi * 32 + lr, kg * 2 + kh));
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) f.b[j][ni] = b_base[(kg * 8 + j) * BN + ni * 32];
  };
  auto mma_group = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float bp[NI], bn[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float p = te_pos(f.b[j][ni]), n = te_neg(f.b[j][ni]);
        bp[ni] = SWAP ? n : p;
        bn[ni] = SWAP ? p : n;
      }
      if constexpr (BOTH || MODE == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[0][mi][ni] = TE_MFMA(f.a[mi][j], bp[ni], acc[0][mi][ni]);
      }
      if constexpr (BOTH || MODE == 2) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[NACC - 1][mi][ni] = TE_MFMA(f.a[mi][j], bn[ni], acc[NACC - 1][mi][ni]);
      }
    }
  };

  auto k_loop = [&](auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    if constexpr (FAST && GLDS) {
      // interior tiles, direct-to-LDS staging: the next K-step's tiles are requested at the top of the step (its
      // stage was released by the barrier that ended the previous step) and are waited for by the step's one barrier
      const int wave_base = __builtin_amdgcn_readfirstlane(threadIdx.x & ~63);
      auto stage = [&](int st, int kt) __attribute__((always_inline)) {
        glds_rows_tile<BM>(smem + st * STAGE, S, K, tc.row0, (int64_t)kt * BK, wave_base);
        glds_kn_tile<BN>(smem + st * STAGE + A_SZ, W, Nn, (int64_t)kt * BK, tc.col0, wave_base);
      };
      stage(0, 0);
      __syncthreads();
      Frag f0, f1;
      read_frag(f0, 0, 0);
      int cur = 0;
      for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) stage(cur ^ 1, kt + 1);
        read_frag(f1, cur, 1);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f0, cur, 2);
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(f1, cur, 3);
        mma_group(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          __syncthreads();
          read_frag(f0, cur ^ 1, 0);
        }
        mma_group(f1);
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
      }
      return;
    }
    f32x4 ra[BM / 32], rb[BN / 32];
    auto load_next = [&](int kt) __attribute__((always_inline)) {
      if constexpr (FAST) {
        load_rows_tile_fast<BM>(S, K, tc.row0, (int64_t)kt * BK, ra);
        load_kn_tile_fast<BN>(W, Nn, (int64_t)kt * BK, tc.col0, rb);
      } else {
        load_rows_tile<BM>(S, T, K, tc.row0, (int64_t)kt * BK, ra);
        load_kn_tile<BN>(W, K, Nn, (int64_t)kt * BK, tc.col0, rb);
      }
    };
    load_next(0);
    store_rows_tile<BM>(smem, ra);
    store_kn_tile<BN>(smem + A_SZ, rb);
    __syncthreads();
    Frag f0, f1;
    read_frag(f0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      read_frag(f1, cur, 1);
      mma_group(f0);
      if (more) load_next(kt + 1);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f0, cur, 2);
      mma_group(f1);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f1, cur, 3);
      mma_group(f0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        store_rows_tile<BM>(smem + (cur ^ 1) * STAGE, ra);
        store_kn_tile<BN>(smem + (cur ^ 1) * STAGE + A_SZ, rb);
        __syncthreads();
        read_frag(f0, cur ^ 1, 0);
      }
      mma_group(f1);
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
  };
  if (interior) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  if constexpr (MODE == 0 && !ACCUM) {
    if (interior) {
      // interior tiles: the 16 X values of a 32x32 block are all requested before the first is used (the guarded loop
      // below issues each load behind a bounds test and in front of a store)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int64_t gc = tc.col0 + wn * WN + ni * 32 + lr;
          const int64_t gr0 = tc.row0 + wm * WM + mi * 32 + 4 * kh;
          float xv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) xv[e] = X[(gr0 + (e & 3) + 8 * (e >> 2)) * Nn + gc];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float xp = fmaxf(xv[e], 0.0f), xn = fminf(xv[e], 0.0f);
            out[(gr0 + (e & 3) + 8 * (e >> 2)) * Nn + gc] = scale * (xp * acc[0][mi][ni][e] + xn * acc[1][mi][ni][e]);
          }
        }
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int64_t gc = tc.col0 + wn * WN + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t gr = tc.row0 + wm * WM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if constexpr (MODE == 3) {
          if (gr < T && gc < Nn) {
            const int64_t b = gr / zb.P;
            const int64_t at = te_zb_index(zb, gr, gc);
            const float Pp = acc[0][mi][ni][e], Pn = acc[1][mi][ni][e];
            out[at] = (X[at] * (Pp + Pn) - zb.lohi[2 * b] * Pp) - zb.lohi[2 * b + 1] * Pn;
          }
        } else if (gr < T && gc < Nn) {
          const float x = X[gr * Nn + gc];
          const float xp = fmaxf(x, 0.0f), xn = fminf(x, 0.0f);
          float val;
          if constexpr (MODE == 0) val = xp * acc[0][mi][ni][e] + xn * acc[1][mi][ni][e];
          else if constexpr (MODE == 1) val = xp * acc[0][mi][ni][e];
          else val = xn * acc[0][mi][ni][e];
          val = scale * val;
          if constexpr (ACCUM) val = out[gr * Nn + gc] - val;          // alpha*act - beta*inh
          else if constexpr (MODE == 2) val = out[gr * Nn + gc] + val;  // C1 + C2 of the lrp variant
          out[gr * Nn + gc] = val;
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

// ------------------------------------------------------------------------------------------------
// host side: tile choice and launches
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
constexpr size_t k1_lds() { return (size_t)2 * (BM + BN) * LDT * sizeof(float); }          // 64 / 48 / 32 KB
template <int BM, int BN>
constexpr size_t k2_lds() { return (size_t)2 * (BM * LDT + BK * BN) * sizeof(float); }      // 64 / 48 / 32 KB

template <typename Kern>
inline void allow_lds(Kern kern, size_t bytes) {
  // > 64 KiB of dynamic LDS must be opted into; cheap and idempotent, no device state besides the
  // function attribute.
  if (bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
}

enum Tile { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x64 = 2 };

// Tile choice per launch.  A launch whose tile count is just above a multiple of the CU count wastes most of its
// last round: 12,608 x 768 outputs are 594 tiles of 128x128 = 2.32 per CU (the third round a third full) but 1188
// of 128x64 = 4.64 per CU.  Estimated chip efficiency of a tile = (tiles / CUs) / ceil(tiles / CUs) x a per-tile
// factor; the best estimate wins.  TE_LINEAR_TILE = 128x128 | 128x64 | 64x64 pins the choice (tuning).
inline Tile pick_tile(int64_t T, int64_t n_out, bool one_product = false) {
#ifdef TE_STUDY      // measurement builds only (TE_BUILD_DEFINES=TE_STUDY): the shipped library reads no environment
  static const int pinned = [] {
    const char* e = getenv("TE_LINEAR_TILE");
    if (!e) return -1;
    if (!strcmp(e, "128x128")) return (int)TILE_128x128;
    if (!strcmp(e, "128x64")) return (int)TILE_128x64;
    if (!strcmp(e, "64x64")) return (int)TILE_64x64;
    return -1;
  }();
  if (pinned >= 0) return (Tile)pinned;
#endif
  auto eff = [&](int bm, int bn, double factor) {
    const int64_t tiles = te_ceil_div(T, bm) * te_ceil_div(n_out, bn);
    return factor * ((double)tiles / kCUs) / (double)te_ceil_div(tiles, kCUs);
  };
  // the one-product Z-pass has half the MFMAs per K-step and is bound by the latency of its global loads: the more
  // co-resident waves the better (64x64: 5 blocks per CU), measured 64x64 >= 128x64 > 128x128 on every ViT-B shape
  // (profiles/r01_zfwd_variants.log)
  const double e0 = eff(128, 128, one_product ? 0.85 : 1.0), e1 = eff(128, 64, one_product ? 0.97 : 0.98),
               e2 = eff(64, 64, one_product ? 1.0 : 0.93);
  if (e2 > e1 && e2 > e0) return TILE_64x64;
  return (e1 > e0) ? TILE_128x64 : TILE_128x128;
}

template <int ZM, bool SWAP, int BM, int BN, int VAR = 0>
inline void launch_k1v(const float* X, const float* W, const float* R, const float* Y, const float* bias, float* S1,
                       float* S2, int64_t T, int64_t in_f, int64_t out_f, hipStream_t stream, RowScale rs) {
  const int nbn = (int)te_ceil_div(out_f, BN);
  const int ntiles = (int)te_ceil_div(T, BM) * nbn;
  constexpr size_t lds = k1_lds<BM, BN>();
  allow_lds(linear_k1_kernel<ZM, SWAP, BM, BN, VAR>, lds);
  linear_k1_kernel<ZM, SWAP, BM, BN, VAR><<<dim3((unsigned)ntiles), dim3(kThreads), lds, stream>>>(
      X, W, R, Y, bias, S1, S2, T, in_f, out_f, nbn, ntiles, rs);
}
template <int ZM, bool SWAP, int BM, int BN>
inline void launch_k1(const float* X, const float* W, const float* R, const float* Y, const float* bias, float* S1,
                      float* S2, int64_t T, int64_t in_f, int64_t out_f, hipStream_t stream,
                      RowScale rs = RowScale{nullptr, 0, 1}) {
  if constexpr (ZM == ZM_FWD) {
    // TE_ZFWD_VARIANT (tuning study, profiles/r01_zfwd_variants.log): 0 = per-element epilogue, distance-1 prefetch;
    // 1 = batched epilogue loads (-3.5 %); 2 = 1 + prefetch distance 2 (-6.3 %, default)
#ifdef TE_STUDY      // variants 0 / 1 (one of which spills) exist in measurement builds only
    static const int var = [] {
      const char* e = getenv("TE_ZFWD_VARIANT");
      return e ? atoi(e) : 2;
    }();
    switch (var) {
      case 0: return launch_k1v<ZM, SWAP, BM, BN, 0>(X, W, R, Y, bias, S1, S2, T, in_f, out_f, stream, rs);
      case 1: return launch_k1v<ZM, SWAP, BM, BN, 1>(X, W, R, Y, bias, S1, S2, T, in_f, out_f, stream, rs);
      default: break;
    }
#endif
    return launch_k1v<ZM, SWAP, BM, BN, 2>(X, W, R, Y, bias, S1, S2, T, in_f, out_f, stream, rs);
  }
  launch_k1v<ZM, SWAP, BM, BN, 0>(X, W, R, Y, bias, S1, S2, T, in_f, out_f, stream, rs);
}
template <int MODE, bool SWAP, bool ACCUM, int BM, int BN>
inline void launch_k2(const float* S, const float* W, const float* X, float* out, int64_t T, int64_t in_f,
                      int64_t out_f, float scale, hipStream_t stream, TeZbGeom zb = TeZbGeom{}) {
  const int nbn = (int)te_ceil_div(in_f, BN);
  const int ntiles = (int)te_ceil_div(T, BM) * nbn;
  constexpr size_t lds = k2_lds<BM, BN>();
  if constexpr (MODE == 0 && !SWAP && !ACCUM) {
#ifdef TE_STUDY      // TE_CPASS_GLDS=1: direct-to-LDS staging of the interior tiles (tuning study; see glds16) -- measurement builds
    static const bool glds = [] {
      const char* e = getenv("TE_CPASS_GLDS");
      return e && atoi(e) != 0;
    }();
    if (glds) {
      allow_lds(linear_k2_kernel<MODE, SWAP, ACCUM, BM, BN, true>, lds);
      linear_k2_kernel<MODE, SWAP, ACCUM, BM, BN, true><<<dim3((unsigned)ntiles), dim3(kThreads), lds, stream>>>(
          S, W, X, out, T, out_f, in_f, nbn, ntiles, scale, zb);
      return;
    }
#endif
  }
  allow_lds(linear_k2_kernel<MODE, SWAP, ACCUM, BM, BN>, lds);
  linear_k2_kernel<MODE, SWAP, ACCUM, BM, BN><<<dim3((unsigned)ntiles), dim3(kThreads), lds, stream>>>(
      S, W, X, out, T, out_f, in_f, nbn, ntiles, scale, zb);
}

#define TE_DISPATCH_TILE(tile, CALL)                  \
  do {                                                \
    switch (tile) {                                   \
      case TILE_128x128: { CALL(128, 128); } break;   \
      case TILE_128x64: { CALL(128, 64); } break;     \
      default: { CALL(64, 64); } break;               \
    }                                                 \
  } while (0)

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
  const Tile t1 = pick_tile(T, out_f), t2 = pick_tile(T, in_f);
  if (lrp) {
    // the lrp variant's K1 holds two accumulator sets; out - beta*(C1 + C2) is formed in two accumulate steps
#define TE_K1(BM_, BN_) launch_k1<ZM_LRP, SWAP, BM_, BN_>(X, W, R, nullptr, nullptr, S1, S2, T, in_f, out_f, stream)
    TE_DISPATCH_TILE(t1, TE_K1);
#undef TE_K1
#define TE_K2(BM_, BN_)                                                                   \
  launch_k2<1, SWAP, ACCUM, BM_, BN_>(S1, W, X, out, T, in_f, out_f, scale, stream);      \
  launch_k2<2, SWAP, ACCUM, BM_, BN_>(S2, W, X, out, T, in_f, out_f, scale, stream)
    TE_DISPATCH_TILE(t2, TE_K2);
#undef TE_K2
  } else {
#define TE_K1(BM_, BN_) launch_k1<ZM_OURS, SWAP, BM_, BN_>(X, W, R, nullptr, nullptr, S1, S1, T, in_f, out_f, stream)
    TE_DISPATCH_TILE(t1, TE_K1);
#undef TE_K1
#define TE_K2(BM_, BN_) launch_k2<0, SWAP, ACCUM, BM_, BN_>(S1, W, X, out, T, in_f, out_f, scale, stream)
    TE_DISPATCH_TILE(t2, TE_K2);
#undef TE_K2
  }
  return TE_OK;
}

}  // namespace

// C-pass of the z^B rule (te_conv.hip): MODE 3 epilogue on the same tiled kernel; false when the shape needs the
// simple kernel there
bool te_internal_zb_cpass_tiled(const float* S, const float* W, const float* X, float* out, int64_t T, int64_t in_f,
                                int64_t out_f, const TeZbGeom& zb, hipStream_t stream) {
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(S) || !te_aligned16(W)) return false;
#define TE_K2(BM_, BN_) launch_k2<3, false, false, BM_, BN_>(S, W, X, out, T, in_f, out_f, 1.0f, stream, zb)
  TE_DISPATCH_TILE(pick_tile(T, in_f), TE_K2);
#undef TE_K2
  return true;
}

// ---- single-pass entry points (variant "ours", alpha = 1): the two kernels of te_linear_relprop_f32
// individually, so that a caller can bracket ONE kernel launch with events (bench.py roofline) or
// interleave other work between the passes.  S is the [T,out_f] scratch of the composed call.
extern "C" int te_linear_zpass_f32(const float* R, const float* X, const float* W, float* S, int64_t T,
                                   int64_t in_f, int64_t out_f, te_stream_t stream_) {
  if (!R || !X || !W || !S || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(R) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(S))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
#define TE_K1(BM_, BN_) launch_k1<ZM_OURS, false, BM_, BN_>(X, W, R, nullptr, nullptr, S, S, T, in_f, out_f, stream)
  TE_DISPATCH_TILE(pick_tile(T, out_f), TE_K1);
#undef TE_K1
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_linear_cpass_f32(const float* S, const float* X, const float* W, float* out, int64_t T,
                                   int64_t in_f, int64_t out_f, te_stream_t stream_) {
  if (!S || !X || !W || !out || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(S) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(out))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
#define TE_K2(BM_, BN_) launch_k2<0, false, false, BM_, BN_>(S, W, X, out, T, in_f, out_f, 1.0f, stream)
  TE_DISPATCH_TILE(pick_tile(T, in_f), TE_K2);
#undef TE_K2
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// ---- Z-pass from the forward output (see K1f in the file header), alone and composed with the C-pass
extern "C" int te_linear_zpass_fwd_scaled_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                              int64_t rows_per_sample, const float* X, const float* W, const float* Y,
                                              const float* bias, float* S, int64_t T, int64_t in_f, int64_t out_f,
                                              te_stream_t stream_) {
  if (!R || !X || !W || !Y || !S || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if (r_scale && (rows_per_sample <= 0 || rows_per_sample > 0x7fffffff || T % rows_per_sample)) return TE_ERR_INVALID_ARG;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(R) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(S))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const RowScale rs{r_scale, r_scale_stride, r_scale ? (int)rows_per_sample : 1};
#define TE_K1(BM_, BN_) launch_k1<ZM_FWD, false, BM_, BN_>(X, W, R, Y, bias, S, S, T, in_f, out_f, stream, rs)
  TE_DISPATCH_TILE(pick_tile(T, out_f, true), TE_K1);
#undef TE_K1
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_linear_zpass_fwd_f32(const float* R, const float* X, const float* W, const float* Y,
                                       const float* bias, float* S, int64_t T, int64_t in_f, int64_t out_f,
                                       te_stream_t stream_) {
  return te_linear_zpass_fwd_scaled_f32(R, nullptr, 0, 1, X, W, Y, bias, S, T, in_f, out_f, stream_);
}

extern "C" int te_linear_relprop_fwd_scaled_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                                int64_t rows_per_sample, const float* X, const float* W,
                                                const float* Y, const float* bias, float* out, int64_t T, int64_t in_f,
                                                int64_t out_f, void* ws, size_t ws_bytes, te_stream_t stream_) {
  if (!R || !X || !W || !Y || !out || T <= 0 || in_f <= 0 || out_f <= 0) return TE_ERR_INVALID_ARG;
  if (!ws || ws_bytes < te_linear_relprop_workspace_bytes(T, in_f, out_f, TE_VARIANT_OURS)) return TE_ERR_WORKSPACE;
  if (!te_aligned16(ws)) return TE_ERR_WORKSPACE;
  if ((in_f % 4) || (out_f % 4) || !te_aligned16(R) || !te_aligned16(X) || !te_aligned16(W) || !te_aligned16(out))
    return TE_ERR_UNSUPPORTED;   // callers fall back to te_linear_relprop_f32 (any shape)
  int rc = te_linear_zpass_fwd_scaled_f32(R, r_scale, r_scale_stride, rows_per_sample, X, W, Y, bias, (float*)ws, T,
                                          in_f, out_f, stream_);
  if (rc != TE_OK) return rc;
  return te_linear_cpass_f32((const float*)ws, X, W, out, T, in_f, out_f, stream_);
}

extern "C" int te_linear_relprop_fwd_f32(const float* R, const float* X, const float* W, const float* Y,
                                         const float* bias, float* out, int64_t T, int64_t in_f, int64_t out_f,
                                         void* ws, size_t ws_bytes, te_stream_t stream_) {
  return te_linear_relprop_fwd_scaled_f32(R, nullptr, 0, 1, X, W, Y, bias, out, T, in_f, out_f, ws, ws_bytes, stream_);
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
