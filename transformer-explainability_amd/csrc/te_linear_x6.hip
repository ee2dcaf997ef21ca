// te_linear_x6.hip -- Linear.relprop (modules/layers_ours.py:207-230, variant "ours", alpha = 1, Z from the cached
// forward output) with its three GEMM-shaped products on bf16 MFMAs at fp32 accuracy.
//
// An fp32 number is exactly the sum of three bf16 numbers, a = a0 + a1 + a2 (8 + 8 + 8 significand bits), and a product
// of two bf16 values is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Of the nine partial products of
// a b the six above 2^-24 |a||b| are kept, smallest first:
//        a b  ~=  a1 b1 + a0 b2 + a2 b0 + a0 b1 + a1 b0 + a0 b0          ("x6": what is dropped is below fp32 rounding)
// so every product of the rule is an fp32-accumulated sum of exact terms -- at least as accurate against fp64 as the
// fp32-MFMA kernels of te_linear.hip (tests/test_gpu_rules.py::test_linear_x6_*).
//
// Round 3 design:
//   plane tensors   an fp32 [R, K] operand lives as bf16 planes in FRAGMENT-MAJOR order
//                        P3[R / 32][K / 16][3 planes][kh 2][r 32][8 bf16]          (1 KiB per plane fragment)
//                   i.e. exactly the image one wave needs as an MFMA operand of a 32-row block and one K16 step: one
//                   global_load_lds_dwordx4 (64 lanes x 16 B, fully coalesced 1 KiB) stages it, one ds_read_b128 per
//                   lane reads it back conflict-free.  The C-pass weights interleave W+^T and W-^T per block:
//                        P6[in / 32][out / 16][sign 2][3 planes][1 KiB].
//   weights         split ONCE per weight version (te_linear_x6_prepare_weights_f32; the host caches the planes)
//   |X| planes      te_linear_x6_split_abs_f32 (one streaming pass)
//   one GEMM kernel x6_kernel<WM, MODE>: the product is evaluated TRANSPOSED, D[w][t] (weight rows x activation rows),
//                   so a lane of the 32x32 accumulator block owns ONE activation row t and runs of four consecutive
//                   features: R / Y / X are read and the fp32 result is stored as 16-byte pieces, and the Z-pass writes
//                   S straight into the plane layout of the C-pass's operand (halves exchanged with
//                   v_permlane32_swap) -- S never exists in fp32.
//                   Tile = (128 WM) weight rows x 256 activation rows, 4 WM waves of 128 x 64 (eight 32x32 blocks, 48
//                   MFMAs per K16 step), two LDS stages filled by direct-to-LDS loads one step ahead, one barrier per
//                   step.  C-pass: the "weight rows" of a wave are (32 i) x {+, -} pairs, P+ and P- side by side.
//   schedule        persistent grid, SEQUENTIAL stream-K: the (tile, K-step) iteration space is cut into equal
//                   contiguous ranges, one per workgroup.  A tile cut in two is computed as ONE k-ordered chain: the
//                   workgroup holding its head (k = 0 ...) runs that fragment FIRST and publishes the accumulators
//                   (agent-scope release), the workgroup holding its tail runs it LAST, starting from those accumulators
//                   (agent-scope acquire).  Every output therefore sees the same fused-multiply-add chain wherever the cut
//                   falls: a batch equals its samples run one by one, bit for bit, and no tile quantisation is left (at
//                   ViT-B batch 64 the 256-wide tiles of the eight launches of a block fill 59-94 % of whole rounds).
//                   Waits only ever go to a LOWER workgroup index of the same XCD slot order.
//                   Where the last round of whole tiles is nearly full (ceil(r) <= 1.15 r, r = tiles per workgroup) the
//                   ranges are cut at tile boundaries instead (launch_x6): the workgroups of an XCD then walk k in
//                   lock-step on shared operand panels and a K16 step takes 1.9 us instead of 2.1-2.6 us.
//   plain GEMM      MODE_G: out = X W^T + b on the same loop (te_gemm_x6_f32: the Linear layers' own forward product and
//                   input gradient); te_linear_x6_split_dual_f32 writes the planes of a layer input and
//                   of its absolute value in one pass, so the rule reuses what the forward product split.
// Round 4:
//   geometries      WM = 2 / 1 / 0: 256 x 256, 128 x 256 and 128 x 128 tiles (X6Geo); the last for launches with few weight rows
//   other rules     variant lrp (layers_lrp.py:188-211) and alpha != 1 (layers_ours.py:225-228) as further epilogue modes of
//                   the same loop: MODE_ZI, MODE_Z1, MODE_CI, MODE_X (te_linear_relprop_x6_general_f32)
//   loud hand-over  a wait for another workgroup's accumulators is bounded (250 ms), sets a sticky status word, poisons the
//                   tile with NaN and makes every later wait give up at once: never a plausible wrong result, never a hang
//   main loop       raw s_barrier + counted s_waitcnt (a __syncthreads() drains direct-to-LDS loads); NST = 3 (prefetch
//                   distance 2) and KSPLIT = 2 (two K segments per output) are measured studies, off by default
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "te_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kFrag = 1024;                   // one plane fragment: 32 rows x 16 k bf16 as [kh][r][8]
constexpr int kRB = 3 * kFrag;                // one 32-row block, one K16 step
constexpr int kMinFrag = 4;                   // a cut closer than this many K-steps to a tile boundary snaps onto it
constexpr float kCancelTol = 0.0078125f;      // as te_linear.hip
constexpr int kSpinMillis = 250;               // bounded wait for a predecessor's accumulators: never hang the GPU
constexpr int kErrWord = 1023;                 // flags[0 .. 767] = hand-over flags of a pass, flags[kErrWord] = its error word


// Z-pass, C-pass, plain GEMM out = B A^T + bias; round 4, for variant lrp and alpha != 1 (layers_lrp.py:188-211,
// layers_ours.py:225-228): MODE_ZI = the inhibitor's Z-pass of variant ours from the same product, Z' = ((Y - b) - |X||W|^T)/2
// = X+ W-^T + X- W+^T; MODE_Z1 = a one-sided Z-pass, S = sd(R, acc) (acc = X+ W+^T ...: same-sign terms, no cancellation);
// MODE_CI = the C-pass with the signs crossed, out += scale (X+ . (S W-) + X- . (S W+)); MODE_X = a product masked by one
// sign of X, out (+)= scale (X+- . acc).
enum { MODE_Z = 0, MODE_C = 1, MODE_G = 2, MODE_ZI = 3, MODE_Z1 = 4, MODE_CI = 5, MODE_X = 6 };

#define TE_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_f32(unsigned b) { return __uint_as_float(b << 16); }
// x = p[0] + p[1] + p[2] exactly: round to nearest even (v_cvt_pk_bf16_f32), subtract (the residual of a round-to-nearest
// bf16 is representable in fp32), repeat.  Pairs: p[q] = the packed bf16 pair (x0 low half, x1 high half) of plane q.
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned (&p)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    p[q] = u;
    x0 = x0 - __uint_as_float(u << 16);
    x1 = x1 - __uint_as_float(u & 0xffff0000u);
  }
}
__device__ __forceinline__ void split3(float x, unsigned (&p)[3]) {
  unsigned pk[3];
  split3_pk(x, 0.0f, pk);
#pragma unroll
  for (int q = 0; q < 3; ++q) p[q] = pk[q] & 0xffffu;
}

// safe_divide (te_common.h: te_sd) of two element pairs on packed fp32 instructions, as in te_attn_rc.hip: den = b + 1e-9 (one
// rounding), an exact-zero den replaced by 1e-9, a / den, zero where b == 0.  The quotient is the hardware's own expansion of an IEEE
// division without its range scaling (v_rcp_f32, one Newton step on the reciprocal, q = a rc, the exact residual r = a - den q by
// fma, q + r rc): correctly rounded wherever no intermediate leaves the normal range -- |den| >= 1e-16 by construction.
__device__ __forceinline__ f32x2 sd2(f32x2 a, f32x2 b) {
  f32x2 den = b + f32x2{1e-9f, 1e-9f};
  den[0] = (den[0] == 0.0f) ? 1e-9f : den[0];
  den[1] = (den[1] == 0.0f) ? 1e-9f : den[1];
  f32x2 rc = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 e = __builtin_elementwise_fma(-den, rc, f32x2{1.0f, 1.0f});
  rc = __builtin_elementwise_fma(e, rc, rc);
  f32x2 q = a * rc;
  const f32x2 r = __builtin_elementwise_fma(-den, q, a);
  q = __builtin_elementwise_fma(r, rc, q);
  q[0] = (b[0] != 0.0f) ? q[0] : 0.0f;
  q[1] = (b[1] != 0.0f) ? q[1] : 0.0f;
  return q;
}

enum { OP_ABS = 0, OP_POS = 1, OP_NEG = 2, OP_ID = 3 };
template <int OP>
__device__ __forceinline__ float apply_op(float x) {
  if constexpr (OP == OP_ID) return x;
  if constexpr (OP == OP_ABS) return __int_as_float(__float_as_int(x) & 0x7fffffff);
  const int b = __float_as_int(x);
  if constexpr (OP == OP_POS) return __int_as_float(b > 0 ? b : 0);
  return __int_as_float(b < 0 ? b : 0);
}

// ------------------------------------------------------------------------------------------------
// fp32 -> planes.  One 256-thread block = 32 rows x 8 K16 steps; thread (r = t & 31, step = t >> 5) reads 16
// consecutive k of its row and writes the six 16-B pieces (3 planes x 2 k-halves) of its (row, step): a half-wave
// writes 512 contiguous bytes per store.  TRANSPOSE: element (r, k) is src[k * R + r] (W^T operands of the C-pass);
// the destination then is the P6 layout with `sign` selecting the W+ / W- half.
// ------------------------------------------------------------------------------------------------
// dst_abs (optional, OP_ID only): the planes of |src| as well, from the same pass -- the three planes of |x| are the planes
// of x times sign(x) exactly (bf16 rounding is sign-symmetric and every residual negates with x), and sign(x) is the
// sign bit of plane 0: clear it there, flip the low planes' sign bits where it was set.
template <int OP, bool TRANSPOSE>
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                    int64_t R, int64_t K, int group, int sign,
                                                    unsigned char* __restrict__ dst_abs = nullptr) {
  const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t rb = blockIdx.x;
  const int64_t ks = (int64_t)blockIdx.y * 8 + sl;
  const int64_t nks = K >> 4;
  if (ks >= nks) return;
  const int64_t row = rb * 32 + r;
  float v[16];
  if (row < R) {
    if constexpr (!TRANSPOSE) {
      const float* s = src + row * K + ks * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(s + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * c + e] = q[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = src[(ks * 16 + e) * R + row];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.0f;
  }
  unsigned p[8][3];                    // [pair of consecutive k][plane]
#pragma unroll
  for (int e = 0; e < 8; ++e) split3_pk(apply_op<OP>(v[2 * e]), apply_op<OP>(v[2 * e + 1]), p[e]);
  unsigned char* d = dst + ((rb * nks + ks) * group + sign * 3) * kFrag + r * 16;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = p[4 * kh + e][q];
      *reinterpret_cast<u32x4*>(d + q * kFrag + kh * 512) = w;
    }
  if constexpr (OP == OP_ID) {
    if (dst_abs) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned m = p[e][0] & 0x80008000u;
        p[e][0] &= 0x7fff7fffu;
        // a zero residual is +0 whatever the sign of x (x - x = +0): flip the sign bit of non-zero halves only, so that
        // the planes are bit for bit those of split3(|x|).  (h & 0x7fff) + 0x7fff has bit 15 set iff the half is non-zero.
        p[e][1] ^= m & (((p[e][1] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
        p[e][2] ^= m & (((p[e][2] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
      }
      unsigned char* da = dst_abs + ((rb * nks + ks) * 3) * kFrag + r * 16;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = p[4 * kh + e][q];
          *reinterpret_cast<u32x4*>(da + q * kFrag + kh * 512) = w;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Producers that emit operand planes themselves (round 5): nn.GELU between the two Linear
// layers of an Mlp block (ViT_LRP.py:57-69, BERT.py: BertIntermediate).  Thread mapping and plane layout of split_kernel.
//   SRC_GELU_BWD  d_h = d_a . gelu'(h) goes straight into the planes of fc1's input-gradient product (te_gemm_x6_f32's
//                 x_planes): the fp32 d_h is never written or read (22 -> 14 B per element against the two passes it replaces)
//   SRC_GELU_FWD  a = gelu(h) is written as fp32 (the rule and the backward pass read it) AND as the signed planes of fc2's
//                 forward product plus the planes of |a| for fc2's rule (te_linear_x6_split_dual_f32's outputs, bit for bit)
// ------------------------------------------------------------------------------------------------
enum { SRC_GELU_BWD = 0, SRC_GELU_FWD = 1 };
template <int SRC>
__global__ __launch_bounds__(256) void gelu_split_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         float* __restrict__ y, unsigned char* __restrict__ dst,
                                                         unsigned char* __restrict__ dst_abs, int64_t R, int64_t K) {
  const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t rb = blockIdx.x;
  const int64_t ks = (int64_t)blockIdx.y * 8 + sl;
  const int64_t nks = K >> 4;
  if (ks >= nks) return;
  const int64_t row = rb * 32 + r;
  float v[16];
  if (row < R) {
    const int64_t at = row * K + ks * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + at + 4 * c);
      if constexpr (SRC == SRC_GELU_BWD) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + at + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * c + e] = te_gelu_grad(gv[e], xv[e]);
      } else {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[4 * c + e] = te_gelu(xv[e]);
        *reinterpret_cast<f32x4*>(y + at + 4 * c) = o;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.0f;
  }
  unsigned p[8][3];                    // [pair of consecutive k][plane]
#pragma unroll
  for (int e = 0; e < 8; ++e) split3_pk(v[2 * e], v[2 * e + 1], p[e]);
  auto store = [&](unsigned char* d) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = p[4 * kh + e][q];
        *reinterpret_cast<u32x4*>(d + q * kFrag + kh * 512) = w;
      }
  };
  store(dst + ((rb * nks + ks) * 3) * kFrag + r * 16);
  if constexpr (SRC == SRC_GELU_FWD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // planes of |a| from the planes of a: as split_kernel's dst_abs
      const unsigned m = p[e][0] & 0x80008000u;
      p[e][0] &= 0x7fff7fffu;
      p[e][1] ^= m & (((p[e][1] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
      p[e][2] ^= m & (((p[e][2] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
    }
    store(dst_abs + ((rb * nks + ks) * 3) * kFrag + r * 16);
  }
}

// The same producers with the elementwise work in a COALESCED mapping: a block's 32 rows x 128 k tile is read as 1024
// float4 (a wave covers 512 contiguous bytes of two rows per instruction, four values per thread and step -- the shape of the
// stand-alone gelu kernels), the results cross to split_kernel's (row, K16 step) mapping through LDS, and only the split and
// the plane stores run there.  gelu_split_kernel above does the erf / exp arithmetic on 16 values per thread behind
// row-strided loads; measured against it in round 5.
constexpr int kGsPitch = 132;               // floats per tile row in LDS: 128 + 4 (16-B reads of 32 rows hit every bank once)
template <int SRC>
__global__ __launch_bounds__(256) void gelu_split_lds_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                             float* __restrict__ y, unsigned char* __restrict__ dst,
                                                             unsigned char* __restrict__ dst_abs, int64_t R, int64_t K) {
  __shared__ __attribute__((aligned(16))) float tile[32 * kGsPitch];
  const int64_t rb = blockIdx.x;
  const int64_t k0 = (int64_t)blockIdx.y * 128;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int tr = i >> 5, c4 = i & 31;
    const int64_t row = rb * 32 + tr, col = k0 + 4 * c4;
    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < R && col < K) {                        // (K % 16 == 0: a float4 never straddles the end of a row)
      const int64_t at = row * K + col;
      const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + at));
      if constexpr (SRC == SRC_GELU_BWD) {
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + at));
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = te_gelu_grad(gv[e], xv[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = te_gelu(xv[e]);
        *reinterpret_cast<f32x4*>(y + at) = o;
      }
    }
    *reinterpret_cast<f32x4*>(&tile[tr * kGsPitch + 4 * c4]) = o;
  }
  __syncthreads();
  const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t ks = (int64_t)blockIdx.y * 8 + sl;
  const int64_t nks = K >> 4;
  if (ks >= nks) return;
  float v[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(&tile[r * kGsPitch + sl * 16 + 4 * c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * c + e] = q[e];
  }
  unsigned p[8][3];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3_pk(v[2 * e], v[2 * e + 1], p[e]);
  auto store = [&](unsigned char* d) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = p[4 * kh + e][q];
        *reinterpret_cast<u32x4*>(d + q * kFrag + kh * 512) = w;
      }
  };
  store(dst + ((rb * nks + ks) * 3) * kFrag + r * 16);
  if constexpr (SRC == SRC_GELU_FWD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned m = p[e][0] & 0x80008000u;
      p[e][0] &= 0x7fff7fffu;
      p[e][1] ^= m & (((p[e][1] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
      p[e][2] ^= m & (((p[e][2] & 0x7fff7fffu) + 0x7fff7fffu) & 0x80008000u);
    }
    store(dst_abs + ((rb * nks + ks) * 3) * kFrag + r * 16);
  }
}

// (measurement builds: TE_GELU_SPLIT=direct runs gelu_split_kernel instead; the shipped build does not instantiate it)
#ifdef TE_STUDY
static bool gelu_split_direct() {
  static const bool on = [] {
    const char* e = getenv("TE_GELU_SPLIT");
    return e && !strcmp(e, "direct");
  }();
  return on;
}
#endif

__global__ __launch_bounds__(256) void zero_words_kernel(u32x4* __restrict__ p) {
  p[(size_t)blockIdx.x * 256 + threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
}

// plain positive-part sum of one output element (the reference's Z), k-ordered: the cancellation fallback
__device__ __forceinline__ float exact_z(const float* __restrict__ x, const float* __restrict__ w, int64_t K) {
  float z1 = 0.0f, z2 = 0.0f;
  for (int64_t k = 0; k < K; ++k) {
    const float xv = x[k], wv = w[k];
    z1 = fmaf(fmaxf(xv, 0.0f), fmaxf(wv, 0.0f), z1);
    z2 = fmaf(fminf(xv, 0.0f), fminf(wv, 0.0f), z2);
  }
  return z1 + z2;
}

// the inhibitor's Z of variant ours, k-ordered: X+ W-^T + X- W+^T (layers_ours.py:226: f(nw, pw, px, nx))
__device__ __forceinline__ float exact_zi(const float* __restrict__ x, const float* __restrict__ w, int64_t K) {
  float z1 = 0.0f, z2 = 0.0f;
  for (int64_t k = 0; k < K; ++k) {
    const float xv = x[k], wv = w[k];
    z1 = fmaf(fmaxf(xv, 0.0f), fminf(wv, 0.0f), z1);
    z2 = fmaf(fminf(xv, 0.0f), fmaxf(wv, 0.0f), z2);
  }
  return z1 + z2;
}

struct X6Params {
  const unsigned char* A;      // weight-side planes: P3 of |W| (Z-pass) or P6 of W+^T / W-^T (C-pass)
  const unsigned char* B;      // activation-side planes (P3): |X| (Z-pass) or S (C-pass)
  int64_t a_group_stride;      // bytes between consecutive 32-row groups of A  = nks * (3 or 6) KiB
  int64_t b_rb_stride;         // bytes between consecutive 32-row blocks of B  = nks * 3 KiB
  int nks;                     // K / 16
  int ksplit;                  // 1, or 2: every output = chain(k < K/2) + chain(k >= K/2), two work items per tile (kseg_rule)
  float* seg_part;             // ksplit == 2: [tile][threads][accumulators] of the first segment ...
  unsigned* seg_flags;         // ... and [tile] "it is there" flags, zero before the launch

  int ntm, ntn;                // tiles along the weight side / the activation side (set by launch_x6 from rows_w, T)
  int rows_w;                  // weight-side rows of the product: out_f (Z-pass, plain product), 2 in_f (C-pass: +/- pairs)
  int t_fast;                  // tile order inside the launch: 0 = weight side fastest, 1 = activation side fastest
  int whole_tiles;             // 1: ranges are cut at tile boundaries only (all workgroups of a round in k lock-step)
  int whole_tiles_forced;      // study builds: whole_tiles was set by the caller (TE_X6_SNAP)
  int ncb;                     // 32-row blocks of the activation side = ceil(T / 32)
  int64_t T;
  int in_f, out_f;
  float* partial;              // [grid][threads][32] float4: accumulators of a cut tile
  unsigned* flags;             // [grid] hand-over flags, zero before the launch; [kErrWord] = error word of this pass
  unsigned* status;            // sticky error word (the caller's, else = flags + kErrWord): OR-ed, never cleared here
  long long spin_ticks;        // wall_clock64 ticks a workgroup waits for a predecessor before it gives up
  int drop_handover;           // TE_X6_TEST_DROP_HANDOVER: publishers keep their flag down (tests: a wait must expire)
  int small_grid;              // TE_X6_TEST_SMALL_GRID: 16 workgroups (tests: stream-K cuts on small shapes)
  int prefer_whole;            // TE_X6_WHOLE_TILES: cut the ranges at tile boundaries whatever the fill of the last round
  // Z-pass epilogue
  const float* R;
  const float* Y;
  const float* bias;
  const float* X;              // fp32 operands: the cancellation fallback (Z) / the sign select (C)
  const float* W;
  unsigned char* S;            // out: P3 planes of S, rows = T, K = out_f
  const float* rs;             // optional per-sample factor on R
  int64_t rs_stride;
  int rps;
  // C-pass epilogue
  float* out;
  float scale;                 // MODE_C / MODE_CI / MODE_X: factor on the result (alpha, -beta)
  int accum;                   // MODE_X: 1 = add to what out holds (MODE_CI always does)
  int x_sign;                  // MODE_X: +1 = X+ masks the product, -1 = X-
  int opt;                     // schedule options (kX6Opt*): results do not depend on them
};
// (round 6, measured and removed: requesting the NEXT tile's first stage before the epilogue, non-temporal epilogue accesses, and
// warm-up requests of the tile's R / Y lines during the last loop steps -- all neutral or negative, profiles/r06_x6_epilogue_study.log)
constexpr int kX6OptStaged = 8;            // Z modes, 128-row wave tiles: R / Y of the epilogue through the LDS, one cache line per load instruction
constexpr int kX6OptSkipEpilogue = 4;      // measurement builds: no epilogue (what it costs in place; garbage results)
constexpr int kX6OptDefault = kX6OptStaged;

// keeps a vector value alive without using it (study paths).  Device pass only: a "v" constraint in the HOST pass of a kernel
// template silently drops the host stub of the instantiation (the ablation kernels 1-4 / 6 stopped linking that way).
#if defined(__HIP_DEVICE_COMPILE__)
#define X6_KEEP(v) asm volatile("" ::"v"(v))
#define X6_PASS4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))      // a no-op four loaded vectors pass through
#else
#define X6_KEEP(v) (void)(v)
#define X6_PASS4(a, b, c, d) (void)0
#endif

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void swap_halves(unsigned& lo_keep, unsigned& hi_keep) {
  // after the call: lanes 0-31 hold {own lo_keep, partner's lo_keep}; lanes 32-63 hold {partner's hi_keep, own hi_keep}
  // (v_permlane32_swap: lanes 32-63 of the first operand <-> lanes 0-31 of the second)
  const u32x2 r = __builtin_amdgcn_permlane32_swap(lo_keep, hi_keep, false, false);
  lo_keep = r[0];
  hi_keep = r[1];
}

// ------------------------------------------------------------------------------------------------
// STUDY (only instantiated under -DTE_X6_STUDY, benchmarks/x6_bench.py --study): timing-only ablations of the main loop --
// 1: no global loads after the first stage; 2: + no barrier; 3: + no LDS reads (MFMAs on resident fragments);
// 4: the shipped loop without the epilogue.  Their results are garbage by construction.
// NST = LDS stages: 2 (stage ks + 1 lands while ks is multiplied) or 3 (prefetch distance 2: a fill that misses the
// XCD's L2 -- workgroups at different k offsets of shared panels, i.e. every stream-K launch -- has two steps to land).
// Tile geometry WM (weight rows x activation rows of a workgroup tile; waves as NWM x NWN, each MI x 2 blocks of 32 x 32):
//   2: 256 x 256, 2 x 4 waves of 128 x 64 (one 512-thread workgroup per CU)
//   1: 128 x 256, 1 x 4 waves of 128 x 64 (two 256-thread workgroups per CU)
//   0: 128 x 128, 2 x 2 waves of  64 x 64 (three 256-thread workgroups per CU, 168 VGPRs) -- round 4, for launches with
//      few weight rows (out_f = 768: 150 / 300 tiles of the larger geometries for 256 CUs): every output is ONE k-ordered
//      chain, so a tile takes K / 16 sequential steps whatever the schedule, and with one 4-wave workgroup on most CUs a
//      SIMD held a single wave (1.38 us per step of 48 MFMAs: half the pipe idle).  Smaller tiles = shorter steps, 594
//      tiles, three waves per SIMD to cover each other's LDS / barrier / fill latency.
template <int WM>
struct X6Geo {
  static constexpr int NWM = (WM == 1) ? 1 : 2, NWN = (WM == 0) ? 2 : 4, MI = (WM == 0) ? 2 : 4;
  static constexpr int NW = NWM * NWN, THREADS = 64 * NW, WPS = (WM == 0) ? 3 : 2;       // waves, waves per SIMD aimed at
  static constexpr int TM = NWM * MI * 32, TT = NWN * 64;                                // tile: weight rows, activation rows
  static constexpr int NPA = 3 * NWM * MI, NPB = 3 * NWN * 2;                            // 1-KiB pieces of one stage
  static constexpr int ACC = MI * 2 * 16;                                                // accumulator floats per lane
  static constexpr int MAX_SPX = (WM == 2) ? 32 : (WM == 1) ? 64 : 96;                   // workgroups per XCD
};

// LDS of a workgroup: the stages, or -- Z modes of the 128-row wave tiles -- the epilogue's staging area if that is larger (16 KiB per
// wave: two buffers of one 32 x 32 block of R and of Y), then one 512-B bias slot per wave and the hand-over word
template <int WM, int MODE, int NST>
struct X6Lds {
  using GEO = X6Geo<WM>;
  static constexpr bool STAGED = (MODE == MODE_Z || MODE == MODE_ZI || MODE == MODE_Z1) && GEO::MI == 4;
  static constexpr int STAGES = NST * (GEO::NPA + GEO::NPB) * kFrag;
  static constexpr int MAIN = (STAGED && GEO::NW * 16384 > STAGES) ? GEO::NW * 16384 : STAGES;
  static constexpr int TOTAL = MAIN + GEO::NW * 512 + 16;
};

// KSPLIT = 2: a separate instantiation, so that launches without the K split carry none of its code or registers (with the
// split as a run-time branch the Z-pass epilogue spilled 277 VGPRs and every un-split launch ran 6-16 % slower, measured).
template <int WM, int MODE, int STUDY = 0, int NST = 2, int KSPLIT = 1>
__global__ __launch_bounds__(X6Geo<WM>::THREADS, X6Geo<WM>::WPS) void x6_kernel(const X6Params p) {
  static_assert(NST == 2 || NST == 3, "two or three LDS stages");
  using GEO = X6Geo<WM>;
  constexpr int NW = GEO::NW, NWM = GEO::NWM, NWN = GEO::NWN, MI = GEO::MI;
  constexpr bool IS_Z = (MODE == MODE_Z || MODE == MODE_ZI || MODE == MODE_Z1);
  constexpr bool IS_C = (MODE == MODE_C || MODE == MODE_CI);
  constexpr int G = IS_C ? 6 : 3;                    // pieces of one A group (32 rows [x 2 signs]) per K16 step
  constexpr int NPA = GEO::NPA, NPB = GEO::NPB;      // 1-KiB pieces of one stage: weight side, activation side
  constexpr int NP = NPA + NPB;
  static_assert(NPA / 3 == NW, "each wave stages one weight-side block");
  constexpr int PBW = NPB / 3 / NW;                  // activation-side 32-row blocks each wave stages (2 or 1)
  constexpr int STAGE = NP * kFrag;
  constexpr int LPS = 3 + 3 * PBW;                   // direct-to-LDS loads one stage_in issues per lane
  constexpr int GROUPS = NPA / G;                    // A groups per tile: 4 WM (Z) or 2 WM (C)
  constexpr bool FULL = (STUDY == 0 || STUDY >= 4);  // study builds: 5 = shipped + time stamps, 6 = no epilogue + stamps
  constexpr bool EPI = (STUDY == 0 || STUDY == 5), PROF = (STUDY == 5 || STUDY == 6);
  // Z modes on 128-row wave tiles stage the R / Y blocks of the epilogue through the (then idle) LDS: 16 KiB per wave (X6Staged)
  constexpr int LDS_MAIN = X6Lds<WM, MODE, NST>::MAIN;
  long long prof_loop = 0, prof_epi = 0, prof_pub = 0, prof_wait = 0, prof_t0 = 0, prof_t1 = 0, prof_steps = 0, prof_nepi = 0;
  const long long prof_start = PROF ? wall_clock64() : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int lane16 = lane * 16;

  // ---- this workgroup's range of the (tile, K-step) space: whole tiles per XCD, equal ranges inside one ----
  const int bid = blockIdx.x, spx = gridDim.x >> 3;
  const int xcd = bid & 7, slot = bid >> 3;
  // A launch with ksplit = 2 runs two work items ("virtual tiles") per tile, K segment 0 and K segment 1, each a k-ordered
  // chain of nks steps; an XCD walks all its segment-0 items, then its segment-1 items.
  const int nks = p.nks / KSPLIT;                    // steps of one work item
  const int tiles = p.ntm * p.ntn;
  const int tx0 = (int)((int64_t)tiles * xcd / 8), tx1 = (int)((int64_t)tiles * (xcd + 1) / 8);
  const int ntx = tx1 - tx0;
  const int64_t iters = (int64_t)ntx * KSPLIT * nks;
  auto cut = [&](int s) -> int64_t {
    int64_t b = iters * s / spx;
    const int r = (int)(b % nks);
    if (p.whole_tiles) return (2 * r < nks) ? b - r : b + (nks - r);
    if (r < kMinFrag) b -= r;
    else if (nks - r < kMinFrag) b += nks - r;
    return b;
  };
  const int64_t ib = cut(slot), ie = cut(slot + 1);
  if (ib >= ie) return;
  const int f_tile = (int)(ib / nks), kb = (int)(ib - (int64_t)f_tile * nks);
  const int l_tile = (int)((ie - 1) / nks), ke = (int)(ie - (int64_t)l_tile * nks);
  const bool tail = kb > 0;                                            // first tile: its head belongs to slot - 1
  const bool head = (ke < nks) && !(f_tile == l_tile && tail);         // last tile: its tail belongs to slot + 1
  const int full0 = tail ? f_tile + 1 : f_tile, full1 = (ke < nks) ? l_tile - 1 : l_tile;
  const int nseq = (head ? 1 : 0) + (full1 >= full0 ? full1 - full0 + 1 : 0) + (tail ? 1 : 0);

  float* my_part = p.partial + (size_t)bid * GEO::THREADS * GEO::ACC;
  const float* in_part = p.partial + (size_t)(bid - 8) * GEO::THREADS * GEO::ACC;

  for (int seq = 0; seq < nseq; ++seq) {
    int tile, k0, k1;
    if (head && seq == 0) {
      tile = l_tile, k0 = 0, k1 = ke;
    } else if (tail && seq == nseq - 1) {
      tile = f_tile, k0 = kb, k1 = (f_tile == l_tile) ? ke : nks;
    } else {
      tile = full0 + seq - (head ? 1 : 0), k0 = 0, k1 = nks;
    }
    const int seg = (KSPLIT == 2 && tile >= ntx) ? 1 : 0;      // K segment of this work item
    tile = tile - seg * ntx + tx0;
    const int kseg0 = seg * nks;                       // its first K16 step
    int tn, tm;
    if (p.t_fast) {
      tm = tile / p.ntn, tn = tile - tm * p.ntn;
    } else {
      tn = tile / p.ntm, tm = tile - tn * p.ntm;
    }

    // ---- what this wave stages per step: the three planes of ONE weight-side 32-row block [one sign] and of PBW
    //      activation-side blocks -- each 3 KiB contiguous in memory and in the stage; wave-uniform pointers ----
    const unsigned char* srcA;
    if constexpr (!IS_C)
      srcA = p.A + (int64_t)(tm * GROUPS + wave) * p.a_group_stride + (int64_t)(kseg0 + k0) * kRB;
    else
      srcA = p.A + (int64_t)(tm * GROUPS + (wave >> 1)) * p.a_group_stride + (int64_t)(kseg0 + k0) * (2 * kRB) + (wave & 1) * kRB;
    const unsigned char* srcB[PBW];
#pragma unroll
    for (int u = 0; u < PBW; ++u) {
      const int cb = min(tn * (GEO::TT / 32) + wave * PBW + u, p.ncb - 1);
      srcB[u] = p.B + (int64_t)cb * p.b_rb_stride + (int64_t)(kseg0 + k0) * kRB;
    }
    unsigned char* const ldsA = smem + wave * kRB;
    unsigned char* const ldsB = smem + NPA * kFrag + wave * (PBW * kRB);
    // adv = 0 re-requests the step already staged (branch-free last step: no control-flow join between the fragment
    // reads and the MFMAs, which is what lets hipcc wait for the reads piecemeal)
    auto stage_in = [&](int stg, int adv) __attribute__((always_inline)) {
      srcA += adv * (G * kFrag);
#pragma unroll
      for (int u = 0; u < PBW; ++u) srcB[u] += adv * kRB;
#pragma unroll
      for (int q = 0; q < 3; ++q) glds16(srcA + q * kFrag + lane16, ldsA + stg * STAGE + q * kFrag);
#pragma unroll
      for (int u = 0; u < PBW; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) glds16(srcB[u] + q * kFrag + lane16, ldsB + stg * STAGE + (u * 3 + q) * kFrag);
    };

    if constexpr (PROF) {
      if (threadIdx.x == 0) prof_t0 = wall_clock64();
    }
    // Bounded, loud wait for a flag another workgroup raises (release) once its accumulators are in memory; uniform
    // result.  On expiry: the caller's sticky status word and this pass's error word are set, `false` comes back.
    auto wait_for = [&](unsigned* flag) -> bool {
      unsigned* const ok_word = reinterpret_cast<unsigned*>(smem + LDS_MAIN + NW * 512);
      if (threadIdx.x == 0) {
        unsigned spins = 0, ok = 1u;
        const long long t_begin = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(8);
          if ((++spins & 63u) == 0u &&
              (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
               wall_clock64() - t_begin > p.spin_ticks)) {
            ok = 0u;
            break;
          }
        }
        if (!ok) {
          __hip_atomic_fetch_or(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.flags + kErrWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ok_word = ok;
      }
      __syncthreads();
      return __builtin_amdgcn_readfirstlane(*ok_word) != 0u;     // (a main loop's barriers lie between two waits)
    };
    f32x16 acc[MI][2];
    if (k0 > 0) {
      // the head of this tile was computed by the workgroup 8 below: wait for its accumulators, continue its chain.
      // The wait is bounded (a predecessor that never becomes resident must not hang the GPU) and LOUD: on expiry the
      // workgroup ORs the caller's sticky status word and its pass's error word and continues from NaN accumulators, so
      // the C-pass / plain-GEMM outputs of the tile are NaN (the Z-pass then takes its exact fallback for every element
      // of the tile: correct values, slowly).  Once any workgroup of any launch has failed, later waits give up at once.
      const bool handed_over = wait_for(p.flags + bid - 8);
      const f32x4* ip = reinterpret_cast<const f32x4*>(in_part) + (size_t)wave * (GEO::ACC / 4) * 64 + lane;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = ip[c * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * c + e] = handed_over ? v[e] : __builtin_nanf("");
          }
          ip += 4 * 64;
          asm volatile("" : "+v"(ip));      // one running pointer, not 32 precomputed ones
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
    }

    if constexpr (PROF) {
      if (threadIdx.x == 0) {
        const long long t = wall_clock64();
        prof_wait += t - prof_t0;
        prof_t0 = t;
        prof_steps += k1 - k0;
      }
    }
    // ---- main loop: stage ks + 1 (and, NST = 3, ks + 2) lands while stage ks is multiplied ----
    stage_in(0, 0);
    if constexpr (NST == 3) stage_in(1, (k0 + 1 < k1) ? 1 : 0);
    int st = 0;
    bf16x8 a[MI][3], b[2][3];
    if constexpr (STUDY == 3) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 3; ++q) b[ni][q] = *reinterpret_cast<const bf16x8*>(smem + NPA * kFrag + wn * (2 * kRB) + lane16 + ni * kRB + q * kFrag);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[mi][q] = *reinterpret_cast<const bf16x8*>(smem + wm * (MI * kRB) + lane16 + mi * kRB + q * kFrag);
    }
    for (int ks = k0; ks < k1; ++ks) {
      // This step's stage has landed (own loads: counted vmcnt -- loads retire in issue order, so with three stages the
      // LPS loads of the youngest fill stay in flight across the barrier; every wave's: the barrier), and the stage of
      // step ks - 1 is free again (every wave finished its fragment reads before it arrived here: lgkmcnt(0) at the last
      // MFMA round).  A RAW s_barrier: __syncthreads() carries a workgroup fence, for which hipcc drains vmcnt(0) -- a
      // direct-to-LDS load is a pending LDS write -- i.e. it would cut the prefetch distance back to one step.
      if constexpr (FULL) {
        if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if constexpr (FULL || STUDY == 1) __builtin_amdgcn_s_barrier();
      const unsigned char* sA = smem + st * STAGE + wm * (MI * kRB) + lane16;
      const unsigned char* sB = smem + st * STAGE + NPA * kFrag + wn * (2 * kRB) + lane16;
      // The order below is pinned with sched_barrier: left to itself hipcc's scheduler flips between an order that
      // chains dependent MFMAs two apart behind piecemeal LDS waits and a good one on unrelated source edits (+-17 % on
      // the whole kernel, measured).  Six partial products per block, smallest first (PA / PB); one ROUND = the same
      // partial product of all eight blocks (eight independent accumulators between dependent MFMAs), so the order of
      // the blocks inside a round is free: it is chosen so that every MFMA needs at most ONE fragment the previous ones
      // did not ((a0,b0) (a1,b0) (a1,b1) (a0,b1) (a2,b1) (a2,b0) (a3,b0) (a3,b1)), the fragment reads are issued in that
      // order (b0 a0 a1 b1 a2 a3; round 0 = plane 1 of both sides, round 1 planes (0, 2), round 2 planes (2, 0); rounds
      // 3-5 reuse what is resident), three to six at a time between the MFMAs, and each MFMA waits for exactly its own
      // operands.  After the barrier all eight waves read at once and the LDS delivers one 1 KiB fragment per 8 clocks:
      // waiting for a whole round (6 fragments x 8 waves) before the first MFMA idled the pipe for ~13 % of the step.
      constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
      // The reads are inline asm with hand-counted waits (guide 5.7): hipcc waits lgkmcnt(0) before the first MFMA once a
      // direct-to-LDS load is in flight.  lgkmcnt counts LDS returns in order: with I reads issued, fragment n (0-based)
      // is present once at most I - 1 - n are outstanding.  (Waits the compiler adds for its own LDS operations can only
      // be longer than needed.)
      const unsigned aA = (unsigned)(uintptr_t)sA, aB = (unsigned)(uintptr_t)sB;
#define X6_SB __builtin_amdgcn_sched_barrier(0)
#define X6_RDB(ni, r) \
  if constexpr (STUDY != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[ni][PB[r]]) : "v"(aB), "i"((ni) * kRB + PB[r] * kFrag))
#define X6_RDA(mi, r) \
  if constexpr (STUDY != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[mi][PA[r]]) : "v"(aA), "i"((mi) * kRB + PA[r] * kFrag))
#define X6_WAIT(n)                                                           \
  X6_SB;                                                                     \
  if constexpr (STUDY != 3) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
  X6_SB
#define X6_MM(mi, ni, q) acc[mi][ni] = TE_MFMA_BF16(a[mi][PA[q]], b[ni][PB[q]], acc[mi][ni])
#define X6_RD_HEAD(r) X6_SB; X6_RDB(0, r); X6_RDA(0, r); X6_RDA(1, r); X6_SB
#define X6_RD_TAIL(r) X6_SB; X6_RDB(1, r); X6_RDA(2, r); X6_RDA(3, r); X6_SB
      if constexpr (MI == 2) {
        // 64 x 64 wave tile: four blocks, 24 MFMAs and 12 fragment reads per step (three waves per SIMD cover the rest).
        // Round r reads (b0 a0 a1 b1) of its plane pair; with I reads issued fragment n is present at lgkmcnt <= I - 1 - n.
#define X6_RD4(r) X6_SB; X6_RDB(0, r); X6_RDA(0, r); X6_RDA(1, r); X6_RDB(1, r); X6_SB
        X6_RD4(0);
        X6_RD4(1);                                                   // 8 issued
        if constexpr (FULL) stage_in((st + NST - 1) % NST, (ks + NST - 1 < k1) ? 1 : 0);
        X6_WAIT(6); X6_MM(0, 0, 0);
        X6_WAIT(5); X6_MM(1, 0, 0);
        X6_WAIT(4); X6_MM(1, 1, 0); X6_MM(0, 1, 0);
        X6_RD4(2);                                                   // 12 issued
        X6_WAIT(6); X6_MM(0, 0, 1);
        X6_WAIT(5); X6_MM(1, 0, 1);
        X6_WAIT(4); X6_MM(1, 1, 1); X6_MM(0, 1, 1);
        X6_WAIT(2); X6_MM(0, 0, 2);
        X6_WAIT(1); X6_MM(1, 0, 2);
        X6_WAIT(0); X6_MM(1, 1, 2); X6_MM(0, 1, 2);
        X6_SB;
#pragma unroll
        for (int q6 = 3; q6 < 6; ++q6) {
          X6_MM(0, 0, q6); X6_MM(1, 0, q6); X6_MM(1, 1, q6); X6_MM(0, 1, q6);
          X6_SB;
        }
#undef X6_RD4
      } else {
      X6_RD_HEAD(0);
      X6_RD_TAIL(0);                                                 // 6 issued
      // the next step's stage is requested before the first MFMA (a direct-to-LDS load between MFMAs costs issue slots)
      if constexpr (FULL) stage_in((st + NST - 1) % NST, (ks + NST - 1 < k1) ? 1 : 0);
      X6_WAIT(4); X6_MM(0, 0, 0);
      X6_WAIT(3); X6_MM(1, 0, 0);
      X6_WAIT(2); X6_MM(1, 1, 0); X6_MM(0, 1, 0);
      X6_RD_HEAD(1);                                                 // 9 issued
      X6_WAIT(4); X6_MM(2, 1, 0); X6_MM(2, 0, 0);
      X6_WAIT(3); X6_MM(3, 0, 0); X6_MM(3, 1, 0);
      X6_RD_TAIL(1);                                                 // 12 issued
      X6_WAIT(4); X6_MM(0, 0, 1);
      X6_WAIT(3); X6_MM(1, 0, 1);
      X6_WAIT(2); X6_MM(1, 1, 1); X6_MM(0, 1, 1);
      X6_RD_HEAD(2);                                                 // 15 issued
      X6_WAIT(4); X6_MM(2, 1, 1); X6_MM(2, 0, 1);
      X6_WAIT(3); X6_MM(3, 0, 1); X6_MM(3, 1, 1);
      X6_RD_TAIL(2);                                                 // 18 issued
      X6_WAIT(4); X6_MM(0, 0, 2);
      X6_WAIT(3); X6_MM(1, 0, 2);
      X6_WAIT(2); X6_MM(1, 1, 2); X6_MM(0, 1, 2);
      X6_WAIT(1); X6_MM(2, 1, 2); X6_MM(2, 0, 2);
      X6_WAIT(0); X6_MM(3, 0, 2); X6_MM(3, 1, 2);
      X6_SB;
#pragma unroll
      for (int q6 = 3; q6 < 6; ++q6) {
        X6_MM(0, 0, q6); X6_MM(1, 0, q6); X6_MM(1, 1, q6); X6_MM(0, 1, q6);
        X6_MM(2, 1, q6); X6_MM(2, 0, q6); X6_MM(3, 0, q6); X6_MM(3, 1, q6);
        X6_SB;
      }
      }
#undef X6_SB
#undef X6_RDB
#undef X6_RDA
#undef X6_WAIT
#undef X6_MM
#undef X6_RD_HEAD
#undef X6_RD_TAIL
      st = (st + 1 == NST) ? 0 : st + 1;
    }

    if constexpr (PROF) {
      if (threadIdx.x == 0) {
        prof_t1 = wall_clock64();
        prof_loop += prof_t1 - prof_t0;
      }
    }
    // ---- accumulators to memory + flag (guide: plain stores -> vmcnt(0) -> barrier -> release -> flag) ----
    auto publish = [&](float* dst, unsigned* flag) __attribute__((always_inline)) {
      f32x4* op = reinterpret_cast<f32x4*>(dst) + (size_t)wave * (GEO::ACC / 4) * 64 + lane;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * c + e];
            op[c * 64] = v;
          }
          op += 4 * 64;
          asm volatile("" : "+v"(op));
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!p.drop_handover) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (PROF) prof_pub += wall_clock64() - prof_t1;
      }
    };
    if (k1 < nks) {            // a cut work item: the workgroup 8 above continues this chain
      publish(my_part, p.flags + bid);
      continue;
    }
    if constexpr (KSPLIT == 2) {
      // Two K segments per tile, each its own k-ordered chain, summed once: out = chain(segment 0) + chain(segment 1),
      // whatever the schedule.  Segment 0 leaves its accumulators in seg_part[tile]; segment 1 -- a later work item of the
      // same XCD, i.e. a HIGHER workgroup slot running at the same time -- waits for them here, at its END.
      float* const sp = p.seg_part + (size_t)tile * GEO::THREADS * GEO::ACC;
      unsigned* const sflag = p.seg_flags + tile;
      if (seg == 0) {
        publish(sp, sflag);
        __syncthreads();
        continue;
      }
      const bool got = wait_for(sflag);
      const f32x4* ip = reinterpret_cast<const f32x4*>(sp) + (size_t)wave * (GEO::ACC / 4) * 64 + lane;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = ip[c * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][ni][4 * c + e] = got ? v[e] + acc[mi][ni][4 * c + e] : __builtin_nanf("");
          }
          ip += 4 * 64;
          asm volatile("" : "+v"(ip));
          __builtin_amdgcn_sched_barrier(0);      // one block (16 registers) of segment 0 in flight at a time: the
        }                                         // accumulators are live here, a batch of all 32 loads would spill
    }

#ifdef TE_X6_STUDY
    if (p.opt & kX6OptSkipEpilogue) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) X6_KEEP(acc[mi][ni]);
      __syncthreads();
      continue;
    }
#endif
    if constexpr (!EPI) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) X6_KEEP(acc[mi][ni]);
      __syncthreads();
      if constexpr (PROF) {
        if (threadIdx.x == 0) {
          prof_epi += wall_clock64() - prof_t1;
          prof_nepi += 1;
        }
      }
      continue;
    }
    // ---- epilogue.  32x32 block (mi, ni): lane (tc = lane & 31, h = lane >> 5) holds activation row t and, for
    //      g = 0..3, weight rows 8 g + 4 h + (0..3) in acc[4 g .. 4 g + 3] ----
    const int tc = lane & 31, h = lane >> 5;
    // The loads of a block are issued as one batch BEFORE the previous block's stores (hipcc keeps loads behind stores
    // that may alias, and a store then costs a full round trip per `s_waitcnt vmcnt`): first light of this kernel spent
    // 110 us per tile in a load -> wait -> store -> wait chain.
    if constexpr (IS_Z) {
      const int nksS = p.out_f >> 4;
      int64_t tl[2];
      bool live[2], blk[2];
      float f[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int cb = tn * (GEO::TT / 32) + wn * 2 + ni;
        const int64_t t = (int64_t)cb * 32 + tc;
        blk[ni] = cb < p.ncb;
        live[ni] = t < p.T;
        tl[ni] = live[ni] ? t : p.T - 1;
        f[ni] = p.rs ? p.rs[(tl[ni] / p.rps) * p.rs_stride] : 1.0f;
      }
      // the wave's 128 bias values go through a private 512-B LDS slot: reading them back counts on lgkmcnt, so no wait
      // for a bias value drains the prefetched R / Y loads of the next block
      float* const bias_lds = reinterpret_cast<float*>(smem + LDS_MAIN) + wave * 128;
      if (lane < MI * 8) {
        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + (tm * (NWM * MI) + wm * MI) * 32 + lane * 4);
        *reinterpret_cast<f32x4*>(bias_lds + lane * 4) = bv;
      }
      bool staged_done = false;
      if constexpr (X6Lds<WM, MODE, NST>::STAGED) {
#ifdef TE_X6_STUDY      // (measurement builds keep the round-3 epilogue of these geometries behind TE_X6_OPT for same-box A/B runs)
        const bool staged = (p.opt & kX6OptStaged) != 0;
#else
        constexpr bool staged = true;
#endif
        if (staged) {
          staged_done = true;
          // Round 6 (profiles/r06_x6_epilogue_study.log (5)): a lane of the accumulator layout owns the 16-byte pieces g = 0..3 of ONE
          // 128-byte line of its row, so the four load instructions of a block touched the same 32 lines back to back and the in-order L1
          // stalled on the pending fills.  Here every line is requested by ONE instruction: a direct-to-LDS load covers 8 rows x 128 B
          // (8 lanes per line), chunk order XOR-permuted by the row so that the read-back in the accumulator layout -- lane (tc, h),
          // piece g = chunk 2g + h of row tc -- is bank-conflict free: [row][128 B] images, row r holds chunk c at slot c ^ ((r >> 1) & 7).
          constexpr int NL = (MODE == MODE_Z1) ? 4 : 8;                 // direct-to-LDS loads per block
          __builtin_amdgcn_s_barrier();                                 // every wave has read its last fragments: the stages are free
          unsigned char* const stg = smem + wave * 16384;               // two buffers of (R block, Y block)
          unsigned rda[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
            rda[g] = (unsigned)(uintptr_t)stg + tc * 128 + (((2 * g + h) ^ ((tc >> 1) & 7)) << 4);
          const unsigned bia = (unsigned)(uintptr_t)bias_lds + h * 16;
          const int r8 = lane >> 3;
          auto request = [&](int bi, int buf) __attribute__((always_inline)) {
            const int ni = bi / MI, mi = bi % MI;
            const int cb = tn * (GEO::TT / 32) + wn * 2 + ni;
            const int j0 = (tm * (NWM * MI) + wm * MI + mi) * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = 8 * i + r8;
              const int64_t t = min((int64_t)cb * 32 + row, p.T - 1);
              const int64_t off = t * p.out_f + j0 + (((lane & 7) ^ ((row >> 1) & 7)) << 2);
              glds16(reinterpret_cast<const unsigned char*>(p.R + off), stg + buf * 8192 + i * 1024);
              if constexpr (MODE != MODE_Z1)
                glds16(reinterpret_cast<const unsigned char*>(p.Y + off), stg + buf * 8192 + 4096 + i * 1024);
            }
          };
          // Two blocks in flight: block bi + 2 is requested into block bi's buffer as soon as block bi sits in registers.  The wait for
          // block bi is hand-counted: younger in the queue are the loads of block bi + 1 and the S stores of blocks bi - 2 and bi - 1
          // (six each, if the wave's row blocks exist: `all_blk`; else only the loads are counted, which waits longer, never shorter).
          const bool all_blk = blk[0] && blk[1];
#ifdef TE_X6_STUDY      // TE_X6_OPT bit 6: shader-clock stamps of the epilogue's phases, waves 0 and 4 (one SIMD) of workgroup 163
          long long* const stamp_out = reinterpret_cast<long long*>(p.flags + 2048) + 4608 + (wave >> 2) * 64;
          const bool stamping = (p.opt & 64) && bid == 163 && (wave & 3) == 0 && lane == 0;
          int stamp_i = 0;
#define X6_STAMP()                                                                              \
  if (p.opt & 64) {                                                                             \
    long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
    if (stamping) stamp_out[stamp_i] = t_;                                                      \
    ++stamp_i;                                                                                  \
  }
#else
#define X6_STAMP()
#endif
          X6_STAMP();
          request(0, 0);
          request(1, 1);
#pragma unroll
          for (int bi = 0; bi < 2 * MI; ++bi) {
            const int ni = bi / MI, mi = bi % MI, buf = bi & 1;
            const int cb = tn * (GEO::TT / 32) + wn * 2 + ni;
            const int j0 = (tm * (NWM * MI) + wm * MI + mi) * 32;
            constexpr int kStore = 6;
            const int yl = (bi + 1 < 2 * MI) ? NL : 0, ys = (bi >= 2 ? kStore : 0) + (bi >= 1 ? kStore : 0);
            X6_STAMP();
            if (all_blk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(yl + ys) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(yl) : "memory");
            X6_STAMP();
            f32x4 r4[4], y4[4], b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[g]) : "v"(rda[g]), "i"(buf * 8192));
              if constexpr (MODE != MODE_Z1)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y4[g]) : "v"(rda[g]), "i"(buf * 8192 + 4096));
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b4[g]) : "v"(bia), "i"((mi * 32 + 8 * g) * 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (bi + 2 < 2 * MI) request(bi + 2, buf);
            X6_STAMP();
            X6_PASS4(r4[0], r4[1], r4[2], r4[3]);
            if constexpr (MODE != MODE_Z1) X6_PASS4(y4[0], y4[1], y4[2], y4[3]);
            X6_PASS4(b4[0], b4[1], b4[2], b4[3]);
            // Element pairs on packed fp32 instructions (z, the division: sd2 -- an IEEE quotient without its range scaling, correctly
            // rounded inside the normal range).  Which elements cancel is kept as wave-level lane masks (the compare's own result,
            // OR-ed on the scalar unit); the per-lane bit mask the fallback wants is rebuilt only if one is set.
            unsigned w[4][3][2];
            unsigned long long any_cancel = 0;
            auto z_of = [&](int g, int c0, f32x2& z2, f32x2& a2) __attribute__((always_inline)) {
              a2 = f32x2{acc[mi][ni][4 * g + c0], acc[mi][ni][4 * g + c0 + 1]};
              if constexpr (MODE == MODE_Z)            // Z  = X+ W+^T + X- W-^T = ((Y - b) + |X||W|^T) / 2  (>= 0)
                z2 = f32x2{0.5f, 0.5f} * ((f32x2{y4[g][c0], y4[g][c0 + 1]} - f32x2{b4[g][c0], b4[g][c0 + 1]}) + a2);
              else if constexpr (MODE == MODE_ZI)      // Z' = X+ W-^T + X- W+^T = ((Y - b) - |X||W|^T) / 2  (<= 0)
                z2 = f32x2{0.5f, 0.5f} * ((f32x2{y4[g][c0], y4[g][c0 + 1]} - f32x2{b4[g][c0], b4[g][c0 + 1]}) - a2);
              else                                     // one-sided product: the accumulator is Z
                z2 = a2;
            };
            auto cancels = [&](float z, float a_abs) __attribute__((always_inline)) -> bool {
              if constexpr (MODE == MODE_Z) return !(z > kCancelTol * a_abs);
              else if constexpr (MODE == MODE_ZI) return !(-z > kCancelTol * a_abs);
              else return false;
            };
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float sv4[4];
#pragma unroll
              for (int c0 = 0; c0 < 4; c0 += 2) {
                f32x2 z2, a2;
                z_of(g, c0, z2, a2);
                f32x2 rr = {r4[g][c0], r4[g][c0 + 1]};
                if (p.rs) rr = rr * f32x2{f[ni], f[ni]};
                const f32x2 sv = sd2(rr, z2);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  const bool cancel = cancels(z2[e], a2[e]);
                  if constexpr (MODE != MODE_Z1) any_cancel |= __builtin_amdgcn_ballot_w64(cancel);
                  sv4[c0 + e] = (live[ni] && !cancel) ? sv[e] : 0.0f;
                }
              }
              unsigned lo[3], hi[3];
              split3_pk(sv4[0], sv4[1], lo);
              split3_pk(sv4[2], sv4[3], hi);
#pragma unroll
              for (int q = 0; q < 3; ++q) {
                w[g][q][0] = lo[q];
                w[g][q][1] = hi[q];
              }
            }
            unsigned bad = 0;                          // elements whose Z needs the cancellation fallback
            if (any_cancel != 0) {
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int c0 = 0; c0 < 4; c0 += 2) {
                  f32x2 z2, a2;
                  z_of(g, c0, z2, a2);
                  bad |= (cancels(z2[0], a2[0]) ? 1u : 0u) << (4 * g + c0);
                  bad |= (cancels(z2[1], a2[1]) ? 1u : 0u) << (4 * g + c0 + 1);
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
              for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int d = 0; d < 2; ++d) swap_halves(w[g][q][d], w[g + 2][q][d]);
#ifdef TE_X6_STUDY
            if (p.opt & 64) {
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 3; ++q) asm volatile("" : "+v"(w[g][q][0]), "+v"(w[g][q][1]));      // (the arithmetic ends before the stamp)
            }
#endif
            X6_STAMP();
            if (blk[ni]) {
              unsigned char* Srow = p.S + (int64_t)cb * nksS * kRB + tc * 16;
              unsigned char* sp = Srow + (int64_t)((j0 >> 4) + h) * kRB;
#pragma unroll
              for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                  u32x4 v = {w[c][q][0], w[c][q][1], w[c + 2][q][0], w[c + 2][q][1]};
                  *reinterpret_cast<u32x4*>(sp + q * kFrag + c * 512) = v;
                }
              if (!live[ni]) bad = 0;
              if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const float* Rrow = p.R + tl[ni] * p.out_f;
#pragma clang loop unroll(disable)
                for (int e = 0; e < 16; ++e) {
                  if ((bad >> e) & 1u) {
                    const int jj = j0 + 8 * (e >> 2) + 4 * h + (e & 3);
                    const float z = (MODE == MODE_ZI) ? exact_zi(p.X + tl[ni] * p.in_f, p.W + (int64_t)jj * p.in_f, p.in_f)
                                                      : exact_z(p.X + tl[ni] * p.in_f, p.W + (int64_t)jj * p.in_f, p.in_f);
                    float rr = Rrow[jj];
                    if (p.rs) rr = rr * f[ni];
                    unsigned pl[3];
                    split3(te_sd(rr, z), pl);
                    unsigned short* d = reinterpret_cast<unsigned short*>(Srow + (int64_t)(jj >> 4) * kRB + ((jj >> 3) & 1) * 512) + (jj & 7);
                    d[0] = (unsigned short)pl[0];
                    d[kFrag / 2] = (unsigned short)pl[1];
                    d[kFrag] = (unsigned short)pl[2];
                  }
                }
              }
            }
            X6_STAMP();
          }
#undef X6_STAMP
        }
      }
      if (!staged_done) {
      // (128 x 128 geometry: three waves per SIMD cover the load latency, and 168 VGPRs do not hold a second buffer)
      constexpr int NBUF = (MI == 4) ? 2 : 1;
      f32x4 r4[NBUF][4], y4[NBUF][4];
      auto load_block = [&](int bi, int buf) __attribute__((always_inline)) {
        const int ni = bi / MI, mi = bi % MI;
        const int j0 = (tm * (NWM * MI) + wm * MI + mi) * 32;
        const float* Rrow = p.R + tl[ni] * p.out_f + j0 + 4 * h;
#ifdef TE_X6_STUDY      // TE_X6_OPT bit 4: the epilogue without its R / Y loads (garbage results: what the loads cost)
        if (p.opt & 16) {
#pragma unroll
          for (int g = 0; g < 4; ++g) r4[buf][g] = f32x4{1.0f, 2.0f, 3.0f, 4.0f}, y4[buf][g] = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
          return;
        }
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g) r4[buf][g] = *reinterpret_cast<const f32x4*>(Rrow + 8 * g);
        if constexpr (MODE != MODE_Z1) {
          const float* Yrow = p.Y + tl[ni] * p.out_f + j0 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) y4[buf][g] = *reinterpret_cast<const f32x4*>(Yrow + 8 * g);
        }
      };
      if constexpr (NBUF == 2) load_block(0, 0);
#pragma unroll
      for (int bi = 0; bi < 2 * MI; ++bi) {
        const int ni = bi / MI, mi = bi % MI, buf = bi % NBUF;
        const int cb = tn * (GEO::TT / 32) + wn * 2 + ni;
        const int j0 = (tm * (NWM * MI) + wm * MI + mi) * 32;
        if constexpr (NBUF == 2) {
          if (bi + 1 < 2 * MI) load_block(bi + 1, buf ^ 1);      // in flight during this block's arithmetic
        } else {
          load_block(bi, 0);
        }
        unsigned w[4][3][2];                       // [g][plane][dword]: four bf16 of one plane = weight rows 8g+4h+0..3
        unsigned bad = 0;                          // elements whose Z needs the cancellation fallback
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + mi * 32 + 8 * g + 4 * h);
          float sv4[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float a_abs = acc[mi][ni][4 * g + c];
            float z;
            bool cancel;
            if constexpr (MODE == MODE_Z) {              // Z  = X+ W+^T + X- W-^T = ((Y - b) + |X||W|^T) / 2  (>= 0)
              z = 0.5f * ((y4[buf][g][c] - b4[c]) + a_abs);
              cancel = !(z > kCancelTol * a_abs);
            } else if constexpr (MODE == MODE_ZI) {      // Z' = X+ W-^T + X- W+^T = ((Y - b) - |X||W|^T) / 2  (<= 0)
              z = 0.5f * ((y4[buf][g][c] - b4[c]) - a_abs);
              cancel = !(-z > kCancelTol * a_abs);
            } else {                                     // one-sided product: the accumulator is Z
              z = a_abs;
              cancel = false;
            }
            bad |= (cancel ? 1u : 0u) << (4 * g + c);
            float rr = r4[buf][g][c];
            if (p.rs) rr = rr * f[ni];
            float sv = te_sd(rr, z);
            asm volatile("" : "+v"(sv));            // evaluated for every lane: a select below, not a branch around the division
            sv4[c] = (live[ni] && !cancel) ? sv : 0.0f;
          }
          unsigned lo[3], hi[3];
          split3_pk(sv4[0], sv4[1], lo);
          split3_pk(sv4[2], sv4[3], hi);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            w[g][q][0] = lo[q];
            w[g][q][1] = hi[q];
          }
        }
        // lanes h = 0 end up with the 16-B pieces g = 0, 1 (8 consecutive features each), lanes h = 1 with g = 2, 3
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int d = 0; d < 2; ++d) swap_halves(w[g][q][d], w[g + 2][q][d]);
#ifdef TE_X6_STUDY      // TE_X6_OPT bit 5: the epilogue without its S stores (garbage results: what the stores cost)
        if (p.opt & 32) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(w[g][q][0]), "v"(w[g][q][1]));
          continue;
        }
#endif
        if (blk[ni]) {
          unsigned char* Srow = p.S + (int64_t)cb * nksS * kRB + tc * 16;
          unsigned char* sp = Srow + (int64_t)((j0 >> 4) + h) * kRB;
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {           // c = k-half of the K16 step (j0 / 16 + h)
              // h = 0: piece g = c is {own w[c], partner's w[c]} = (w[c], w[c + 2]) after the swap
              // h = 1: piece g = 2 + c is {partner's w[2 + c], own w[2 + c]} = (w[c], w[c + 2]) after the swap
              u32x4 v = {w[c][q][0], w[c][q][1], w[c + 2][q][0], w[c + 2][q][1]};
              *reinterpret_cast<u32x4*>(sp + q * kFrag + c * 512) = v;
            }
          // rare: (Y - b) and |X||W|^T cancel (nearly every product of the element is negative): the reference's plain
          // positive-part sum, k-ordered, written over the element's three plane values
          if (!live[ni]) bad = 0;
          if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the 16-B pieces above are in memory first
            const float* Rrow = p.R + tl[ni] * p.out_f;
#pragma clang loop unroll(disable)
            for (int e = 0; e < 16; ++e) {
              if ((bad >> e) & 1u) {
                const int jj = j0 + 8 * (e >> 2) + 4 * h + (e & 3);
                const float z = (MODE == MODE_ZI) ? exact_zi(p.X + tl[ni] * p.in_f, p.W + (int64_t)jj * p.in_f, p.in_f)
                                                  : exact_z(p.X + tl[ni] * p.in_f, p.W + (int64_t)jj * p.in_f, p.in_f);
                float rr = Rrow[jj];
                if (p.rs) rr = rr * f[ni];
                unsigned pl[3];
                split3(te_sd(rr, z), pl);
                unsigned short* d = reinterpret_cast<unsigned short*>(Srow + (int64_t)(jj >> 4) * kRB + ((jj >> 3) & 1) * 512) + (jj & 7);
                d[0] = (unsigned short)pl[0];
                d[kFrag / 2] = (unsigned short)pl[1];
                d[kFrag] = (unsigned short)pl[2];
              }
            }
          }
        }
      }
      }      // (!staged_done)
    } else if constexpr (MODE == MODE_G) {
      // plain product: out[t][m] = acc + bias[m] (fp32 row-major [T, M]); the wave's 128 bias values through its LDS slot
      float* const bias_lds = reinterpret_cast<float*>(smem + LDS_MAIN) + wave * 128;
      if (lane < MI * 8) {
        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + (tm * (NWM * MI) + wm * MI) * 32 + lane * 4);
        *reinterpret_cast<f32x4*>(bias_lds + lane * 4) = bv;
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t t = ((int64_t)tn * (GEO::TT / 32) + wn * 2 + ni) * 32 + tc;
        float* orow = p.out + (t < p.T ? t : 0) * p.out_f + (tm * (NWM * MI) + wm * MI) * 32 + 4 * h;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + mi * 32 + 8 * g + 4 * h);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = acc[mi][ni][4 * g + c] + b4[c];
            if (t < p.T) *reinterpret_cast<f32x4*>(orow + mi * 32 + 8 * g) = o;
          }
      }
    } else if constexpr (MODE == MODE_X) {
      // masked product (variant lrp's C-pass, one sign at a time: layers_lrp.py:201-202): out[t][m] (+)= scale * (X+-[t][m] * acc)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t t = ((int64_t)tn * (GEO::TT / 32) + wn * 2 + ni) * 32 + tc;
        const int64_t off = (t < p.T ? t : 0) * p.out_f + (tm * (NWM * MI) + wm * MI) * 32 + 4 * h;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 xv[4], pv[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            xv[g] = *reinterpret_cast<const f32x4*>(p.X + off + mi * 32 + 8 * g);
            pv[g] = p.accum ? *reinterpret_cast<const f32x4*>(p.out + off + mi * 32 + 8 * g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float xm = (p.x_sign > 0) ? fmaxf(xv[g][c], 0.0f) : fminf(xv[g][c], 0.0f);
              const float v = p.scale * (xm * acc[mi][ni][4 * g + c]);
              o[c] = p.accum ? pv[g][c] + v : v;
            }
            if (t < p.T) *reinterpret_cast<f32x4*>(p.out + off + mi * 32 + 8 * g) = o;
          }
        }
      }
    } else {
      f32x4 x4[2][MI / 2][4];
      int64_t tt[2];
      bool live[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t t = ((int64_t)tn * (GEO::TT / 32) + wn * 2 + ni) * 32 + tc;
        live[ni] = t < p.T;
        tt[ni] = live[ni] ? t : p.T - 1;
#pragma unroll
        for (int il = 0; il < MI / 2; ++il) {
          const float* xr = p.X + tt[ni] * p.in_f + (tm * (NWM * MI / 2) + wm * (MI / 2) + il) * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) x4[ni][il][g] = *reinterpret_cast<const f32x4*>(xr + 8 * g);
        }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int il = 0; il < MI / 2; ++il) {
          float* orow = p.out + tt[ni] * p.in_f + (tm * (NWM * MI / 2) + wm * (MI / 2) + il) * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float xv = x4[ni][il][g][c];
              const float xp = fmaxf(xv, 0.0f), xn = fminf(xv, 0.0f);
              if constexpr (MODE == MODE_C)     // alpha * (X+ . (S W+) + X- . (S W-))                  layers_ours.py:222-228
                o[c] = p.scale * (xp * acc[2 * il][ni][4 * g + c] + xn * acc[2 * il + 1][ni][4 * g + c]);
              else                              // out - beta * (X+ . (S' W-) + X- . (S' W+)): scale = -beta
                o[c] = p.scale * (xp * acc[2 * il + 1][ni][4 * g + c] + xn * acc[2 * il][ni][4 * g + c]);
            }
            if constexpr (MODE == MODE_CI) {
              const f32x4 prev = *reinterpret_cast<const f32x4*>(orow + 8 * g);
#pragma unroll
              for (int c = 0; c < 4; ++c) o[c] = prev[c] + o[c];
            }
            if (live[ni]) *reinterpret_cast<f32x4*>(orow + 8 * g) = o;
          }
        }
    }
    __syncthreads();       // every wave is done with the stages before the next fragment's first loads land in them
    if constexpr (PROF) {
      if (threadIdx.x == 0) {
        prof_epi += wall_clock64() - prof_t1;
        prof_nepi += 1;
      }
    }
  }
  if constexpr (PROF) {
    if (threadIdx.x == 0) {
      long long* o = reinterpret_cast<long long*>(p.flags + 2048) + (size_t)bid * 8;
      o[0] = prof_loop, o[1] = prof_epi, o[2] = prof_pub, o[3] = prof_wait, o[4] = prof_steps, o[5] = prof_nepi;
      o[6] = wall_clock64() - prof_start;
      o[7] = prof_start;
    }
  }
}

inline size_t planes_bytes(int64_t rows, int64_t K) {
  return (size_t)te_ceil_div(rows, 32) * 32 * (size_t)K * 6;
}
// Fixed K split -- a STUDY, off by default (the flag TE_X6_KSPLIT turns it on; it changes which chains an output is summed
// from, i.e. the bits: a caller sets it for every launch of a process or for none).  Products with a long K and few weight rows --
// out_f = 768 against K = 2304 / 3072: fc2's forward, the input gradients of qkv and fc1, fc2's Z-pass -- are 150 tiles
// of 256 x 256 for 256 CUs, and a tile is a SEQUENTIAL chain of K / 16 steps however it is scheduled.  Two chains per output
// (the same two for every tile geometry, batch size and schedule: a property of the layer's shape only, so results stay
// bitwise batch- and geometry-invariant) halve that floor and double the work items.  Measured:
// isolated launches gain 10-17 % over 128 x 256 tiles, but only 5 % over the 128 x 128 geometry that shipped with it, and
// IN THE STEP nothing (ViT-B/16: 893 vs 896 maps/s, same box, A B A B) or less than nothing (BERT-512: 286 vs 292
// sequences/s: at T = 16 384 the un-split 128 x 128 launch is exactly one tile per workgroup slot) -- the CUs a narrow
// launch leaves idle are used by the other stream of the step anyway.  Not shipped.
inline int kseg_shape(int64_t K, int64_t rows_w) { return (K >= 1536 && rows_w <= 768 && (K / 16) % 2 == 0) ? 2 : 1; }
#ifdef TE_X6_STUDY
inline int kseg_rule(int64_t K, int64_t rows_w, int flags) { return (flags & TE_X6_KSPLIT) ? kseg_shape(K, rows_w) : 1; }
#else
inline int kseg_rule(int64_t, int64_t, int) { return 1; }
#endif
// study-only schedules (TE_X6_STAGES_3, TE_X6_KSPLIT) are not compiled into the shipped library: a caller that asks for one
// gets TE_ERR_UNSUPPORTED instead of a heavily spilling kernel
inline bool study_flags_refused(int flags) {
#ifdef TE_X6_STUDY
  (void)flags;
  return false;
#else
  return (flags & (TE_X6_STAGES_3 | TE_X6_KSPLIT)) != 0;
#endif
}
inline size_t seg_part_bytes(int64_t T, int64_t K, int64_t rows_w) {       // accumulators of segment 0, any geometry
  return kseg_shape(K, rows_w) == 2 ? te_align_up((size_t)te_ceil_div(T, 256) * 256 * (size_t)rows_w * 4, 256) : 0;
}
inline size_t seg_flag_words(int64_t T, int64_t K, int64_t rows_w) {       // one per tile of the smallest geometry, x 1024
  return kseg_shape(K, rows_w) == 2 ? te_align_up((size_t)te_ceil_div(T, 128) * (size_t)(rows_w / 128), 1024) : 0;
}
inline size_t seg_region_bytes(int64_t T, int64_t K, int64_t rows_w) {
  return seg_part_bytes(T, K, rows_w) + seg_flag_words(T, K, rows_w) * 4;
}
constexpr size_t kPartialBytes = (size_t)512 * 256 * 128 * 4;       // grid x threads x accumulators: 64 MiB covers every geometry
constexpr double kWholeTileSlack = 1.15;                          // whole tiles if ceil(r) <= 1.15 r (launch_x6)
constexpr size_t kFlagBytes = 65536;                                 // 512 flags + the error word (+ study time stamps), per pass

inline int pick_wm(int64_t in_f, int64_t out_f) {
  if (out_f % 256 == 0 && in_f % 128 == 0) return 2;
  if (out_f % 128 == 0 && in_f % 64 == 0) return 1;
  return 0;
}

// wall_clock64 rate of the current device (constant-frequency counter), per device, cached
inline long long spin_ticks_for_current_device() {
  static long long cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cached[dev] == 0) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    cached[dev] = (long long)khz * kSpinMillis;
  }
  return cached[dev];
}

template <int WM, int MODE, int STUDY = 0, int NST = 2, int KSPLIT = 1>
int launch_x6(const X6Params& p, hipStream_t stream) {
  using GEO = X6Geo<WM>;
  constexpr int lds = X6Lds<WM, MODE, NST>::TOTAL;
  static_assert(lds * (WM == 0 ? 3 : WM == 1 ? 2 : 1) <= 160 * 1024, "LDS of the workgroups of one CU");
  static_assert((size_t)8 * GEO::MAX_SPX * GEO::THREADS * GEO::ACC * 4 <= kPartialBytes && 8 * GEO::MAX_SPX < kErrWord, "workspace");
  auto kern = x6_kernel<WM, MODE, STUDY, NST, KSPLIT>;
  // an attribute of the code object ON THE CURRENT DEVICE: set per launch (idempotent, no data-path state)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  X6Params q = p;
  q.ksplit = KSPLIT;
  q.ntm = p.rows_w / GEO::TM;
  q.ntn = (int)te_ceil_div(p.T, GEO::TT);
  const int64_t tiles = (int64_t)q.ntm * q.ntn * q.ksplit;      // work items
  int max_spx = p.small_grid ? 2 : GEO::MAX_SPX;
#ifdef TE_X6_STUDY      // TE_X6_CUS_PER_XCD=<n <= 32>: leave 32 - n CUs of every XCD to the kernels of other streams (overlapped-step study)
  if (const char* e_cu = getenv("TE_X6_CUS_PER_XCD")) {
    const int n = atoi(e_cu);
    if (n >= 1 && n <= 32 && !p.small_grid) max_spx = GEO::MAX_SPX * n / 32;
  }
#endif
  const int spx = (int)std::min<int64_t>(max_spx, std::max<int64_t>(1, te_ceil_div(tiles, 8)));
  // Stream-K or whole tiles?  With equal (tile, k) ranges the workgroups of an XCD sit at different k offsets of their
  // tiles and nothing one of them fetches is still in the 4 MB L2 when its neighbour needs it; cut at tile boundaries
  // they run a round in k lock-step on shared operand panels and a K16 step takes 1.9 instead of 2.1-2.6 us (measured,
  // profiles/r03_x6_whole_tiles.log) -- which pays as long as the last round is nearly full.  r = tiles per workgroup;
  // whole tiles cost ceil(r) rounds.  Either way every output is the same k-ordered chain: results do not change.
  const double r = (double)tiles / (8.0 * spx);
  // TE_X6_WHOLE_TILES (the caller runs concurrent streams): whole tiles always -- the last round's idle CUs are not lost, the
  // other streams' kernels use them, while stream-K keeps ALL CUs for the whole launch at the slower skewed step
  if (!q.whole_tiles_forced)
    q.whole_tiles = ((p.prefer_whole || std::ceil(r) <= kWholeTileSlack * r) && !p.small_grid) ? 1 : 0;
  if (!q.status) q.status = q.flags + kErrWord;
  q.spin_ticks = spin_ticks_for_current_device();
  q.opt = kX6OptDefault;
#ifdef TE_X6_STUDY      // measurement builds: TE_X6_OPT=<bits> overrides the schedule options, read per launch (benchmarks/x6_variants.py
  if (const char* e_opt = getenv("TE_X6_OPT")) q.opt = atoi(e_opt);      // interleaves the variants in one process)
#endif
  kern<<<dim3(8 * spx), dim3(GEO::THREADS), lds, stream>>>(q);
  return TE_OK;
}

// Two LDS stages everywhere; TE_X6_STAGES_3 (measurement) runs the 256 x 256 geometry with three (prefetch distance 2 behind
// a raw barrier, 148 KiB).  Measured on the MI355X, ViT-B/16 batch 64 (profiles/r04_x6_variants.log): Z-pass and plain
// products +-0.5 %, C-pass 7 % SLOWER -- a fill that misses the L2 is not what the loop waits for; the third stage's
// traffic in flight costs more than its latency cover buys.
template <int MODE>
int launch_x6_mode(int wm, bool three_stages, const X6Params& p, hipStream_t stream) {
#ifdef TE_X6_STUDY      // NST = 3 and KSPLIT = 2 are measured studies: compiled into study builds only
  if constexpr (MODE != MODE_C && MODE != MODE_CI) {
    if (p.seg_part && p.seg_flags && kseg_shape((int64_t)p.nks * 16, p.rows_w) == 2) {
      if (wm == 2) return launch_x6<2, MODE, 0, 2, 2>(p, stream);
      if (wm == 1) return launch_x6<1, MODE, 0, 2, 2>(p, stream);
      return launch_x6<0, MODE, 0, 2, 2>(p, stream);
    }
  }
  if (wm == 2 && three_stages) return launch_x6<2, MODE, 0, 3>(p, stream);
#else
  if (three_stages || p.seg_part || p.seg_flags) return TE_ERR_UNSUPPORTED;      // (the entry points reject the flags first)
#endif
  if (wm == 2) return launch_x6<2, MODE, 0, 2>(p, stream);
  if (wm == 1) return launch_x6<1, MODE, 0, 2>(p, stream);
  return launch_x6<0, MODE, 0, 2>(p, stream);
}

// Tile geometry of a launch (2 / 1 / 0 = 256 x 256 / 128 x 256 / 128 x 128; the result does not depend on it, bit for bit).
// wm_max = the largest geometry the feature counts divide into; rows_w = weight-side rows; pin = TE_X6_TILE_* or 0.
// small_ok: the launch may use the 128 x 128 geometry (Z-pass, plain products; the C-pass never gains from it).
// Measured per launch on the MI355X (profiles/r04_x6_geometry.log; ViT-B/16 batch 64, BERT-512 batch 32, ViT-L/16-384 batch
// 32): out_f = 768 at T = 12 608 / 16 384 (300 / 384 tiles of 128 x 256) runs 8-15 % faster on 128 x 128 tiles (Z-pass
// 143 -> 121 us and 431 -> 368 us, products 119 -> 106, 401 -> 360, 313 -> 280, 408 -> 368 us); out_f = 1024 at T = 18 464
// (584 tiles) is 10 % faster on 128 x 256.
inline int choose_geo(int wm_max, int64_t T, int64_t rows_w, int pin, int min_tiles_256 = 256, bool small_ok = false,
                      int ksplit = 1) {
  if (pin == TE_X6_TILE_128) return 1;
  if (pin == TE_X6_TILE_256) return wm_max;
  if (pin == TE_X6_TILE_128x128) return 0;
  const int64_t t256 = te_ceil_div(T, 256);
  // 256-row tiles (one 512-thread workgroup per CU, every operand byte staged once for eight waves) where that gives
  // every CU a tile ...
  if (wm_max == 2 && t256 * (rows_w / 256) * ksplit >= min_tiles_256) return 2;       // (work items: tiles x K segments)
  // ... 128 x 256 (two workgroups per CU) while there are at least ~1.75 tiles per CU, else 128 x 128 (three per CU)
  static const bool no_small = [] { const char* e = getenv("TE_X6_SMALL_TILES"); return e && atoi(e) == 0; }();      // (A/B knob)
  if (small_ok && !no_small && t256 * (rows_w / 128) * ksplit < 448) return 0;
  return 1;
}

}  // namespace

// ----------------------------------------------------------------------------------------------------------------------
extern "C" int te_linear_relprop_x6_supported(int64_t T, int64_t in_f, int64_t out_f) {
  return (T >= 1 && in_f >= 128 && out_f >= 128 && pick_wm(in_f, out_f) != 0 && in_f <= (1 << 20) && out_f <= (1 << 20) &&
          T <= (int64_t)1 << 26) ? 1 : 0;
}

extern "C" size_t te_linear_x6_weight_planes_bytes(int64_t in_f, int64_t out_f) {
  if (!te_linear_relprop_x6_supported(1, in_f, out_f)) return 0;
  // P3 of |W| (rows = out_f, K = in_f) followed by P6 of W+^T / W-^T (rows = in_f, K = out_f)
  return te_align_up(planes_bytes(out_f, in_f), 256) + te_align_up(2 * planes_bytes(in_f, out_f), 256);
}

extern "C" int te_linear_x6_prepare_weights_f32(const float* W, int64_t in_f, int64_t out_f, void* planes,
                                                size_t planes_bytes_, te_stream_t stream_) {
  if (!W || !planes) return TE_ERR_INVALID_ARG;
  if (!te_linear_relprop_x6_supported(1, in_f, out_f) || !te_aligned16(W)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < te_linear_x6_weight_planes_bytes(in_f, out_f) || !te_aligned16(planes)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* pz = (unsigned char*)planes;
  unsigned char* pc = pz + te_align_up(planes_bytes(out_f, in_f), 256);
  split_kernel<OP_ABS, false><<<dim3((unsigned)(out_f / 32), (unsigned)te_ceil_div(in_f / 16, 8)), dim3(256), 0, stream>>>(
      W, pz, out_f, in_f, 3, 0);
  const dim3 gc((unsigned)(in_f / 32), (unsigned)te_ceil_div(out_f / 16, 8));
  split_kernel<OP_POS, true><<<gc, dim3(256), 0, stream>>>(W, pc, in_f, out_f, 6, 0);
  split_kernel<OP_NEG, true><<<gc, dim3(256), 0, stream>>>(W, pc, in_f, out_f, 6, 1);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" size_t te_linear_x6_planes_bytes(int64_t rows, int64_t K) {
  return (rows < 1 || K < 16 || K % 16) ? 0 : planes_bytes(rows, K);
}

extern "C" int te_linear_x6_split_abs_f32(const float* X, int64_t rows, int64_t K, void* planes, size_t planes_bytes_,
                                          te_stream_t stream_) {
  if (!X || !planes || rows < 1) return TE_ERR_INVALID_ARG;
  if (K < 16 || K % 16 || !te_aligned16(X) || rows > ((int64_t)1 << 26)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < planes_bytes(rows, K) || !te_aligned16(planes)) return TE_ERR_WORKSPACE;
  split_kernel<OP_ABS, false><<<dim3((unsigned)te_ceil_div(rows, 32), (unsigned)te_ceil_div(K / 16, 8)), dim3(256), 0,
                                (hipStream_t)stream_>>>(X, (unsigned char*)planes, rows, K, 3, 0);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// ---- plain GEMM on the same machinery (the forward / input-gradient products of a Linear layer) -------
// out [T, M] = X [T, K] . Wp^T + bias [M], Wp = signed P3 planes of an [M, K] matrix (te_linear_x6_split_matrix_f32: of W
// itself for the forward product, of W^T for the input gradient d_x = d_y W).
extern "C" int te_gemm_x6_supported(int64_t T, int64_t K, int64_t M) {
  return (T >= 1 && K >= 128 && M >= 128 && K % 16 == 0 && M % 128 == 0 && K <= (1 << 20) && M <= (1 << 20) &&
          T <= (int64_t)1 << 26) ? 1 : 0;
}

extern "C" int te_linear_x6_split_matrix_f32(const float* A, int64_t rows, int64_t K, int transposed, void* planes,
                                             size_t planes_bytes_, te_stream_t stream_) {
  if (!A || !planes || rows < 1) return TE_ERR_INVALID_ARG;
  if (K < 16 || K % 16 || !te_aligned16(A) || rows > ((int64_t)1 << 26)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < planes_bytes(rows, K) || !te_aligned16(planes)) return TE_ERR_WORKSPACE;
  const dim3 grid((unsigned)te_ceil_div(rows, 32), (unsigned)te_ceil_div(K / 16, 8));
  if (transposed)       // A is [K, rows] in memory
    split_kernel<OP_ID, true><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(A, (unsigned char*)planes, rows, K, 3, 0);
  else
    split_kernel<OP_ID, false><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(A, (unsigned char*)planes, rows, K, 3, 0);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// planes of A [rows, K] AND of |A| from one pass over A (the forward product of a Linear layer wants the first, its
// relprop rule the second: te_linear_relprop_x6_f32's x_planes)
extern "C" int te_linear_x6_split_dual_f32(const float* A, int64_t rows, int64_t K, void* planes, void* planes_abs,
                                           size_t planes_bytes_, te_stream_t stream_) {
  if (!A || !planes || !planes_abs || rows < 1) return TE_ERR_INVALID_ARG;
  if (K < 16 || K % 16 || !te_aligned16(A) || rows > ((int64_t)1 << 26)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < planes_bytes(rows, K) || !te_aligned16(planes) || !te_aligned16(planes_abs)) return TE_ERR_WORKSPACE;
  const dim3 grid((unsigned)te_ceil_div(rows, 32), (unsigned)te_ceil_div(K / 16, 8));
  split_kernel<OP_ID, false><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(A, (unsigned char*)planes, rows, K, 3, 0,
                                                                          (unsigned char*)planes_abs);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// GELU producers that write operand planes (gelu_split_kernel above); planes / planes_abs: te_linear_x6_planes_bytes(rows, K)
extern "C" int te_gelu_backward_x6_planes_f32(const float* dy, const float* x, int64_t rows, int64_t K, void* planes,
                                              size_t planes_bytes_, te_stream_t stream_) {
  if (!dy || !x || !planes || rows < 1) return TE_ERR_INVALID_ARG;
  if (K < 16 || K % 16 || !te_aligned16(dy) || !te_aligned16(x) || rows > ((int64_t)1 << 26)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < planes_bytes(rows, K) || !te_aligned16(planes)) return TE_ERR_WORKSPACE;
  const dim3 grid((unsigned)te_ceil_div(rows, 32), (unsigned)te_ceil_div(K / 16, 8));
#ifdef TE_STUDY
  if (gelu_split_direct()) {
    gelu_split_kernel<SRC_GELU_BWD><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(dy, x, nullptr, (unsigned char*)planes,
                                                                                 nullptr, rows, K);
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
#endif
  gelu_split_lds_kernel<SRC_GELU_BWD><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(dy, x, nullptr, (unsigned char*)planes,
                                                                                   nullptr, rows, K);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" int te_gelu_forward_x6_planes_f32(const float* x, float* y, int64_t rows, int64_t K, void* planes, void* planes_abs,
                                             size_t planes_bytes_, te_stream_t stream_) {
  if (!x || !y || !planes || !planes_abs || rows < 1) return TE_ERR_INVALID_ARG;
  if (K < 16 || K % 16 || !te_aligned16(x) || !te_aligned16(y) || rows > ((int64_t)1 << 26)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < planes_bytes(rows, K) || !te_aligned16(planes) || !te_aligned16(planes_abs)) return TE_ERR_WORKSPACE;
  const dim3 grid((unsigned)te_ceil_div(rows, 32), (unsigned)te_ceil_div(K / 16, 8));
#ifdef TE_STUDY
  if (gelu_split_direct()) {
    gelu_split_kernel<SRC_GELU_FWD><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(nullptr, x, y, (unsigned char*)planes,
                                                                                 (unsigned char*)planes_abs, rows, K);
    TE_RETURN_IF_LAUNCH_FAILED();
    return TE_OK;
  }
#endif
  gelu_split_lds_kernel<SRC_GELU_FWD><<<grid, dim3(256), 0, (hipStream_t)stream_>>>(nullptr, x, y, (unsigned char*)planes,
                                                                                   (unsigned char*)planes_abs, rows, K);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

extern "C" size_t te_gemm_x6_workspace_bytes(int64_t T, int64_t K, int64_t M) {
  if (!te_gemm_x6_supported(T, K, M)) return 0;
  return te_align_up(planes_bytes(T, K), 256) + kPartialBytes + kFlagBytes + seg_region_bytes(T, K, M);
}

extern "C" int te_gemm_x6_f32(const float* X, const void* x_planes, const void* w_planes, const float* bias, float* out,
                              int64_t T, int64_t K, int64_t M, int flags, unsigned* status, void* ws, size_t ws_bytes,
                              te_stream_t stream_) {
  if ((!X && !x_planes) || !w_planes || !out) return TE_ERR_INVALID_ARG;
  if ((flags & ~(3 | TE_X6_STAGES_3 | TE_X6_KSPLIT | TE_X6_WHOLE_TILES | TE_X6_TEST_DROP_HANDOVER | TE_X6_TEST_SMALL_GRID)) != 0) return TE_ERR_INVALID_ARG;
  if (study_flags_refused(flags)) return TE_ERR_UNSUPPORTED;
  if (!te_gemm_x6_supported(T, K, M)) return TE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < te_gemm_x6_workspace_bytes(T, K, M) || !te_aligned16(ws)) return TE_ERR_WORKSPACE;
  if ((X && !te_aligned16(X)) || !te_aligned16(out) || !te_aligned16(w_planes) || (bias && !te_aligned16(bias)) ||
      (x_planes && !te_aligned16(x_planes)))
    return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* q = (unsigned char*)ws;
  unsigned char* Xs = q;
  q += te_align_up(planes_bytes(T, K), 256);
  float* partial = (float*)q;
  q += kPartialBytes;
  unsigned* flag_words = (unsigned*)q;
  q += kFlagBytes;
  zero_words_kernel<<<dim3(kFlagBytes / 16 / 256), dim3(256), 0, stream>>>(reinterpret_cast<u32x4*>(flag_words));
  float* seg_part = nullptr;
  unsigned* seg_flags = nullptr;
  if (kseg_rule(K, M, flags) == 2) {
    seg_part = (float*)q;
    seg_flags = (unsigned*)(q + seg_part_bytes(T, K, M));
    zero_words_kernel<<<dim3((unsigned)(seg_flag_words(T, K, M) / 1024)), dim3(256), 0, stream>>>(reinterpret_cast<u32x4*>(seg_flags));
  }
  if (!x_planes) {
    int rc = te_linear_x6_split_matrix_f32(X, T, K, 0, Xs, planes_bytes(T, K), stream_);
    if (rc != TE_OK) return rc;
    x_planes = Xs;
  }
  int wm = choose_geo((M % 256 == 0) ? 2 : 1, T, M, flags & 3, 192, true, kseg_rule(K, M, flags));
  X6Params p{};
  p.status = status;
  p.drop_handover = (flags & TE_X6_TEST_DROP_HANDOVER) ? 1 : 0;
  p.small_grid = (flags & TE_X6_TEST_SMALL_GRID) ? 1 : 0;
  p.prefer_whole = (flags & TE_X6_WHOLE_TILES) ? 1 : 0;
#ifdef TE_X6_STUDY
  if (const char* e = getenv("TE_X6_ORDER")) p.t_fast = atoi(e);
  if (const char* e = getenv("TE_X6_SNAP")) p.whole_tiles = atoi(e), p.whole_tiles_forced = 1;
#endif
  p.T = T;
  p.in_f = (int)K;
  p.out_f = (int)M;
  p.ncb = (int)te_ceil_div(T, 32);
  p.rows_w = (int)M;
  p.partial = partial;
  p.flags = flag_words;
  p.seg_part = seg_part;
  p.seg_flags = seg_flags;
  p.bias = bias;
  p.out = out;
  p.A = (const unsigned char*)w_planes;
  p.B = (const unsigned char*)x_planes;
  p.nks = (int)(K / 16);
  p.a_group_stride = (int64_t)p.nks * kRB;
  p.b_rb_stride = (int64_t)p.nks * kRB;
  int rc;
#ifdef TE_X6_STUDY
  // study builds (benchmarks/x6_gemm_bench.py): TE_X6_G_WM pins the tile geometry, TE_X6_G_PROF=1 runs the time-stamped kernel
  if (const char* e = getenv("TE_X6_G_WM")) wm = (atoi(e) == 2 && M % 256 == 0) ? 2 : 1;
  const char* pe = getenv("TE_X6_G_PROF");
  if (pe && atoi(pe) == 1) rc = (wm == 2) ? launch_x6<2, MODE_G, 5>(p, stream) : launch_x6<1, MODE_G, 5>(p, stream);
  else
#endif
  rc = launch_x6_mode<MODE_G>(wm, (flags & TE_X6_STAGES_3) != 0, p, stream);
  if (rc != TE_OK) return rc;
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// workspace: |X| planes, S planes, the accumulators of cut tiles, flags (Z-pass, C-pass)
extern "C" size_t te_linear_relprop_x6_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f) {
  if (!te_linear_relprop_x6_supported(T, in_f, out_f)) return 0;
  return te_align_up(planes_bytes(T, in_f), 256) + te_align_up(planes_bytes(T, out_f), 256) + kPartialBytes + 2 * kFlagBytes +
         seg_region_bytes(T, in_f, out_f);
}

extern "C" int te_linear_relprop_x6_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                        int64_t rows_per_sample, const float* X, const float* W, const void* w_planes,
                                        const void* x_planes, const float* Y, const float* bias, float* out, int64_t T,
                                        int64_t in_f, int64_t out_f, int flags, unsigned* status, void* ws,
                                        size_t ws_bytes, te_stream_t stream_) {
  if (!R || !X || !W || !w_planes || !Y || !out || T <= 0) return TE_ERR_INVALID_ARG;
  if (!te_linear_relprop_x6_supported(T, in_f, out_f)) return TE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < te_linear_relprop_x6_workspace_bytes(T, in_f, out_f) || !te_aligned16(ws)) return TE_ERR_WORKSPACE;
  if (!te_aligned16(X) || !te_aligned16(W) || !te_aligned16(R) || !te_aligned16(Y) || !te_aligned16(out) ||
      !te_aligned16(w_planes) || (bias && !te_aligned16(bias)) || (x_planes && !te_aligned16(x_planes)))
    return TE_ERR_UNSUPPORTED;
  if (r_scale && (rows_per_sample <= 0 || rows_per_sample > 0x7fffffff || T % rows_per_sample)) return TE_ERR_INVALID_ARG;
  const int phases = (flags & TE_X6_PHASE_MASK) ? (flags & TE_X6_PHASE_MASK) : TE_X6_PHASE_MASK;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* q = (unsigned char*)ws;
  unsigned char* Xs = q;
  q += te_align_up(planes_bytes(T, in_f), 256);
  unsigned char* Ss = q;
  q += te_align_up(planes_bytes(T, out_f), 256);
  float* partial = (float*)q;
  q += kPartialBytes;
  unsigned* flag_words = (unsigned*)q;
  q += 2 * kFlagBytes;
  unsigned char* const seg_region = q;          // shared by the two passes (they run one after the other)
  const unsigned char* wz = (const unsigned char*)w_planes;
  const unsigned char* wc = wz + te_align_up(planes_bytes(out_f, in_f), 256);

  if (phases & TE_X6_PHASE_SPLIT) {
    // the hand-over flags of both passes start from zero (a kernel, not a memset node: the whole rule stays a chain of
    // kernel nodes when it is captured in a HIP graph)
    zero_words_kernel<<<dim3(2 * kFlagBytes / 16 / 256), dim3(256), 0, stream>>>(reinterpret_cast<u32x4*>(flag_words));
  }
  if (!x_planes) {
    if (phases & TE_X6_PHASE_SPLIT) {
      int rc = te_linear_x6_split_abs_f32(X, T, in_f, Xs, planes_bytes(T, in_f), stream_);
      if (rc != TE_OK) return rc;
    }
    x_planes = Xs;
  }
  // Tile geometry per pass (the result does not depend on it, bit for bit): 256 weight rows / one 512-thread workgroup per
  // CU where that gives every CU at least one tile, else 128 rows / two 256-thread workgroups per CU.
  const int wm_max = pick_wm(in_f, out_f);
  const int pin_z = ((flags >> TE_X6_TILE_Z_SHIFT) & 3) ? ((flags >> TE_X6_TILE_Z_SHIFT) & 3) : (flags & 3);      // per-pass pins win
  const int pin_c = ((flags >> TE_X6_TILE_C_SHIFT) & 3) ? ((flags >> TE_X6_TILE_C_SHIFT) & 3) : (flags & 3);
  const int wm_z = choose_geo(wm_max, T, out_f, pin_z, 256, true, kseg_rule(in_f, out_f, flags));
  const int wm_c = choose_geo(wm_max, T, 2 * in_f, pin_c);
#ifdef TE_X6_STUDY
  const int study = (flags >> 5) & 7;      // study builds: run ablation `study` of the main loop instead
  flags &= ~0xe0;
#endif
  if ((flags & ~(0x1f | TE_X6_STAGES_3 | TE_X6_KSPLIT | TE_X6_WHOLE_TILES | TE_X6_TEST_DROP_HANDOVER | TE_X6_TEST_SMALL_GRID | 0x3c00)) != 0) return TE_ERR_INVALID_ARG;
  if (study_flags_refused(flags)) return TE_ERR_UNSUPPORTED;
  const bool three_stages = (flags & TE_X6_STAGES_3) != 0;
  int wm = 0;
  X6Params p{};
#ifdef TE_X6_STUDY
  if (const char* e = getenv("TE_X6_ORDER")) p.t_fast = atoi(e);
  if (const char* e = getenv("TE_X6_SNAP")) p.whole_tiles = atoi(e), p.whole_tiles_forced = 1;
#endif
  p.T = T;
  p.in_f = (int)in_f;
  p.out_f = (int)out_f;
  p.ncb = (int)te_ceil_div(T, 32);
  p.partial = partial;
  p.R = R;
  p.Y = Y;
  p.bias = bias;
  p.X = X;
  p.W = W;
  p.S = Ss;
  p.rs = r_scale;
  p.rs_stride = r_scale_stride;
  p.rps = (int)(r_scale ? rows_per_sample : 1);
  p.out = out;
  p.scale = 1.0f;
  p.status = status;
  p.drop_handover = (flags & TE_X6_TEST_DROP_HANDOVER) ? 1 : 0;
  p.small_grid = (flags & TE_X6_TEST_SMALL_GRID) ? 1 : 0;
  p.prefer_whole = (flags & TE_X6_WHOLE_TILES) ? 1 : 0;
  int rc;
  if (phases & TE_X6_PHASE_Z) {   // Z-pass: D[j][t] = sum_k |W|[j][k] |X|[t][k]
    p.A = wz;
    p.B = (const unsigned char*)x_planes;
    p.nks = (int)(in_f / 16);
    p.a_group_stride = (int64_t)p.nks * kRB;
    p.b_rb_stride = (int64_t)p.nks * kRB;
    wm = wm_z;
    p.rows_w = (int)out_f;
    p.flags = flag_words;
    p.seg_part = nullptr, p.seg_flags = nullptr;
    if (kseg_rule(in_f, out_f, flags) == 2) {
      p.seg_part = (float*)seg_region;
      p.seg_flags = (unsigned*)(seg_region + seg_part_bytes(T, in_f, out_f));
      zero_words_kernel<<<dim3((unsigned)(seg_flag_words(T, in_f, out_f) / 1024)), dim3(256), 0, stream>>>(
          reinterpret_cast<u32x4*>(p.seg_flags));
    }
#ifdef TE_X6_STUDY
    if (wm == 2 && study == 1) rc = launch_x6<2, MODE_Z, 1>(p, stream);
    else if (wm == 2 && study == 2) rc = launch_x6<2, MODE_Z, 2>(p, stream);
    else if (wm == 2 && study == 3) rc = launch_x6<2, MODE_Z, 3>(p, stream);
    else if (wm == 2 && study == 4) rc = launch_x6<2, MODE_Z, 4>(p, stream);
    else if (wm == 2 && study == 5) rc = launch_x6<2, MODE_Z, 5>(p, stream);
    else if (wm == 2 && study == 6) rc = launch_x6<2, MODE_Z, 6>(p, stream);
    else if (wm == 1 && study == 5) rc = launch_x6<1, MODE_Z, 5>(p, stream);
    else
#endif
    rc = launch_x6_mode<MODE_Z>(wm, three_stages, p, stream);
    if (rc != TE_OK) return rc;
  }
  if (phases & TE_X6_PHASE_C) {   // C-pass: D[(i, +-)][t] = sum_j W+-[j][i] S[t][j]
    p.A = wc;
    p.B = Ss;
    p.nks = (int)(out_f / 16);
    p.a_group_stride = (int64_t)p.nks * 2 * kRB;
    p.b_rb_stride = (int64_t)p.nks * kRB;
    wm = wm_c;
    p.rows_w = (int)(2 * in_f);
    p.flags = flag_words + kFlagBytes / 4;
    p.seg_part = nullptr, p.seg_flags = nullptr;       // (the C-pass has 2 in_f weight rows: never a launch the split helps)
#ifdef TE_X6_STUDY
    if (wm == 2 && study == 1) rc = launch_x6<2, MODE_C, 1>(p, stream);
    else if (wm == 2 && study == 2) rc = launch_x6<2, MODE_C, 2>(p, stream);
    else if (wm == 2 && study == 3) rc = launch_x6<2, MODE_C, 3>(p, stream);
    else if (wm == 2 && study == 4) rc = launch_x6<2, MODE_C, 4>(p, stream);
    else if (wm == 2 && study == 5) rc = launch_x6<2, MODE_C, 5>(p, stream);
    else if (wm == 2 && study == 6) rc = launch_x6<2, MODE_C, 6>(p, stream);
    else if (wm == 1 && study == 5) rc = launch_x6<1, MODE_C, 5>(p, stream);
    else
#endif
    rc = launch_x6_mode<MODE_C>(wm, three_stages, p, stream);
    if (rc != TE_OK) return rc;
  }
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// ----------------------------------------------------------------------------------------------------------------------
// The rule for EVERY variant and alpha on the same kernels: modules/layers_lrp.py:188-211 (variant lrp:
// S1 = sd(R, X+ W+^T), S2 = sd(R, X- W-^T) -- separate denominators) and the inhibitor half of both variants
// (layers_ours.py:225-228: R = alpha * f(pw, nw, px, nx) - beta * f(nw, pw, px, nx), beta = alpha - 1).
//   ours: Z-pass (S) -> C-pass scaled by alpha -> [beta != 0] MODE_ZI (S' from the SAME |X||W|^T product, Z' = ((Y - b) - A)/2)
//         -> MODE_CI: out += -beta (X+ . (S' W-) + X- . (S' W+))                                   18 (+18) bf16 product units
//   lrp : four one-sided launches per half: MODE_Z1 on (W+, X+) -> S1, on (W-, X-) -> S2; MODE_X on (W+^T, S1) masked by X+,
//         on (W-^T, S2) masked by X- (accumulating); the inhibitor half crosses the weight signs.   24 (+24) units
// Every product is an x6 product (six bf16 partial products, fp32 accumulation), every output one k-ordered chain (two where
// kseg_rule splits K): bitwise batch-invariant like the default rule.
extern "C" int te_linear_relprop_x6_general_supported(int64_t T, int64_t in_f, int64_t out_f, int variant) {
  if (!te_linear_relprop_x6_supported(T, in_f, out_f)) return 0;
  if (variant == TE_VARIANT_LRP) return (in_f % 128 == 0) ? 1 : 0;       // the masked products have in_f weight-side rows
  return variant == TE_VARIANT_OURS ? 1 : 0;
}

extern "C" size_t te_linear_x6_weight_planes_lrp_bytes(int64_t in_f, int64_t out_f) {
  if (!te_linear_relprop_x6_general_supported(1, in_f, out_f, TE_VARIANT_LRP)) return 0;
  // P3 of W+ and W- (rows = out_f, K = in_f), then P3 of W+^T and W-^T (rows = in_f, K = out_f)
  return 2 * te_align_up(planes_bytes(out_f, in_f), 256) + 2 * te_align_up(planes_bytes(in_f, out_f), 256);
}

extern "C" int te_linear_x6_prepare_weights_lrp_f32(const float* W, int64_t in_f, int64_t out_f, void* planes,
                                                    size_t planes_bytes_, te_stream_t stream_) {
  if (!W || !planes) return TE_ERR_INVALID_ARG;
  if (!te_linear_relprop_x6_general_supported(1, in_f, out_f, TE_VARIANT_LRP) || !te_aligned16(W)) return TE_ERR_UNSUPPORTED;
  if (planes_bytes_ < te_linear_x6_weight_planes_lrp_bytes(in_f, out_f) || !te_aligned16(planes)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  unsigned char* q = (unsigned char*)planes;
  const size_t za = te_align_up(planes_bytes(out_f, in_f), 256), ca = te_align_up(planes_bytes(in_f, out_f), 256);
  const dim3 gz((unsigned)(out_f / 32), (unsigned)te_ceil_div(in_f / 16, 8)), gc((unsigned)(in_f / 32), (unsigned)te_ceil_div(out_f / 16, 8));
  split_kernel<OP_POS, false><<<gz, dim3(256), 0, stream>>>(W, q, out_f, in_f, 3, 0);
  split_kernel<OP_NEG, false><<<gz, dim3(256), 0, stream>>>(W, q + za, out_f, in_f, 3, 0);
  split_kernel<OP_POS, true><<<gc, dim3(256), 0, stream>>>(W, q + 2 * za, in_f, out_f, 3, 0);
  split_kernel<OP_NEG, true><<<gc, dim3(256), 0, stream>>>(W, q + 2 * za + ca, in_f, out_f, 3, 0);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

constexpr int kGeneralPasses = 8;

extern "C" size_t te_linear_relprop_x6_general_workspace_bytes(int64_t T, int64_t in_f, int64_t out_f, int variant) {
  if (!te_linear_relprop_x6_general_supported(T, in_f, out_f, variant)) return 0;
  // two plane sets of the input side (|X|, or X+ and X-), two of the output side (S / S', or S1 and S2)
  return 2 * te_align_up(planes_bytes(T, in_f), 256) + 2 * te_align_up(planes_bytes(T, out_f), 256) + kPartialBytes +
         kGeneralPasses * kFlagBytes + std::max(seg_region_bytes(T, in_f, out_f), seg_region_bytes(T, out_f, in_f));
}

extern "C" int te_linear_relprop_x6_general_f32(const float* R, const float* r_scale, int64_t r_scale_stride,
                                                int64_t rows_per_sample, const float* X, const float* W,
                                                const void* w_planes, const void* w_planes_lrp, const void* x_abs_planes,
                                                const float* Y, const float* bias, float* out, int64_t T, int64_t in_f,
                                                int64_t out_f, float alpha, int variant, int flags, unsigned* status,
                                                void* ws, size_t ws_bytes, te_stream_t stream_) {
  if (!R || !X || !W || !out || T <= 0) return TE_ERR_INVALID_ARG;
  if (!te_linear_relprop_x6_general_supported(T, in_f, out_f, variant)) return TE_ERR_UNSUPPORTED;
  const bool lrp = variant == TE_VARIANT_LRP;
  if (lrp ? !w_planes_lrp : (!w_planes || !Y)) return TE_ERR_INVALID_ARG;
  if (!ws || ws_bytes < te_linear_relprop_x6_general_workspace_bytes(T, in_f, out_f, variant) || !te_aligned16(ws))
    return TE_ERR_WORKSPACE;
  if (!te_aligned16(X) || !te_aligned16(W) || !te_aligned16(R) || (Y && !te_aligned16(Y)) || !te_aligned16(out) ||
      (w_planes && !te_aligned16(w_planes)) || (w_planes_lrp && !te_aligned16(w_planes_lrp)) ||
      (bias && !te_aligned16(bias)) || (x_abs_planes && !te_aligned16(x_abs_planes)))
    return TE_ERR_UNSUPPORTED;
  if (r_scale && (rows_per_sample <= 0 || rows_per_sample > 0x7fffffff || T % rows_per_sample)) return TE_ERR_INVALID_ARG;
  if ((flags & ~(3 | TE_X6_STAGES_3 | TE_X6_KSPLIT | TE_X6_WHOLE_TILES | TE_X6_TEST_DROP_HANDOVER | TE_X6_TEST_SMALL_GRID)) != 0) return TE_ERR_INVALID_ARG;
  if (study_flags_refused(flags)) return TE_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const float beta = alpha - 1.0f;
  unsigned char* q = (unsigned char*)ws;
  unsigned char* Xa = q;                                   // |X|   or X+
  q += te_align_up(planes_bytes(T, in_f), 256);
  unsigned char* Xb = q;                                   //        or X-
  q += te_align_up(planes_bytes(T, in_f), 256);
  unsigned char* Sa = q;                                   // S / S' or S1
  q += te_align_up(planes_bytes(T, out_f), 256);
  unsigned char* Sb = q;                                   //        or S2
  q += te_align_up(planes_bytes(T, out_f), 256);
  float* partial = (float*)q;
  q += kPartialBytes;
  unsigned* flag_words = (unsigned*)q;
  q += kGeneralPasses * kFlagBytes;
  unsigned char* const seg_region = q;
  zero_words_kernel<<<dim3(kGeneralPasses * kFlagBytes / 16 / 256), dim3(256), 0, stream>>>(reinterpret_cast<u32x4*>(flag_words));

  X6Params base{};
  base.T = T;
  base.ncb = (int)te_ceil_div(T, 32);
  base.partial = partial;
  base.R = R;
  base.Y = Y;
  base.bias = bias;
  base.X = X;
  base.W = W;
  base.rs = r_scale;
  base.rs_stride = r_scale_stride;
  base.rps = (int)(r_scale ? rows_per_sample : 1);
  base.out = out;
  // status = NULL: ONE per-call error word for all launches of the rule, in the last flag region -- which no launch uses (at
  // most four passes per half) and which the reset between the two halves of variant lrp leaves alone, so a failure of the
  // first half is not lost
  base.status = status ? status : flag_words + (size_t)(kGeneralPasses - 1) * (kFlagBytes / 4) + kErrWord;
  base.drop_handover = (flags & TE_X6_TEST_DROP_HANDOVER) ? 1 : 0;
  base.small_grid = (flags & TE_X6_TEST_SMALL_GRID) ? 1 : 0;
  base.prefer_whole = (flags & TE_X6_WHOLE_TILES) ? 1 : 0;
  const bool three_stages = (flags & TE_X6_STAGES_3) != 0;
  const int pin = flags & 3;
  int pass = 0;
  // one launch: a Z-like pass (K = in_f, weight-side rows = out_f) or an output-side pass (K = out_f)
  auto seg_setup = [&](X6Params& p, int64_t K, int64_t rows_w) {
    p.seg_part = nullptr, p.seg_flags = nullptr;
    if (kseg_rule(K, rows_w, flags) == 2) {
      p.seg_part = (float*)seg_region;
      p.seg_flags = (unsigned*)(seg_region + seg_part_bytes(T, K, rows_w));
      zero_words_kernel<<<dim3((unsigned)(seg_flag_words(T, K, rows_w) / 1024)), dim3(256), 0, stream>>>(
          reinterpret_cast<u32x4*>(p.seg_flags));
    }
  };
  auto z_like = [&](auto mode_tag, const unsigned char* A, const unsigned char* B, unsigned char* S) -> int {
    constexpr int MODE = decltype(mode_tag)::value;
    X6Params p = base;
    p.in_f = (int)in_f, p.out_f = (int)out_f;
    p.A = A, p.B = B, p.S = S;
    p.nks = (int)(in_f / 16);
    p.a_group_stride = (int64_t)p.nks * kRB;
    p.b_rb_stride = (int64_t)p.nks * kRB;
    p.rows_w = (int)out_f;
    p.flags = flag_words + (size_t)(pass++) * (kFlagBytes / 4);
    seg_setup(p, in_f, out_f);
    const int wm = choose_geo(pick_wm(in_f, out_f), T, out_f, pin, 256, true, kseg_rule(in_f, out_f, flags));
    return launch_x6_mode<MODE>(wm, three_stages, p, stream);
  };
  int rc;
  if (!lrp) {
    const unsigned char* wz = (const unsigned char*)w_planes;
    const unsigned char* wc = wz + te_align_up(planes_bytes(out_f, in_f), 256);
    if (!x_abs_planes) {
      rc = te_linear_x6_split_abs_f32(X, T, in_f, Xa, planes_bytes(T, in_f), stream_);
      if (rc != TE_OK) return rc;
      x_abs_planes = Xa;
    }
    auto c_like = [&](auto mode_tag, float scale) -> int {
      constexpr int MODE = decltype(mode_tag)::value;
      X6Params p = base;
      p.in_f = (int)in_f, p.out_f = (int)out_f;
      p.A = wc, p.B = Sa;
      p.nks = (int)(out_f / 16);
      p.a_group_stride = (int64_t)p.nks * 2 * kRB;
      p.b_rb_stride = (int64_t)p.nks * kRB;
      p.rows_w = (int)(2 * in_f);
      p.scale = scale;
      p.flags = flag_words + (size_t)(pass++) * (kFlagBytes / 4);
      return launch_x6_mode<MODE>(choose_geo(pick_wm(in_f, out_f), T, 2 * in_f, pin), three_stages, p, stream);
    };
    rc = z_like(std::integral_constant<int, MODE_Z>{}, wz, (const unsigned char*)x_abs_planes, Sa);
    if (rc != TE_OK) return rc;
    rc = c_like(std::integral_constant<int, MODE_C>{}, alpha);
    if (rc != TE_OK) return rc;
    if (beta != 0.0f) {
      rc = z_like(std::integral_constant<int, MODE_ZI>{}, wz, (const unsigned char*)x_abs_planes, Sa);      // S' over S
      if (rc != TE_OK) return rc;
      rc = c_like(std::integral_constant<int, MODE_CI>{}, -beta);
      if (rc != TE_OK) return rc;
    }
  } else {
    const size_t za = te_align_up(planes_bytes(out_f, in_f), 256), ca = te_align_up(planes_bytes(in_f, out_f), 256);
    const unsigned char* wq = (const unsigned char*)w_planes_lrp;
    const unsigned char* Wp = wq, *Wn = wq + za, *WpT = wq + 2 * za, *WnT = wq + 2 * za + ca;
    const dim3 gx((unsigned)te_ceil_div(T, 32), (unsigned)te_ceil_div(in_f / 16, 8));
    split_kernel<OP_POS, false><<<gx, dim3(256), 0, stream>>>(X, Xa, T, in_f, 3, 0);
    split_kernel<OP_NEG, false><<<gx, dim3(256), 0, stream>>>(X, Xb, T, in_f, 3, 0);
    // out (+)= scale * (X^sign . (S W^t)):  product [T, out_f] x [out_f, in_f], weight-side rows = in_f
    auto x_like = [&](const unsigned char* AT, const unsigned char* S, int sign, float scale, int accum) -> int {
      X6Params p = base;
      p.in_f = (int)out_f, p.out_f = (int)in_f;           // (plain-product naming: K, M)
      p.A = AT, p.B = S;
      p.nks = (int)(out_f / 16);
      p.a_group_stride = (int64_t)p.nks * kRB;
      p.b_rb_stride = (int64_t)p.nks * kRB;
      p.rows_w = (int)in_f;
      p.scale = scale, p.accum = accum, p.x_sign = sign;
      p.flags = flag_words + (size_t)(pass++ % kGeneralPasses) * (kFlagBytes / 4);
      seg_setup(p, out_f, in_f);
      const int wm = choose_geo((in_f % 256 == 0) ? 2 : 1, T, in_f, pin, 192, true, kseg_rule(out_f, in_f, flags));
      return launch_x6_mode<MODE_X>(wm, three_stages, p, stream);
    };
    for (int half = 0; half < (beta != 0.0f ? 2 : 1); ++half) {
      // half 0: f(pw, nw, px, nx) (activator); half 1: f(nw, pw, px, nx) (inhibitor) -- the weight signs cross
      const unsigned char* w1 = half ? Wn : Wp, *w2 = half ? Wp : Wn, *w1T = half ? WnT : WpT, *w2T = half ? WpT : WnT;
      const float sc = half ? -beta : alpha;
      if (half) {       // the second half reuses the flag regions of the first: reset them (stream-ordered after their use)
        pass = 0;
        static_assert(kGeneralPasses >= 5, "four passes per half + the region of the per-call error word");
        zero_words_kernel<<<dim3(4 * kFlagBytes / 16 / 256), dim3(256), 0, stream>>>(reinterpret_cast<u32x4*>(flag_words));
      }
      rc = z_like(std::integral_constant<int, MODE_Z1>{}, w1, Xa, Sa);
      if (rc != TE_OK) return rc;
      rc = z_like(std::integral_constant<int, MODE_Z1>{}, w2, Xb, Sb);
      if (rc != TE_OK) return rc;
      rc = x_like(w1T, Sa, +1, sc, half);
      if (rc != TE_OK) return rc;
      rc = x_like(w2T, Sb, -1, sc, 1);
      if (rc != TE_OK) return rc;
    }
  }
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}

// 1 if a bounded wait of the last te_linear_relprop_x6_f32 call on this workspace expired (a predecessor workgroup never
// published its accumulators): the result of that call is invalid.  Synchronises the stream.
extern "C" int te_linear_relprop_x6_check(const void* ws, int64_t T, int64_t in_f, int64_t out_f, te_stream_t stream_) {
  if (!ws || !te_linear_relprop_x6_supported(T, in_f, out_f)) return TE_ERR_INVALID_ARG;
  const unsigned char* q = (const unsigned char*)ws + te_align_up(planes_bytes(T, in_f), 256) +
                           te_align_up(planes_bytes(T, out_f), 256) + kPartialBytes;
  unsigned host[2][1024];      // the 512 flags + the error word of each pass
  hipError_t e = hipSuccess;
  for (int pass = 0; pass < 2 && e == hipSuccess; ++pass)
    e = hipMemcpyAsync(host[pass], q + pass * kFlagBytes, sizeof(host[pass]), hipMemcpyDeviceToHost, (hipStream_t)stream_);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream_);
  if (e != hipSuccess) return (int)e;
  // any non-zero word after a completed call is an error (a flag nobody consumed, or the error word)
  for (int pass = 0; pass < 2; ++pass)
    for (unsigned i = 0; i < 1024; ++i)
      if (host[pass][i]) return 1;
  return 0;
}
