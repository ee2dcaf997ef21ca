// gemm_bf16x6.hip -- STUDY (not part of libte_relprop.so): an fp32-accurate GEMM on bf16 MFMAs.
//
//   C[M,N] = A[M,K] B[N,K]^T   with A, B given in fp32 and used as the exact sum of three bf16 parts each,
//   a = a0 + a1 + a2, and the six partial products above 2^-24 |a||b| kept ("x6"):
//   a0 b0 + a0 b1 + a1 b0 + a0 b2 + a2 b0 + a1 b1  -- every product exact in the fp32 accumulator of
//   v_mfma_f32_32x32x16_bf16, six of them for one fp32 product at 16x the fp32-MFMA rate per instruction.
//
// Two kernels:
//   split3_kernel    fp32 [R,K] -> bf16 planes, stored per row and per 32-k block as [3][32] (192 B: the tile row of
//                    one K-step is contiguous in memory and in LDS)
//   gemm_x6_kernel   128x128 tile, 256 threads (2 x 2 waves, 64x64 per wave = four 32x32 accumulators), K-step 32 = two
//                    K16 slices; operands staged global -> registers -> LDS (rows padded to 208 B: 16-B fragments of
//                    eight rows hit eight bank groups), next K-step's loads in flight during the MFMAs
//
// build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC benchmarks/studies/gemm_bf16x6.hip -o benchmarks/studies/libgemm_bf16x6.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int ROWB = 192;            // bytes of one row of one K-step in memory: 3 planes x 32 bf16
constexpr int LROW = 208;            // ... in LDS (padded: 52 dwords, 52 mod 32 = 20 -> conflict-free 16-B reads over 8 rows)

__device__ __forceinline__ unsigned short bf16_bits_rn(float x) {      // round to nearest even (finite inputs)
  const unsigned u = __float_as_uint(x);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// one thread per (row, 8 consecutive k): reads 8 floats, writes 8 bf16 into each of the three planes of its 32-k block
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                     int64_t R, int64_t K) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over R * K / 8
  const int64_t per_row = K >> 3;
  if (idx >= R * per_row) return;
  const int64_t row = idx / per_row;
  const int c8 = (int)(idx - row * per_row);                         // which group of 8 k
  const float* s = src + row * K + (int64_t)c8 * 8;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
  float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  unsigned short p[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float r = v[e];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      p[q][e] = bf16_bits_rn(r);
      r = r - bf16_to_f32(p[q][e]);      // exact: the residual of a round-to-nearest bf16 fits fp32
    }
  }
  // row layout: [K/32 blocks][3 planes][32 k]
  unsigned short* d = dst + row * (K / 32) * 96 + (int64_t)(c8 >> 2) * 96 + (c8 & 3) * 8;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = (unsigned)p[q][2 * e] | ((unsigned)p[q][2 * e + 1] << 16);
    *reinterpret_cast<u32x4*>(d + q * 32) = w;
  }
}

// A_s, B_s: split operands ([rows][K/32][3][32] bf16).  M, N multiples of 128, K a multiple of 32 (study shapes).
__global__ __launch_bounds__(kThreads) void gemm_x6_kernel(const unsigned short* __restrict__ A_s,
                                                           const unsigned short* __restrict__ B_s,
                                                           float* __restrict__ C, int M, int N, int K, int nbn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* At = smem;                    // [BM][LROW]
  unsigned char* Bt = smem + BM * LROW;        // [BN][LROW]
  const int tile = blockIdx.x;
  const int row0 = (tile / nbn) * BM, col0 = (tile % nbn) * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = K / BK;
  const int64_t rowbytes = (int64_t)(K / 32) * ROWB;

  // staging: 128 rows x 12 chunks of 16 B per operand = 1536 chunks, 6 per thread
  u32x4 ra[6], rb[6];
  auto load_next = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int row = idx / 12, c = idx - row * 12;
      ra[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(A_s) + (int64_t)(row0 + row) * rowbytes +
                                              (int64_t)kt * ROWB + c * 16);
      rb[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(B_s) + (int64_t)(col0 + row) * rowbytes +
                                              (int64_t)kt * ROWB + c * 16);
    }
  };
  auto store_tiles = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int row = idx / 12, c = idx - row * 12;
      *reinterpret_cast<u32x4*>(At + row * LROW + c * 16) = ra[i];
      *reinterpret_cast<u32x4*>(Bt + row * LROW + c * 16) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;

  // fragment of plane q, K16 slice s of row r: 16 B at r * LROW + q * 64 + s * 32 + kh * 16
  const unsigned char* ap = At + (wm * 64 + lr) * LROW + kh * 16;
  const unsigned char* bp = Bt + (wn * 64 + lr) * LROW + kh * 16;

  load_next(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                           // the previous step's fragment reads are done
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_next(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          a[mi][q] = *reinterpret_cast<const bf16x8*>(ap + mi * 32 * LROW + q * 64 + s * 32);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          b[ni][q] = *reinterpret_cast<const bf16x8*>(bp + ni * 32 * LROW + q * 64 + s * 32);
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first within the slice
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          f32x16 c = acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
          acc[mi][ni] = c;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // C: row = (e & 3) + 8 (e >> 2) + 4 kh, column = lr of each 32x32 block
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      float* c = C + (int64_t)(row0 + wm * 64 + mi * 32 + 4 * kh) * N + col0 + wn * 64 + ni * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) c[(int64_t)((e & 3) + 8 * (e >> 2)) * N] = acc[mi][ni][e];
    }
}

}  // namespace

extern "C" int split3_f32(const float* src, void* dst, int64_t R, int64_t K, void* stream) {
  if (K % 32) return -1;
  const int64_t n = R * (K / 8);
  split3_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(src, (unsigned short*)dst, R, K);
  return (int)hipGetLastError();
}

extern "C" int gemm_x6_f32(const void* A_s, const void* B_s, float* C, int M, int N, int K, void* stream) {
  if (M % BM || N % BN || K % BK) return -1;
  const size_t lds = (size_t)(BM + BN) * LROW;
  const int nbn = N / BN;
  gemm_x6_kernel<<<dim3((unsigned)((M / BM) * nbn)), dim3(kThreads), lds, (hipStream_t)stream>>>(
      (const unsigned short*)A_s, (const unsigned short*)B_s, C, M, N, K, nbn);
  return (int)hipGetLastError();
}
