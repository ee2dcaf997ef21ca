// mfma_peak.hip -- measurement infrastructure (not product code): what fp32-MFMA rate does THIS MI355X sustain
// on random operands under DVFS?  A register-only v_mfma_f32_32x32x2_f32 loop (no LDS, no HBM) is the
// speed-of-light reference the Linear.relprop kernels are priced against next to the 157.3 TF datasheet figure.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC independent accumulators per wave; operands change every iteration (cheap VALU) so the data path toggles
// like a real GEMM's does.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* seed, int iters) {
  f32x16 acc[NACC];
  const float s = seed[threadIdx.x & 63];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;
  float x = s, y = 0.5f - s;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
      x = -x * 0.999f + 0.001f;   // sign flips + mantissa churn
      y = y * 0.998f - 0.002f * x;
    }
  }
  float r = 0.0f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// returns TFLOP/s; blocks_per_cu x 256 CUs blocks of 256 threads (4 waves), `ms_target` ~ run length
extern "C" double mfma_peak_tflops(int nacc, int blocks_per_cu, int iters, float* scratch, const float* seed,
                                   double* ms_out) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() {
    if (nacc == 4) mfma_loop<4><<<grid, 256>>>(scratch, seed, iters);
    else mfma_loop<8><<<grid, 256>>>(scratch, seed, iters);
  };
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  if (ms_out) *ms_out = ms;
  const double flops = (double)grid * 4 /*waves*/ * (double)iters * 8 * nacc * 4096.0;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return flops / (ms * 1e-3) / 1e12;
}

// ---- bf16: v_mfma_f32_32x32x16_bf16 (the instruction a split-operand fp32 GEMM would run six times per fp32 product,
// DESIGN.md section 7).  Same structure: NACC independent accumulators, operands perturbed every iteration.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_bf16(float* out, const float* seed, int iters) {
  f32x16 acc[NACC];
  const float s = seed[threadIdx.x & 63];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;
  bf16x8 x, y;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = (__bf16)(s + 0.01f * e);
    y[e] = (__bf16)(0.5f - s - 0.02f * e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
      // cheap churn: flip sign bits / low mantissa bits of the packed operands with two integer ops
      s16x8 xi = __builtin_bit_cast(s16x8, x), yi = __builtin_bit_cast(s16x8, y);
      xi = xi ^ (short)0x8003;
      yi = yi ^ (short)0x0005;
      x = __builtin_bit_cast(bf16x8, xi);
      y = __builtin_bit_cast(bf16x8, yi);
    }
  }
  float r = 0.0f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// returns TFLOP/s of bf16 MFMA (2 * 32 * 32 * 16 flops per instruction)
extern "C" double mfma_peak_bf16_tflops(int nacc, int blocks_per_cu, int iters, float* scratch, const float* seed,
                                        double* ms_out) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&]() {
    if (nacc == 4) mfma_loop_bf16<4><<<grid, 256>>>(scratch, seed, iters);
    else mfma_loop_bf16<8><<<grid, 256>>>(scratch, seed, iters);
  };
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  if (ms_out) *ms_out = ms;
  const double flops = (double)grid * 4 /*waves*/ * (double)iters * 8 * nacc * 32768.0;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return flops / (ms * 1e-3) / 1e12;
}

// ---- shader-clock probe: one small block that samples the shader-cycle counter (s_memtime, DVFS-dependent) against
// the constant 100 MHz real-time counter (s_memrealtime) while another stream runs the kernel under test.
__global__ void clock_probe_kernel(unsigned long long* out, long long spin_ref_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long c0 = clock64(), r0 = wall_clock64();
  unsigned long long r1 = r0;
  while ((long long)(r1 - r0) < spin_ref_ticks) r1 = wall_clock64();
  const unsigned long long c1 = clock64();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}

// launches the probe on `stream`; out = 2 x uint64 device words {shader cycles, 100 MHz ticks}
extern "C" void clock_probe_launch(unsigned long long* out, long long spin_us, void* stream) {
  clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out, spin_us * 100);
}
