// te_heatmap.hip -- the consumer of a relevance map: bilinear x`scale` up-sampling of the [g,g]
// patch map to image resolution, per-map min-max normalisation and the mean-threshold foreground mask, as
// baselines/ViT/imagenet_seg_eval.py:214-222 and generate_visualizations.py:99-100 do with
// F.interpolate(scale_factor=16, mode='bilinear') + (Res - Res.min()) / (Res.max() - Res.min()) + Res.gt(Res.mean()).
// One block per map (batch = independent maps): the 224 x 224 image is 196 KB, the source 784 B -- every pass
// recomputes the interpolation from the LDS copy of the patch map instead of re-reading an intermediate.
#include "te_common.h"

namespace {

constexpr int kThreads = 1024;

// ATen's area_pixel_compute_source_index (align_corners = false, not cubic) and 2-tap weights
__device__ __forceinline__ void src_index(int o, float rscale, int in_size, int& i0, int& step, float& l0, float& l1) {
  float s = rscale * ((float)o + 0.5f) - 0.5f;
  s = s < 0.0f ? 0.0f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  step = (i0 < in_size - 1) ? 1 : 0;
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

__device__ __forceinline__ float bilinear(const float* __restrict__ m, int g, int oy, int ox, float rscale) {
  int y0, ys, x0, xs;
  float ly0, ly1, lx0, lx1;
  src_index(oy, rscale, g, y0, ys, ly0, ly1);
  src_index(ox, rscale, g, x0, xs, lx0, lx1);
  const float* r0 = m + y0 * g + x0;
  const float* r1 = r0 + ys * g;
  const float t0 = lx0 * r0[0] + lx1 * r0[xs];
  const float t1 = lx0 * r1[0] + lx1 * r1[xs];
  return ly0 * t0 + ly1 * t1;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}

__global__ __launch_bounds__(kThreads) void heatmap_kernel(const float* __restrict__ maps, float* __restrict__ heat,
                                                           float* __restrict__ mask, int g, int scale, int normalise) {
  extern __shared__ float sm[];                 // [g*g] patch map, then 16 floats + 16 doubles of reduction scratch
  float* pm = sm;
  const int gg = g * g;
  float* red_f = sm + ((gg + 3) & ~3);
  double* red_d = reinterpret_cast<double*>(red_f + 2 * (kThreads / 64));
  const int b = blockIdx.x, side = g * scale, P = side * side;
  const float rscale = 1.0f / (float)scale;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kThreads / 64;
  for (int i = threadIdx.x; i < gg; i += kThreads) pm[i] = maps[(int64_t)b * gg + i];
  __syncthreads();
  float lo = 0.0f, den = 1.0f;
  if (normalise) {
    float mn = INFINITY, mx = -INFINITY;
    for (int p = threadIdx.x; p < P; p += kThreads) {
      const float v = bilinear(pm, g, p / side, p % side, rscale);
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) {
      red_f[wave] = mn;
      red_f[nw + wave] = mx;
    }
    __syncthreads();
    mn = red_f[0];
    mx = red_f[nw];
    for (int w = 1; w < nw; ++w) {
      mn = fminf(mn, red_f[w]);
      mx = fmaxf(mx, red_f[nw + w]);
    }
    lo = mn;
    den = mx - mn;
    __syncthreads();
  }
  float* hb = heat + (int64_t)b * P;
  double sum = 0.0;
  for (int p = threadIdx.x; p < P; p += kThreads) {
    float v = bilinear(pm, g, p / side, p % side, rscale);
    if (normalise) v = (v - lo) / den;
    hb[p] = v;
    sum += (double)v;
  }
  if (mask == nullptr) return;
  sum = te_wave_sum(sum);
  if (lane == 0) red_d[wave] = sum;
  __syncthreads();
  double tot = 0.0;
  for (int w = 0; w < nw; ++w) tot += red_d[w];
  const float mean = (float)(tot / (double)P);
  float* mb = mask + (int64_t)b * P;
  for (int p = threadIdx.x; p < P; p += kThreads) {
    float v = bilinear(pm, g, p / side, p % side, rscale);
    if (normalise) v = (v - lo) / den;
    mb[p] = (v > mean) ? 1.0f : 0.0f;       // Res.gt(Res.mean()) ; NaN compares false like the reference's scrub to 0
  }
}

}  // namespace

extern "C" int te_heatmap_f32(const float* maps, float* heat, float* fg_mask, int64_t B, int64_t g, int64_t scale,
                              int normalise, te_stream_t stream_) {
  if (!maps || !heat || B <= 0 || g <= 0 || scale <= 0) return TE_ERR_INVALID_ARG;
  if (g * g > 8192 || g * scale > 32768) return TE_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(((g * g + 3) & ~3) + 2 * (kThreads / 64)) * sizeof(float) + (kThreads / 64) * sizeof(double) + 8;
  heatmap_kernel<<<dim3((unsigned)B), dim3(kThreads), lds, (hipStream_t)stream_>>>(maps, heat, fg_mask, (int)g, (int)scale,
                                                                                    normalise);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
