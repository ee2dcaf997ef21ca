// te_perturb.hip -- perturbation-test input builder (gfx950), SURVEY.md section 8(f) row 4.
//
// Reference (baselines/ViT/pertubation_eval_from_hdf5.py:88-101), per batch and per perturbation step s:
//     _, idx = torch.topk(vis, k_s, dim=-1)                       vis [B, HW]: relevance per pixel
//     _data  = data.clone().reshape(B, C, HW).scatter_(-1, idx.unsqueeze(1).repeat(1, C, 1), 0)
//     _norm  = normalize(_data)                                   (x - mean[c]) / std[c]
// nine times (k_s = 10 % ... 90 % of the pixels): 9 x (sort-based top-k over 50,176 values per row + index tensor
// repeat + clone + scatter + normalise).  Here: ONE selection kernel finds, for all steps at once, the k_s-th largest
// value of every row (4-pass 8-bit radix select on order-preserving keys, one block per sample, the row stays in
// L2), and ONE streaming kernel writes the S masked + normalised copies (HBM-bound: S*B*C*HW*4 B written, the image
// read once per step).
//
// Ties: torch.topk leaves the choice among equal values unspecified (and bilinearly up-sampled maps DO have ties:
// the rows / columns outside the outermost patch centres are replicated).  This library removes exactly k_s pixels,
// taking tied values in ascending index order; any tie-break yields the same multiset of removed relevance values.
#include "te_common.h"

namespace {

constexpr int kSelThreads = 1024;
constexpr int kMaxSteps = TE_PERTURB_MAX_STEPS;

struct Steps {
  int n;
  int64_t k[kMaxSteps];
};
struct Norm {
  float mean[TE_PERTURB_MAX_CHANNELS], std[TE_PERTURB_MAX_CHANNELS];
};

// order-preserving key: a > b (as floats, -0 == +0, NaN largest as in torch.topk) <=> key(a) > key(b)
__device__ __forceinline__ uint32_t te_key(float v) {
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0;                       // -0 -> +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// sel[b][s] = {threshold key, cut}: pixel i of sample b is removed at step s iff
//   key_i > thr  ||  (key_i == thr && i < cut)
__global__ __launch_bounds__(kSelThreads) void perturb_select_kernel(const float* __restrict__ vis,
                                                                   uint32_t* __restrict__ sel, int64_t HW, Steps st) {
  __shared__ uint32_t hist[kMaxSteps][256];
  __shared__ uint32_t prefix[kMaxSteps];     // key bits fixed so far (high bits)
  __shared__ int64_t rank[kMaxSteps];        // remaining rank (1-based, from the top) inside the prefix group
  __shared__ uint32_t ties[kMaxSteps];       // number of elements equal to the final threshold key
  __shared__ uint32_t wave_cnt[kSelThreads / TE_WAVE];
  __shared__ uint32_t carry;
  const int S = st.n;
  const float* row = vis + (int64_t)blockIdx.x * HW;
  if (threadIdx.x < S) {
    prefix[threadIdx.x] = 0;
    rank[threadIdx.x] = st.k[threadIdx.x];
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < S * 256; i += kSelThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < HW; i += kSelThreads) {
      const uint32_t key = te_key(row[i]);
      const uint32_t digit = (key >> shift) & 0xffu;
      // (pass 0: every element is a candidate of every step; later: only those matching the step's prefix)
      const uint32_t hi = (pass == 0) ? 0u : (key >> (shift + 8));
      for (int s = 0; s < S; ++s)
        if (pass == 0 || hi == (prefix[s] >> (shift + 8))) atomicAdd(&hist[s][digit], 1u);
    }
    __syncthreads();
    if (threadIdx.x < S) {
      const int s = threadIdx.x;
      int64_t r = rank[s];
      int d = 255;
      if (r >= 1) {
        for (; d > 0; --d) {
          const uint32_t c = hist[s][d];
          if (r <= (int64_t)c) break;
          r -= c;
        }
        // (if k exceeds the row length the walk ends in digit 0 with r > hist: handled by the host-side clamp)
      }
      prefix[s] |= ((uint32_t)d) << shift;
      rank[s] = r;
      if (pass == 3) ties[s] = hist[s][d];
    }
    __syncthreads();
  }
  // tie cut: the index just past the rank[s]-th element equal to the threshold, in ascending index order
  for (int s = 0; s < S; ++s) {
    const int64_t k = st.k[s];
    uint32_t* out = sel + ((int64_t)blockIdx.x * S + s) * 2;
    if (k <= 0) {                                    // nothing removed
      if (threadIdx.x == 0) {
        out[0] = 0xffffffffu;
        out[1] = 0;
      }
      continue;
    }
    const uint32_t thr = prefix[s];
    const int64_t need = rank[s];
    if (k >= HW || need >= (int64_t)ties[s]) {       // every tied element is taken
      if (threadIdx.x == 0) {
        out[0] = (k >= HW) ? 0u : thr;
        out[1] = (uint32_t)HW;
      }
      continue;
    }
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < HW; base += kSelThreads) {
      const int64_t i = base + threadIdx.x;
      const bool tie = (i < HW) && te_key(row[i]) == thr;
      const uint64_t bal = __ballot(tie);
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t before = carry;
      for (int w = 0; w < wave; ++w) before += wave_cnt[w];
      const uint32_t incl = before + (uint32_t)__popcll(bal & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1)));
      if (tie && (int64_t)incl == need) {
        out[0] = thr;
        out[1] = (uint32_t)(i + 1);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t tot = carry;
        for (int w = 0; w < kSelThreads / TE_WAVE; ++w) tot += wave_cnt[w];
        carry = tot;
      }
      __syncthreads();
      if ((int64_t)carry >= need) break;             // block-uniform
    }
    __syncthreads();
  }
}

// out[s][b][c][i] = ((removed(s,b,i) ? 0 : data[b][c][i]) - mean[c]) / std[c]
template <int VEC>
__global__ __launch_bounds__(256) void perturb_apply_kernel(const float* __restrict__ vis,
                                                            const float* __restrict__ data,
                                                            const uint32_t* __restrict__ sel, float* __restrict__ out,
                                                            int64_t B, int64_t C, int64_t HW, int S, Norm nm) {
  const int64_t b = blockIdx.y;
  const int s = blockIdx.z;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i0 >= HW) return;
  const uint32_t thr = sel[(b * S + s) * 2], cut = sel[(b * S + s) * 2 + 1];
  float keep[VEC];
  if constexpr (VEC == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(vis + b * HW + i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t key = te_key(v[j]);
      keep[j] = (key > thr || (key == thr && (uint32_t)(i0 + j) < cut)) ? 0.0f : 1.0f;
    }
  } else {
    const uint32_t key = te_key(vis[b * HW + i0]);
    keep[0] = (key > thr || (key == thr && (uint32_t)i0 < cut)) ? 0.0f : 1.0f;
  }
  for (int64_t c = 0; c < C; ++c) {
    const float* src = data + (b * C + c) * HW + i0;
    float* dst = out + (((int64_t)s * B + b) * C + c) * HW + i0;
    const float mean = nm.mean[c], sd = nm.std[c];
    if constexpr (VEC == 4) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(src);
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = ((keep[j] != 0.0f ? x[j] : 0.0f) - mean) / sd;
      __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(dst));
    } else {
      dst[0] = ((keep[0] != 0.0f ? src[0] : 0.0f) - mean) / sd;
    }
  }
}

}  // namespace

extern "C" size_t te_perturb_workspace_bytes(int64_t B, int64_t n_steps) {
  if (B <= 0 || n_steps <= 0 || n_steps > kMaxSteps) return 0;
  return te_align_up((size_t)B * (size_t)n_steps * 2 * sizeof(uint32_t), 256);
}

extern "C" int te_perturb_f32(const float* vis, const float* data, float* out, int64_t B, int64_t C, int64_t HW,
                              const int64_t* ks, int64_t n_steps, const float* mean, const float* std_, void* ws,
                              size_t ws_bytes, te_stream_t stream_) {
  if (!vis || !data || !out || !ks || B <= 0 || C <= 0 || HW <= 0 || n_steps <= 0) return TE_ERR_INVALID_ARG;
  if (n_steps > kMaxSteps || C > TE_PERTURB_MAX_CHANNELS || HW > (int64_t)0x7fffffff) return TE_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < te_perturb_workspace_bytes(B, n_steps)) return TE_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  Steps st;
  st.n = (int)n_steps;
  for (int s = 0; s < kMaxSteps; ++s) st.k[s] = 0;
  for (int s = 0; s < st.n; ++s) st.k[s] = ks[s] < 0 ? 0 : (ks[s] > HW ? HW : ks[s]);
  Norm nm;
  for (int c = 0; c < TE_PERTURB_MAX_CHANNELS; ++c) {
    nm.mean[c] = (mean && c < C) ? mean[c] : 0.0f;
    nm.std[c] = (std_ && c < C) ? std_[c] : 1.0f;
  }
  uint32_t* sel = (uint32_t*)ws;
  perturb_select_kernel<<<dim3((unsigned)B), dim3(kSelThreads), 0, stream>>>(vis, sel, HW, st);
  const bool vec = (HW % 4 == 0) && te_aligned16(vis) && te_aligned16(data) && te_aligned16(out);
  if (vec)
    perturb_apply_kernel<4><<<dim3((unsigned)te_ceil_div(HW / 4, 256), (unsigned)B, (unsigned)n_steps), dim3(256), 0,
                              stream>>>(vis, data, sel, out, B, C, HW, (int)n_steps, nm);
  else
    perturb_apply_kernel<1><<<dim3((unsigned)te_ceil_div(HW, 256), (unsigned)B, (unsigned)n_steps), dim3(256), 0,
                              stream>>>(vis, data, sel, out, B, C, HW, (int)n_steps, nm);
  TE_RETURN_IF_LAUNCH_FAILED();
  return TE_OK;
}
