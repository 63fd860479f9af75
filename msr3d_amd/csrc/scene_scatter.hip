// scene_scatter.hip -- the hand-off from the hot path to the LLM
// (/root/reference/model/msr3d/msr3d.py:277-287):
//     scene_embeds = llm_proj(obj_tokens).to(inputs_embeds.dtype)
//     inputs_embeds[where(input_ids == scene_sp_token)] = scene_embeds.reshape(-1, E)
//     attention_mask[same positions] = obj_masks
// The reference's torch.where is a host sync per step.  Here a
// one-block kernel builds the placeholder map on the device (row-major order, k-th
// placeholder <- k-th scene token, exactly the indexed-assignment order), a second kernel
// writes rows and mask entries through it, casting fp32 -> fp16 / bf16 / fp32 on the way.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"

namespace {

// dest[k] = flat (b*T + t) position of the k-th placeholder in row-major order, k < cap;
// *count = total number of placeholders found.  One block; rows are walked in order with a
// running offset, each row scanned 256 positions at a time with a block-wide prefix count.
__global__ __launch_bounds__(256) void scatter_map_kernel(int B, int T,
                                                          const long long *__restrict__ ids,
                                                          long long token, int cap,
                                                          int *__restrict__ dest,
                                                          int *__restrict__ count) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    for (int t0 = 0; t0 < T; t0 += 256) {
      const int t = t0 + tid;
      const bool hit = t < T && ids[(size_t)b * T + t] == token;
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int before = running;
      for (int w = 0; w < wave; ++w) before += wave_cnt[w];
      const int k = before + __popcll(m & ((1ull << lane) - 1ull));
      if (hit && k < cap) dest[k] = b * T + t;
      __syncthreads();
      if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
  }
  if (tid == 0) *count = running;
}

template <typename OutT>
__device__ __forceinline__ OutT cvt(float v);
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ __half cvt<__half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ unsigned short cvt<unsigned short>(float v) {   // bf16, RNE
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// rows: n = min(count, cap) scene tokens.  embeds_out[dest[k]] = cvt(scene[k]); mask likewise.
template <typename OutT>
__global__ void scatter_rows_kernel(int cap, int E, const int *__restrict__ dest,
                                    const int *__restrict__ count,
                                    const float *__restrict__ scene, OutT *__restrict__ embeds,
                                    const unsigned char *__restrict__ scene_mask,
                                    long long *__restrict__ attn_mask) {
  const int n = min(*count, cap);
  const int k = blockIdx.x;
  if (k >= n) return;
  const size_t row = (size_t)dest[k];
  const float *src = scene + (size_t)k * E;
  OutT *dst = embeds + row * E;
  for (int c = threadIdx.x; c < E; c += blockDim.x) dst[c] = cvt<OutT>(src[c]);
  if (threadIdx.x == 0 && attn_mask && scene_mask) attn_mask[row] = scene_mask[k] ? 1 : 0;
}

}  // namespace

extern "C" {

int msr3d_scene_scatter(int B, int T, int n_scene, int E, const long long *input_ids,
                        long long scene_token, const float *scene_embeds,
                        const unsigned char *scene_mask, int out_dtype, void *inputs_embeds,
                        long long *attention_mask, int *map_ws, int *count_out,
                        msr3d_stream_t stream) {
  if (B < 0 || T <= 0 || n_scene < 0 || E <= 0) return MSR3D_EINVAL;
  if (B == 0 || n_scene == 0) return 0;
  if (!input_ids || !scene_embeds || !inputs_embeds || !map_ws || !count_out) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  scatter_map_kernel<<<1, 256, 0, st>>>(B, T, input_ids, scene_token, n_scene, map_ws, count_out);
  const int thr = E >= 256 ? 256 : 64;
  switch (out_dtype) {
    case 0:
      scatter_rows_kernel<float><<<n_scene, thr, 0, st>>>(n_scene, E, map_ws, count_out, scene_embeds,
                                                         (float *)inputs_embeds, scene_mask,
                                                         attention_mask);
      break;
    case 1:
      scatter_rows_kernel<__half><<<n_scene, thr, 0, st>>>(n_scene, E, map_ws, count_out, scene_embeds,
                                                          (__half *)inputs_embeds, scene_mask,
                                                          attention_mask);
      break;
    case 2:
      scatter_rows_kernel<unsigned short><<<n_scene, thr, 0, st>>>(
          n_scene, E, map_ws, count_out, scene_embeds, (unsigned short *)inputs_embeds, scene_mask,
          attention_mask);
      break;
    default:
      return MSR3D_EINVAL;
  }
  return (int)hipGetLastError();
}

}  // extern "C"
