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

#include <cstdint>

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


// ---------------------------------------------------------------------------------------
// Fused project-and-scatter: inputs_embeds[dest[m]] = cast(tokens[m] W^T + b)
// without materialising the fp32 (n_scene, E) projector output.  The product runs on bf16 MFMA
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate) -- the output is cast to the LLM's 16-bit embedding
// dtype anyway; tokens and weights are read as fp32 and rounded (RNE) on the way into LDS.
// Workgroup tile 64 rows x 128 columns, 4 waves of 64 x 32, K in slabs of 32 (one MFMA step),
// register-prefetched double buffer.  HBM-bound: 2 * n_scene * E bytes written, W read once
// per 64-row band through L2.
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4v = __attribute__((ext_vector_type(4))) float;

constexpr int PS_BM = 64, PS_BN = 128, PS_BK = 32;
constexpr int PS_LD = PS_BK + 8;          // bf16 elements per LDS row (80 B: 16-B aligned fragments)

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return (unsigned)cvt<unsigned short>(a) | ((unsigned)cvt<unsigned short>(b) << 16);
}

// rows x 32 fp32 (row-major, ld) -> registers: ROWS/32 float4 per thread (thread t: row t/8 + 32p, k4 = (t%8)*4)
template <int ROWS>
__device__ __forceinline__ void ps_load(const float *__restrict__ P, int ld, int r0, int k0, int R,
                                        float4 (&v)[ROWS / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < ROWS / 32; ++p) {
    const int row = r0 + (t >> 3) + 32 * p;
    v[p] = row < R ? *reinterpret_cast<const float4 *>(P + (size_t)row * ld + k0 + (t & 7) * 4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int ROWS>
__device__ __forceinline__ void ps_store(unsigned short *s, const float4 (&v)[ROWS / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < ROWS / 32; ++p) {
    uint2 w;
    w.x = pack_bf16x2(v[p].x, v[p].y);
    w.y = pack_bf16x2(v[p].z, v[p].w);
    *reinterpret_cast<uint2 *>(s + ((t >> 3) + 32 * p) * PS_LD + (t & 7) * 4) = w;
  }
}

template <typename OutT>
__global__ __launch_bounds__(256) void proj_scatter_kernel(
    int cap, int E, int K, const float *__restrict__ tokens, const float *__restrict__ W,
    const float *__restrict__ bias, const int *__restrict__ dest, const int *__restrict__ count,
    OutT *__restrict__ embeds, const unsigned char *__restrict__ scene_mask,
    long long *__restrict__ attn_mask) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[2][PS_BM * PS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short sB[2][PS_BN * PS_LD];
  __shared__ __attribute__((aligned(16))) OutT sC[PS_BM * PS_BN];
  const int n = min(*count, cap);
  const int m0 = blockIdx.y * PS_BM, n0 = blockIdx.x * PS_BN;
  if (m0 >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;

  f32x4v acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

  float4 ra[PS_BM / 32], rb[PS_BN / 32];
  ps_load<PS_BM>(tokens, K, m0, 0, n, ra);
  ps_load<PS_BN>(W, K, n0, 0, E, rb);
  ps_store<PS_BM>(sA[0], ra);
  ps_store<PS_BN>(sB[0], rb);
  __syncthreads();
  const int nk = K / PS_BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      ps_load<PS_BM>(tokens, K, m0, (kt + 1) * PS_BK, n, ra);
      ps_load<PS_BN>(W, K, n0, (kt + 1) * PS_BK, E, rb);
    }
    // lane (i, g) supplies A[row i][k = 8g .. 8g+7] and B[k = 8g .. 8g+7][col i]
    bf16x8 fa[4], fb[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      fa[rt] = *reinterpret_cast<const bf16x8 *>(&sA[cur][(rt * 16 + i) * PS_LD + g * 8]);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
      fb[ct] = *reinterpret_cast<const bf16x8 *>(&sB[cur][(wave * 32 + ct * 16 + i) * PS_LD + g * 8]);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    if (kt + 1 < nk) {
      ps_store<PS_BM>(sA[cur ^ 1], ra);
      ps_store<PS_BN>(sB[cur ^ 1], rb);
    }
    __syncthreads();
  }

  // epilogue: + bias, cast, through LDS so that rows leave as whole 16-byte vectors
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = wave * 32 + ct * 16 + i;
    const float bv = bias ? bias[n0 + col] : 0.f;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(rt * 16 + g * 4 + r) * PS_BN + col] = cvt<OutT>(acc[rt][ct][r] + bv);
  }
  __syncthreads();
  constexpr int VPR = PS_BN * (int)sizeof(OutT) / 16;      // 16-byte vectors per row
  for (int e = threadIdx.x; e < PS_BM * VPR; e += 256) {
    const int row = e / VPR, v = e % VPR;
    const int m = m0 + row;
    if (m >= n) continue;
    const size_t drow = (size_t)dest[m];
    const uint4 val = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sC) +
                                                       ((size_t)row * PS_BN * sizeof(OutT) + v * 16));
    *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(embeds + drow * E + n0) + v * 16) = val;
  }
  if (blockIdx.x == 0 && attn_mask && scene_mask && threadIdx.x < PS_BM && m0 + (int)threadIdx.x < n)
    attn_mask[dest[m0 + threadIdx.x]] = scene_mask[m0 + threadIdx.x] ? 1 : 0;
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

int msr3d_project_scatter_bf16(int B, int T, int n_scene, int E, int K, const long long *input_ids,
                               long long scene_token, const float *tokens, const float *weight,
                               const float *bias, const unsigned char *scene_mask, int out_dtype,
                               void *inputs_embeds, long long *attention_mask, int *map_ws,
                               int *count_out, msr3d_stream_t stream) {
  if (B < 0 || T <= 0 || n_scene < 0 || E <= 0 || K <= 0 || E % PS_BN != 0 || K % PS_BK != 0)
    return MSR3D_EINVAL;
  if (B == 0 || n_scene == 0) return 0;
  if (!input_ids || !tokens || !weight || !inputs_embeds || !map_ws || !count_out) return MSR3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(tokens) | reinterpret_cast<uintptr_t>(weight) |
       reinterpret_cast<uintptr_t>(inputs_embeds)) & 15u)
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  scatter_map_kernel<<<1, 256, 0, st>>>(B, T, input_ids, scene_token, n_scene, map_ws, count_out);
  const dim3 grid(E / PS_BN, (n_scene + PS_BM - 1) / PS_BM);
  switch (out_dtype) {
    case 0:
      proj_scatter_kernel<float><<<grid, 256, 0, st>>>(n_scene, E, K, tokens, weight, bias, map_ws,
                                                        count_out, (float *)inputs_embeds, scene_mask,
                                                        attention_mask);
      break;
    case 1:
      proj_scatter_kernel<__half><<<grid, 256, 0, st>>>(n_scene, E, K, tokens, weight, bias, map_ws,
                                                         count_out, (__half *)inputs_embeds, scene_mask,
                                                         attention_mask);
      break;
    case 2:
      proj_scatter_kernel<unsigned short><<<grid, 256, 0, st>>>(
          n_scene, E, K, tokens, weight, bias, map_ws, count_out, (unsigned short *)inputs_embeds,
          scene_mask, attention_mask);
      break;
    default:
      return MSR3D_EINVAL;
  }
  return (int)hipGetLastError();
}

}  // extern "C"
