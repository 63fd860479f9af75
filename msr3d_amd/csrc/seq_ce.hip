// seq_ce.hip -- the language-model loss of MSR3D's training forward, fused:
//
//     logits = outputs.logits.float()
//     shift_logits = logits[..., :-1, :];  shift_labels = targets[..., 1:]
//     num_tokens_for_loss = (shift_labels >= 0).sum(1)
//     loss = F.cross_entropy(shift_logits, shift_labels, reduction='none').sum(1) / num_tokens_for_loss
//
// (/root/reference/model/msr3d/msr3d.py:426-441: the mean over each SEQUENCE's supervised tokens,
// not the model's own token-mean).  The reference materialises an fp32 copy of the (B, T, 32000)
// logits, the shifted copy, and log-softmax intermediates -- ~1 GB of HBM traffic per step at 4 x 576
// tokens; here the 16-bit logits are read ONCE per pass where they lie (the shift is an index), the
// row maximum / sum of exponentials are carried online in registers, and the backward writes the
// gradient in the logits' dtype.
//
//   forward   one workgroup per (b, t) row, t < T - 1: lse = logsumexp(row), tok_loss = lse - row[label]
//             (0 if label < 0); a second tiny launch sums each sequence's token losses in index order
//             (bit-reproducible) and divides by its count
//   backward  d row = (exp(row - lse) - onehot(label)) * g[b] / count[b]; rows without a target
//             (ignored label, and the last position of every sequence) get zeros
//
// HBM-bound byte work: algorithmic bytes = rows x V x sizeof(logit) (forward), twice that (backward).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace {

template <typename T> struct Ld;
template <> struct Ld<float> {
  static constexpr int VEC = 4;      // elements per 16-byte load
  __device__ static void load(const float *p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4 *>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
  __device__ static float one(const float *p) { return *p; }
  __device__ static void store(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ static void store1(float *p, float v) { *p = v; }
};
template <> struct Ld<__hip_bfloat16> {
  static constexpr int VEC = 8;
  __device__ static void load(const __hip_bfloat16 *p, float (&v)[8]) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(w[j] << 16);
      v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
    }
  }
  __device__ static float one(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
  __device__ static void store(__hip_bfloat16 *p, const float (&v)[8]) {
    __hip_bfloat16 h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = __float2bfloat16(v[j]);
    *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(h);
  }
  __device__ static void store1(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }
};
template <> struct Ld<__half> {
  static constexpr int VEC = 8;
  __device__ static void load(const __half *p, float (&v)[8]) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    const __half *h = reinterpret_cast<const __half *>(&q);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __half2float(h[j]);
  }
  __device__ static float one(const __half *p) { return __half2float(*p); }
  __device__ static void store(__half *p, const float (&v)[8]) {
    __half h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = __float2half(v[j]);
    *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(h);
  }
  __device__ static void store1(__half *p, float v) { *p = __float2half(v); }
};

// combine two (max, sum-of-exp) partials
__device__ __forceinline__ void lse_merge(float &m, float &s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
  m = mm;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(int T_len, int V, const T *__restrict__ logits,
                                                     const long long *__restrict__ targets,
                                                     float *__restrict__ lse_out,
                                                     float *__restrict__ tok_loss) {
  constexpr int VEC = Ld<T>::VEC;
  const int row = blockIdx.x;                        // (b, t), t < T - 1
  const int b = row / (T_len - 1), t = row - b * (T_len - 1);
  const long long label = targets[(size_t)b * T_len + t + 1];
  const int tid = threadIdx.x;
  if (label < 0 || label >= V) {                     // ignored position: nothing to read
    if (tid == 0) { tok_loss[row] = 0.f; lse_out[row] = 0.f; }
    return;
  }
  const T *x = logits + ((size_t)b * T_len + t) * V;
  float m = -INFINITY, s = 0.f;
  const int nvec = V / VEC;
  constexpr int U = 8;                               // 16-byte loads in flight per thread: half a 32000-wide bf16 row per pass
  for (int e0 = tid; e0 < nvec; e0 += 256 * U) {
    // unconditional loads from a clamped index, masked afterwards: a load inside `if (e < nvec)` is its own basic
    // block and the compiler waits vmcnt(0) behind each one -- the loads then run one after the other
    // (round 2: 1.6 TB/s with four round trips per row per pass)
    float v[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) Ld<T>::load(x + (size_t)min(e0 + u * 256, nvec - 1) * VEC, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u * 256 >= nvec) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[u][j] = -INFINITY;
      }
    }
    float mx = m;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < VEC; ++j) mx = fmaxf(mx, v[u][j]);
    if (mx > -INFINITY) {
      float acc = (m == -INFINITY) ? 0.f : s * __expf(m - mx);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += __expf(v[u][j] - mx);       // exp(-inf) = 0
      m = mx; s = acc;
    }
  }
  for (int e = nvec * VEC + tid; e < V; e += 256) {   // ragged tail (V % VEC != 0)
    const float v = Ld<T>::one(x + e);
    lse_merge(m, s, v, 1.f);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
    lse_merge(m, s, m2, s2);
  }
  __shared__ float sm[4], ss[4];
  if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = sm[0], S = ss[0];
    for (int w = 1; w < 4; ++w) lse_merge(M, S, sm[w], ss[w]);
    const float lse = M + logf(S);
    lse_out[row] = lse;
    tok_loss[row] = lse - Ld<T>::one(x + label);
  }
}

// loss[b] = (sum_t tok_loss[b][t]) / count[b], summed in index order: one workgroup per sequence
__global__ __launch_bounds__(256) void ce_reduce_kernel(int T_len, const long long *__restrict__ targets,
                                                        const float *__restrict__ tok_loss,
                                                        float *__restrict__ loss, int *__restrict__ count) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (T_len - 1 + 255) / 256;
  float s = 0.f;
  int c = 0;
  for (int t = tid * per; t < min(T_len - 1, (tid + 1) * per); ++t) {   // contiguous chunk per thread
    s += tok_loss[(size_t)b * (T_len - 1) + t];
    c += targets[(size_t)b * T_len + t + 1] >= 0 ? 1 : 0;
  }
  __shared__ float ps[256];
  __shared__ int pc[256];
  ps[tid] = s; pc[tid] = c;
  __syncthreads();
  if (tid == 0) {
    float S = 0.f; int C = 0;
    for (int j = 0; j < 256; ++j) { S += ps[j]; C += pc[j]; }            // fixed order
    count[b] = C;
    loss[b] = S / (float)C;                                              // 0/0 = NaN, as the reference
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(int T_len, int V, const T *__restrict__ logits,
                                                     const long long *__restrict__ targets,
                                                     const float *__restrict__ lse,
                                                     const int *__restrict__ count,
                                                     const float *__restrict__ g, T *__restrict__ dlogits) {
  constexpr int VEC = Ld<T>::VEC;
  const int b = blockIdx.x / T_len, t = blockIdx.x - b * T_len;          // ALL T rows: every one is written
  const int tid = threadIdx.x;
  T *dx = dlogits + ((size_t)b * T_len + t) * V;
  const long long label = t + 1 < T_len ? targets[(size_t)b * T_len + t + 1] : -1;
  const int nvec = V / VEC;
  if (label < 0 || label >= V) {
    float z[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) z[j] = 0.f;
    for (int e = tid; e < nvec; e += 256) Ld<T>::store(dx + (size_t)e * VEC, z);
    for (int e = nvec * VEC + tid; e < V; e += 256) Ld<T>::store1(dx + e, 0.f);
    return;
  }
  const T *x = logits + ((size_t)b * T_len + t) * V;
  const float l = lse[(size_t)b * (T_len - 1) + t];
  const float scale = g[b] / (float)count[b];
  for (int e0 = tid; e0 < nvec; e0 += 256 * 4) {     // 4 x 16-byte loads in flight per thread, then 4 stores
    float v[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) Ld<T>::load(x + (size_t)min(e0 + u * 256, nvec - 1) * VEC, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float p = __expf(v[u][j] - l);
        v[u][j] = (p - ((long long)e * VEC + j == label ? 1.f : 0.f)) * scale;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256;
      if (e < nvec) Ld<T>::store(dx + (size_t)e * VEC, v[u]);
    }
  }
  for (int e = nvec * VEC + tid; e < V; e += 256)
    Ld<T>::store1(dx + e, (__expf(Ld<T>::one(x + e) - l) - (e == label ? 1.f : 0.f)) * scale);
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

int msr3d_seq_ce_fwd(int B, int T, int V, const void *logits, int dtype, const long long *targets,
                     float *lse, float *tok_loss, float *loss, int *count, msr3d_stream_t stream) {
  if (B < 0 || T < 2 || V <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!logits || !targets || !lse || !tok_loss || !loss || !count || !al16(logits)) return MSR3D_EINVAL;
  const size_t esz = dtype == 0 ? 4 : 2;
  if (((size_t)V * esz) % 16 != 0) return MSR3D_EINVAL;        // every row starts 16-byte aligned
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * (T - 1);
  switch (dtype) {
    case 0: ce_fwd_kernel<float><<<rows, 256, 0, st>>>(T, V, (const float *)logits, targets, lse, tok_loss); break;
    case 1: ce_fwd_kernel<__half><<<rows, 256, 0, st>>>(T, V, (const __half *)logits, targets, lse, tok_loss); break;
    case 2: ce_fwd_kernel<__hip_bfloat16><<<rows, 256, 0, st>>>(T, V, (const __hip_bfloat16 *)logits, targets, lse, tok_loss); break;
    default: return MSR3D_EINVAL;
  }
  ce_reduce_kernel<<<B, 256, 0, st>>>(T, targets, tok_loss, loss, count);
  return (int)hipGetLastError();
}

int msr3d_seq_ce_bwd(int B, int T, int V, const void *logits, int dtype, const long long *targets,
                     const float *lse, const int *count, const float *grad_loss, void *dlogits,
                     msr3d_stream_t stream) {
  if (B < 0 || T < 2 || V <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!logits || !targets || !lse || !count || !grad_loss || !dlogits || !al16(logits) || !al16(dlogits))
    return MSR3D_EINVAL;
  const size_t esz = dtype == 0 ? 4 : 2;
  if (((size_t)V * esz) % 16 != 0) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * T;
  switch (dtype) {
    case 0: ce_bwd_kernel<float><<<rows, 256, 0, st>>>(T, V, (const float *)logits, targets, lse, count, grad_loss, (float *)dlogits); break;
    case 1: ce_bwd_kernel<__half><<<rows, 256, 0, st>>>(T, V, (const __half *)logits, targets, lse, count, grad_loss, (__half *)dlogits); break;
    case 2: ce_bwd_kernel<__hip_bfloat16><<<rows, 256, 0, st>>>(T, V, (const __hip_bfloat16 *)logits, targets, lse, count, grad_loss, (__hip_bfloat16 *)dlogits); break;
    default: return MSR3D_EINVAL;
  }
  return (int)hipGetLastError();
}

}  // extern "C"
