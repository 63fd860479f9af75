// llm_layer.hip -- the row-local and element-wise kernels of one LoRA-Llama decoder layer (SURVEY.md §8(f)
// rank 4; /root/reference/model/msr3d/msr3d.py:103-112, 409-415: Vicuna-7B under bf16 autocast with peft
// LoRA on q/k/v/o/gate/up/down_proj).  The layer, as transformers' LlamaDecoderLayer computes it:
//
//     h = RMSNorm(x) ; q, k, v = LoRALinear(h) ; RoPE(q, k) ; P = softmax(q k^T / sqrt(d) + causal + padding)
//     x = x + LoRALinear_o(P v) ; h = RMSNorm(x) ; x = x + LoRALinear_down( silu(LoRALinear_gate(h)) * LoRALinear_up(h) )
//
// The seven projections and the attention's per-head products are msr3d_bf16_gemm_lowrank / _batched
// (lora_linear.hip).  Here: RMSNorm (+ the residual add in front of it) forward / backward, rotary embedding
// (forward and its transpose), the causal + key-padding softmax forward / backward on fp32 scores, SwiGLU
// forward / backward, and a batched bf16 transpose (the backward products and P V need operands with the
// contraction index contiguous).  bf16 storage, fp32 arithmetic, one rounding per stored value.
// All HBM-bound: algorithmic bytes = the tensors read + written once.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace {

using u16 = unsigned short;

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ void unpack8(uint4 v, float (&f)[8]) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- RMSNorm: one wave per row, D <= 8192, D % 512 == 0 (a lane holds D / 64 values as D / 512 x 8) ------
// forward:  s = x (+ delta);  y = w * bf16( s * rsqrt(mean(s^2) + eps) )       (modeling_llama.LlamaRMSNorm)
template <int V>     // V = D / 512 16-byte vectors per lane
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(int M, const u16 *__restrict__ x, const u16 *__restrict__ delta,
                                                          const u16 *__restrict__ w, float eps, u16 *__restrict__ sum_out,
                                                          u16 *__restrict__ y, float *__restrict__ rstd_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  constexpr int D = V * 512;
  float s[V][8];
  float ss = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const size_t o = (size_t)row * D + (v * 64 + lane) * 8;
    unpack8(*reinterpret_cast<const uint4 *>(x + o), s[v]);
    if (delta) {
      float d[8];
      unpack8(*reinterpret_cast<const uint4 *>(delta + o), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[v][e] = bf2f(f2bf(s[v][e] + d[e]));      // the residual stream is stored in bf16
      if (sum_out) *reinterpret_cast<uint4 *>(sum_out + o) = pack8(s[v]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(s[v][e], s[v][e], ss);
  }
  const float rstd = rsqrtf(wave_sum(ss) * (1.0f / D) + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float g[8], o8[8];
    unpack8(*reinterpret_cast<const uint4 *>(w + (v * 64 + lane) * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = g[e] * bf2f(f2bf(s[v][e] * rstd));
    *reinterpret_cast<uint4 *>(y + (size_t)row * D + (v * 64 + lane) * 8) = pack8(o8);
  }
}
// backward (w frozen): g = dy * w;  dx = rstd * (g - xh * mean(g * xh)) (+ dres),  xh = s * rstd
template <int V>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(int M, const u16 *__restrict__ dy, const u16 *__restrict__ s_in,
                                                          const u16 *__restrict__ w, const float *__restrict__ rstd_in,
                                                          const u16 *__restrict__ dres, u16 *__restrict__ dx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  constexpr int D = V * 512;
  const float rstd = rstd_in[row];
  float g[V][8], xh[V][8];
  float c = 0.f;
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const size_t o = (size_t)row * D + (v * 64 + lane) * 8;
    float d[8], ww[8], s[8];
    unpack8(*reinterpret_cast<const uint4 *>(dy + o), d);
    unpack8(*reinterpret_cast<const uint4 *>(w + (v * 64 + lane) * 8), ww);
    unpack8(*reinterpret_cast<const uint4 *>(s_in + o), s);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      g[v][e] = d[e] * ww[e];
      xh[v][e] = s[e] * rstd;
      c = fmaf(g[v][e], xh[v][e], c);
    }
  }
  c = wave_sum(c) * (1.0f / D);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const size_t o = (size_t)row * D + (v * 64 + lane) * 8;
    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o8[8];
    if (dres) unpack8(*reinterpret_cast<const uint4 *>(dres + o), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = rstd * (g[v][e] - xh[v][e] * c) + r[e];
    *reinterpret_cast<uint4 *>(dx + o) = pack8(o8);
  }
}

// ---- rotary embedding, in place on (B, T, H, D) bf16; cos / sin (T, D) fp32 (position = t) ----------------
// x' = x cos + rotate_half(x) sin, rotate_half(x) = [-x2, x1] (modeling_llama.apply_rotary_pos_emb);
// sign = -1 applies the transpose (the backward): dx = dx' cos - rotate_half(dx' ... ) i.e. sin -> -sin.
__global__ __launch_bounds__(256) void rope_kernel(long long n_pairs, int T, int H, int D, u16 *__restrict__ x,
                                                   const float *__restrict__ cs, const float *__restrict__ sn, float sign) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;     // one (b, t, h, i < D/2) pair per thread
  if (p >= n_pairs) return;
  const int half = D >> 1;
  const int i = (int)(p % half);
  const long long bth = p / half;
  const int t = (int)((bth / H) % T);
  u16 *q = x + bth * D;
  const float a = bf2f(q[i]), b = bf2f(q[i + half]);
  const float c1 = cs[(size_t)t * D + i], c2 = cs[(size_t)t * D + i + half];
  const float s1 = sign * sn[(size_t)t * D + i], s2 = sign * sn[(size_t)t * D + i + half];
  q[i] = f2bf(a * c1 - b * s1);
  q[i + half] = f2bf(b * c2 + a * s2);
}
// D % 16 == 0, H % HG == 0: a thread owns eight adjacent pairs (two 16-byte pieces) of HG heads of one token -- the
// token's cos / sin values are loaded once and used HG times (the pair-per-thread kernel above reads 16 table bytes
// per 4 data bytes, 2-byte accesses: 15 us for 19 MB in place); up to two tensors (q and k) in one launch.
template <int HG>
__global__ __launch_bounds__(256) void rope_vec_kernel(long long n_items, int T, int H, int D, u16 *__restrict__ x0,
                                                       u16 *__restrict__ x1, const float *__restrict__ cs,
                                                       const float *__restrict__ sn, float sign) {
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  if (id >= n_items) return;
  const int half = D >> 1, npc = D >> 4, ngr = H / HG;
  const int c = (int)(id % npc);
  const int hg = (int)((id / npc) % ngr);
  const long long bt = id / ((long long)npc * ngr);
  const int t = (int)(bt % T), i0 = 8 * c;
  float c1[8], c2[8], s1[8], s2[8];
  {
    const float *cr = cs + (size_t)t * D + i0, *sr = sn + (size_t)t * D + i0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float4 a = *reinterpret_cast<const float4 *>(cr + 4 * v), b = *reinterpret_cast<const float4 *>(cr + half + 4 * v);
      const float4 e = *reinterpret_cast<const float4 *>(sr + 4 * v), f = *reinterpret_cast<const float4 *>(sr + half + 4 * v);
      c1[4 * v] = a.x; c1[4 * v + 1] = a.y; c1[4 * v + 2] = a.z; c1[4 * v + 3] = a.w;
      c2[4 * v] = b.x; c2[4 * v + 1] = b.y; c2[4 * v + 2] = b.z; c2[4 * v + 3] = b.w;
      s1[4 * v] = sign * e.x; s1[4 * v + 1] = sign * e.y; s1[4 * v + 2] = sign * e.z; s1[4 * v + 3] = sign * e.w;
      s2[4 * v] = sign * f.x; s2[4 * v + 1] = sign * f.y; s2[4 * v + 2] = sign * f.z; s2[4 * v + 3] = sign * f.w;
    }
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    u16 *x = which ? x1 : x0;
    if (!x) continue;
    u16 *row = x + (bt * H + (long long)hg * HG) * D + i0;
    uint4 va[HG], vb[HG];
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
      va[hh] = *reinterpret_cast<const uint4 *>(row + (size_t)hh * D);
      vb[hh] = *reinterpret_cast<const uint4 *>(row + (size_t)hh * D + half);
    }
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
      float a[8], b[8], ya[8], yb[8];
      unpack8(va[hh], a);
      unpack8(vb[hh], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) { ya[e] = a[e] * c1[e] - b[e] * s1[e]; yb[e] = b[e] * c2[e] + a[e] * s2[e]; }
      *reinterpret_cast<uint4 *>(row + (size_t)hh * D) = pack8(ya);
      *reinterpret_cast<uint4 *>(row + (size_t)hh * D + half) = pack8(yb);
    }
  }
}

// ---- causal + key-padding softmax over fp32 scores (B H, T, T) -> bf16 probabilities; one wave per row ------
// key t' of row t is visible iff t' <= t and keep[b][t'] != 0; a row with no visible key gives zeros.
// The row lives in registers: NV float4 per lane (T <= 256 NV), loaded unconditionally in one go, one pass for the
// maximum, one for the exponentials, 8-byte stores of four bf16 (the first version walked the row three times with
// 4-byte loads behind a branch per element: 109 us = 2.3 TB/s at 4 x 32 x 576 x 576).
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int BH, int H, int T, const float *__restrict__ S,
                                                          const unsigned char *__restrict__ keep, u16 *__restrict__ P) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)BH * T) return;
  const int t = (int)(row % T), b = (int)(row / T / H), n4 = T >> 2;
  const float4 *s = reinterpret_cast<const float4 *>(S + row * T);
  const unsigned *kp = keep ? reinterpret_cast<const unsigned *>(keep + (size_t)b * T) : nullptr;
  float4 v[NV];
  unsigned kk[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = min(lane + 64 * i, n4 - 1);
    v[i] = s[q];
    kk[i] = kp ? kp[q] : 0x01010101u;
  }
  float x[NV][4];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = 4 * (lane + 64 * i);
    const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool vis = lane + 64 * i < n4 && c0 + e <= t && ((kk[i] >> (8 * e)) & 0xffu);
      x[i][e] = vis ? f[e] : -INFINITY;
      m = fmaxf(m, x[i][e]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[i][e] = x[i][e] == -INFINITY ? 0.f : __expf(x[i][e] - m);
      z += x[i][e];
    }
  z = wave_sum(z);
  const float inv = z > 0.f ? 1.0f / z : 0.f;
  uint2 *p = reinterpret_cast<uint2 *>(P + row * T);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < n4)
      p[lane + 64 * i] = make_uint2(f2bf(x[i][0] * inv) | ((unsigned)f2bf(x[i][1] * inv) << 16),
                                    f2bf(x[i][2] * inv) | ((unsigned)f2bf(x[i][3] * inv) << 16));
}
// dS = P * (dP - sum_c dP P): dP fp32 (B H, T, T), P bf16 -> dS bf16; the row in registers as above
template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(long long rows, int T, const float *__restrict__ dP,
                                                          const u16 *__restrict__ P, u16 *__restrict__ dS) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int n4 = T >> 2;
  const float4 *d = reinterpret_cast<const float4 *>(dP + row * T);
  const uint2 *p = reinterpret_cast<const uint2 *>(P + row * T);
  float4 dv[NV];
  uint2 pv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = min(lane + 64 * i, n4 - 1);
    dv[i] = d[q];
    pv[i] = p[q];
  }
  float pr[NV][4], dd[NV][4];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool in = lane + 64 * i < n4;
    pr[i][0] = in ? bf2f((u16)(pv[i].x & 0xffffu)) : 0.f; pr[i][1] = in ? bf2f((u16)(pv[i].x >> 16)) : 0.f;
    pr[i][2] = in ? bf2f((u16)(pv[i].y & 0xffffu)) : 0.f; pr[i][3] = in ? bf2f((u16)(pv[i].y >> 16)) : 0.f;
    dd[i][0] = dv[i].x; dd[i][1] = dv[i].y; dd[i][2] = dv[i].z; dd[i][3] = dv[i].w;
#pragma unroll
    for (int e = 0; e < 4; ++e) dot = fmaf(dd[i][e], pr[i][e], dot);
  }
  dot = wave_sum(dot);
  uint2 *o = reinterpret_cast<uint2 *>(dS + row * T);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + 64 * i < n4)
      o[lane + 64 * i] = make_uint2(f2bf(pr[i][0] * (dd[i][0] - dot)) | ((unsigned)f2bf(pr[i][1] * (dd[i][1] - dot)) << 16),
                                    f2bf(pr[i][2] * (dd[i][2] - dot)) | ((unsigned)f2bf(pr[i][3] * (dd[i][3] - dot)) << 16));
}

// ---- SwiGLU: h = silu(gate) * up ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(long long n8, const uint4 *__restrict__ gate, const uint4 *__restrict__ up,
                                                         uint4 *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float g[8], u[8], o[8];
  unpack8(gate[i], g);
  unpack8(up[i], u);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sg = bf2f(f2bf(g[e] / (1.0f + expf(-g[e]))));      // silu is its own (bf16) op in the reference graph
    o[e] = sg * u[e];
  }
  out[i] = pack8(o);
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(long long n8, const uint4 *__restrict__ gate, const uint4 *__restrict__ up,
                                                         const uint4 *__restrict__ dh, uint4 *__restrict__ dgate,
                                                         uint4 *__restrict__ dup) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float g[8], u[8], d[8], og[8], ou[8];
  unpack8(gate[i], g);
  unpack8(up[i], u);
  unpack8(dh[i], d);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float sig = 1.0f / (1.0f + expf(-g[e]));
    const float sl = g[e] * sig;
    ou[e] = d[e] * sl;
    og[e] = d[e] * u[e] * (sig * (1.0f + g[e] * (1.0f - sig)));
  }
  dgate[i] = pack8(og);
  dup[i] = pack8(ou);
}

// ---- batched bf16 transpose: dst[b][c][r] = src[b][r][c]; 64 x 64 tiles through LDS; two-level batch -----------
// VEC: rows / cols multiples of 64 and both leading dimensions and bases 8-byte aligned: 8-byte loads and stores
// (four bf16), the transposition itself as 2-byte LDS reads (2-byte global accesses: 1.1 TB/s, a third of this).
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_kernel(int rows, int cols, int inner, const u16 *__restrict__ src, int lds_,
                                                        long long so, long long si, u16 *__restrict__ dst, int ldd,
                                                        long long dout, long long din) {
  __shared__ u16 tile[64][68];                         // (pitch 136 B: 8-byte row writes stay aligned)
  const int bz = blockIdx.z, bo = bz / inner, bi = bz - bo * inner;
  const u16 *s = src + bo * so + bi * si;
  u16 *d = dst + bo * dout + bi * din;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if constexpr (VEC) {
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;           // 16 column quads x 16 rows per pass
    uint2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint2 *>(s + (size_t)(r0 + rr + 16 * k) * lds_ + c0 + 4 * q);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2 *>(&tile[rr + 16 * k][4 * q]) = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = rr + 16 * k;                                      // output row = source column
      const unsigned lo = tile[4 * q][c] | ((unsigned)tile[4 * q + 1][c] << 16);
      const unsigned hi = tile[4 * q + 2][c] | ((unsigned)tile[4 * q + 3][c] << 16);
      *reinterpret_cast<uint2 *>(d + (size_t)(c0 + c) * ldd + r0 + 4 * q) = make_uint2(lo, hi);
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4)
      tile[r][tx] = (r0 + r < rows && c0 + tx < cols) ? s[(size_t)(r0 + r) * lds_ + c0 + tx] : (u16)0;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
      if (c0 + c < cols && r0 + tx < rows) d[(size_t)(c0 + c) * ldd + r0 + tx] = tile[tx][c];
  }
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

int msr3d_rmsnorm_fwd(int M, int D, const void *x, const void *delta, const void *w, float eps, void *sum_out,
                      void *y, float *rstd, msr3d_stream_t stream) {
  if (M < 0 || (D != 512 && D != 1024 && D != 2048 && D != 4096 && D != 5120 && D != 8192)) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!x || !w || !y || !al16(x) || !al16(w) || !al16(y) || (delta && !al16(delta)) || (sum_out && !al16(sum_out)))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int g = (M + 3) / 4;
#define MSR3D_RMS(V)                                                                                              \
  rmsnorm_fwd_kernel<V><<<g, 256, 0, st>>>(M, (const u16 *)x, (const u16 *)delta, (const u16 *)w, eps, (u16 *)sum_out, \
                                           (u16 *)y, rstd)
  switch (D / 512) {
    case 1: MSR3D_RMS(1); break;
    case 2: MSR3D_RMS(2); break;
    case 4: MSR3D_RMS(4); break;
    case 8: MSR3D_RMS(8); break;
    case 10: MSR3D_RMS(10); break;
    default: MSR3D_RMS(16); break;
  }
#undef MSR3D_RMS
  return (int)hipGetLastError();
}

int msr3d_rmsnorm_bwd(int M, int D, const void *dy, const void *s, const void *w, const float *rstd,
                      const void *dres, void *dx, msr3d_stream_t stream) {
  if (M < 0 || (D != 512 && D != 1024 && D != 2048 && D != 4096 && D != 5120 && D != 8192)) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!dy || !s || !w || !rstd || !dx || !al16(dy) || !al16(s) || !al16(w) || !al16(dx) || (dres && !al16(dres)))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int g = (M + 3) / 4;
#define MSR3D_RMSB(V)                                                                                          \
  rmsnorm_bwd_kernel<V><<<g, 256, 0, st>>>(M, (const u16 *)dy, (const u16 *)s, (const u16 *)w, rstd, (const u16 *)dres, \
                                           (u16 *)dx)
  switch (D / 512) {
    case 1: MSR3D_RMSB(1); break;
    case 2: MSR3D_RMSB(2); break;
    case 4: MSR3D_RMSB(4); break;
    case 8: MSR3D_RMSB(8); break;
    case 10: MSR3D_RMSB(10); break;
    default: MSR3D_RMSB(16); break;
  }
#undef MSR3D_RMSB
  return (int)hipGetLastError();
}

static int rope_launch(int B, int T, int H, int D, void *x0, void *x1, const float *cos_td, const float *sin_td, int transpose,
                       msr3d_stream_t stream) {
  if (B < 0 || T <= 0 || H <= 0 || D <= 0 || (D & 1)) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!x0 || !cos_td || !sin_td) return MSR3D_EINVAL;
  const float sign = transpose ? -1.0f : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (D % 16) == 0 && !(reinterpret_cast<uintptr_t>(x0) & 15u) && !(reinterpret_cast<uintptr_t>(x1) & 15u) &&
                   !(reinterpret_cast<uintptr_t>(cos_td) & 15u) && !(reinterpret_cast<uintptr_t>(sin_td) & 15u);
  if (vec) {
    const int hg = (H % 4) == 0 ? 4 : 1;
    const long long n = (long long)B * T * (H / hg) * (D / 16);
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (hg == 4) rope_vec_kernel<4><<<grid, 256, 0, st>>>(n, T, H, D, (u16 *)x0, (u16 *)x1, cos_td, sin_td, sign);
    else rope_vec_kernel<1><<<grid, 256, 0, st>>>(n, T, H, D, (u16 *)x0, (u16 *)x1, cos_td, sin_td, sign);
    return (int)hipGetLastError();
  }
  const long long n = (long long)B * T * H * (D / 2);
  rope_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, T, H, D, (u16 *)x0, cos_td, sin_td, sign);
  if (x1) rope_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, T, H, D, (u16 *)x1, cos_td, sin_td, sign);
  return (int)hipGetLastError();
}

int msr3d_rope_inplace(int B, int T, int H, int D, void *x, const float *cos_td, const float *sin_td, int transpose,
                       msr3d_stream_t stream) {
  return rope_launch(B, T, H, D, x, nullptr, cos_td, sin_td, transpose, stream);
}

int msr3d_rope_inplace2(int B, int T, int H, int D, void *x0, void *x1, const float *cos_td, const float *sin_td,
                        int transpose, msr3d_stream_t stream) {
  if (!x1) return MSR3D_EINVAL;
  return rope_launch(B, T, H, D, x0, x1, cos_td, sin_td, transpose, stream);
}

int msr3d_causal_softmax_fwd(int B, int H, int T, const float *scores, const unsigned char *key_keep, void *probs,
                             msr3d_stream_t stream) {
  if (B < 0 || H <= 0 || T <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!scores || !probs) return MSR3D_EINVAL;
  const long long rows = (long long)B * H * T;
  if ((T & 3) || T > 2048 || (reinterpret_cast<uintptr_t>(scores) & 15u) || (reinterpret_cast<uintptr_t>(probs) & 7u) ||
      (key_keep && (reinterpret_cast<uintptr_t>(key_keep) & 3u)))
    return MSR3D_EINVAL;
  const unsigned g = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  const int nv = (T / 4 + 63) / 64;
#define MSR3D_SM(NV) softmax_fwd_kernel<NV><<<g, 256, 0, st>>>(B * H, H, T, scores, key_keep, (u16 *)probs)
  if (nv <= 1) MSR3D_SM(1); else if (nv <= 2) MSR3D_SM(2); else if (nv <= 3) MSR3D_SM(3); else if (nv <= 4) MSR3D_SM(4);
  else MSR3D_SM(8);
#undef MSR3D_SM
  return (int)hipGetLastError();
}

int msr3d_causal_softmax_bwd(int B, int H, int T, const float *dprobs, const void *probs, void *dscores,
                             msr3d_stream_t stream) {
  if (B < 0 || H <= 0 || T <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!dprobs || !probs || !dscores) return MSR3D_EINVAL;
  const long long rows = (long long)B * H * T;
  if ((T & 3) || T > 2048 || (reinterpret_cast<uintptr_t>(dprobs) & 15u) || (reinterpret_cast<uintptr_t>(probs) & 7u) ||
      (reinterpret_cast<uintptr_t>(dscores) & 7u))
    return MSR3D_EINVAL;
  const unsigned g = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  const int nv = (T / 4 + 63) / 64;
#define MSR3D_SM(NV) softmax_bwd_kernel<NV><<<g, 256, 0, st>>>(rows, T, dprobs, (const u16 *)probs, (u16 *)dscores)
  if (nv <= 1) MSR3D_SM(1); else if (nv <= 2) MSR3D_SM(2); else if (nv <= 3) MSR3D_SM(3); else if (nv <= 4) MSR3D_SM(4);
  else MSR3D_SM(8);
#undef MSR3D_SM
  return (int)hipGetLastError();
}

int msr3d_swiglu_fwd(long long n, const void *gate, const void *up, void *out, msr3d_stream_t stream) {
  if (n < 0 || (n % 8)) return MSR3D_EINVAL;
  if (n == 0) return 0;
  if (!gate || !up || !out || !al16(gate) || !al16(up) || !al16(out)) return MSR3D_EINVAL;
  swiglu_fwd_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      n / 8, (const uint4 *)gate, (const uint4 *)up, (uint4 *)out);
  return (int)hipGetLastError();
}

int msr3d_swiglu_bwd(long long n, const void *gate, const void *up, const void *dh, void *dgate, void *dup,
                     msr3d_stream_t stream) {
  if (n < 0 || (n % 8)) return MSR3D_EINVAL;
  if (n == 0) return 0;
  if (!gate || !up || !dh || !dgate || !dup || !al16(gate) || !al16(up) || !al16(dh) || !al16(dgate) || !al16(dup))
    return MSR3D_EINVAL;
  swiglu_bwd_kernel<<<(unsigned)((n / 8 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      n / 8, (const uint4 *)gate, (const uint4 *)up, (const uint4 *)dh, (uint4 *)dgate, (uint4 *)dup);
  return (int)hipGetLastError();
}

int msr3d_transpose_bf16(int outer, int inner, int rows, int cols, const void *src, int ld_src, long long src_outer,
                         long long src_inner, void *dst, int ld_dst, long long dst_outer, long long dst_inner,
                         msr3d_stream_t stream) {
  if (outer < 0 || inner <= 0 || rows <= 0 || cols <= 0 || (long long)outer * inner > 65535) return MSR3D_EINVAL;
  if (outer == 0) return 0;
  if (!src || !dst || ld_src < cols || ld_dst < rows) return MSR3D_EINVAL;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64, outer * inner);
  const bool vec = !(rows % 64) && !(cols % 64) && !(ld_src % 4) && !(ld_dst % 4) && !(src_outer % 4) && !(src_inner % 4) &&
                   !(dst_outer % 4) && !(dst_inner % 4) && !(reinterpret_cast<uintptr_t>(src) & 7u) &&
                   !(reinterpret_cast<uintptr_t>(dst) & 7u);
  if (vec)
    transpose_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(rows, cols, inner, (const u16 *)src, ld_src, src_outer,
                                                                  src_inner, (u16 *)dst, ld_dst, dst_outer, dst_inner);
  else
    transpose_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(rows, cols, inner, (const u16 *)src, ld_src, src_outer,
                                                                   src_inner, (u16 *)dst, ld_dst, dst_outer, dst_inner);
  return (int)hipGetLastError();
}

}  // extern "C"
