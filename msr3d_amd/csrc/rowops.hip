// rowops.hip -- row-wise fused pieces of the spatial encoder layer:
//     y = LayerNorm( dropout(a) + r ) * gamma + beta          (forward)
// and its backward, one wave per row.  Three call sites per layer in the reference
// (/root/reference/modules/layers/transformers.py:250-251 attention tail, :324-325 and
// :326-328 layer tails) and the two embedding encoders (model/ose3d_situation.py:399-404,
// Linear -> LayerNorm, no residual, no dropout).  Replaces dropout + add + layer_norm
// (3 launches forward, 5 backward) by one launch each way.
//
// Dropout mask: counter-based hash of (seed word on the device, call-site salt, element
// index), regenerated in backward, so a captured HIP graph draws a fresh mask on every
// replay (the seed word is bumped inside the graph) and nothing but the pre-norm sum is
// saved.  Inverted-dropout scaling 1/(1-p) like torch.
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "dropout_rng.h"

namespace {

using msr3d::keep_elem;

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// D = 256 * VPL floats per row; lane holds VPL float4 at columns (j*64 + lane)*4
template <int VPL>
__global__ __launch_bounds__(256) void dal_fwd_kernel(int M, const float *__restrict__ a,
                                                      const float *__restrict__ r,
                                                      const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float eps,
                                                      float p_drop,
                                                      const unsigned long long *__restrict__ seed,
                                                      unsigned salt, float *__restrict__ y,
                                                      float *__restrict__ s_out,
                                                      float *__restrict__ stats) {
  constexpr int D = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bool drop = p_drop > 0.f;
  const unsigned thresh = msr3d::drop_thresh(drop ? p_drop : 0.f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned long long sd = drop ? *seed : 0ull;
  float4 v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    float4 x = *reinterpret_cast<const float4 *>(a + (size_t)row * D + c);
    if (drop) {
      const unsigned base = (unsigned)row * D + c;
      x.x = keep_elem(sd, salt, base + 0, thresh) ? x.x * scale : 0.f;
      x.y = keep_elem(sd, salt, base + 1, thresh) ? x.y * scale : 0.f;
      x.z = keep_elem(sd, salt, base + 2, thresh) ? x.z * scale : 0.f;
      x.w = keep_elem(sd, salt, base + 3, thresh) ? x.w * scale : 0.f;
    }
    if (r) {
      const float4 q = *reinterpret_cast<const float4 *>(r + (size_t)row * D + c);
      x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
    }
    v[j] = x;
    sum += (x.x + x.y) + (x.z + x.w);
  }
  const float mean = wave_sum(sum) * (1.0f / D);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
    var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  const float rstd = rsqrtf(wave_sum(var) * (1.0f / D) + eps);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 b = *reinterpret_cast<const float4 *>(beta + c);
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x + b.x;
    o.y = (v[j].y - mean) * rstd * g.y + b.y;
    o.z = (v[j].z - mean) * rstd * g.z + b.z;
    o.w = (v[j].w - mean) * rstd * g.w + b.w;
    *reinterpret_cast<float4 *>(y + (size_t)row * D + c) = o;
    if (s_out) *reinterpret_cast<float4 *>(s_out + (size_t)row * D + c) = v[j];
  }
  if (stats && lane == 0) { stats[row * 2 + 0] = mean; stats[row * 2 + 1] = rstd; }
}

// ROWS rows per block (4 waves x ROWS/4 iterations); dgamma/dbeta partials reduced in LDS,
// then one atomicAdd per column per block onto the (pre-zeroed or accumulating) destinations.
template <int VPL>
__global__ __launch_bounds__(256) void dal_bwd_kernel(int M, int rows_per_block,
                                                      const float *__restrict__ dy,
                                                      const float *__restrict__ s,
                                                      const float *__restrict__ stats,
                                                      const float *__restrict__ gamma, float p_drop,
                                                      const unsigned long long *__restrict__ seed,
                                                      unsigned salt, float *__restrict__ da,
                                                      float *__restrict__ dr, int dr_accumulate,
                                                      float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta,
                                                      float *__restrict__ partials) {
  constexpr int D = 256 * VPL;
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool drop = p_drop > 0.f;
  const unsigned thresh = msr3d::drop_thresh(drop ? p_drop : 0.f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned long long sd = drop ? *seed : 0ull;
  float4 gg[VPL], accg[VPL], accb[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    gg[j] = *reinterpret_cast<const float4 *>(gamma + (j * 64 + lane) * 4);
    accg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    accb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r0 = blockIdx.x * rows_per_block;
  for (int row = r0 + wave; row < min(M, r0 + rows_per_block); row += 4) {
    const float mean = stats[row * 2 + 0], rstd = stats[row * 2 + 1];
    float4 xh[VPL], g[VPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int c = (j * 64 + lane) * 4;
      const float4 d = *reinterpret_cast<const float4 *>(dy + (size_t)row * D + c);
      const float4 sv = *reinterpret_cast<const float4 *>(s + (size_t)row * D + c);
      xh[j].x = (sv.x - mean) * rstd; xh[j].y = (sv.y - mean) * rstd;
      xh[j].z = (sv.z - mean) * rstd; xh[j].w = (sv.w - mean) * rstd;
      g[j].x = d.x * gg[j].x; g[j].y = d.y * gg[j].y; g[j].z = d.z * gg[j].z; g[j].w = d.w * gg[j].w;
      c1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      c2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
      accg[j].x += d.x * xh[j].x; accg[j].y += d.y * xh[j].y;
      accg[j].z += d.z * xh[j].z; accg[j].w += d.w * xh[j].w;
      accb[j].x += d.x; accb[j].y += d.y; accb[j].z += d.z; accb[j].w += d.w;
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int c = (j * 64 + lane) * 4;
      float4 dx;
      dx.x = rstd * (g[j].x - c1 - xh[j].x * c2);
      dx.y = rstd * (g[j].y - c1 - xh[j].y * c2);
      dx.z = rstd * (g[j].z - c1 - xh[j].z * c2);
      dx.w = rstd * (g[j].w - c1 - xh[j].w * c2);
      if (dr) {
        float4 *o = reinterpret_cast<float4 *>(dr + (size_t)row * D + c);
        if (dr_accumulate) {           // the residual's gradient joins one that is already there
          const float4 e = *o;
          *o = make_float4(e.x + dx.x, e.y + dx.y, e.z + dx.z, e.w + dx.w);
        } else {
          *o = dx;
        }
      }
      if (da) {
        if (drop) {
          const unsigned base = (unsigned)row * D + c;
          dx.x = keep_elem(sd, salt, base + 0, thresh) ? dx.x * scale : 0.f;
          dx.y = keep_elem(sd, salt, base + 1, thresh) ? dx.y * scale : 0.f;
          dx.z = keep_elem(sd, salt, base + 2, thresh) ? dx.z * scale : 0.f;
          dx.w = keep_elem(sd, salt, base + 3, thresh) ? dx.w * scale : 0.f;
        }
        *reinterpret_cast<float4 *>(da + (size_t)row * D + c) = dx;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    *reinterpret_cast<float4 *>(&red[0][wave][c]) = accg[j];
    *reinterpret_cast<float4 *>(&red[1][wave][c]) = accb[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float sg = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float sb = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    if (partials) {        // ordered mode: per-block partials, summed in block order by reduce_partials
      partials[((size_t)blockIdx.x * 2 + 0) * D + c] = sg;
      partials[((size_t)blockIdx.x * 2 + 1) * D + c] = sb;
    } else {
      atomicAdd(dgamma + c, sg);
      atomicAdd(dbeta + c, sb);
    }
  }
}

// dst_k[c] += sum over blocks (ascending) of partials[block][k][c], k < NP: the bit-reproducible
// meeting point of the LayerNorm parameter gradients (the default one is atomicAdd)
struct PartialDst { float *p[4]; };
__global__ void reduce_partials_kernel(int n_blocks, int NP, int D, const float *__restrict__ partials,
                                       PartialDst dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  for (int k = 0; k < NP; ++k) {
    float acc = 0.f;
    for (int blk = 0; blk < n_blocks; ++blk) acc += partials[((size_t)blk * NP + k) * D + c];
    dst.p[k][c] += acc;
  }
}

// ---------------------------------------------------------------------------------------
// Two chained tails in one launch: t = LN2( drop2( LN1( drop1(a) + r ) ) + r ) -- the attention
// block's own residual + LayerNorm followed by the encoder layer's first one, which adds the SAME
// residual again (/root/reference/modules/layers/transformers.py:250-251 then :324-325).  Row-local,
// so forward and backward each need one pass over the row instead of two launches.
// ---------------------------------------------------------------------------------------
template <int VPL>
__device__ __forceinline__ void row_dropout(float4 (&v)[VPL], bool drop, unsigned long long sd,
                                            unsigned salt, unsigned thresh, float scale, int row,
                                            int lane) {
  if (!drop) return;
  constexpr int D = 256 * VPL;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const unsigned base = (unsigned)row * D + (j * 64 + lane) * 4;
    v[j].x = keep_elem(sd, salt, base + 0, thresh) ? v[j].x * scale : 0.f;
    v[j].y = keep_elem(sd, salt, base + 1, thresh) ? v[j].y * scale : 0.f;
    v[j].z = keep_elem(sd, salt, base + 2, thresh) ? v[j].z * scale : 0.f;
    v[j].w = keep_elem(sd, salt, base + 3, thresh) ? v[j].w * scale : 0.f;
  }
}

// v (the pre-norm sum, kept) -> y = LN(v) * gamma + beta; returns mean / rstd
template <int VPL>
__device__ __forceinline__ void row_layernorm(const float4 (&v)[VPL], const float *__restrict__ gamma,
                                              const float *__restrict__ beta, float eps, int lane,
                                              float4 (&y)[VPL], float &mean, float &rstd) {
  constexpr int D = 256 * VPL;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  mean = wave_sum(sum) * (1.0f / D);
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
    var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  rstd = rsqrtf(wave_sum(var) * (1.0f / D) + eps);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 b = *reinterpret_cast<const float4 *>(beta + c);
    y[j].x = (v[j].x - mean) * rstd * g.x + b.x;
    y[j].y = (v[j].y - mean) * rstd * g.y + b.y;
    y[j].z = (v[j].z - mean) * rstd * g.z + b.z;
    y[j].w = (v[j].w - mean) * rstd * g.w + b.w;
  }
}

template <int VPL>
__global__ __launch_bounds__(256) void dal2_fwd_kernel(
    int M, const float *__restrict__ a, const float *__restrict__ r, const float *__restrict__ g1,
    const float *__restrict__ b1, float eps1, float p1, unsigned salt1, const float *__restrict__ g2,
    const float *__restrict__ b2, float eps2, float p2, unsigned salt2,
    const unsigned long long *__restrict__ seed, float *__restrict__ y, float *__restrict__ s1_out,
    float *__restrict__ stats1, float *__restrict__ s2_out, float *__restrict__ stats2) {
  constexpr int D = 256 * VPL;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const unsigned long long sd = (p1 > 0.f || p2 > 0.f) ? *seed : 0ull;
  float4 v[VPL], res[VPL], mid[VPL], out[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    v[j] = *reinterpret_cast<const float4 *>(a + (size_t)row * D + c);
    res[j] = *reinterpret_cast<const float4 *>(r + (size_t)row * D + c);
  }
  row_dropout<VPL>(v, p1 > 0.f, sd, salt1, msr3d::drop_thresh(p1), p1 > 0.f ? 1.0f / (1.0f - p1) : 1.0f, row, lane);
#pragma unroll
  for (int j = 0; j < VPL; ++j) { v[j].x += res[j].x; v[j].y += res[j].y; v[j].z += res[j].z; v[j].w += res[j].w; }
  float mean, rstd;
  row_layernorm<VPL>(v, g1, b1, eps1, lane, mid, mean, rstd);
#pragma unroll
  for (int j = 0; j < VPL; ++j)
    *reinterpret_cast<float4 *>(s1_out + (size_t)row * D + (j * 64 + lane) * 4) = v[j];
  if (lane == 0) { stats1[row * 2 + 0] = mean; stats1[row * 2 + 1] = rstd; }
  row_dropout<VPL>(mid, p2 > 0.f, sd, salt2, msr3d::drop_thresh(p2), p2 > 0.f ? 1.0f / (1.0f - p2) : 1.0f, row, lane);
#pragma unroll
  for (int j = 0; j < VPL; ++j) { mid[j].x += res[j].x; mid[j].y += res[j].y; mid[j].z += res[j].z; mid[j].w += res[j].w; }
  row_layernorm<VPL>(mid, g2, b2, eps2, lane, out, mean, rstd);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    *reinterpret_cast<float4 *>(s2_out + (size_t)row * D + c) = mid[j];
    *reinterpret_cast<float4 *>(y + (size_t)row * D + c) = out[j];
  }
  if (lane == 0) { stats2[row * 2 + 0] = mean; stats2[row * 2 + 1] = rstd; }
}

// dx of one LayerNorm row; accumulates this row's dgamma / dbeta contributions
template <int VPL>
__device__ __forceinline__ void row_layernorm_bwd(const float4 (&d)[VPL], const float *__restrict__ s,
                                                  float mean, float rstd, const float4 (&gg)[VPL],
                                                  int row, int lane, float4 (&dx)[VPL],
                                                  float4 (&accg)[VPL], float4 (&accb)[VPL]) {
  constexpr int D = 256 * VPL;
  float4 xh[VPL], g[VPL];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float4 sv = *reinterpret_cast<const float4 *>(s + (size_t)row * D + (j * 64 + lane) * 4);
    xh[j].x = (sv.x - mean) * rstd; xh[j].y = (sv.y - mean) * rstd;
    xh[j].z = (sv.z - mean) * rstd; xh[j].w = (sv.w - mean) * rstd;
    g[j].x = d[j].x * gg[j].x; g[j].y = d[j].y * gg[j].y; g[j].z = d[j].z * gg[j].z; g[j].w = d[j].w * gg[j].w;
    c1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
    c2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    accg[j].x += d[j].x * xh[j].x; accg[j].y += d[j].y * xh[j].y;
    accg[j].z += d[j].z * xh[j].z; accg[j].w += d[j].w * xh[j].w;
    accb[j].x += d[j].x; accb[j].y += d[j].y; accb[j].z += d[j].z; accb[j].w += d[j].w;
  }
  c1 = wave_sum(c1) * (1.0f / D);
  c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    dx[j].x = rstd * (g[j].x - c1 - xh[j].x * c2);
    dx[j].y = rstd * (g[j].y - c1 - xh[j].y * c2);
    dx[j].z = rstd * (g[j].z - c1 - xh[j].z * c2);
    dx[j].w = rstd * (g[j].w - c1 - xh[j].w * c2);
  }
}

// dy = d t;  outputs da (gradient of `a`, both masks applied in turn) and dr (the residual's
// gradient: both tails feed it); dgamma / dbeta of both LayerNorms are accumulated into.
template <int VPL>
__global__ __launch_bounds__(256) void dal2_bwd_kernel(
    int M, int rows_per_block, const float *__restrict__ dy, const float *__restrict__ s1,
    const float *__restrict__ stats1, const float *__restrict__ g1, float p1, unsigned salt1,
    const float *__restrict__ s2, const float *__restrict__ stats2, const float *__restrict__ g2,
    float p2, unsigned salt2, const unsigned long long *__restrict__ seed, float *__restrict__ da,
    float *__restrict__ dr, float *__restrict__ dgamma1, float *__restrict__ dbeta1,
    float *__restrict__ dgamma2, float *__restrict__ dbeta2, float *__restrict__ partials) {
  constexpr int D = 256 * VPL;
  __shared__ float red[4][4][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long sd = (p1 > 0.f || p2 > 0.f) ? *seed : 0ull;
  float4 gg1[VPL], gg2[VPL], ag1[VPL], ab1[VPL], ag2[VPL], ab2[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    gg1[j] = *reinterpret_cast<const float4 *>(g1 + (j * 64 + lane) * 4);
    gg2[j] = *reinterpret_cast<const float4 *>(g2 + (j * 64 + lane) * 4);
    ag1[j] = ab1[j] = ag2[j] = ab2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r0 = blockIdx.x * rows_per_block;
  for (int row = r0 + wave; row < min(M, r0 + rows_per_block); row += 4) {
    float4 d[VPL], dx2[VPL], dx1[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      d[j] = *reinterpret_cast<const float4 *>(dy + (size_t)row * D + (j * 64 + lane) * 4);
    row_layernorm_bwd<VPL>(d, s2, stats2[row * 2], stats2[row * 2 + 1], gg2, row, lane, dx2, ag2, ab2);
#pragma unroll
    for (int j = 0; j < VPL; ++j) d[j] = dx2[j];
    row_dropout<VPL>(d, p2 > 0.f, sd, salt2, msr3d::drop_thresh(p2), p2 > 0.f ? 1.0f / (1.0f - p2) : 1.0f, row, lane);
    row_layernorm_bwd<VPL>(d, s1, stats1[row * 2], stats1[row * 2 + 1], gg1, row, lane, dx1, ag1, ab1);
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int c = (j * 64 + lane) * 4;
      *reinterpret_cast<float4 *>(dr + (size_t)row * D + c) =
          make_float4(dx2[j].x + dx1[j].x, dx2[j].y + dx1[j].y, dx2[j].z + dx1[j].z, dx2[j].w + dx1[j].w);
    }
    row_dropout<VPL>(dx1, p1 > 0.f, sd, salt1, msr3d::drop_thresh(p1), p1 > 0.f ? 1.0f / (1.0f - p1) : 1.0f, row, lane);
#pragma unroll
    for (int j = 0; j < VPL; ++j)
      *reinterpret_cast<float4 *>(da + (size_t)row * D + (j * 64 + lane) * 4) = dx1[j];
  }
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (j * 64 + lane) * 4;
    *reinterpret_cast<float4 *>(&red[0][wave][c]) = ag1[j];
    *reinterpret_cast<float4 *>(&red[1][wave][c]) = ab1[j];
    *reinterpret_cast<float4 *>(&red[2][wave][c]) = ag2[j];
    *reinterpret_cast<float4 *>(&red[3][wave][c]) = ab2[j];
  }
  __syncthreads();
  float *const dsts[4] = {dgamma1, dbeta1, dgamma2, dbeta2};
  for (int c = threadIdx.x; c < D; c += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sum = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
      if (partials) partials[((size_t)blockIdx.x * 4 + k) * D + c] = sum;
      else atomicAdd(dsts[k] + c, sum);
    }
  }
}

__global__ void bump_seed_kernel(unsigned long long *seed) { *seed = *seed * 6364136223846793005ull + 1442695040888963407ull; }

}  // namespace

extern "C" {

int msr3d_dropout_add_ln_fwd(int M, int D, const float *a, const float *r, const float *gamma,
                             const float *beta, float eps, float p_drop,
                             const unsigned long long *seed, unsigned salt, float *y, float *s_out,
                             float *stats, msr3d_stream_t stream) {
  if (M < 0 || (D != 256 && D != 512 && D != 768 && D != 1024)) return MSR3D_EINVAL;
  if (p_drop < 0.f || p_drop >= 1.f) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!a || !gamma || !beta || !y || (p_drop > 0.f && !seed)) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (M + 3) / 4;
#define L(V) dal_fwd_kernel<V><<<grid, 256, 0, st>>>(M, a, r, gamma, beta, eps, p_drop, seed, salt, y, s_out, stats)
  switch (D / 256) { case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; default: L(4); }
#undef L
  return (int)hipGetLastError();
}

int msr3d_dropout_add_ln_bwd(int M, int D, const float *dy, const float *s, const float *stats,
                             const float *gamma, float p_drop, const unsigned long long *seed,
                             unsigned salt, float *da, float *dr, int dr_accumulate,
                             float *dgamma_acc, float *dbeta_acc, float *partial_ws,
                             msr3d_stream_t stream) {
  if (M < 0 || (D != 256 && D != 512 && D != 768 && D != 1024)) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!dy || !s || !stats || !gamma || !dgamma_acc || !dbeta_acc || (p_drop > 0.f && !seed))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = MSR3D_LN_BWD_ROWS;
  const int grid = (M + rpb - 1) / rpb;
#define L(V) dal_bwd_kernel<V><<<grid, 256, 0, st>>>(M, rpb, dy, s, stats, gamma, p_drop, seed, salt, da, dr, dr_accumulate, dgamma_acc, dbeta_acc, partial_ws)
  switch (D / 256) { case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; default: L(4); }
#undef L
  if (partial_ws) {
    PartialDst dst = {{dgamma_acc, dbeta_acc, nullptr, nullptr}};
    reduce_partials_kernel<<<(D + 255) / 256, 256, 0, st>>>(grid, 2, D, partial_ws, dst);
  }
  return (int)hipGetLastError();
}

int msr3d_dropout_add_ln2_fwd(int M, int D, const float *a, const float *r, const float *gamma1,
                              const float *beta1, float eps1, float p1, unsigned salt1,
                              const float *gamma2, const float *beta2, float eps2, float p2,
                              unsigned salt2, const unsigned long long *seed, float *y, float *s1,
                              float *stats1, float *s2, float *stats2, msr3d_stream_t stream) {
  if (M < 0 || (D != 256 && D != 512)) return MSR3D_EINVAL;     // red[][] of the backward: D <= 512
  if (M == 0) return 0;
  if (!a || !r || !gamma1 || !beta1 || !gamma2 || !beta2 || !y || !s1 || !stats1 || !s2 || !stats2 ||
      ((p1 > 0.f || p2 > 0.f) && !seed) || p1 >= 1.f || p2 >= 1.f)
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (M + 3) / 4;
#define L(V) dal2_fwd_kernel<V><<<grid, 256, 0, st>>>(M, a, r, gamma1, beta1, eps1, p1, salt1, gamma2, beta2, eps2, p2, salt2, seed, y, s1, stats1, s2, stats2)
  if (D == 256) L(1); else L(2);
#undef L
  return (int)hipGetLastError();
}

int msr3d_dropout_add_ln2_bwd(int M, int D, const float *dy, const float *s1, const float *stats1,
                              const float *gamma1, float p1, unsigned salt1, const float *s2,
                              const float *stats2, const float *gamma2, float p2, unsigned salt2,
                              const unsigned long long *seed, float *da, float *dr,
                              float *dgamma1_acc, float *dbeta1_acc, float *dgamma2_acc,
                              float *dbeta2_acc, float *partial_ws, msr3d_stream_t stream) {
  if (M < 0 || (D != 256 && D != 512)) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!dy || !s1 || !stats1 || !gamma1 || !s2 || !stats2 || !gamma2 || !da || !dr || !dgamma1_acc ||
      !dbeta1_acc || !dgamma2_acc || !dbeta2_acc || ((p1 > 0.f || p2 > 0.f) && !seed))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rpb = MSR3D_LN_BWD_ROWS;
  const int grid = (M + rpb - 1) / rpb;
#define L(V) dal2_bwd_kernel<V><<<grid, 256, 0, st>>>(M, rpb, dy, s1, stats1, gamma1, p1, salt1, s2, stats2, gamma2, p2, salt2, seed, da, dr, dgamma1_acc, dbeta1_acc, dgamma2_acc, dbeta2_acc, partial_ws)
  if (D == 256) L(1); else L(2);
#undef L
  if (partial_ws) {
    PartialDst dst = {{dgamma1_acc, dbeta1_acc, dgamma2_acc, dbeta2_acc}};
    reduce_partials_kernel<<<(D + 255) / 256, 256, 0, st>>>(grid, 4, D, partial_ws, dst);
  }
  return (int)hipGetLastError();
}

int msr3d_bump_seed(unsigned long long *seed, msr3d_stream_t stream) {
  if (!seed) return MSR3D_EINVAL;
  bump_seed_kernel<<<1, 1, 0, (hipStream_t)stream>>>(seed);
  return (int)hipGetLastError();
}

}  // extern "C"
