// rowmath.h -- one token row (D = 256 floats) held by one wave, lane l owning columns 4l..4l+3:
// the dropout / residual / LayerNorm arithmetic of the spatial encoder layer
// (/root/reference/modules/layers/transformers.py:250-251,324-328) in the exact operation order of
// rowops.hip's kernels (dal_fwd / dal2_fwd / dal_bwd / dal2_bwd), so that a kernel which applies it
// while staging a GEMM operand (strip_gemm.hip) produces the same bits as the stand-alone row kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "dropout_rng.h"

namespace msr3d {

constexpr int ROW_D = 256;

__device__ __forceinline__ float row_wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// inverted dropout on the row's four elements; mask index = row * 256 + column (as rowops.hip)
__device__ __forceinline__ float4 row_drop(float4 v, bool drop, unsigned long long sd, unsigned salt,
                                           unsigned thresh, float scale, int row, int lane) {
  if (!drop) return v;
  const unsigned base = (unsigned)row * ROW_D + lane * 4;
  v.x = keep_elem(sd, salt, base + 0, thresh) ? v.x * scale : 0.f;
  v.y = keep_elem(sd, salt, base + 1, thresh) ? v.y * scale : 0.f;
  v.z = keep_elem(sd, salt, base + 2, thresh) ? v.z * scale : 0.f;
  v.w = keep_elem(sd, salt, base + 3, thresh) ? v.w * scale : 0.f;
  return v;
}

// y = LN(v) * gamma + beta over the 256 columns spread across the wave
__device__ __forceinline__ float4 row_ln(float4 v, float4 g, float4 b, float eps, float &mean,
                                         float &rstd) {
  mean = row_wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / ROW_D);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  rstd = rsqrtf(row_wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / ROW_D) + eps);
  return make_float4(dx * rstd * g.x + b.x, dy * rstd * g.y + b.y, dz * rstd * g.z + b.z,
                     dw * rstd * g.w + b.w);
}

// LayerNorm backward of one row: d = upstream gradient, s = the saved pre-norm sum; returns dx and
// adds this row's contributions to the gamma / beta gradient accumulators
__device__ __forceinline__ float4 row_ln_bwd(float4 d, float4 s, float mean, float rstd, float4 gg,
                                             float4 &accg, float4 &accb) {
  const float4 xh = make_float4((s.x - mean) * rstd, (s.y - mean) * rstd, (s.z - mean) * rstd,
                                (s.w - mean) * rstd);
  const float4 g = make_float4(d.x * gg.x, d.y * gg.y, d.z * gg.z, d.w * gg.w);
  float c1 = (g.x + g.y) + (g.z + g.w);
  float c2 = (g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w);
  accg.x += d.x * xh.x; accg.y += d.y * xh.y; accg.z += d.z * xh.z; accg.w += d.w * xh.w;
  accb.x += d.x; accb.y += d.y; accb.z += d.z; accb.w += d.w;
  c1 = row_wave_sum(c1) * (1.0f / ROW_D);
  c2 = row_wave_sum(c2) * (1.0f / ROW_D);
  return make_float4(rstd * (g.x - c1 - xh.x * c2), rstd * (g.y - c1 - xh.y * c2),
                     rstd * (g.z - c1 - xh.z * c2), rstd * (g.w - c1 - xh.w * c2));
}

__device__ __forceinline__ float gelu_exact(float x) {     // erf form (F.gelu default)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_exact_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) +
         x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

}  // namespace msr3d
