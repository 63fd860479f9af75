// colsum.h -- dst[c] += sum_{i < n} part[i * 256 + c], c < 256, in row order: the second half of the LayerNorm
// parameter gradients (include/msr3d_hip.h: msr3d_colsum_partials).  A device function so that the jobs can ride as extra
// workgroups of another launch (msr3d_wgrad_split_colsum) as well as in their own.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"

namespace msr3d {

// threads 0..255 of the workgroup work (thread = (float4 column q, row group rg): rows rg, rg + 4, ... in batches of 16
// loads in flight; the four row groups meet in `red` ([4][256] floats of LDS) in a fixed order); EVERY thread of the
// workgroup must call it (one barrier inside).
__device__ __forceinline__ void colsum_job(const msr3d_colsum_job_t &j, float *red) {
  constexpr int D = 256;
  const bool work = threadIdx.x < 256;
  const int q = threadIdx.x & 63, rg = (threadIdx.x >> 6) & 3;
  if (work) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 16;
    for (int i0 = rg; i0 < j.n; i0 += 4 * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 4 * u;
        v[u] = *reinterpret_cast<const float4 *>(j.part + (size_t)min(i, j.n - 1) * D + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + 4 * u < j.n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    *reinterpret_cast<float4 *>(red + rg * D + 4 * q) = s;
  }
  __syncthreads();
  if (work && rg == 0) {
    const float4 a = *reinterpret_cast<const float4 *>(red + 4 * q), b = *reinterpret_cast<const float4 *>(red + D + 4 * q),
                 c = *reinterpret_cast<const float4 *>(red + 2 * D + 4 * q), d = *reinterpret_cast<const float4 *>(red + 3 * D + 4 * q);
    float4 t = *reinterpret_cast<const float4 *>(j.dst + 4 * q);
    t.x += (a.x + b.x) + (c.x + d.x); t.y += (a.y + b.y) + (c.y + d.y);
    t.z += (a.z + b.z) + (c.z + d.z); t.w += (a.w + b.w) + (c.w + d.w);
    *reinterpret_cast<float4 *>(j.dst + 4 * q) = t;
  }
}

}  // namespace msr3d
