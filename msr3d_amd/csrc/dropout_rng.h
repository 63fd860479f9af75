// dropout_rng.h -- counter-based dropout mask shared by the row kernels (rowops.hip, scene_rows.hip),
// the GEMM epilogues (gemm_f32.hip, strip_gemm.hip) and the scene blocks (scene_block.hip): a hash of
// (seed word on the device, call-site salt, element index), regenerated in backward, so a captured HIP
// graph draws a fresh mask on every replay (the seed word is bumped inside the graph) and no mask is
// stored.
//
// One murmur3 finaliser per PAIR of elements (index >> 1), 16 bits of it per element, compared with a
// 16-bit threshold: P(drop) = round(p * 65536) / 65536 (p = 0.1 -> 0.100006).  The mask generation is
// ~7 VALU operations per element instead of ~20 (two chained finalisers per element) -- the row-local
// chains of the spatial layer draw two masks per element and were bound by exactly this.
#pragma once
#include <hip/hip_runtime.h>

namespace msr3d {

__device__ __forceinline__ unsigned drop_mix32(unsigned h) {   // murmur3 finaliser
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool keep_elem(unsigned long long seed, unsigned salt, unsigned idx,
                                          unsigned thresh) {
  const unsigned c = drop_mix32((unsigned)seed ^ (salt * 0x7FEB352Du)) + (unsigned)(seed >> 32);   // wave-uniform
  const unsigned h = drop_mix32((idx >> 1) * 0x9E3779B1u + c);
  const unsigned bits = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
  return bits >= thresh;                // P(keep) = 1 - thresh / 2^16
}
__host__ __device__ __forceinline__ unsigned drop_thresh(float p) {
  return p > 0.f ? (unsigned)(p * 65536.0 + 0.5) : 0u;
}

}  // namespace msr3d
