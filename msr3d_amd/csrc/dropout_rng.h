// dropout_rng.h -- counter-based dropout mask shared by the row kernels (rowops.hip) and the
// GEMM epilogue (gemm_f32.hip): a hash of (seed word on the device, call-site salt, element
// index), regenerated in backward, so a captured HIP graph draws a fresh mask on every replay
// (the seed word is bumped inside the graph) and no mask is stored.
#pragma once
#include <hip/hip_runtime.h>

namespace msr3d {

__device__ __forceinline__ unsigned drop_mix32(unsigned h) {   // murmur3 finaliser
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool keep_elem(unsigned long long seed, unsigned salt, unsigned idx,
                                          unsigned thresh) {
  const unsigned h = drop_mix32(idx * 0x9E3779B1u + drop_mix32((unsigned)seed ^ (salt * 0x7FEB352Du)) +
                                (unsigned)(seed >> 32));
  return drop_mix32(h) >= thresh;       // P(keep) = 1 - thresh / 2^32
}
__host__ __device__ __forceinline__ unsigned drop_thresh(float p) {
  return p > 0.f ? (unsigned)(p * 4294967296.0) : 0u;
}

}  // namespace msr3d
