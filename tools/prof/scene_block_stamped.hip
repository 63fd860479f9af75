// Profiling build of csrc/scene_block.hip: every wave records s_memtime at the kernel's phase marks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I include -shared \
//         tools/prof/scene_block_stamped.hip -o tools/_prof/libsb_stamped.so
// Exports msr3d_scene_block / msr3d_split_pack (same ABI) plus msr3d_prof_stamps(host_buffer).
#include <hip/hip_runtime.h>

#define SB_NSTAMP 16
#define SB_MAXWAVES 4096
__device__ unsigned long long g_sb_stamps[SB_MAXWAVES * SB_NSTAMP];
#define SB_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      const unsigned w_ = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);                 \
      if (w_ < SB_MAXWAVES) g_sb_stamps[w_ * SB_NSTAMP + (i)] = __builtin_readcyclecounter();              \
    }                                                                                                      \
  } while (0)

#include "../../msr3d_amd/csrc/scene_block.hip"

extern "C" int msr3d_prof_stamps(unsigned long long *host, int clear) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sb_stamps), sizeof(unsigned long long) * SB_MAXWAVES * SB_NSTAMP);
  if (e != hipSuccess) return (int)e;
  if (clear) {
    static unsigned long long z[SB_MAXWAVES * SB_NSTAMP];
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_sb_stamps), z, sizeof(z));
  }
  return (int)e;
}
