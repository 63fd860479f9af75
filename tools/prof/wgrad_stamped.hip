// Profiling build of csrc/wgrad_split.hip: every wave records the cycle counter at the kernel's marks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I include -shared \
//         tools/prof/wgrad_stamped.hip -o tools/_prof/libwg_stamped.so
#include <hip/hip_runtime.h>

#define WG_NSTAMP 8
#define WG_MAXWAVES 8192
__device__ unsigned long long g_wg_stamps[WG_MAXWAVES * WG_NSTAMP];
#define WG_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      const unsigned w_ = blockIdx.x * 16 + (threadIdx.x >> 6);                                             \
      if (w_ < WG_MAXWAVES) g_wg_stamps[w_ * WG_NSTAMP + (i)] = __builtin_readcyclecounter();              \
    }                                                                                                      \
  } while (0)

#define WG_CLOCK() __builtin_readcyclecounter()
#define WG_PUT(i, v)                                                                                       \
  do {                                                                                                     \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      const unsigned w_ = blockIdx.x * 16 + (threadIdx.x >> 6);                                             \
      if (w_ < WG_MAXWAVES) g_wg_stamps[w_ * WG_NSTAMP + (i)] = (v);                                       \
    }                                                                                                      \
  } while (0)

#include "../../msr3d_amd/csrc/wgrad_split.hip"

extern "C" int msr3d_prof_wgrad_stamps(unsigned long long *host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * WG_MAXWAVES * WG_NSTAMP);
}
