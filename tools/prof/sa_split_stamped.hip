// Profiling build of csrc/sa_split.hip: every wave of the level-2 kernel records s_memtime at the phase
// marks of its third tile (steady state).  tools/ab_split.py builds it into tools/_prof/abl/stamp/stamp.so:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I include \
//         tools/prof/sa_split_stamped.hip -o tools/_prof/abl/stamp/stamp.so
#include <hip/hip_runtime.h>

__device__ unsigned long long g_sa2_stamps[512 * 4 * 16];
#define STAMP_DECL                                                                                     \
  unsigned long long *stamps = g_sa2_stamps + ((size_t)(blockIdx.x & 511) * 4 + wave) * 16;            \
  int tile_no = 0
#define STAMP(i) if (tile_no == 2 && (tid & 63) == 0) stamps[i] = __builtin_amdgcn_s_memtime()
#define STAMP_TILE_TOP if (tile_no == 3 && (tid & 63) == 0) stamps[10] = __builtin_amdgcn_s_memtime()
#define STAMP_TILE_END ++tile_no

#include "../../msr3d_amd/csrc/sa_split.hip"

extern "C" int msr3d_prof_sa2_stamps(unsigned long long *host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sa2_stamps), sizeof(unsigned long long) * 512 * 4 * 16);
}
