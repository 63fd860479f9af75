// Profiling build of csrc/sa_split.hip for the distinct-row kernels: every wave logs (mark id, s_memtime) pairs,
// up to 240 per wave, at the phase marks RSTAMP(i) of sa2_rows_kernel.  tools/prof_sa_rows.py builds and reads it:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I include \
//         tools/prof/sa_rows_stamped.hip -o tools/_prof/sa_rows_stamp.so
#include <hip/hip_runtime.h>

constexpr int kRLog = 240;
__device__ unsigned long long g_rows_log[512 * 8 * kRLog];
#define RSTAMP_DECL                                                                                         \
  unsigned long long *rlog = g_rows_log + ((size_t)(blockIdx.x & 511) * 8 + (threadIdx.x >> 6)) * kRLog;    \
  int rlog_n = 0
#define RSTAMP(i)                                                                                            \
  if ((threadIdx.x & 63) == 0 && rlog_n < kRLog)                                                             \
    rlog[rlog_n++] = ((unsigned long long)(i) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull)

#include "../../msr3d_amd/csrc/sa_split.hip"

extern "C" int msr3d_prof_rows_log(unsigned long long *host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rows_log), sizeof(unsigned long long) * 512 * 8 * kRLog);
}
extern "C" int msr3d_prof_rows_clear() {
  static unsigned long long zeros[512 * 8 * kRLog];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_log), zeros, sizeof(zeros));
}
