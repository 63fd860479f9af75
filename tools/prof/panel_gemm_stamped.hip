// Profiling build of csrc/panel_gemm.hip: thread 0 of every workgroup records the shader clock at the
// kernel's phase marks (tools/prof_panel.py builds and reads it).
#include <hip/hip_runtime.h>

__device__ unsigned long long g_panel_stamps[8192 * 8];
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_panel_stamps[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)

#include "../../msr3d_amd/csrc/panel_gemm.hip"

extern "C" int msr3d_prof_panel_stamps(unsigned long long *host_out, int n_blocks) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_panel_stamps), sizeof(unsigned long long) * 8 * n_blocks);
}
