// Where do the four waves of a 256-thread workgroup land?  Launches the sampling kernel's shape (960 workgroups x 4 waves,
// 25 KB of LDS) and records HW_ID / XCC_ID per wave.  Prints, per wave index, the histogram of SIMD ids, and the largest
// number of wave-0s that shared one SIMD of one CU.     hipcc --offload-arch=gfx950 -O2 wave_placement.hip -o wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;       // stay resident so that all 960 workgroups coexist
  lds[threadIdx.x] = a;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 0] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + (lds[threadIdx.x] == 12345.f);
  }
}

int main() {
  const int blocks = 960;
  unsigned *d;
  hipMalloc(&d, blocks * 4 * 2 * sizeof(unsigned));
  probe<<<blocks, 256, 25 * 1024>>>(d, 200000);
  std::vector<unsigned> h(blocks * 4 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int hist[4][4] = {};
  std::map<unsigned, int> wave0_per_simd;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      hist[w][simd]++;
      if (w == 0) wave0_per_simd[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd]++;
    }
  for (int w = 0; w < 4; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  int worst = 0, simds = 0;
  std::map<int, int> dist;
  for (auto &kv : wave0_per_simd) { worst = kv.second > worst ? kv.second : worst; ++simds; dist[kv.second]++; }
  printf("wave-0s per (xcc, se, sh, cu, simd): %d SIMDs hold one or more; worst %d\n", simds, worst);
  for (auto &kv : dist) printf("  %d SIMDs hold %d wave-0s\n", kv.second, kv.first);
  for (int b = 0; b < 8; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 4; ++w) printf(" hw %08x xcc %u |", h[(b * 4 + w) * 2], h[(b * 4 + w) * 2 + 1]);
    printf("\n");
  }
  return 0;
}
