// What does s_memtime count, and how fast does a lone wave issue dependent VALU instructions?
//   hipcc --offload-arch=gfx950 -O2 tools/prof/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void probe(unsigned long long *out, int n) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float c[8];
  for (int k = 0; k < 8; ++k) c[k] = threadIdx.x + k;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(c[k % CHAINS]));   // 16 FMAs over CHAINS chains
  }
  float a = 0.f;
  for (int k = 0; k < 8; ++k) a += c[k];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t1 - t0;
    out[blockIdx.x * 4 + 1] = r1 - r0;
  }
  if (a == 12345.f) out[3] = 1;
}

int main() {
  unsigned long long *d, h[4];
  hipMalloc(&d, 1024 * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int chains : {1, 2, 3, 4, 8})
  for (int blocks : {1, 1024, 2048, 4096}) {
    const int n = 20000;
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0, 0);
      switch (chains) {
        case 1: probe<1><<<blocks, 64>>>(d, n); break;
        case 2: probe<2><<<blocks, 64>>>(d, n); break;
        case 3: probe<3><<<blocks, 64>>>(d, n); break;
        case 4: probe<4><<<blocks, 64>>>(d, n); break;
        default: probe<8><<<blocks, 64>>>(d, n); break;
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%d chains, %4d waves: kernel %.1f us; s_memtime ticks %llu, s_memrealtime ticks %llu -> s_memtime = %.1f MHz if realtime is 100 MHz; "
           "%.2f memtime ticks per v_fma_f32, %.2f ns each\n", chains, blocks, ms * 1e3, h[0], h[1], 100.0 * h[0] / h[1],
           (double)h[0] / (16.0 * n), ms * 1e6 / (16.0 * n));
  }
  return 0;
}
