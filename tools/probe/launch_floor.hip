// launch_floor.hip -- what does one dependent kernel boundary cost on this box, eagerly and
// inside a captured graph?  hipcc --offload-arch=gfx950 -O2 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void touch(float *p, int n) {           // every thread reads + writes one float4
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { float4 v = reinterpret_cast<float4 *>(p)[i]; v.x += 1.f; reinterpret_cast<float4 *>(p)[i] = v; }
}

int main() {
  float *buf; const int n4 = 960 * 256 / 4;      // one (960 x 256) fp32 activation
  CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 200;
  for (int mode = 0; mode < 3; ++mode) {          // 0: 1-thread kernel, 1: 240 WGs touching 1 MB, 2: 1024 WGs touching 4 MB
    auto launch = [&]() {
      if (mode == 0) tiny<<<1, 64, 0, st>>>(buf);
      else if (mode == 1) touch<<<n4 / 256, 256, 0, st>>>(buf, n4);
      else touch<<<4 * n4 / 256, 256, 0, st>>>(buf, 4 * n4);
    };
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a, st));
      for (int i = 0; i < N; ++i) launch();
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("mode %d eager: %.2f us per kernel\n", mode, best * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("mode %d graph: %.2f us per kernel\n", mode, best * 1e3 / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
