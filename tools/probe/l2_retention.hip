// l2_retention.hip -- does an XCD's L2 keep what a kernel WROTE (or read) for the NEXT kernel's loads on the same XCD?
// Kernel A: workgroup n writes chunk n of a buffer (64 KB a workgroup; workgroup n runs on XCD n mod 8 -- checked with
// s_getreg XCC_ID).  Kernel B: workgroup n reads chunk (n + shift) -- shift 0: the chunk its own XCD wrote, shift 1: a
// chunk another XCD wrote -- and stamps the latency of its first load and the time for the whole chunk.
// hipcc --offload-arch=gfx950 -O2 l2_retention.hip -o l2_retention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

constexpr int CHUNK_V = 4096;   // float4 per workgroup chunk = 64 KB

__global__ __launch_bounds__(256) void writer(float4 *buf, int *xcc) {
  float4 *c = buf + (size_t)blockIdx.x * CHUNK_V;
  for (int i = threadIdx.x; i < CHUNK_V; i += 256) c[i] = make_float4(i, blockIdx.x, 1.f, 2.f);
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
}

__global__ __launch_bounds__(256) void reader(const float4 *buf, int nblk, int shift, long long *stamps, float *out, int *xcc) {
  const int src = (blockIdx.x + shift) % nblk;
  const float4 *c = buf + (size_t)src * CHUNK_V;
  const long long t0 = __builtin_readcyclecounter();
  float4 v0 = c[threadIdx.x];
  float s = v0.x + v0.y;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float4 v[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) v[k] = c[threadIdx.x + 256 * (k + 1)];
#pragma unroll
  for (int k = 0; k < 15; ++k) s += v[k].x + v[k].w;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = t2 - t0;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  const int nblk = 256;
  float4 *buf; long long *st; float *out; int *xa, *xb;
  CK(hipMalloc(&buf, (size_t)nblk * CHUNK_V * sizeof(float4)));
  CK(hipMalloc(&st, nblk * 2 * sizeof(long long)));
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&xa, nblk * 4)); CK(hipMalloc(&xb, nblk * 4));
  std::vector<long long> h(nblk * 2);
  std::vector<int> ha(nblk), hb(nblk);
  for (int mode = 0; mode < 3; ++mode) {            // 0: written by the previous kernel; 1: READ by the previous kernel; 2: cold
    for (int shift : {0, 1, 8}) {
      for (int rep = 0; rep < 3; ++rep) {
        if (mode == 0) writer<<<nblk, 256>>>(buf, xa);
        if (mode == 1) { writer<<<nblk, 256>>>(buf, xa); reader<<<nblk, 256>>>(buf, nblk, shift, st, out, xb); }
        if (mode == 2) { writer<<<nblk, 256>>>(buf, xa); CK(hipDeviceSynchronize());
                         // evict: stream 512 MB through
                         float4 *big; CK(hipMalloc(&big, 512u << 20)); CK(hipMemset(big, 1, 512u << 20)); CK(hipDeviceSynchronize()); CK(hipFree(big)); }
        reader<<<nblk, 256>>>(buf, nblk, shift, st, out, xb);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(h.data(), st, nblk * 2 * sizeof(long long), hipMemcpyDeviceToHost));
      CK(hipMemcpy(ha.data(), xa, nblk * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), xb, nblk * 4, hipMemcpyDeviceToHost));
      std::vector<long long> f, a;
      int same = 0, rr = 0;
      for (int i = 0; i < nblk; ++i) {
        f.push_back(h[2 * i]); a.push_back(h[2 * i + 1]);
        same += ha[(i + shift) % nblk] == hb[i];
        rr += ha[i] == (i % 8);
      }
      std::sort(f.begin(), f.end()); std::sort(a.begin(), a.end());
      printf("mode %d (%s) shift %d: first load median %6lld p90 %6lld cycles; 64 KB median %6lld p90 %6lld; reader on the writer's XCD: %3d / %d; writer n on XCD n mod 8: %d / %d\n",
             mode, mode == 0 ? "written just before" : mode == 1 ? "read just before" : "evicted", shift, f[nblk / 2], f[nblk * 9 / 10],
             a[nblk / 2], a[nblk * 9 / 10], same, nblk, rr, nblk);
    }
  }
  return 0;
}
