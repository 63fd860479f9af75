// l2_stream.hip -- how fast can ONE CU pull an L2-resident weight stream (1 KB coalesced wave loads)
// as a function of loads in flight per wave?  hipcc --offload-arch=gfx950 -O2 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void stream(const float4 *__restrict__ w, int n_vec, int iters, float *out) {
  // every block reads the SAME buffer (like a weight matrix): wave v of the block starts at a different offset
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  int pos = (wave * 997 + blockIdx.x * 131) * 64;
  for (int it = 0; it < iters; ++it) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { v[d] = w[(pos + d * 64 + lane) % n_vec]; }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
    pos += DEPTH * 64;
  }
  if (acc.x == 123.456f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH>
int run(const float4 *w, int n_vec, float *out, int blocks, int wgs_per_cu_label) {
  const int total_loads = 4096;                      // per wave
  const int iters = total_loads / DEPTH;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  stream<DEPTH><<<blocks, 256>>>(w, n_vec, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  stream<DEPTH><<<blocks, 256>>>(w, n_vec, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)blocks * 4 * iters * DEPTH * 1024.0;
  const double per_cu = bytes / 256.0 / (ms * 1e-3) / 2.4e9;     // B/clk/CU at 2.4 GHz, 256 CUs
  printf("depth %2d  blocks %4d (%d/CU): %7.1f GB/s total, %5.1f B/clk/CU\n", DEPTH, blocks, wgs_per_cu_label,
         bytes / (ms * 1e-3) / 1e9, per_cu);
  return 0;
}

int main() {
  const int n_vec = (2 << 20) / 16;                  // 2 MB buffer: L2-resident in every XCD
  float4 *w; float *out;
  CK(hipMalloc(&w, n_vec * 16)); CK(hipMemset(w, 0, n_vec * 16)); CK(hipMalloc(&out, 64));
  for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
    const int blocks = 256 * per_cu;
    run<1>(w, n_vec, out, blocks, per_cu); run<2>(w, n_vec, out, blocks, per_cu); run<4>(w, n_vec, out, blocks, per_cu);
    run<8>(w, n_vec, out, blocks, per_cu); run<16>(w, n_vec, out, blocks, per_cu);
  }
  return 0;
}
