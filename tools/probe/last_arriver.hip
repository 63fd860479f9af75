// last_arriver.hip -- what would folding msr3d_scene_rows into the producing scene block cost?  (round-5 verdict item 1)
//
// The attention forward block ends with 8 heads' partial out-projections of a (scene, 32-row half): 8 slabs x 32 rows x
// 1 KB, written by 8 workgroups that sit on 8 different XCDs (blockIdx.x = head: scene_block.hip).  Today a second launch
// (240 workgroups, a wave per row) adds them up and applies the row-local chain.  The alternative asked for: every
// workgroup takes a ticket per (scene, half) after its stores have left the CU, and the LAST of the eight reads the
// slabs back and does the rows' work itself -- one launch instead of two.
//
//   mode A   producer launch (256 workgroups x 512 threads, plain 16-byte stores) + rows launch (one wave per row:
//            8 slab loads, a row sum standing for the LayerNorm chain, 3 row outputs + 3 bf16 planes)
//   mode B   ONE launch: stores -> s_waitcnt vmcnt(0) -> barrier -> agent-scope release + ticket (thread 0) -> the last
//            arriver's acquire -> its 8 waves take 4 rows each (all 32 slab loads of a wave in flight) -> same outputs
//   mode C   as B with the slabs stored / re-read through device-coherent (sc1) accesses instead of the release /
//            acquire fences (buffer_wbl2 / buffer_inv of the whole L2)
// Each mode: N = 200 repetitions inside one captured graph, time per repetition; mode B / C also print the last
// arrivers' phase stamps (s_memtime ticks): stores issued -> drained -> ticket known -> slabs loaded -> outputs stored.
// hipcc --offload-arch=gfx950 -O2 last_arriver.hip -o last_arriver
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

constexpr int B = 16, H = 8, L = 60, D = 256, M = B * L;
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// the row-local stand-in: sum of the slabs (+ residual), mean / rstd over the row, 3 outputs + 3 planes
__device__ __forceinline__ void finish_row(const float4 (&v)[8], const float *__restrict__ resid, int row, int lane,
                                           float *__restrict__ o0, float *__restrict__ o1, float *__restrict__ o2,
                                           unsigned short *__restrict__ xp) {
  float4 a = v[0];
#pragma unroll
  for (int s = 1; s < 8; ++s) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
  const float4 r = *reinterpret_cast<const float4 *>(resid + (size_t)row * D + 4 * lane);
  a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
  const float mean = row_sum((a.x + a.y) + (a.z + a.w)) * (1.f / D);
  const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
  const float rstd = rsqrtf(row_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / D) + 1e-5f);
  const float4 y = make_float4(dx * rstd, dy * rstd, dz * rstd, dw * rstd);
  const size_t o = (size_t)row * D + 4 * lane;
  *reinterpret_cast<float4 *>(o0 + o) = a;
  *reinterpret_cast<float4 *>(o1 + o) = y;
  *reinterpret_cast<float4 *>(o2 + o) = make_float4(y.x + r.x, y.y + r.y, y.z + r.z, y.w + r.w);
  const int b = row / L, rr = row - b * L;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint2 pl = make_uint2(__float_as_uint(y.x) >> 16 | (__float_as_uint(y.y) & 0xffff0000u),
                                __float_as_uint(y.z) >> 16 | (__float_as_uint(y.w) & 0xffff0000u));
    *reinterpret_cast<uint2 *>(xp + ((size_t)(b * 3 + k) * 64 + rr) * D + 4 * lane) = pl;
  }
}

template <int MODE>      // 0: producer only (mode A), 1: fences (B), 2: sc1 accesses (C)
__global__ __launch_bounds__(512) void producer(float *__restrict__ part, const float *__restrict__ resid, int *ticket,
                                                float *o0, float *o1, float *o2, unsigned short *xp,
                                                unsigned long long *stamps, float seedv) {
  __shared__ int last_s;
  const int h = blockIdx.x & 7, half = blockIdx.x >> 3, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = 32 * half, nq = min(32, L - q0);
  float *slab = part + (size_t)h * M * D;
  unsigned long long t0 = __builtin_readcyclecounter();
  // the block's result: 32 rows x 256 floats = 2048 float4, 4 per thread
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = tid + 512 * k, r = e >> 6, c4 = e & 63;
    if (r < nq) {
      const float4 v = make_float4(seedv + e, seedv - e, 0.5f * e, 1.f);
      float4 *d = reinterpret_cast<float4 *>(slab + (size_t)(b * L + q0 + r) * D) + c4;
      if (MODE == 2) {                            // agent-scope write-through store
        const vf4 vv = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(vv) : "memory");
      } else {
        *d = v;
      }
    }
  }
  if (MODE == 0) return;
  unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned long long t2 = __builtin_readcyclecounter();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const int k = __hip_atomic_fetch_add(ticket + b * 2 + half, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = k == H - 1;
    if (k == H - 1) __hip_atomic_store(ticket + b * 2 + half, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_s) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned long long t3 = __builtin_readcyclecounter();
  // 8 waves x 4 rows: all 32 slab loads of the wave first
  float4 v[4][8];
  if (MODE == 2) {
    vf4 tv[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = min(4 * wave + j, nq - 1);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float4 *src = reinterpret_cast<const float4 *>(part + (size_t)s * M * D + (size_t)(b * L + q0 + r) * D) + lane;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(tv[j][s]) : "v"(src) : "memory");
      }
    }
    // (the compiler does not count these loads: the wait names every destination so that no use moves above it)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(tv[j][0]), "+v"(tv[j][1]), "+v"(tv[j][2]), "+v"(tv[j][3]), "+v"(tv[j][4]), "+v"(tv[j][5]),
                     "+v"(tv[j][6]), "+v"(tv[j][7])::"memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s = 0; s < 8; ++s) v[j][s] = make_float4(tv[j][s].x, tv[j][s].y, tv[j][s].z, tv[j][s].w);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = min(4 * wave + j, nq - 1);
#pragma unroll
      for (int s = 0; s < 8; ++s)
        v[j][s] = *(reinterpret_cast<const float4 *>(part + (size_t)s * M * D + (size_t)(b * L + q0 + r) * D) + lane);
    }
    // (stamp t4 below must see the data: touch one element of every row)
  }
  unsigned long long t4 = __builtin_readcyclecounter();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * wave + j < nq) finish_row(v[j], resid, b * L + q0 + 4 * wave + j, lane, o0, o1, o2, xp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t5 = __builtin_readcyclecounter();
  if (tid == 0 && stamps) {
    unsigned long long *s = stamps + (size_t)(b * 2 + half) * 8;
    s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t4 - t3; s[4] = t5 - t4; s[5] = t5 - t0;
  }
}

__global__ __launch_bounds__(256) void rows(const float *__restrict__ part, const float *__restrict__ resid, float *o0,
                                            float *o1, float *o2, unsigned short *xp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  float4 v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) v[s] = *(reinterpret_cast<const float4 *>(part + (size_t)s * M * D + (size_t)row * D) + lane);
  finish_row(v, resid, row, lane, o0, o1, o2, xp);
}

int main() {
  float *part, *resid, *o0, *o1, *o2; unsigned short *xp; int *ticket; unsigned long long *stamps;
  CK(hipMalloc(&part, (size_t)8 * M * D * 4)); CK(hipMalloc(&resid, (size_t)M * D * 4));
  CK(hipMalloc(&o0, (size_t)M * D * 4)); CK(hipMalloc(&o1, (size_t)M * D * 4)); CK(hipMalloc(&o2, (size_t)M * D * 4));
  CK(hipMalloc(&xp, (size_t)B * 3 * 64 * D * 2)); CK(hipMalloc(&ticket, 64 * 4)); CK(hipMalloc(&stamps, 32 * 8 * 8));
  CK(hipMemset(resid, 0, (size_t)M * D * 4)); CK(hipMemset(ticket, 0, 64 * 4)); CK(hipMemset(xp, 0, (size_t)B * 3 * 64 * D * 2));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 200;
  const char *names[3] = {"A  producer launch + rows launch", "B  one launch, last arriver (release / acquire fences)",
                          "C  one launch, last arriver (write-through stores, sc1 loads)"};
  for (int mode = 0; mode < 3; ++mode) {
    auto launch = [&](int i) {
      const float sv = 0.001f * i;
      if (mode == 0) {
        producer<0><<<dim3(16, B), 512, 0, st>>>(part, resid, ticket, o0, o1, o2, xp, nullptr, sv);
        rows<<<(M + 3) / 4, 256, 0, st>>>(part, resid, o0, o1, o2, xp);
      } else if (mode == 1) {
        producer<1><<<dim3(16, B), 512, 0, st>>>(part, resid, ticket, o0, o1, o2, xp, stamps, sv);
      } else {
        producer<2><<<dim3(16, B), 512, 0, st>>>(part, resid, ticket, o0, o1, o2, xp, stamps, sv);
      }
    };
    for (int i = 0; i < 10; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("mode %s: %.2f us per repetition (graph of %d)\n", names[mode], best * 1e3 / N, N);
    if (mode > 0) {
      std::vector<unsigned long long> h(32 * 8);
      CK(hipMemcpy(h.data(), stamps, 32 * 8 * 8, hipMemcpyDeviceToHost));
      const char *ph[6] = {"stores issued", "stores drained + barrier", "release + ticket + acquire", "8 slabs x 4 rows loaded",
                           "rows' arithmetic + outputs stored", "whole workgroup"};
      for (int k = 0; k < 6; ++k) {
        std::vector<unsigned long long> v;
        for (int i = 0; i < 32; ++i) v.push_back(h[i * 8 + k]);
        std::sort(v.begin(), v.end());
        printf("    last arrivers, %-34s median %6llu  max %6llu cycles\n", ph[k], v[16], v[31]);
      }
    }
    // sanity: the outputs of the three modes agree (checksum of o0)
    std::vector<float> ho((size_t)M * D);
    CK(hipMemcpy(ho.data(), o0, ho.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float x : ho) cs += x;
    printf("    checksum(o0) %.6e\n", cs);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
