// mfma_peak.hip -- what the matrix pipes SUSTAIN on this box: a register-only MFMA loop on every SIMD of
// every CU, long enough (~0.3-3 ms) for the chip to settle at the clock its power budget allows.  The
// spec-sheet peaks assume 2.4 GHz; roofline fractions in DESIGN.md are also given against these numbers.
//   hipcc --offload-arch=gfx950 -O2 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// KIND 2: v_mfma_f32_32x32x16_bf16 (16 accumulator registers each); ZERO: all-zero operands (DVFS give-back)
template <bool ZERO>
__global__ __launch_bounds__(256) void spin32(float *out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf16x8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = ZERO ? 0 : (short)(0x3f80 + threadIdx.x + i); hb[i] = ZERO ? 0 : (short)(0x3f00 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
__global__ __launch_bounds__(256) void spin(float *out, int iters, float seed) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  bf16x8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = seed == 0.f ? 0 : (short)(0x3f80 + threadIdx.x + i); hb[i] = seed == 0.f ? 0 : (short)(0x3f00 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // inline asm: with the builtin hipcc rotates the accumulators through AGPR copies every iteration
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;       // never true: keeps the loop alive
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float *out; CK(hipMalloc(&out, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int kind = 0; kind < 2; ++kind) {
    const double flop_per = kind == 0 ? 16.0 * 16 * 4 * 2 : 16.0 * 16 * 32 * 2;
    for (int waves = 1; waves <= 2; ++waves) {            // waves per SIMD
      const int blocks = cus * waves;                     // 4 waves per block: one per SIMD
      for (int iters : {20000, 200000}) {
        auto go = [&]() {
          if (kind == 0) spin<0><<<blocks, 256>>>(out, iters, 1.25f); else spin<1><<<blocks, 256>>>(out, iters, 1.25f);
        };
        go(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double n = (double)blocks * 4 * iters * 8;   // MFMA instructions
        const double cyc = kind == 0 ? 32.0 : 16.0;        // issue cycles per instruction per SIMD
        printf("%s  %d wave(s)/SIMD  %7.3f ms  %8.1f TFLOP/s  implied clock %.2f GHz\n",
               kind == 0 ? "v_mfma_f32_16x16x4_f32  " : "v_mfma_f32_16x16x32_bf16", waves, ms, n * flop_per / (ms * 1e-3) / 1e12,
               n * cyc / (cus * 4.0) / (ms * 1e-3) / 1e9 / (waves > 1 ? 1.0 : 1.0));
      }
    }
  }
  for (int zero = 0; zero < 2; ++zero)
    for (int waves = 1; waves <= 2; ++waves) {
      const int blocks = cus * waves, iters = 100000;
      auto go = [&]() { if (zero) spin32<true><<<blocks, 256>>>(out, iters); else spin32<false><<<blocks, 256>>>(out, iters); };
      go(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); go(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double n = (double)blocks * 4 * iters * 4;
      printf("v_mfma_f32_32x32x16_bf16 %s %d wave(s)/SIMD  %7.3f ms  %8.1f TFLOP/s  implied clock %.2f GHz\n", zero ? "zeros " : "values",
             waves, ms, n * 32.0 * 32 * 16 * 2 / (ms * 1e-3) / 1e12, n * 32.0 / (cus * 4.0) / (ms * 1e-3) / 1e9);
    }
  {   // 16x16x32 with all-zero operands
    const int blocks = cus * 2, iters = 100000;
    spin<1><<<blocks, 256>>>(out, iters, 0.f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); spin<1><<<blocks, 256>>>(out, iters, 0.f); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)blocks * 4 * iters * 8;
    printf("v_mfma_f32_16x16x32_bf16 zeros  2 wave(s)/SIMD  %7.3f ms  %8.1f TFLOP/s\n", ms, n * 16.0 * 16 * 32 * 2 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
