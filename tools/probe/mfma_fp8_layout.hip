// Pins the operand layout and scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950:
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_fp8_layout.hip -o tools/_prof/mfma_fp8_layout && tools/_prof/mfma_fp8_layout
// Hypothesis: lane l supplies row (A) / column (B) i = l & 15 and the 32 consecutive k values 32 (l >> 4) .. + 31, as 32 bytes in
// 8 VGPRs (byte b of the 32 = k offset b); scale byte 127 (E8M0) = 1.0; D as every 16x16 MFMA: lane (j, g) holds D[4 g + r][j].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned char *A, const unsigned char *B, float *D, int scale_a, int scale_b) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = *reinterpret_cast<const int *>(A + i * 128 + 32 * g + 4 * w);
    b[w] = *reinterpret_cast<const int *>(B + i * 128 + 32 * g + 4 * w);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}

__global__ void cvt(const float *x, unsigned char *q, int n) {       // v_cvt_pk_fp8_f32 of pairs
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * t + 1 < n) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * t], x[2 * t + 1], w, false);
    q[2 * t] = w & 0xff;
    q[2 * t + 1] = (w >> 8) & 0xff;
  }
}

static float e4m3_to_float(unsigned char v) {      // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = std::ldexp((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -f : f;
}

int main() {
  std::vector<unsigned char> A(16 * 128), B(16 * 128);
  srand(1);
  for (auto &v : A) v = (unsigned char)(rand() % 0x78) | ((rand() & 1) << 7);     // finite e4m3 codes
  for (auto &v : B) v = (unsigned char)(rand() % 0x78) | ((rand() & 1) << 7);
  unsigned char *dA, *dB;
  float *dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int sc : {(int)0x7f7f7f7f, (int)0x80808080, 0}) {
    probe<<<1, 64>>>(dA, dB, dD, sc, 0x7f7f7f7f);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double err = 0, ref2 = 0, ratio = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)e4m3_to_float(A[m * 128 + k]) * e4m3_to_float(B[n * 128 + k]);
        err += (D[m * 16 + n] - s) * (D[m * 16 + n] - s);
        ref2 += s * s;
        if (m == 3 && n == 5) ratio = D[m * 16 + n] / s;
      }
    printf("scale_a %08x: rel err vs D[m][n] = sum_k A[m][k] B[n][k] : %.3e   (D/ref at (3,5): %.6g)\n", sc, std::sqrt(err / ref2), ratio);
  }
  // conversion: which format does v_cvt_pk_fp8_f32 write?
  std::vector<float> x = {1.0f, -1.0f, 0.5f, 448.0f, 500.0f, 0.0625f, 3.3f, -0.017f, 240.f, 1e-3f};
  float *dx; unsigned char *dq;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dq, x.size());
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  cvt<<<1, 64>>>(dx, dq, (int)x.size());
  std::vector<unsigned char> q(x.size());
  hipMemcpy(q.data(), dq, q.size(), hipMemcpyDeviceToHost);
  for (size_t k = 0; k < x.size(); ++k) printf("cvt %g -> 0x%02x = %g (as OCP e4m3fn)\n", x[k], q[k], e4m3_to_float(q[k]));
  return 0;
}
