// lora_linear.hip -- a LoRA-augmented linear layer of the frozen bf16 language model
// (/root/reference/model/msr3d/msr3d.py:103-112: peft LoraConfig r = 16, alpha = 16, dropout 0 on
// q/k/v/o/gate/up/down_proj; the LLM runs under bf16 autocast, msr3d.py:409-415):
//
//     y = x W^T + s (x A^T) B^T                     W (N,K) frozen bf16, A (r,K), B (N,r) trainable
//     dx = dy W + s (dy B) A        dA = s (dy B)^T x        dB = s dy^T (x A^T)         (no dW)
//
// One bf16 MFMA kernel (v_mfma_f32_16x16x32_bf16, fp32 accumulate) serves forward and dx:
//
//     C (M,N) = P Q^T + P2 Q2^T        P (M,K), Q (N,K), P2 (M,R), Q2 (N,R): all k-contiguous bf16
//
// i.e. the low-rank term rides as ONE extra K step of the same product (P2 = s x A^T computed by a
// first call of the same kernel against the zero-padded A; R = 32).  The frozen weight is kept in
// BOTH orientations (W for the forward, W^T for dx): 13 GB extra for a 7B model, which is what a
// 288 GB part is for -- no transposing load path, both products read k-contiguous rows.
// 128 x 128 tile, BK = 64, LDS double-buffered through registers, 4 waves of 64 x 64 (4 x 4 MFMA
// tiles).  The weight gradients' token reduction (M = a few thousand, output r x K / N x r) is a
// column-per-thread VALU kernel: it is bound by reading x / dy once.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LD = BK + 8;                 // bf16 units: row stride 144 B

__device__ __forceinline__ unsigned short f2bf(float f) {        // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

struct GemmArgs {
  int M, N, K, R;
  const unsigned short *P; int ldp;
  const unsigned short *Q; int ldq;
  const unsigned short *P2; int ldp2;
  const unsigned short *Q2; int ldq2;
  void *C; int ldc; int c_f32;
  float scale;                             // applied to the whole result (the forward's u = s x A^T)
};

// rows x 64 k of a row-major bf16 matrix -> registers: 4 x 16 B per thread (row = t/8 + 32 j, k8 = t%8)
__device__ __forceinline__ void tile_load(uint4 (&r)[4], const unsigned short *__restrict__ S, int ld, int row0,
                                          int rows, int k0, int kmax) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = min(row0 + (t >> 3) + 32 * j, rows - 1), k = k0 + (t & 7) * 8;
    r[j] = k < kmax ? *reinterpret_cast<const uint4 *>(S + (size_t)row * ld + k) : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void tile_store(unsigned short *L, const uint4 (&r)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<uint4 *>(L + ((t >> 3) + 32 * j) * LD + (t & 7) * 8) = r[j];
}

__global__ __launch_bounds__(256, 2) void bf16_gemm_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int STAGE = 2 * BM * LD;             // one stage = A tile then B tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the K walk: main operands, then (if R > 0) the low-rank pair as extra stages
  const int nk_main = a.K / BK, nk = nk_main + (a.R + BK - 1) / BK;
  auto fetch = [&](int kt, uint4 (&ra)[4], uint4 (&rb)[4]) {
    if (kt < nk_main) {
      tile_load(ra, a.P, a.ldp, m0, a.M, kt * BK, a.K);
      tile_load(rb, a.Q, a.ldq, n0, a.N, kt * BK, a.K);
    } else {
      const int k0 = (kt - nk_main) * BK;
      tile_load(ra, a.P2, a.ldp2, m0, a.M, k0, a.R);
      tile_load(rb, a.Q2, a.ldq2, n0, a.N, k0, a.R);
    }
  };
  uint4 ra[4], rb[4];
  fetch(0, ra, rb);
  tile_store(lds, ra);
  tile_store(lds + BM * LD, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned short *Acur = lds + (kt & 1) * STAGE, *Bcur = Acur + BM * LD;
    unsigned short *Anxt = lds + ((kt + 1) & 1) * STAGE;
    if (kt + 1 < nk) fetch(kt + 1, ra, rb);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        fa[x] = *reinterpret_cast<const bf16x8 *>(Acur + (wm * 64 + x * 16 + i) * LD + ks * 32 + g * 8);
#pragma unroll
      for (int y = 0; y < 4; ++y)
        fb[y] = *reinterpret_cast<const bf16x8 *>(Bcur + (wn * 64 + y * 16 + i) * LD + ks * 32 + g * 8);
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[x], fb[y], acc[x][y], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      tile_store(Anxt, ra);
      tile_store(Anxt + BM * LD, rb);
    }
    __syncthreads();
  }
  // epilogue: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int y = 0; y < 4; ++y) {
    const int col = n0 + wn * 64 + y * 16 + i;
    if (col >= a.N) continue;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + x * 16 + g * 4 + r;
        if (row >= a.M) continue;
        const float v = acc[x][y][r] * a.scale;
        if (a.c_f32) reinterpret_cast<float *>(a.C)[(size_t)row * a.ldc + col] = v;
        else reinterpret_cast<unsigned short *>(a.C)[(size_t)row * a.ldc + col] = f2bf(v);
      }
  }
}

// out (R, C) (or its transpose) += sum_m P[m][r] * Q[m][c]: thread = column c, R accumulators;
// rows split over blockIdx.y, meeting by atomicAdd.  P (M, R) bf16 is broadcast from LDS.
template <int R>
__global__ __launch_bounds__(256) void lora_grad_kernel(int M, int C, const unsigned short *__restrict__ P, int ldp,
                                                        const unsigned short *__restrict__ Q, int ldq,
                                                        float *__restrict__ out, int transpose_out, float scale) {
  __shared__ float ps[64][R];
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int mb = blockIdx.y * per, me = min(M, mb + per);
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  for (int m0 = mb; m0 < me; m0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * R; e += 256) {
      const int mm = e / R, r = e - mm * R;
      ps[mm][r] = (m0 + mm < me) ? __uint_as_float((unsigned)P[(size_t)(m0 + mm) * ldp + r] << 16) : 0.f;
    }
    __syncthreads();
    if (c < C) {
      const int lim = min(64, me - m0);
      for (int mm = 0; mm < lim; ++mm) {
        const float q = __uint_as_float((unsigned)Q[(size_t)(m0 + mm) * ldq + c] << 16);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fmaf(ps[mm][r], q, acc[r]);
      }
    }
  }
  if (c < C) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      atomicAdd(out + (transpose_out ? (size_t)c * R + r : (size_t)r * C + c), acc[r] * scale);
  }
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

int msr3d_bf16_gemm_lowrank(int M, int N, int K, int R, const void *P, int ldp, const void *Q, int ldq,
                            const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                            int c_f32, float scale, msr3d_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || R < 0 || (K % BK) != 0 || (R % 8) != 0) return MSR3D_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!P || !Q || !C || ldp < K || ldq < K || ldc < N || (ldp % 8) || (ldq % 8) || !al16(P) || !al16(Q))
    return MSR3D_EINVAL;
  if (R > 0 && (!P2 || !Q2 || ldp2 < R || ldq2 < R || (ldp2 % 8) || (ldq2 % 8) || !al16(P2) || !al16(Q2)))
    return MSR3D_EINVAL;
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.R = R;
  a.P = (const unsigned short *)P; a.ldp = ldp; a.Q = (const unsigned short *)Q; a.ldq = ldq;
  a.P2 = (const unsigned short *)P2; a.ldp2 = ldp2; a.Q2 = (const unsigned short *)Q2; a.ldq2 = ldq2;
  a.C = C; a.ldc = ldc; a.c_f32 = c_f32; a.scale = scale;
  constexpr size_t lds = sizeof(unsigned short) * 4 * BM * LD;         // 73.7 KB: two stages of A and B
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_gemm_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  bf16_gemm_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(a);
  return (int)hipGetLastError();
}

int msr3d_lora_grad(int M, int R, int C, const void *P, int ldp, const void *Q, int ldq, float *out,
                    int transpose_out, float scale, msr3d_stream_t stream) {
  if (M < 0 || C < 0 || (R != 16 && R != 32)) return MSR3D_EINVAL;
  if (M == 0 || C == 0) return 0;
  if (!P || !Q || !out || ldp < R || ldq < C) return MSR3D_EINVAL;
  int splits = (M + 255) / 256;
  if (splits > 32) splits = 32;
  dim3 grid((C + 255) / 256, splits);
  hipStream_t st = (hipStream_t)stream;
  if (R == 16)
    lora_grad_kernel<16><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                               out, transpose_out, scale);
  else
    lora_grad_kernel<32><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                               out, transpose_out, scale);
  return (int)hipGetLastError();
}

}  // extern "C"
