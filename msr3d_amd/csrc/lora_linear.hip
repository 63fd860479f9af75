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
// 192 x 128 (or 128 x 128) tile, BK = 64, one LDS stage + register prefetch, 4 waves of 96 x 64 (6 x 4 MFMA
// tiles); msr3d_bf16_gemm_batched: the same kernel over a batch (attention's per-head products).  The weight gradients' token reduction (M = a few thousand, output r x K / N x r) is a
// column-per-thread VALU kernel: it is bound by reading x / dy once.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "../../include/msr3d_hip.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BN = 128, BK = 64;
constexpr int LD = BK + 8;                 // bf16 units: row stride 144 B (conflict-free 16-byte fragment reads)

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
  int inner;                               // blockIdx.z = outer * inner + inner index (e.g. sequence, head)
  long long spo, spi, sqo, sqi, sco, sci;   // batch strides (elements) of P, Q, C, outer and inner
};

// ROWS x 64 k of a row-major bf16 matrix -> registers: 16 B per thread and pass (row = t/8 + 32 j, k8 = t%8)
template <int ROWS>
__device__ __forceinline__ void tile_load(uint4 (&r)[ROWS / 32], const unsigned short *__restrict__ S, int ld, int row0,
                                          int rows, int k0, int kmax) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < ROWS / 32; ++j) {
    const int row = min(row0 + (t >> 3) + 32 * j, rows - 1), k = k0 + (t & 7) * 8;
    r[j] = k < kmax ? *reinterpret_cast<const uint4 *>(S + (size_t)row * ld + k) : make_uint4(0, 0, 0, 0);
  }
}
template <int ROWS>
__device__ __forceinline__ void tile_store(unsigned short *L, const uint4 (&r)[ROWS / 32]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < ROWS / 32; ++j)
    *reinterpret_cast<uint4 *>(L + ((t >> 3) + 32 * j) * LD + (t & 7) * 8) = r[j];
}

// BM x 128 tile, 4 waves as 2 x 2, each (BM / 2) x 64 = MT x 4 MFMA tiles.  BM = 192 is the shape for the
// language model's token counts (4 x 576 = 2304 = 12 x 192): 6 x 4 tiles per wave read 10 fragments per 24
// MFMAs -- the 64 x 64 sub-tile of the 128 x 128 variant reads 8 per 16, which with two workgroups per CU is
// exactly the LDS's 128 B/clk (round 2: 275 TFLOP/s, the 576 tiles of 2304 x 4096 also being 1.125 rounds of
// the chip's 512 slots).  The accumulators are transposed (D = Q P^T: a lane holds FOUR CONSECUTIVE output
// columns of one row), so the epilogue is one 8-byte (bf16) or 16-byte (fp32) store per tile instead of four
// scattered 2-byte ones.  Single LDS stage + register prefetch: next tile's loads fly under this tile's MFMAs.
template <int BM>
__global__ __launch_bounds__(256, 2) void bf16_gemm_kernel(const GemmArgs a) {
  constexpr int MT = BM / 32;                    // 16-row tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short *As = lds, *Bs = lds + BM * LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int bo = blockIdx.z / a.inner, bi = blockIdx.z - bo * a.inner;
  const unsigned short *P = a.P + bo * a.spo + bi * a.spi, *Q = a.Q + bo * a.sqo + bi * a.sqi;

  f32x4 acc[MT][4];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the K walk: main operands, then (if R > 0) the low-rank pair as extra stages
  const int nk_main = a.K / BK, nk = nk_main + (a.R + BK - 1) / BK;
  uint4 ra[BM / 32], rb[BN / 32];
  auto fetch = [&](int kt) {
    if (kt < nk_main) {
      tile_load<BM>(ra, P, a.ldp, m0, a.M, kt * BK, a.K);
      tile_load<BN>(rb, Q, a.ldq, n0, a.N, kt * BK, a.K);
    } else {
      const int k0 = (kt - nk_main) * BK;
      tile_load<BM>(ra, a.P2, a.ldp2, m0, a.M, k0, a.R);
      tile_load<BN>(rb, a.Q2, a.ldq2, n0, a.N, k0, a.R);
    }
  };
  fetch(0);
  for (int kt = 0; kt < nk; ++kt) {
    tile_store<BM>(As, ra);
    tile_store<BN>(Bs, rb);
    __syncthreads();
    if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 fa[MT], fb[4];
#pragma unroll
      for (int y = 0; y < 4; ++y)
        fb[y] = *reinterpret_cast<const bf16x8 *>(Bs + (wn * 64 + y * 16 + i) * LD + ks * 32 + g * 8);
#pragma unroll
      for (int x = 0; x < MT; ++x)
        fa[x] = *reinterpret_cast<const bf16x8 *>(As + (wm * (BM / 2) + x * 16 + i) * LD + ks * 32 + g * 8);
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[y], fa[x], acc[x][y], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: D = Q P^T -- lane (i, g) holds columns n = 4 g + r of row m = i
#pragma unroll
  for (int x = 0; x < MT; ++x) {
    const int row = m0 + wm * (BM / 2) + x * 16 + i;
    if (row >= a.M) continue;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int col = n0 + wn * 64 + y * 16 + 4 * g;
      if (col >= a.N) continue;
      const float v0 = acc[x][y][0] * a.scale, v1 = acc[x][y][1] * a.scale, v2 = acc[x][y][2] * a.scale,
                  v3 = acc[x][y][3] * a.scale;
      const size_t o = (size_t)(bo * a.sco + bi * a.sci) + (size_t)row * a.ldc + col;
      if (col + 3 < a.N) {
        if (a.c_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.C) + o) = make_float4(v0, v1, v2, v3);
        else *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(a.C) + o) =
            make_uint2(f2bf(v0) | ((unsigned)f2bf(v1) << 16), f2bf(v2) | ((unsigned)f2bf(v3) << 16));
      } else {
        const float vv[4] = {v0, v1, v2, v3};
        for (int r = 0; r < 4 && col + r < a.N; ++r) {
          if (a.c_f32) reinterpret_cast<float *>(a.C)[o + r] = vv[r];
          else reinterpret_cast<unsigned short *>(a.C)[o + r] = f2bf(vv[r]);
        }
      }
    }
  }
}

// The same product with the operands staged by LDS-DMA (global_load_lds, 16 bytes per lane) into THREE LDS
// stages, two K steps in flight: the register-staged kernel above has one K step (768 cycles of MFMA) to
// cover a ~2k-cycle load, and its prefetch registers (40 per stage) leave no room for a second one.  The DMA
// writes wave-uniform base + lane x 16, so the LDS image is unpadded ([rows][64] bf16 = 8 chunks of 16 B per
// row) and the bank-conflict fix is on the SOURCE side: chunk c of row r is stored at position c ^ (r & 7), and
// the fragment read of lane (i, g) takes position (4 ks + g) ^ (i & 7) -- every ds_read_b128 lane group then
// covers sixteen distinct 16-byte slots of the 256-byte bank row.  One raw s_barrier per K step; the loads are
// retired by a counted vmcnt (this stage's 10 done, the next stage's 10 still flying), never by a full drain.
template <int BM>
__global__ __launch_bounds__(256, 1) void bf16_gemm_glds_kernel(const GemmArgs a) {
  constexpr int MT = BM / 32;
  constexpr int STAGE = (BM + BN) * BK;          // bf16 elements per stage
  constexpr int NA = BM * 8 / 256, NB = BN * 8 / 256;     // 16-byte chunks per thread and stage: 6 + 4
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int bo = blockIdx.z / a.inner, bi = blockIdx.z - bo * a.inner;
  const unsigned short *P = a.P + bo * a.spo + bi * a.spi, *Q = a.Q + bo * a.sqo + bi * a.sqi;

  f32x4 acc[MT][4];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_main = a.K / BK, nk = nk_main + (a.R + BK - 1) / BK;
  // per-thread source rows / chunks of the stage image (fixed over the K walk)
  const unsigned short *srcA[NA], *srcA2[NA], *srcB[NB], *srcB2[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int q = tid + 256 * j, row = q >> 3, c = (q & 7) ^ (row & 7);
    const size_t r = (size_t)min(m0 + row, a.M - 1);
    srcA[j] = P + r * a.ldp + c * 8;
    srcA2[j] = a.R ? a.P2 + r * a.ldp2 + c * 8 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int q = tid + 256 * j, row = q >> 3, c = (q & 7) ^ (row & 7);
    const size_t r = (size_t)min(n0 + row, a.N - 1);
    srcB[j] = Q + r * a.ldq + c * 8;
    srcB2[j] = a.R ? a.Q2 + r * a.ldq2 + c * 8 : nullptr;
  }
  auto issue = [&](int kt) {                     // stage kt -> LDS buffer kt % 3
    unsigned short *buf = lds + (kt % 3) * STAGE;
    const bool main = kt < nk_main;
    const int k0 = main ? kt * BK : (kt - nk_main) * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((main ? srcA[j] : srcA2[j]) + k0),
                                       (__attribute__((address_space(3))) void *)(buf + (256 * j + 64 * wave) * 8), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((main ? srcB[j] : srcB2[j]) + k0),
                                       (__attribute__((address_space(3))) void *)(buf + BM * BK + (256 * j + 64 * wave) * 8), 16, 0, 0);
  };
  issue(0);
  if (nk > 1) issue(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // stage kt landed; kt + 1 may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // ... for every wave; and every wave is done reading stage kt - 1
    if (kt + 2 < nk) issue(kt + 2);             // (into the buffer stage kt - 1 occupied)
    const unsigned short *As = lds + (kt % 3) * STAGE, *Bs = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int pos = ((4 * ks + g) ^ (i & 7)) * 8;
      bf16x8 fa[MT], fb[4];
#pragma unroll
      for (int y = 0; y < 4; ++y) fb[y] = *reinterpret_cast<const bf16x8 *>(Bs + (wn * 64 + y * 16 + i) * BK + pos);
#pragma unroll
      for (int x = 0; x < MT; ++x) fa[x] = *reinterpret_cast<const bf16x8 *>(As + (wm * (BM / 2) + x * 16 + i) * BK + pos);
#pragma unroll
      for (int x = 0; x < MT; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[y], fa[x], acc[x][y], 0, 0, 0);
    }
  }
  // epilogue: D = Q P^T -- lane (i, g) holds columns n = 4 g + r of row m = i
#pragma unroll
  for (int x = 0; x < MT; ++x) {
    const int row = m0 + wm * (BM / 2) + x * 16 + i;
    if (row >= a.M) continue;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int col = n0 + wn * 64 + y * 16 + 4 * g;
      if (col >= a.N) continue;
      const float v0 = acc[x][y][0] * a.scale, v1 = acc[x][y][1] * a.scale, v2 = acc[x][y][2] * a.scale,
                  v3 = acc[x][y][3] * a.scale;
      const size_t o = (size_t)(bo * a.sco + bi * a.sci) + (size_t)row * a.ldc + col;
      if (col + 3 < a.N) {
        if (a.c_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.C) + o) = make_float4(v0, v1, v2, v3);
        else *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(a.C) + o) =
            make_uint2(f2bf(v0) | ((unsigned)f2bf(v1) << 16), f2bf(v2) | ((unsigned)f2bf(v3) << 16));
      } else {
        const float vv[4] = {v0, v1, v2, v3};
        for (int r = 0; r < 4 && col + r < a.N; ++r) {
          if (a.c_f32) reinterpret_cast<float *>(a.C)[o + r] = vv[r];
          else reinterpret_cast<unsigned short *>(a.C)[o + r] = f2bf(vv[r]);
        }
      }
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

template <int BM>
int launch_gemm(const GemmArgs &a, int batch, hipStream_t st) {
  constexpr size_t lds = sizeof(unsigned short) * (BM + BN) * LD;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_gemm_kernel<BM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return (int)attr;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, batch);
  bf16_gemm_kernel<BM><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}

template <int BM>
int launch_gemm_glds(const GemmArgs &a, int batch, hipStream_t st) {
  constexpr size_t lds = sizeof(unsigned short) * 3 * (BM + BN) * BK;         // 192: 122,880 B
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_gemm_glds_kernel<BM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return (int)attr;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, batch);
  bf16_gemm_glds_kernel<BM><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}

std::atomic<int> g_gemm_path{-1};                // -1: read MSR3D_BF16_GEMM (glds | reg) on first use

int gemm_dispatch(GemmArgs &a, int batch, hipStream_t st) {
  int path = g_gemm_path.load(std::memory_order_relaxed);
  if (path < 0) {
    const char *e = getenv("MSR3D_BF16_GEMM");
    path = (e && e[0] == 'r') ? 0 : 1;
    g_gemm_path.store(path, std::memory_order_relaxed);
  }
  // LDS-DMA path: rows >= 16 bytes-aligned sources (checked by the callers) and R a multiple of 64 (a stage
  // reads 64 columns of the low-rank pair)
  if (path == 1 && a.M >= 192 && (a.R % BK) == 0) return launch_gemm_glds<192>(a, batch, st);
  // tile height: the one that wastes fewer padded rows; 192 on a tie at >= 192 rows (fewer LDS reads per MFMA)
  const long long w128 = (long long)((a.M + 127) / 128) * 128, w192 = (long long)((a.M + 191) / 192) * 192;
  return (a.M >= 192 && w192 <= w128) ? launch_gemm<192>(a, batch, st) : launch_gemm<128>(a, batch, st);
}

}  // namespace

extern "C" {

int msr3d_bf16_gemm_lowrank(int M, int N, int K, int R, const void *P, int ldp, const void *Q, int ldq,
                            const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                            int c_f32, float scale, msr3d_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || R < 0 || (K % BK) != 0 || (R % 8) != 0) return MSR3D_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!P || !Q || !C || ldp < K || ldq < K || ldc < N || (ldp % 8) || (ldq % 8) || !al16(P) || !al16(Q))
    return MSR3D_EINVAL;
  if ((ldc % 4) || (reinterpret_cast<uintptr_t>(C) & 15u)) return MSR3D_EINVAL;
  if (R > 0 && (!P2 || !Q2 || ldp2 < R || ldq2 < R || (ldp2 % 8) || (ldq2 % 8) || !al16(P2) || !al16(Q2)))
    return MSR3D_EINVAL;
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.R = R;
  a.P = (const unsigned short *)P; a.ldp = ldp; a.Q = (const unsigned short *)Q; a.ldq = ldq;
  a.P2 = (const unsigned short *)P2; a.ldp2 = ldp2; a.Q2 = (const unsigned short *)Q2; a.ldq2 = ldq2;
  a.C = C; a.ldc = ldc; a.c_f32 = c_f32; a.scale = scale;
  a.inner = 1;
  a.spo = a.spi = a.sqo = a.sqi = a.sco = a.sci = 0;
  return gemm_dispatch(a, 1, (hipStream_t)stream);
}

int msr3d_bf16_gemm_batched(int outer, int inner, int M, int N, int K, const void *P, int ldp, long long p_outer,
                            long long p_inner, const void *Q, int ldq, long long q_outer, long long q_inner, void *C,
                            int ldc, long long c_outer, long long c_inner, int c_f32, float scale,
                            msr3d_stream_t stream) {
  if (outer < 0 || inner <= 0 || M < 0 || N < 0 || K < 0 || (K % BK) != 0) return MSR3D_EINVAL;
  if (outer == 0 || M == 0 || N == 0) return 0;
  if ((long long)outer * inner > 65535 || !P || !Q || !C || ldp < K || ldq < K || ldc < N || (ldp % 8) || (ldq % 8) ||
      !al16(P) || !al16(Q))
    return MSR3D_EINVAL;
  if ((p_outer % 8) || (p_inner % 8) || (q_outer % 8) || (q_inner % 8) || (c_outer % 4) || (c_inner % 4) || (ldc % 4) ||
      (reinterpret_cast<uintptr_t>(C) & 15u))
    return MSR3D_EINVAL;
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.R = 0;
  a.P = (const unsigned short *)P; a.ldp = ldp; a.Q = (const unsigned short *)Q; a.ldq = ldq;
  a.P2 = nullptr; a.ldp2 = 0; a.Q2 = nullptr; a.ldq2 = 0;
  a.C = C; a.ldc = ldc; a.c_f32 = c_f32; a.scale = scale;
  a.inner = inner;
  a.spo = p_outer; a.spi = p_inner; a.sqo = q_outer; a.sqi = q_inner; a.sco = c_outer; a.sci = c_inner;
  return gemm_dispatch(a, outer * inner, (hipStream_t)stream);
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
