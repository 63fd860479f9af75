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
#include <type_traits>

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

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct GemmArgs {
  int M, N, K, R;
  const unsigned short *P; int ldp;
  const unsigned short *Q; int ldq;
  const unsigned short *P2; int ldp2;
  const unsigned short *Q2; int ldq2;
  void *C; int ldc; int c_f32;
  int accumulate;                          // wide kernel, bf16 C only: C += the product
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

// ---- the wide-tile kernel: (16 MTB) x 256 tile, 8 waves side by side along N ----------------------------------
// The 4-wave kernels above keep one wave per SIMD in lockstep: every LDS-DMA piece a wave issues (~60-180 cycles
// of issue each, 10 per K step) and every wait at the K step's barrier is time its SIMD's matrix pipe idles.  Here
// a workgroup is 8 waves = two per SIMD, each owning ALL the tile's rows (MTB MFMA tiles) x 32 columns, so one
// wave's staging runs under the other's MFMAs, and the tile height is chosen per problem from {128, 144, 160} so
// that the tile count lands on whole rounds of the 256 CUs: the language model's M = 2304 tokens x N = 4096 is
// 16 x 16 = 256 tiles of 144 x 256 -- one per CU, one round (192 x 128: 384 tiles, 1.5).
// Three LDS stages of (BM + 256) x 64 bf16 filled by LDS-DMA (source-side XOR swizzle as above), two K steps in
// flight, one raw s_barrier per K step, counted vmcnt; the pieces of the next-but-one stage are issued between
// the MFMAs.  Tiles are numbered so that each XCD (blockIdx % 8) owns a compact 8 (M) x 4 (N) patch of them.
//
// Measured on 2304 x 4096 x 4096 (tools/bench_bf16_gemm.py, ablation builds of this file): 85 us = 900 TFLOP/s;
// alone, the fragment reads take 30 us (the LDS delivers ~156 B/clk to 8 waves reading 22 KB each per K step),
// the MFMAs 27 us (= the matrix pipe's rate), the staging 25 us, over a 20 us floor (launch, 64 barriers, 19 MB
// of C) -- and the three overlap little: all waves read, then all multiply.  A finer pipeline (ring of six
// 32-deep units, next unit's fragments read under this unit's MFMAs) was built and measured SLOWER (104 us):
// 64-byte rows double the DMA's cache-line requests (staging alone 61 -> 85 us) and twice the barriers eat what
// the read/MFMA overlap gives.
constexpr int WBN = 256;

template <int MTB>
__global__ __launch_bounds__(512, 1) void bf16_gemm_wide_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int BM = 16 * MTB;
  constexpr int STAGE = (BM + WBN) * BK;                 // bf16 elements per stage
  constexpr int PIECES = (BM + WBN) / 8;                 // 1 KB wave-pieces per stage (8 rows x 128 B)
  constexpr int NP = (PIECES + 7) / 8;                   // per wave, at most
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;

  // tile of this workgroup: XCD x = id % 8 takes a contiguous run of the patch order (4-tile-wide panels, M-major)
  int tm, tn;
  {
    const int T = tiles_m * tiles_n, id = blockIdx.x;
    const int q = T / 8, r = T % 8, x = id % 8;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
    const int per_panel = tiles_m * 4, p = t / per_panel, rem = t - p * per_panel;
    const int w = min(4, tiles_n - 4 * p);
    tm = rem / w;
    tn = 4 * p + rem - tm * w;
  }
  const int m0 = tm * BM, n0 = tn * WBN;
  const int bo = blockIdx.z / a.inner, bi = blockIdx.z - bo * a.inner;
  const unsigned short *P = a.P + bo * a.spo + bi * a.spi, *Q = a.Q + bo * a.sqo + bi * a.sqi;

  f32x4 acc[MTB][2];
#pragma unroll
  for (int x = 0; x < MTB; ++x) acc[x][0] = acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_main = a.K / BK, nk = nk_main + (a.R + BK - 1) / BK;
  // this wave's pieces: piece = wave + 8 j; rows 8 piece .. + 7 of the stage image (first BM rows: P, then Q)
  const int np = (PIECES - wave + 7) / 8;                // wave-uniform
  unsigned off1[NP], off2[NP];                           // element offsets of this lane's 16 bytes, main / low-rank pass
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int piece = min(wave + 8 * j, PIECES - 1), row = 8 * piece + (lane >> 3), c = (lane & 7) ^ (row & 7);
    const bool isA = row < BM;
    const int r = isA ? min(m0 + row, a.M - 1) : min(n0 + row - BM, a.N - 1);
    off1[j] = (unsigned)r * (unsigned)(isA ? a.ldp : a.ldq) + c * 8;
    off2[j] = (unsigned)r * (unsigned)(isA ? a.ldp2 : a.ldq2) + c * 8;
  }
  auto issue_piece = [&](int kt, int j) {                // piece j of stage kt -> LDS buffer kt % 3
    const bool main = kt < nk_main;
    const int k0 = main ? kt * BK : (kt - nk_main) * BK;
    const bool isA = wave + 8 * j < BM / 8;              // (wave-uniform)
    const unsigned short *base = isA ? (main ? P : a.P2) : (main ? Q : a.Q2);
    const unsigned short *src = base + (main ? off1[j] : off2[j]) + k0;
    unsigned short *dst = lds + (kt % 3) * STAGE + (wave + 8 * j) * 512;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (j < np) issue_piece(kt, j);
  };
  // Two wave groups half a K step apart (waves 0-3 = group A, 4-7 = group B: one of each per SIMD).  Every wave
  // runs the same stream  read(kt) -> multiply(kt);  group A meets the step's barrier BEFORE its reads, group B
  // BETWEEN its reads and its multiplies (its barrier kt + 1: it reads stage kt while A multiplies it, and multiplies
  // it while A reads stage kt + 1).  So A's reads share the CU with B's MFMAs and the other way round -- with all
  // eight waves in step, read time and multiply time simply added up (85 us for 27 us of MFMA).  Both groups pass
  // the barrier nk times; a wave issues its pieces of stage kt + 2 (A) / kt + 3 (B) among its multiplies of stage
  // kt, i.e. after the barrier that certifies everyone is done with the buffer they go into.
  const int grp_b = wave >= 4 ? 1 : 0;           // (wave-uniform)
  auto wait_stage = [&](bool last) {              // this wave's pieces of the stage about to be certified have landed
    if (!last) {                                  // (the stage after it may still fly)
      if (np == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  issue(0);
  if (nk > 1) issue(1);
  if (grp_b) {                                    // barrier 0 (stage 0 landed); B runs one stage further ahead
    if (nk > 2) issue(2);
    if (nk > 2) {
      if (np == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NP - 1)) : "memory");
    } else {
      wait_stage(nk == 1);
    }
    __builtin_amdgcn_s_barrier();
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (!grp_b) {                                 // A: barrier kt
      wait_stage(kt + 1 >= nk);
      __builtin_amdgcn_s_barrier();
    }
    const unsigned short *As = lds + (kt % 3) * STAGE, *Bs = As + BM * BK;
    bf16x8 fa[2][MTB], fb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pos = ((4 * ks + g) ^ (i & 7)) * 8;
#pragma unroll
      for (int y = 0; y < 2; ++y) fb[ks][y] = *reinterpret_cast<const bf16x8 *>(Bs + (wave * 32 + y * 16 + i) * BK + pos);
#pragma unroll
      for (int x = 0; x < MTB; ++x) fa[ks][x] = *reinterpret_cast<const bf16x8 *>(As + (x * 16 + i) * BK + pos);
    }
    if (grp_b && kt + 1 < nk) {                   // B: barrier kt + 1 -- its reads of stage kt are in registers
      wait_stage(kt + 2 >= nk);
      __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0), as a builtin: the compiler's scoreboard must see it
      __builtin_amdgcn_s_barrier();
    }
    const int st = kt + 2 + grp_b;                // the stage whose buffer the last barrier has freed
    const bool more = st < nk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int x = 0; x < MTB; ++x) {
        acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][0], fa[ks][x], acc[x][0], 0, 0, 0);
        acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][1], fa[ks][x], acc[x][1], 0, 0, 0);
        const int slot = ks * MTB + x;            // 0 .. 2 MTB - 1: one piece every other slot
        if (more && (slot & 1) == 0 && slot / 2 < NP && slot / 2 < np) issue_piece(st, slot / 2);
      }
    }
  }
  // epilogue: D = Q P^T -- lane (i, g) holds columns n = 4 g + r of row m = i
#pragma unroll
  for (int x = 0; x < MTB; ++x) {
    const int row = m0 + x * 16 + i;
    if (row >= a.M) continue;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int col = n0 + wave * 32 + y * 16 + 4 * g;
      if (col >= a.N) continue;
      float v0 = acc[x][y][0] * a.scale, v1 = acc[x][y][1] * a.scale, v2 = acc[x][y][2] * a.scale,
            v3 = acc[x][y][3] * a.scale;
      const size_t o = (size_t)(bo * a.sco + bi * a.sci) + (size_t)row * a.ldc + col;
      if (a.accumulate) {                  // (bf16 C, N % 4 == 0: checked by the caller)
        const uint2 old = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(a.C) + o);
        v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
      }
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

// out (R, C) (or its transpose) += scale * sum_m P[m][r] * Q[m][c]      (the LoRA weight gradients: the token
// reduction of a (M, 16) against a (M, 4096..11008) bf16 matrix).  Bound by reading Q once (19-51 MB), so the rows
// are cut into enough chunks to put a workgroup on every CU (up to kGradChunks): a lane owns 128 / R adjacent
// columns (one 16-byte load per row at R = 16) and walks its chunk with R x 8 fp32 accumulators in registers, eight
// rows' loads in flight; the chunk's P values are broadcast from LDS as float4.  The chunks' partial results go to
// a workspace as plain stores and a second launch adds them up in chunk order (deterministic; 64 atomicAdds per
// output element were 4 M memory-side atomics) -- without a workspace: 16 chunks and atomicAdd.
// (Round 2's column-per-thread kernel, one 2-byte load and 16 LDS reads per row and thread: 112 us at 2304 x 4096.)
constexpr int kGradChunks = 64, kGradChunksAtomic = 16;


template <int R>
__global__ __launch_bounds__(256) void lora_grad_kernel(int M, int C, const unsigned short *__restrict__ P, int ldp,
                                                        const unsigned short *__restrict__ Q, int ldq,
                                                        float *__restrict__ out, int transpose_out, float scale,
                                                        float *__restrict__ ws) {
  constexpr int CPL = 128 / R;                         // columns per lane: 8 (R = 16) or 4 (R = 32)
  constexpr int ROWS = 64;                             // rows of P staged per pass
  constexpr int PF = 8;                                // rows in flight
  __shared__ __attribute__((aligned(16))) float ps[ROWS][R];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 4 + wave) * 64 * CPL + lane * CPL;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int mb = blockIdx.y * per, me = min(M, mb + per);
  const bool live = c0 < C;                            // (C is a multiple of CPL: checked by the caller)
  using qvec = typename std::conditional<CPL == 8, uint4, uint2>::type;
  float acc[R][CPL];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[r][j] = 0.f;
  for (int m0 = mb; m0 < me; m0 += ROWS) {
    __syncthreads();
    for (int e = threadIdx.x; e < ROWS * R; e += 256) {
      const int mm = e / R, r = e - mm * R;
      ps[mm][r] = (m0 + mm < me) ? bf_lo(P[(size_t)(m0 + mm) * ldp + r]) : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const int lim = min(ROWS, me - m0);
    const unsigned short *q = Q + (size_t)m0 * ldq + c0;
    for (int mb8 = 0; mb8 < lim; mb8 += PF) {
      qvec v[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u)                     // (rows past the chunk re-read its last row: their P is zero)
        v[u] = *reinterpret_cast<const qvec *>(q + (size_t)min(mb8 + u, lim - 1) * ldq);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int mm = mb8 + u;                        // (< ROWS: ROWS is a multiple of PF)
        float qv[CPL];
        if constexpr (CPL == 8) {
          qv[0] = bf_lo(v[u].x); qv[1] = bf_hi(v[u].x); qv[2] = bf_lo(v[u].y); qv[3] = bf_hi(v[u].y);
          qv[4] = bf_lo(v[u].z); qv[5] = bf_hi(v[u].z); qv[6] = bf_lo(v[u].w); qv[7] = bf_hi(v[u].w);
        } else {
          qv[0] = bf_lo(v[u].x); qv[1] = bf_hi(v[u].x); qv[2] = bf_lo(v[u].y); qv[3] = bf_hi(v[u].y);
        }
        const float live_row = mm < lim ? 1.f : 0.f;
#pragma unroll
        for (int r4 = 0; r4 < R; r4 += 4) {
          const float4 pv = *reinterpret_cast<const float4 *>(&ps[mm][r4]);
          const float pp[4] = {pv.x * live_row, pv.y * live_row, pv.z * live_row, pv.w * live_row};
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[r4 + k][j] = fmaf(pp[k], qv[j], acc[r4 + k][j]);
        }
      }
    }
  }
  if (!live) return;
  if (ws) {                                            // partial (chunk, R, C), plain 16-byte stores
    float *w = ws + (size_t)blockIdx.y * R * C + c0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < CPL; j += 4)
        *reinterpret_cast<float4 *>(w + (size_t)r * C + j) = make_float4(acc[r][j], acc[r][j + 1], acc[r][j + 2], acc[r][j + 3]);
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      atomicAdd(out + (transpose_out ? (size_t)(c0 + j) * R + r : (size_t)r * C + c0 + j), acc[r][j] * scale);
}

// The same reduction on the matrix pipe (round 4; the workspace path).  D[c][r] = sum_m Q[m][c] P[m][r] contracts over
// the ROW index of both operands, so an MFMA fragment (8 consecutive tokens of one column) is a strided gather: a
// workgroup stages 64 tokens x 256 columns of Q (and the tokens' R values of P) row-major in LDS with coalesced 16-byte
// loads -- all of a pass's loads in flight together -- and each wave gathers its fragments with eight 2-byte LDS reads
// (32 contiguous bytes per 16-lane row group).  4 waves x 64 columns; one v_mfma_f32_16x16x32_bf16 per (column tile,
// r tile, 32 tokens).  The arithmetic is ~20 k MFMAs per call: the kernel is bound by reading Q once.  Round 3's VALU
// kernel (a lane = 8 columns x R accumulators, P broadcast from LDS) spent 28 us per 2304 x 4096 call on 128
// workgroups; the 14 calls of a decoder layer were 14 % of its forward + backward.
constexpr int kGQ = 256 + 8;       // LDS pitch of a staged Q row, bf16 units (528 B: 16-byte aligned)
template <int R>
__global__ __launch_bounds__(256) void lora_grad_mfma_kernel(int M, int C, const unsigned short *__restrict__ P, int ldp,
                                                             const unsigned short *__restrict__ Q, int ldq,
                                                             float *__restrict__ ws, int per) {
  constexpr int RT = R / 16;
  constexpr int GP = R + 2;                            // LDS pitch of a staged P row (36 / 68 B: 4-byte aligned)
  __shared__ __attribute__((aligned(16))) unsigned short qs[64 * kGQ];
  __shared__ __attribute__((aligned(16))) unsigned short ps[64 * GP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int cb = blockIdx.x * 256;
  const int mb = blockIdx.y * per, me = min(M, mb + per);
  f32x4 acc[4][RT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < RT; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging roles: Q -- thread t covers 16 bytes (8 columns) of row t / 32 + 8 j; P -- R / 8 chunks of 16 bytes per row
  const int qrow = tid >> 5, qc = (tid & 31) * 8;
  const bool qlive = cb + qc < C;                      // (C % 8 == 0: a chunk is inside or outside)
  constexpr int PCH = R / 8;                           // 16-byte chunks per P row
  const int prow = tid / PCH, pc = (tid % PCH) * 8;
  uint4 qv[8], pv;
  auto fetch = [&](int m0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + qrow + 8 * j;
      qv[j] = (qlive && m < me) ? *reinterpret_cast<const uint4 *>(Q + (size_t)m * ldq + cb + qc) : make_uint4(0, 0, 0, 0);
    }
    const int m = m0 + prow;
    pv = (prow < 64 && m < me) ? *reinterpret_cast<const uint4 *>(P + (size_t)m * ldp + pc) : make_uint4(0, 0, 0, 0);
  };
  fetch(mb);
  for (int m0 = mb; m0 < me; m0 += 64) {
    __syncthreads();                                   // the previous pass's fragments have been read
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4 *>(qs + (qrow + 8 * j) * kGQ + qc) = qv[j];
    if (prow < 64) {
      unsigned *d = reinterpret_cast<unsigned *>(ps + prow * GP + pc);
      d[0] = pv.x; d[1] = pv.y; d[2] = pv.z; d[3] = pv.w;
    }
    __syncthreads();
    if (m0 + 64 < me) fetch(m0 + 64);                  // the next pass's loads fly under this pass's gathers
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int tk = 32 * sl + 8 * g;                  // this lane's eight tokens
      bf16x8 fb[RT];
#pragma unroll
      for (int u = 0; u < RT; ++u) {
        union { unsigned short h[8]; bf16x8 v; } w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w.h[e] = ps[(tk + e) * GP + 16 * u + i];
        fb[u] = w.v;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union { unsigned short h[8]; bf16x8 v; } w;
#pragma unroll
        for (int e = 0; e < 8; ++e) w.h[e] = qs[(tk + e) * kGQ + wave * 64 + 16 * t + i];
#pragma unroll
        for (int u = 0; u < RT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, fb[u], acc[t][u], 0, 0, 0);
      }
    }
  }
  // D[c][r]: lane (i, g) of tile (t, u) holds columns c = 16 t + 4 g + q (q = 0..3) of r = 16 u + i
  float *w = ws + (size_t)blockIdx.y * R * C;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = cb + wave * 64 + 16 * t + 4 * g;
    if (c >= C) continue;                              // (C % 4 == 0)
#pragma unroll
    for (int u = 0; u < RT; ++u)
      *reinterpret_cast<float4 *>(w + (size_t)(16 * u + i) * C + c) = make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
  }
}

// out (+)= scale * sum over the chunks' partials, in chunk order; thread = 4 adjacent columns of one r
__global__ __launch_bounds__(256) void lora_grad_reduce_kernel(int R, int C, int chunks, const float *__restrict__ ws,
                                                               float *__restrict__ out, int transpose_out, float scale,
                                                               int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;       // over R * C / 4
  if (e >= R * (C / 4)) return;
  const int r = e / (C / 4), c = 4 * (e - r * (C / 4));
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float *w = ws + (size_t)r * C + c;
  constexpr int U = 16;                                // loads in flight (one dependent load per chunk: 19 us)
  for (int k0 = 0; k0 < chunks; k0 += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4 *>(w + (size_t)min(k0 + u, chunks - 1) * R * C);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k0 + u < chunks) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  const float keep = accumulate ? 1.f : 0.f;           // (accumulate == 0: `out` is written, whatever it held)
  if (transpose_out) {
    float *o = out + (size_t)c * R + r;
    const float o0 = accumulate ? o[0] : 0.f, o1 = accumulate ? o[R] : 0.f, o2 = accumulate ? o[2 * R] : 0.f,
                o3 = accumulate ? o[3 * R] : 0.f;
    o[0] = o0 * keep + s.x * scale; o[R] = o1 * keep + s.y * scale; o[2 * R] = o2 * keep + s.z * scale;
    o[3 * R] = o3 * keep + s.w * scale;
  } else {
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)r * C + c);
    float4 t = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
    t.x += s.x * scale; t.y += s.y * scale; t.z += s.z * scale; t.w += s.w * scale;
    *o = t;
  }
}

// dA and dB of ONE LoRA pair in ONE launch, no partial sums at all (round 4 cut the rows into 64 chunks for
// parallelism, wrote the chunks' partials to a workspace and added them up in a second launch: four launches per pair,
// 896 a step for a 32-layer stack, 8.5 ms of an 83 ms step; a first one-launch version -- tickets + agent-scope fences, the
// last workgroup of a column block adding the partials -- measured 80-150 us per pair against 37-41: an L2 write-back per
// wave).  Here a workgroup OWNS 64 output columns over ALL rows: 64 + 172 = 236 workgroups for a 4096 / 11008 pair, one
// per CU; the sum over the tokens never leaves the accumulators, so the result is bit-reproducible by construction.  One
// CU's share of the bandwidth needs ~64 KB in flight: a register ring of kGD passes (64 rows x 128 bytes each), refilled
// as it drains; the pass's rows go row-major into LDS (double-buffered: one barrier per pass) and each wave gathers the
// fragments of its 16 columns with the LDS transpose read (an MFMA fragment here is 8 consecutive TOKENS of one column:
// ds_read_b64_tr_b16 hands a 16-lane group's [4 tokens][16 columns] block out column-wise).  (Measured and dropped: 32
// columns per workgroup where 64-column blocks number fewer than the CUs -- 4096 + 4096 columns are 128 -- with 128-row
// passes and the two token halves met in LDS: 15.2 us against 14.1.)
struct GradJob {
  int C;
  const unsigned short *P; int ldp;          // (M, R)
  const unsigned short *Q; int ldq;          // (M, C)
  float *out; int transpose_out;
  int blocks;                                // 64-column blocks of this job
};
struct GradPair { GradJob j[2]; };
using v4s = __attribute__((ext_vector_type(4))) short;
__device__ __forceinline__ v4s tr_read(const unsigned short *p) {          // ds_read_b64_tr_b16
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)p);
}
constexpr int kGD = 8;                       // passes in flight per workgroup
constexpr int kGC = 64 + 8;                  // LDS pitch of a staged Q row, bf16 units (144 B)

template <int R>
__global__ __launch_bounds__(256) void lora_grad_cols_kernel(int M, const GradPair pr, float scale, int accumulate) {
  constexpr int RT = R / 16;
  constexpr int GP = R + 4;                  // LDS pitch of a staged P row (40 / 72 B: 8-byte aligned for the transpose reads)
  constexpr int PCH = R / 8;                 // 16-byte chunks per P row
  const int z = (int)blockIdx.x >= pr.j[0].blocks ? 1 : 0;
  const GradJob &jb = pr.j[z];
  const int C = jb.C;
  const int cb = 64 * ((int)blockIdx.x - (z ? pr.j[0].blocks : 0));
  __shared__ __attribute__((aligned(16))) unsigned short qs[2][64 * kGC];
  __shared__ __attribute__((aligned(16))) unsigned short ps[2][64 * GP];
  const unsigned short *__restrict__ P = jb.P;
  const unsigned short *__restrict__ Q = jb.Q;
  const int ldp = jb.ldp, ldq = jb.ldq;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc[RT];
#pragma unroll
  for (int u = 0; u < RT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging roles: Q -- thread t covers 16 bytes (8 columns) of rows t / 8 and 32 + t / 8; P -- threads 0 .. 64 PCH - 1
  const int qrow = tid >> 3, qc = (tid & 7) * 8;
  const bool qlive = cb + qc < C;            // (C % 8 == 0: a 16-byte piece is inside or outside)
  const int prow = tid / PCH, pc = (tid % PCH) * 8;
  const int np = (M + 63) >> 6;
  uint4 qv[kGD][2], pv[kGD];
  auto fetch = [&](int d, int pass) {
    const int m0 = pass * 64;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + qrow + 32 * h;
      qv[d][h] = (qlive && m < M) ? *reinterpret_cast<const uint4 *>(Q + (size_t)m * ldq + cb + qc) : make_uint4(0, 0, 0, 0);
    }
    const int m = m0 + prow;
    pv[d] = (prow < 64 && m < M) ? *reinterpret_cast<const uint4 *>(P + (size_t)m * ldp + pc) : make_uint4(0, 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < kGD; ++d) fetch(d, d);                // (passes beyond the last read nothing: zeros)
  for (int base = 0; base < np; base += kGD) {
#pragma unroll
    for (int d = 0; d < kGD; ++d) {
      const int pass = base + d;
      if (pass < np) {                                      // (uniform)
        unsigned short *q_ = qs[d & 1], *p_ = ps[d & 1];    // (kGD is even: pass parity == d parity)
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<uint4 *>(q_ + (qrow + 32 * h) * kGC + qc) = qv[d][h];
        if (prow < 64) {
          uint2 *dst = reinterpret_cast<uint2 *>(p_ + prow * GP + pc);
          dst[0] = make_uint2(pv[d].x, pv[d].y); dst[1] = make_uint2(pv[d].z, pv[d].w);
        }
        fetch(d, pass + kGD);                               // refill the ring slot just drained
        __syncthreads();                                    // (the other buffer's readers finished before the last barrier)
        // fragments by the LDS transpose read: a 16-lane group hands in the addresses of a [4 tokens][16 columns] block
        // (lane i': token i' >> 2, columns 4 (i' & 3) ..) and lane i receives column i of it -- four consecutive tokens
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int tr = 32 * sl + 8 * g + (i >> 2), tc = 4 * (i & 3);
          union { v4s h[2]; bf16x8 v; } wq;
#pragma unroll
          for (int h = 0; h < 2; ++h) wq.h[h] = tr_read(q_ + (tr + 4 * h) * kGC + wave * 16 + tc);
#pragma unroll
          for (int u = 0; u < RT; ++u) {
            union { v4s h[2]; bf16x8 v; } wp;
#pragma unroll
            for (int h = 0; h < 2; ++h) wp.h[h] = tr_read(p_ + (tr + 4 * h) * GP + 16 * u + tc);
            acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq.v, wp.v, acc[u], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[c][r]: lane (i, g) holds columns c = cb + 16 wave + 4 g + q (q = 0..3) of r = 16 u + i
  const int c = cb + wave * 16 + 4 * g;
  if (c >= C) return;                                       // (C % 4 == 0)
  float *out = jb.out;
#pragma unroll
  for (int u = 0; u < RT; ++u) {
    const int r = 16 * u + i;
    if (jb.transpose_out) {
      float *o = out + (size_t)c * R + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q * R] = (accumulate ? o[q * R] : 0.f) + acc[u][q] * scale;
    } else {
      float4 *o = reinterpret_cast<float4 *>(out + (size_t)r * C + c);
      float4 t = accumulate ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
      t.x += acc[u][0] * scale; t.y += acc[u][1] * scale; t.z += acc[u][2] * scale; t.w += acc[u][3] * scale;
      *o = t;
    }
  }
}

// The bf16 images of EVERY LoRA pair of a stack in the four orientations its products read, in one launch (once per
// optimiser step; round 4 issued four conversion launches per pair: 896 a step):
//   a_pad (r, K) = bf16(A)         bt_pad (r, N) = bf16(B^T)
//   at2   (K, 64)[:, :r] = bf16(A^T)    b2 (N, 64)[:, :r] = bf16(B)     (columns r..63 are zero and are not touched)
// blockIdx.y = pair; a thread owns one column k of A (then one row n of B).
template <int R>
__device__ __forceinline__ void shadow_pair(const msr3d_lora_shadow_job_t &jb) {
  const int K = jb.K, N = jb.N;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < K + N; t += gridDim.x * 256) {
    if (t < K) {
      const int k = t;
      unsigned w[R / 2];
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        const unsigned lo = f2bf(jb.A[(size_t)r * K + k]), hi = f2bf(jb.A[(size_t)(r + 1) * K + k]);
        jb.a_pad[(size_t)r * K + k] = (unsigned short)lo;
        jb.a_pad[(size_t)(r + 1) * K + k] = (unsigned short)hi;
        w[r / 2] = lo | (hi << 16);
      }
      uint4 *d = reinterpret_cast<uint4 *>(jb.at2 + (size_t)k * 64);
#pragma unroll
      for (int q = 0; q < R / 8; ++q) d[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    } else {
      const int n = t - K;
      const float4 *src = reinterpret_cast<const float4 *>(jb.B + (size_t)n * R);
      unsigned w[R / 2];
#pragma unroll
      for (int q = 0; q < R / 4; ++q) {
        const float4 v = src[q];
        const unsigned b0 = f2bf(v.x), b1 = f2bf(v.y), b2 = f2bf(v.z), b3 = f2bf(v.w);
        w[2 * q] = b0 | (b1 << 16);
        w[2 * q + 1] = b2 | (b3 << 16);
        jb.bt_pad[(size_t)(4 * q) * N + n] = (unsigned short)b0;
        jb.bt_pad[(size_t)(4 * q + 1) * N + n] = (unsigned short)b1;
        jb.bt_pad[(size_t)(4 * q + 2) * N + n] = (unsigned short)b2;
        jb.bt_pad[(size_t)(4 * q + 3) * N + n] = (unsigned short)b3;
      }
      uint4 *d = reinterpret_cast<uint4 *>(jb.b2 + (size_t)n * 64);
#pragma unroll
      for (int q = 0; q < R / 8; ++q) d[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
  }
}
__global__ __launch_bounds__(256) void lora_shadows_kernel(const msr3d_lora_shadow_job_t *__restrict__ jobs) {
  const msr3d_lora_shadow_job_t jb = jobs[blockIdx.y];
  if (jb.r == 16) shadow_pair<16>(jb);
  else shadow_pair<32>(jb);
}

// C (M, N <= 64) = scale * P Q^T for a Q of a few rows (the LoRA down-projections x A^T and dy B: N = 16 of
// 64 padded columns), and C[:, N:zero_to] = 0.  A tile of the big kernels would put this on 18 CUs; here a
// workgroup owns 16 rows of P, its four waves each a quarter of K, fragments loaded straight from global memory
// (P is read once; the 16 x K operand Q stays in L2), and the four partial tiles meet in LDS.
// waves per workgroup, each a share of K: sixteen (four per SIMD) while their partial tiles fit 32 KB of LDS (N <= 32), eight
// otherwise (round 3: four).  2304 tokens are only 144 workgroups: what a CU has in flight is what its own waves issue.
template <int NT> constexpr int kSkinnyWaves = NT <= 2 ? 16 : 8;
// (Measured and dropped: 8 rows per workgroup, half of the MFMA tile empty, so that 2304 tokens are 288 workgroups
// instead of 144 on 256 CUs -- 19 us against 13 per call.)
// QUANT: the same pass also leaves the e4m3 image of P with one scale per row (= msr3d_quant_rows_fp8, bit for bit) --
// the frozen-weight product that follows reads exactly this tensor, and its own quantisation launch read it a second
// time: the rows' absolute maxima are taken from the fragments as they go by (integer maxima of the bf16 bit patterns),
// met across the waves in LDS, and a second walk over the wave's K slice (L2 hits) converts and stores 8 bytes per lane.
template <int NT, bool QUANT = false>
__global__ __launch_bounds__(64 * kSkinnyWaves<NT>) void bf16_gemm_skinny_kernel(int M, int N, int K, const unsigned short *__restrict__ P,
                                                               int ldp, const unsigned short *__restrict__ Q, int ldq,
                                                               unsigned short *__restrict__ C, int ldc, int zero_to,
                                                               float scale, unsigned char *__restrict__ q8 = nullptr,
                                                               int ldq8 = 0, float *__restrict__ row_scale = nullptr) {
  constexpr int NW = kSkinnyWaves<NT>;
  __shared__ __attribute__((aligned(16))) float red[NW][NT][64][4];
  __shared__ unsigned rmax[NW][16];
  unsigned amax = 0;                                   // two 15-bit maxima of |bf16| bit patterns, packed
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int nks = K / 32, ks0 = wave * nks / NW, ks1 = (wave + 1) * nks / NW;
  const unsigned short *p = P + (size_t)min(m0 + i, M - 1) * ldp + 8 * g;
  const unsigned short *q[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) q[t] = Q + (size_t)min(16 * t + i, N - 1) * ldq + 8 * g;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // batches of U K steps, the NEXT batch's loads issued before this batch's MFMAs (round 3 loaded a batch, waited a full
  // memory latency, multiplied, and started over: 8 latencies per 16 rows x 4096 -- 15 us for 19 MB on 144 workgroups)
  constexpr int U = 4;
  struct Batch { bf16x8 fp[U], fq[U][NT]; };
  auto fetch = [&](Batch &b_, int ks) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      b_.fp[u] = *reinterpret_cast<const bf16x8 *>(p + (ks + u) * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) b_.fq[u][t] = *reinterpret_cast<const bf16x8 *>(q[t] + (ks + u) * 32);
    }
  };
  auto track = [&](const bf16x8 &f) {                  // |x| as integers: for non-negative floats the orders agree
    if constexpr (QUANT) {
      union { bf16x8 v; unsigned w[4]; } x;
      x.v = f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned a = x.w[e] & 0x7fff7fffu;
        amax = (max(amax >> 16, a >> 16) << 16) | max(amax & 0xffffu, a & 0xffffu);
      }
    }
  };
  auto mma = [&](const Batch &b_) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      track(b_.fp[u]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_.fq[u][t], b_.fp[u], acc[t], 0, 0, 0);
    }
  };
  const int nb = (ks1 - ks0) / U;
  int ks = ks0;
  if (nb > 0) {
    Batch ba, bb;
    fetch(ba, ks0);
    for (int b2 = 0; b2 < nb; b2 += 2) {
      if (b2 + 1 < nb) fetch(bb, ks0 + (b2 + 1) * U);
      mma(ba);
      if (b2 + 2 < nb) fetch(ba, ks0 + (b2 + 2) * U);
      if (b2 + 1 < nb) mma(bb);
    }
    ks = ks0 + nb * U;
  }
  for (; ks < ks1; ++ks) {
    const bf16x8 fp = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
    track(fp);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(q[t] + ks * 32), fp, acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(&red[wave][t][lane][0]) = acc[t];
  constexpr int QU = 8;                                // 16-byte pieces per thread and round of the second walk
  constexpr int TPR = 4 * NW, QS = 8 * TPR;            // threads per row of the second walk; elements per row and step
  const int qr = threadIdx.x / TPR, qc = (threadIdx.x % TPR) * 8;
  const unsigned short *src = P + (size_t)min(m0 + qr, M - 1) * ldp + qc;
  uint4 xv[QUANT ? QU : 1];
  if constexpr (QUANT) {
    unsigned m = max(amax >> 16, amax & 0xffffu);
    m = max(m, (unsigned)__shfl_xor((int)m, 16));
    m = max(m, (unsigned)__shfl_xor((int)m, 32));
    if (g == 0) rmax[wave][i] = m;
    // the second walk's first round of loads leaves BEFORE the barrier (it needs the scale only to convert)
#pragma unroll
    for (int u = 0; u < QU; ++u) xv[u] = QS * u + qc < K ? *reinterpret_cast<const uint4 *>(src + QS * u) : make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  if constexpr (QUANT) {
    unsigned m = rmax[0][i];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = max(m, rmax[w][i]);
    const float mx = __uint_as_float(m << 16);
    const float sc = mx > 0.f ? mx * (1.0f / 448.0f) : 1.0f;
    if (m0 + i < M && wave == 0 && g == 0) row_scale[m0 + i] = sc;
    // second walk, ROW-major over the workgroup's 16 x K block (L2 hits): 32 threads per row, 16 bytes each -- 512
    // contiguous bytes read and 256 written per row and step (the fragment order of pass 1 would store 32-byte pieces)
    unsigned mr = rmax[0][qr];
#pragma unroll
    for (int w = 1; w < NW; ++w) mr = max(mr, rmax[w][qr]);
    const float mxr = __uint_as_float(mr << 16);
    const float inv = 1.0f / (mxr > 0.f ? mxr * (1.0f / 448.0f) : 1.0f);
    const bool live = m0 + qr < M;
    unsigned char *dst = q8 + (size_t)(m0 + qr) * ldq8 + qc;
    for (int c0 = 0; c0 < K; c0 += QS * QU) {
      if (c0 > 0) {
#pragma unroll
        for (int u = 0; u < QU; ++u)
          xv[u] = c0 + QS * u + qc < K ? *reinterpret_cast<const uint4 *>(src + c0 + QS * u) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int c = c0 + QS * u + qc;
        const unsigned w4[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w4[0] << 16) * inv, __uint_as_float(w4[0] & 0xffff0000u) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w4[1] << 16) * inv, __uint_as_float(w4[1] & 0xffff0000u) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w4[2] << 16) * inv, __uint_as_float(w4[2] & 0xffff0000u) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(w4[3] << 16) * inv, __uint_as_float(w4[3] & 0xffff0000u) * inv, hi, true);
        if (live && c < K) *reinterpret_cast<uint2 *>(dst + c0 + QS * u) = make_uint2((unsigned)lo, (unsigned)hi);
      }
    }
  }
  // D = Q P^T: lane (i, g) of tile t holds columns n = 16 t + 4 g + r of row m = i; wave w finishes tiles w, w + 4 ..
  const int row = m0 + i;
  for (int t = wave; t < NT; t += NW) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(&red[0][t][lane][0]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 o = *reinterpret_cast<const f32x4 *>(&red[w][t][lane][0]);
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    const int col = 16 * t + 4 * g;
    if (row < M && col < N)              // (N is a multiple of 4)
      *reinterpret_cast<uint2 *>(C + (size_t)row * ldc + col) =
          make_uint2(f2bf(v[0] * scale) | ((unsigned)f2bf(v[1] * scale) << 16),
                     f2bf(v[2] * scale) | ((unsigned)f2bf(v[3] * scale) << 16));
  }
  // the padding columns the next product reads as part of its K step
  for (int e = threadIdx.x; e < 16 * ((zero_to - N) / 4); e += 64 * NW) {
    const int r = e / ((zero_to - N) / 4), c = N + 4 * (e - r * ((zero_to - N) / 4));
    if (m0 + r < M) *reinterpret_cast<uint2 *>(C + (size_t)(m0 + r) * ldc + c) = make_uint2(0u, 0u);
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

template <int MTB>
int launch_gemm_wide(const GemmArgs &a, int batch, hipStream_t st) {
  constexpr int BM = 16 * MTB;
  constexpr size_t lds = sizeof(unsigned short) * 3 * (BM + WBN) * BK;        // 144: 153,600 B
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_gemm_wide_kernel<MTB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return (int)attr;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + WBN - 1) / WBN;
  bf16_gemm_wide_kernel<MTB><<<dim3(tiles_m * tiles_n, 1, batch), 512, lds, st>>>(a, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

std::atomic<int> g_gemm_path{-1};                // -1: read MSR3D_BF16_GEMM (wide | glds | reg) on first use

int gemm_dispatch(GemmArgs &a, int batch, hipStream_t st) {
  int path = g_gemm_path.load(std::memory_order_relaxed);
  if (path < 0) {
    const char *e = getenv("MSR3D_BF16_GEMM");
    path = (e && e[0] == 'r') ? 0 : (e && e[0] == 'g') ? 1 : 2;
    g_gemm_path.store(path, std::memory_order_relaxed);
  }
  // wide tiles: big products whose offsets fit the kernel's 32-bit element offsets; tile height = the one with the
  // least (rounds of 256 CUs) x height
  if (path == 2 && a.M >= 128 && a.N >= 256 && (a.R % BK) == 0 &&
      (long long)a.M * a.ldp < (1ll << 31) && (long long)a.N * a.ldq < (1ll << 31) &&
      (a.R == 0 || ((long long)a.M * a.ldp2 < (1ll << 31) && (long long)a.N * a.ldq2 < (1ll << 31)))) {
    const int tn = (a.N + WBN - 1) / WBN;
    long long best = -1;
    int bm = 0;
    for (int h : {160, 144, 128}) {
      const long long tiles = (long long)((a.M + h - 1) / h) * tn * batch, cost = (tiles + 255) / 256 * h;
      if (best < 0 || cost < best) best = cost, bm = h;
    }
    // (fewer tiles than half the chip: the 128-wide tiles of the 4-wave kernels spread the product further)
    if ((long long)((a.M + bm - 1) / bm) * tn * batch >= 128)
      return bm == 160 ? launch_gemm_wide<10>(a, batch, st) : bm == 144 ? launch_gemm_wide<9>(a, batch, st)
                                                                        : launch_gemm_wide<8>(a, batch, st);
  }
  // LDS-DMA path: rows >= 16 bytes-aligned sources (checked by the callers) and R a multiple of 64 (a stage
  // reads 64 columns of the low-rank pair)
  if (path >= 1 && a.M >= 192 && (a.R % BK) == 0) return launch_gemm_glds<192>(a, batch, st);
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
  a.accumulate = 0;
  a.inner = 1;
  a.spo = a.spi = a.sqo = a.sqi = a.sco = a.sci = 0;
  return gemm_dispatch(a, 1, (hipStream_t)stream);
}

int msr3d_bf16_gemm_lowrank_acc(int M, int N, int K, int R, const void *P, int ldp, const void *Q, int ldq,
                                const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc, float scale,
                                msr3d_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || R < 0 || (K % BK) != 0 || (R % BK) != 0 || (N % 4) != 0) return MSR3D_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!P || !Q || !C || ldp < K || ldq < K || ldc < N || (ldp % 8) || (ldq % 8) || !al16(P) || !al16(Q))
    return MSR3D_EINVAL;
  if ((ldc % 4) || (reinterpret_cast<uintptr_t>(C) & 15u)) return MSR3D_EINVAL;
  if (R > 0 && (!P2 || !Q2 || ldp2 < R || ldq2 < R || (ldp2 % 8) || (ldq2 % 8) || !al16(P2) || !al16(Q2)))
    return MSR3D_EINVAL;
  // the wide-tile kernel only (the language model's products): its domain, as gemm_dispatch states it
  if (M < 128 || N < 256 || (long long)M * ldp >= (1ll << 31) || (long long)N * ldq >= (1ll << 31) ||
      (R > 0 && ((long long)M * ldp2 >= (1ll << 31) || (long long)N * ldq2 >= (1ll << 31))))
    return MSR3D_EINVAL;
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.R = R;
  a.P = (const unsigned short *)P; a.ldp = ldp; a.Q = (const unsigned short *)Q; a.ldq = ldq;
  a.P2 = (const unsigned short *)P2; a.ldp2 = ldp2; a.Q2 = (const unsigned short *)Q2; a.ldq2 = ldq2;
  a.C = C; a.ldc = ldc; a.c_f32 = 0; a.scale = scale;
  a.accumulate = 1;
  a.inner = 1;
  a.spo = a.spi = a.sqo = a.sqi = a.sco = a.sci = 0;
  const int tn = (N + WBN - 1) / WBN;
  long long best = -1;
  int bm = 0;
  for (int h : {160, 144, 128}) {
    const long long tiles = (long long)((M + h - 1) / h) * tn, cost = (tiles + 255) / 256 * h;
    if (best < 0 || cost < best) best = cost, bm = h;
  }
  hipStream_t st = (hipStream_t)stream;
  return bm == 160 ? launch_gemm_wide<10>(a, 1, st) : bm == 144 ? launch_gemm_wide<9>(a, 1, st) : launch_gemm_wide<8>(a, 1, st);
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
  a.accumulate = 0;
  a.inner = inner;
  a.spo = p_outer; a.spi = p_inner; a.sqo = q_outer; a.sqi = q_inner; a.sco = c_outer; a.sci = c_inner;
  return gemm_dispatch(a, outer * inner, (hipStream_t)stream);
}

int msr3d_lora_grad(int M, int R, int C, const void *P, int ldp, const void *Q, int ldq, float *out,
                    int transpose_out, float scale, int accumulate, float *workspace, long long workspace_floats,
                    msr3d_stream_t stream) {
  if (M < 0 || C < 0 || (R != 16 && R != 32)) return MSR3D_EINVAL;
  if (M == 0 || C == 0) return 0;
  if (!P || !Q || !out || ldp < R || ldq < C) return MSR3D_EINVAL;
  if (!accumulate && !workspace) return MSR3D_EINVAL;     // (the atomic path can only add)
  const int cpl = 128 / R;                      // a lane's columns: one 16- / 8-byte load per row
  if ((C % cpl) || (C % 4) || (ldq % cpl) || (reinterpret_cast<uintptr_t>(Q) & (2 * cpl - 1))) return MSR3D_EINVAL;
  if (workspace && ((reinterpret_cast<uintptr_t>(workspace) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u)))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int chunks;
  if (workspace && (C % 8) == 0 && (ldq % 8) == 0 && (ldp % 8) == 0 && al16(P) && al16(Q)) {
    // matrix-pipe kernel: chunks of whole 64-token passes, at most kGradChunks of them
    int per = (M + kGradChunks - 1) / kGradChunks;
    per = (per + 63) / 64 * 64;
    chunks = (M + per - 1) / per;
    if (workspace_floats < (long long)chunks * R * C) return MSR3D_EINVAL;
    dim3 grid((C + 255) / 256, chunks);
    if (R == 16)
      lora_grad_mfma_kernel<16><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                                      workspace, per);
    else
      lora_grad_mfma_kernel<32><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                                      workspace, per);
  } else {
    chunks = (M + 31) / 32;
    const int cap = workspace ? kGradChunks : kGradChunksAtomic;
    if (chunks > cap) chunks = cap;
    if (workspace && workspace_floats < (long long)chunks * R * C) return MSR3D_EINVAL;
    dim3 grid((C + 256 * cpl - 1) / (256 * cpl), chunks);
    if (R == 16)
      lora_grad_kernel<16><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                                 out, transpose_out, scale, workspace);
    else
      lora_grad_kernel<32><<<grid, 256, 0, st>>>(M, C, (const unsigned short *)P, ldp, (const unsigned short *)Q, ldq,
                                                 out, transpose_out, scale, workspace);
  }
  if (workspace)
    lora_grad_reduce_kernel<<<(R * (C / 4) + 255) / 256, 256, 0, st>>>(R, C, chunks, workspace, out, transpose_out, scale,
                                                                       accumulate);
  return (int)hipGetLastError();
}

int msr3d_lora_grad_pair(int M, int R, int njobs, const msr3d_lora_grad_job_t *jobs, float scale, int accumulate,
                         msr3d_stream_t stream) {
  if (M < 0 || (R != 16 && R != 32) || njobs < 1 || njobs > 2 || !jobs) return MSR3D_EINVAL;
  if (M == 0) return 0;
  GradPair pr{};
  int total = 0;
  for (int z = 0; z < njobs; ++z) {
    const msr3d_lora_grad_job_t &j = jobs[z];
    if (j.C <= 0 || (j.C % 8) || !j.P || !j.Q || !j.out || j.ldp < R || j.ldq < j.C || (j.ldp % 8) || (j.ldq % 8) ||
        !al16(j.P) || !al16(j.Q) || (reinterpret_cast<uintptr_t>(j.out) & 15u))
      return MSR3D_EINVAL;
    GradJob &g = pr.j[z];
    g.C = j.C; g.P = (const unsigned short *)j.P; g.ldp = j.ldp; g.Q = (const unsigned short *)j.Q; g.ldq = j.ldq;
    g.out = j.out; g.transpose_out = j.transpose_out;
    g.blocks = (j.C + 63) / 64;
    total += g.blocks;
  }
  hipStream_t st = (hipStream_t)stream;
  if (R == 16) lora_grad_cols_kernel<16><<<total, 256, 0, st>>>(M, pr, scale, accumulate);
  else lora_grad_cols_kernel<32><<<total, 256, 0, st>>>(M, pr, scale, accumulate);
  return (int)hipGetLastError();
}

int msr3d_lora_shadows(int njobs, const msr3d_lora_shadow_job_t *jobs_device, msr3d_stream_t stream) {
  if (njobs < 0) return MSR3D_EINVAL;
  if (njobs == 0) return 0;
  if (!jobs_device) return MSR3D_EINVAL;
  lora_shadows_kernel<<<dim3(16, njobs), 256, 0, (hipStream_t)stream>>>(jobs_device);
  return (int)hipGetLastError();
}

static int skinny_launch(int M, int N, int K, const void *P, int ldp, const void *Q, int ldq, void *C, int ldc, int zero_to,
                         float scale, void *q8, int ldq8, float *row_scale, msr3d_stream_t stream) {
  if (M < 0 || N <= 0 || N > 64 || (N % 16) || K <= 0 || (K % 32) || zero_to < N || (zero_to % 4) || zero_to > ldc)
    return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!P || !Q || !C || ldp < K || ldq < K || ldc < N || (ldp % 8) || (ldq % 8) || (ldc % 4) || !al16(P) || !al16(Q) ||
      (reinterpret_cast<uintptr_t>(C) & 7u))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *p = (const unsigned short *)P, *q = (const unsigned short *)Q;
  unsigned short *c = (unsigned short *)C;
  const int grid = (M + 15) / 16;
  if (q8) {
    unsigned char *q8p = (unsigned char *)q8;
    switch (N / 16) {
      case 1: bf16_gemm_skinny_kernel<1, true><<<grid, 64 * kSkinnyWaves<1>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale, q8p, ldq8, row_scale); break;
      case 2: bf16_gemm_skinny_kernel<2, true><<<grid, 64 * kSkinnyWaves<2>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale, q8p, ldq8, row_scale); break;
      case 3: bf16_gemm_skinny_kernel<3, true><<<grid, 64 * kSkinnyWaves<3>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale, q8p, ldq8, row_scale); break;
      default: bf16_gemm_skinny_kernel<4, true><<<grid, 64 * kSkinnyWaves<4>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale, q8p, ldq8, row_scale); break;
    }
    return (int)hipGetLastError();
  }
  switch (N / 16) {
    case 1: bf16_gemm_skinny_kernel<1><<<grid, 64 * kSkinnyWaves<1>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale); break;
    case 2: bf16_gemm_skinny_kernel<2><<<grid, 64 * kSkinnyWaves<2>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale); break;
    case 3: bf16_gemm_skinny_kernel<3><<<grid, 64 * kSkinnyWaves<3>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale); break;
    default: bf16_gemm_skinny_kernel<4><<<grid, 64 * kSkinnyWaves<4>, 0, st>>>(M, N, K, p, ldp, q, ldq, c, ldc, zero_to, scale); break;
  }
  return (int)hipGetLastError();
}

int msr3d_bf16_gemm_skinny(int M, int N, int K, const void *P, int ldp, const void *Q, int ldq, void *C, int ldc,
                           int zero_to, float scale, msr3d_stream_t stream) {
  return skinny_launch(M, N, K, P, ldp, Q, ldq, C, ldc, zero_to, scale, nullptr, 0, nullptr, stream);
}

int msr3d_bf16_gemm_skinny_quant(int M, int N, int K, const void *P, int ldp, const void *Q, int ldq, void *C, int ldc,
                                 int zero_to, float scale, void *q8, int ldq8, float *row_scale, msr3d_stream_t stream) {
  if (!q8 || !row_scale || ldq8 < K || (ldq8 % 8) || (reinterpret_cast<uintptr_t>(q8) & 7u)) return MSR3D_EINVAL;
  return skinny_launch(M, N, K, P, ldp, Q, ldq, C, ldc, zero_to, scale, q8, ldq8, row_scale, stream);
}

}  // extern "C"
