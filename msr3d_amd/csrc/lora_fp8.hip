// lora_fp8.hip -- the frozen projections of the LoRA-Llama layers with fp8 (OCP e4m3) operands on the MX matrix
// instruction (/root/reference/model/msr3d/msr3d.py:103-112,409-415: the LLM's seven projections per layer; SURVEY.md
// §8(f) rank 4 "bf16/fp8 GEMMs"):
//
//     y = diag(sx) (Xq Wq^T) diag(sw) + (s x A^T) B^T        Xq, Wq e4m3; sx per token row, sw per output channel
//
// The frozen W is quantised ONCE per checkpoint (per-output-channel scale, both orientations: W for the forward, W^T for
// dx), the activations per call (msr3d_quant_rows_fp8: a row's absolute maximum -> scale, round to nearest even);
// the LoRA pair, its activations and every accumulator stay bf16 / fp32.
//
// The product is the wide-tile kernel of lora_linear.hip -- (16 MTB) x 256 tile, 8 waves side by side along N, three
// LDS stages filled by LDS-DMA with the source-side XOR swizzle, two wave groups half a K step apart, counted vmcnt,
// one raw s_barrier per K step -- with the SAME byte geometry: a K step is 128 e4m3 = 128 bytes per row, as 64 bf16
// were, so the stage image, the pieces and the swizzle carry over unchanged; what changes is the arithmetic:
// v_mfma_scale_f32_16x16x128_f8f6f4 (both formats e4m3, block scales 2^0) consumes a whole K step of a 16 x 16 tile in
// one instruction at twice the bf16 rate, with half the LDS bytes and half the fragment reads per flop.
// The LoRA term rides as the FIRST stage (64 zero-padded bf16 = 128 bytes per row: same image) on the bf16
// instruction; its accumulators are then divided by sx[m] sw[n] so that the fp8 partial sums can be added on top and the
// epilogue's multiplication by sx[m] sw[n] restores it -- no second accumulator set (72 more registers a lane).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/msr3d_hip.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WBN = 256;                 // tile width
constexpr int ROWB = 128;                // bytes of one row of a stage image (128 e4m3 / 64 bf16)
constexpr int kOne = 0x7f7f7f7f;         // four E8M0 block scales of 2^0

__device__ __forceinline__ unsigned short f2bf(float f) {        // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- rows of a bf16 matrix -> e4m3 with one scale per row ---------------------------------------------------
// one wave per row, 8 columns (16 bytes) per lane and pass, the row held in registers between the two passes
// (K <= 64 * 8 * MAXP); |x| / s <= 448 = the largest e4m3, so the conversion never saturates
constexpr int MAXP = 24;                 // K <= 12,288
__global__ __launch_bounds__(256) void quant_rows_kernel(int M, int K, const unsigned short *__restrict__ x, int ldx,
                                                         unsigned char *__restrict__ q, int ldq, float *__restrict__ scale) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const unsigned short *src = x + (size_t)row * ldx;
  const int np = (K + 511) / 512;
  uint4 v[MAXP];
  float m = 0.f;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p < np) {
      const int c = p * 512 + lane * 8;
      v[p] = c < K ? *reinterpret_cast<const uint4 *>(src + c) : make_uint4(0, 0, 0, 0);
      const unsigned w[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(bf_lo(w[e])), fabsf(bf_hi(w[e]))));
    }
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  const float s = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
  unsigned char *dst = q + (size_t)row * ldq;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p < np) {
      const int c = p * 512 + lane * 8;
      if (c < K) {
        const unsigned w[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w[0]) * inv, bf_hi(w[0]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w[1]) * inv, bf_hi(w[1]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w[2]) * inv, bf_hi(w[2]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(w[3]) * inv, bf_hi(w[3]) * inv, hi, true);
        *reinterpret_cast<uint2 *>(dst + c) = make_uint2((unsigned)lo, (unsigned)hi);
      }
    }
  }
}

struct Fp8Args {
  int M, N, K, R;
  const unsigned char *P; int ldp; const float *sp;       // (M, K) e4m3 + row scales
  const unsigned char *Q; int ldq; const float *sq;       // (N, K) e4m3 + row scales
  const unsigned short *P2; int ldp2;                      // (M, 64) bf16: s x A^T, zero-padded
  const unsigned short *Q2; int ldq2;                      // (N, 64) bf16: B, zero-padded
  unsigned short *C; int ldc;                              // (M, N) bf16
  int accumulate;                                          // C += the product (one more bf16 rounding per call)
};

template <int MTB>
__global__ __launch_bounds__(512, 1) void fp8_gemm_wide_kernel(const Fp8Args a, int tiles_m, int tiles_n) {
  constexpr int BM = 16 * MTB;
  constexpr int STAGE = (BM + WBN) * ROWB;               // bytes per stage
  constexpr int PIECES = (BM + WBN) / 8;                 // 1 KB wave-pieces per stage (8 rows x 128 B)
  constexpr int NP = (PIECES + 7) / 8;                   // per wave, at most
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
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

  f32x4 acc[MTB][2];
#pragma unroll
  for (int x = 0; x < MTB; ++x) acc[x][0] = acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // stages: [the LoRA pair (bf16) if R > 0] then K / 128 fp8 steps
  const int lr = a.R > 0 ? 1 : 0;
  const int nk = lr + a.K / 128;
  const int np = (PIECES - wave + 7) / 8;                // wave-uniform
  unsigned off1[NP];                                     // BYTE offsets of this lane's 16 bytes of the fp8 operands
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int piece = min(wave + 8 * j, PIECES - 1), row = 8 * piece + (lane >> 3), c = (lane & 7) ^ (row & 7);
    const bool isA = row < BM;
    const int r = isA ? min(m0 + row, a.M - 1) : min(n0 + row - BM, a.N - 1);
    off1[j] = (unsigned)r * (unsigned)(isA ? a.ldp : a.ldq) + c * 16;
  }
  auto issue_piece = [&](int kt, int j) {                // piece j of fp8 stage kt (>= lr) -> LDS buffer kt % 3
    const bool isA = wave + 8 * j < BM / 8;              // (wave-uniform)
    const unsigned char *src = (isA ? a.P : a.Q) + off1[j] + (size_t)(kt - lr) * 128;
    unsigned char *dst = lds + (kt % 3) * STAGE + (wave + 8 * j) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };
  auto issue_low = [&]() {                               // stage 0 = the LoRA pair, bf16 (prologue only: offsets not kept)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (j >= np) continue;
      const bool isA = wave + 8 * j < BM / 8;
      const int piece = min(wave + 8 * j, PIECES - 1), row = 8 * piece + (lane >> 3), c = (lane & 7) ^ (row & 7);
      const int r = isA ? min(m0 + row, a.M - 1) : min(n0 + row - BM, a.N - 1);
      const unsigned char *src = reinterpret_cast<const unsigned char *>(isA ? a.P2 : a.Q2) +
                                 ((size_t)r * (isA ? a.ldp2 : a.ldq2)) * 2 + c * 16;
      unsigned char *dst = lds + (wave + 8 * j) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };
  auto issue = [&](int kt) {
    if (kt < lr) { issue_low(); return; }
#pragma unroll
    for (int j = 0; j < NP; ++j)
      if (j < np) issue_piece(kt, j);
  };
  // the two wave groups and their barriers: exactly lora_linear.hip's bf16_gemm_wide_kernel (see there)
  const int grp_b = wave >= 4 ? 1 : 0;
  auto wait_stage = [&](bool last) {
    if (!last) {
      if (np == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  issue(0);
  if (nk > 1) issue(1);
  if (grp_b) {
    if (nk > 2) issue(2);
    if (nk > 2) {
      if (np == NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NP - 1)) : "memory");
    } else {
      wait_stage(nk == 1);
    }
    __builtin_amdgcn_s_barrier();
  }
  // one K step; LOW (compile time): the LoRA stage -- peeled out of the loop, or the accumulators of the two
  // instruction kinds meet in phi nodes and hipcc stops updating them in place (72 more registers: spills)
  auto step = [&](int kt, auto low_tag) {
    constexpr bool low = decltype(low_tag)::value;
    if (!grp_b) {                                 // A: barrier kt
      wait_stage(kt + 1 >= nk);
      __builtin_amdgcn_s_barrier();
    }
    const unsigned char *As = lds + (kt % 3) * STAGE, *Bs = As + BM * ROWB;
    // a lane's 32 bytes of a row: chunks 2 g and 2 g + 1 (fp8: k = 32 g .. + 31; bf16: ks = 0 -> chunk g, ks = 1 -> 4 + g)
    const int c0 = low ? g : 2 * g, c1 = low ? 4 + g : 2 * g + 1;
    const int p0 = (c0 ^ (i & 7)) * 16, p1 = (c1 ^ (i & 7)) * 16;
    union Frag { i32x8 v; i32x4 h[2]; };                 // a lane's 32 bytes of one row: the MX operand as it is
    Frag fa[MTB], fb[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      fb[y].h[0] = *reinterpret_cast<const i32x4 *>(Bs + (wave * 32 + y * 16 + i) * ROWB + p0);
      fb[y].h[1] = *reinterpret_cast<const i32x4 *>(Bs + (wave * 32 + y * 16 + i) * ROWB + p1);
    }
#pragma unroll
    for (int x = 0; x < MTB; ++x) {
      fa[x].h[0] = *reinterpret_cast<const i32x4 *>(As + (x * 16 + i) * ROWB + p0);
      fa[x].h[1] = *reinterpret_cast<const i32x4 *>(As + (x * 16 + i) * ROWB + p1);
    }
    if (grp_b && kt + 1 < nk) {                   // B: barrier kt + 1 -- its reads of stage kt are in registers
      wait_stage(kt + 2 >= nk);
      __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0)
      __builtin_amdgcn_s_barrier();
    }
    const int st = kt + 2 + grp_b;                // the stage whose buffer the last barrier has freed
    const bool more = st < nk;
    if constexpr (low) {
      // ---- the LoRA pair: two bf16 K steps of 32, then parked under the fp8 sums as lora / (sp[m] sq[n])
#pragma unroll
      for (int x = 0; x < MTB; ++x) {
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[y].h[0]), __builtin_bit_cast(bf16x8, fa[x].h[0]), acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[y].h[1]), __builtin_bit_cast(bf16x8, fa[x].h[1]), acc[x][y], 0, 0, 0);
        }
        if (more && x < NP && x < np) issue_piece(st, x);
      }
      float4 rsn[2];
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int col = min(n0 + wave * 32 + y * 16 + 4 * g, a.N - 4);
        const float4 s = *reinterpret_cast<const float4 *>(a.sq + col);
        rsn[y] = make_float4(1.0f / s.x, 1.0f / s.y, 1.0f / s.z, 1.0f / s.w);
      }
#pragma unroll
      for (int x = 0; x < MTB; ++x) {
        const float rs = 1.0f / a.sp[min(m0 + x * 16 + i, a.M - 1)];
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          acc[x][y][0] *= rs * rsn[y].x; acc[x][y][1] *= rs * rsn[y].y;
          acc[x][y][2] *= rs * rsn[y].z; acc[x][y][3] *= rs * rsn[y].w;
        }
      }
    } else {
#pragma unroll
      for (int x = 0; x < MTB; ++x) {
        acc[x][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[0].v, fa[x].v, acc[x][0], 0, 0, 0, kOne, 0, kOne);
        acc[x][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[1].v, fa[x].v, acc[x][1], 0, 0, 0, kOne, 0, kOne);
        if (more && x < NP && x < np) issue_piece(st, x);
      }
    }
  };
  if (lr) step(0, std::true_type{});
  for (int kt = lr; kt < nk; ++kt) step(kt, std::false_type{});
  // epilogue: D = Q P^T -- lane (i, g) holds columns n = 4 g + r of row m = i; y = acc * sp[m] * sq[n]
#pragma unroll
  for (int x = 0; x < MTB; ++x) {
    const int row = m0 + x * 16 + i;
    if (row >= a.M) continue;
    const float sm = a.sp[row];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int col = n0 + wave * 32 + y * 16 + 4 * g;
      if (col >= a.N) continue;                          // (N % 4 == 0: checked by the caller)
      const float4 sn = *reinterpret_cast<const float4 *>(a.sq + col);
      float v0 = acc[x][y][0] * sm * sn.x, v1 = acc[x][y][1] * sm * sn.y, v2 = acc[x][y][2] * sm * sn.z,
            v3 = acc[x][y][3] * sm * sn.w;
      if (a.accumulate) {
        const uint2 old = *reinterpret_cast<const uint2 *>(a.C + (size_t)row * a.ldc + col);
        v0 += bf_lo(old.x); v1 += bf_hi(old.x); v2 += bf_lo(old.y); v3 += bf_hi(old.y);
      }
      *reinterpret_cast<uint2 *>(a.C + (size_t)row * a.ldc + col) =
          make_uint2(f2bf(v0) | ((unsigned)f2bf(v1) << 16), f2bf(v2) | ((unsigned)f2bf(v3) << 16));
    }
  }
}

template <int MTB>
int launch(const Fp8Args &a, hipStream_t st) {
  constexpr int BM = 16 * MTB;
  constexpr size_t lds = (size_t)3 * (BM + WBN) * ROWB;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&fp8_gemm_wide_kernel<MTB>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return (int)attr;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + WBN - 1) / WBN;
  fp8_gemm_wide_kernel<MTB><<<dim3(tiles_m * tiles_n), 512, lds, st>>>(a, tiles_m, tiles_n);
  return (int)hipGetLastError();
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

int msr3d_quant_rows_fp8(int M, int K, const void *x, int ldx, void *q, int ldq, float *scale, msr3d_stream_t stream) {
  if (M < 0 || K <= 0 || K > 512 * MAXP || (K % 8) || (ldx % 8) || (ldq % 8) || ldx < K || ldq < K) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!x || !q || !scale || !al16(x) || (reinterpret_cast<uintptr_t>(q) & 7u)) return MSR3D_EINVAL;
  quant_rows_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(M, K, (const unsigned short *)x, ldx, (unsigned char *)q,
                                                                 ldq, scale);
  return (int)hipGetLastError();
}

static int fp8_gemm(int M, int N, int K, const void *Pq, int ldp, const float *sp, const void *Qq, int ldq,
                    const float *sq, const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc, int accumulate,
                    msr3d_stream_t stream) {
  if (M < 0 || N < 0 || K <= 0 || (K % 128) || (N % 4)) return MSR3D_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (M < 128 || N < 256) return MSR3D_EINVAL;             // (the wide tile's domain: the language model's products)
  if (!Pq || !Qq || !sp || !sq || !C || ldp < K || ldq < K || ldc < N || (ldp % 16) || (ldq % 16) || (ldc % 4) ||
      !al16(Pq) || !al16(Qq) || !al16(sq) || (reinterpret_cast<uintptr_t>(C) & 7u))
    return MSR3D_EINVAL;
  const bool low = P2 != nullptr || Q2 != nullptr;
  if (low && (!P2 || !Q2 || ldp2 < 64 || ldq2 < 64 || (ldp2 % 8) || (ldq2 % 8) || !al16(P2) || !al16(Q2))) return MSR3D_EINVAL;
  if ((long long)M * ldp >= (1ll << 31) || (long long)N * ldq >= (1ll << 31)) return MSR3D_EINVAL;
  Fp8Args a;
  a.M = M; a.N = N; a.K = K; a.R = low ? 64 : 0;
  a.P = (const unsigned char *)Pq; a.ldp = ldp; a.sp = sp;
  a.Q = (const unsigned char *)Qq; a.ldq = ldq; a.sq = sq;
  a.P2 = (const unsigned short *)P2; a.ldp2 = ldp2;
  a.Q2 = (const unsigned short *)Q2; a.ldq2 = ldq2;
  a.C = (unsigned short *)C; a.ldc = ldc;
  a.accumulate = accumulate;
  // tile height = the one with the least (rounds of 256 CUs) x height, as the bf16 kernel picks it
  const int tn = (N + WBN - 1) / WBN;
  long long best = -1;
  int bm = 0;
  for (int h : {160, 144, 128}) {
    const long long tiles = (long long)((M + h - 1) / h) * tn, cost = ((tiles + 255) / 256) * h;
    if (best < 0 || cost < best) { best = cost; bm = h; }
  }
  hipStream_t st = (hipStream_t)stream;
  return bm == 160 ? launch<10>(a, st) : bm == 144 ? launch<9>(a, st) : launch<8>(a, st);
}

int msr3d_fp8_gemm_lowrank(int M, int N, int K, const void *Pq, int ldp, const float *sp, const void *Qq, int ldq,
                           const float *sq, const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                           msr3d_stream_t stream) {
  return fp8_gemm(M, N, K, Pq, ldp, sp, Qq, ldq, sq, P2, ldp2, Q2, ldq2, C, ldc, 0, stream);
}

int msr3d_fp8_gemm_lowrank_acc(int M, int N, int K, const void *Pq, int ldp, const float *sp, const void *Qq, int ldq,
                               const float *sq, const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                               msr3d_stream_t stream) {
  return fp8_gemm(M, N, K, Pq, ldp, sp, Qq, ldq, sq, P2, ldp2, Q2, ldq2, C, ldc, 1, stream);
}

}  // extern "C"
