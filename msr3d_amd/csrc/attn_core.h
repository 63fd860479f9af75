// attn_core.h -- the device-side core of MultiHeadAttentionSpatial with 'cond' fusion
// (/root/reference/modules/layers/transformers.py:200-252) on LDS tiles: shared by the stand-alone
// launches of attn_spatial.hip (one workgroup per (sample, head)) and by the fused attention blocks of
// scene_block.hip, which produce q / k / v / cond in the same workgroup.  See attn_spatial.hip for the
// arithmetic and its citations.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace msr3d_attn {


using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DH = 32;        // head dim
constexpr int SD = 5;         // spatial dims
constexpr int LD32 = DH + 4;  // [LT][36] tiles
constexpr int kMaxL = 128;
constexpr float kSqrtDh = 5.656854249492381f;   // sqrt(32): s = dot / this (:205)
constexpr float kInvSqrtDh = 0.17677669529663687f;   // (multiplied: 1 ulp from the division, a tenth of its instructions)

// acc[rn] += sum_{k<KD} a(row0+i.., k) * b(rn*16+.., k) for this wave's 16-row strip.
// a(r,k) = A_KC ? As[r*lda + k] : As[k*lda + r];  b(c,k) likewise.
//   f32 : 16x16x4 operand map, lane (i = lane&15, g = lane>>4) supplies element (i, k0+g);
//   bf16 / fp8 : 16x16x32 operand map, lane (i, g) supplies elements (i, k0+8g .. k0+8g+7).
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <bool KC>
__device__ __forceinline__ void frag8(const float *S, int ld, int r, int k, float (&f)[8]) {
  if (KC) {
    const float4 lo = *reinterpret_cast<const float4 *>(S + r * ld + k);
    const float4 hi = *reinterpret_cast<const float4 *>(S + r * ld + k + 4);
    f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
    f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = S[(k + j) * ld + r];
  }
}

__device__ __forceinline__ unsigned bf16_rne(float v) {        // finite inputs (LDS tiles)
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ bf16x8 pack_bf16(const float (&f)[8]) {
  union { unsigned u[4]; bf16x8 v; } w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w.u[j] = bf16_rne(f[2 * j]) | (bf16_rne(f[2 * j + 1]) << 16);
  return w.v;
}
__device__ __forceinline__ long pack_fp8(const float (&f)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

// SWAP (f32 only): the two operands trade roles, D^T instead of D -- acc[rn][r] is then element (row = row0 + i,
// col = 16 rn + 4 g + r): a lane holds four CONSECUTIVE columns of one row (one 16-byte store, one sm_split4) instead of
// four rows of one column.  The same products summed over k in the same order.
template <int MMA, int RN, int KD, bool A_KC, bool B_KC, bool SWAP = false>
__device__ __forceinline__ void strip_mma(const float *As, int lda, const float *Bs, int ldb,
                                          int row0, f32x4 (&acc)[RN], int lane) {
  const int i = lane & 15, g = lane >> 4;
  static_assert(!SWAP || MMA != MSR3D_MMA_FP8, "swapped roles: f32 and bf16 operand maps");
  if (MMA == MSR3D_MMA_F32) {
#pragma unroll
    for (int k0 = 0; k0 < KD; k0 += 4) {
      const int k = k0 + g;
      const float a = A_KC ? As[(row0 + i) * lda + k] : As[k * lda + row0 + i];
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float b = B_KC ? Bs[(rn * 16 + i) * ldb + k] : Bs[k * ldb + rn * 16 + i];
        acc[rn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[rn], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[rn], 0, 0, 0);
      }
    }
  } else {
    static_assert(KD % 32 == 0, "16x16x32 operand map");
#pragma unroll
    for (int k0 = 0; k0 < KD; k0 += 32) {
      float fa[8], fb[8];
      frag8<A_KC>(As, lda, row0 + i, k0 + 8 * g, fa);
      if (MMA == MSR3D_MMA_BF16) {
        const bf16x8 a = pack_bf16(fa);
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          frag8<B_KC>(Bs, ldb, rn * 16 + i, k0 + 8 * g, fb);
          acc[rn] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack_bf16(fb), a, acc[rn], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pack_bf16(fb), acc[rn], 0, 0, 0);
        }
      } else {
        const long a = pack_fp8(fa);
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          frag8<B_KC>(Bs, ldb, rn * 16 + i, k0 + 8 * g, fb);
          acc[rn] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, pack_fp8(fb), acc[rn], 0, 0, 0);
        }
      }
    }
  }
}

// reductions over the 16 lanes of a DPP row (= one row group of the C/D layout)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL,
                                                    0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// token-major (B*L, ld) head slice -> LDS [LT][36], rows >= L zero
template <int LT>
__device__ __forceinline__ void load_head_tile(const float *__restrict__ src, int ld, int b, int h,
                                               int L, float *dst) {
  for (int e = threadIdx.x; e < LT * (DH / 4); e += LT * 4) {
    const int row = e >> 3, c4 = (e & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < L) v = *reinterpret_cast<const float4 *>(src + ((size_t)b * L + row) * ld + h * DH + c4);
    *reinterpret_cast<float4 *>(dst + row * LD32 + c4) = v;
  }
}

// pairwise_locs of one sample, (L, L, SD) floats, is one contiguous slab: the 64-token tile copies
// it into LDS with coalesced 16-byte loads (the logits loop reads 5 floats per (query, key) pair:
// straight from global memory that is 80 uncoalesced 4-byte loads per lane; from LDS the stride-5
// pattern is bank-conflict-free).  The 128-token tile (328 KB slab) keeps reading global memory.
template <int LT>
__device__ __forceinline__ const float *stage_ploc(const float *__restrict__ ploc, int b, int L,
                                                   float *spl) {
  const float *src = ploc + (size_t)b * L * L * SD;
  if (LT != 64) return src;
  const int n = L * L * SD, n4 = n >> 2;
  const bool vec = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  if (vec) {
#pragma unroll 4
    for (int e = threadIdx.x; e < n4; e += LT * 4)
      reinterpret_cast<float4 *>(spl)[e] = reinterpret_cast<const float4 *>(src)[e];
    for (int e = n4 * 4 + threadIdx.x; e < n; e += LT * 4) spl[e] = src[e];
  } else {
    for (int e = threadIdx.x; e < n; e += LT * 4) spl[e] = src[e];
  }
  return spl;      // valid after the caller's next __syncthreads()
}

// saved probabilities of one (scene, head), (L, L) dense -> LDS [LT][LT+4], zero outside
template <int LT>
__device__ __forceinline__ void load_probs_tile(const float *__restrict__ probs_bh, int L, float *sp) {
  constexpr int LDP = LT + 4;
  for (int e = threadIdx.x; e < LT * LT; e += LT * 4) {
    const int row = e / LT, col = e % LT;
    sp[row * LDP + col] = (row < L && col < L) ? probs_bh[(size_t)row * L + col] : 0.f;
  }
}

// The same two tiles fetched into REGISTERS first (the fused blocks issue these loads ahead of a matrix
// product whose operand still occupies the LDS they are headed for) and stored later.  64-token tile,
// 256 threads; `vec` (16-byte aligned source, element count % 4 == 0) is the caller's to check.
constexpr int kPlocRegs = 64 * 64 * SD / 4 / 256;       // 20 float4 per thread
__device__ __forceinline__ void ploc_fetch(const float *__restrict__ src, int n4, float4 (&v)[kPlocRegs]) {
#pragma unroll
  for (int k = 0; k < kPlocRegs; ++k) {
    const int e = threadIdx.x + 256 * k;
    v[k] = e < n4 ? reinterpret_cast<const float4 *>(src)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ploc_store(float *spl, int n4, const float4 (&v)[kPlocRegs]) {
#pragma unroll
  for (int k = 0; k < kPlocRegs; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e < n4) reinterpret_cast<float4 *>(spl)[e] = v[k];
  }
}
constexpr int kProbRegs = 64 * 64 / 4 / 256;            // 4 float4 per thread
__device__ __forceinline__ void probs_fetch(const float *__restrict__ probs_bh, int L, float4 (&v)[kProbRegs]) {
  const int n4 = (L * L) >> 2;
#pragma unroll
  for (int k = 0; k < kProbRegs; ++k) {
    const int e = threadIdx.x + 256 * k;
    v[k] = e < n4 ? reinterpret_cast<const float4 *>(probs_bh)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
// L % 4 == 0: a float4 never straddles a row; [64][68] tile, zero outside (L, L)
__device__ __forceinline__ void probs_store(float *sp, int L, const float4 (&v)[kProbRegs]) {
  constexpr int LDP = 64 + 4;
  const int n4 = (L * L) >> 2;
#pragma unroll
  for (int k = 0; k < kProbRegs; ++k) {
    const int e = threadIdx.x + 256 * k;
    if (e < n4) {
      const int idx = 4 * e, row = idx / L, col = idx - row * L;
      *reinterpret_cast<float4 *>(sp + row * LDP + col) = v[k];
    }
  }
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int row = e >> 6, col = e & 63;
    if (row >= L || col >= L) sp[row * LDP + col] = 0.f;
  }
}

struct RowCond { float bias, w[SD]; };

// `cond0`: the (bias, w[5]) sextet of row 0 of this (scene, head); rows are `ldc` floats apart
// (global: cond + b * L * ldc + h * 6; the fused block keeps them in LDS)
__device__ __forceinline__ RowCond load_cond(const float *cond0, int ldc, int L, int row) {
  RowCond c;
  c.bias = 0.f;
#pragma unroll
  for (int d = 0; d < SD; ++d) c.w[d] = 0.f;
  if (row < L) {
    const float *p = cond0 + (size_t)row * ldc;
    c.bias = p[0];
#pragma unroll
    for (int d = 0; d < SD; ++d) c.w[d] = p[1 + d];
  }
  return c;
}


// =================================================================================
// forward core.  LT/16 waves; wave w owns query rows [16w, 16w+16).  On entry sq / sk / sv hold the
// head's q, k, v tiles ([LT][36] fp32, rows >= L zero) and `plb` the scene's pairwise slab (LDS or
// global), all visible to every wave.  Leaves P in sp ([LT][LT+4]) and returns ctx = P V in o[2]
// (element (row = 16 wave + 4 g + r, col = 16 rn + i)), scaled by kPScale for fp8.
// `probs_bh`: this (scene, head)'s (L, L) slice of the saved probabilities, or null.
// =================================================================================
template <int LT, int MMA>
__device__ __forceinline__ void attn_fwd_core(int L, const float *sq, const float *sk, const float *sv, float *sp,
                                              const float *plb, const float *cond0, int ldc,
                                              const unsigned char *pad_b, float *probs_bh, f32x4 (&o)[2]) {
  constexpr int NT = LT / 16, LDP = LT + 4;
  constexpr float kPScale = (MMA == MSR3D_MMA_FP8) ? 256.f : 1.f;   // P into e4m3's normal range
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  // a workgroup may hold more waves than query tiles (the fused block with two waves per SIMD): the extra waves only
  // keep the barrier company
  const bool act = wave < NT;
  if (act) {
  f32x4 acc[NT];
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) acc[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, NT, DH, true, true>(sq, LD32, sk, LD32, row0, acc, lane);

  // logits on the accumulators: element (row = row0 + 4g + r, col = 16 rn + i)
  bool keyok[NT];                        // the lane's NT key columns: inside the scene and not padding
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) keyok[rn] = rn * 16 + i < L && !pad_b[min(rn * 16 + i, L - 1)];
  float mx[4], sm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    const RowCond c = load_cond(cond0, ldc, L, row);
    float m = -INFINITY;
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const int col = rn * 16 + i;
      float lg = -INFINITY;
      if (row < L && keyok[rn]) {
        const float *pl = plb + ((size_t)row * L + col) * SD;
        float z = c.bias;
#pragma unroll
        for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
        // sigmoid -> clamp -> log as the reference states it, on the hardware's exp2 / rcp / log2 (1 ulp each):
        // expf / logf / a true division are ~45 VALU instructions per pair, these are 8 -- and this loop is
        // where the block's time goes (16 pairs per lane, 18 k of its 50 k cycles)
        const float loc = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        lg = __logf(fmaxf(loc, 1e-6f)) + acc[rn][r] * kInvSqrtDh;
      }
      acc[rn][r] = lg;
      m = fmaxf(m, lg);
    }
    mx[r] = row16_max(m);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const float e = (acc[rn][r] == -INFINITY) ? 0.f : __expf(acc[rn][r] - mx[r]);
      acc[rn][r] = e;
      s += e;
    }
    sm[r] = row16_sum(s);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    const float inv = 1.0f / sm[r];      // a fully padded row gives NaN, as the reference would
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const int col = rn * 16 + i;
      const float p = (row < L) ? acc[rn][r] * inv : 0.f;
      sp[row * LDP + col] = p * kPScale;
      if (probs_bh && row < L && col < L) probs_bh[(size_t)row * L + col] = p;
    }
  }
  }
  __syncthreads();
  o[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (act) strip_mma<MMA, 2, LT, true, false>(sp, LDP, sv, LD32, row0, o, lane);   // ctx = P V
}

// =================================================================================
// backward core.  On entry sq / sk / sv / sdo hold q, k, v and d ctx of the head, sp the saved
// probabilities ([LT][LT+4], zero outside (L, L)), all visible.  Returns dq (oq), dk (ok), dv (ov) in
// the accumulator layout (row = 16 wave + 4 g + r, col = 16 rn + i; dq / dk still to be divided by
// sqrt(dh)) and writes the (bias, w[5]) gradients of each row to dcond0 + row * lddc (lane i == 0).
// =================================================================================
// TR: dq / dk / dv returned TRANSPOSED per tile (strip_mma's SWAP): element (row = 16 wave + i, col = 16 rn + 4 g + r).
template <int LT, int MMA, bool TR = false>
__device__ __forceinline__ void attn_bwd_core(int L, const float *sq, const float *sk, const float *sv,
                                              const float *sdo, float *sp, const float *plb, const float *cond0,
                                              int ldc, const unsigned char *pad_b, float *dcond0, int lddc,
                                              f32x4 (&oq)[2], f32x4 (&ok)[2], f32x4 (&ov)[2]) {
  constexpr int NT = LT / 16, LDP = LT + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  // dP = dctx V^T   (rows = queries, cols = keys)
  f32x4 acc[NT];
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) acc[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, NT, DH, true, true>(sdo, LD32, sv, LD32, row0, acc, lane);
  // dv = P^T dctx (rows = keys) while P is still intact
  ov[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  ov[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, 2, LT, false, false, TR>(sp, LDP, sdo, LD32, row0, ov, lane);
  __syncthreads();                       // every wave is done reading P as a matrix operand

  bool keyok[NT];
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) keyok[rn] = rn * 16 + i < L && !pad_b[min(rn * 16 + i, L - 1)];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    float dot = 0.f;
    float p[NT];
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      p[rn] = sp[row * LDP + rn * 16 + i];
      dot = fmaf(p[rn], acc[rn][r], dot);
    }
    dot = row16_sum(dot);
    const RowCond c = load_cond(cond0, ldc, L, row);
    float gb = 0.f, gw[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) gw[d] = 0.f;
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const int col = rn * 16 + i;
      const float dlogit = p[rn] * (acc[rn][r] - dot);     // softmax backward
      sp[row * LDP + col] = dlogit;         // in place: this lane owns the element
      if (row < L && keyok[rn]) {
        const float *pl = plb + ((size_t)row * L + col) * SD;
        float z = c.bias;
#pragma unroll
        for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
        const float loc = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        // d log(max(loc,1e-6)) / dz = (1 - loc) where the clamp is inactive, else 0
        const float dz = (loc >= 1e-6f) ? dlogit * (1.0f - loc) : 0.f;
        gb += dz;
#pragma unroll
        for (int d = 0; d < SD; ++d) gw[d] = fmaf(dz, pl[d], gw[d]);
      }
    }
    gb = row16_sum(gb);
#pragma unroll
    for (int d = 0; d < SD; ++d) gw[d] = row16_sum(gw[d]);
    if (i == 0 && row < L) {
      float *o = dcond0 + (size_t)row * lddc;
      o[0] = gb;
#pragma unroll
      for (int d = 0; d < SD; ++d) o[1 + d] = gw[d];
    }
  }
  __syncthreads();

  // dq = (dS K) / sqrt(dh): rows = queries;  dk = (dS^T Q) / sqrt(dh), dv = P^T dctx: rows = keys
#pragma unroll
  for (int rn = 0; rn < 2; ++rn) {
    oq[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
    ok[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  strip_mma<MMA, 2, LT, true, false, TR>(sp, LDP, sk, LD32, row0, oq, lane);
  strip_mma<MMA, 2, LT, false, false, TR>(sp, LDP, sq, LD32, row0, ok, lane);
}

}  // namespace msr3d_attn
