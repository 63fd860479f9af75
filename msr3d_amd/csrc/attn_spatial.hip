// attn_spatial.hip -- the core of MultiHeadAttentionSpatial with 'cond' fusion
// (/root/reference/modules/layers/transformers.py:200-252), forward and backward, one
// workgroup per (sample, head):
//
//   s[l,t]   = (q_l . k_t) / sqrt(dh)                                       (:205)
//   z[l,t]   = bias[l] + sum_d w[l,d] * pairwise_locs[b,l,t,d]              (:225-226)
//   loc      = sigmoid(z);  padded keys: s = -inf, loc = 0                   (:227,229-235)
//   P        = softmax_t( log(max(loc, 1e-6)) + s )                          (:241-244)
//   ctx_l    = sum_t P[l,t] v_t                                              (:248)
//
// The reference spends ~25 launches per layer on this (einsum / rearrange / masked_fill /
// sigmoid / clamp / log / softmax / einsum) and a host sync (`assert isnan`, :246); here it
// is one launch forward and one backward.  The four small matrix products of each
// direction run on f32-input MFMA (16x16x4) from LDS tiles; softmax and the spatial
// term are applied on the accumulators (row statistics by DPP inside the 16-lane rows of
// the MFMA C/D layout).  fp32 like the reference (autocast is disabled around the
// encoder, model/ose3d_situation.py:377) by default.
//
// Optional operand precisions for the matrix products (`mma` argument; softmax, the spatial term
// and every accumulation stay fp32, the LDS tiles stay fp32 and are rounded as fragments are read):
//   MSR3D_MMA_BF16  v_mfma_f32_16x16x32_bf16, forward and backward (BASELINE.json north_star's
//                   wording for QK^T / AV);
//   MSR3D_MMA_FP8   v_mfma_f32_16x16x32_fp8_fp8 (OCP e4m3), forward only -- the "fp8 object
//                   attention" of the stress configuration; P is scaled by 256 into e4m3's normal
//                   range before rounding and the product scaled back.
// The C/D layout of the 16x16x32 instructions is the 16x16x4 one, so only the operand fetch differs.
//
// Shapes: L <= 128 tokens, dh = 32, spatial_dim = 5, one (bias, w[5]) sextet per (token, head)
// in `cond` (B*L, H*6).  Two instantiations: a 64-token tile (60 objects, or 61 with the agent
// token: every shipped config; 4 waves) and a 128-token tile (the 120-object stress config,
// BASELINE.json configs[4]; 8 waves, LDS tiles in dynamic shared memory).  Larger L takes the
// composite path on the host side.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int DH = 32;        // head dim
constexpr int SD = 5;         // spatial dims
constexpr int LD32 = DH + 4;  // [LT][36] tiles
constexpr int kMaxL = 128;
constexpr float kSqrtDh = 5.656854249492381f;   // sqrt(32): s = dot / this (a division, as :205)

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

template <int MMA, int RN, int KD, bool A_KC, bool B_KC>
__device__ __forceinline__ void strip_mma(const float *As, int lda, const float *Bs, int ldb,
                                          int row0, f32x4 (&acc)[RN], int lane) {
  const int i = lane & 15, g = lane >> 4;
  if (MMA == MSR3D_MMA_F32) {
#pragma unroll
    for (int k0 = 0; k0 < KD; k0 += 4) {
      const int k = k0 + g;
      const float a = A_KC ? As[(row0 + i) * lda + k] : As[k * lda + row0 + i];
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float b = B_KC ? Bs[(rn * 16 + i) * ldb + k] : Bs[k * ldb + rn * 16 + i];
        acc[rn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[rn], 0, 0, 0);
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
          acc[rn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pack_bf16(fb), acc[rn], 0, 0, 0);
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

struct RowCond { float bias, w[SD]; };

__device__ __forceinline__ RowCond load_cond(const float *__restrict__ cond, int ldc, int b, int h,
                                             int L, int row) {
  RowCond c;
  c.bias = 0.f;
#pragma unroll
  for (int d = 0; d < SD; ++d) c.w[d] = 0.f;
  if (row < L) {
    const float *p = cond + ((size_t)b * L + row) * ldc + h * (SD + 1);
    c.bias = p[0];
#pragma unroll
    for (int d = 0; d < SD; ++d) c.w[d] = p[1 + d];
  }
  return c;
}

// =================================================================================
// forward.  grid (H, B), LT/16 waves; wave w owns query rows [16w, 16w+16).
// =================================================================================
template <int LT, int MMA>
__global__ __launch_bounds__(LT * 4) void attn_fwd_kernel(int B, int L, int H,
                                                       const float *__restrict__ q,
                                                       const float *__restrict__ k,
                                                       const float *__restrict__ v, int ldqkv,
                                                       const float *__restrict__ cond, int ldc,
                                                       const float *__restrict__ ploc,
                                                       const unsigned char *__restrict__ pad,
                                                       float *__restrict__ ctx,
                                                       float *__restrict__ probs) {
  constexpr int NT = LT / 16, LDP = LT + 4;      // key tiles per row strip; [LT][LT+4] P tile
  constexpr float kPScale = (MMA == MSR3D_MMA_FP8) ? 256.f : 1.f;   // P into e4m3's normal range
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sq = smem, *sk = sq + LT * LD32, *sv = sk + LT * LD32, *sp = sv + LT * LD32;
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  load_head_tile<LT>(q, ldqkv, b, h, L, sq);
  load_head_tile<LT>(k, ldqkv, b, h, L, sk);
  load_head_tile<LT>(v, ldqkv, b, h, L, sv);
  const float *plb = stage_ploc<LT>(ploc, b, L, sp + LT * LDP);
  __syncthreads();

  f32x4 acc[NT];
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) acc[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, NT, DH, true, true>(sq, LD32, sk, LD32, row0, acc, lane);

  // logits on the accumulators: element (row = row0 + 4g + r, col = 16 rn + i)
  float mx[4], sm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    const RowCond c = load_cond(cond, ldc, b, h, L, row);
    float m = -INFINITY;
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const int col = rn * 16 + i;
      float lg = -INFINITY;
      if (row < L && col < L && !pad[(size_t)b * L + col]) {
        const float *pl = plb + ((size_t)row * L + col) * SD;
        float z = c.bias;
#pragma unroll
        for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
        const float loc = 1.0f / (1.0f + expf(-z));
        lg = logf(fmaxf(loc, 1e-6f)) + acc[rn][r] / kSqrtDh;
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
      const float e = (acc[rn][r] == -INFINITY) ? 0.f : expf(acc[rn][r] - mx[r]);
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
      if (probs && row < L && col < L) probs[(((size_t)b * H + h) * L + row) * L + col] = p;
    }
  }
  __syncthreads();

  f32x4 o[2];
  o[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, 2, LT, true, false>(sp, LDP, sv, LD32, row0, o, lane);   // ctx = P V
#pragma unroll
  for (int rn = 0; rn < 2; ++rn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      if (row < L)
        ctx[((size_t)b * L + row) * (H * DH) + h * DH + rn * 16 + i] = o[rn][r] * (1.0f / kPScale);
    }
}

// =================================================================================
// backward.  Inputs as forward + probs (saved) + dctx; outputs dq, dk, dv (token-major,
// ld = ldg) and dcond (B*L, H*6).  pairwise_locs and the mask get no gradient.
// =================================================================================
template <int LT, int MMA>
__global__ __launch_bounds__(LT * 4) void attn_bwd_kernel(int B, int L, int H,
                                                       const float *__restrict__ q,
                                                       const float *__restrict__ k,
                                                       const float *__restrict__ v, int ldqkv,
                                                       const float *__restrict__ cond, int ldc,
                                                       const float *__restrict__ ploc,
                                                       const unsigned char *__restrict__ pad,
                                                       const float *__restrict__ probs,
                                                       const float *__restrict__ dctx,
                                                       float *__restrict__ dq,
                                                       float *__restrict__ dk,
                                                       float *__restrict__ dv, int ldg,
                                                       float *__restrict__ dcond, int lddc) {
  constexpr int NT = LT / 16, LDP = LT + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sq = smem, *sk = sq + LT * LD32, *sv = sk + LT * LD32, *sdo = sv + LT * LD32;
  float *sp = sdo + LT * LD32;                    // P, then dS in place
  const float *plb = stage_ploc<LT>(ploc, blockIdx.y, L, sp + LT * LDP);
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  load_head_tile<LT>(q, ldqkv, b, h, L, sq);
  load_head_tile<LT>(k, ldqkv, b, h, L, sk);
  load_head_tile<LT>(v, ldqkv, b, h, L, sv);
  load_head_tile<LT>(dctx, H * DH, b, h, L, sdo);
  for (int e = threadIdx.x; e < LT * LT; e += LT * 4) {
    const int row = e / LT, col = e % LT;
    sp[row * LDP + col] = (row < L && col < L) ? probs[(((size_t)b * H + h) * L + row) * L + col] : 0.f;
  }
  __syncthreads();

  // dP = dctx V^T   (rows = queries, cols = keys)
  f32x4 acc[NT];
#pragma unroll
  for (int rn = 0; rn < NT; ++rn) acc[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, NT, DH, true, true>(sdo, LD32, sv, LD32, row0, acc, lane);
  // dv = P^T dctx (rows = keys) while P is still intact
  f32x4 ov[2];
  ov[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  ov[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  strip_mma<MMA, 2, LT, false, false>(sp, LDP, sdo, LD32, row0, ov, lane);
  __syncthreads();                       // every wave is done reading P as a matrix operand

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
    const RowCond c = load_cond(cond, ldc, b, h, L, row);
    float gb = 0.f, gw[SD];
#pragma unroll
    for (int d = 0; d < SD; ++d) gw[d] = 0.f;
#pragma unroll
    for (int rn = 0; rn < NT; ++rn) {
      const int col = rn * 16 + i;
      const float dlogit = p[rn] * (acc[rn][r] - dot);     // softmax backward
      sp[row * LDP + col] = dlogit;         // in place: this lane owns the element
      if (row < L && col < L && !pad[(size_t)b * L + col]) {
        const float *pl = plb + ((size_t)row * L + col) * SD;
        float z = c.bias;
#pragma unroll
        for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
        const float loc = 1.0f / (1.0f + expf(-z));
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
      float *o = dcond + ((size_t)b * L + row) * lddc + h * (SD + 1);
      o[0] = gb;
#pragma unroll
      for (int d = 0; d < SD; ++d) o[1 + d] = gw[d];
    }
  }
  __syncthreads();

  // dq = (dS K) / sqrt(dh): rows = queries;  dk = (dS^T Q) / sqrt(dh), dv = P^T dctx: rows = keys
  f32x4 oq[2], ok[2];
#pragma unroll
  for (int rn = 0; rn < 2; ++rn) {
    oq[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
    ok[rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  strip_mma<MMA, 2, LT, true, false>(sp, LDP, sk, LD32, row0, oq, lane);
  strip_mma<MMA, 2, LT, false, false>(sp, LDP, sq, LD32, row0, ok, lane);
#pragma unroll
  for (int rn = 0; rn < 2; ++rn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      if (row < L) {
        const size_t o = ((size_t)b * L + row) * ldg + h * DH + rn * 16 + i;
        dq[o] = oq[rn][r] / kSqrtDh;
        dk[o] = ok[rn][r] / kSqrtDh;
        dv[o] = ov[rn][r];
      }
    }
}

// dynamic LDS: forward 3 head tiles + P, backward 4 head tiles + P
template <int LT> constexpr size_t ploc_lds() { return LT == 64 ? sizeof(float) * 64 * 64 * SD : 0; }
template <int LT> constexpr size_t fwd_lds() { return sizeof(float) * (3 * LT * LD32 + LT * (LT + 4)) + ploc_lds<LT>(); }
template <int LT> constexpr size_t bwd_lds() { return sizeof(float) * (4 * LT * LD32 + LT * (LT + 4)) + ploc_lds<LT>(); }

template <int LT, int MMA, typename... Args>
hipError_t launch_fwd(int B, int H, hipStream_t stream, Args... args) {
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void *>(&attn_fwd_kernel<LT, MMA>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<LT>());
  if (attr != hipSuccess) return attr;
  attn_fwd_kernel<LT, MMA><<<dim3(H, B), LT * 4, fwd_lds<LT>(), stream>>>(B, args...);
  return hipGetLastError();
}

template <int LT, int MMA, typename... Args>
hipError_t launch_bwd(int B, int H, hipStream_t stream, Args... args) {
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void *>(&attn_bwd_kernel<LT, MMA>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<LT>());
  if (attr != hipSuccess) return attr;
  attn_bwd_kernel<LT, MMA><<<dim3(H, B), LT * 4, bwd_lds<LT>(), stream>>>(B, args...);
  return hipGetLastError();
}

template <int LT, typename... Args>
hipError_t dispatch_fwd(int mma, Args... args) {
  switch (mma) {
    case MSR3D_MMA_F32: return launch_fwd<LT, MSR3D_MMA_F32>(args...);
    case MSR3D_MMA_BF16: return launch_fwd<LT, MSR3D_MMA_BF16>(args...);
    default: return launch_fwd<LT, MSR3D_MMA_FP8>(args...);
  }
}
template <int LT, typename... Args>
hipError_t dispatch_bwd(int mma, Args... args) {
  if (mma == MSR3D_MMA_BF16) return launch_bwd<LT, MSR3D_MMA_BF16>(args...);
  return launch_bwd<LT, MSR3D_MMA_F32>(args...);
}

}  // namespace

extern "C" {

int msr3d_spatial_attn_fwd(int B, int L, int H, int dh, int spatial_dim, const float *q,
                           const float *k, const float *v, int ld_qkv, const float *cond,
                           int ld_cond, const float *pairwise_locs,
                           const unsigned char *key_padding_mask, float *ctx, float *probs,
                           int mma, msr3d_stream_t stream) {
  if (B < 0 || L <= 0 || L > kMaxL || H <= 0 || dh != DH || spatial_dim != SD) return MSR3D_EINVAL;
  if (mma != MSR3D_MMA_F32 && mma != MSR3D_MMA_BF16 && mma != MSR3D_MMA_FP8) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!q || !k || !v || !cond || !pairwise_locs || !key_padding_mask || !ctx) return MSR3D_EINVAL;
  if (ld_qkv % 4 != 0) return MSR3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64)
    return (int)dispatch_fwd<64>(mma, B, H, s, L, H, q, k, v, ld_qkv, cond, ld_cond, pairwise_locs,
                                 key_padding_mask, ctx, probs);
  return (int)dispatch_fwd<128>(mma, B, H, s, L, H, q, k, v, ld_qkv, cond, ld_cond, pairwise_locs,
                                key_padding_mask, ctx, probs);
}

int msr3d_spatial_attn_bwd(int B, int L, int H, int dh, int spatial_dim, const float *q,
                           const float *k, const float *v, int ld_qkv, const float *cond,
                           int ld_cond, const float *pairwise_locs,
                           const unsigned char *key_padding_mask, const float *probs,
                           const float *dctx, float *dq, float *dk, float *dv, int ld_grad,
                           float *dcond, int ld_dcond, int mma, msr3d_stream_t stream) {
  if (B < 0 || L <= 0 || L > kMaxL || H <= 0 || dh != DH || spatial_dim != SD) return MSR3D_EINVAL;
  if (mma != MSR3D_MMA_F32 && mma != MSR3D_MMA_BF16) return MSR3D_EINVAL;   // fp8 is forward-only
  if (B == 0) return 0;
  if (!q || !k || !v || !cond || !pairwise_locs || !key_padding_mask || !probs || !dctx || !dq ||
      !dk || !dv || !dcond)
    return MSR3D_EINVAL;
  if (ld_qkv % 4 != 0) return MSR3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64)
    return (int)dispatch_bwd<64>(mma, B, H, s, L, H, q, k, v, ld_qkv, cond, ld_cond, pairwise_locs,
                                 key_padding_mask, probs, dctx, dq, dk, dv, ld_grad, dcond, ld_dcond);
  return (int)dispatch_bwd<128>(mma, B, H, s, L, H, q, k, v, ld_qkv, cond, ld_cond, pairwise_locs,
                                key_padding_mask, probs, dctx, dq, dk, dv, ld_grad, dcond, ld_dcond);
}

}  // extern "C"
