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

#include "../../include/msr3d_hip.h"
#include "attn_core.h"

namespace {

using namespace msr3d_attn;

// (scene, head) of a workgroup.  Speed only: a workgroup runs on XCD (linear id mod 8); with eight heads and a multiple of
// eight scenes a scene's heads are put on ONE XCD -- they share the scene's pairwise rows (L L 5 floats: 293 KB at the
// stress shape's 121 tokens), which every L2 otherwise pulls for every scene (csrc/scene_block.hip's head has the figures).
__device__ __forceinline__ void scene_head(int &h, int &b) {
  h = blockIdx.x; b = blockIdx.y;
  if (gridDim.x == 8 && (gridDim.y & 7) == 0) {
    const int id = blockIdx.y * 8 + blockIdx.x, jj = id >> 3;
    b = (id & 7) + 8 * (jj >> 3); h = jj & 7;
  }
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
  constexpr int LDP = LT + 4;      // [LT][LT+4] P tile
  constexpr float kPScale = (MMA == MSR3D_MMA_FP8) ? 256.f : 1.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sq = smem, *sk = sq + LT * LD32, *sv = sk + LT * LD32, *sp = sv + LT * LD32;
  int h, b;
  scene_head(h, b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  load_head_tile<LT>(q, ldqkv, b, h, L, sq);
  load_head_tile<LT>(k, ldqkv, b, h, L, sk);
  load_head_tile<LT>(v, ldqkv, b, h, L, sv);
  const float *plb = stage_ploc<LT>(ploc, b, L, sp + LT * LDP);
  __syncthreads();

  f32x4 o[2];
  attn_fwd_core<LT, MMA>(L, sq, sk, sv, sp, plb, cond + (size_t)b * L * ldc + h * (SD + 1), ldc,
                         pad + (size_t)b * L, probs ? probs + ((size_t)b * H + h) * L * L : nullptr, o);
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
  constexpr int LDP = LT + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sq = smem, *sk = sq + LT * LD32, *sv = sk + LT * LD32, *sdo = sv + LT * LD32;
  float *sp = sdo + LT * LD32;                    // P, then dS in place
  int h, b;
  scene_head(h, b);
  const float *plb = stage_ploc<LT>(ploc, b, L, sp + LT * LDP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4, row0 = wave * 16;
  load_head_tile<LT>(q, ldqkv, b, h, L, sq);
  load_head_tile<LT>(k, ldqkv, b, h, L, sk);
  load_head_tile<LT>(v, ldqkv, b, h, L, sv);
  load_head_tile<LT>(dctx, H * DH, b, h, L, sdo);
  load_probs_tile<LT>(probs + ((size_t)b * H + h) * L * L, L, sp);
  __syncthreads();

  f32x4 oq[2], ok[2], ov[2];
  attn_bwd_core<LT, MMA>(L, sq, sk, sv, sdo, sp, plb, cond + (size_t)b * L * ldc + h * (SD + 1), ldc,
                         pad + (size_t)b * L, dcond + (size_t)b * L * lddc + h * (SD + 1), lddc, oq, ok, ov);
#pragma unroll
  for (int rn = 0; rn < 2; ++rn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      if (row < L) {
        const size_t o = ((size_t)b * L + row) * ldg + h * DH + rn * 16 + i;
        dq[o] = oq[rn][r] * kInvSqrtDh;
        dk[o] = ok[rn][r] * kInvSqrtDh;
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
