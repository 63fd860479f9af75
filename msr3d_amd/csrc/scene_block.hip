// scene_block.hip -- the trainable part of the hot path as SCENE-LOCAL fused blocks on the bf16 matrix
// pipe at fp32 accuracy (include/msr3d_hip.h, "scene-local fused blocks").
//
// The spatial encoder layer (/root/reference/modules/layers/transformers.py:200-252,314-329) mixes only
// the <= 64 tokens of ONE scene, and every product of it has the model width (256) on one side.  So a
// scene is a 64-row tile, and each half of the layer -- attention block, feed-forward block, forward
// and backward -- is the same sandwich
//
//     rows  = three bf16 planes of the block's 64 x 256 input (written by msr3d_scene_rows)
//     mid   = middle( rows . Op1_slice^T )       K = 256 -> slice width (a head, 128 hidden units)
//     part[slice] = mid . Op2_slice^T            K = slice -> 256; the slices' partial products are summed,
//                                                in slab order, by the next msr3d_scene_rows launch
//
// with the middle = spatial attention of one head (attn_core.h), GELU + dropout, or their backward.
// One workgroup (4 waves) per (scene, slice): 128 workgroups for the attention blocks, 256 for the
// feed-forward blocks.  The (tokens x 2048) activation, q / k / v and the context vectors never go
// through HBM between the two products (they are written once, as side outputs for the backward).
//
// Matrix products: six v_mfma_f32_16x16x32_bf16 per block of products on exactly-split operands
// (split_mma.h).  `mid` is split by product 1's epilogue into fragment-order planes in LDS; the weights
// arrive pre-split and fragment-packed (msr3d_split_pack, once per optimiser step) through a buffer
// descriptor and a register ring four pieces deep, so there is no LDS staging of weights and no barrier
// inside a product.  Waves are laid out 1 x 4: each owns all 64 rows and a quarter of the columns, so a
// weight fragment is fetched once per workgroup.
//
// Why partial slabs and not atomics: scene_rows.hip.
//
// Workgroup -> XCD (speed only; a workgroup runs on XCD (linear id mod 8)).  Feed-forward / projector blocks:
// blockIdx.x = slice and gridDim.x = 16, so all scenes' workgroups of a slice land on the XCD (slice mod 8) -- a slice's
// weights (0.4 MB) are fetched into ONE L2, every L2 sees every scene's planes (1.5 MB).  ATTENTION blocks (round 6,
// MSR3D_ATTN_SCENE_XCD): the other way round -- a scene's eight (sixteen) workgroups on XCD (scene mod 8) when the
// batch is a multiple of eight scenes: a head's weights are 0.15-0.2 MB, so every L2 holding all heads (1.6 MB) and two
// scenes' planes + pairwise rows (0.3 MB) is less fabric traffic than one head and sixteen scenes (2.7 MB): 7-9 us a
// step (profiles/r06_v4_ab_attn_xcd.txt); results do not depend on it.  The same placement for the feed-forward blocks
// (MSR3D_FFN_SCENE_XCD=1 at build time: all sixteen slices' weights, 6.3 MB, past every 4 MB L2) is neutral on the day's
// fast boxes and 14 us a step slower on the others: off.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/msr3d_hip.h"
#include "attn_core.h"
#include "rowmath.h"
#include "split_mma.h"

// phase marks: empty here; tools/prof/scene_block_stamped.hip defines SB_STAMP and includes this file
#ifndef SB_STAMP
#define SB_STAMP(i)
#endif

#ifndef MSR3D_ATTN_SCENE_XCD
#define MSR3D_ATTN_SCENE_XCD 1
#endif
#ifndef MSR3D_FFN_SCENE_XCD
#define MSR3D_FFN_SCENE_XCD 0
#endif

namespace {

using namespace msr3d;
using msr3d_attn::DH;
using msr3d_attn::LD32;
using msr3d_attn::SD;
using msr3d_attn::kInvSqrtDh;

using SB = msr3d_scene_block_t;

constexpr int KD = 256;                 // model width
constexpr int TM = 64;                  // rows of a scene tile
constexpr int PITCH = KD + 16;          // bf16 units: 544 B rows, conflict-free ds_read_b128 fragments
constexpr int PLANE = TM * PITCH;       // bf16 units per plane
constexpr int XS_BYTES = 3 * PLANE * 2; // 104,448
constexpr int RING = 4;                 // weight pieces in flight per wave (12 KB)
// the attention core's operand form: f32-input MFMA, or (MSR3D_TRAIN_PLANES = 1, the labelled bf16 variant) bf16 operands
constexpr int kCoreMma = MSR3D_TRAIN_PLANES == 1 ? MSR3D_MMA_BF16 : MSR3D_MMA_F32;
constexpr int MID_BYTES = 4 * 4 * 3 * 1024;   // FRAG planes of a 64 x 128 middle: 49,152

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// erf-form GELU (F.gelu default, transformers.py:323) with erf by Abramowitz-Stegun 7.1.26:
// |error| <= 1.5e-7 on erf, i.e. below the fp32 resolution of (1 + erf).  Branch-free, ~15 instructions:
// the library erff (two polynomial branches, both executed by a diverged wave) made the feed-forward
// block's epilogue -- 32 activations per lane with one wave per SIMD -- as long as its two products.
__device__ __forceinline__ float erfc_pos(float z, float e) {      // erfc(z), z >= 0, e = exp(-z z)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  return q * t * e;
}
__device__ __forceinline__ float gelu_as(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float ec = erfc_pos(z, __expf(-z * z));                     // erfc(|x| / sqrt 2)
  const float phi = x >= 0.f ? 1.0f - 0.5f * ec : 0.5f * ec;        // Phi(x)
  return x * phi;
}
__device__ __forceinline__ float gelu_as_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float e = __expf(-z * z);                                   // exp(-x x / 2)
  const float ec = erfc_pos(z, e);
  const float phi = x >= 0.f ? 1.0f - 0.5f * ec : 0.5f * ec;
  return fmaf(x * 0.39894228040143267794f, e, phi);
}

// Input staging.  Every kind but LINEAR_KSPLIT: the scene's three planes (B, 3, 64, 256) bf16, one
// contiguous 96 KB block, copied into the ROWS layout (pitch 272) -- 24 16-byte loads per thread, all in
// flight together; rows past L are zero in the source.  LINEAR_KSPLIT: fp32 rows a0[row][col0 .. col0 + 256)
// split on the way in (the upstream gradient of llm_proj comes from outside the schedule).
template <int NTHR = 256>
__device__ __forceinline__ void stage_planes(const unsigned short *__restrict__ xp, unsigned short *xs, int b) {
  const uint4 *src = reinterpret_cast<const uint4 *>(xp + (size_t)b * 3 * TM * KD);
  constexpr int NV = 8 * kPlanes * 256 / NTHR;        // (kPlanes = 1: the first plane's 2048 uint4 only)
  uint4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = src[threadIdx.x + NTHR * k];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int q = threadIdx.x + NTHR * k;
    const int plane = q >> 11, row = (q >> 5) & 63, c8 = q & 31;
    *reinterpret_cast<uint4 *>(xs + plane * PLANE + row * PITCH + c8 * 8) = v[k];
  }
}

// stage_planes in two steps (256 threads): the loads, and -- after the caller has issued other loads behind them -- the
// LDS stores
__device__ __forceinline__ void planes_fetch(const unsigned short *__restrict__ xp, int b, uint4 (&v)[24]) {
  const uint4 *src = reinterpret_cast<const uint4 *>(xp + (size_t)b * 3 * TM * KD);
#pragma unroll
  for (int k = 0; k < 8 * kPlanes; ++k) v[k] = src[threadIdx.x + 256 * k];
}
__device__ __forceinline__ void planes_store(unsigned short *xs, const uint4 (&v)[24]) {
#pragma unroll
  for (int k = 0; k < 8 * kPlanes; ++k) {
    const int q = threadIdx.x + 256 * k;
    const int plane = q >> 11, row = (q >> 5) & 63, c8 = q & 31;
    *reinterpret_cast<uint4 *>(xs + plane * PLANE + row * PITCH + c8 * 8) = v[k];
  }
}

__device__ __forceinline__ void stage_f32(const float *__restrict__ a0, int lda, int col0, unsigned short *xs,
                                          int row_base, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cseg = lane & 15, sub = lane >> 4;
  float4 in[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 16 + 4 * j + sub;
    const float *src = a0 + (size_t)(row_base + min(r, L - 1)) * lda + col0 + 4 * cseg;
#pragma unroll
    for (int q = 0; q < 4; ++q) in[j][q] = ld4(src + 64 * q);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 16 + 4 * j + sub;
    const bool ok = r < L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v[4] = {ok ? in[j][q].x : 0.f, ok ? in[j][q].y : 0.f, ok ? in[j][q].z : 0.f, ok ? in[j][q].w : 0.f};
      uint2 pl[3];
      sm_split4(v, pl);
      unsigned short *d = xs + r * PITCH + 64 * q + 4 * cseg;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * PLANE) = pl[k];
    }
  }
}

// product-2 epilogue: this slice's partial product -> its slab.  D = W X^T: lane (j, g) holds channels
// n0 + 16 rn + 4 g .. + 3 of token 16 mt + j -- one 16-byte store per tile, 64 contiguous bytes per row
template <int RN>
__device__ __forceinline__ void store_partials(const f32x4 (&acc)[RN][4], float *__restrict__ slab, int row_base,
                                               int L, int n0, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = 16 * mt + j;
      if (row < L)
        st4(slab + (size_t)(row_base + row) * KD + n0 + 16 * rn + 4 * g,
            make_float4(acc[rn][mt][0], acc[rn][mt][1], acc[rn][mt][2], acc[rn][mt][3]));
    }
}

// A dense fp32 slab (the scene's pairwise features, a head's probabilities) whose start is 4- but not 16-byte aligned --
// L = 61: the agent token of situation_type 'as_object' -- still travels as 16-byte vectors: the loads start at the
// 16-byte boundary below it (`mis` floats early) through a buffer descriptor that ends with the slab (the tail's lanes
// past it read 0, nothing outside [start - mis, end) is touched), and the consumer finds element i at position i + mis.
struct Slab4 {
  __amdgpu_buffer_rsrc_t rsrc;
  int mis, n4;                           // floats in front of the slab; 16-byte vectors covering mis + n floats
};
__device__ __forceinline__ Slab4 make_slab4(const float *src, int n) {
  Slab4 s;
  s.mis = (int)((reinterpret_cast<uintptr_t>(src) >> 2) & 3u);
  s.n4 = (s.mis + n + 3) >> 2;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src - s.mis), 0, (s.mis + n) * 4, 0x00020000);
  return s;
}
__device__ __forceinline__ float4 slab4_load(const Slab4 &s, int e) {
  const sm_i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, e * 16, 0, 0);       // (e >= n4: out of range, zeros)
  return make_float4(__int_as_float(r.x), __int_as_float(r.y), __int_as_float(r.z), __int_as_float(r.w));
}

// LDS carve-up behind the ROWS planes, by kind
constexpr int kTile = TM * LD32 * 4;                       // one [64][36] fp32 head tile: 9,216 B
constexpr int kAttnFwdAux = 3 * kTile + 2048 + 4 * 3 * 1024;          // q k v | cond [64][8] | ctx FRAG (1 slab)
constexpr int kAttnBwdAux = 4 * kTile + 2048 + 2048;                   // q k v do | cond | dcond

template <int KIND>
constexpr int lds_bytes() {
  return KIND == MSR3D_BLK_ATTN_FWD ? XS_BYTES + kAttnFwdAux
       : KIND == MSR3D_BLK_ATTN_BWD ? XS_BYTES + kAttnBwdAux
       : (KIND == MSR3D_BLK_FFN_FWD || KIND == MSR3D_BLK_FFN_BWD) ? XS_BYTES + MID_BYTES
       : XS_BYTES;
}

// NW: waves per workgroup.  8 (two per SIMD: one wave alone issues at most 44 % of a SIMD's VALU rate,
// profiles/r05_hw_probes.txt) for the attention forward and the feed-forward blocks -- each wave then owns half the
// column tiles -- 4 for the rest; MSR3D_ATTN_FWD_WAVES=4 / MSR3D_FFN_WAVES=4 restore four.
template <int KIND, int NW = 4>
__global__ __launch_bounds__(64 * NW) void scene_block_kernel(const SB p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
  unsigned char *aux = smem + XS_BYTES;
  // (rows_total > 0: row tiles that ignore scene boundaries -- tile b holds rows [b L, min((b + 1) L, rows_total)))
  int slice = blockIdx.x, b = blockIdx.y;
#if MSR3D_ATTN_SCENE_XCD
  if ((KIND == MSR3D_BLK_ATTN_FWD || KIND == MSR3D_BLK_ATTN_BWD) && gridDim.x == 8 && (gridDim.y & 7) == 0) {
    const int id = blockIdx.y * 8 + blockIdx.x, jj = id >> 3;   // (a scene's eight heads on ONE XCD: see the file's head)
    b = (id & 7) + 8 * (jj >> 3); slice = jj & 7;
  }
#endif
#if MSR3D_FFN_SCENE_XCD
  if ((KIND == MSR3D_BLK_FFN_FWD || KIND == MSR3D_BLK_FFN_BWD) && gridDim.x == 16 && (gridDim.y & 7) == 0 && p.rows_total == 0) {
    const int id = blockIdx.y * 16 + blockIdx.x, jj = id >> 3;
    b = (id & 7) + 8 * (jj >> 4); slice = jj & 15;
  }
#endif
  const int row_base = b * p.L;
  const int L = p.rows_total > 0 ? min(p.L, p.rows_total - row_base) : p.L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float *const slab = p.part + (size_t)slice * p.part_stride;

  // ---- product 1's stream: the first pieces fly under the prologue ----
  constexpr int RN1 = (KIND == MSR3D_BLK_LINEAR || KIND == MSR3D_BLK_LINEAR_KSPLIT) ? 4
                    : (KIND == MSR3D_BLK_ATTN_BWD || NW == 8) ? 1 : 2;
  constexpr int KS1 = KD / 32;
  WStream w1;
  if (KIND == MSR3D_BLK_ATTN_FWD)          // per head: [8 slabs][8 tiles]: q q k k v v cond -
    w1 = make_wstream(p.w1 + (size_t)slice * (KS1 * 8 * kPieceBytes / 2), KS1 * 8 * kPieceBytes, 8, 0,
                      NW == 8 ? wave : 2 * wave, lane);
  else if (KIND == MSR3D_BLK_ATTN_BWD)     // Wfc as [k = fc row][n = ctx column]: [8 slabs][16 tiles], tiles 2 h, 2 h + 1
    w1 = make_wstream(p.w1, p.w1_bytes, 16, 0, 2 * slice + (wave & 1), lane);
  else if (KIND == MSR3D_BLK_LINEAR)       // [8 slabs][N / 16 tiles]
    w1 = make_wstream(p.w1, p.w1_bytes, p.N / 16, 0, 16 * slice + 4 * wave, lane);
  else if (KIND == MSR3D_BLK_LINEAR_KSPLIT)   // [K / 32 slabs][16 tiles]
    w1 = make_wstream(p.w1, p.w1_bytes, 16, KS1 * slice, 4 * wave, lane);
  else                                     // FFN: [8 slabs][ff / 16 tiles], this slice's 8 tiles
    w1 = make_wstream(p.w1, p.w1_bytes, p.ff / 16, 0, 8 * slice + (8 / NW) * wave, lane);
  SB_STAMP(0);
  WPiece ring1[RING];
  preload_wring<RN1, RING>(ring1, w1);

  // ---- the block's input -> ROWS planes ----
  if (KIND == MSR3D_BLK_LINEAR_KSPLIT) stage_f32(p.a0, p.lda0, KD * slice, xs, row_base, L);
  else stage_planes<64 * NW>(p.xp, xs, b);
  SB_STAMP(1);
  __syncthreads();
  SB_STAMP(2);
  const XRows xr = make_xrows(xs, PITCH, TM, lane);

  if constexpr (KIND == MSR3D_BLK_LINEAR) {
    // ------------------------------------------------------------------ C = rows W^T + b
    const int n0 = 256 * slice + 64 * wave + 4 * g;
    float4 bias[4];                        // ahead of the product: a load next to the stores would make every store wait
#pragma unroll
    for (int rn = 0; rn < 4; ++rn) bias[rn] = p.bias1 ? ld4(p.bias1 + n0 + 16 * rn) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[4][4];
    zero_acc3(acc);
    gemm_split3<true, 4, 4, KS1, RING>(xr, 0, w1, acc, ring1);
    SB_STAMP(3);
#pragma unroll
    for (int rn = 0; rn < 4; ++rn) {
      const float4 bv = bias[rn];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = 16 * mt + j;
        if (row < L)
          st4(p.C + (size_t)(row_base + row) * p.ldc + n0 + 16 * rn,
              make_float4(acc[rn][mt][0] + bv.x, acc[rn][mt][1] + bv.y, acc[rn][mt][2] + bv.z, acc[rn][mt][3] + bv.w));
      }
    }
  } else if constexpr (KIND == MSR3D_BLK_LINEAR_KSPLIT) {
    // ------------------------------------------------------------------ part[slice] = a0[:, slice] W[.., slice]^T
    f32x4 acc[4][4];
    zero_acc3(acc);
    gemm_split3<true, 4, 4, KS1, RING>(xr, 0, w1, acc, ring1);
    SB_STAMP(3);
    store_partials<4>(acc, slab, row_base, L, 64 * wave, lane);
  } else if constexpr (KIND == MSR3D_BLK_FFN_FWD || KIND == MSR3D_BLK_FFN_BWD) {
    // ------------------------------------------------------------------ feed-forward block
    constexpr bool FWD = KIND == MSR3D_BLK_FFN_FWD;
    const int ff = p.ff;
    // NW = 4: a wave owns 32 of the slice's 128 middle columns and 64 of product 2's 256; NW = 8: 16 and 32
    constexpr int RN2 = 16 / NW, RING2 = RING;         // (product 2: 4 slabs x RN2 pieces >= RING)
    const int lc0 = 16 * RN1 * wave + 4 * g;           // + 16 rn: the lane's four middle columns inside the slice
    const int hc0 = 128 * slice + lc0;
    // product 2's stream: [ff / 32 slabs][16 tiles], this slice's four slabs
    const WStream w2 = make_wstream(p.w2, p.w2_bytes, 16, 4 * slice, RN2 * wave, lane);
    float4 pre_in[RN1][4];
    if (!FWD) {                                        // gelu'(pre): fetched under product 1
#pragma unroll
      for (int rn = 0; rn < RN1; ++rn)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          pre_in[rn][mt] = ld4(p.pre + (size_t)(row_base + min(16 * mt + j, L - 1)) * ff + hc0 + 16 * rn);
    }
    float4 bias[RN1];
#pragma unroll
    for (int rn = 0; rn < RN1; ++rn)
      bias[rn] = (FWD && p.bias1) ? ld4(p.bias1 + hc0 + 16 * rn) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[RN1][4];
    zero_acc3(acc);
    gemm_split3<true, RN1, 4, KS1, RING>(xr, 0, w1, acc, ring1);
    SB_STAMP(3);
    WPiece ring2[RING2];
    preload_wring<RN2, RING2>(ring2, w2);
    const bool drop = p.p_drop > 0.f;
    const unsigned thresh = drop_thresh(p.p_drop);
    const float dscale = drop ? 1.0f / (1.0f - p.p_drop) : 1.0f;
    const unsigned long long sd = drop ? *p.seed : 0ull;
    unsigned char *mid = aux;
#pragma unroll
    for (int rn = 0; rn < RN1; ++rn) {
      const float b4[4] = {bias[rn].x, bias[rn].y, bias[rn].z, bias[rn].w};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = 16 * mt + j;
        const bool ok = row < L;
        const size_t o = (size_t)(row_base + row) * ff + hc0 + 16 * rn;
        float v[4];
        if (FWD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[rn][mt][r] + b4[r];
          if (ok) st4(p.pre + o, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = gelu_as(v[r]);
            if (drop) v[r] = keep_elem(sd, p.salt, (unsigned)(o + r), thresh) ? v[r] * dscale : 0.f;
            if (!ok) v[r] = 0.f;
          }
        } else {
          const float pi[4] = {pre_in[rn][mt].x, pre_in[rn][mt].y, pre_in[rn][mt].z, pre_in[rn][mt].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float d = acc[rn][mt][r];
            if (drop) d = keep_elem(sd, p.salt, (unsigned)(o + r), thresh) ? d * dscale : 0.f;
            v[r] = ok ? d * gelu_as_grad(pi[r]) : 0.f;
          }
        }
        if (ok) st4(p.h + o, make_float4(v[0], v[1], v[2], v[3]));
        uint2 pl[3];
        sm_split4(v, pl);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          *reinterpret_cast<uint2 *>(mid + frag_off4<4>(mt, j, lc0 + 16 * rn, k)) = pl[k];
      }
    }
    SB_STAMP(4);
    __syncthreads();
    SB_STAMP(5);
    f32x4 acc2[RN2][4];
    zero_acc3(acc2);
    const XFrag<4> xm{reinterpret_cast<const unsigned short *>(mid) + lane * 8};
    gemm_split3<true, RN2, 4, 4, RING2>(xm, 0, w2, acc2, ring2);
    SB_STAMP(6);
    store_partials<RN2>(acc2, slab, row_base, L, 16 * RN2 * wave, lane);
  } else if constexpr (KIND == MSR3D_BLK_ATTN_FWD) {
    // ------------------------------------------------------------------ attention block, forward
    const int h = slice, H = p.H, ldq = p.ldq;
    float *sq = reinterpret_cast<float *>(aux), *sk = sq + TM * LD32, *sv = sk + TM * LD32;
    float *scond = sv + TM * LD32;                                  // [64][8]
    unsigned char *ctxp = reinterpret_cast<unsigned char *>(scond + TM * 8);   // FRAG, 1 slab x 4 row tiles
    // product 2's stream: Wfc [8 slabs][16 tiles], slab h
    constexpr int RN2 = 16 / NW, RING2 = RN2 < RING ? RN2 : RING;
    const WStream w2 = make_wstream(p.w2, p.w2_bytes, 16, h, RN2 * wave, lane);
    // the scene's pairwise slab is headed for the LDS the planes occupy: fetched into registers now (threads 0..255)
    const float *plsrc = p.ploc + (size_t)b * L * L * SD;
    const int pn = L * L * SD;
    const Slab4 pls = make_slab4(plsrc, pn);          // (16-byte vectors at any alignment: L = 61)
    const bool plt = NW == 4 || tid < 256;
    float4 plv[msr3d_attn::kPlocRegs];
    if (plt) {
#pragma unroll
      for (int k = 0; k < msr3d_attn::kPlocRegs; ++k) plv[k] = slab4_load(pls, tid + 256 * k);
    }
    f32x4 acc[RN1][4];
    zero_acc3(acc);
    gemm_split3<true, RN1, 4, KS1, RING>(xr, 0, w1, acc, ring1);
    SB_STAMP(3);
    WPiece ring2[RING2];
    preload_wring<RN2, RING2>(ring2, w2);
    // the head's column tiles: q q k k v v cond -.  NW = 4: wave 0 / 1 / 2 the head's q / k / v (32 columns each),
    // wave 3 cond; NW = 8: one tile a wave
    constexpr int TPW = 8 / NW;                                       // column tiles per wave
    const int which = (TPW * wave) >> 1;                              // 0 q, 1 k, 2 v, 3 cond
    if (which < 3) {
      float *tile = sq + which * TM * LD32;
#pragma unroll
      for (int rn = 0; rn < RN1; ++rn) {
        const int c = 16 * ((TPW * wave + rn) & 1) + 4 * g;
        const float4 bv = ld4(p.bias1 + which * KD + h * DH + c);     // bias of the packed [q | k | v | cond] rows
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row = 16 * mt + j;
          const bool ok = row < L;
          const float4 v = ok ? make_float4(acc[rn][mt][0] + bv.x, acc[rn][mt][1] + bv.y, acc[rn][mt][2] + bv.z,
                                            acc[rn][mt][3] + bv.w)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
          st4(tile + row * LD32 + c, v);
          if (ok) st4(p.qkvc + (size_t)(row_base + row) * ldq + which * KD + h * DH + c, v);
        }
      }
    } else if (TPW * wave == 6) {
      float b4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) b4[r] = 4 * g + r < SD + 1 ? p.bias1[3 * KD + h * (SD + 1) + 4 * g + r] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int row = 16 * mt + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * g + r;
          const float v = row < L ? acc[0][mt][r] + b4[r] : 0.f;
          if (c < 8) scond[row * 8 + c] = c < SD + 1 ? v : 0.f;
          if (c < SD + 1 && row < L) p.qkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] = v;
        }
      }
    }
    SB_STAMP(4);
    __syncthreads();                       // every wave is done with the ROWS planes; q / k / v / cond visible
    float *sp = reinterpret_cast<float *>(xs);                      // P [64][68], then the pairwise slab
    const float *plb = sp + TM * (TM + 4) + pls.mis;
    if (plt) {
#pragma unroll
      for (int k = 0; k < msr3d_attn::kPlocRegs; ++k) {
        const int e = tid + 256 * k;
        if (e < pls.n4) reinterpret_cast<float4 *>(sp + TM * (TM + 4))[e] = plv[k];
      }
    }
    __syncthreads();
    SB_STAMP(5);
    f32x4 o[2];
    msr3d_attn::attn_fwd_core<TM, kCoreMma>(L, sq, sk, sv, sp, plb, scond, 8, p.pad + (size_t)b * L,
                                                 p.probs ? p.probs + ((size_t)b * H + h) * L * L : nullptr, o);
    // ctx_h: side output + product 2's operand (FRAG planes, one slab; row tile = wave)
    if (wave < 4) {
#pragma unroll
      for (int rn = 0; rn < 2; ++rn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = 4 * g + r, row = 16 * wave + rr, kk = 16 * rn + j;
          const float v = row < L ? o[rn][r] : 0.f;
          if (row < L) p.ctx[(size_t)(row_base + row) * KD + h * DH + kk] = v;
          unsigned short pl[3];
          sm_split1(v, pl);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            *reinterpret_cast<unsigned short *>(ctxp + (((wave * 3 + k) * 64 + rr + 16 * (kk >> 3)) * 16 + (kk & 7) * 2)) = pl[k];
        }
    }
    SB_STAMP(6);
    __syncthreads();
    f32x4 acc2[RN2][4];
    zero_acc3(acc2);
    const XFrag<4> xm{reinterpret_cast<const unsigned short *>(ctxp) + lane * 8};
    gemm_split3<true, RN2, 4, 1, RING2>(xm, 0, w2, acc2, ring2);
    SB_STAMP(7);
    store_partials<RN2>(acc2, slab, row_base, L, 16 * RN2 * wave, lane);
  } else {
    // ------------------------------------------------------------------ attention block, backward
    const int h = slice, H = p.H, ldq = p.ldq;
    float *sq = reinterpret_cast<float *>(aux), *sk = sq + TM * LD32, *sv = sk + TM * LD32, *sdo = sv + TM * LD32;
    float *scond = sdo + TM * LD32, *sdc = scond + TM * 8;
    // product 2's stream: the head's gathered rows of W_qkvc as [k = 128 head columns][n = 256]: [4 slabs][16 tiles]
    const WStream w2 = make_wstream(p.w2 + (size_t)h * (4 * 16 * kPieceBytes / 2), 4 * 16 * kPieceBytes, 16, 0, 4 * wave, lane);
    // the head's saved q, k, v, cond
    msr3d_attn::load_head_tile<TM>(p.qkvc, ldq, b, h, L, sq);
    msr3d_attn::load_head_tile<TM>(p.qkvc + KD, ldq, b, h, L, sk);
    msr3d_attn::load_head_tile<TM>(p.qkvc + 2 * KD, ldq, b, h, L, sv);
    for (int e = tid; e < TM * 8; e += 256) {
      const int row = e >> 3, c = e & 7;
      scond[e] = (row < L && c < SD + 1) ? p.qkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] : 0.f;
    }
    const float *plsrc = p.ploc + (size_t)b * L * L * SD, *prsrc = p.probs + ((size_t)b * H + h) * L * L;
    const int pn = L * L * SD;
    const bool pvec = (reinterpret_cast<uintptr_t>(plsrc) & 15u) == 0 && (pn & 3) == 0;
    const bool qvec = (reinterpret_cast<uintptr_t>(prsrc) & 15u) == 0 && (L & 3) == 0;
    float4 plv[msr3d_attn::kPlocRegs], prv[msr3d_attn::kProbRegs];
    if (pvec) msr3d_attn::ploc_fetch(plsrc, pn >> 2, plv);
    if (qvec) msr3d_attn::probs_fetch(prsrc, L, prv);
    // d ctx_h = d_fc Wfc[:, 32 h : 32 h + 32]: wave (wr, wc) owns row tiles 2 wr, 2 wr + 1 and column tile wc
    f32x4 acc[1][2];
    zero_acc3(acc);
    gemm_split3<true, 1, 2, KS1, RING>(xr, 2 * (wave >> 1), w1, acc, ring1);
    SB_STAMP(3);
    WPiece ring2[RING];
    preload_wring<4, RING>(ring2, w2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = 16 * (2 * (wave >> 1) + mt) + j;
      const float4 v = row < L ? make_float4(acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3])
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      st4(sdo + row * LD32 + 16 * (wave & 1) + 4 * g, v);
    }
    __syncthreads();                       // planes free; q / k / v / cond / d ctx visible
    float *sp = reinterpret_cast<float *>(xs);
    if (qvec) msr3d_attn::probs_store(sp, L, prv);
    else msr3d_attn::load_probs_tile<TM>(prsrc, L, sp);
    const float *plb = sp + TM * (TM + 4);
    if (pvec) msr3d_attn::ploc_store(sp + TM * (TM + 4), pn >> 2, plv);
    else plb = msr3d_attn::stage_ploc<TM>(p.ploc, b, L, sp + TM * (TM + 4));
    __syncthreads();
    SB_STAMP(5);
    f32x4 oq[2], ok[2], ov[2];
    msr3d_attn::attn_bwd_core<TM, kCoreMma>(L, sq, sk, sv, sdo, sp, plb, scond, 8, p.pad + (size_t)b * L, sdc, 8,
                                                 oq, ok, ov);
    SB_STAMP(6);
    // every wave is past the core's last barrier: the pairwise slab is dead, product 2's operand
    // (FRAG planes, 4 slabs: [dq | dk | dv | dcond, 0]) goes on top of it
    unsigned char *mid = reinterpret_cast<unsigned char *>(sp + TM * (TM + 4));
#pragma unroll
    for (int rn = 0; rn < 2; ++rn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r, row = 16 * wave + rr, kk = 16 * rn + j;
        const bool okr = row < L;
        const float v3[3] = {okr ? oq[rn][r] * kInvSqrtDh : 0.f, okr ? ok[rn][r] * kInvSqrtDh : 0.f, okr ? ov[rn][r] : 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (okr) p.dqkvc[(size_t)(row_base + row) * ldq + a * KD + h * DH + kk] = v3[a];
          unsigned short pl[3];
          sm_split1(v3[a], pl);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            *reinterpret_cast<unsigned short *>(mid + ((((a * 4 + wave) * 3 + k) * 64 + rr + 16 * (kk >> 3)) * 16 + (kk & 7) * 2)) = pl[k];
        }
      }
    if (tid < TM) {                        // slab 3 of row `tid`: the six cond gradients, zeros behind them
      const int row = tid;
      float dc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) dc[c] = (row < L && c < SD + 1) ? sdc[row * 8 + c] : 0.f;
      if (row < L) {
#pragma unroll
        for (int c = 0; c < SD + 1; ++c) p.dqkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] = dc[c];
      }
      uint4 pl[3];
      sm_split8(dc, pl);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned char *d = mid + (((3 * 4 + (row >> 4)) * 3 + k) * 64 + (row & 15)) * 16;
        *reinterpret_cast<uint4 *>(d) = pl[k];
#pragma unroll
        for (int gg = 1; gg < 4; ++gg) *reinterpret_cast<uint4 *>(d + gg * 256) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    __syncthreads();
    f32x4 acc2[4][4];
    zero_acc3(acc2);
    const XFrag<4> xm{reinterpret_cast<const unsigned short *>(mid) + lane * 8};
    gemm_split3<true, 4, 4, 4, RING>(xm, 0, w2, acc2, ring2);
    SB_STAMP(7);
    store_partials<4>(acc2, slab, row_base, L, 64 * wave, lane);
  }
  SB_STAMP(8);
}

// =====================================================================================================
// Attention block, forward, as TWO workgroups per (scene, head) (round 5): workgroup
// (h, qh) owns the query rows [32 qh, 32 qh + 32) -- q, cond, the probabilities, the context and the head's
// partial out-projection of those rows -- and computes k and v of ALL rows for itself (each workgroup saves
// the k / v rows of its own half for the backward).  256 workgroups at the bench shape instead of 128, and
// inside the core the eight waves split the KEYS as well: wave (qt, kq) owns the 16 x 16 logits of query
// tile qt and key quarter kq (4 (query, key) pairs per lane instead of 16); the quarters' row maxima and row
// sums meet through LDS in quarter order (deterministic).
//
// Column tiles of product 1 by wave -- k0 k1 v0 v1 (all four row tiles) | q0 q1 cond (this half's two row
// tiles) | idle -- so that the two waves of a SIMD (w, w + 4) issue 6, 6, 6 and 4 tile-units.
// =====================================================================================================
constexpr int kPloc2Regs = 32 * 64 * SD / 4 / 256;          // 10 float4 per thread of waves 4 .. 7: the half's (32, L, 5) slab
constexpr int kAttnFwd2Aux = 3 * kTile + 2048 + 2 * 3 * 1024 + 2 * 4 * 32 * 4;   // q k v | cond | ctx FRAG (2 row tiles) | max, sum
constexpr int kAttnFwd2Lds = XS_BYTES + kAttnFwd2Aux;

__global__ __launch_bounds__(512) void scene_attn_fwd2_kernel(const SB p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
  unsigned char *aux = smem + XS_BYTES;
#if MSR3D_ATTN_SCENE_XCD
  int h, qh, b;
  if ((gridDim.y & 7) == 0) {                // a scene's sixteen workgroups on ONE XCD (workgroup id mod 8)
    const int id = blockIdx.y * 16 + blockIdx.x, c = id & 7, jj = id >> 3, x = jj & 15;
    b = c + 8 * (jj >> 4); h = x & 7; qh = x >> 3;
  } else { h = blockIdx.x & 7; qh = blockIdx.x >> 3; b = blockIdx.y; }
  const int L = p.L, H = p.H, ldq = p.ldq;
#else
  const int h = blockIdx.x & 7, qh = blockIdx.x >> 3, b = blockIdx.y, L = p.L, H = p.H, ldq = p.ldq;
#endif
  const int q0 = 32 * qh;
  if (q0 >= L) return;                       // (a scene of <= 32 tokens: the first workgroup owns every row)
  const int row_base = b * L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float *const slab = p.part + (size_t)h * p.part_stride;
  float *sq = reinterpret_cast<float *>(aux), *sk = sq + TM * LD32, *sv = sk + TM * LD32;
  float *scond = sv + TM * LD32;                                           // [64][8]
  unsigned char *ctxp = reinterpret_cast<unsigned char *>(scond + TM * 8); // FRAG, 1 slab x 2 row tiles
  float *xmax = reinterpret_cast<float *>(ctxp + 2 * 3 * 1024), *xsum = xmax + 4 * 32;   // [4 quarters][32 rows] each

  constexpr int KS1 = KD / 32;
  const int tile = wave < 4 ? 2 + wave : wave == 7 ? 7 : wave - 4 + (wave == 6 ? 4 : 0);   // k0 k1 v0 v1 q0 q1 cond -
  const bool kv = wave < 4, busy = wave < 7;
  WStream w1 = make_wstream(p.w1 + (size_t)h * (KS1 * 8 * kPieceBytes / 2), KS1 * 8 * kPieceBytes, 8, 0, tile, lane);
  SB_STAMP(0);
  WPiece ring1[RING];
  if (busy) preload_wring<1, RING>(ring1, w1);
  // this half's rows of the scene's pairwise slab: registers now, LDS when the planes are dead
  const int nq = min(32, L - q0);
  const float *plsrc = p.ploc + ((size_t)b * L + q0) * L * SD;
  const int pn = nq * L * SD;
  const Slab4 pls = make_slab4(plsrc, pn);
  constexpr bool pvec = true;            // (any alignment, any count: Slab4)
  float4 plv[kPloc2Regs];
  // The scene's planes, THIS half's 32 rows first and the other half's only when those have arrived: the pair's two
  // workgroups (same XCD) then miss on disjoint halves and hit, in L2, on what the partner has fetched -- all 64 rows
  // requested at once by both doubled the traffic over the fabric and the staging time (5.3 k -> 11.6 k cycles).
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(p.xp + (size_t)b * 3 * TM * KD);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const int r0 = 32 * (part == 0 ? qh : 1 - qh);
      uint4 v[6];
#pragma unroll
      for (int k = 0; k < 2 * kPlanes; ++k) {
        const int q = tid + 512 * k, plane = q >> 10, row = r0 + ((q >> 5) & 31), c8 = q & 31;
        v[k] = src[plane * 2048 + row * 32 + c8];
      }
#pragma unroll
      for (int k = 0; k < 2 * kPlanes; ++k) {
        const int q = tid + 512 * k, plane = q >> 10, row = r0 + ((q >> 5) & 31), c8 = q & 31;
        *reinterpret_cast<uint4 *>(xs + plane * PLANE + row * PITCH + c8 * 8) = v[k];
      }
      if (part == 0) __builtin_amdgcn_sched_barrier(0);       // (the second half's loads stay behind the first half's stores)
      if (part == 1) {
        // the half's pairwise rows last: they return behind the planes (loads complete in order) and are not needed
        // before product 1 is over -- their 1.15 MB per XCD cross the fabric under the product, not in front of it.
        // By waves 4 .. 7 only (q, cond, idle: half the product of the k / v waves, whose weight pieces would queue
        // behind these loads).
        __builtin_amdgcn_sched_barrier(0);
        if (pvec && !kv) {
#pragma unroll
          for (int k2 = 0; k2 < kPloc2Regs; ++k2) plv[k2] = slab4_load(pls, tid - 256 + 256 * k2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  SB_STAMP(1);
  __syncthreads();
  SB_STAMP(2);
  const XRows xr = make_xrows(xs, PITCH, TM, lane);
  // product 2's stream: Wfc [8 slabs][16 tiles], slab h, two column tiles a wave
  const WStream w2 = make_wstream(p.w2, p.w2_bytes, 16, h, 2 * wave, lane);
  WPiece ring2[2];

  if (kv) {
    f32x4 acc[1][4];
    zero_acc3(acc);
    gemm_split3<true, 1, 4, KS1, RING>(xr, 0, w1, acc, ring1);
    SB_STAMP(3);
    preload_wring<2, 2>(ring2, w2);
    const int which = 1 + (wave >> 1);                                      // 1 k, 2 v
    float *t = sq + which * TM * LD32;
    const int c = 16 * (wave & 1) + 4 * g;
    const float4 bv = ld4(p.bias1 + which * KD + h * DH + c);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = 16 * mt + j;
      const bool ok = row < L;
      const float4 v = ok ? make_float4(acc[0][mt][0] + bv.x, acc[0][mt][1] + bv.y, acc[0][mt][2] + bv.z, acc[0][mt][3] + bv.w)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
      st4(t + row * LD32 + c, v);
      if (ok && (mt >> 1) == qh) st4(p.qkvc + (size_t)(row_base + row) * ldq + which * KD + h * DH + c, v);
    }
  } else if (busy) {
    f32x4 acc[1][2];
    zero_acc3(acc);
    gemm_split3<true, 1, 2, KS1, RING>(xr, 2 * qh, w1, acc, ring1);
    SB_STAMP(3);
    preload_wring<2, 2>(ring2, w2);
    if (wave < 6) {                                                         // q
      const int c = 16 * (wave & 1) + 4 * g;
      const float4 bv = ld4(p.bias1 + h * DH + c);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = q0 + 16 * mt + j;
        const bool ok = row < L;
        const float4 v = ok ? make_float4(acc[0][mt][0] + bv.x, acc[0][mt][1] + bv.y, acc[0][mt][2] + bv.z, acc[0][mt][3] + bv.w)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        st4(sq + row * LD32 + c, v);
        if (ok) st4(p.qkvc + (size_t)(row_base + row) * ldq + h * DH + c, v);
      }
    } else {                                                                // cond
      float b4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) b4[r] = 4 * g + r < SD + 1 ? p.bias1[3 * KD + h * (SD + 1) + 4 * g + r] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = q0 + 16 * mt + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * g + r;
          const float v = row < L ? acc[0][mt][r] + b4[r] : 0.f;
          if (c < 8) scond[row * 8 + c] = c < SD + 1 ? v : 0.f;
          if (c < SD + 1 && row < L) p.qkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] = v;
        }
      }
    }
  } else {
    SB_STAMP(3);
    preload_wring<2, 2>(ring2, w2);
  }
  SB_STAMP(4);
  __syncthreads();                           // every wave is done with the ROWS planes; q / k / v / cond visible
  constexpr int LDP = TM + 4;
  float *sp = reinterpret_cast<float *>(xs);                                // P [32][68], then the half's pairwise slab
  float *spl = sp + 32 * LDP;
  if (!kv) {
#pragma unroll
    for (int k = 0; k < kPloc2Regs; ++k) {
      const int e = tid - 256 + 256 * k;
      if (e < pls.n4) reinterpret_cast<float4 *>(spl)[e] = plv[k];
    }
  }
  spl += pls.mis;                                                           // (element i of the slab)
  __syncthreads();
  SB_STAMP(5);
  // ---- the core: wave (qt, kq) ----
  {
    const int qt = wave & 1, kq = wave >> 1;
    const int i = j;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    msr3d_attn::strip_mma<kCoreMma, 1, DH, true, true>(sq, LD32, sk + 16 * kq * LD32, LD32, q0 + 16 * qt, acc, lane);
    const int col = 16 * kq + i;
    const unsigned char *pad_b = p.pad + (size_t)b * L;
    const bool keyok = col < L && !pad_b[min(col, L - 1)];
    float lg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = 16 * qt + 4 * g + r, row = q0 + lr;
      const msr3d_attn::RowCond c = msr3d_attn::load_cond(scond, 8, L, row);
      float v = -INFINITY;
      if (row < L && keyok) {
        const float *pl = spl + ((size_t)lr * L + col) * SD;
        float z = c.bias;
#pragma unroll
        for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
        const float loc = __builtin_amdgcn_rcpf(1.0f + __expf(-z));      // (attn_core.h: the reference's sigmoid -> clamp -> log)
        v = __logf(fmaxf(loc, 1e-6f)) + acc[0][r] * kInvSqrtDh;
      }
      lg[r] = v;
      const float m = msr3d_attn::row16_max(v);
      if (i == 0) xmax[kq * 32 + lr] = m;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = 16 * qt + 4 * g + r;
      const float m = fmaxf(fmaxf(xmax[lr], xmax[32 + lr]), fmaxf(xmax[64 + lr], xmax[96 + lr]));
      const float e = (lg[r] == -INFINITY) ? 0.f : __expf(lg[r] - m);
      lg[r] = e;
      const float sm = msr3d_attn::row16_sum(e);
      if (i == 0) xsum[kq * 32 + lr] = sm;
    }
    __syncthreads();
    float *probs_bh = p.probs ? p.probs + ((size_t)b * H + h) * L * L : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = 16 * qt + 4 * g + r, row = q0 + lr;
      const float tot = ((xsum[lr] + xsum[32 + lr]) + xsum[64 + lr]) + xsum[96 + lr];
      const float inv = 1.0f / tot;          // a fully padded row gives NaN, as the reference would
      const float pr = (row < L) ? lg[r] * inv : 0.f;
      sp[lr * LDP + col] = pr;
      if (probs_bh && row < L && col < L) probs_bh[(size_t)row * L + col] = pr;
    }
    __syncthreads();
    // ctx = P V: waves 0 .. 3 as (query tile, half of the head's 32 columns); side output + product 2's operand
    if (wave < 4) {
      const int dt = wave >> 1;
      f32x4 o[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      msr3d_attn::strip_mma<kCoreMma, 1, TM, true, false>(sp, LDP, sv + 16 * dt, LD32, 16 * qt, o, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * g + r, row = q0 + 16 * qt + rr, kk = 16 * dt + i;
        const float v = row < L ? o[0][r] : 0.f;
        if (row < L) p.ctx[(size_t)(row_base + row) * KD + h * DH + kk] = v;
        unsigned short pl[3];
        sm_split1(v, pl);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          *reinterpret_cast<unsigned short *>(ctxp + (((qt * 3 + k) * 64 + rr + 16 * (kk >> 3)) * 16 + (kk & 7) * 2)) = pl[k];
      }
    }
  }
  SB_STAMP(6);
  __syncthreads();
  f32x4 acc2[2][2];
  zero_acc3(acc2);
  const XFrag<2> xm{reinterpret_cast<const unsigned short *>(ctxp) + lane * 8};
  gemm_split3<true, 2, 2, 1, 2>(xm, 0, w2, acc2, ring2);
  SB_STAMP(7);
#pragma unroll
  for (int rn = 0; rn < 2; ++rn)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = q0 + 16 * mt + j;
      if (row < L)
        st4(slab + (size_t)(row_base + row) * KD + 32 * wave + 16 * rn + 4 * g,
            make_float4(acc2[rn][mt][0], acc2[rn][mt][1], acc2[rn][mt][2], acc2[rn][mt][3]));
    }
  SB_STAMP(8);
}

// =====================================================================================================
// Attention block, backward (one workgroup of four waves per (scene, head), as scene_block_kernel<ATTN_BWD> was), with
// ALL of the block's inputs crossing the fabric together: the head's saved q / k / v / cond and probabilities are asked
// for with the planes (they used to be fetched, one dependent round trip after another, once the planes had arrived:
// 13 k cycles in front of an 8-piece product), all eight weight pieces of product 1 sit in registers before the pairwise
// slab's 72 KB are requested (loads complete in order), and the core returns dq / dk / dv transposed per tile
// (attn_core.h: strip_mma's SWAP) so that the side outputs are 16-byte stores and product 2's operand 8-byte LDS stores.
// MSR3D_ATTN_BWD_V2=0 restores scene_block_kernel<ATTN_BWD>.
// =====================================================================================================
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void scene_attn_bwd2_kernel(const SB p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
  unsigned char *aux = smem + XS_BYTES;
  const int slice = blockIdx.x, b = blockIdx.y, L = p.L;
  const int row_base = b * L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float *const slab = p.part + (size_t)slice * p.part_stride;
  constexpr int KS1 = KD / 32;
  {
    const int h = slice, H = p.H, ldq = p.ldq;
    float *sq = reinterpret_cast<float *>(aux), *sk = sq + TM * LD32, *sv = sk + TM * LD32, *sdo = sv + TM * LD32;
    float *scond = sdo + TM * LD32, *sdc = scond + TM * 8;
    // Wfc as [k = fc row][n = ctx column]: [8 slabs][16 tiles], tiles 2 h, 2 h + 1
    const WStream w1 = make_wstream(p.w1, p.w1_bytes, 16, 0, 2 * slice + (wave & 1), lane);
    SB_STAMP(0);
    WPiece ring1[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) load_wpiece(ring1[q], w1, q, 0);
    uint4 xv[24];
    planes_fetch(p.xp, b, xv);
    // saved q / k / v (two float4 a thread and tile), cond, probabilities
    float4 tq[6];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + 256 * k, row = e >> 3, c4 = (e & 7) * 4;
        tq[2 * t + k] = row < L ? ld4(p.qkvc + (size_t)(row_base + row) * ldq + t * KD + h * DH + c4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float tc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 256 * k, row = e >> 3, c = e & 7;
      tc[k] = (row < L && c < SD + 1) ? p.qkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] : 0.f;
    }
    const float *plsrc = p.ploc + (size_t)b * L * L * SD, *prsrc = p.probs + ((size_t)b * H + h) * L * L;
    const int pn = L * L * SD;
    const bool pvec = (reinterpret_cast<uintptr_t>(plsrc) & 15u) == 0 && (pn & 3) == 0;
    const bool qvec = (reinterpret_cast<uintptr_t>(prsrc) & 15u) == 0 && (L & 3) == 0;
    float4 plv[msr3d_attn::kPlocRegs], prv[msr3d_attn::kProbRegs];
    if (qvec) msr3d_attn::probs_fetch(prsrc, L, prv);
    {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = tid + 256 * k, row = e >> 3, c4 = (e & 7) * 4;
          st4(sq + t * TM * LD32 + row * LD32 + c4, tq[2 * t + k]);
        }
#pragma unroll
      for (int k = 0; k < 2; ++k) scond[tid + 256 * k] = tc[k];
      planes_store(xs, xv);
    }
    // the planes' registers are free: the rest of product 1's pieces, then the pairwise slab (consumed after product 1).
    // (Their addresses pass through an empty asm so that no pass hoists these loads above the planes' stores.  No
    //  sched_barrier in this prologue: with them the planes went to scratch, one serialised load at a time.)
    {
      WStream w1b = w1;
      const float *plsrc2 = plsrc;
      asm volatile("" : "+s"(w1b.soff), "+s"(plsrc2) : : "memory");
#pragma unroll
      for (int q = 4; q < 8; ++q) load_wpiece(ring1[q], w1b, q, 0);
      if (pvec) msr3d_attn::ploc_fetch(plsrc2, pn >> 2, plv);
    }
    SB_STAMP(1);
    __syncthreads();
    SB_STAMP(2);
    const XRows xr = make_xrows(xs, PITCH, TM, lane);
    // ------------------------------------------------------------------ attention block, backward
    // product 2's stream: the head's gathered rows of W_qkvc as [k = 128 head columns][n = 256]: [4 slabs][16 tiles]
    const WStream w2 = make_wstream(p.w2 + (size_t)h * (4 * 16 * kPieceBytes / 2), 4 * 16 * kPieceBytes, 16, 0, 4 * wave, lane);
    // d ctx_h = d_fc Wfc[:, 32 h : 32 h + 32]: wave (wr, wc) owns row tiles 2 wr, 2 wr + 1 and column tile wc
    f32x4 acc[1][2];
    zero_acc3(acc);
    gemm_split3<true, 1, 2, KS1, 8>(xr, 2 * (wave >> 1), w1, acc, ring1);
    SB_STAMP(3);
    WPiece ring2[RING];
    preload_wring<4, RING>(ring2, w2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = 16 * (2 * (wave >> 1) + mt) + j;
      const float4 v = row < L ? make_float4(acc[0][mt][0], acc[0][mt][1], acc[0][mt][2], acc[0][mt][3])
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      st4(sdo + row * LD32 + 16 * (wave & 1) + 4 * g, v);
    }
    __syncthreads();                       // planes free; q / k / v / cond / d ctx visible
    float *sp = reinterpret_cast<float *>(xs);
    if (qvec) msr3d_attn::probs_store(sp, L, prv);
    else msr3d_attn::load_probs_tile<TM>(prsrc, L, sp);
    const float *plb = sp + TM * (TM + 4);
    if (pvec) msr3d_attn::ploc_store(sp + TM * (TM + 4), pn >> 2, plv);
    else plb = msr3d_attn::stage_ploc<TM>(p.ploc, b, L, sp + TM * (TM + 4));
    __syncthreads();
    SB_STAMP(5);
    f32x4 oq[2], ok[2], ov[2];
    msr3d_attn::attn_bwd_core<TM, kCoreMma, true>(L, sq, sk, sv, sdo, sp, plb, scond, 8, p.pad + (size_t)b * L, sdc, 8,
                                                       oq, ok, ov);
    SB_STAMP(6);
    // every wave is past the core's last barrier: the pairwise slab is dead, product 2's operand
    // (FRAG planes, 4 slabs: [dq | dk | dv | dcond, 0]) goes on top of it.  The core hands dq / dk / dv over transposed:
    // a lane holds columns 16 rn + 4 g .. + 3 of row 16 wave + j -- one 16-byte side-output store and one 8-byte LDS
    // store per plane (they were four 4-byte and twelve 2-byte stores per tile and operand: 11 k cycles of the block)
    unsigned char *mid = reinterpret_cast<unsigned char *>(sp + TM * (TM + 4));
    {
      const int row = 16 * wave + j;
      const bool okr = row < L;
#pragma unroll
      for (int rn = 0; rn < 2; ++rn)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const f32x4 &src = a == 0 ? oq[rn] : a == 1 ? ok[rn] : ov[rn];
          const float sc = a < 2 ? kInvSqrtDh : 1.0f;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = okr ? src[r] * sc : 0.f;
          if (okr) st4(p.dqkvc + (size_t)(row_base + row) * ldq + a * KD + h * DH + 16 * rn + 4 * g, make_float4(v[0], v[1], v[2], v[3]));
          uint2 pl[3];
          sm_split4(v, pl);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            *reinterpret_cast<uint2 *>(mid + frag_off4<4>(wave, j, 32 * a + 16 * rn + 4 * g, k)) = pl[k];
        }
    }
    if (tid < TM) {                        // slab 3 of row `tid`: the six cond gradients, zeros behind them
      const int row = tid;
      float dc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) dc[c] = (row < L && c < SD + 1) ? sdc[row * 8 + c] : 0.f;
      if (row < L) {
#pragma unroll
        for (int c = 0; c < SD + 1; ++c) p.dqkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] = dc[c];
      }
      uint4 pl[3];
      sm_split8(dc, pl);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned char *d = mid + (((3 * 4 + (row >> 4)) * 3 + k) * 64 + (row & 15)) * 16;
        *reinterpret_cast<uint4 *>(d) = pl[k];
#pragma unroll
        for (int gg = 1; gg < 4; ++gg) *reinterpret_cast<uint4 *>(d + gg * 256) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    __syncthreads();
    f32x4 acc2[4][4];
    zero_acc3(acc2);
    const XFrag<4> xm{reinterpret_cast<const unsigned short *>(mid) + lane * 8};
    gemm_split3<true, 4, 4, 4, RING>(xm, 0, w2, acc2, ring2);
    SB_STAMP(7);
    store_partials<4>(acc2, slab, row_base, L, 64 * wave, lane);
  }
  SB_STAMP(8);
}

// =====================================================================================================
// Attention block, backward, EIGHT waves (two per SIMD) with the core's work split over all of them
// (scene_attn_bwd2_kernel's four waves each carried 16 (query, key) pairs per lane through the softmax /
// sigmoid backward: 14-15 k of the block's 47 k cycles).  Wave (t, kh): query / key tile t, key half kh.
//   d P     16 queries x the half's 32 keys                      (was: x 64)
//   d v     tile t's 16 keys x column half kh, over all queries   (was: x 32 columns)
//   softmax / sigmoid backward on 8 pairs per lane; the row's  sum_k p dP  and the six cond gradients are partial
//           sums over a key half that meet through LDS in half order (deterministic)
//   d q, d k  tile t's 16 rows x column half kh
// Product 1 (d ctx) is one tile-unit per wave, product 2 two column tiles per wave.  Everything else as bwd2: inputs
// requested up front, product 1's eight pieces in registers before the pairwise slab is asked for, transposed core
// outputs.  MSR3D_ATTN_BWD=2 selects scene_attn_bwd2_kernel, =0 scene_block_kernel<ATTN_BWD>.
// =====================================================================================================
constexpr int kAttnBwd3Aux = kAttnBwdAux + 2 * 64 * 4 + 2 * 64 * 8 * 4;     // + row dots [2][64] + cond partials [2][64][8]
constexpr int kAttnBwd3Lds = XS_BYTES + kAttnBwd3Aux;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void scene_attn_bwd3_kernel(const SB p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
  unsigned char *aux = smem + XS_BYTES;
#if MSR3D_ATTN_SCENE_XCD
  int h, b;
  if ((gridDim.y & 7) == 0) {
    const int id = blockIdx.y * 8 + blockIdx.x, c = id & 7, jj = id >> 3;
    b = c + 8 * (jj >> 3); h = jj & 7;
  } else { h = blockIdx.x; b = blockIdx.y; }
  const int L = p.L, H = p.H, ldq = p.ldq;
#else
  const int h = blockIdx.x, b = blockIdx.y, L = p.L, H = p.H, ldq = p.ldq;
#endif
  const int row_base = b * L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float *const slab = p.part + (size_t)h * p.part_stride;
  constexpr int KS1 = KD / 32, LDP = TM + 4;
  float *sq = reinterpret_cast<float *>(aux), *sk = sq + TM * LD32, *sv = sk + TM * LD32, *sdo = sv + TM * LD32;
  float *scond = sdo + TM * LD32, *sdc = scond + TM * 8;
  float *xdot = sdc + TM * 8, *xg = xdot + 2 * 64;
  // Wfc as [k = fc row][n = ctx column]: [8 slabs][16 tiles], tiles 2 h, 2 h + 1
  const WStream w1 = make_wstream(p.w1, p.w1_bytes, 16, 0, 2 * h + (wave & 1), lane);
  SB_STAMP(0);
  WPiece ring1[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) load_wpiece(ring1[q], w1, q, 0);
  // the scene's planes, the head's saved q / k / v (one float4 a thread and tile), cond, probabilities
  uint4 xv[12];
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(p.xp + (size_t)b * 3 * TM * KD);
#pragma unroll
    for (int k = 0; k < 4 * kPlanes; ++k) xv[k] = src[tid + 512 * k];
  }
  float4 tq[3];
  {
    const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      tq[t] = row < L ? ld4(p.qkvc + (size_t)(row_base + row) * ldq + t * KD + h * DH + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float tc;
  {
    const int row = tid >> 3, c = tid & 7;
    tc = (row < L && c < SD + 1) ? p.qkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] : 0.f;
  }
  const float *plsrc = p.ploc + (size_t)b * L * L * SD, *prsrc = p.probs + ((size_t)b * H + h) * L * L;
  const int pn = L * L * SD, qn = L * L;
  // both slabs as 16-byte vectors whatever their alignment (L = 61: the agent token of 'as_object'): Slab4
  const Slab4 pls = make_slab4(plsrc, pn), prs = make_slab4(prsrc, qn);
  const bool qrow4 = prs.mis == 0 && (L & 3) == 0;      // a vector of probabilities never straddles a row
  float4 prv[2], plv[10];
  // (the slab is asked for WITH the planes: behind them and ahead of product 1's last pieces it was waited for with those
  //  pieces -- loads complete in order)
#pragma unroll
  for (int k = 0; k < 10; ++k) plv[k] = slab4_load(pls, tid + 512 * k);
#pragma unroll
  for (int k = 0; k < 2; ++k) prv[k] = slab4_load(prs, tid + 512 * k);
  {
    const int row = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int t = 0; t < 3; ++t) st4(sq + t * TM * LD32 + row * LD32 + c4, tq[t]);
    scond[tid] = tc;
#pragma unroll
    for (int k = 0; k < 4 * kPlanes; ++k) {
      const int q = tid + 512 * k;
      const int plane = q >> 11, row2 = (q >> 5) & 63, c8 = q & 31;
      *reinterpret_cast<uint4 *>(xs + plane * PLANE + row2 * PITCH + c8 * 8) = xv[k];
    }
  }
  // (the planes' registers are free) the rest of product 1's pieces
#pragma unroll
  for (int q = 4; q < 8; ++q) load_wpiece(ring1[q], w1, q, 0);
  SB_STAMP(1);
  __syncthreads();
  SB_STAMP(2);
  const XRows xr = make_xrows(xs, PITCH, TM, lane);
  // product 2's stream: the head's gathered rows of W_qkvc as [k = 128 head columns][n = 256]: [4 slabs][16 tiles]
  const WStream w2 = make_wstream(p.w2 + (size_t)h * (4 * 16 * kPieceBytes / 2), 4 * 16 * kPieceBytes, 16, 0, 2 * wave, lane);
  {
    // d ctx_h = d_fc Wfc[:, 32 h : 32 h + 32]: wave (rt, ct) owns row tile rt and column tile ct
    const int rt = wave >> 1;
    f32x4 acc[1][1];
    zero_acc3(acc);
    gemm_split3<true, 1, 1, KS1, 8>(xr, rt, w1, acc, ring1);
    const int row = 16 * rt + j;
    const float4 v = row < L ? make_float4(acc[0][0][0], acc[0][0][1], acc[0][0][2], acc[0][0][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    st4(sdo + row * LD32 + 16 * (wave & 1) + 4 * g, v);
  }
  SB_STAMP(3);
  WPiece ring2[RING];
  preload_wring<2, RING>(ring2, w2);
  __syncthreads();                         // planes free; q / k / v / cond / d ctx visible
  float *sp = reinterpret_cast<float *>(xs);
  float *spl = sp + TM * LDP;
  if (qrow4) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 512 * k;
      if (e < prs.n4) {
        const int idx = 4 * e, row = idx / L, col = idx - row * L;
        *reinterpret_cast<float4 *>(sp + row * LDP + col) = prv[k];
      }
    }
  } else {
    const float invL = 1.0f / (float)L;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + 512 * k;
      const float v4[4] = {prv[k].x, prv[k].y, prv[k].z, prv[k].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = 4 * e + q - prs.mis;
        if (idx >= 0 && idx < qn) {
          const int row = (int)(((float)idx + 0.5f) * invL);      // exact for idx < 2^22: idx / L
          sp[row * LDP + idx - row * L] = v4[q];
        }
      }
    }
  }
  for (int e = tid; e < 64 * 64; e += 512) {
    const int row = e >> 6, col = e & 63;
    if (row >= L || col >= L) sp[row * LDP + col] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int e = tid + 512 * k;
    if (e < pls.n4) reinterpret_cast<float4 *>(spl)[e] = plv[k];
  }
  const float *plb = spl + pls.mis;
  __syncthreads();
  SB_STAMP(5);
  // ---- the core: wave (t, kh) ----
  const int t = wave & 3, kh = wave >> 2, row0 = 16 * t, i = j;
  f32x4 oq[1], ok[1], ov[1];
  {
    f32x4 dP[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    msr3d_attn::strip_mma<kCoreMma, 2, DH, true, true>(sdo, LD32, sv + 32 * kh * LD32, LD32, row0, dP, lane);
    ov[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    msr3d_attn::strip_mma<kCoreMma, 1, TM, false, false, true>(sp, LDP, sdo + 16 * kh, LD32, row0, ov, lane);
    __syncthreads();                       // every wave is done reading P as a matrix operand
    const unsigned char *pad_b = p.pad + (size_t)b * L;
    bool keyok[2];
#pragma unroll
    for (int rn = 0; rn < 2; ++rn) {
      const int col = 32 * kh + 16 * rn + i;
      keyok[rn] = col < L && !pad_b[min(col, L - 1)];
    }
    float pr[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      float dot = 0.f;
#pragma unroll
      for (int rn = 0; rn < 2; ++rn) {
        pr[r][rn] = sp[row * LDP + 32 * kh + 16 * rn + i];
        dot = fmaf(pr[r][rn], dP[rn][r], dot);
      }
      dot = msr3d_attn::row16_sum(dot);
      if (i == 0) xdot[kh * 64 + row] = dot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * g + r;
      const float dot = xdot[row] + xdot[64 + row];
      const msr3d_attn::RowCond c = msr3d_attn::load_cond(scond, 8, L, row);
      float gb = 0.f, gw[SD];
#pragma unroll
      for (int d = 0; d < SD; ++d) gw[d] = 0.f;
#pragma unroll
      for (int rn = 0; rn < 2; ++rn) {
        const int col = 32 * kh + 16 * rn + i;
        const float dlogit = pr[r][rn] * (dP[rn][r] - dot);          // softmax backward
        sp[row * LDP + col] = dlogit;                                // in place: this lane owns the element
        if (row < L && keyok[rn]) {
          const float *pl = plb + ((size_t)row * L + col) * SD;
          float z = c.bias;
#pragma unroll
          for (int d = 0; d < SD; ++d) z = fmaf(c.w[d], pl[d], z);
          const float loc = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
          const float dz = (loc >= 1e-6f) ? dlogit * (1.0f - loc) : 0.f;   // (attn_core.h: d log(max(loc, 1e-6)) / dz)
          gb += dz;
#pragma unroll
          for (int d = 0; d < SD; ++d) gw[d] = fmaf(dz, pl[d], gw[d]);
        }
      }
      gb = msr3d_attn::row16_sum(gb);
#pragma unroll
      for (int d = 0; d < SD; ++d) gw[d] = msr3d_attn::row16_sum(gw[d]);
      if (i == 0) {
        float *o = xg + (size_t)(kh * 64 + row) * 8;
        o[0] = gb;
#pragma unroll
        for (int d = 0; d < SD; ++d) o[1 + d] = gw[d];
      }
    }
    __syncthreads();                       // d S complete; the pairwise slab is dead; the cond partials visible
    oq[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    ok[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    msr3d_attn::strip_mma<kCoreMma, 1, TM, true, false, true>(sp, LDP, sk + 16 * kh, LD32, row0, oq, lane);
    msr3d_attn::strip_mma<kCoreMma, 1, TM, false, false, true>(sp, LDP, sq + 16 * kh, LD32, row0, ok, lane);
  }
  SB_STAMP(6);
  // product 2's operand (FRAG planes, 4 slabs: [dq | dk | dv | dcond, 0]) on top of the dead pairwise slab; side outputs
  unsigned char *mid = reinterpret_cast<unsigned char *>(spl);
  {
    const int row = row0 + j;
    const bool okr = row < L;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const f32x4 &src = a == 0 ? oq[0] : a == 1 ? ok[0] : ov[0];
      const float sc = a < 2 ? kInvSqrtDh : 1.0f;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = okr ? src[r] * sc : 0.f;
      if (okr) st4(p.dqkvc + (size_t)(row_base + row) * ldq + a * KD + h * DH + 16 * kh + 4 * g, make_float4(v[0], v[1], v[2], v[3]));
      uint2 pl[3];
      sm_split4(v, pl);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *reinterpret_cast<uint2 *>(mid + frag_off4<4>(t, j, 32 * a + 16 * kh + 4 * g, k)) = pl[k];
    }
  }
  if (tid < TM) {                          // slab 3 of row `tid`: the six cond gradients (half 0 + half 1), zeros behind them
    const int row = tid;
    float dc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) dc[c] = (row < L && c < SD + 1) ? xg[row * 8 + c] + xg[(64 + row) * 8 + c] : 0.f;
    if (row < L) {
#pragma unroll
      for (int c = 0; c < SD + 1; ++c) p.dqkvc[(size_t)(row_base + row) * ldq + 3 * KD + h * (SD + 1) + c] = dc[c];
    }
    uint4 pl[3];
    sm_split8(dc, pl);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      unsigned char *d = mid + (((3 * 4 + (row >> 4)) * 3 + k) * 64 + (row & 15)) * 16;
      *reinterpret_cast<uint4 *>(d) = pl[k];
#pragma unroll
      for (int gg = 1; gg < 4; ++gg) *reinterpret_cast<uint4 *>(d + gg * 256) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  __syncthreads();
  f32x4 acc2[2][4];
  zero_acc3(acc2);
  const XFrag<4> xm{reinterpret_cast<const unsigned short *>(mid) + lane * 8};
  gemm_split3<true, 2, 4, 4, RING>(xm, 0, w2, acc2, ring2);
  SB_STAMP(7);
  store_partials<2>(acc2, slab, row_base, L, 32 * wave, lane);
  SB_STAMP(8);
}

template <int KIND, int NW = 4>
int launch_block(const SB &p, int slices, hipStream_t s) {
  constexpr int lds = lds_bytes<KIND>();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&scene_block_kernel<KIND, NW>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != hipSuccess) return (int)attr;
  scene_block_kernel<KIND, NW><<<dim3(slices, p.B), 64 * NW, lds, s>>>(p);
  return (int)hipGetLastError();
}

// =====================================================================================================
// msr3d_split_pack: one wave per (slab, tile) piece, lane (j, g) -> operand row 16 tile + j,
// k = 32 slab + 8 g .. + 7: reads eight fp32 (contiguous for transposed = 0, one row apart otherwise),
// writes the lane's 16 bytes of each of the three planes (coalesced 1 KB per wave and plane).
// =====================================================================================================
__device__ __forceinline__ int map_index(const msr3d_pack_job_t *jb, int nseg, int i) {
  int r = -1;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d0 = jb->seg_dst[s], len = jb->seg_len[s], s0 = jb->seg_src[s];
    if (s < nseg && i >= d0 && i < d0 + len) r = s0 + (i - d0);
  }
  return r;
}

constexpr int kPackPer = 1;        // pieces per wave (measured: 4, all 32 loads issued before the first split: 27.2 us against 24.3)

// workgroups past `pack_wgs` (msr3d_split_pack_begin): the step's zero fill + dropout seed bump (= msr3d_step_begin)
__global__ __launch_bounds__(256) void split_pack_kernel(int njobs, const msr3d_pack_job_t *__restrict__ jobs,
                                                         const int *__restrict__ prefix, int total, int pack_wgs,
                                                         float4 *__restrict__ z, long long n4, unsigned long long *seed) {
  if ((int)blockIdx.x >= pack_wgs) {
    const int zb = blockIdx.x - pack_wgs, nz = gridDim.x - pack_wgs;
    if (seed && zb == 0 && threadIdx.x == 0) *seed = *seed * 6364136223846793005ull + 1442695040888963407ull;
    for (long long t = zb * 256ll + threadIdx.x; t < n4; t += nz * 256ll) z[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  // A wave takes kPackPer consecutive pieces and issues all their loads before the first split.
  const int piece0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kPackPer;
  if (piece0 >= total) return;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  float v[kPackPer][8];
  unsigned short *dst[kPackPer];
#pragma unroll
  for (int u = 0; u < kPackPer; ++u) {
    const int piece = piece0 + u;
    dst[u] = nullptr;
    if (piece >= total) continue;
    // the job holding this piece = (number of prefix entries <= piece) - 1: one coalesced load of the table + a ballot
    // per 64 jobs.  (Round 4's binary search was six DEPENDENT round trips to L2 in front of every wave's real loads:
    // 27.9 -> 24.3 us a launch of ~22 k one-piece waves.)
    int cnt = 0;
    for (int base = 0; base < njobs; base += 64) {
      const int q = base + lane;
      const bool le = q < njobs && prefix[q] <= piece;
      cnt += __popcll(__ballot(le));
    }
    const int lo = __builtin_amdgcn_readfirstlane(cnt - 1);
    const msr3d_pack_job_t *jb = jobs + lo;
    const float *__restrict__ src = jb->src;
    const int ld = jb->ld, nseg = jb->nseg;
    const int local = piece - prefix[lo];
    const int nt = jb->rows >> 4;
    const int slab = local / nt, tile = local - slab * nt;
    const int n = 16 * tile + j, k0 = 32 * slab + 8 * g;
    if (!jb->transposed) {
      const int r = map_index(jb, nseg, n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = r >= 0 ? src[(size_t)r * ld + k0 + e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = map_index(jb, nseg, k0 + e);
        v[u][e] = r >= 0 ? src[(size_t)r * ld + n] : 0.f;
      }
    }
    dst[u] = jb->dst + (size_t)local * (3 * 512) + lane * 8;
  }
#pragma unroll
  for (int u = 0; u < kPackPer; ++u) {
    if (!dst[u]) continue;
    uint4 pl[3];
    sm_split8(v[u], pl);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(dst[u] + k * 512) = pl[k];
  }
}

// the attention forward block's form: -1 not chosen yet (environment at first use), 0 one workgroup per (scene, head),
// 1 two (msr3d_attn_fwd_form)
int g_attn_fwd_form = -1;
bool attn_fwd_split() {
  if (g_attn_fwd_form < 0) {
    const char *v = getenv("MSR3D_ATTN_FWD_SPLIT");
    g_attn_fwd_form = (v && v[0] == '0') ? 0 : 1;
  }
  return g_attn_fwd_form != 0;
}

}  // namespace

extern "C" {

int msr3d_attn_fwd_form(int form) {
  if (form == 0 || form == 1) g_attn_fwd_form = form;
  else if (form != -1) return MSR3D_EINVAL;
  return attn_fwd_split() ? 1 : 0;
}

int msr3d_split_pack(int njobs, const msr3d_pack_job_t *jobs, const int *piece_prefix, int total_pieces,
                     msr3d_stream_t stream) {
  if (njobs < 0 || total_pieces < 0) return MSR3D_EINVAL;
  if (njobs == 0 || total_pieces == 0) return 0;
  if (!jobs || !piece_prefix) return MSR3D_EINVAL;
  const int wgs = (total_pieces + 4 * kPackPer - 1) / (4 * kPackPer);
  split_pack_kernel<<<wgs, 256, 0, (hipStream_t)stream>>>(njobs, jobs, piece_prefix, total_pieces, wgs, nullptr, 0, nullptr);
  return (int)hipGetLastError();
}

int msr3d_split_pack_begin(int njobs, const msr3d_pack_job_t *jobs, const int *piece_prefix, int total_pieces,
                           float *zero_region, long long n_floats, unsigned long long *seed, msr3d_stream_t stream) {
  if (njobs < 0 || total_pieces < 0 || n_floats < 0 || (n_floats % 4)) return MSR3D_EINVAL;
  if (njobs > 0 && total_pieces > 0 && (!jobs || !piece_prefix)) return MSR3D_EINVAL;
  if (n_floats > 0 && (!zero_region || (reinterpret_cast<uintptr_t>(zero_region) & 15u))) return MSR3D_EINVAL;
  const int wgs = (njobs > 0) ? (total_pieces + 4 * kPackPer - 1) / (4 * kPackPer) : 0;
  const long long n4 = n_floats / 4;
  long long zw = (n4 + 255) / 256;
  zw = zw > 512 ? 512 : zw;
  if (zw < 1 && seed) zw = 1;
  if (wgs + zw == 0) return 0;
  split_pack_kernel<<<wgs + (int)zw, 256, 0, (hipStream_t)stream>>>(njobs, jobs, piece_prefix, total_pieces, wgs,
                                                                    reinterpret_cast<float4 *>(zero_region), n4, seed);
  return (int)hipGetLastError();
}

int msr3d_scene_block(const msr3d_scene_block_t *pp, msr3d_stream_t stream) {
  if (!pp) return MSR3D_EINVAL;
  const SB &p = *pp;
  if (p.B < 0 || p.L <= 0 || p.L > TM) return MSR3D_EINVAL;
  if (p.B == 0) return 0;
  if (!p.w1) return MSR3D_EINVAL;
  if (p.kind == MSR3D_BLK_LINEAR_KSPLIT ? !p.a0 : !p.xp) return MSR3D_EINVAL;
  const long long rows_all = p.rows_total > 0 ? p.rows_total : (long long)p.B * p.L;
  if (p.rows_total < 0 || (p.rows_total > 0 && (p.rows_total > p.B * p.L || p.rows_total <= (p.B - 1) * p.L))) return MSR3D_EINVAL;
  if (p.rows_total > 0 && (p.kind == MSR3D_BLK_ATTN_FWD || p.kind == MSR3D_BLK_ATTN_BWD)) return MSR3D_EINVAL;
  if (p.kind != MSR3D_BLK_LINEAR && (!p.part || p.part_stride < rows_all * KD)) return MSR3D_EINVAL;
  if (p.p_drop > 0.f && !p.seed) return MSR3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (p.kind) {
    case MSR3D_BLK_ATTN_FWD:
      if (p.H != 8 || !p.w2 || !p.qkvc || !p.ploc || !p.pad || !p.ctx || !p.bias1 || p.ldq % 4) return MSR3D_EINVAL;
      if (p.w1_bytes < 8u * 8u * 8u * kPieceBytes || p.w2_bytes < 8u * 16u * kPieceBytes) return MSR3D_EINVAL;
      {
        // two workgroups per (scene, head), keys split over the core's eight waves (round 5): the library's default;
        // MSR3D_ATTN_FWD_SPLIT=0 / msr3d_attn_fwd_form(0): one workgroup per (scene, head)
        if (attn_fwd_split()) {
          static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&scene_attn_fwd2_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kAttnFwd2Lds);
          if (attr != hipSuccess) return (int)attr;
          scene_attn_fwd2_kernel<<<dim3(16, p.B), 512, kAttnFwd2Lds, s>>>(p);
          return (int)hipGetLastError();
        }
      }
      {
        // eight waves per workgroup (two per SIMD) by default: -6 us a step in three interleaved same-box runs
        // (1.2421 -> 1.2362 ms); MSR3D_ATTN_FWD_WAVES=4 restores round 3's four
        static const bool eight = [] { const char *v = getenv("MSR3D_ATTN_FWD_WAVES"); return !(v && v[0] == '4'); }();
        return eight ? launch_block<MSR3D_BLK_ATTN_FWD, 8>(p, 8, s) : launch_block<MSR3D_BLK_ATTN_FWD>(p, 8, s);
      }
    case MSR3D_BLK_FFN_FWD:
    case MSR3D_BLK_FFN_BWD:
      if (p.ff <= 0 || p.ff % 128 || p.ff / 128 > 16 || !p.w2 || !p.pre || !p.h) return MSR3D_EINVAL;
      if (p.w1_bytes < (unsigned)(8 * (p.ff / 16)) * kPieceBytes || p.w2_bytes < (unsigned)((p.ff / 32) * 16) * kPieceBytes)
        return MSR3D_EINVAL;
      {
        // eight waves here too (round 5): -1.6 us a step over the six feed-forward blocks in three interleaved same-box
        // pairs (0.9497 / 0.9502 / 0.9479 -> 0.9492 / 0.9484 / 0.9453 ms) -- the blocks wait on their weight pieces and
        // planes, not on instruction issue; MSR3D_FFN_WAVES=4 restores four
        static const bool eight = [] { const char *v = getenv("MSR3D_FFN_WAVES"); return !(v && v[0] == '4'); }();
        if (eight)
          return p.kind == MSR3D_BLK_FFN_FWD ? launch_block<MSR3D_BLK_FFN_FWD, 8>(p, p.ff / 128, s)
                                             : launch_block<MSR3D_BLK_FFN_BWD, 8>(p, p.ff / 128, s);
      }
      return p.kind == MSR3D_BLK_FFN_FWD ? launch_block<MSR3D_BLK_FFN_FWD>(p, p.ff / 128, s)
                                         : launch_block<MSR3D_BLK_FFN_BWD>(p, p.ff / 128, s);
    case MSR3D_BLK_ATTN_BWD:
      if (p.H != 8 || !p.w2 || !p.qkvc || !p.dqkvc || !p.ploc || !p.pad || !p.probs || p.ldq % 4) return MSR3D_EINVAL;
      if (p.w1_bytes < 8u * 16u * kPieceBytes || p.w2_bytes < 8u * 4u * 16u * kPieceBytes) return MSR3D_EINVAL;
      {
        // 3 (default): eight waves, the core split over all of them; 2: scene_attn_bwd2_kernel; 0: scene_block_kernel<ATTN_BWD>
        static const int ver = [] { const char *v = getenv("MSR3D_ATTN_BWD"); return (v && v[0] >= '0' && v[0] <= '3') ? v[0] - '0' : 3; }();
        if (ver == 3) {
          static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&scene_attn_bwd3_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, kAttnBwd3Lds);
          if (attr != hipSuccess) return (int)attr;
          scene_attn_bwd3_kernel<<<dim3(8, p.B), 512, kAttnBwd3Lds, s>>>(p);
          return (int)hipGetLastError();
        }
        static const bool v2 = [] { const char *v = getenv("MSR3D_ATTN_BWD_V2"); return !(v && v[0] == '0'); }();
        if (v2 && ver != 0) {
          static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&scene_attn_bwd2_kernel),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             lds_bytes<MSR3D_BLK_ATTN_BWD>());
          if (attr != hipSuccess) return (int)attr;
          scene_attn_bwd2_kernel<<<dim3(8, p.B), 256, lds_bytes<MSR3D_BLK_ATTN_BWD>(), s>>>(p);
          return (int)hipGetLastError();
        }
      }
      return launch_block<MSR3D_BLK_ATTN_BWD>(p, 8, s);
    case MSR3D_BLK_LINEAR:
      if (p.N <= 0 || p.N % 256 || !p.C || p.ldc % 4 || p.w1_bytes < (unsigned)(8 * (p.N / 16)) * kPieceBytes) return MSR3D_EINVAL;
      return launch_block<MSR3D_BLK_LINEAR>(p, p.N / 256, s);
    case MSR3D_BLK_LINEAR_KSPLIT:
      if (p.lda0 <= 0 || p.lda0 % 256 || p.lda0 / 256 > 20) return MSR3D_EINVAL;
      if (p.w1_bytes < (unsigned)((p.lda0 / 32) * 16) * kPieceBytes) return MSR3D_EINVAL;
      return launch_block<MSR3D_BLK_LINEAR_KSPLIT>(p, p.lda0 / 256, s);
    default:
      return MSR3D_EINVAL;
  }
}

#if MSR3D_TRAIN_PLANES == 1
// libmsr3d_hip_bf16.so (this file + wgrad_split.hip, MSR3D_TRAIN_PLANES = 1) carries the version of the header it was
// compiled against: msr3d_amd/_lib.py::load_bf16 refuses a stale build (the main library exports this from pn2_ops.hip)
int msr3d_abi_version(void) { return MSR3D_ABI_VERSION; }
#endif

}  // extern "C"
