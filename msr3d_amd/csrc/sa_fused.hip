// sa_fused.hip -- fused PointNet++ set-abstraction levels for gfx950 (MI355X).
//
// What the reference does per level with ~25 launches
// (/root/reference/modules/third_party/pointnet2/pointnet2_modules.py:34-75:
//  FPS -> gather -> ball_query -> group(xyz) -> subtract -> group(feat) -> cat ->
//  3 x [conv1x1, BN, ReLU] -> max_pool -> squeeze), each intermediate a round trip
// through HBM, is here ONE launch per level plus one index launch:
//
//   msr3d_sa_fps2   FPS of level 1 and of level 2 (on level 1's winners), one wave per cloud
//   msr3d_sa_level  ball query (wave ballot; level 1: its own launch over the 1024-point clouds,
//                   level 2: inside the block) -> neighbourhood gather + recentre straight
//                   into LDS -> three GEMM layers on f32-input MFMA (16x16x4) with the
//                   BN(eval) affine + ReLU applied on the accumulators, activations kept
//                   in LDS -> max over the neighbourhood taken on the accumulators ->
//                   only the pooled (centres x C_out) tile is written.
//
// Arithmetic is fp32 end to end (the reference runs the backbone in fp32): products and
// sums are exact-f32 fma chains on the matrix cores, the k-order of a dot product differs
// from MIOpen's, hence a tolerance (not bit-exactness) on the features; the INDEX part
// (FPS, ball query) is bit-exact and shares its code with pn2_ops.hip.
//
// Layout choices: features between levels are POINT-major (b, points, C) so a
// neighbour's feature row is one contiguous 512 B / 1 KB read; weights are pre-packed
// by the host as [N][KP] rows (K permuted "features first, xyz last", zero-padded to a
// multiple of 16) followed by scale[N], shift[N].
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "pn2_device.h"

namespace {

using namespace msr3d;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Tile choices, each the winner of a measured sweep (the losing variants -- register
// pipeline pinned with sched_barrier, phase-skewed co-resident blocks, two tile groups per block,
// forced single residency -- lived here as build switches in round 1 and are now only in the table).
constexpr int kSa1Cpb = 2;      // centres per block, level 1: 64-row tile, 37 KB LDS, 4 blocks per CU
constexpr int kSa2Cpb = 2;      // centres per block, level 2: 64-row tile, 75 KB LDS, 2 blocks per CU
constexpr int kSa1Wm = 2;       // level 1 waves 2 x 2
constexpr int kSa2Wm = 1;       // level 2 waves 1 x 4: every wave owns all 64 rows and a quarter of the
                                // columns, so each weight fragment is fetched once per block (621 -> 537 us)

// Weights arrive in MFMA-fragment order (include/msr3d_hip.h, msr3d_sa_level): the 16 columns x 16 k
// of (slab, column tile) are ONE contiguous 1 KB block whose lane-th 16 bytes are what lane `lane`
// feeds the matrix pipe, so a wave's B-fragment load is a single fully coalesced 1 KB read (8 cache
// lines).  Row-major weights made the same load touch 16 rows x 64 B, i.e. 64 tag look-ups of 16 B per
// instruction: measured on the layout change alone, sa1 302 -> 280 us and sa3 190 -> 170 us (sa2
// unchanged until its waves were re-laid 1 x 4, which the contiguous blocks made worthwhile).
constexpr int kFrag = 256;   // floats per (slab, column tile) block: 64 lanes x 4
constexpr int kLdsPad = 8;   // row stride = K + 8 floats (K % 16 == 0): stride/4 == 2 (mod 4)
                             // makes every 16-lane group of a ds_read_b128 fragment read hit
                             // 16 distinct 16-B slots of the 256-B bank row

struct Layer {
  const float *w;       // [KP/16][N/16][64][4], fragment order
  const float *scale;   // [N]
  const float *shift;   // [N]
};

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------
// acc[RM][RN] (16x16 tiles) += X[rows][k] * W[cols][k]^T over k in [0, KP).
// X: LDS, row-major, leading dim ldx.  W: global (L2-resident), fragment-ordered
// [KP/16][NT][64][4] with NT = N/16 column tiles; `wg` points at this wave's first column tile.
// K is consumed in slabs of 16: lane (i = lane&15, g = lane>>4) holds k = 4g..4g+3 of
// row i, ONE 16-byte read for A (LDS) and for B (global); MFMA step s of the slab
// multiplies the k = 4g+s elements, i.e. a fixed permutation of k inside the slab --
// legal because a dot product does not care, and it turns 4+4 scalar fragment loads into
// 1+1 vector loads.
// ---------------------------------------------------------------------------------
// first B slab of a layer, issued early (before the previous layer's epilogue / the loader's
// barrier) so its L2 latency is off the critical path
template <int RN>
__device__ __forceinline__ void load_b_first(const float *__restrict__ wg, int lane,
                                             float4 (&b)[RN]) {
  const float *wp = wg + lane * 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) b[rn] = *reinterpret_cast<const float4 *>(wp + rn * kFrag);
}

// per-lane BN affine of this wave's columns (col = 16 rn + (lane & 15)), fetched with the B prefetch
template <int RN>
__device__ __forceinline__ void load_affine(const float *__restrict__ scale,
                                            const float *__restrict__ shift, int lane,
                                            float (&sc)[RN], float (&sh)[RN]) {
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    sc[rn] = scale[rn * 16 + (lane & 15)];
    sh[rn] = shift[rn * 16 + (lane & 15)];
  }
}

template <int RM, int RN, int KP, int NT>
__device__ __forceinline__ void gemm_lds_global(const float *xs, int ldx,
                                                const float *__restrict__ wg,
                                                f32x4 (&acc)[RM][RN], int lane,
                                                const float4 (&bfirst)[RN]) {
  const int i = lane & 15, g = lane >> 4;
  const float *xp = xs + i * ldx + 4 * g;
  const float *wp = wg + lane * 4;
  float4 bcur[RN];
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) bcur[rn] = bfirst[rn];
#pragma unroll 2
  for (int k0 = 0; k0 < KP; k0 += 16) {
    float4 bnext[RN];
    const int kn = (k0 + 16 < KP) ? k0 + 16 : k0;   // last slab: harmless re-read
#pragma unroll
    for (int rn = 0; rn < RN; ++rn)
      bnext[rn] = *reinterpret_cast<const float4 *>(wp + (size_t)(kn / 16) * (NT * kFrag) + rn * kFrag);
    float4 a[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
      a[rm] = *reinterpret_cast<const float4 *>(xp + rm * 16 * ldx + k0);
    // step-major order: consecutive MFMAs hit DIFFERENT accumulators (RM*RN of them), so the
    // 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 never gates its 32-cycle issue rate
#define MSR3D_STEP(c)                                                                           \
    _Pragma("unroll") for (int rm = 0; rm < RM; ++rm)                                           \
    _Pragma("unroll") for (int rn = 0; rn < RN; ++rn)                                           \
        acc[rm][rn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rm].c, bcur[rn].c, acc[rm][rn], 0, 0, 0);
    MSR3D_STEP(x)
    MSR3D_STEP(y)
    MSR3D_STEP(z)
    MSR3D_STEP(w)
#undef MSR3D_STEP
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) bcur[rn] = bnext[rn];
  }
}

template <int RM, int RN>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RM][RN]) {
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) acc[rm][rn] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// y = relu(acc * scale[col] + shift[col]) -> LDS tile.  C/D map of the 16x16 MFMA:
// col = lane & 15, row = (lane >> 4) * 4 + reg.
template <int RM, int RN>
__device__ __forceinline__ void store_bn_relu_lds(const f32x4 (&acc)[RM][RN],
                                                  const float (&scv)[RN], const float (&shv)[RN],
                                                  float *ys, int ldy, int lane) {
  const int c = lane & 15, r4 = (lane >> 4) * 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float sc = scv[rn], sh = shv[rn];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ys[(rm * 16 + r4 + r) * ldy + rn * 16 + c] =
            fmaxf(__builtin_fmaf(acc[rm][rn][r], sc, sh), 0.0f);
  }
}

// max over groups of GT*16 consecutive rows of relu(acc*scale+shift) -> global (group, col).
// Starting the running max at 0 IS the ReLU (max and ReLU commute).
template <int RM, int RN, int GT>
__device__ __forceinline__ void store_bn_relu_groupmax(const f32x4 (&acc)[RM][RN],
                                                       const float (&scv)[RN],
                                                       const float (&shv)[RN],
                                                       float *__restrict__ out, int ldo,
                                                       int groups_valid, int lane) {
  const int c = lane & 15;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float sc = scv[rn], sh = shv[rn];
#pragma unroll
    for (int gq = 0; gq < RM / GT; ++gq) {
      float m = 0.0f;
#pragma unroll
      for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          m = fmaxf(m, __builtin_fmaf(acc[gq * GT + t][rn][r], sc, sh));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (lane < 16 && gq < groups_valid) out[(size_t)gq * ldo + rn * 16 + c] = m;
    }
  }
}

// ---------------------------------------------------------------------------------
// Three chained layers on a TM-row tile held in LDS.  bufA holds the input tile
// [TM][K0P+8] and is re-used for layer 2's output [TM][N2+8]; bufB holds layer 1's
// output [TM][N1+8].  4 waves as WM x WN; every wave owns TM/WM rows x N/WN columns.
// ---------------------------------------------------------------------------------
template <int TM, int K0P, int N1, int N2, int N3, int G, int WM, int WN>
struct Chain {
  static constexpr int RM = TM / (16 * WM);
  static constexpr int LDA = cmax(K0P, N2) + kLdsPad;
  static constexpr int LDB = N1 + kLdsPad;
  static constexpr int GT = G / 16;
  static_assert(WM * WN == 4 && RM * 16 * WM == TM, "tile / wave grid mismatch");
  static_assert(N1 % (16 * WN) == 0 && N2 % (16 * WN) == 0 && N3 % (16 * WN) == 0, "N split");
  static_assert(K0P % 16 == 0 && N1 % 16 == 0 && N2 % 16 == 0, "K must be a multiple of 16");
  static_assert(RM % GT == 0, "a pooling group must live inside one wave");
  static constexpr int LDS_FLOATS = TM * LDA + TM * LDB;

  static constexpr int RN1 = N1 / (16 * WN), RN2 = N2 / (16 * WN), RN3 = N3 / (16 * WN);

  struct Pre1 {            // layer-1 operands fetched before the loader phase
    float4 b[RN1];
    float sc[RN1], sh[RN1];
  };
  __device__ static void preload(const Layer &l1, Pre1 &p, int tid) {
    const int lane = tid & 63, wn = (tid >> 6) % WN;
    const int col0 = wn * RN1 * 16;
    load_b_first<RN1>(l1.w + (col0 / 16) * kFrag, lane, p.b);
    load_affine<RN1>(l1.scale + col0, l1.shift + col0, lane, p.sc, p.sh);
  }

  // out: first pooled row of this block; `groups_valid_block`: pooled rows of this block that exist
  // `tid`: thread index inside the 256-thread block.
  __device__ static void run(float *bufA, float *bufB, const Pre1 &p1, const Layer &l1,
                             const Layer &l2, const Layer &l3, float *__restrict__ out,
                             int groups_valid_block, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int row0 = wm * RM * 16;
    float4 b2[RN2], b3[RN3];
    float sc2[RN2], sh2[RN2], sc3[RN3], sh3[RN3];
    {
      const int col0 = wn * RN1 * 16;
      f32x4 acc[RM][RN1];
      zero_acc(acc);
      gemm_lds_global<RM, RN1, K0P, N1 / 16>(bufA + row0 * LDA, LDA, l1.w + (col0 / 16) * kFrag, acc, lane, p1.b);
      // next layer's first operands fly while this layer's epilogue and barrier run
      load_b_first<RN2>(l2.w + (wn * RN2) * kFrag, lane, b2);
      load_affine<RN2>(l2.scale + wn * RN2 * 16, l2.shift + wn * RN2 * 16, lane, sc2, sh2);
      store_bn_relu_lds<RM, RN1>(acc, p1.sc, p1.sh, bufB + row0 * LDB + col0, LDB, lane);
    }
    __syncthreads();
    {
      const int col0 = wn * RN2 * 16;
      f32x4 acc[RM][RN2];
      zero_acc(acc);
      gemm_lds_global<RM, RN2, N1, N2 / 16>(bufB + row0 * LDB, LDB, l2.w + (col0 / 16) * kFrag, acc, lane, b2);
      load_b_first<RN3>(l3.w + (wn * RN3) * kFrag, lane, b3);
      load_affine<RN3>(l3.scale + wn * RN3 * 16, l3.shift + wn * RN3 * 16, lane, sc3, sh3);
      store_bn_relu_lds<RM, RN2>(acc, sc2, sh2, bufA + row0 * LDA + col0, LDA, lane);
    }
    __syncthreads();
    {
      const int col0 = wn * RN3 * 16;
      f32x4 acc[RM][RN3];
      zero_acc(acc);
      gemm_lds_global<RM, RN3, N2, N3 / 16>(bufA + row0 * LDA, LDA, l3.w + (col0 / 16) * kFrag, acc, lane, b3);
      const int g0 = wm * (RM / GT);            // first pooled row owned by this wave
      int gv = groups_valid_block - g0;
      store_bn_relu_groupmax<RM, RN3, GT>(acc, sc3, sh3, out + (size_t)g0 * N3 + col0, N3, gv, lane);
    }
  }
};

// ---------------------------------------------------------------------------------
// Ball query of ONE centre by ONE wave over a cloud staged in LDS (packed xyz), result in
// an LDS row of `nsample` ints.  Same semantics as ball_query_kernel in pn2_ops.hip
// (ball_query_gpu.cu:9-44): index order, strict '<', first-hit fill, zeros when empty.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void wave_ball_query(const float *sx, int n, float cx, float cy, float cz,
                                                float radius2, int nsample, int *row, int lane) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < nsample; base += kWave) {
    const int k = base + lane;
    bool hit = false;
    if (k < n) {
      const float d2 = sq3(cx - sx[k * 3 + 0], cy - sx[k * 3 + 1], cz - sx[k * 3 + 2]);
      hit = d2 < radius2;
    }
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      if (cnt == 0) first = base + __ffsll((long long)mask) - 1;
      const int slot = cnt + __popcll(mask & lt);
      if (hit && slot < nsample) row[slot] = k;
      cnt += __popcll(mask);
    }
  }
  const int filled = cnt < nsample ? cnt : nsample;
  const int fill = cnt > 0 ? first : 0;
  for (int l = filled + lane; l < nsample; l += kWave) row[l] = fill;
}

// =================================================================================
// Level 1: points (b, n, 6) = [xyz, rgb] as the dataset stores them; centres (b, m, 3).
// Block = CPB centres x 32 neighbours (default 2 -> 64 rows).  MLP 6 -> 64 -> 64 -> 128.  The ball
// query runs as its own launch and hands its indices over in `ball_idx`.  out: (b, m, 128) point-major.
// =================================================================================
constexpr int kNS = 32;   // neighbours per centre in both query levels (configs/msr3d.yaml:199)
// CPB centres per block: 2 -> 64-row tile, 37 KB of LDS and (capped) <= 128 registers: four blocks per
// CU, so one block's gather phase hides under the others' MFMA phases (128- / 256-row tiles: 297 / 358 us
// against 279; waves 1 x 4: 300 us).
template <int CPB> struct Sa1 {
  using C = Chain<CPB * kNS, 16, 64, 64, 128, kNS, kSa1Wm, 4 / kSa1Wm>;
};

// 128-register cap: the 37 KB of LDS allow four blocks per CU, 136 registers did not
template <int CPB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void sa1_kernel(int n, int m, const float *__restrict__ pts,
                                                  const float *__restrict__ new_xyz,
                                                  const int *__restrict__ ball_idx, Layer l1,
                                                  Layer l2, Layer l3, float *__restrict__ out,
                                                  const unsigned char *__restrict__ valid) {
  if (valid && !valid[blockIdx.y]) return;          // padding object (see msr3d_sa_level)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Chain1 = typename Sa1<CPB>::C;
  constexpr int TM = CPB * kNS;
  float *bufA = reinterpret_cast<float *>(smem);
  float *bufB = bufA + TM * Chain1::LDA;

  const int obj = blockIdx.y, c0 = blockIdx.x * CPB;
  const int tid = threadIdx.x;
  typename Chain1::Pre1 pre;
  Chain1::preload(l1, pre, tid);     // layer-1 weights/affine in flight during the gather
  const float *P = pts + (size_t)obj * n * 6;
  // gather: row = (centre w, sample k); cols [x-cx, y-cy, z-cz, r, g, b, 0 x10].  The ball
  // indices come from the wave-ballot query launched just before (msr3d_sa_level does both);
  // each point row is 24 B = three 8-byte loads.
  {   // indices first, then ALL point loads, then the LDS stores: two L2 round trips per block in all
    constexpr int IT = TM * 8 / 256;
    int pidx[IT];
    float2 v[IT];
    float cx[IT], cy[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256, row = e >> 3, c2 = e & 7, cj = c0 + (row >> 5);
      pidx[it] = -1;
      cx[it] = cy[it] = 0.f;
      if (c2 < 3 && cj < m) {
        pidx[it] = ball_idx[((size_t)obj * m + cj) * kNS + (row & 31)];
        const float *c = new_xyz + ((size_t)obj * m + cj) * 3;
        if (c2 == 0) { cx[it] = c[0]; cy[it] = c[1]; }
        else if (c2 == 1) { cx[it] = c[2]; }
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int c2 = (tid + it * 256) & 7;
      v[it] = make_float2(0.f, 0.f);
      if (pidx[it] >= 0) v[it] = *reinterpret_cast<const float2 *>(P + (size_t)pidx[it] * 6 + c2 * 2);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256, row = e >> 3, c2 = e & 7;
      *reinterpret_cast<float2 *>(bufA + row * Chain1::LDA + c2 * 2) = make_float2(v[it].x - cx[it], v[it].y - cy[it]);
    }
  }
  __syncthreads();
  const int groups = (m - c0) < CPB ? (m - c0) : CPB;
  Chain1::run(bufA, bufB, pre, l1, l2, l3, out + ((size_t)obj * m + c0) * 128, groups, tid);
}

// =================================================================================
// Level 2: xyz (b, n<=64, 3) = level-1 centres, feat (b, n, 128) point-major; centres
// (b, m, 3).  Block = 4 centres x 32 neighbours.  MLP 131 -> 128 -> 128 -> 256 with the
// K order [feat(128), dxyz(3), 0 x13].  out: (b, m, 256).
// =================================================================================
template <int CPB> struct Sa2 {
  using C = Chain<CPB * kNS, 144, 128, 128, 256, kNS, kSa2Wm, 4 / kSa2Wm>;
};

template <int CPB>
__global__ __launch_bounds__(256) void sa2_kernel(int n, int m, float radius2,
                                                       const float *__restrict__ xyz,
                                                       const float *__restrict__ feat,
                                                       const float *__restrict__ new_xyz, Layer l1,
                                                       Layer l2, Layer l3, float *__restrict__ out,
                                                       int *__restrict__ dbg_idx,
                                                       const unsigned char *__restrict__ valid) {
  if (valid && !valid[blockIdx.y]) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Chain2 = typename Sa2<CPB>::C;
  constexpr int TM = CPB * kNS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *bufA = reinterpret_cast<float *>(smem);
  float *bufB = bufA + TM * Chain2::LDA;
  int *nbr = reinterpret_cast<int *>(bufB + TM * Chain2::LDB);   // [CPB][32]
  float *ctr = reinterpret_cast<float *>(nbr + 4 * kNS);          // [4][4]
  float *sx = ctr + 16;                                            // [n][3], n <= 64

  const int obj = blockIdx.y, c0 = blockIdx.x * CPB;
  typename Chain2::Pre1 pre;
  Chain2::preload(l1, pre, tid);     // layer-1 weights/affine in flight during the loader phase
  if (tid < n * 3) sx[tid] = xyz[(size_t)obj * n * 3 + tid];
  if (tid >= 192 && tid < 192 + 3 * CPB) {
    const int t = tid - 192, w = t / 3, c = t - w * 3;
    ctr[w * 4 + c] = (c0 + w < m) ? new_xyz[((size_t)obj * m + c0 + w) * 3 + c] : 0.f;
  }
  __syncthreads();
  if (wave < CPB) {
    if (c0 + wave < m)
      wave_ball_query(sx, n, ctr[wave * 4 + 0], ctr[wave * 4 + 1], ctr[wave * 4 + 2], radius2, kNS,
                      nbr + wave * kNS, lane);
    else if (lane < kNS)
      nbr[wave * kNS + lane] = 0;
  }
  __syncthreads();
  if (dbg_idx && tid < CPB * kNS && c0 + tid / kNS < m)
    dbg_idx[((size_t)obj * m + c0) * kNS + tid] = nbr[tid];
  const float *F = feat + (size_t)obj * n * 128;
  {   // 32 float4 per row; indices first, then ALL loads, then the LDS stores: one L2 round trip
    constexpr int IT = TM * 32 / 256;
    int pidx[IT];
    float4 val[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) pidx[it] = nbr[(tid + it * 256) >> 5];
#pragma unroll
    for (int it = 0; it < IT; ++it)
      val[it] = *reinterpret_cast<const float4 *>(F + (size_t)pidx[it] * 128 + ((tid + it * 256) & 31) * 4);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      *reinterpret_cast<float4 *>(bufA + (e >> 5) * Chain2::LDA + (e & 31) * 4) = val[it];
    }
  }
  if (tid < TM) {
    const int row = tid, p = nbr[row], w = row >> 5;
    float *d = bufA + row * Chain2::LDA + 128;
    d[0] = sx[p * 3 + 0] - ctr[w * 4 + 0];
    d[1] = sx[p * 3 + 1] - ctr[w * 4 + 1];
    d[2] = sx[p * 3 + 2] - ctr[w * 4 + 2];
#pragma unroll
    for (int c = 3; c < 16; ++c) d[c] = 0.f;
  }
  __syncthreads();
  int groups = m - c0;
  groups = groups < 0 ? 0 : (groups < CPB ? groups : CPB);
  Chain2::run(bufA, bufB, pre, l1, l2, l3, out + ((size_t)obj * m + c0) * 256, groups, tid);
}

// =================================================================================
// Level 3 (GroupAll): every object's 16 points form one group.  rows = (object, point),
// block = 2 objects = 32 rows.  MLP 259 -> 256 -> 512 -> 768 with the K order
// [feat(256), xyz(3), 0 x13].  out: (b, 768).  A third LDS buffer is needed because
// N2 = 512 does not fit the input tile's buffer.
// =================================================================================
constexpr int kN3Pts = 16;
struct Chain3Cfg {
  static constexpr int TM = 32, K0P = 272, N1 = 256, N2 = 512, N3 = 768;
  static constexpr int LDX = K0P + kLdsPad, LD1 = N1 + kLdsPad, LD2 = N2 + kLdsPad;
  static constexpr int LDS_FLOATS = TM * (LDX + LD1 + LD2);
};

__global__ __launch_bounds__(256) void sa3_kernel(int b, const float *__restrict__ xyz,
                                                  const float *__restrict__ feat, Layer l1,
                                                  Layer l2, Layer l3, float *__restrict__ out,
                                                  const unsigned char *__restrict__ valid) {
  if (valid) {      // both objects of the block are padding: skip (one valid: the other rides along)
    const int o0 = blockIdx.x * 2;
    if (!valid[o0] && !(o0 + 1 < b && valid[o0 + 1])) return;
  }
  using C = Chain3Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *bufX = reinterpret_cast<float *>(smem);
  float *buf1 = bufX + C::TM * C::LDX;
  float *buf2 = buf1 + C::TM * C::LD1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int obj0 = blockIdx.x * 2;
  float4 b1[C::N1 / 64];
  float sc1[C::N1 / 64], sh1[C::N1 / 64];
  load_b_first<C::N1 / 64>(l1.w + (wave * (C::N1 / 64)) * kFrag, lane, b1);
  load_affine<C::N1 / 64>(l1.scale + wave * (C::N1 / 64) * 16, l1.shift + wave * (C::N1 / 64) * 16,
                          lane, sc1, sh1);
  for (int e = tid; e < C::TM * 64; e += 256) {        // 64 float4 of features per row
    const int row = e >> 6, c4 = e & 63;
    const int obj = obj0 + (row >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (obj < b)
      v = *reinterpret_cast<const float4 *>(feat + ((size_t)obj * kN3Pts + (row & 15)) * 256 + c4 * 4);
    *reinterpret_cast<float4 *>(bufX + row * C::LDX + c4 * 4) = v;
  }
  if (tid < C::TM) {
    const int row = tid, obj = obj0 + (row >> 4);
    float *d = bufX + row * C::LDX + 256;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      d[c] = (c < 3 && obj < b) ? xyz[((size_t)obj * kN3Pts + (row & 15)) * 3 + c] : 0.f;
  }
  __syncthreads();
  constexpr int RM = 2;    // 32 rows, all waves; waves split N four ways
  constexpr int RN1 = C::N1 / 64, RN2 = C::N2 / 64, RN3 = C::N3 / 64;
  float4 b2[RN2], b3[RN3];
  float sc2[RN2], sh2[RN2], sc3[RN3], sh3[RN3];
  {
    const int col0 = wave * RN1 * 16;
    f32x4 acc[RM][RN1];
    zero_acc(acc);
    gemm_lds_global<RM, RN1, C::K0P, C::N1 / 16>(bufX, C::LDX, l1.w + (col0 / 16) * kFrag, acc, lane, b1);
    load_b_first<RN2>(l2.w + (wave * RN2) * kFrag, lane, b2);
    load_affine<RN2>(l2.scale + wave * RN2 * 16, l2.shift + wave * RN2 * 16, lane, sc2, sh2);
    store_bn_relu_lds<RM, RN1>(acc, sc1, sh1, buf1 + col0, C::LD1, lane);
  }
  __syncthreads();
  {
    const int col0 = wave * RN2 * 16;
    f32x4 acc[RM][RN2];
    zero_acc(acc);
    gemm_lds_global<RM, RN2, C::N1, C::N2 / 16>(buf1, C::LD1, l2.w + (col0 / 16) * kFrag, acc, lane, b2);
    load_b_first<RN3>(l3.w + (wave * RN3) * kFrag, lane, b3);
    load_affine<RN3>(l3.scale + wave * RN3 * 16, l3.shift + wave * RN3 * 16, lane, sc3, sh3);
    store_bn_relu_lds<RM, RN2>(acc, sc2, sh2, buf2 + col0, C::LD2, lane);
  }
  __syncthreads();
  {
    const int col0 = wave * RN3 * 16;
    f32x4 acc[RM][RN3];
    zero_acc(acc);
    gemm_lds_global<RM, RN3, C::N2, C::N3 / 16>(buf2, C::LD2, l3.w + (col0 / 16) * kFrag, acc, lane, b3);
    const int gv = (b - obj0) < 2 ? (b - obj0) : 2;
    store_bn_relu_groupmax<RM, RN3, 1>(acc, sc3, sh3, out + (size_t)obj0 * C::N3 + col0, C::N3, gv,
                                       lane);
  }
}

inline Layer make_layer(const float *packed, int n, int kp) {
  Layer l;
  l.w = packed;
  l.scale = packed + (size_t)n * kp;
  l.shift = l.scale + n;
  return l;
}

// Raise a kernel's dynamic-LDS limit above the 64 KB default, once per (kernel, size).
template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  static size_t granted = 0;        // one static per instantiation = per kernel
  if (bytes <= granted) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) granted = bytes;
  return e;
}

}  // namespace

extern "C" {

int msr3d_sa_fps2_flags(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                        float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid,
                        unsigned char *constant_out, msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m1 <= 0 || m2 < 0 || point_stride < 3) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!pts) return MSR3D_EINVAL;
  const hipError_t e = dispatch_fps(b, n, point_stride, m1, pts, idx1, new_xyz1, m2, idx2,
                                    new_xyz2, (hipStream_t)stream, valid, constant_out);
  return e == hipErrorInvalidValue ? MSR3D_EINVAL : (int)e;
}

int msr3d_sa_fps2(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                  float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid,
                  msr3d_stream_t stream) {
  return msr3d_sa_fps2_flags(b, n, point_stride, m1, m2, pts, idx1, new_xyz1, idx2, new_xyz2, valid, nullptr, stream);
}

int msr3d_sa_fps2_query_flags(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                              float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid, float radius1,
                              int nsample1, int *ball_idx1, unsigned char *constant_out, msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m1 <= 0 || m2 < 0 || point_stride < 3 || nsample1 <= 0 || !(radius1 > 0.f)) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!pts || !ball_idx1) return MSR3D_EINVAL;
  const hipError_t e = launch_fps_query(b, n, point_stride, m1, pts, idx1, new_xyz1, m2, idx2, new_xyz2,
                                        radius1 * radius1, nsample1, ball_idx1, (hipStream_t)stream, valid, constant_out);
  return e == hipErrorInvalidValue ? MSR3D_EINVAL : (int)e;
}

int msr3d_sa_fps2_query(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                        float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid, float radius1,
                        int nsample1, int *ball_idx1, msr3d_stream_t stream) {
  return msr3d_sa_fps2_query_flags(b, n, point_stride, m1, m2, pts, idx1, new_xyz1, idx2, new_xyz2, valid, radius1,
                                   nsample1, ball_idx1, nullptr, stream);
}

int msr3d_sa_level(int level, int b, int n, int m, float radius, const float *pts,
                   const float *feat, const float *new_xyz, const int *dims,
                   const float *params1, const float *params2, const float *params3, float *out,
                   int *dbg_ball_idx, const unsigned char *valid, msr3d_stream_t stream) {
  if (b < 0 || !dims) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!params1 || !params2 || !params3 || !out) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float r2 = radius * radius;   // f32 product, as ball_query_gpu.cu:22
  hipError_t e;
  if (level == 1) {
    // pts (b,n,6); dims = {6, 64, 64, 128}
    if (!(dims[0] == 6 && dims[1] == 64 && dims[2] == 64 && dims[3] == 128)) return MSR3D_EINVAL;
    if (!pts || !new_xyz || n <= 0 || m <= 0) return MSR3D_EINVAL;
    // two launches: the wave-ballot ball query over the in-place (stride 6) point rows into
    // ball_idx (required here: it is this level's workspace), then gather + MLP + max.  Keeping
    // the query out of the MLP kernel leaves it 37 KB of LDS -> 4 blocks per CU.
    if (!dbg_ball_idx) return MSR3D_EINVAL;
    // (radius <= 0: ball_idx already holds the neighbour lists -- msr3d_sa_fps2_query wrote them beside the FPS)
    if (radius > 0.f && (e = launch_ball_query(b, n, 6, m, r2, kNS, new_xyz, pts, dbg_ball_idx, st, valid)) != hipSuccess)
      return (int)e;
    constexpr int CPB = kSa1Cpb;
    const size_t lds = sizeof(float) * Sa1<CPB>::C::LDS_FLOATS;
    if ((e = allow_lds(sa1_kernel<CPB>, lds)) != hipSuccess) return (int)e;
    dim3 grid((m + CPB - 1) / CPB, b);
    sa1_kernel<CPB><<<grid, 256, lds, st>>>(n, m, pts, new_xyz, dbg_ball_idx,
                                            make_layer(params1, 64, 16), make_layer(params2, 64, 64),
                                            make_layer(params3, 128, 64), out, valid);
  } else if (level == 2) {
    // pts = xyz (b,n,3), feat (b,n,128); dims = {131, 128, 128, 256}
    if (!(dims[0] == 131 && dims[1] == 128 && dims[2] == 128 && dims[3] == 256)) return MSR3D_EINVAL;
    if (!pts || !feat || !new_xyz || n <= 0 || n > 64 || m <= 0) return MSR3D_EINVAL;
    constexpr int CPB = kSa2Cpb;
    const size_t lds = sizeof(float) * (Sa2<CPB>::C::LDS_FLOATS + 4 * kNS + 16 + 64 * 3);
    if ((e = allow_lds(sa2_kernel<CPB>, lds)) != hipSuccess) return (int)e;
    dim3 grid((m + CPB - 1) / CPB, b);
    sa2_kernel<CPB><<<grid, 256, lds, st>>>(n, m, r2, pts, feat, new_xyz,
                                                     make_layer(params1, 128, 144),
                                                     make_layer(params2, 128, 128),
                                                     make_layer(params3, 256, 128), out, dbg_ball_idx,
                                                     valid);
  } else if (level == 3) {
    // group-all over n = 16 points: pts = xyz (b,16,3), feat (b,16,256); dims = {259,256,512,768}
    if (!(dims[0] == 259 && dims[1] == 256 && dims[2] == 512 && dims[3] == 768)) return MSR3D_EINVAL;
    if (!pts || !feat || n != kN3Pts) return MSR3D_EINVAL;
    const size_t lds = sizeof(float) * Chain3Cfg::LDS_FLOATS;
    if ((e = allow_lds(sa3_kernel, lds)) != hipSuccess) return (int)e;
    sa3_kernel<<<(b + 1) / 2, 256, lds, st>>>(b, pts, feat, make_layer(params1, 256, 272),
                                              make_layer(params2, 512, 256),
                                              make_layer(params3, 768, 512), out, valid);
  } else {
    return MSR3D_EINVAL;
  }
  return (int)hipGetLastError();
}

}  // extern "C"
