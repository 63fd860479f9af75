// sa_split.hip -- the set-abstraction SharedMLPs on the bf16 matrix pipe at fp32 accuracy.
//
// gfx950 has no reduced-precision fast path for f32 inputs (no xf32): the f32-input MFMA runs at the
// vector rate, 1/16 of the bf16 MFMA rate, and sa_fused.hip's kernels already sit at 65-77 % of that
// 157 TFLOP/s peak.  The only way further down is the pipe that is 16x faster.  Every fp32 operand is
// split EXACTLY into three bf16 terms (round-to-nearest-even residuals),
//
//     x = x0 + x1 + x2 (+ e),   |e| <= 2^-27 |x|        w = w0 + w1 + w2 (+ e')
//
// and a product is evaluated as the six bf16 MFMA products whose magnitudes exceed 2^-24 of the result,
//
//     x w  ~  x0 w0 + x0 w1 + x1 w0 + x1 w1 + x0 w2 + x2 w0            (dropped: ~2^-26 |x w|)
//
// each exact in the fp32 accumulator (8-bit x 8-bit significands) and summed there.  The relative
// error per product is below ONE fp32 rounding of it (2^-24); measured against float64 the level
// outputs are as close as the f32-MFMA kernels' (tests/test_sa_split_gpu.py), so the tolerance of the
// path (2e-5 rel-L2 on features) is unchanged.  6 MFMAs x 16 cycles replace 8 x 32 per 16x16x32
// block of products: 2.67x less matrix-pipe time.
//
// What moves instead is data: three bf16 planes are 6 bytes per operand element instead of 4, and
// the matrix pipe now consumes them 2.67x faster, so the weight stream out of L2 (every block reads
// all of a level's weights) becomes the co-limiter: ~32 B/clk/CU at the full MFMA rate, which takes
// >= 8 KB of loads in flight per wave (tools/probe/l2_stream.hip: 10 B/clk/CU at one load in flight
// per wave, 31-40 at eight).  Hence: weights pre-split on the host and packed in 16x16x32 fragment
// order (one coalesced 1 KB read per (slab, tile, plane)), a whole slab of fragments prefetched ahead,
// activations split ONCE where they are produced (epilogue / gather) and kept in LDS as three bf16
// planes, operand roles swapped (D = W X^T) so that a lane ends up with four consecutive CHANNELS of
// one row -- one 8-byte LDS store per plane -- instead of four rows of one channel.
//
// Same contract as sa_fused.hip (msr3d_sa_level): same index ops (shared code), same folded BN affine
// and ReLU on the accumulators, same max over the neighbourhood; /root/reference/modules/third_party/
// pointnet2/pointnet2_modules.py:34-75, pytorch_utils.py:11-36.
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "pn2_device.h"

namespace {

using namespace msr3d;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kNS = 32;           // neighbours per centre (configs/msr3d.yaml:199)
constexpr int kFragS = 512;       // bf16 per (slab, tile, plane) block: 64 lanes x 8
constexpr int kPadH = 16;         // LDS row stride = K + 16 bf16: conflict-free 16-byte fragment reads

struct LayerS {
  const unsigned short *w;        // [K/32][N/16][3][64][8] bf16, fragment order
  const float *scale, *shift;     // [N]
};

// ---- exact three-way split of fp32 values into bf16 (RNE) ------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {       // v_cvt_pk_bf16_f32
  const f32x2 v = {a, b};
  const bf16x2 r = __builtin_convertvector(v, bf16x2);
  return *reinterpret_cast<const unsigned *>(&r);
}
// four values -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&p)[3]) {
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned lo = pk_bf16(a0, a1), hi = pk_bf16(a2, a3);
    p[k] = make_uint2(lo, hi);
    if (k < 2) {
      a0 -= __uint_as_float(lo << 16); a1 -= __uint_as_float(lo & 0xffff0000u);
      a2 -= __uint_as_float(hi << 16); a3 -= __uint_as_float(hi & 0xffff0000u);
    }
  }
}

// one slab's weight fragments of RN column tiles, all three planes: RN x 3 coalesced 1 KB reads
template <int RN>
struct WFrag { bf16x8 v[RN][3]; };

template <int RN, int NT>
__device__ __forceinline__ void load_w(WFrag<RN> &f, const unsigned short *__restrict__ wg, int s, int lane) {
#pragma unroll
  for (int rn = 0; rn < RN; ++rn)
#pragma unroll
    for (int p = 0; p < 3; ++p)
      f.v[rn][p] = *reinterpret_cast<const bf16x8 *>(wg + ((size_t)(s * NT + rn) * 3 + p) * kFragS + lane * 8);
}

// acc[rn][mt] (+)= W[n-tile rn] X[m-tile mt]^T over KS slabs of 32.  X: LDS planes [3][TM][ldh] bf16.
// D layout: lane (j = lane & 15, g = lane >> 4) holds rows n = 4 g + r (r = 0..3), column m = j.
template <int RN, int MT, int KS, int NT>
__device__ __forceinline__ void gemm_split(const unsigned short *xs, int ldh, int plane, const unsigned short *__restrict__ wg,
                                           f32x4 (&acc)[RN][MT], int lane, const WFrag<RN> &first) {
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *xp = xs + j * ldh + 8 * g;
  WFrag<RN> wc = first;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    WFrag<RN> wn;
    load_w<RN, NT>(wn, wg, s + 1 < KS ? s + 1 : s, lane);        // last slab: harmless re-read
    bf16x8 x[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        x[mt][p] = *reinterpret_cast<const bf16x8 *>(xp + p * plane + mt * 16 * ldh + 32 * s);
    // the six significant products, SMALLEST first (the accumulator meets the big terms last);
    // consecutive MFMAs hit different accumulators
#define MSR3D_TERM(PW, PX)                                                                             \
    _Pragma("unroll") for (int rn = 0; rn < RN; ++rn)                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
        acc[rn][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc.v[rn][PW], x[mt][PX], acc[rn][mt], 0, 0, 0);
    MSR3D_TERM(2, 0)
    MSR3D_TERM(0, 2)
    MSR3D_TERM(1, 1)
    MSR3D_TERM(1, 0)
    MSR3D_TERM(0, 1)
    MSR3D_TERM(0, 0)
#undef MSR3D_TERM
    wc = wn;
  }
}

template <int RN, int MT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RN][MT]) {
#pragma unroll
  for (int a = 0; a < RN; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// y = relu(acc * scale[n] + shift[n]), split, -> LDS planes [3][TM][ldy] at columns n0 + 16 rn + 4 g ..
template <int RN, int MT>
__device__ __forceinline__ void store_split(const f32x4 (&acc)[RN][MT], const float4 (&sc)[RN], const float4 (&sh)[RN],
                                            unsigned short *ys, int ldy, int plane, int n0, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
    const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(__builtin_fmaf(acc[rn][mt][r], s4[r], h4[r]), 0.0f);
      uint2 p[3];
      split4(v, p);
      unsigned short *d = ys + (mt * 16 + j) * ldy + n0 + rn * 16 + 4 * g;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * plane) = p[k];
    }
  }
}

__device__ __forceinline__ float row16_max(float v) {     // all-reduce max over the lane's 16-lane row
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)));
  return v;
}

// max over groups of GT m-tiles of relu(acc * scale + shift) -> global out[group][n] (n0 + 16 rn + 4 g ..)
template <int RN, int MT, int GT>
__device__ __forceinline__ void store_groupmax(const f32x4 (&acc)[RN][MT], const float4 (&sc)[RN], const float4 (&sh)[RN],
                                               float *__restrict__ out, int ldo, int n0, int groups_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
    const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
#pragma unroll
    for (int gq = 0; gq < MT / GT; ++gq) {
      float m[4] = {0.f, 0.f, 0.f, 0.f};              // starting the max at 0 IS the ReLU
#pragma unroll
      for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], __builtin_fmaf(acc[rn][gq * GT + t][r], s4[r], h4[r]));
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = row16_max(m[r]);
      if (j == 0 && gq < groups_valid)
        *reinterpret_cast<float4 *>(out + (size_t)gq * ldo + n0 + rn * 16 + 4 * g) = make_float4(m[0], m[1], m[2], m[3]);
    }
  }
}

template <int RN>
__device__ __forceinline__ void load_affine4(const float *__restrict__ scale, const float *__restrict__ shift, int n0,
                                             int lane, float4 (&sc)[RN], float4 (&sh)[RN]) {
  const int g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    sc[rn] = *reinterpret_cast<const float4 *>(scale + n0 + rn * 16 + 4 * g);
    sh[rn] = *reinterpret_cast<const float4 *>(shift + n0 + rn * 16 + 4 * g);
  }
}

// ---------------------------------------------------------------------------------------------------
// Three chained layers on a TM-row tile.  bufA: input planes [3][TM][K0P + 16], re-used for layer 2's
// output [3][TM][N2 + 16]; bufB: layer 1's output [3][TM][N1 + 16].  4 waves as 1 x 4: every wave owns
// all TM rows and a quarter of the channels, so each weight fragment is fetched once per block.
// ---------------------------------------------------------------------------------------------------
template <int TM, int K0P, int N1, int N2, int N3, int G>
struct ChainS {
  static constexpr int MT = TM / 16, GT = G / 16;
  static constexpr int LDA = (K0P > N2 ? K0P : N2) + kPadH, LDB = N1 + kPadH;
  static constexpr int PA = TM * LDA, PB = TM * LDB;          // plane strides (bf16 units)
  static constexpr int LDS_HALVES = 3 * PA + 3 * PB;
  static constexpr int RN1 = N1 / 64, RN2 = N2 / 64, RN3 = N3 / 64;
  static_assert(K0P % 32 == 0 && N1 % 64 == 0 && N2 % 64 == 0 && N3 % 64 == 0 && MT % GT == 0, "shape");

  struct Pre1 { WFrag<RN1> w; float4 sc[RN1], sh[RN1]; };
  __device__ static void preload(const LayerS &l1, Pre1 &p, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    load_w<RN1, N1 / 16>(p.w, l1.w + (size_t)(wave * RN1) * 3 * kFragS, 0, lane);
    load_affine4<RN1>(l1.scale, l1.shift, wave * RN1 * 16, lane, p.sc, p.sh);
  }

  __device__ static void run(unsigned short *bufA, unsigned short *bufB, const Pre1 &p1, const LayerS &l1,
                             const LayerS &l2, const LayerS &l3, float *__restrict__ out, int groups_valid, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    WFrag<RN2> w2;
    WFrag<RN3> w3;
    float4 sc2[RN2], sh2[RN2], sc3[RN3], sh3[RN3];
    {
      f32x4 acc[RN1][MT];
      zero_acc(acc);
      gemm_split<RN1, MT, K0P / 32, N1 / 16>(bufA, LDA, PA, l1.w + (size_t)(wave * RN1) * 3 * kFragS, acc, lane, p1.w);
      // the next layer's first operands fly while this layer's epilogue and barrier run
      load_w<RN2, N2 / 16>(w2, l2.w + (size_t)(wave * RN2) * 3 * kFragS, 0, lane);
      load_affine4<RN2>(l2.scale, l2.shift, wave * RN2 * 16, lane, sc2, sh2);
      store_split<RN1, MT>(acc, p1.sc, p1.sh, bufB, LDB, PB, wave * RN1 * 16, lane);
    }
    __syncthreads();
    {
      f32x4 acc[RN2][MT];
      zero_acc(acc);
      gemm_split<RN2, MT, N1 / 32, N2 / 16>(bufB, LDB, PB, l2.w + (size_t)(wave * RN2) * 3 * kFragS, acc, lane, w2);
      load_w<RN3, N3 / 16>(w3, l3.w + (size_t)(wave * RN3) * 3 * kFragS, 0, lane);
      load_affine4<RN3>(l3.scale, l3.shift, wave * RN3 * 16, lane, sc3, sh3);
      store_split<RN2, MT>(acc, sc2, sh2, bufA, N2 + kPadH, TM * (N2 + kPadH), wave * RN2 * 16, lane);
    }
    __syncthreads();
    {
      f32x4 acc[RN3][MT];
      zero_acc(acc);
      gemm_split<RN3, MT, N2 / 32, N3 / 16>(bufA, N2 + kPadH, TM * (N2 + kPadH), l3.w + (size_t)(wave * RN3) * 3 * kFragS,
                                            acc, lane, w3);
      store_groupmax<RN3, MT, GT>(acc, sc3, sh3, out, N3, wave * RN3 * 16, groups_valid, lane);
    }
  }
};

// ball query of ONE centre by ONE wave over a cloud staged in LDS (sa_fused.hip's, verbatim semantics:
// ball_query_gpu.cu:9-44 -- index order, strict '<', first-hit fill, zeros when empty)
__device__ __forceinline__ void wave_ball_query(const float *sx, int n, float cx, float cy, float cz, float radius2,
                                                int nsample, int *row, int lane) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < nsample; base += kWave) {
    const int k = base + lane;
    bool hit = false;
    if (k < n) hit = sq3(cx - sx[k * 3 + 0], cy - sx[k * 3 + 1], cz - sx[k * 3 + 2]) < radius2;
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

// =====================================================================================================
// Level 2: xyz (b, n <= 64, 3), feat (b, n, 128) point-major fp32; centres (b, m, 3).  Block = 2 centres
// x 32 neighbours.  MLP 131 -> 128 -> 128 -> 256, K order [feat(128), dxyz(3), 0 x 29].  out (b, m, 256).
// =====================================================================================================
using Chain2S = ChainS<2 * kNS, 160, 128, 128, 256, kNS>;

__global__ __launch_bounds__(256) void sa2_split_kernel(int n, int m, float radius2, const float *__restrict__ xyz,
                                                        const float *__restrict__ feat, const float *__restrict__ new_xyz,
                                                        LayerS l1, LayerS l2, LayerS l3, float *__restrict__ out,
                                                        int *__restrict__ dbg_idx, const unsigned char *__restrict__ valid) {
  if (valid && !valid[blockIdx.y]) return;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  constexpr int CPB = 2, TM = CPB * kNS;
  unsigned short *bufA = smem, *bufB = smem + 3 * Chain2S::PA;
  int *nbr = reinterpret_cast<int *>(smem + Chain2S::LDS_HALVES);       // [CPB][32]
  float *ctr = reinterpret_cast<float *>(nbr + 4 * kNS);                // [4][4]
  float *sx = ctr + 16;                                                  // [n][3], n <= 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int obj = blockIdx.y, c0 = blockIdx.x * CPB;

  Chain2S::Pre1 pre;
  Chain2S::preload(l1, pre, tid);           // layer-1 weights / affine in flight during the loader phase
  if (tid < n * 3) sx[tid] = xyz[(size_t)obj * n * 3 + tid];
  if (tid >= 192 && tid < 192 + 3 * CPB) {
    const int t = tid - 192, w = t / 3, c = t - w * 3;
    ctr[w * 4 + c] = (c0 + w < m) ? new_xyz[((size_t)obj * m + c0 + w) * 3 + c] : 0.f;
  }
  __syncthreads();
  if (wave < CPB) {
    if (c0 + wave < m)
      wave_ball_query(sx, n, ctr[wave * 4 + 0], ctr[wave * 4 + 1], ctr[wave * 4 + 2], radius2, kNS, nbr + wave * kNS, lane);
    else if (lane < kNS)
      nbr[wave * kNS + lane] = 0;
  }
  __syncthreads();
  if (dbg_idx && tid < CPB * kNS && c0 + tid / kNS < m) dbg_idx[((size_t)obj * m + c0) * kNS + tid] = nbr[tid];
  const float *F = feat + (size_t)obj * n * 128;
  {   // 32 float4 per row; indices first, then ALL loads, then split + LDS stores: one L2 round trip
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
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 p[3];
      split4(v, p);
      unsigned short *d = bufA + (e >> 5) * Chain2S::LDA + (e & 31) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * Chain2S::PA) = p[k];
    }
  }
  if (tid < TM) {                  // columns 128..159: [dx, dy, dz, 0 ...]
    const int row = tid, pi = nbr[row], w = row >> 5;
    const float v[4] = {sx[pi * 3 + 0] - ctr[w * 4 + 0], sx[pi * 3 + 1] - ctr[w * 4 + 1], sx[pi * 3 + 2] - ctr[w * 4 + 2], 0.f};
    uint2 p[3];
    split4(v, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      unsigned short *d = bufA + k * Chain2S::PA + row * Chain2S::LDA + 128;
      *reinterpret_cast<uint2 *>(d) = p[k];
#pragma unroll
      for (int c = 4; c < 32; c += 4) *reinterpret_cast<uint2 *>(d + c) = make_uint2(0u, 0u);
    }
  }
  __syncthreads();
  int groups = m - c0;
  groups = groups < 0 ? 0 : (groups < CPB ? groups : CPB);
  Chain2S::run(bufA, bufB, pre, l1, l2, l3, out + ((size_t)obj * m + c0) * 256, groups, tid);
}

template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  static size_t granted = 0;
  if (bytes <= granted) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) granted = bytes;
  return e;
}

inline LayerS make_layer(const void *w, const float *affine, int n) {
  LayerS l;
  l.w = reinterpret_cast<const unsigned short *>(w);
  l.scale = affine;
  l.shift = affine + n;
  return l;
}

}  // namespace

extern "C" int msr3d_sa_level_split(int level, int b, int n, int m, float radius, const float *pts, const float *feat,
                                    const float *new_xyz, const void *w1, const float *affine1, const void *w2,
                                    const float *affine2, const void *w3, const float *affine3, float *out,
                                    int *dbg_ball_idx, const unsigned char *valid, msr3d_stream_t stream) {
  if (b < 0) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!w1 || !w2 || !w3 || !affine1 || !affine2 || !affine3 || !out) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float r2 = radius * radius;   // f32 product, as ball_query_gpu.cu:22
  hipError_t e;
  if (level == 2) {
    if (!pts || !feat || !new_xyz || n <= 0 || n > 64 || m <= 0) return MSR3D_EINVAL;
    const size_t lds = sizeof(unsigned short) * Chain2S::LDS_HALVES + sizeof(int) * (4 * kNS + 16 + 64 * 3);
    if ((e = allow_lds(sa2_split_kernel, lds)) != hipSuccess) return (int)e;
    dim3 grid((m + 1) / 2, b);
    sa2_split_kernel<<<grid, 256, lds, st>>>(n, m, r2, pts, feat, new_xyz, make_layer(w1, affine1, 128),
                                             make_layer(w2, affine2, 128), make_layer(w3, affine3, 256), out,
                                             dbg_ball_idx, valid);
  } else {
    return MSR3D_EINVAL;
  }
  return (int)hipGetLastError();
}
