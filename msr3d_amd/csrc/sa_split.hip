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
// What moves instead is data: three bf16 planes are 6 bytes per operand element instead of 4, and the
// matrix pipe consumes them 2.67x faster, so the weight stream out of L2 (every tile reads all of a
// level's weights: 418 KB per 64 rows at level 2) and every global round trip of a tile's query + gather
// show.  Hence:
//   * weights pre-split on the host, packed in 16x16x32 fragment order (one coalesced 1 KB read per
//     (slab, tile, plane)), streamed through a buffer descriptor and a register ring four pieces
//     (12 KB per wave) deep -- tools/probe/l2_stream.hip: 10 B/clk/CU at one load in flight per wave,
//     31-40 at eight;
//   * activations split ONCE where they are produced (epilogue / gather) and kept in LDS as three bf16
//     planes; operand roles swapped (D = W X^T) so that a lane ends up with four consecutive CHANNELS
//     of one row -- one 8-byte LDS store per plane -- instead of four rows of one channel;
//   * one operand buffer rewritten in place by each epilogue (66 KB, two blocks per CU) and blocks
//     persistent over runs of tiles with the next tile's geometry, ball query and neighbour rows
//     fetched under the current tile's layers (see the kernel).
// Measured (16 x 60 objects): 0.531 ms for the f32-MFMA kernel (122 of the 155.6 TFLOP/s a pure
// issue loop sustains, tools/probe/mfma_peak.hip) -> 0.300 ms; the bf16 pipe is 56 % busy (PMC), the rest is each block's serial chain
// of epilogues and barriers (phase stamps: tools/ab_split.py with tools/prof/sa_split_stamped.hip).
//
// Same contract as sa_fused.hip (msr3d_sa_level): same index ops (shared code), same folded BN affine
// and ReLU on the accumulators, same max over the neighbourhood; /root/reference/modules/third_party/
// pointnet2/pointnet2_modules.py:34-75, pytorch_utils.py:11-36.
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "../../include/msr3d_hip.h"
#include "pn2_device.h"

// phase marks of the level-2 kernel: empty here; tools/prof/sa_split_stamped.hip defines them and includes
// this file (tools/ab_split.py reads the stamps)
// MSR3D_SPLIT_TERMS = 3 (python -m msr3d_amd.build builds libmsr3d_hip_split2.so from this file with it): the LABELLED
// reduced variant MSR3D_SA_MMA=split2 -- two bf16 terms per operand, the three products x0 w0 + x0 w1 + x1 w0 (~16
// significant bits per product; the reference's cuDNN convolutions run TF32, 10 bits, by default on any Ampere-or-later
// GPU: README.md:81 pins torch 1.12.1 and nothing in the reference touches allow_tf32).  Never the headline.
#ifndef MSR3D_SPLIT_TERMS
#define MSR3D_SPLIT_TERMS 6
#endif
#if MSR3D_SPLIT_TERMS == 6     // smallest first: the accumulator meets the big terms last
#define MSR3D_TERMS_ALL MSR3D_TERM(2, 0) MSR3D_TERM(0, 2) MSR3D_TERM(1, 1) MSR3D_TERM(1, 0) MSR3D_TERM(0, 1) MSR3D_TERM(0, 0)
#elif MSR3D_SPLIT_TERMS == 3
#define MSR3D_TERMS_ALL MSR3D_TERM(1, 0) MSR3D_TERM(0, 1) MSR3D_TERM(0, 0)
#else
#error "MSR3D_SPLIT_TERMS must be 6 or 3"
#endif
#ifndef RSTAMP                 // phase marks of the distinct-row kernels (tools/prof/sa_rows_stamped.hip)
#define RSTAMP(i)
#define RSTAMP_DECL
#endif
#ifndef STAMP
#define STAMP(i)
#define STAMP_DECL
#define STAMP_TILE_TOP
#define STAMP_TILE_END
#endif
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

// One PIECE of weights = one (slab, column tile): three planes, 3 x 1 KB coalesced reads per wave.
// Pieces are consumed in (slab, tile) order and fetched D pieces ahead through a register ring, so a
// wave keeps D x 3 KB in flight whatever the layer width (tools/probe/l2_stream.hip: the L2 -> CU
// stream needs >= 8 loads in flight per wave to pass 30 B/clk/CU).
struct WPiece { bf16x8 v[3]; };

// The stream goes through a buffer descriptor: wave-uniform base (SGPRs) + one per-lane byte offset +
// an SGPR piece offset.  With flat loads hipcc materialises every piece's 64-bit address in its own
// VGPR pair and hoists them all out of the persistent loop (~180 VGPRs, all spilled).
struct WStream {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;                        // lane * 16
  int soff;                        // the wave's first column tile, bytes
};
template <int RN>
__device__ __forceinline__ WStream make_stream(const unsigned short *w, int bytes, int wave_uniform, int lane) {
  WStream st;
  st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, bytes, 0x00020000);
  st.voff = lane * 16;
  st.soff = wave_uniform * RN * 3 * kFragS * 2;
  return st;
}

template <int NT>
__device__ __forceinline__ void load_piece(WPiece &f, const WStream &st, int s, int rn, int /*lane*/) {
  const int piece = st.soff + (s * NT + rn) * 3 * kFragS * 2;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, st.voff, piece + p * kFragS * 2, 0);
    f.v[p] = *reinterpret_cast<const bf16x8 *>(&r);
  }
}

template <int RN, int NT, int D>
__device__ __forceinline__ void preload_ring(WPiece (&ring)[D], const WStream &st, int lane) {
#pragma unroll
  for (int q = 0; q < D; ++q) load_piece<NT>(ring[q], st, q / RN, q % RN, lane);
}

// acc[rn][mt] (+)= W[n-tile rn] X[m-tile mt]^T over KS slabs of 32.  X: LDS planes [3][TM][ldh] bf16.
// D layout: lane (j = lane & 15, g = lane >> 4) holds rows n = 4 g + r (r = 0..3), column m = j.
// `ring` holds pieces 0..D-1 on entry (fetched under the previous phase).
template <int RN, int MT, int KS, int NT, int D>
__device__ __forceinline__ void gemm_split(const unsigned short *xs, int ldh, int plane, const WStream &wg,
                                           f32x4 (&acc)[RN][MT], int lane, const WPiece (&ring)[D]) {
  constexpr int NP = KS * RN;
  static_assert(D <= NP, "ring deeper than the layer");
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *xp = xs + j * ldh + 8 * g;
  WPiece w[NP];                    // fully unrolled: only a window of D + 1 pieces is ever live
#pragma unroll
  for (int q = 0; q < D; ++q) w[q] = ring[q];
  bf16x8 x[MT][3];                 // one slab of the operand (prefetching the next one a piece early: no gain)
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int s = q / RN, rn = q % RN;
    __builtin_amdgcn_sched_barrier(0);          // this piece's fetches stay below the previous piece's MFMAs ..
    if (q + D < NP) load_piece<NT>(w[q + D], wg, (q + D) / RN, (q + D) % RN, lane);
    if (rn == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          x[mt][p] = *reinterpret_cast<const bf16x8 *>(xp + p * plane + mt * 16 * ldh + 32 * s);
    }
    __builtin_amdgcn_sched_barrier(0);          // .. and above its own
    // the six significant products, SMALLEST first (the accumulator meets the big terms last);
    // consecutive MFMAs hit different accumulators
#define MSR3D_TERM(PW, PX)                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
        acc[rn][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q].v[PW], x[mt][PX], acc[rn][mt], 0, 0, 0);
    MSR3D_TERMS_ALL
#undef MSR3D_TERM
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

// ---- level 2 (round 3): operand planes in two layouts ------------------------------------------------------
// ROWS  [3][rows][pitch] row-major: what the gather writes (16 lanes per row, 8 contiguous bytes each: one
//       bank row per store group); pitch = K + 16 bf16 keeps the 16-byte fragment reads conflict-free.
// FRAG  [slab][4 row tiles][3 planes][64 lanes][16 B] MFMA-fragment order: what an epilogue writes.  A lane
//       of D = W X^T holds four consecutive channels of ONE row, so its 8-byte store goes to slot
//       (row, channel / 8); the 16 lanes of a store group are 16 rows = 16 consecutive 16-byte slots: two
//       lanes per bank row instead of the four the row-major image gave (38 % of this kernel's LDS cycles
//       were those conflicts, profiles/r02_v6_pmc_sa.txt), and every fragment READ is a contiguous 1 KB.
struct XRowsS {
  const unsigned short *base;      // + j * pitch + 8 g
  int pitch, plane;
  __device__ __forceinline__ bf16x8 operator()(int mt, int s, int p) const {
    return *reinterpret_cast<const bf16x8 *>(base + p * plane + mt * 16 * pitch + 32 * s);
  }
};
struct XFragS {
  const unsigned short *base;      // + lane * 8
  __device__ __forceinline__ bf16x8 operator()(int mt, int s, int p) const {
    return *reinterpret_cast<const bf16x8 *>(base + ((s * 4 + mt) * 3 + p) * 512);
  }
};

template <int RN, int MT, int KS, int NT, int D, typename XF>
__device__ __forceinline__ void gemm_split_x(const XF &xf, const WStream &wg, f32x4 (&acc)[RN][MT], int lane,
                                             const WPiece (&ring)[D]) {
  constexpr int NP = KS * RN;
  static_assert(D <= NP, "ring deeper than the layer");
  WPiece w[NP];
#pragma unroll
  for (int q = 0; q < D; ++q) w[q] = ring[q];
  bf16x8 x[MT][3];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int s = q / RN, rn = q % RN;
    __builtin_amdgcn_sched_barrier(0);
    if (q + D < NP) load_piece<NT>(w[q + D], wg, (q + D) / RN, (q + D) % RN, lane);
    if (rn == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p) x[mt][p] = xf(mt, s, p);
    }
    __builtin_amdgcn_sched_barrier(0);
#define MSR3D_TERM(PW, PX)                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
        acc[rn][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q].v[PW], x[mt][PX], acc[rn][mt], 0, 0, 0);
    MSR3D_TERMS_ALL
#undef MSR3D_TERM
  }
}

// y = relu(acc * scale[n] + shift[n]), split -> FRAG planes (4 row tiles) at channels n0 + 16 rn + 4 g ..
template <int RN, int MT>
__device__ __forceinline__ void store_split_frag(const f32x4 (&acc)[RN][MT], const float4 (&sc)[RN], const float4 (&sh)[RN],
                                                 unsigned char *ys, int n0, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
    const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
    const int c = n0 + rn * 16 + 4 * g, slab = c >> 5, kk = c & 31;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(__builtin_fmaf(acc[rn][mt][r], s4[r], h4[r]), 0.0f);
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        *reinterpret_cast<uint2 *>(ys + ((((slab * 4 + mt) * 3 + k) * 64 + j + 16 * (kk >> 3)) * 16 + (kk & 7) * 2)) = p[k];
    }
  }
}

// max over each lane's 16-lane row (rotations by 8, 4, 2, 1), four values at a time so that every DPP
// read sits three instructions behind the write it depends on (a DPP read needs two wait states after a
// VALU write of the same register and inline asm gets no hazard padding); v_max_f32 with a DPP source is
// one instruction per step -- hipcc does not fold update_dpp + fmaxf into it.
__device__ __forceinline__ void row16_max4(float (&m)[4]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]));
}

// relu(acc * scale + shift) maxed over each group's GT m-tiles, per lane: the accumulators die here
template <int RN, int MT, int GT>
__device__ __forceinline__ void group_reduce(const f32x4 (&acc)[RN][MT], const float4 (&sc)[RN], const float4 (&sh)[RN],
                                             float (&m)[RN][MT / GT][4]) {
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
    const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
#pragma unroll
    for (int gq = 0; gq < MT / GT; ++gq) {
#pragma unroll
      for (int r = 0; r < 4; ++r) m[rn][gq][r] = 0.f;              // starting the max at 0 IS the ReLU
#pragma unroll
      for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          m[rn][gq][r] = fmaxf(m[rn][gq][r], __builtin_fmaf(acc[rn][gq * GT + t][r], s4[r], h4[r]));
    }
  }
}
// ... over the 16 rows a tile holds across lanes, -> global out[group][n] (n0 + 16 rn + 4 g ..)
template <int RN, int NG>
__device__ __forceinline__ void group_finish(float (&m)[RN][NG][4], float *__restrict__ out, int ldo, int n0,
                                             int groups_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn)
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      row16_max4(m[rn][gq]);
      if (j == 0 && gq < groups_valid)
        *reinterpret_cast<float4 *>(out + (size_t)gq * ldo + n0 + rn * 16 + 4 * g) =
            make_float4(m[rn][gq][0], m[rn][gq][1], m[rn][gq][2], m[rn][gq][3]);
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
// Level 2: xyz (b, n <= 64, 3), feat (b, n, 128) point-major fp32; centres (b, m, 3).  A TILE is 2 centres
// x 32 neighbours = 64 rows; MLP 131 -> 128 -> 128 -> 256, K order [feat(128), dxyz(3), 0 x 29];
// out (b, m, 256).
//
// The block is PERSISTENT over a contiguous run of tiles and software-pipelined across them: one tile
// costs ~14k cycles of matrix pipe per wave but its query + gather is a chain of three dependent global
// round trips (~10k cycles when exposed, measured with s_memtime stamps), so the next tile's geometry is
// fetched under layer 1, its ball query runs under layer 1's epilogue, its 32 KB of neighbour features
// fly under layer 3 and are split into the operand planes while layer 3's maxima are stored.  ONE LDS
// operand buffer, rewritten in place by each epilogue (66 KB): two blocks per CU, so one block's
// epilogues and barriers sit under the other's MFMAs.  4 waves as 1 x 4: every wave owns all 64 rows and
// a quarter of the channels -- each weight fragment is fetched once per tile.
// =====================================================================================================
constexpr int kTM = 2 * kNS;                 // rows per tile
// Layer 1's K is the 128 feature channels: the three recentred coordinates enter as fp32 FMAs on the
// accumulators (a fifth, 29/32-empty slab of MFMAs in round 2: 9 % of the layer's matrix work).
constexpr int kK0 = 128, kN1 = 128, kN2 = 128, kN3 = 256;
constexpr int kLd = kK0 + kPadH;             // ROWS pitch of layer 1's operand (144: conflict-free b128 reads)
constexpr int kPlane = kTM * kLd;            // plane stride, bf16 units
constexpr int kFragBytes = 4 * 4 * 3 * 1024; // FRAG image of a 64 x 128 operand (layers 2, 3): 49,152 B
static_assert(3 * kPlane * 2 >= kFragBytes, "the two images share one buffer");
constexpr int kRing = 4;                     // weight pieces in flight per wave (12 KB)
constexpr int kSa2Lds = 3 * kPlane * 2 + (2 * (kN1 + kN2 + kN3) + 64 * 3 + 16 + kTM * 4) * 4 + 2 * kNS * 4 + 16;
constexpr int kChunk = 3;                   // tiles per queue fetch (15 per block at the bench shape: five fetches)

__global__ __launch_bounds__(256, 2) void sa2_split_kernel(int n, int m, int tiles, int *__restrict__ queue, float radius2,
                                                           const float *__restrict__ xyz, const float *__restrict__ feat,
                                                           const float *__restrict__ new_xyz, LayerS l1, LayerS l2, LayerS l3,
                                                           float *__restrict__ out, int *__restrict__ dbg_idx,
                                                           const unsigned char *__restrict__ valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *buf = smem;                                             // [3][64][176] bf16
  float *aff = reinterpret_cast<float *>(smem + 3 * kPlane);               // sc1 sh1 sc2 sh2 sc3 sh3
  float *sx = aff + 2 * (kN1 + kN2 + kN3);                                 // [n][3], n <= 64
  float *ctr = sx + 64 * 3;                                                // [2][4]
  int *nbr = reinterpret_cast<int *>(ctr + 16);                            // [2][32]
  int *s_next = nbr + 2 * kNS;                                             // [2]: tile hand-over from thread 0
  float *dxs = reinterpret_cast<float *>(s_next + 4);                      // [64][4]: the rows' recentred coordinates
  unsigned char *fbuf = reinterpret_cast<unsigned char *>(buf);            // the FRAG image of layers 2 / 3 (same buffer)
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpo = (m + 1) >> 1;                                            // tiles per object
  const int t_end = tiles;                                                 // (also the "no tile" value)
  STAMP_DECL;

  // Tiles are handed out in CHUNKS of kChunk consecutive tiles from a device-wide queue (queue[0]: next
  // chunk, queue[1]: blocks finished; the last block to leave resets both for the next launch).  A static
  // split would be just as good on an idle chip -- but when other kernels hold CUs (the gradient
  // all-reduce runs beside the next batch's encoder in the data-parallel step) the blocks that start late
  // would carry a whole share each and double the kernel's time; from a queue they simply take less.
  // Thread 0 walks the queue; the tile AFTER the next one is published through LDS between two of the
  // loop's barriers, so the atomic's round trip sits under a layer.
  int chunk_end = 0;                         // (thread 0's view of the chunk it is consuming)
  auto advance = [&](int t) {                // thread 0 only: the next encoded tile after t, or t_end
    ++t;
    while (true) {
      if (t >= chunk_end) {
        const int c = __hip_atomic_fetch_add(queue, kChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c >= tiles) return t_end;
        t = c;
        chunk_end = min(c + kChunk, tiles);
      }
      if (!valid || valid[t / tpo]) return t;
      ++t;
    }
  };
  auto leave = [&]() {                       // thread 0 only
    const int done = __hip_atomic_fetch_add(queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (tid == 0) {
    const int t0 = advance(-1);              // (chunk_end = 0: fetches the first chunk)
    s_next[0] = t0;
    s_next[1] = t0 < t_end ? advance(t0) : t_end;
  }
  __syncthreads();
  int T = s_next[0], Tn = s_next[1];
  if (T >= t_end) {
    if (tid == 0) leave();
    return;
  }

  // folded BN affines: once per block
  for (int i = tid; i < kN1; i += 256) { aff[i] = l1.scale[i]; aff[kN1 + i] = l1.shift[i]; }
  for (int i = tid; i < kN2; i += 256) { aff[2 * kN1 + i] = l2.scale[i]; aff[2 * kN1 + kN2 + i] = l2.shift[i]; }
  for (int i = tid; i < kN3; i += 256) { aff[2 * (kN1 + kN2) + i] = l3.scale[i]; aff[2 * (kN1 + kN2) + kN3 + i] = l3.shift[i]; }
  const float *sc1 = aff, *sh1 = aff + kN1, *sc2 = aff + 2 * kN1, *sh2 = sc2 + kN2, *sc3 = aff + 2 * (kN1 + kN2), *sh3 = sc3 + kN3;

  constexpr int RN1 = kN1 / 64, RN2 = kN2 / 64, RN3 = kN3 / 64, MT = kTM / 16, GT = kNS / 16;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);                 // uniform to the compiler as well
  const WStream w1 = make_stream<RN1>(l1.w, kK0 * kN1 * 6, wave_u, lane);
  const float *__restrict__ wxyz = l1.shift + kN1;                         // [128][4] fp32: layer 1's weights on (dx, dy, dz)
  const WStream w2 = make_stream<RN2>(l2.w, kN1 * kN2 * 6, wave_u, lane);
  const WStream w3 = make_stream<RN3>(l3.w, kN2 * kN3 * 6, wave_u, lane);

  // ---- the pieces of a tile's query + gather ----
  float geo = 0.f;                           // one xyz / centre coordinate of the NEXT tile per thread
  auto geo_fetch = [&](int t) {              // global -> register
    const int obj = t / tpo, c0 = (t - obj * tpo) * 2;
    if (tid < n * 3) geo = xyz[(size_t)obj * n * 3 + tid];
    else if (tid >= 192 && tid < 198) {
      const int q = tid - 192, w = q / 3, c = q - w * 3;
      geo = (c0 + w < m) ? new_xyz[((size_t)obj * m + c0 + w) * 3 + c] : 0.f;
    }
  };
  auto geo_store = [&]() {                   // register -> LDS
    if (tid < n * 3) sx[tid] = geo;
    else if (tid >= 192 && tid < 198) { const int q = tid - 192, w = q / 3; ctr[w * 4 + (q - w * 3)] = geo; }
  };
  auto query = [&](int t) {                  // waves 0, 1: one centre each
    const int obj = t / tpo, c0 = (t - obj * tpo) * 2;
    if (wave < 2) {
      if (c0 + wave < m)
        wave_ball_query(sx, n, ctr[wave * 4 + 0], ctr[wave * 4 + 1], ctr[wave * 4 + 2], radius2, kNS, nbr + wave * kNS, lane);
      else if (lane < kNS)
        nbr[wave * kNS + lane] = 0;
    }
  };
  constexpr int IT = kTM * 32 / 256;         // float4 per thread per tile: 32 float4 per row
  float4 val[IT];
  auto feat_fetch = [&](int t) {             // indices first, then ALL loads: one L2 round trip
    const int obj = t / tpo, c0 = (t - obj * tpo) * 2;
    if (dbg_idx && tid < kTM && c0 + tid / kNS < m) dbg_idx[((size_t)obj * m + c0) * kNS + tid] = nbr[tid];
    const float *F = feat + (size_t)obj * n * 128;
    int pidx[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) pidx[it] = nbr[(tid + it * 256) >> 5];
#pragma unroll
    for (int it = 0; it < IT; ++it)
      val[it] = *reinterpret_cast<const float4 *>(F + (size_t)pidx[it] * 128 + ((tid + it * 256) & 31) * 4);
  };
  auto feat_store = [&]() {                  // split + LDS planes; columns 128..159: [dx, dy, dz, 0 ...]
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 p[3];
      split4(v, p);
      unsigned short *d = buf + (e >> 5) * kLd + (e & 31) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * kPlane) = p[k];
    }
    if (tid < kTM) {
      const int row = tid, pi = nbr[row], w = row >> 5;
      *reinterpret_cast<float4 *>(dxs + row * 4) =
          make_float4(sx[pi * 3 + 0] - ctr[w * 4 + 0], sx[pi * 3 + 1] - ctr[w * 4 + 1], sx[pi * 3 + 2] - ctr[w * 4 + 2], 0.f);
    }
  };

  // ---- prologue: the first tile's operand, unpipelined ----
  WPiece ring1[kRing];
  preload_ring<RN1, kN1 / 16, kRing>(ring1, w1, lane);
  geo_fetch(T);
  geo_store();
  __syncthreads();
  query(T);
  __syncthreads();
  feat_fetch(T);
  feat_store();
  __syncthreads();

  while (true) {
    const int obj = T / tpo, c0 = (T - obj * tpo) * 2;
    const bool more = Tn < t_end;
    // per-lane addresses are re-derived every tile: hoisted out of the loop they would sit in ~40 VGPRs
    // across the layer-3 phase and spill (and a scratch reload's vmcnt(0) drains the weight ring)
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wave = tid >> 6;
    STAMP_TILE_TOP;
    STAMP(0);
    if (more) geo_fetch(Tn);
    WPiece ring2[kRing], ring3[kRing];
    {
      f32x4 acc[RN1][MT];
      zero_acc(acc);
      const XRowsS x1{buf + (lane & 15) * kLd + 8 * (lane >> 4), kLd, kPlane};
      gemm_split_x<RN1, MT, kK0 / 32, kN1 / 16, kRing>(x1, w1, acc, lane, ring1);
      preload_ring<RN2, kN2 / 16, kRing>(ring2, w2, lane);        // the next layer's operands fly under the epilogue
      float4 sc[RN1], sh[RN1];
      load_affine4<RN1>(sc1, sh1, wave * RN1 * 16, lane, sc, sh);
      {   // + W_xyz (dx, dy, dz): lane (j, g) holds row 16 mt + j, channels n0 + 16 rn + 4 g + r
        float4 d[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) d[mt] = *reinterpret_cast<const float4 *>(dxs + (16 * mt + (lane & 15)) * 4);
#pragma unroll
        for (int rn = 0; rn < RN1; ++rn)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float4 wv = *reinterpret_cast<const float4 *>(wxyz + (wave * RN1 * 16 + rn * 16 + 4 * (lane >> 4) + r) * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[rn][mt][r] = __builtin_fmaf(wv.z, d[mt].z, __builtin_fmaf(wv.y, d[mt].y, __builtin_fmaf(wv.x, d[mt].x, acc[rn][mt][r])));
          }
      }
      STAMP(1);
      __syncthreads();                                            // (A) every wave is done READING the operand; sx/ctr free
      if (tid == 0) s_next[0] = more ? advance(Tn) : t_end;       // the tile after the next one
      if (more) geo_store();
      store_split_frag<RN1, MT>(acc, sc, sh, fbuf, wave * RN1 * 16, lane);
    }
    __syncthreads();                                              // (B) layer-1 planes + next geometry visible
    const int Tnn = s_next[0];
    STAMP(2);
    if (more) query(Tn);
    {
      f32x4 acc[RN2][MT];
      zero_acc(acc);
      const XFragS x2{buf + lane * 8};
      gemm_split_x<RN2, MT, kN1 / 32, kN2 / 16, kRing>(x2, w2, acc, lane, ring2);
      preload_ring<RN3, kN3 / 16, kRing>(ring3, w3, lane);
      float4 sc[RN2], sh[RN2];
      load_affine4<RN2>(sc2, sh2, wave * RN2 * 16, lane, sc, sh);
      STAMP(3);
      __syncthreads();                                            // (C)
      store_split_frag<RN2, MT>(acc, sc, sh, fbuf, wave * RN2 * 16, lane);
    }
    __syncthreads();                                              // (D) layer-2 planes + next neighbour lists visible
    STAMP(4);
    float touch = 0.f;                                            // the neighbour rows' 256 cache lines start moving
    float gm[RN3][MT / GT][4];
    if (more) touch = feat[((size_t)(Tn / tpo) * n + nbr[tid >> 2]) * 128 + (tid & 3) * 32];   // towards this XCD's L2
    {
      f32x4 acc[RN3][MT];
      zero_acc(acc);
      const XFragS x3{buf + lane * 8};
      gemm_split_x<RN3, MT, kN2 / 32, kN3 / 16, kRing>(x3, w3, acc, lane, ring3);
      STAMP(5);
      float4 sc[RN3], sh[RN3];
      load_affine4<RN3>(sc3, sh3, wave * RN3 * 16, lane, sc, sh);
      group_reduce<RN3, MT, GT>(acc, sc, sh, gm);
    }
    // (touched lines: L2 hits) fly under the row maxima.  UNCONDITIONAL -- after the last tile it re-reads
    // that tile: a conditional assignment would make val / ring1 loop-carried and live across the GEMMs
    feat_fetch(more ? Tn : T);
    preload_ring<RN1, kN1 / 16, kRing>(ring1, w1, lane);
    {
      int groups = m - c0;
      groups = groups < 2 ? groups : 2;
      group_finish<RN3, MT / GT>(gm, out + ((size_t)obj * m + c0) * kN3, kN3, wave * RN3 * 16, groups, lane);
    }
    if (!more) break;
    STAMP(6);
    __syncthreads();                                              // (E) every wave is done reading layer-2 planes
    asm volatile("" ::"v"(touch));
    STAMP(7);
    feat_store();
    STAMP(8);
    __syncthreads();                                              // (F) the next tile's operand is in place
    STAMP(9);
    STAMP_TILE_END;
    T = Tn;
    Tn = Tnn;
  }
  if (tid == 0) leave();
}

// =====================================================================================================
// Level 2 over DISTINCT neighbourhood rows (round 5).
//
// ball_query pads a neighbourhood that has fewer than nsample hits by repeating its first hit
// (ball_query_gpu.cu:35-39), so the rows a centre's SharedMLP sees are `filled = min(hits, nsample)`
// different (centre, point) pairs followed by nsample - filled bit-identical copies of the first one.  A row's
// way through the three layers depends on nothing but that row, and max is idempotent: the level's output is
// THE SAME BITS when only the distinct rows are multiplied.  On the benchmark's scenes a level-2 centre
// (32 points, r = 0.4) has 2.9 distinct neighbours on average -- 49 rows per object instead of 512.
//
// A tile is up to 64 distinct rows of ONE object (an object with more takes several chunks), packed densely
// across its centres; every row keeps the arithmetic of sa2_split_kernel to the letter (same gather, same
// split, same MFMA order per row, xyz as the same three FMAs, same affine + ReLU), only the 16-row tiles
// that hold no row are not multiplied (MT = 1..4 row tiles per chunk).  The maximum is SEGMENTED: rows of
// different centres share a 16-row tile, so the last epilogue is an LDS atomic max (values are >= 0 after
// the ReLU: the order of their bit patterns as unsigned integers is their order as floats) into an
// (m <= 16) x 256 table that starts at zero -- starting at zero IS the ReLU -- and is written out once per
// object.  A centre without hits keeps the reference's row of index 0 (one row).
// `constant` (b bytes, optional): objects whose cloud is one repeated point (the dataset's padding slots,
// dataset_wrapper.py:156-158; msr3d_sa_fps2* reports them): every (centre, point) row of such an object is the
// same row, so ONE row is multiplied and its result written to all m centres.
// =====================================================================================================
constexpr int kRowsMaxM = 16;                // centres per object the out table holds
constexpr int kOutPitch = 257;               // floats per centre in the table: centre c, channel n -> bank (c + n) % 32
// The PLAN of a launch (caller's workspace, kPlanBytes per object): header int = R | constant << 16 (R = the
// object's distinct rows, 0 for an object the valid mask skips), then the row list, R x (centre << 8 | point),
// sorted by centre.  Written by sa2_plan_kernel (one wave per object: the m ball queries + the list), read by
// sa2_rows_kernel an object ahead -- the dependent chain centres -> queries -> list -> gather of an object is
// off the multiplying blocks' critical path.
constexpr int kPlanRows = kRowsMaxM * kNS;   // 512
constexpr int kPlanBytes = 32 + kPlanRows * 2;

// One wave per object, ALL of its m <= 16 ball queries at once (ball_query_gpu.cu:9-44: index order, strict '<', first-hit
// fill, zeros when empty): lane 4 c + q tests its quarter of the n <= 64 points against centre c, the quad ORs the four
// partial hit masks, and slot s of the centre's row is the s-th set bit of the mask (or the first hit past the last one).
struct Sa2PlanArgs {
  int b, n, m;
  float radius2;
  const float *xyz, *new_xyz;
  int *rows_of;
  unsigned char *plan;
  int *dbg_idx;
  const unsigned char *valid, *constant;
  float *out;
};

// The plan of ONE object by ONE wave, its n <= 64 points in LDS (sx: packed xyz) and this lane's centre (lane 4 c + q:
// centre c) in registers; fcnt: kRowsMaxM words of LDS of the wave's own.
__device__ __forceinline__ void sa2_plan_wave(const int obj, const int lane, const int n, const int m, const float radius2,
                                              const float *sx, const float cx, const float cy, const float cz, int *fcnt,
                                              const bool is_const, int *__restrict__ rows_of, unsigned char *__restrict__ plan,
                                              int *__restrict__ dbg_idx, float *__restrict__ out) {
  int *hdr = reinterpret_cast<int *>(plan + (size_t)obj * kPlanBytes);
  unsigned short *list = reinterpret_cast<unsigned short *>(plan + (size_t)obj * kPlanBytes + 32);
  const int c = lane >> 2, q = lane & 3;
  const int ppl = (n + 3) >> 2;
  unsigned long long mask = 0ull;
  if (c < m)
    for (int t = 0; t < ppl; ++t) {
      const int k = q * ppl + t;
      if (k < n && sq3(cx - sx[k * 3 + 0], cy - sx[k * 3 + 1], cz - sx[k * 3 + 2]) < radius2) mask |= 1ull << k;
    }
  {   // OR over the quad (xor 1, xor 2)
    unsigned lo = (unsigned)mask, hi = (unsigned)(mask >> 32);
    lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, false);
    lo |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    hi |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, false);
    mask = ((unsigned long long)hi << 32) | lo;
  }
  const int cnt = __popcll(mask);
  const int filled = cnt > 0 ? (cnt < kNS ? cnt : kNS) : 1;
  const int fill = cnt > 0 ? __ffsll((long long)mask) - 1 : 0;
  if (q == 0 && c < m) fcnt[c] = filled;
  // this lane's eight slots 8 q .. 8 q + 7 of the centre's row
  int slot[8];
  {
    unsigned long long w = mask;
    for (int t = 0; t < 8 * q; ++t) w &= w - 1;          // drop the 8 q lowest hits
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      slot[t] = w ? __ffsll((long long)w) - 1 : fill;
      w &= w - 1;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (dbg_idx && c < m) {
#pragma unroll
    for (int t = 0; t < 8; ++t) dbg_idx[((size_t)obj * m + c) * kNS + 8 * q + t] = slot[t];
  }
  if (is_const) {                                        // one repeated point: every row is the same row
    if (lane == 0) {
      hdr[0] = 1 | (1 << 16);
      rows_of[obj] = 1;
      list[0] = (unsigned short)slot[0];
    }
    return;
  }
  int off = 0, R = 0;
  for (int k = 0; k < m; ++k) {
    const int f = fcnt[k];
    off += k < c ? f : 0;
    R += f;
  }
  if (c < m) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (8 * q + t < filled) list[off + 8 * q + t] = (unsigned short)((c << 8) | slot[t]);
  }
  if (lane == 0) hdr[0] = rows_of[obj] = R;
  if (R > kTM) {     // more than one chunk: its units may run in different workgroups and meet in `out` by atomic max
    float4 *o = reinterpret_cast<float4 *>(out + (size_t)obj * m * kN3);
    for (int i = lane; i < m * kN3 / 4; i += kWave) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// (`blk`: the workgroup's index among the level-2 planners -- blockIdx.x of sa2_plan_kernel, an offset one in sa12_plan_kernel)
__device__ __forceinline__ void sa2_plan_body(const int blk, const int b, const int n, const int m, const float radius2,
                                              const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                              int *__restrict__ rows_of, unsigned char *__restrict__ plan,
                                              int *__restrict__ dbg_idx, const unsigned char *__restrict__ valid,
                                              const unsigned char *__restrict__ constant, float *__restrict__ out) {
  __shared__ float s_x[4][64 * 3];
  __shared__ int s_f[4][kRowsMaxM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obj = blk * 4 + wave;
  if (obj >= b) return;                                  // (no block barrier below: every wave is on its own)
  if (valid && !valid[obj]) {
    if (lane == 0) *reinterpret_cast<int *>(plan + (size_t)obj * kPlanBytes) = rows_of[obj] = 0;
    return;
  }
  float *sx = s_x[wave];
  int *fcnt = s_f[wave];
  const int c = lane >> 2;
  for (int i = lane; i < n * 3; i += kWave) sx[i] = xyz[(size_t)obj * n * 3 + i];
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (c < m) {
    const float *ct = new_xyz + ((size_t)obj * m + c) * 3;
    cx = ct[0]; cy = ct[1]; cz = ct[2];
  }
  const bool is_const = constant && constant[obj];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  sa2_plan_wave(obj, lane, n, m, radius2, sx, cx, cy, cz, fcnt, is_const, rows_of, plan, dbg_idx, out);
}

__global__ __launch_bounds__(256) void sa2_plan_kernel(const Sa2PlanArgs a) {
  sa2_plan_body(blockIdx.x, a.b, a.n, a.m, a.radius2, a.xyz, a.new_xyz, a.rows_of, a.plan, a.dbg_idx, a.valid, a.constant, a.out);
}

template <int V> struct IntTag { static constexpr int value = V; };

constexpr int kMineMax = 32;                 // units one block can be dealt (units <= kMineMax x blocks)

// The unit of work is a CHUNK: up to 64 consecutive rows of one object's list (an object of R rows: ceil(R / 64) units).
// Units differ 4-fold in work (1 .. 4 row tiles), there are only a few per block (~1,150 on 512 resident blocks at the
// bench shape), and a block reads a unit's plan a unit ahead -- a device-wide queue would be drained by that look-ahead
// before any block knew how long its share takes.  So the deal is STATIC and balanced: units are ordered by row tiles,
// heaviest first (a STABLE counting sort on 4 .. 1 tiles; ranks by ballot / popcount over the b row counts -- every block
// computes the same order for itself, ~1 us, and keeps only its own entries), and block k of B takes positions k,
// 2B-1-k, 2B+k, 4B-1-k, ... of that order.  (Round 5's first form dealt whole OBJECTS: the heaviest, three chunks, set
// the launch's length -- median block 68 k cycles, slowest 98 k.)  An object cut over several blocks meets in `out`
// through an integer atomic max (sa2_plan_kernel zeroes its rows): values are >= 0, the order of arrival changes nothing.
// Returns with s_mine[r] = this block's r-th unit, object | chunk << 20 (-1: none), after a barrier.
__device__ __forceinline__ void deal_units(int b, const int *__restrict__ rows_of, int *s_cnt /* [4][5] */,
                                           int *s_off /* [4][5] */, int *s_mine, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int per_wave = ((b + 255) / 256) * 64, i0 = wave * per_wave, i1 = min(b, i0 + per_wave);
  if (tid <= kMineMax) s_mine[tid] = -1;
  // per object: `full` units of 4 tiles and one LAST unit of 1 .. 4 tiles (none when it has no rows)
  auto units_of = [&](int i, int &full, int &last_tiles) {
    const int R = i < i1 ? rows_of[i] : 0;
    const int nch = (R + kTM - 1) / kTM;
    full = nch > 0 ? nch - 1 : 0;                                   // (0 .. 7: three bits)
    last_tiles = nch > 0 ? (R - full * kTM + 15) >> 4 : 0;
  };
  auto wave_sum = [&](int v) {                                      // sum of a 3-bit value over the wave
    return __popcll(__ballot(v & 1)) + 2 * __popcll(__ballot(v & 2)) + 4 * __popcll(__ballot(v & 4));
  };
  auto wave_before = [&](int v) {                                   // ... over the lanes below this one
    return __popcll(__ballot(v & 1) & lt) + 2 * __popcll(__ballot(v & 2) & lt) + 4 * __popcll(__ballot(v & 4) & lt);
  };
  int cnt[5] = {0, 0, 0, 0, 0};
  for (int base = i0; base < i1; base += 64) {
    int full, lt_;
    units_of(base + lane, full, lt_);
    cnt[4] += wave_sum(full);
#pragma unroll
    for (int c = 1; c <= 4; ++c) cnt[c] += __popcll(__ballot(lt_ == c));
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 1; c <= 4; ++c) s_cnt[wave * 5 + c] = cnt[c];
  }
  __syncthreads();
  if (tid < 4) {                                 // thread w: where wave w's units of each class start
    int run = 0;
    for (int c = 4; c >= 1; --c)
      for (int w = 0; w < 4; ++w) {
        if (w == tid) s_off[tid * 5 + c] = run;
        run += s_cnt[w * 5 + c];
      }
  }
  __syncthreads();
  int run[5];
#pragma unroll
  for (int c = 1; c <= 4; ++c) run[c] = s_off[wave * 5 + c];
  const int B = gridDim.x, k = blockIdx.x;
  auto claim = [&](int pos, int unit) {
    const int r = pos / B, rem = pos - r * B;
    if (((r & 1) ? B - 1 - rem : rem) == k && r < kMineMax) s_mine[r] = unit;
  };
  for (int base = i0; base < i1; base += 64) {
    const int i = base + lane;
    int full, lt_;
    units_of(i, full, lt_);
    // class 4: the object's full chunks first, then (if its last chunk also has 4 tiles) that one
    const int p4 = run[4] + wave_before(full) + __popcll(__ballot(lt_ == 4) & lt);
    for (int c = 0; c < full; ++c) claim(p4 + c, i | (c << 20));
    if (lt_ == 4) claim(p4 + full, i | (full << 20));
    run[4] += wave_sum(full) + __popcll(__ballot(lt_ == 4));
#pragma unroll
    for (int c = 3; c >= 1; --c) {
      const unsigned long long mk = __ballot(lt_ == c);
      if (lt_ == c) claim(run[c] + __popcll(mk & lt), i | (full << 20));
      run[c] += __popcll(mk);
    }
  }
  __syncthreads();
}

constexpr int kSa2RowsLds = 3 * kPlane * 2 + 2 * (kPlanRows * 2 + 64 * 3 * 4 + kRowsMaxM * 4 * 4) + kTM * 4 * 4 + kTM * 4 +
                            kRowsMaxM * kOutPitch * 4;

__global__ __launch_bounds__(256, 2) void sa2_rows_kernel(int b, int n, int m, const int *__restrict__ rows_of,
                                                          const unsigned char *__restrict__ plan,
                                                          const float *__restrict__ xyz, const float *__restrict__ feat,
                                                          const float *__restrict__ new_xyz, LayerS l1, LayerS l2, LayerS l3,
                                                          float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *buf = smem;                                             // ROWS [3][64][144] / FRAG image
  unsigned short *rowmap2 = smem + 3 * kPlane;                             // [2][512]: this unit's object's list and the next one's
  float *sx2 = reinterpret_cast<float *>(rowmap2 + 2 * kPlanRows);         // [2][64 * 3]
  float *ctr2 = sx2 + 2 * 64 * 3;                                          // [2][16 * 4]
  float *dxs = ctr2 + 2 * kRowsMaxM * 4;                                   // [64][4]: the chunk rows' recentred coordinates
  int *rowc = reinterpret_cast<int *>(dxs + kTM * 4);                      // [64]: centre of each row of the chunk
  unsigned *outb = reinterpret_cast<unsigned *>(rowc + kTM);               // [16][257]: the chunk's maxima (bit patterns)
  unsigned char *fbuf = reinterpret_cast<unsigned char *>(buf);
  __shared__ int s_mine[kMineMax + 1], s_cnt[4 * 5], s_off[4 * 5];
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  RSTAMP_DECL;
  deal_units(b, rows_of, s_cnt, s_off, s_mine, tid);
  // (the folded BN affines are read where they are used, from global memory -- 4 KB that every block shares in L2: the
  // table's 4 KB of LDS are what lets TWO blocks share a CU)
  const float *__restrict__ sc1 = l1.scale, *__restrict__ sh1 = l1.shift, *__restrict__ sc2 = l2.scale,
                           *__restrict__ sh2 = l2.shift, *__restrict__ sc3 = l3.scale, *__restrict__ sh3 = l3.shift;
  constexpr int RN1 = kN1 / 64, RN2 = kN2 / 64, RN3 = kN3 / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const WStream w1 = make_stream<RN1>(l1.w, kK0 * kN1 * 6, wave_u, lane);
  const float *__restrict__ wxyz = l1.shift + kN1;                         // [128][4] fp32: layer 1's weights on (dx, dy, dz)
  const WStream w2 = make_stream<RN2>(l2.w, kN1 * kN2 * 6, wave_u, lane);
  const WStream w3 = make_stream<RN3>(l3.w, kN2 * kN3 * 6, wave_u, lane);
  for (int i = tid; i < m * kOutPitch; i += 256) outb[i] = 0u;
  int ui = 0, un = s_mine[1];
  if (s_mine[0] < 0) return;
  int obj = s_mine[0] & 0xfffff, base = (s_mine[0] >> 20) * kTM;

  // ---- an object's plan + geometry: global -> registers -> LDS (parity p) ----
  unsigned plw = 0u;                           // two entries of the row list per thread
  float geo = 0.f;                             // one coordinate of a point / of a centre per thread
  int hdrn = 0;
  auto plan_fetch = [&](int o) {
    const unsigned char *P = plan + (size_t)o * kPlanBytes;
    hdrn = *reinterpret_cast<const int *>(P);
    plw = reinterpret_cast<const unsigned *>(P + 32)[tid];
    if (tid < n * 3) geo = xyz[(size_t)o * n * 3 + tid];
    else if (tid >= 192 && tid < 192 + m * 3) geo = new_xyz[(size_t)o * m * 3 + (tid - 192)];
  };
  auto plan_store = [&](int p) {
    reinterpret_cast<unsigned *>(rowmap2 + p * kPlanRows)[tid] = plw;
    if (tid < n * 3) sx2[p * 192 + tid] = geo;
    else if (tid >= 192 && tid < 192 + m * 3) {
      const int q = tid - 192, c = q / 3;
      ctr2[p * kRowsMaxM * 4 + c * 4 + (q - c * 3)] = geo;
    }
  };
  // ---- a chunk's rows: global -> registers (all loads first), then split -> operand planes, dx and centre per row ----
  constexpr int IT = kTM * 32 / 256;
  float4 val[IT];
  auto rows_fetch = [&](int o, int p, int bs, int R) {            // (slots past the last row repeat it: a duplicate
    const float *F = feat + (size_t)o * n * 128;                   //  does not move a maximum)
    const unsigned short *rm = rowmap2 + p * kPlanRows;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      int rr = bs + (e >> 5);
      rr = rr < R ? rr : R - 1;
      val[it] = *reinterpret_cast<const float4 *>(F + (size_t)(rm[rr] & 255) * 128 + (e & 31) * 4);
    }
  };
  auto rows_store = [&](int p, int bs, int R) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 q[3];
      split4(v, q);
      unsigned short *d = buf + (e >> 5) * kLd + (e & 31) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * kPlane) = q[k];
    }
    if (tid < kTM) {
      int rr = bs + tid;
      rr = rr < R ? rr : R - 1;
      const int rm = rowmap2[p * kPlanRows + rr], pi = rm & 255, c = rm >> 8;
      const float *sx = sx2 + p * 192, *ct = ctr2 + p * kRowsMaxM * 4 + c * 4;
      rowc[tid] = c;
      *reinterpret_cast<float4 *>(dxs + tid * 4) = make_float4(sx[pi * 3 + 0] - ct[0], sx[pi * 3 + 1] - ct[1], sx[pi * 3 + 2] - ct[2], 0.f);
    }
  };

  // ---- prologue: the first unit's plan and rows, unpipelined ----
  plan_fetch(obj);
  int hdr = hdrn, pb = 0;
  plan_store(0);
  __syncthreads();
  rows_fetch(obj, 0, base, hdr & 0xffff);
  rows_store(0, base, hdr & 0xffff);
  __syncthreads();

  while (true) {
    const int R = hdr & 0xffff;
    const int rows = R - base < kTM ? R - base : kTM;
    const int nmt = (rows + 15) >> 4;
    const bool more = un >= 0;
    const int objn = more ? un & 0xfffff : obj, basen = more ? (un >> 20) * kTM : base;
    // per-lane addresses are re-derived every unit (see sa2_split_kernel)
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wave = tid >> 6;
    RSTAMP(4);
    plan_fetch(objn);                          // the NEXT unit's plan flies under layer 1 (no next unit: a harmless re-read)
    auto layers = [&](auto tag) {
      constexpr int MT = decltype(tag)::value;
      // (each variant derives its lane addresses from its own opaque copy of the thread id: expressions common to the
      // four variants would otherwise be hoisted above the switch and stay live through a whole variant -- 71 spills)
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      const int lane = t_ & 63, wave = t_ >> 6;
      WPiece ring1[kRing], ring2[kRing], ring3[kRing];
      preload_ring<RN1, kN1 / 16, kRing>(ring1, w1, lane);
      {
        f32x4 acc[RN1][MT];
        zero_acc(acc);
        const XRowsS x1{buf + (lane & 15) * kLd + 8 * (lane >> 4), kLd, kPlane};
        gemm_split_x<RN1, MT, kK0 / 32, kN1 / 16, kRing>(x1, w1, acc, lane, ring1);
        preload_ring<RN2, kN2 / 16, kRing>(ring2, w2, lane);
        float4 sc[RN1], sh[RN1];
        load_affine4<RN1>(sc1, sh1, wave * RN1 * 16, lane, sc, sh);
        {   // + W_xyz (dx, dy, dz), exactly as sa2_split_kernel
          float4 d[MT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) d[mt] = *reinterpret_cast<const float4 *>(dxs + (16 * mt + (lane & 15)) * 4);
#pragma unroll
          for (int rn = 0; rn < RN1; ++rn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float4 wv = *reinterpret_cast<const float4 *>(wxyz + (wave * RN1 * 16 + rn * 16 + 4 * (lane >> 4) + r) * 4);
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[rn][mt][r] = __builtin_fmaf(wv.z, d[mt].z, __builtin_fmaf(wv.y, d[mt].y, __builtin_fmaf(wv.x, d[mt].x, acc[rn][mt][r])));
            }
        }
        RSTAMP(6);
        __syncthreads();                                          // (A) every wave is done READING the operand
        plan_store(pb ^ 1);                                       // the next unit's plan: registers -> LDS (other parity)
        store_split_frag<RN1, MT>(acc, sc, sh, fbuf, wave * RN1 * 16, lane);
      }
      __syncthreads();                                            // (B)
      RSTAMP(7);
      {
        f32x4 acc[RN2][MT];
        zero_acc(acc);
        const XFragS x2{buf + lane * 8};
        gemm_split_x<RN2, MT, kN1 / 32, kN2 / 16, kRing>(x2, w2, acc, lane, ring2);
        preload_ring<RN3, kN3 / 16, kRing>(ring3, w3, lane);
        float4 sc[RN2], sh[RN2];
        load_affine4<RN2>(sc2, sh2, wave * RN2 * 16, lane, sc, sh);
        RSTAMP(8);
        __syncthreads();                                          // (C)
        store_split_frag<RN2, MT>(acc, sc, sh, fbuf, wave * RN2 * 16, lane);
      }
      __syncthreads();                                            // (D)
      RSTAMP(9);
      {
        f32x4 acc[RN3][MT];
        zero_acc(acc);
        const XFragS x3{buf + lane * 8};
        gemm_split_x<RN3, MT, kN2 / 32, kN3 / 16, kRing>(x3, w3, acc, lane, ring3);
        RSTAMP(10);
        float4 sc[RN3], sh[RN3];
        load_affine4<RN3>(sc3, sh3, wave * RN3 * 16, lane, sc, sh);
        // Segmented maximum.  The chunk's rows are sorted by centre, so inside a 16-row tile (the 16 lanes j of a lane
        // group) every centre is a run of consecutive lanes: an inclusive max-scan over the runs (offsets 1, 2, 4, 8 by
        // DPP row shifts; a lane takes its neighbour's value only when that neighbour belongs to the same centre) leaves
        // each run's maximum in its LAST lane, and only those lanes touch the table -- different centres, different
        // words: no two lanes of an instruction meet on an address.  (One atomic per row and channel instead: the 16 rows
        // of a dense neighbourhood are 16 read-modify-writes of ONE word, served one after the other.)  Values are
        // >= 0 after the ReLU: the unsigned order of their bit patterns is their order, and the bitwise AND with an
        // all-ones / all-zeros word is the select.
        const int j = lane & 15, g = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int c = rowc[16 * mt + j];
          unsigned same[4];
          bool any[4];
#define MSR3D_SHR(K, X, OLD) __builtin_amdgcn_update_dpp((OLD), (X), 0x110 + (K), 0xf, 0xf, false)
          same[0] = MSR3D_SHR(1, c, -1) == c ? 0xffffffffu : 0u;
          same[1] = MSR3D_SHR(2, c, -1) == c ? 0xffffffffu : 0u;
          same[2] = MSR3D_SHR(4, c, -1) == c ? 0xffffffffu : 0u;
          same[3] = MSR3D_SHR(8, c, -1) == c ? 0xffffffffu : 0u;
#pragma unroll
          for (int q = 0; q < 4; ++q) any[q] = __ballot(same[q] != 0u) != 0ull;      // wave-uniform: runs longer than 2^q
          const bool tail = __builtin_amdgcn_update_dpp(-1, c, 0x101, 0xf, 0xf, false) != c;   // row_shl:1 (lane 15: -1)
          unsigned u[RN3][4];
#pragma unroll
          for (int rn = 0; rn < RN3; ++rn) {
            const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
            const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) u[rn][r] = __float_as_uint(fmaxf(__builtin_fmaf(acc[rn][mt][r], s4[r], h4[r]), 0.0f));
          }
#define MSR3D_STEP(Q, K)                                                                               \
          if (any[Q]) {                                                                                \
            _Pragma("unroll") for (int rn = 0; rn < RN3; ++rn)                                         \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
              const unsigned t = (unsigned)MSR3D_SHR(K, (int)u[rn][r], 0) & same[Q];                   \
              u[rn][r] = t > u[rn][r] ? t : u[rn][r];                                                  \
            }                                                                                          \
          }
          MSR3D_STEP(0, 1)
          MSR3D_STEP(1, 2)
          MSR3D_STEP(2, 4)
          MSR3D_STEP(3, 8)
#undef MSR3D_STEP
#undef MSR3D_SHR
          if (tail) {
            unsigned *o = outb + c * kOutPitch + wave * RN3 * 16 + 4 * g;
#pragma unroll
            for (int rn = 0; rn < RN3; ++rn)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                __hip_atomic_fetch_max(o + rn * 16 + r, u[rn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    };
    switch (nmt) {
      case 1: layers(IntTag<1>{}); break;
      case 2: layers(IntTag<2>{}); break;
      case 3: layers(IntTag<3>{}); break;
      default: layers(IntTag<4>{}); break;
    }
    RSTAMP(11);
    // the NEXT unit's rows start moving (UNCONDITIONAL: after the last unit a harmless re-read of it -- a conditional
    // assignment would make `val` loop-carried and live across the three products); its plan sits in the other parity
    const int Rn = hdrn & 0xffff;
    rows_fetch(objn, pb ^ 1, basen, Rn);
    __syncthreads();                                              // (E) the chunk's maxima are in; the operand buffer is free
    RSTAMP(12);
    {   // ---- the chunk's maxima -> out.  An object of ONE chunk: plain stores (a constant object: centre 0's row for every
        // centre); an object cut into several units: integer atomic max onto rows sa2_plan_kernel zeroed ----
      float *O = out + (size_t)obj * m * kN3;
      if ((hdr >> 16) != 0) {
        for (int i = tid; i < m * kN3; i += 256) O[i] = __uint_as_float(outb[i & 255]);
        __syncthreads();
        if (tid < kN3) outb[tid] = 0u;
      } else if (R <= kTM) {
        for (int i = tid; i < m * kN3; i += 256) {
          const int w = (i >> 8) * kOutPitch + (i & 255);
          O[i] = __uint_as_float(outb[w]);
          outb[w] = 0u;
        }
      } else {
        for (int i = tid; i < m * kN3; i += 256) {
          const int w = (i >> 8) * kOutPitch + (i & 255);
          const unsigned v = outb[w];
          if (v) {
            __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(O) + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            outb[w] = 0u;
          }
        }
      }
    }
    RSTAMP(13);
    if (!more) break;
    rows_store(pb ^ 1, basen, Rn);
    obj = objn;
    base = basen;
    hdr = hdrn;
    pb ^= 1;
    un = s_mine[++ui + 1];                                        // (s_mine[kMineMax] = -1)
    __syncthreads();                                              // (F) the next unit's operand is in place
    RSTAMP(5);
  }
}

// =====================================================================================================
// Level 3 (group-all): xyz (b, 16, 3), feat (b, 16, 256) -> MLP 259 -> 256 -> 512 -> 768, max over the 16
// points; out (b, 768).  K order [feat(256), xyz(3), 0 x 29].  A tile is TWO objects (32 rows): layer 3's
// operand (512 wide, three planes) is 101 KB of LDS, so one block per CU, and the level's 3.6 MB of split
// weights stream past every tile -- the kernel is bound by that stream (~40 B/clk/CU out of L2), not by
// the matrix pipe.  Hence wide register panels: each wave owns RN = 4 / 8 / 6+6 column tiles and keeps TWO
// slabs of them (up to 36 KB per wave) in flight, rolled over slab pairs (ping-pong, no register copies).
// =====================================================================================================
constexpr int k3TM = 32, k3K0 = 288, k3N1 = 256, k3N2 = 512, k3N3 = 768;
constexpr int k3Ld = k3N2 + kPadH;           // 528: same bank residue as 144 -- conflict-free b128 reads
constexpr int k3Plane = k3TM * k3Ld;
constexpr int kSa3Lds = 3 * k3Plane * 2 + 2 * (k3N1 + k3N2 + k3N3) * 4;

template <int RN, int MT, int NT>
__device__ __forceinline__ void gemm_split_rolled(const unsigned short *xs, int ldh, int plane, const WStream &wg, int t0,
                                                  int KS, f32x4 (&acc)[RN][MT], int lane, int ws0 = 0) {
  const int j = lane & 15, g = lane >> 4;
  const unsigned short *xp = xs + j * ldh + 8 * g;
  WPiece wa[RN], wb[RN];
  auto fetch = [&](WPiece (&w)[RN], int s) {
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) load_piece<NT>(w[rn], wg, ws0 + s, t0 + rn, lane);     // (ws0: the weights' first slab)
  };
  auto mma = [&](const WPiece (&w)[RN], int s) {
    bf16x8 x[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        x[mt][p] = *reinterpret_cast<const bf16x8 *>(xp + p * plane + mt * 16 * ldh + 32 * s);
#define MSR3D_TERM(PW, PX)                                                                             \
    _Pragma("unroll") for (int rn = 0; rn < RN; ++rn)                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
        acc[rn][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[rn].v[PW], x[mt][PX], acc[rn][mt], 0, 0, 0);
    MSR3D_TERMS_ALL
#undef MSR3D_TERM
  };
  fetch(wa, 0);
#pragma clang loop unroll(disable)      // (unrolled -- it is, at three row tiles -- every slab's pieces are hoisted: 3.8 KB of scratch a lane)
  for (int s = 0; s < KS; s += 2) {
    fetch(wb, s + 1 < KS ? s + 1 : s);          // (odd slab counts: a harmless re-read)
    __builtin_amdgcn_sched_barrier(0);
    mma(wa, s);
    __builtin_amdgcn_sched_barrier(0);
    fetch(wa, s + 2 < KS ? s + 2 : s);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < KS) mma(wb, s + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256) void sa3_split_kernel(int b, const float *__restrict__ xyz, const float *__restrict__ feat,
                                                        LayerS l1, LayerS l2, LayerS l3, float *__restrict__ out,
                                                        const unsigned char *__restrict__ valid) {
  const int obj0 = blockIdx.x * 2;
  if (valid && !valid[obj0] && !(obj0 + 1 < b && valid[obj0 + 1])) return;   // (one valid: the other rides along)
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *buf = smem;                                              // [3][32][528] bf16
  float *aff = reinterpret_cast<float *>(smem + 3 * k3Plane);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int i = tid; i < k3N1; i += 256) { aff[i] = l1.scale[i]; aff[k3N1 + i] = l1.shift[i]; }
  for (int i = tid; i < k3N2; i += 256) { aff[2 * k3N1 + i] = l2.scale[i]; aff[2 * k3N1 + k3N2 + i] = l2.shift[i]; }
  for (int i = tid; i < k3N3; i += 256) { aff[2 * (k3N1 + k3N2) + i] = l3.scale[i]; aff[2 * (k3N1 + k3N2) + k3N3 + i] = l3.shift[i]; }
  const float *sc1 = aff, *sh1 = aff + k3N1, *sc2 = aff + 2 * k3N1, *sh2 = sc2 + k3N2, *sc3 = aff + 2 * (k3N1 + k3N2), *sh3 = sc3 + k3N3;
  const WStream w1 = make_stream<0>(l1.w, k3K0 * k3N1 * 6, 0, lane);
  const WStream w2 = make_stream<0>(l2.w, k3N1 * k3N2 * 6, 0, lane);
  const WStream w3 = make_stream<0>(l3.w, k3N2 * k3N3 * 6, 0, lane);
  {   // operand: 32 rows x [feat(256), x, y, z, 0 ...]; all loads first, then split + LDS stores
    constexpr int IT = k3TM * 64 / 256;
    float4 val[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256, row = e >> 6, obj = obj0 + (row >> 4);
      val[it] = obj < b ? *reinterpret_cast<const float4 *>(feat + ((size_t)obj * 16 + (row & 15)) * 256 + (e & 63) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    if (tid < k3TM) {
      const int obj = obj0 + (tid >> 4);
      if (obj < b) {
        const float *q = xyz + ((size_t)obj * 16 + (tid & 15)) * 3;
        px = q[0]; py = q[1]; pz = q[2];
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 p[3];
      split4(v, p);
      unsigned short *d = buf + (e >> 6) * k3Ld + (e & 63) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * k3Plane) = p[k];
    }
    if (tid < k3TM) {
      const float v[4] = {px, py, pz, 0.f};
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned short *d = buf + k * k3Plane + tid * k3Ld + 256;
        *reinterpret_cast<uint2 *>(d) = p[k];
#pragma unroll
        for (int c = 4; c < 32; c += 4) *reinterpret_cast<uint2 *>(d + c) = make_uint2(0u, 0u);
      }
    }
  }
  __syncthreads();
  constexpr int MT = k3TM / 16;
  {
    constexpr int RN = k3N1 / 64;
    f32x4 acc[RN][MT];
    zero_acc(acc);
    gemm_split_rolled<RN, MT, k3N1 / 16>(buf, k3Ld, k3Plane, w1, wave_u * RN, k3K0 / 32, acc, lane);
    float4 sc[RN], sh[RN];
    load_affine4<RN>(sc1, sh1, wave * RN * 16, lane, sc, sh);
    __syncthreads();                                            // every wave is done reading the operand
    store_split<RN, MT>(acc, sc, sh, buf, k3Ld, k3Plane, wave * RN * 16, lane);
  }
  __syncthreads();
  {
    constexpr int RN = k3N2 / 64;
    f32x4 acc[RN][MT];
    zero_acc(acc);
    gemm_split_rolled<RN, MT, k3N2 / 16>(buf, k3Ld, k3Plane, w2, wave_u * RN, k3N1 / 32, acc, lane);
    float4 sc[RN], sh[RN];
    load_affine4<RN>(sc2, sh2, wave * RN * 16, lane, sc, sh);
    __syncthreads();
    store_split<RN, MT>(acc, sc, sh, buf, k3Ld, k3Plane, wave * RN * 16, lane);
  }
  __syncthreads();
  int groups = b - obj0;
  groups = groups < 2 ? groups : 2;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {                        // the wave's 12 column tiles, six at a time
    constexpr int RN = k3N3 / 128;
    f32x4 acc[RN][MT];
    zero_acc(acc);
    const int t0 = wave_u * 2 * RN + pass * RN;
    gemm_split_rolled<RN, MT, k3N3 / 16>(buf, k3Ld, k3Plane, w3, t0, k3N2 / 32, acc, lane);
    float4 sc[RN], sh[RN];
    load_affine4<RN>(sc3, sh3, t0 * 16, lane, sc, sh);
    float gm[RN][MT][4];
    group_reduce<RN, MT, 1>(acc, sc, sh, gm);
    group_finish<RN, MT>(gm, out + (size_t)obj0 * k3N3, k3N3, t0 * 16, groups, lane);
  }
}

// ---- level 3, FOUR objects per tile (round 4) ----------------------------------------------------------------------
// The two-object tile above streams the level's 3.6 MB of split weights past every 32 rows: 480 tiles = 1.7 GB per launch
// out of L2 (73.8 MB of HBM traffic for 18.7 MB of unique data) in 1.9 rounds of the 256 CUs, matrix pipe 41 % busy.  A
// 64-row tile halves the stream per object and is ONE round (240 tiles) -- but layer 3's operand (512 wide, three
// planes, 64 rows) is 203 KB.  So the 512-wide activation never exists as a whole: layer 2 is accumulated entirely in
// registers (each wave: 4 column tiles of either half), its first half goes to LDS (104 KB), layer 3 runs over that
// K half into ALL of its accumulators (12 column tiles x 4 row tiles per wave = 192 registers), then the second half
// takes the buffer's place and layer 3 finishes.  LDS: one 64 x 304 x 3 bf16 buffer (the 288-wide input; 272 pitch for
// the 256-wide images -- both pitches leave every 16-byte fragment read on its own bank quad) + the affine tables.
constexpr int k4TM = 64, k4LdIn = k3K0 + kPadH, k4Ld = 256 + kPadH;
constexpr int k4PlaneIn = k4TM * k4LdIn, k4Plane = k4TM * k4Ld;
constexpr int kSa3x4Lds = 3 * k4PlaneIn * 2 + 2 * (k3N1 + k3N2 + k3N3) * 4;

__global__ __launch_bounds__(256, 1) void sa3_split4_kernel(int b, const float *__restrict__ xyz, const float *__restrict__ feat,
                                                            LayerS l1, LayerS l2, LayerS l3, float *__restrict__ out,
                                                            const unsigned char *__restrict__ valid) {
  const int obj0 = blockIdx.x * 4;
  if (valid) {                                               // (any valid object: the others ride along)
    bool any = false;
    for (int q = 0; q < 4; ++q) any = any || (obj0 + q < b && valid[obj0 + q]);
    if (!any) return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *buf = smem;
  float *aff = reinterpret_cast<float *>(smem + 3 * k4PlaneIn);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  for (int i = tid; i < k3N1; i += 256) { aff[i] = l1.scale[i]; aff[k3N1 + i] = l1.shift[i]; }
  for (int i = tid; i < k3N2; i += 256) { aff[2 * k3N1 + i] = l2.scale[i]; aff[2 * k3N1 + k3N2 + i] = l2.shift[i]; }
  for (int i = tid; i < k3N3; i += 256) { aff[2 * (k3N1 + k3N2) + i] = l3.scale[i]; aff[2 * (k3N1 + k3N2) + k3N3 + i] = l3.shift[i]; }
  const float *sc1 = aff, *sh1 = aff + k3N1, *sc2 = aff + 2 * k3N1, *sh2 = sc2 + k3N2, *sc3 = aff + 2 * (k3N1 + k3N2), *sh3 = sc3 + k3N3;
  const WStream w1 = make_stream<0>(l1.w, k3K0 * k3N1 * 6, 0, lane);
  const WStream w2 = make_stream<0>(l2.w, k3N1 * k3N2 * 6, 0, lane);
  const WStream w3 = make_stream<0>(l3.w, k3N2 * k3N3 * 6, 0, lane);
  {   // operand: 64 rows x [feat(256), x, y, z, 0 ...]; all loads first, then split + LDS stores
    constexpr int IT = k4TM * 64 / 256;
    float4 val[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256, row = e >> 6, obj = obj0 + (row >> 4);
      val[it] = obj < b ? *reinterpret_cast<const float4 *>(feat + ((size_t)obj * 16 + (row & 15)) * 256 + (e & 63) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    if (tid < k4TM) {
      const int obj = obj0 + (tid >> 4);
      if (obj < b) {
        const float *q = xyz + ((size_t)obj * 16 + (tid & 15)) * 3;
        px = q[0]; py = q[1]; pz = q[2];
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 p[3];
      split4(v, p);
      unsigned short *d = buf + (e >> 6) * k4LdIn + (e & 63) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * k4PlaneIn) = p[k];
    }
    if (tid < k4TM) {
      const float v[4] = {px, py, pz, 0.f};
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned short *d = buf + k * k4PlaneIn + tid * k4LdIn + 256;
        *reinterpret_cast<uint2 *>(d) = p[k];
#pragma unroll
        for (int c = 4; c < 32; c += 4) *reinterpret_cast<uint2 *>(d + c) = make_uint2(0u, 0u);
      }
    }
  }
  __syncthreads();
  constexpr int MT = k4TM / 16;
  {   // layer 1: 288 -> 256, each wave 4 column tiles
    f32x4 acc[4][MT];
    zero_acc(acc);
    gemm_split_rolled<4, MT, k3N1 / 16>(buf, k4LdIn, k4PlaneIn, w1, wave_u * 4, k3K0 / 32, acc, lane);
    float4 sc[4], sh[4];
    load_affine4<4>(sc1, sh1, wave * 64, lane, sc, sh);
    __syncthreads();                                            // every wave is done reading the operand
    store_split<4, MT>(acc, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  // layer 2: 256 -> 512, all of it in registers: tiles 4 w .. 4 w + 3 of the first half, 16 + 4 w .. of the second
  f32x4 acc2b[4][MT];
  {
    f32x4 acc2a[4][MT];
    zero_acc(acc2a);
    gemm_split_rolled<4, MT, k3N2 / 16>(buf, k4Ld, k4Plane, w2, wave_u * 4, k3N1 / 32, acc2a, lane);
    zero_acc(acc2b);
    gemm_split_rolled<4, MT, k3N2 / 16>(buf, k4Ld, k4Plane, w2, 16 + wave_u * 4, k3N1 / 32, acc2b, lane);
    float4 sc[4], sh[4];
    load_affine4<4>(sc2, sh2, wave * 64, lane, sc, sh);
    __syncthreads();                                            // layer 1's image has been read by everyone
    store_split<4, MT>(acc2a, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  // layer 3 over the first K half (weight slabs 0..7), the wave's 12 column tiles as 6 + 6
  f32x4 acc3a[6][MT], acc3b[6][MT];
  zero_acc(acc3a);
  zero_acc(acc3b);
  const int t3 = wave_u * 12;
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3, 8, acc3a, lane, 0);
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3 + 6, 8, acc3b, lane, 0);
  {
    float4 sc[4], sh[4];
    load_affine4<4>(sc2, sh2, 256 + wave * 64, lane, sc, sh);
    __syncthreads();                                            // the first half has been read by everyone
    store_split<4, MT>(acc2b, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3, 8, acc3a, lane, 8);
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3 + 6, 8, acc3b, lane, 8);
  int groups = b - obj0;
  groups = groups < 4 ? groups : 4;
  {
    float4 sc[6], sh[6];
    float gm[6][MT][4];
    load_affine4<6>(sc3, sh3, t3 * 16, lane, sc, sh);
    group_reduce<6, MT, 1>(acc3a, sc, sh, gm);
    group_finish<6, MT>(gm, out + (size_t)obj0 * k3N3, k3N3, t3 * 16, groups, lane);
    load_affine4<6>(sc3, sh3, (t3 + 6) * 16, lane, sc, sh);
    group_reduce<6, MT, 1>(acc3b, sc, sh, gm);
    group_finish<6, MT>(gm, out + (size_t)obj0 * k3N3, k3N3, (t3 + 6) * 16, groups, lane);
  }
}

// ---- level 3 over a LIST of objects: three real objects a tile, the constant ones one row each (round 6) -----------
// A tile's duration is ~11 us + ~19.5 us per 16-row tile of it (two-object tiles 50 us, four-object tiles 89 us): the
// launch is one round of tiles either way, and the weights' stream past a CU costs the same whatever the tile holds.
// A third of the bench's object slots are the dataset's padding cloud (dataset_wrapper.py:156-158): one repeated point,
// sixteen identical rows here.  So when the objects allow it the launch is dealt differently, decided ON THE DEVICE from
// the flags the sampling launch left (msr3d_sa_fps2*_flags) -- no host read-back:
//   every workgroup ranks the objects for itself (ballots of the flags -> two bit masks per 64 objects in LDS, ~1 us):
//   R real objects (valid, not constant), C constant ones; if ceil(R / 3) + ceil(C / 48) <= gridDim.x
//     workgroup k < ceil(R / 3)       three real objects (ranks 3 k ..), 48 rows, group maximum over each object's 16 rows
//     the next ceil(C / 48)           48 constant objects each: ONE row an object (its point 0), no maximum
//   else (dense batches)              four consecutive objects a workgroup, as sa3_split4_kernel.
// A row's arithmetic is the same in every form (same gather, same split, the same MFMA sequence per row; which rows
// share a 16-row tile enters no result) and max over sixteen identical rows is that row: the same bits.
// (Measured and not kept: THREE stages of weight pieces in flight instead of two at three row tiles -- the registers are
// there -- 84-86 us against 78-81: the tile is not short of loads in flight.)
template <int RN, int NG>
__device__ __forceinline__ void group_finish_ids(float (&m)[RN][NG][4], float *__restrict__ out, int ldo, int n0,
                                                 const int *s_obj, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn)
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      row16_max4(m[rn][gq]);
      const int obj = s_obj[gq * 16];
      if (j == 0 && obj >= 0)
        *reinterpret_cast<float4 *>(out + (size_t)obj * ldo + n0 + rn * 16 + 4 * g) =
            make_float4(m[rn][gq][0], m[rn][gq][1], m[rn][gq][2], m[rn][gq][3]);
    }
}
// every row an object of its own: relu(acc * scale + shift) of row 16 mt + j -> out[s_obj[16 mt + j]]
template <int RN, int MT>
__device__ __forceinline__ void rows_finish_ids(const f32x4 (&acc)[RN][MT], const float4 (&sc)[RN], const float4 (&sh)[RN],
                                                float *__restrict__ out, int ldo, int n0, const int *s_obj, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int obj = s_obj[mt * 16 + j];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) {
      const float s4[4] = {sc[rn].x, sc[rn].y, sc[rn].z, sc[rn].w};
      const float h4[4] = {sh[rn].x, sh[rn].y, sh[rn].z, sh[rn].w};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(0.f, __builtin_fmaf(acc[rn][mt][r], s4[r], h4[r]));
      if (obj >= 0) *reinterpret_cast<float4 *>(out + (size_t)obj * ldo + n0 + rn * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// position of the k-th (0-based) set bit of m (k < popcount(m))
__device__ __forceinline__ int kth_set_bit(unsigned long long m, int k) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const unsigned long long low = m & ((1ull << w) - 1ull);
    const int c = __popcll(low);
    if (k >= c) { k -= c; m >>= w; pos += w; } else { m = low; }
  }
  return pos;
}

// One tile of 16 MT rows: row r is point (ONE_ROW ? 0 : r & 15) of object s_obj[r] (-1: an empty row, zeros).  The
// layer chain of sa3_split4_kernel (layer 2 in registers, layer 3 over the two K halves) with MT row tiles.
template <int MT, bool one_row>
__device__ __forceinline__ void sa3_tile(const int *s_obj, unsigned short *buf, const float *aff,
                                                   const float *__restrict__ xyz, const float *__restrict__ feat, const LayerS &l1,
                                                   const LayerS &l2, const LayerS &l3, float *__restrict__ out) {
  constexpr int TM = 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const float *sc1 = aff, *sh1 = aff + k3N1, *sc2 = aff + 2 * k3N1, *sh2 = sc2 + k3N2, *sc3 = aff + 2 * (k3N1 + k3N2), *sh3 = sc3 + k3N3;
  const WStream w1 = make_stream<0>(l1.w, k3K0 * k3N1 * 6, 0, lane);
  const WStream w2 = make_stream<0>(l2.w, k3N1 * k3N2 * 6, 0, lane);
  const WStream w3 = make_stream<0>(l3.w, k3N2 * k3N3 * 6, 0, lane);
  {   // operand: TM rows x [feat(256), x, y, z, 0 ...]; all loads first, then split + LDS stores
    constexpr int IT = TM * 64 / 256;
    float4 val[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256, row = e >> 6, obj = s_obj[row];
      val[it] = obj >= 0 ? *reinterpret_cast<const float4 *>(feat + ((size_t)obj * 16 + (one_row ? 0 : (row & 15))) * 256 + (e & 63) * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    if (tid < TM) {
      const int obj = s_obj[tid];
      if (obj >= 0) {
        const float *q = xyz + ((size_t)obj * 16 + (one_row ? 0 : (tid & 15))) * 3;
        px = q[0]; py = q[1]; pz = q[2];
      }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const float v[4] = {val[it].x, val[it].y, val[it].z, val[it].w};
      uint2 p[3];
      split4(v, p);
      unsigned short *d = buf + (e >> 6) * k4LdIn + (e & 63) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * k4PlaneIn) = p[k];
    }
    if (tid < TM) {
      const float v[4] = {px, py, pz, 0.f};
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned short *d = buf + k * k4PlaneIn + tid * k4LdIn + 256;
        *reinterpret_cast<uint2 *>(d) = p[k];
#pragma unroll
        for (int c = 4; c < 32; c += 4) *reinterpret_cast<uint2 *>(d + c) = make_uint2(0u, 0u);
      }
    }
  }
  __syncthreads();
  {   // layer 1: 288 -> 256, each wave 4 column tiles
    f32x4 acc[4][MT];
    zero_acc(acc);
    gemm_split_rolled<4, MT, k3N1 / 16>(buf, k4LdIn, k4PlaneIn, w1, wave_u * 4, k3K0 / 32, acc, lane);
    float4 sc[4], sh[4];
    load_affine4<4>(sc1, sh1, wave * 64, lane, sc, sh);
    __syncthreads();                                            // every wave is done reading the operand
    store_split<4, MT>(acc, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  // layer 2: 256 -> 512, all of it in registers: tiles 4 w .. 4 w + 3 of the first half, 16 + 4 w .. of the second
  f32x4 acc2b[4][MT];
  {
    f32x4 acc2a[4][MT];
    zero_acc(acc2a);
    gemm_split_rolled<4, MT, k3N2 / 16>(buf, k4Ld, k4Plane, w2, wave_u * 4, k3N1 / 32, acc2a, lane);
    zero_acc(acc2b);
    gemm_split_rolled<4, MT, k3N2 / 16>(buf, k4Ld, k4Plane, w2, 16 + wave_u * 4, k3N1 / 32, acc2b, lane);
    float4 sc[4], sh[4];
    load_affine4<4>(sc2, sh2, wave * 64, lane, sc, sh);
    __syncthreads();                                            // layer 1's image has been read by everyone
    store_split<4, MT>(acc2a, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  // layer 3 over the first K half (weight slabs 0..7), the wave's 12 column tiles as 6 + 6
  f32x4 acc3a[6][MT], acc3b[6][MT];
  zero_acc(acc3a);
  zero_acc(acc3b);
  const int t3 = wave_u * 12;
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3, 8, acc3a, lane, 0);
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3 + 6, 8, acc3b, lane, 0);
  {
    float4 sc[4], sh[4];
    load_affine4<4>(sc2, sh2, 256 + wave * 64, lane, sc, sh);
    __syncthreads();                                            // the first half has been read by everyone
    store_split<4, MT>(acc2b, sc, sh, buf, k4Ld, k4Plane, wave * 64, lane);
  }
  __syncthreads();
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3, 8, acc3a, lane, 8);
  gemm_split_rolled<6, MT, k3N3 / 16>(buf, k4Ld, k4Plane, w3, t3 + 6, 8, acc3b, lane, 8);
  float4 sc[6], sh[6];
  if (one_row) {
    load_affine4<6>(sc3, sh3, t3 * 16, lane, sc, sh);
    rows_finish_ids<6, MT>(acc3a, sc, sh, out, k3N3, t3 * 16, s_obj, lane);
    load_affine4<6>(sc3, sh3, (t3 + 6) * 16, lane, sc, sh);
    rows_finish_ids<6, MT>(acc3b, sc, sh, out, k3N3, (t3 + 6) * 16, s_obj, lane);
  } else {
    float gm[6][MT][4];
    load_affine4<6>(sc3, sh3, t3 * 16, lane, sc, sh);
    group_reduce<6, MT, 1>(acc3a, sc, sh, gm);
    group_finish_ids<6, MT>(gm, out, k3N3, t3 * 16, s_obj, lane);
    load_affine4<6>(sc3, sh3, (t3 + 6) * 16, lane, sc, sh);
    group_reduce<6, MT, 1>(acc3b, sc, sh, gm);
    group_finish_ids<6, MT>(gm, out, k3N3, (t3 + 6) * 16, s_obj, lane);
  }
}

constexpr int kSa3MaxGroups = 64;            // objects are ranked on the device up to 64 x 64 of them

__global__ __launch_bounds__(256, 1) void sa3_tiles_kernel(int b, const float *__restrict__ xyz, const float *__restrict__ feat,
                                                           LayerS l1, LayerS l2, LayerS l3, float *__restrict__ out,
                                                           const unsigned char *__restrict__ valid,
                                                           const unsigned char *__restrict__ constant) {
  __shared__ unsigned long long s_real[kSa3MaxGroups], s_const[kSa3MaxGroups];
  __shared__ int s_obj[64];
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *buf = smem;
  float *aff = reinterpret_cast<float *>(smem + 3 * k4PlaneIn);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int groups = (b + 63) >> 6;
  int tiles_r = 0, tiles_c = 0, R = 0, C = 0;
  bool listed = false;
  if (groups <= kSa3MaxGroups) {
    for (int g = wave; g < groups; g += 4) {
      const int o = g * 64 + lane;
      int cls = 0;
      if (o < b && (!valid || valid[o])) cls = (constant && constant[o]) ? 2 : 1;
      const unsigned long long mr = __builtin_amdgcn_ballot_w64(cls == 1), mc = __builtin_amdgcn_ballot_w64(cls == 2);
      if (lane == 0) { s_real[g] = mr; s_const[g] = mc; }
    }
    __syncthreads();
    for (int g = 0; g < groups; ++g) { R += __popcll(s_real[g]); C += __popcll(s_const[g]); }
    tiles_r = (R + 2) / 3;
    tiles_c = (C + 47) / 48;
    listed = tiles_r + tiles_c <= (int)gridDim.x;
  }
  const int k = blockIdx.x;
  bool one_row = false;
  if (listed) {
    if (k >= tiles_r + tiles_c) return;
    one_row = k >= tiles_r;
    if (tid < 48) {
      const int want = one_row ? 48 * (k - tiles_r) + tid : 3 * k + tid / 16;
      const unsigned long long *m = one_row ? s_const : s_real;
      int id = -1;
      if (want < (one_row ? C : R)) {
        int cum = 0;
        for (int g = 0; g < groups; ++g) {
          const int c = __popcll(m[g]);
          if (want < cum + c) { id = g * 64 + kth_set_bit(m[g], want - cum); break; }
          cum += c;
        }
      }
      s_obj[tid] = id;
    }
  } else {
    const int obj0 = k * 4;
    if (obj0 >= b) return;
    if (valid) {                                             // (any valid object: the others ride along)
      bool any = false;
      for (int q = 0; q < 4; ++q) any = any || (obj0 + q < b && valid[obj0 + q]);
      if (!any) return;
    }
    if (tid < 64) s_obj[tid] = obj0 + (tid >> 4) < b ? obj0 + (tid >> 4) : -1;
  }
  for (int i = tid; i < k3N1; i += 256) { aff[i] = l1.scale[i]; aff[k3N1 + i] = l1.shift[i]; }
  for (int i = tid; i < k3N2; i += 256) { aff[2 * k3N1 + i] = l2.scale[i]; aff[2 * k3N1 + k3N2 + i] = l2.shift[i]; }
  for (int i = tid; i < k3N3; i += 256) { aff[2 * (k3N1 + k3N2) + i] = l3.scale[i]; aff[2 * (k3N1 + k3N2) + k3N3 + i] = l3.shift[i]; }
  __syncthreads();
  if (!listed) sa3_tile<4, false>(s_obj, buf, aff, xyz, feat, l1, l2, l3, out);
  else if (one_row) sa3_tile<3, true>(s_obj, buf, aff, xyz, feat, l1, l2, l3, out);
  else sa3_tile<3, false>(s_obj, buf, aff, xyz, feat, l1, l2, l3, out);
}

// =====================================================================================================
// Level 1: pts (b, n, 6) rows [xyz, rgb]; centres (b, m, 3); ball_idx (b, m, 32) from the ball-query
// launch.  MLP 6 -> 64 -> 64 -> 128, max over the 32 neighbours; out (b, m, 128).
//
// The layers are NARROW (13 k MAC per row against level 2's 70 k): with level 2's layout -- waves split
// the channels, block barriers around every in-place epilogue -- a tile would be six barriers around
// 170 MFMAs per wave.  Here the roles are turned: all of the level's split weights (86 KB) sit in LDS for
// the life of a persistent block, and every WAVE owns one whole neighbourhood (32 rows) and ALL channels,
// so the layer chain is wave-local.  And it never leaves the registers: with D = W X^T a lane (j, g) ends a
// layer holding, for row j, channels 16 t + 4 g + r -- eight values per 32-channel slab (tiles 2s, 2s+1) --
// which is exactly a lane's share of the next layer's X fragment if that layer's K axis is numbered
// accordingly.  The host packs layers 2 and 3 with that K permutation inside each slab (position (g, e) <->
// channel 32 s + 16 (e >> 2) + 4 g + (e & 3)), so an epilogue is affine + ReLU + split + two register
// packs: no LDS writes, no operand reads, no barrier anywhere in the loop.  LDS carries only the weight
// fragments (1 KB contiguous reads, conflict-free, a group ahead of the MFMAs that use them).
// The gather is two dependent global round trips (neighbour index -> point row); both are issued a round
// ahead (the index at the top of the previous round, the point row after its layer 1).
// =====================================================================================================
constexpr int k1K0 = 32, k1N1 = 64, k1N2 = 64, k1N3 = 128;
constexpr int k1W1 = (k1K0 / 32) * (k1N1 / 16) * 3 * kFragS, k1W2 = (k1N1 / 32) * (k1N2 / 16) * 3 * kFragS,
              k1W3 = (k1N2 / 32) * (k1N3 / 16) * 3 * kFragS;             // bf16 counts: 6144, 12288, 24576
constexpr int k1Waves = 8;
#ifndef MSR3D_SA1_ROWS_WAVES
#define MSR3D_SA1_ROWS_WAVES 8
#endif
constexpr int k1RowsWaves = MSR3D_SA1_ROWS_WAVES;     // waves per workgroup of sa1_rows_kernel (226 registers a lane: 12 waves
                                                      // = 3 per SIMD spill 59 of them, 103 us against 88; 16 spill 119)
constexpr int k1Chunk = 3;                   // rounds per queue fetch (15 per block at the bench shape)
constexpr int kSa1Lds = (k1W1 + k1W2 + k1W3) * 2 + 2 * (k1N1 + k1N2 + k1N3) * 4;

// one layer of a wave's 32-row neighbourhood: acc[t][mt] = W tile t (LDS, fragment order [s][t][3][64][8])
// x X^T (registers).  Pieces are taken TWO at a time (x 2 row tiles = four independent accumulators per
// product term) and the next two fly under these 24 MFMAs.
// SWAP: the operands change roles (D^T = X W^T): the same registers, the same products summed over k in the same order
// -- the same bits (tests/test_sa_rows_gpu.py compares with the unswapped kernel) -- but a lane (j, g) then holds rows
// 4 g + r of CHANNEL j instead of channels 4 g + r of row j: the maximum over a tile's 16 rows is three in-lane
// maxima and two exchanges between the four lane groups instead of a 16-lane DPP reduction per value.
template <int NT, int KS, bool SWAP = false>
__device__ __forceinline__ void wave_layer(const unsigned short *wl, const bf16x8 (&x)[KS][2][3], f32x4 (&acc)[NT][2], int lane) {
  static_assert(NT % 2 == 0, "column tiles in pairs");
  constexpr int NG = KS * NT / 2;
  const unsigned short *wp = wl + lane * 8;
  WPiece w[NG][2];                 // fully unrolled; two groups live
  auto fetch = [&](int grp) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        w[grp][i].v[p] = *reinterpret_cast<const bf16x8 *>(wp + ((grp * 2 + i) * 3 + p) * kFragS);
  };
  fetch(0);
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int grp = 0; grp < NG; ++grp) {
    const int q0 = grp * 2, s = q0 / NT, t0 = q0 % NT;
    __builtin_amdgcn_sched_barrier(0);
    if (grp + 1 < NG) fetch(grp + 1);
    __builtin_amdgcn_sched_barrier(0);
#define MSR3D_TERM(PW, PX)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
        acc[t0 + i][mt] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[s][mt][PX], w[grp][i].v[PW], acc[t0 + i][mt], 0, 0, 0) \
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[grp][i].v[PW], x[s][mt][PX], acc[t0 + i][mt], 0, 0, 0);
    MSR3D_TERMS_ALL
#undef MSR3D_TERM
  }
}

// relu(acc * scale + shift), split: the lane's accumulators of tiles 2s, 2s+1 ARE its share of slab s of
// the next layer's operand (K numbered by the host accordingly)
template <int NT>
__device__ __forceinline__ void wave_next(const f32x4 (&acc)[NT][2], const float *sc, const float *sh,
                                          bf16x8 (&x)[NT / 2][2][3], int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 s4 = *reinterpret_cast<const float4 *>(sc + t * 16 + 4 * g);
    const float4 h4 = *reinterpret_cast<const float4 *>(sh + t * 16 + 4 * g);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float v[4] = {fmaxf(__builtin_fmaf(acc[t][mt][0], s4.x, h4.x), 0.f), fmaxf(__builtin_fmaf(acc[t][mt][1], s4.y, h4.y), 0.f),
                          fmaxf(__builtin_fmaf(acc[t][mt][2], s4.z, h4.z), 0.f), fmaxf(__builtin_fmaf(acc[t][mt][3], s4.w, h4.w), 0.f)};
      uint2 p[3];
      split4(v, p);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned *d = reinterpret_cast<unsigned *>(&x[t >> 1][mt][k]) + (t & 1) * 2;
        d[0] = p[k].x;
        d[1] = p[k].y;
      }
    }
  }
}

__global__ __launch_bounds__(64 * k1Waves, 1) void sa1_split_kernel(int n, int m, int centres, int rounds, int *__restrict__ queue,
                                                                   const float *__restrict__ pts,
                                                                   const float *__restrict__ new_xyz,
                                                                   const int *__restrict__ ball_idx, LayerS l1, LayerS l2,
                                                                   LayerS l3, float *__restrict__ out,
                                                                   const unsigned char *__restrict__ valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *wl1 = smem, *wl2 = wl1 + k1W1, *wl3 = wl2 + k1W2;
  float *aff = reinterpret_cast<float *>(wl3 + k1W3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // Rounds (eight neighbourhoods, one per wave) are handed out per BLOCK in chunks of k1Chunk from a device-
  // wide queue (queue[0]: next round, queue[1]: blocks finished; the last block to leave resets both) -- see
  // the level-2 kernel for why not a static split.  Thread 0 walks the queue one chunk AHEAD (the chunk after
  // the one being processed is always known, so the gather prefetch can cross a chunk boundary) and the
  // waves, otherwise independent, meet at one barrier per chunk to take the next pair over from LDS.
  // (The same queue per WAVE -- 2,048 waves x 5 returning device-scope atomics on one address -- cost +60 us.)
  __shared__ int s_chunk[4];                 // [0..1]: the first two chunks; [2..3]: hand-over slots, alternating
  auto fetch_chunk = [&]() { return __hip_atomic_fetch_add(queue, k1Chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto leave = [&]() {
    const int done = __hip_atomic_fetch_add(queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (tid == 0) {
    const int c0 = fetch_chunk();
    s_chunk[0] = c0;
    s_chunk[1] = c0 < rounds ? fetch_chunk() : rounds;
  }
  {   // the level's weights and affines: once per block
    const uint4 *s1 = reinterpret_cast<const uint4 *>(l1.w), *s2 = reinterpret_cast<const uint4 *>(l2.w),
                *s3 = reinterpret_cast<const uint4 *>(l3.w);
    uint4 *d = reinterpret_cast<uint4 *>(smem);
    for (int i = tid; i < k1W1 / 8; i += 64 * k1Waves) d[i] = s1[i];
    for (int i = tid; i < k1W2 / 8; i += 64 * k1Waves) d[k1W1 / 8 + i] = s2[i];
    for (int i = tid; i < k1W3 / 8; i += 64 * k1Waves) d[(k1W1 + k1W2) / 8 + i] = s3[i];
    for (int i = tid; i < k1N1; i += 64 * k1Waves) { aff[i] = l1.scale[i]; aff[k1N1 + i] = l1.shift[i]; }
    for (int i = tid; i < k1N2; i += 64 * k1Waves) { aff[2 * k1N1 + i] = l2.scale[i]; aff[2 * k1N1 + k1N2 + i] = l2.shift[i]; }
    for (int i = tid; i < k1N3; i += 64 * k1Waves) { aff[2 * (k1N1 + k1N2) + i] = l3.scale[i]; aff[2 * (k1N1 + k1N2) + k1N3 + i] = l3.shift[i]; }
  }
  const float *sc1 = aff, *sh1 = aff + k1N1, *sc2 = aff + 2 * k1N1, *sh2 = sc2 + k1N2, *sc3 = aff + 2 * (k1N1 + k1N2), *sh3 = sc3 + k1N3;
  __syncthreads();

  // lane (j, g = 0) carries rows j and 16 + j of the wave's neighbourhood through the gather
  auto centre_of = [&](int rr) { return min(rr * k1Waves + wave, centres - 1); };   // (a ragged last round repeats the last centre)
  int pi[2];
  float2 pa[2], pb[2], pc[2];
  float cx, cy, cz;
  auto fetch_idx = [&](int cg) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) pi[mt] = min(max(ball_idx[(size_t)cg * kNS + mt * 16 + j], 0), n - 1);
  };   // (clamped: an object the valid mask skips has no indices -- any row will do)
  auto fetch_pts = [&](int cg) {
    const int obj = cg / m;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float2 *q = reinterpret_cast<const float2 *>(pts + ((size_t)obj * n + pi[mt]) * 6);
      pa[mt] = q[0]; pb[mt] = q[1]; pc[mt] = q[2];
    }
    const float *ct = new_xyz + (size_t)cg * 3;
    cx = ct[0]; cy = ct[1]; cz = ct[2];
  };
  int cur = s_chunk[0], nxt = s_chunk[1];               // (published before the barrier above)
  if (cur >= rounds) {
    if (tid == 0) leave();
    return;
  }
  int r = cur, r_end = min(cur + k1Chunk, rounds), par = 0;
  int pending = rounds;                                   // thread 0: the chunk after `nxt`, fetched a chunk early
  if (tid == 0 && nxt < rounds) pending = fetch_chunk();
  fetch_idx(centre_of(r));
  fetch_pts(centre_of(r));
  while (true) {
    const int cg = centre_of(r);
    const bool live = !(valid && !valid[cg / m]);        // wave-uniform
    const bool last_of_chunk = r + 1 >= r_end;
    const bool more = !last_of_chunk || nxt < rounds;
    const int cgn = centre_of(!more ? r : (last_of_chunk ? nxt : r + 1));
    // ---- layer-1 operand straight into registers: slab 0, k = 8 g + e; only g = 0 is non-zero ----
    bf16x8 x1[1][2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float v0[4] = {pa[mt].x - cx, pa[mt].y - cy, pb[mt].x - cz, pb[mt].y};
      const float v1[4] = {pc[mt].x, pc[mt].y, 0.f, 0.f};
      uint2 p0[3], p1[3];
      split4(v0, p0);
      split4(v1, p1);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const uint4 q = g == 0 ? make_uint4(p0[k].x, p0[k].y, p1[k].x, p1[k].y) : make_uint4(0u, 0u, 0u, 0u);
        x1[0][mt][k] = *reinterpret_cast<const bf16x8 *>(&q);
      }
    }
    fetch_idx(cgn);                                      // next round's neighbours fly under layer 1
    bf16x8 x2[k1N1 / 32][2][3], x3[k1N2 / 32][2][3];
    if (live) {
      f32x4 acc[k1N1 / 16][2];
      wave_layer<k1N1 / 16, k1K0 / 32>(wl1, x1, acc, lane);
      wave_next<k1N1 / 16>(acc, sc1, sh1, x2, lane);
    }
    fetch_pts(cgn);                                      // next round's point rows fly under layers 2 and 3
    if (live) {
      {
        f32x4 acc[k1N2 / 16][2];
        wave_layer<k1N2 / 16, k1N1 / 32>(wl2, x2, acc, lane);
        wave_next<k1N2 / 16>(acc, sc2, sh2, x3, lane);
      }
      f32x4 acc[k1N3 / 16][2];
      wave_layer<k1N3 / 16, k1N2 / 32>(wl3, x3, acc, lane);
      // relu(affine), max over the 32 rows: the two row tiles per lane, then the 16 lanes of a row
#pragma unroll
      for (int t = 0; t < k1N3 / 16; ++t) {
        const float4 s4 = *reinterpret_cast<const float4 *>(sc3 + t * 16 + 4 * g);
        const float4 h4 = *reinterpret_cast<const float4 *>(sh3 + t * 16 + 4 * g);
        float mx[4] = {0.f, 0.f, 0.f, 0.f};              // starting the max at 0 IS the ReLU
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          mx[0] = fmaxf(mx[0], __builtin_fmaf(acc[t][mt][0], s4.x, h4.x));
          mx[1] = fmaxf(mx[1], __builtin_fmaf(acc[t][mt][1], s4.y, h4.y));
          mx[2] = fmaxf(mx[2], __builtin_fmaf(acc[t][mt][2], s4.z, h4.z));
          mx[3] = fmaxf(mx[3], __builtin_fmaf(acc[t][mt][3], s4.w, h4.w));
        }
        row16_max4(mx);
        if (j == 0) *reinterpret_cast<float4 *>(out + (size_t)cg * k1N3 + t * 16 + 4 * g) = make_float4(mx[0], mx[1], mx[2], mx[3]);
      }
    }
    if (!more) break;
    if (!last_of_chunk) { ++r; continue; }
    // chunk boundary: everyone moves on to `nxt`; thread 0 hands over the chunk after it
    if (tid == 0) s_chunk[2 + par] = pending;
    __syncthreads();
    cur = nxt;
    nxt = s_chunk[2 + par];
    par ^= 1;                                              // (the other slot is rewritten a whole chunk later)
    r = cur;
    r_end = min(cur + k1Chunk, rounds);
    if (tid == 0) pending = nxt < rounds ? fetch_chunk() : rounds;
  }
  if (tid == 0) leave();
}

// =====================================================================================================
// Level 1 over DISTINCT neighbourhood rows (round 5).  Same kernel as sa1_split_kernel -- the level's weights in LDS, a
// wave owns 32 rows and all channels, the layer chain never leaves the registers -- but the 32 rows of a wave are a TASK:
//   BIG    a centre with more than 16 different neighbours: its 32 row slots, as before;
//   PAIR   TWO centres with at most 16 different neighbours each (slots 16..31 of such a centre's ball-query row repeat
//          its first hit, ball_query_gpu.cu:35-39): the first 16 slots of one in row tile 0, of the other in row tile 1,
//          and the two maxima taken per row tile (a centre without a partner is paired with itself);
//   CONST  an object whose cloud is one repeated point (msr3d_sa_fps2*_flags): every row of every centre is the same
//          row; one PAIR task of its centre 0, the result written to all m centres.
// Every row's arithmetic is unchanged and max is idempotent: the level's output is the same bits.  On the benchmark's
// scenes 40 % of the real centres are small and a third of the objects are padding: 30,720 wave tasks become ~16,700.
// sa1_plan_kernel (16 lanes ... one wave per object) classifies the centres (slot 16 == slot 0 <=> at most 16 different
// neighbours), pairs the small ones in index order and writes, per task, a header word and its 32 point indices into the
// caller's workspace; tasks of different objects are appended through ONE atomic per workgroup of 4 objects (the order of
// the list is not reproducible; no result depends on it).
// =====================================================================================================
constexpr int kTaskBig = 0, kTaskPair = 1, kTaskConst = 2;      // header: kind | obj << 2 | cA << 20 | cB << 26

// The tasks of one object, its tasks' first index `base` known: lane = centre (lane < m, the object live); `src`: the
// centre's 32 indices (global or LDS).
__device__ __forceinline__ void sa1_emit_tasks(const int obj, const int lane, const bool is_const, const bool small,
                                               const unsigned long long bigs, const unsigned long long smalls, const int nbig,
                                               const int base, const int4 *src, int *__restrict__ thdr, int *__restrict__ trow) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (is_const) {
    if (lane == 0) {
      thdr[base] = kTaskConst | (obj << 2);
      int4 *d = reinterpret_cast<int4 *>(trow + (size_t)base * kNS);
      for (int q = 0; q < 4; ++q) d[q] = d[4 + q] = src[q];
    }
    return;
  }
  if (!small) {
    const int t = base + __popcll(bigs & lt);
    thdr[t] = kTaskBig | (obj << 2) | (lane << 20) | (lane << 26);
    int4 *d = reinterpret_cast<int4 *>(trow + (size_t)t * kNS);
    for (int q = 0; q < 8; ++q) d[q] = src[q];
  } else {
    const int r = __popcll(smalls & lt), t = base + nbig + (r >> 1);
    int4 *d = reinterpret_cast<int4 *>(trow + (size_t)t * kNS) + 4 * (r & 1);
    for (int q = 0; q < 4; ++q) d[q] = src[q];
    if (!(r & 1)) {                                      // first of its pair: the header (and both halves when alone)
      const unsigned long long above = smalls & ~lt & ~(1ull << lane);
      const int partner = above ? __ffsll((long long)above) - 1 : lane;
      thdr[t] = kTaskPair | (obj << 2) | (lane << 20) | (partner << 26);
      if (!above)
        for (int q = 0; q < 4; ++q) d[4 + q] = src[q];
    }
  }
}

__device__ __forceinline__ void sa1_plan_body(const int blk, const int b, const int m, const int *__restrict__ ball_idx,
                                              int *__restrict__ total, int *__restrict__ thdr, int *__restrict__ trow,
                                              const unsigned char *__restrict__ valid,
                                              const unsigned char *__restrict__ constant) {
  __shared__ int s_n[4], s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obj = blk * 4 + wave;
  const bool live = obj < b && !(valid && !valid[obj]);
  const bool is_const = live && constant && constant[obj];
  const int *row = ball_idx + ((size_t)(live ? obj : 0) * m + (lane < m ? lane : 0)) * kNS;
  bool small = false;
  if (live && lane < m) small = row[16] == row[0];
  const unsigned long long in = (live && lane < m && !is_const) ? ~0ull : 0ull;
  const unsigned long long bigs = __ballot(in && !small), smalls = __ballot(in && small);
  const int nbig = __popcll(bigs), nsmall = __popcll(smalls);
  const int ntask = !live ? 0 : (is_const ? 1 : nbig + ((nsmall + 1) >> 1));
  if (lane == 0) s_n[wave] = ntask;
  __syncthreads();
  if (threadIdx.x == 0) s_base = atomicAdd(total, s_n[0] + s_n[1] + s_n[2] + s_n[3]);
  __syncthreads();
  if (!live || lane >= m) return;
  int base = s_base;
  for (int w = 0; w < wave; ++w) base += s_n[w];
  sa1_emit_tasks(obj, lane, is_const, small, bigs, smalls, nbig, base, reinterpret_cast<const int4 *>(row), thdr, trow);
}

struct Sa1PlanArgs {
  int b, m;
  const int *ball_idx;
  int *total, *thdr, *trow;
  const unsigned char *valid, *constant;
};

__global__ __launch_bounds__(256) void sa1_plan_kernel(const Sa1PlanArgs a) {
  sa1_plan_body(blockIdx.x, a.b, a.m, a.ball_idx, a.total, a.thdr, a.trow, a.valid, a.constant);
}

// Both planners in ONE launch (msr3d_sa_plan12): each needs only what the sampling launch wrote -- level 1 the ball
// rows, level 2 the two sets of centres -- and a launch on this part costs ~5 us before it does anything; workgroups
// [0, nb1) plan level 1, the rest level 2.
__global__ __launch_bounds__(256) void sa12_plan_kernel(const Sa1PlanArgs a1, const Sa2PlanArgs a2, const int nb1) {
  if ((int)blockIdx.x < nb1)
    sa1_plan_body(blockIdx.x, a1.b, a1.m, a1.ball_idx, a1.total, a1.thdr, a1.trow, a1.valid, a1.constant);
  else
    sa2_plan_body(blockIdx.x - nb1, a2.b, a2.n, a2.m, a2.radius2, a2.xyz, a2.new_xyz, a2.rows_of, a2.plan, a2.dbg_idx, a2.valid,
                  a2.constant, a2.out);
}

// Both plans INSIDE the sampling launch (msr3d_sa_fps2_query_plan, round 6): what the planners read is in that
// launch's LDS when it ends -- the object's 32 ball rows, its two sets of centres -- and its waves have time: level 1's
// plan is run by the last query wave to finish, beside the FPS wave's second level (a dependent chain of one wave); level
// 2's by the FPS wave itself when that chain ends.  One atomic an object appends its tasks to the list (its order is not
// reproducible; no result depends on it).  The 9 us launch of sa12_plan_kernel is gone; the sampling launch is ~1 us longer.
struct FpsPlan12 {
  static constexpr bool kOn = true;
  static constexpr int kLdsInts = kRowsMaxM;             // sa2_plan_wave's fcnt
  int *total, *thdr, *trow;                              // level 1 (Sa1PlanArgs)
  float radius2_l2;                                      // level 2 (Sa2PlanArgs)
  int *rows_of;
  unsigned char *plan;
  int *dbg_idx2;
  float *out2;
  __device__ void skipped(int obj, int lane) const {
    if (lane == 0) *reinterpret_cast<int *>(plan + (size_t)obj * kPlanBytes) = rows_of[obj] = 0;
  }
  __device__ void after_queries(int obj, int lane, int m, int nsample, const int *rows, bool is_const) const {
    const int *row = rows + (size_t)(lane < m ? lane : 0) * nsample;
    const bool small = lane < m && row[16] == row[0];
    const unsigned long long in = (lane < m && !is_const) ? ~0ull : 0ull;
    const unsigned long long bigs = __ballot(in && !small), smalls = __ballot(in && small);
    const int nbig = __popcll(bigs), nsmall = __popcll(smalls);
    const int ntask = is_const ? 1 : nbig + ((nsmall + 1) >> 1);
    int base = 0;
    if (lane == 0) base = atomicAdd(total, ntask);
    base = __builtin_amdgcn_readfirstlane(base);
    if (lane >= m) return;
    sa1_emit_tasks(obj, lane, is_const, small, bigs, smalls, nbig, base, reinterpret_cast<const int4 *>(row), thdr, trow);
  }
  __device__ void after_sampling(int obj, int lane, int m, int m2, const float *keep, const float *keep2, int *scratch,
                                 bool is_const) const {
    const int c = lane >> 2;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (c < m2) { cx = keep2[c * 3 + 0]; cy = keep2[c * 3 + 1]; cz = keep2[c * 3 + 2]; }
    sa2_plan_wave(obj, lane, m, m2, radius2_l2, keep, cx, cy, cz, scratch, is_const, rows_of, plan, dbg_idx2, out2);
  }
};

template <int PS>
__global__ __launch_bounds__(kWave * 4) void fps_query_plan_kernel(
    int n, int ps_arg, int m, int bs, int log2bs, int q, const float *__restrict__ pts, int *__restrict__ idxs,
    float *__restrict__ new_xyz, int m2, int bs2, int log2bs2, int q2, int *__restrict__ idxs2, float *__restrict__ new_xyz2,
    const unsigned char *__restrict__ valid, float radius2, int nsample, int *__restrict__ ball_idx,
    unsigned char *__restrict__ constant_out, const FpsPlan12 plan) {
  extern __shared__ __attribute__((aligned(16))) char fq_smem[];
  fps_query_body<16, 3, PS>(n, ps_arg, m, bs, log2bs, q, pts, idxs, new_xyz, m2, bs2, log2bs2, q2, idxs2, new_xyz2, valid, radius2,
                            nsample, ball_idx, constant_out, fq_smem, plan);
}

__global__ __launch_bounds__(64 * k1RowsWaves, 1) void sa1_rows_kernel(int n, int m, int *__restrict__ queue,
                                                                  const float *__restrict__ pts,
                                                                  const float *__restrict__ new_xyz,
                                                                  const int *__restrict__ thdr, const int *__restrict__ trow,
                                                                  LayerS l1, LayerS l2, LayerS l3, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short *wl1 = smem, *wl2 = wl1 + k1W1, *wl3 = wl2 + k1W2;
  float *aff = reinterpret_cast<float *>(wl3 + k1W3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // Tasks cost the same (32 rows each), so they are dealt STATICALLY: wave w of block k takes tasks 8 k + w, + 8 B, ...
  // -- no queue, no atomics and no barrier in the loop: every wave runs on its own.  (sa1_split_kernel's queue of
  // three-round chunks, fetched two chunks ahead, is drained by that look-ahead once a block's share is only ~8 rounds:
  // the stamps showed blocks with 6 and with 9 rounds and 11k cycles per chunk spent at the chunk barrier.)
  // queue[2] holds the number of tasks sa1_plan_kernel appended, queue[1] counts the blocks that are done: the last
  // one resets both for the next launch.
  const int tasks = __hip_atomic_load(queue + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  auto leave = [&]() {
    const int done = __hip_atomic_fetch_add(queue + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  {   // the level's weights and affines: once per block
    const uint4 *s1 = reinterpret_cast<const uint4 *>(l1.w), *s2 = reinterpret_cast<const uint4 *>(l2.w),
                *s3 = reinterpret_cast<const uint4 *>(l3.w);
    uint4 *d = reinterpret_cast<uint4 *>(smem);
    for (int i = tid; i < k1W1 / 8; i += 64 * k1RowsWaves) d[i] = s1[i];
    for (int i = tid; i < k1W2 / 8; i += 64 * k1RowsWaves) d[k1W1 / 8 + i] = s2[i];
    for (int i = tid; i < k1W3 / 8; i += 64 * k1RowsWaves) d[(k1W1 + k1W2) / 8 + i] = s3[i];
    for (int i = tid; i < k1N1; i += 64 * k1RowsWaves) { aff[i] = l1.scale[i]; aff[k1N1 + i] = l1.shift[i]; }
    for (int i = tid; i < k1N2; i += 64 * k1RowsWaves) { aff[2 * k1N1 + i] = l2.scale[i]; aff[2 * k1N1 + k1N2 + i] = l2.shift[i]; }
    for (int i = tid; i < k1N3; i += 64 * k1RowsWaves) { aff[2 * (k1N1 + k1N2) + i] = l3.scale[i]; aff[2 * (k1N1 + k1N2) + k1N3 + i] = l3.shift[i]; }
  }
  const float *sc1 = aff, *sh1 = aff + k1N1, *sc2 = aff + 2 * k1N1, *sh2 = sc2 + k1N2, *sc3 = aff + 2 * (k1N1 + k1N2), *sh3 = sc3 + k1N3;
  __syncthreads();

  int pi[2], hw = 0, hwn = 0;                  // this lane's two point indices; header of the current / the fetched task
  float2 pa[2], pb[2], pc[2];
  float cx[2], cy[2], cz[2];
  auto fetch_idx = [&](int t) {
    hwn = thdr[t];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) pi[mt] = min(max(trow[(size_t)t * kNS + mt * 16 + j], 0), n - 1);
  };
  auto fetch_pts = [&](int h) {
    const int obj = (h >> 2) & 0x3ffff, kind = h & 3, ca = (h >> 20) & 63, cb = kind == kTaskBig ? ca : (h >> 26) & 63;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float2 *q = reinterpret_cast<const float2 *>(pts + ((size_t)obj * n + pi[mt]) * 6);
      pa[mt] = q[0]; pb[mt] = q[1]; pc[mt] = q[2];
      const float *ct = new_xyz + ((size_t)obj * m + (mt ? cb : ca)) * 3;
      cx[mt] = ct[0]; cy[mt] = ct[1]; cz[mt] = ct[2];
    }
  };
  const int stride = gridDim.x * k1RowsWaves;
  int t = blockIdx.x * k1RowsWaves + wave;
  RSTAMP_DECL;
  if (t < tasks) {
  fetch_idx(t);
  fetch_pts(hwn);
  while (true) {
    RSTAMP(0);
    hw = hwn;
    const bool more = t + stride < tasks;
    const int tn = more ? t + stride : t;                // (no next task: a harmless re-read)
    bf16x8 x1[1][2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float v0[4] = {pa[mt].x - cx[mt], pa[mt].y - cy[mt], pb[mt].x - cz[mt], pb[mt].y};
      const float v1[4] = {pc[mt].x, pc[mt].y, 0.f, 0.f};
      uint2 p0[3], p1[3];
      split4(v0, p0);
      split4(v1, p1);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const uint4 q = g == 0 ? make_uint4(p0[k].x, p0[k].y, p1[k].x, p1[k].y) : make_uint4(0u, 0u, 0u, 0u);
        x1[0][mt][k] = *reinterpret_cast<const bf16x8 *>(&q);
      }
    }
    fetch_idx(tn);                                       // next round's header and neighbours fly under layer 1
    bf16x8 x2[k1N1 / 32][2][3], x3[k1N2 / 32][2][3];
    {
      f32x4 acc[k1N1 / 16][2];
      wave_layer<k1N1 / 16, k1K0 / 32>(wl1, x1, acc, lane);
      wave_next<k1N1 / 16>(acc, sc1, sh1, x2, lane);
    }
    RSTAMP(1);
    fetch_pts(hwn);                                      // next round's point rows fly under layers 2 and 3
    RSTAMP(2);
    {
      {
        f32x4 acc[k1N2 / 16][2];
        wave_layer<k1N2 / 16, k1N1 / 32>(wl2, x2, acc, lane);
        wave_next<k1N2 / 16>(acc, sc2, sh2, x3, lane);
      }
      f32x4 acc[k1N3 / 16][2];
      wave_layer<k1N3 / 16, k1N2 / 32, true>(wl3, x3, acc, lane);        // lane (j, g): rows 4 g + r of channel 16 t + j
      RSTAMP(3);
      const int hu = __builtin_amdgcn_readfirstlane(hw);
      const int obj = (hu >> 2) & 0x3ffff, kind = hu & 3, ca = (hu >> 20) & 63, cb = (hu >> 26) & 63;
      float *oa = out + ((size_t)obj * m + ca) * k1N3 + j, *ob = out + ((size_t)obj * m + cb) * k1N3 + j;
      const int x16 = (lane ^ 16) * 4, x32 = (lane ^ 32) * 4;
      constexpr int NT3 = k1N3 / 16;
      // per lane: the maximum over its four rows (and, for a big centre, over both row tiles) of every channel tile ...
      float m0[NT3], m1[NT3];
#pragma unroll
      for (int t = 0; t < NT3; ++t) {
        const float sc = sc3[t * 16 + j], sh = sh3[t * 16 + j];
        m0[t] = m1[t] = 0.f;                               // starting the max at 0 IS the ReLU
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          m0[t] = fmaxf(m0[t], __builtin_fmaf(acc[t][0][r], sc, sh));
          m1[t] = fmaxf(m1[t], __builtin_fmaf(acc[t][1][r], sc, sh));
        }
        if (kind == kTaskBig) m0[t] = fmaxf(m0[t], m1[t]);
      }
      // ... then over the four lane groups: two exchanges per value, all of a stage in flight together (every lane ends
      // with the maximum)
      auto exchange = [&](float (&v)[NT3], int addr) {
        int got[NT3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) got[t] = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(v[t]));
#pragma unroll
        for (int t = 0; t < NT3; ++t) v[t] = fmaxf(v[t], __int_as_float(got[t]));
      };
      exchange(m0, x16);
      if (kind != kTaskBig) exchange(m1, x16);
      exchange(m0, x32);
      if (kind != kTaskBig) exchange(m1, x32);
      if (kind == kTaskConst) {                            // constant cloud: every centre of the object
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
          float *o0 = out + (size_t)obj * m * k1N3 + t * 16 + j;
          for (int c = g; c < m; c += 4) o0[(size_t)c * k1N3] = m0[t];
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
          if (g == 0) oa[t * 16] = m0[t];
          if (kind == kTaskPair && g == 1) ob[t * 16] = m1[t];
        }
      }
    }
    RSTAMP(4);
    if (!more) break;
    t = tn;
  }
  }
  __syncthreads();                                        // every wave of the block is done
  if (tid == 0) leave();
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

// CUs the persistent kernels leave to others (msr3d_set_reserved_cus): their grids are sized to the chip, so
// a kernel that holds CUs beside them -- RCCL's all-reduce in the data-parallel step -- would push the last
// blocks of a launch into a second round.
std::atomic<int> g_reserved_cus{0};
inline int usable_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
      c = 256;
    cus = c;
  }
  const int r = g_reserved_cus.load(std::memory_order_relaxed);
  return cus - r > 8 ? cus - r : 8;
}

// Work queues of the persistent kernels: four ints per (device, stream, level), zero between launches: the
// last block of a launch resets them.  (The default stream's handle is the same on every device, hence the
// device in the key.)  Cleared in stream order on first use and again whenever the previous launch through
// the queue did not report success -- a launch that never ran to its last block would otherwise leave the
// counters non-zero and every later launch would silently skip tiles.
struct WorkQueue {
  int *q = nullptr;
  bool suspect = true;
  bool planned = false;             // msr3d_sa_plan12 appended a task list that no msr3d_sa_level1_rows has consumed yet
};
inline WorkQueue *work_queue(hipStream_t st, int level, hipError_t *err) {
  static std::mutex mu;
  static std::map<std::tuple<int, hipStream_t, int>, WorkQueue> queues;
  int dev = 0;
  if ((*err = hipGetDevice(&dev)) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  WorkQueue &w = queues[{dev, st, level}];
  if (!w.q && (*err = hipMalloc(&w.q, 4 * sizeof(int))) != hipSuccess) return nullptr;
  if (w.suspect) {
    // (the caller's streams do not synchronise with the null stream, where a plain hipMemset would run)
    if ((*err = hipMemsetAsync(w.q, 0, 4 * sizeof(int), st)) != hipSuccess) return nullptr;
  }
  w.suspect = true;                 // until the launch it is handed to reports success
  *err = hipSuccess;
  return &w;
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
  WorkQueue *wq = nullptr;
  const float r2 = radius * radius;   // f32 product, as ball_query_gpu.cu:22
  hipError_t e;
  if (level == 2) {
    if (!pts || !feat || !new_xyz || n <= 0 || n > 64 || m <= 0) return MSR3D_EINVAL;
    if ((e = allow_lds(sa2_split_kernel, kSa2Lds)) != hipSuccess) return (int)e;
    const int slots = 2 * usable_cus();        // resident blocks: two per CU
    const long long tiles = (long long)b * ((m + 1) / 2);
    if (tiles > 0x7fffffffLL - kChunk) return MSR3D_EINVAL;
    const long long chunks = (tiles + kChunk - 1) / kChunk;
    const int blocks = (int)(chunks < slots ? chunks : slots);
    wq = work_queue(st, 2, &e);
    if (!wq) return (int)e;
    int *queue = wq->q;
    sa2_split_kernel<<<blocks, 256, kSa2Lds, st>>>(n, m, (int)tiles, queue, r2, pts, feat, new_xyz, make_layer(w1, affine1, 128),
                                                   make_layer(w2, affine2, 128), make_layer(w3, affine3, 256), out, dbg_ball_idx, valid);
  } else if (level == 1) {
    // pts (b, n, 6); `feat` unused; dbg_ball_idx is this level's WORKSPACE (b, m, 32), filled by the query launch
    if (!pts || !new_xyz || !dbg_ball_idx || n <= 0 || m <= 0) return MSR3D_EINVAL;
    // (radius <= 0: ball_idx already holds the neighbour lists -- msr3d_sa_fps2_query wrote them beside the FPS)
    if (radius > 0.f && (e = launch_ball_query(b, n, 6, m, r2, kNS, new_xyz, pts, dbg_ball_idx, st, valid)) != hipSuccess)
      return (int)e;
    if ((e = allow_lds(sa1_split_kernel, kSa1Lds)) != hipSuccess) return (int)e;
    const int cus = usable_cus();
    const long long rounds = ((long long)b * m + k1Waves - 1) / k1Waves;
    if (rounds > 0x7fffffffLL) return MSR3D_EINVAL;
    const long long chunks = (rounds + k1Chunk - 1) / k1Chunk;
    const int blocks = (int)(chunks < cus ? chunks : cus);
    wq = work_queue(st, 1, &e);
    if (!wq) return (int)e;
    int *queue = wq->q;
    sa1_split_kernel<<<blocks, 64 * k1Waves, kSa1Lds, st>>>(n, m, b * m, (int)rounds, queue, pts, new_xyz, dbg_ball_idx,
                                                           make_layer(w1, affine1, 64), make_layer(w2, affine2, 64),
                                                           make_layer(w3, affine3, 128), out, valid);
  } else if (level == 3) {
    if (!pts || !feat || n != 16 || m != 1) return MSR3D_EINVAL;
#if MSR3D_SPLIT_TERMS == 3
    // (the four-object tile's register allocation -- 256 VGPRs + 200 AGPRs with six products in flight per piece -- falls
    //  into scratch with three: the reduced variant takes the two-object tile)
    static const bool two = true;
#else
    static const bool two = [] { const char *v = getenv("MSR3D_SA3_TILE"); return v && v[0] == '2'; }();
#endif
    if (two) {
      if ((e = allow_lds(sa3_split_kernel, kSa3Lds)) != hipSuccess) return (int)e;
      sa3_split_kernel<<<(b + 1) / 2, 256, kSa3Lds, st>>>(b, pts, feat, make_layer(w1, affine1, 256), make_layer(w2, affine2, 512),
                                                         make_layer(w3, affine3, 768), out, valid);
    } else {
      if ((e = allow_lds(sa3_split4_kernel, kSa3x4Lds)) != hipSuccess) return (int)e;
      sa3_split4_kernel<<<(b + 3) / 4, 256, kSa3x4Lds, st>>>(b, pts, feat, make_layer(w1, affine1, 256),
                                                            make_layer(w2, affine2, 512), make_layer(w3, affine3, 768), out, valid);
    }
  } else {
    return MSR3D_EINVAL;
  }
  e = hipGetLastError();
  if (wq && e == hipSuccess) wq->suspect = false;
  return (int)e;
}

// Level 3 over the objects' flags (see sa3_tiles_kernel): valid (b) / constant (b) may be NULL.
extern "C" int msr3d_sa_level3_tiles(int b, const float *xyz, const float *feat, const void *w1, const float *affine1,
                                     const void *w2, const float *affine2, const void *w3, const float *affine3, float *out, const unsigned char *valid,
                                     const unsigned char *constant, msr3d_stream_t stream) {
#if MSR3D_SPLIT_TERMS == 3
  return MSR3D_EINVAL;        // (the reduced variant keeps its two-object tile: msr3d_sa_level_split(3, ...))
#else
  if (b < 0) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!xyz || !feat || !w1 || !affine1 || !w2 || !affine2 || !w3 || !affine3 || !out) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if ((e = allow_lds(sa3_tiles_kernel, kSa3x4Lds)) != hipSuccess) return (int)e;
  const int cus = usable_cus();
  const int four = (b + 3) / 4, listed = (b + 2) / 3 + (b + 47) / 48;
  const int grid = four > (listed < cus ? listed : cus) ? four : (listed < cus ? listed : cus);
  sa3_tiles_kernel<<<grid, 256, kSa3x4Lds, st>>>(b, xyz, feat, make_layer(w1, affine1, 256), make_layer(w2, affine2, 512),
                                                make_layer(w3, affine3, 768), out, valid, constant);
  return (int)hipGetLastError();
#endif
}

inline size_t plan_costs_bytes(int b) { return ((size_t)b * sizeof(int) + 15) & ~(size_t)15; }
extern "C" size_t msr3d_sa_level2_rows_ws_bytes(int b) { return b > 0 ? plan_costs_bytes(b) + (size_t)b * kPlanBytes : 0; }

static bool sa2_plan_args(int b, int n, int m, float radius, const float *xyz, const float *new_xyz, float *out,
                          int *dbg_ball_idx, const unsigned char *valid, const unsigned char *constant, void *plan_ws,
                          Sa2PlanArgs *a) {
  if (!out || !xyz || !new_xyz || !plan_ws) return false;
  if (n <= 0 || n > 64 || m <= 0 || m > kRowsMaxM || (reinterpret_cast<uintptr_t>(plan_ws) & 15u)) return false;
  a->b = b; a->n = n; a->m = m;
  a->radius2 = radius * radius;       // f32 product, as ball_query_gpu.cu:22
  a->xyz = xyz; a->new_xyz = new_xyz;
  a->rows_of = reinterpret_cast<int *>(plan_ws);
  a->plan = reinterpret_cast<unsigned char *>(plan_ws) + plan_costs_bytes(b);
  a->dbg_idx = dbg_ball_idx; a->valid = valid; a->constant = constant; a->out = out;
  return true;
}

extern "C" int msr3d_sa_level2_rows(int b, int n, int m, float radius, const float *xyz, const float *feat,
                                    const float *new_xyz, const void *w1, const float *affine1, const void *w2,
                                    const float *affine2, const void *w3, const float *affine3, float *out,
                                    int *dbg_ball_idx, const unsigned char *valid, const unsigned char *constant,
                                    void *plan_ws, int planned, msr3d_stream_t stream) {
  if (b < 0) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!w1 || !w2 || !w3 || !affine1 || !affine2 || !affine3 || !feat) return MSR3D_EINVAL;
  Sa2PlanArgs pa;
  if (!sa2_plan_args(b, n, m, radius, xyz, new_xyz, out, dbg_ball_idx, valid, constant, plan_ws, &pa)) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  int *rows_of = pa.rows_of;
  unsigned char *plan = pa.plan;
  if (!planned) {                     // (planned: msr3d_sa_plan12 wrote plan_ws on this stream)
    sa2_plan_kernel<<<(b + 3) / 4, 256, 0, st>>>(pa);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  if ((e = allow_lds(sa2_rows_kernel, kSa2RowsLds)) != hipSuccess) return (int)e;
  static const int per_cu = [] { const char *v = getenv("MSR3D_SA2_ROWS_BLOCKS"); return v ? atoi(v) : 2; }();
  const int slots = per_cu * usable_cus();
  int blocks = b < slots ? b : slots;
  const long long max_units = (long long)b * (kPlanRows / kTM);                  // (an object: at most 8 chunks)
  if ((long long)blocks * kMineMax < max_units) blocks = (int)((max_units + kMineMax - 1) / kMineMax);   // (more blocks than resident slots)
  sa2_rows_kernel<<<blocks, 256, kSa2RowsLds, st>>>(b, n, m, rows_of, plan, xyz, feat, new_xyz, make_layer(w1, affine1, 128),
                                                    make_layer(w2, affine2, 128), make_layer(w3, affine3, 256), out);
  return (int)hipGetLastError();
}

extern "C" size_t msr3d_sa_level1_rows_ws_bytes(int b, int m) {
  return (b > 0 && m > 0) ? (size_t)b * m * (sizeof(int) + kNS * sizeof(int)) + 16 : 0;
}

static bool sa1_plan_args(int b, int m, const int *ball_idx, const unsigned char *valid, const unsigned char *constant,
                          void *task_ws, WorkQueue *wq, Sa1PlanArgs *a) {
  if (!ball_idx || !task_ws || m <= 0 || m > 64 || b >= (1 << 18) || (reinterpret_cast<uintptr_t>(task_ws) & 15u)) return false;
  a->b = b; a->m = m; a->ball_idx = ball_idx;
  a->total = wq->q + 2;
  a->trow = reinterpret_cast<int *>(task_ws);                            // [b m][32], 16-byte aligned rows
  a->thdr = a->trow + (size_t)b * m * kNS;
  a->valid = valid; a->constant = constant;
  return true;
}

// The level-1 task list and the level-2 row lists of one batch in ONE launch, for a caller that then passes planned = 1
// to msr3d_sa_level1_rows / msr3d_sa_level2_rows on the same stream (same arguments as those take).
extern "C" int msr3d_sa_plan12(int b, int m1, const int *ball_idx1, void *task_ws1, int n2, int m2, float radius2,
                               const float *xyz2, const float *new_xyz2, float *out2, int *dbg_ball_idx2, void *plan_ws2,
                               const unsigned char *valid, const unsigned char *constant, msr3d_stream_t stream) {
  if (b < 0) return MSR3D_EINVAL;
  if (b == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  WorkQueue *wq = work_queue(st, 5, &e);
  if (!wq) return (int)e;
  Sa1PlanArgs a1;
  Sa2PlanArgs a2;
  if (!sa1_plan_args(b, m1, ball_idx1, valid, constant, task_ws1, wq, &a1)) return MSR3D_EINVAL;
  if (!sa2_plan_args(b, n2, m2, radius2, xyz2, new_xyz2, out2, dbg_ball_idx2, valid, constant, plan_ws2, &a2)) return MSR3D_EINVAL;
  if (wq->planned) {                  // a plan nobody consumed: its task count is still in the queue
    if ((e = hipMemsetAsync(wq->q, 0, 4 * sizeof(int), st)) != hipSuccess) return (int)e;
    wq->planned = false;
  }
  const int nb = (b + 3) / 4;
  sa12_plan_kernel<<<2 * nb, 256, 0, st>>>(a1, a2, nb);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  wq->suspect = false;
  wq->planned = true;
  return 0;
}

// msr3d_sa_fps2_query_flags AND msr3d_sa_plan12 as one launch (see FpsPlan12): the caller then passes planned = 1 to
// msr3d_sa_level1_rows / msr3d_sa_level2_rows on the same stream.  MSR3D_EINVAL for a shape the fused sampling kernel or
// the planners do not take (the caller then makes the two calls).
extern "C" int msr3d_sa_fps2_query_plan(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                                        float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid,
                                        float radius1, int nsample1, int *ball_idx1, unsigned char *constant_out,
                                        void *task_ws1, float radius_l2, float *out2, int *dbg_ball_idx2, void *plan_ws2,
                                        msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m1 <= 0 || m2 <= 0 || point_stride < 3 || !(radius1 > 0.f)) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!pts || !new_xyz1 || !new_xyz2 || !ball_idx1 || !constant_out) return MSR3D_EINVAL;
  // the planners' shapes: 32 slots a row, a lane a centre, <= 16 centres of <= 64 points at level 2; rows read 16 bytes at a time
  if (nsample1 != kNS || m1 > 64 || m2 > kRowsMaxM || m2 > m1 || ((long long)n * point_stride) % 4) return MSR3D_EINVAL;
  const FpsShape s = fps_shape(n);
  const size_t cloud = (size_t)n * point_stride * sizeof(float);
  if (s.slots > 1024 || s.slots <= 256 || cloud > 48 * 1024) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  WorkQueue *wq = work_queue(st, 5, &e);
  if (!wq) return (int)e;
  Sa1PlanArgs a1;
  Sa2PlanArgs a2;
  if (!sa1_plan_args(b, m1, ball_idx1, valid, constant_out, task_ws1, wq, &a1)) return MSR3D_EINVAL;
  if (!sa2_plan_args(b, m1, m2, radius_l2, new_xyz1, new_xyz2, out2, dbg_ball_idx2, valid, constant_out, plan_ws2, &a2))
    return MSR3D_EINVAL;
  if (wq->planned) {                  // a plan nobody consumed: its task count is still in the queue
    if ((e = hipMemsetAsync(wq->q, 0, 4 * sizeof(int), st)) != hipSuccess) return (int)e;
    wq->planned = false;
  }
  FpsPlan12 plan;
  plan.total = a1.total; plan.thdr = a1.thdr; plan.trow = a1.trow;
  plan.radius2_l2 = a2.radius2; plan.rows_of = a2.rows_of; plan.plan = a2.plan; plan.dbg_idx2 = a2.dbg_idx; plan.out2 = a2.out;
  const FpsShape s2 = fps_shape(m1);
  const size_t lds = fps_query_lds(n, point_stride, m1, nsample1, 3, true, FpsPlan12::kLdsInts);
#define MSR3D_FQP(PS)                                                                                                  \
  fps_query_plan_kernel<PS><<<b, kWave * 4, lds, st>>>(s.n, point_stride, m1, s.bs, s.log2bs, s.q, pts, idx1, new_xyz1, m2, \
                                                      s2.bs, s2.log2bs, s2.q, idx2, new_xyz2, valid, radius1 * radius1,  \
                                                      nsample1, ball_idx1, constant_out, plan)
  if (point_stride == 6) MSR3D_FQP(6);
  else if (point_stride == 3) MSR3D_FQP(3);
  else MSR3D_FQP(0);
#undef MSR3D_FQP
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  wq->suspect = false;
  wq->planned = true;
  return 0;
}

extern "C" int msr3d_sa_level1_rows(int b, int n, int m, const float *pts, const float *new_xyz, const int *ball_idx,
                                    const void *w1, const float *affine1, const void *w2, const float *affine2,
                                    const void *w3, const float *affine3, float *out, const unsigned char *valid,
                                    const unsigned char *constant, void *task_ws, int planned, msr3d_stream_t stream) {
  if (b < 0) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!w1 || !w2 || !w3 || !affine1 || !affine2 || !affine3 || !out || !pts || !new_xyz || n <= 0) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  WorkQueue *wq = work_queue(st, 5, &e);
  if (!wq) return (int)e;
  Sa1PlanArgs pa;
  if (!sa1_plan_args(b, m, ball_idx, valid, constant, task_ws, wq, &pa)) return MSR3D_EINVAL;
  int *trow = pa.trow, *thdr = pa.thdr;
  if (planned) {                      // msr3d_sa_plan12 appended the tasks on this stream
    if (!wq->planned) return MSR3D_EINVAL;
  } else {
    if (wq->planned && (e = hipMemsetAsync(wq->q, 0, 4 * sizeof(int), st)) != hipSuccess) return (int)e;   // (a stale plan)
    sa1_plan_kernel<<<(b + 3) / 4, 256, 0, st>>>(pa);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  wq->planned = false;
  if ((e = allow_lds(sa1_rows_kernel, kSa1Lds)) != hipSuccess) return (int)e;
  const int cus = usable_cus();
  const long long max_rounds = ((long long)b * m + k1RowsWaves - 1) / k1RowsWaves;
  const int blocks = (int)(max_rounds < cus ? max_rounds : cus);
  sa1_rows_kernel<<<blocks, 64 * k1RowsWaves, kSa1Lds, st>>>(n, m, wq->q, pts, new_xyz, thdr, trow, make_layer(w1, affine1, 64),
                                                        make_layer(w2, affine2, 64), make_layer(w3, affine3, 128), out);
  e = hipGetLastError();
  if (e == hipSuccess) wq->suspect = false;
  return (int)e;
}

extern "C" int msr3d_set_reserved_cus(int n) {
  if (n < 0 || n > 128) return MSR3D_EINVAL;
  g_reserved_cus.store(n, std::memory_order_relaxed);
  return 0;
}

#if MSR3D_SPLIT_TERMS == 3
// libmsr3d_hip_split2.so is this file alone: it carries the version of the header it was compiled against, so that
// msr3d_amd/_lib.py::load_split2 can refuse a stale build (the main library exports this from pn2_ops.hip)
extern "C" int msr3d_abi_version(void) { return MSR3D_ABI_VERSION; }
#endif
