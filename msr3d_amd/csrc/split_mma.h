// split_mma.h -- fp32-accurate matrix products on the bf16 matrix pipe, shared by the trainable part's
// kernels (scene_block.hip, wgrad_split.hip).  Same arithmetic as sa_split.hip:
//
//     x = x0 + x1 + x2 (+e),  |e| <= 2^-27 |x|     (three bf16 terms, round-to-nearest-even residuals)
//     x w ~ x0 w0 + x0 w1 + x1 w0 + x1 w1 + x0 w2 + x2 w0            (dropped: ~2^-26 |x w|)
//
// six v_mfma_f32_16x16x32_bf16 products per 16 x 16 x 32 block, each exact in the fp32 accumulator and
// summed there smallest first.  An operand matrix Op[rows][k] lives in one of two forms:
//
//   PACK      global memory, pre-split by msr3d_split_pack, [k/32 slabs][rows/16 tiles][3 planes][64 lanes][8]
//             bf16: the lane-th 16 bytes of a (slab, tile, plane) block are what lane `lane` feeds the matrix
//             pipe (lane (j, g): row 16 tile + j, k = 32 slab + 8 g .. + 7), so a wave's fetch is ONE
//             coalesced 1 KB read, streamed through a buffer descriptor and a register ring;
//   FRAG      LDS, the same block order ([slab][tile][plane][64][8]): what an MFMA epilogue can write
//             with one 8-byte store per plane (a lane of D = W X^T holds four consecutive channels of one row);
//   ROWS      LDS, three row-major planes [3][rows][pitch]: what a row-local prologue writes (16 lanes
//             per row, 8 contiguous bytes each: conflict-free stores); pitch = 544 B (K = 256) puts the
//             sixteen 16-byte reads of every ds_read_b128 lane group on distinct bank quads.
#pragma once
#include <hip/hip_runtime.h>

namespace msr3d {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
typedef float sm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sm_bf16x2 __attribute__((ext_vector_type(2)));
typedef int sm_i32x4 __attribute__((ext_vector_type(4)));

constexpr int kPieceBytes = 3 * 1024;       // one (slab, tile): three planes of 64 lanes x 16 B

// MSR3D_TRAIN_PLANES = 1 (python -m msr3d_amd.build builds libmsr3d_hip_bf16.so from scene_block.hip and wgrad_split.hip
// with it): the LABELLED reduced variant MSR3D_TRAIN_MMA=bf16 of the trainable part -- every operand is its FIRST plane
// only (= the value rounded to bf16, nearest even) and every product ONE v_mfma_f32_16x16x32_bf16 instead of six, fp32
// accumulate; the attention core's QK^T / PV / backward products take bf16 operands as well (attn_core.h's bf16 operand
// map).  Memory layouts are unchanged (the planes nobody reads are still written by their producers), so the two
// libraries work on the same buffers.  Not fp32 accuracy: tests/test_train_bf16_gpu.py states the tolerance.
#ifndef MSR3D_TRAIN_PLANES
#define MSR3D_TRAIN_PLANES 3
#endif
#if MSR3D_TRAIN_PLANES != 3 && MSR3D_TRAIN_PLANES != 1
#error "MSR3D_TRAIN_PLANES must be 3 or 1"
#endif
constexpr int kPlanes = MSR3D_TRAIN_PLANES;  // planes of an operand that are fetched and multiplied

// ---- exact three-way split -----------------------------------------------------------------------------
__device__ __forceinline__ unsigned sm_pk_bf16(float a, float b) {       // v_cvt_pk_bf16_f32 (RNE)
  const sm_f32x2 v = {a, b};
  const sm_bf16x2 r = __builtin_convertvector(v, sm_bf16x2);
  return *reinterpret_cast<const unsigned *>(&r);
}
// four values -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void sm_split4(const float (&v)[4], uint2 (&p)[3]) {
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned lo = sm_pk_bf16(a0, a1), hi = sm_pk_bf16(a2, a3);
    p[k] = make_uint2(lo, hi);
    if (k < 2) {
      a0 -= __uint_as_float(lo << 16); a1 -= __uint_as_float(lo & 0xffff0000u);
      a2 -= __uint_as_float(hi << 16); a3 -= __uint_as_float(hi & 0xffff0000u);
    }
  }
}
// eight values -> three planes of eight bf16 (16 bytes each)
__device__ __forceinline__ void sm_split8(const float (&v)[8], uint4 (&p)[3]) {
  float a[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = v[e];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = sm_pk_bf16(a[2 * e], a[2 * e + 1]);
    p[k] = make_uint4(w[0], w[1], w[2], w[3]);
    if (k < 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] -= __uint_as_float(w[e] << 16);
        a[2 * e + 1] -= __uint_as_float(w[e] & 0xffff0000u);
      }
    }
  }
}
// one value -> three bf16 (as the low 16 bits of three words)
__device__ __forceinline__ void sm_split1(float v, unsigned short (&p)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned w = sm_pk_bf16(v, 0.f);
    p[k] = (unsigned short)(w & 0xffffu);
    v -= __uint_as_float(w << 16);
  }
}

// ---- the PACK stream --------------------------------------------------------------------------------------
struct WPiece { bf16x8 v[3]; };

struct WStream {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;                        // lane * 16
  int soff;                        // the wave's first piece, bytes (wave-uniform)
  int nt;                          // column tiles per slab of the packed operand
};
// `first_tile`, `first_slab`: wave-uniform (readfirstlane them where the compiler cannot prove it)
__device__ __forceinline__ WStream make_wstream(const unsigned short *pack, unsigned bytes, int nt, int first_slab,
                                                int first_tile, int lane) {
  WStream st;
  st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(pack), 0, bytes, 0x00020000);
  st.voff = lane * 16;
  st.soff = (first_slab * nt + first_tile) * kPieceBytes;
  st.nt = nt;
  return st;
}
__device__ __forceinline__ void load_wpiece(WPiece &f, const WStream &st, int s, int rn) {
  const int piece = st.soff + (s * st.nt + rn) * kPieceBytes;
#pragma unroll
  for (int p = 0; p < kPlanes; ++p) {
    const sm_i32x4 r = __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, st.voff, piece + p * 1024, 0);
    f.v[p] = *reinterpret_cast<const bf16x8 *>(&r);
  }
}
template <int RN, int D>
__device__ __forceinline__ void preload_wring(WPiece (&ring)[D], const WStream &st) {
#pragma unroll
  for (int q = 0; q < D; ++q) load_wpiece(ring[q], st, q / RN, q % RN);
}

// ---- X fragment sources in LDS ------------------------------------------------------------------------------
// ROWS: planes [3][rows][pitch] (bf16 units); fragment (row tile mt, slab s, plane p) of lane (j, g)
struct XRows {
  const unsigned short *base;      // already offset to this lane: + j * pitch + 8 g
  int pitch, plane;                // bf16 units
  __device__ __forceinline__ bf16x8 operator()(int mt, int s, int p) const {
    return *reinterpret_cast<const bf16x8 *>(base + p * plane + mt * 16 * pitch + 32 * s);
  }
};
__device__ __forceinline__ XRows make_xrows(const unsigned short *planes, int pitch, int rows, int lane) {
  XRows x;
  x.base = planes + (lane & 15) * pitch + 8 * (lane >> 4);
  x.pitch = pitch;
  x.plane = rows * pitch;
  return x;
}
// FRAG: [slab][MTT tiles][3][64][8]
template <int MTT>
struct XFrag {
  const unsigned short *base;      // + lane * 8
  __device__ __forceinline__ bf16x8 operator()(int mt, int s, int p) const {
    return *reinterpret_cast<const bf16x8 *>(base + ((s * MTT + mt) * 3 + p) * 512);
  }
};
// byte offset, inside a FRAG buffer of MTT row tiles, of the 8-byte group (row tile mt, row j, columns
// c .. c + 3, c % 4 == 0) of plane p
template <int MTT>
__device__ __forceinline__ int frag_off4(int mt, int j, int c, int p) {
  const int s = c >> 5, kk = c & 31;
  return (((s * MTT + mt) * 3 + p) * 64 + j + 16 * (kk >> 3)) * 16 + (kk & 7) * 2;
}

// ---- the product --------------------------------------------------------------------------------------------
// acc[rn][mt] (+)= Op[tile rn] . X[tile mt]^T over KS slabs of 32.  W_FIRST: D = W X^T (lane (j, g) holds
// operand rows 4 g + r of token j); otherwise D = X W^T (lane (j, g) holds tokens 4 g + r of operand row j).
// `ring` holds pieces 0..D-1 on entry.  MT0: first row tile of X this wave works on.
template <bool W_FIRST, int RN, int MT, int KS, int D, typename XF>
__device__ __forceinline__ void gemm_split3(const XF &xf, int mt0, const WStream &wg, f32x4 (&acc)[RN][MT],
                                            const WPiece (&ring)[D]) {
  constexpr int NP = KS * RN;
  static_assert(D <= NP, "ring deeper than the product");
  WPiece w[NP];                    // fully unrolled: only a window of D + 1 pieces is ever live
#pragma unroll
  for (int q = 0; q < D; ++q) w[q] = ring[q];
  bf16x8 x[MT][3];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int s = q / RN, rn = q % RN;
    __builtin_amdgcn_sched_barrier(0);          // this piece's fetches stay below the previous piece's MFMAs ..
    if (q + D < NP) load_wpiece(w[q + D], wg, (q + D) / RN, (q + D) % RN);
    if (rn == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < kPlanes; ++p) x[mt][p] = xf(mt0 + mt, s, p);
    }
    __builtin_amdgcn_sched_barrier(0);          // .. and above its own
#define MSR3D_TERM(PW, PX)                                                                                  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                       \
        acc[rn][mt] = W_FIRST ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[q].v[PW], x[mt][PX], acc[rn][mt], 0, 0, 0) \
                              : __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mt][PX], w[q].v[PW], acc[rn][mt], 0, 0, 0);
#if MSR3D_TRAIN_PLANES == 3
    MSR3D_TERM(2, 0)
    MSR3D_TERM(0, 2)
    MSR3D_TERM(1, 1)
    MSR3D_TERM(1, 0)
    MSR3D_TERM(0, 1)
#endif
    MSR3D_TERM(0, 0)
#undef MSR3D_TERM
  }
}

template <int RN, int MT>
__device__ __forceinline__ void zero_acc3(f32x4 (&acc)[RN][MT]) {
#pragma unroll
  for (int a = 0; a < RN; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace msr3d
