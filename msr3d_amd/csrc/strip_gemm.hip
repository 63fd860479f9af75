// strip_gemm.hip -- the token GEMMs of the spatial encoder layer whose reduction runs over the
// model width (K = D = 256), fused with the ROW-LOCAL work in front of and behind them:
//
//     C[m][n] = epilogue( sum_k prologue(inputs)[m][k] * b(n, k) )
//
// A workgroup owns a 64-row strip of tokens.  Because K is the whole row, everything the reference
// does to a row before the product -- dropout + residual + LayerNorm (one or two in a chain), adding
// the positional term, or the BACKWARD of those LayerNorms -- can be applied while the strip is
// staged into LDS (/root/reference/modules/layers/transformers.py:250-251,324-328 and their
// autograd).  Each of those was a launch of its own (rowops.hip) with a round trip through HBM; here
// it rides in the GEMM that consumes it, and the workgroups of column block 0 write the row results
// the rest of the step needs (pre-norm sums, statistics, residual gradients).
//
// Per layer, forward:  [LN(ffn) + pos -> q|k|v|cond]   [fc]   [LN(LN(fc)+x)+x -> linear1, GELU, dropout]
//            backward: [LN-bwd -> dy W2 -> GELU-bwd]   [LN-LN-bwd -> d_fc Wfc]
// (the K = 2048 products and the weight gradients stay on gemm_f32.hip's split-K kernel).
//
// Latency, not bandwidth, is what these launches pay for (M = 960 tokens: 15 strips): every load a
// workgroup needs is issued before anything waits -- the weight panel of the first column group goes
// to registers ahead of the prologue, the strip's inputs follow, and each weight fragment is
// re-loaded for the NEXT column group right after its last use, one whole group (3.4 us of MFMA)
// ahead.  No LDS staging of weights, no barrier inside the product.
//
// 512 threads = 8 waves = 2 per SIMD; f32-input MFMA 16x16x4 (exact fp32 products and sums).
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "rowmath.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using msr3d::f4_add;
using msr3d::row_drop;
using msr3d::row_ln;
using msr3d::row_ln_bwd;

constexpr int KD = 256;            // reduction length = row width
constexpr int LDA = KD + 8;        // LDS row stride: conflict-free ds_read_b128 fragments (gemm_f32.hip)
// Strip height: 64 rows (eight waves: 8 rows each in the prologue, 2 x 4 over a 64 x 64 tile in the product)
// or, where a launch would otherwise have fewer workgroups than half the chip's CUs (the 256-wide
// products at 960 tokens: 60), 32 rows -- the same eight waves with half the prologue rows and half the
// MFMAs each, twice the workgroups.

using P = msr3d_strip_gemm_t;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ------------------------------------------------------------------------------------------------
// Prologue.  Wave w stages rows 8w..8w+7 of the strip in two passes of four rows: a row is held by
// SIXTEEN lanes (lane l: row 4*pass + l/16, columns 64q + 4(l%16) .. +3 for q = 0..3), so a row
// reduction is 15 in-lane adds + 4 DPP rotate-adds inside the 16-lane row instead of a 6-step
// cross-lane butterfly per row (one wave per row: 8 rows x up to 4 reductions x 6 dependent LDS-crossbar
// shuffles = the 8-10 us this prologue first cost).  PRO is a compile-time switch and rows past M are
// read from row M-1 (results discarded): straight-line code, every load of the strip -- inputs and
// saved statistics -- issued before the first use.  `red` (LDS behind the strip, backward prologues
// only): the strip's LayerNorm parameter-gradient column sums, one atomicAdd per column and strip.
// Same operation order per element as rowmath.h / rowops.hip; only the summation tree of the row
// reductions differs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {    // all-reduce over the lane's 16-lane row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

struct Row4 { float4 q[4]; };                            // a lane's 16 elements of one row

__device__ __forceinline__ Row4 r4_add(const Row4 &a, const Row4 &b) {
  Row4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o.q[q] = f4_add(a.q[q], b.q[q]);
  return o;
}
__device__ __forceinline__ Row4 r4_drop(const Row4 &v, bool drop, unsigned long long sd, unsigned salt,
                                        unsigned thresh, float scale, int row, int cseg) {
  if (!drop) return v;
  Row4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned base = (unsigned)row * KD + 64 * q + 4 * cseg;
    o.q[q].x = msr3d::keep_elem(sd, salt, base + 0, thresh) ? v.q[q].x * scale : 0.f;
    o.q[q].y = msr3d::keep_elem(sd, salt, base + 1, thresh) ? v.q[q].y * scale : 0.f;
    o.q[q].z = msr3d::keep_elem(sd, salt, base + 2, thresh) ? v.q[q].z * scale : 0.f;
    o.q[q].w = msr3d::keep_elem(sd, salt, base + 3, thresh) ? v.q[q].w * scale : 0.f;
  }
  return o;
}
__device__ __forceinline__ Row4 r4_ln(const Row4 &v, const Row4 &g, const Row4 &b, float eps, float &mean,
                                      float &rstd) {
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) sum += (v.q[q].x + v.q[q].y) + (v.q[q].z + v.q[q].w);
  mean = row16_sum(sum) * (1.0f / KD);
  float var = 0.f;
  Row4 d;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    d.q[q] = make_float4(v.q[q].x - mean, v.q[q].y - mean, v.q[q].z - mean, v.q[q].w - mean);
    var += (d.q[q].x * d.q[q].x + d.q[q].y * d.q[q].y) + (d.q[q].z * d.q[q].z + d.q[q].w * d.q[q].w);
  }
  rstd = rsqrtf(row16_sum(var) * (1.0f / KD) + eps);
  Row4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    o.q[q] = make_float4(d.q[q].x * rstd * g.q[q].x + b.q[q].x, d.q[q].y * rstd * g.q[q].y + b.q[q].y,
                         d.q[q].z * rstd * g.q[q].z + b.q[q].z, d.q[q].w * rstd * g.q[q].w + b.q[q].w);
  return o;
}
// LayerNorm backward of one row; tg / tb receive this row's gamma / beta gradient contributions
__device__ __forceinline__ Row4 r4_ln_bwd(const Row4 &d, const Row4 &s, float mean, float rstd, const Row4 &gg,
                                          Row4 &tg, Row4 &tb) {
  Row4 xh, g;
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xh.q[q] = make_float4((s.q[q].x - mean) * rstd, (s.q[q].y - mean) * rstd, (s.q[q].z - mean) * rstd,
                          (s.q[q].w - mean) * rstd);
    g.q[q] = make_float4(d.q[q].x * gg.q[q].x, d.q[q].y * gg.q[q].y, d.q[q].z * gg.q[q].z, d.q[q].w * gg.q[q].w);
    c1 += (g.q[q].x + g.q[q].y) + (g.q[q].z + g.q[q].w);
    c2 += (g.q[q].x * xh.q[q].x + g.q[q].y * xh.q[q].y) + (g.q[q].z * xh.q[q].z + g.q[q].w * xh.q[q].w);
    tg.q[q] = make_float4(d.q[q].x * xh.q[q].x, d.q[q].y * xh.q[q].y, d.q[q].z * xh.q[q].z, d.q[q].w * xh.q[q].w);
    tb.q[q] = d.q[q];
  }
  c1 = row16_sum(c1) * (1.0f / KD);
  c2 = row16_sum(c2) * (1.0f / KD);
  Row4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    o.q[q] = make_float4(rstd * (g.q[q].x - c1 - xh.q[q].x * c2), rstd * (g.q[q].y - c1 - xh.q[q].y * c2),
                         rstd * (g.q[q].z - c1 - xh.q[q].z * c2), rstd * (g.q[q].w - c1 - xh.q[q].w * c2));
  return o;
}

template <int PRO, int ROWS>
__device__ __forceinline__ void stage_strip(const P &p, float *As, float *red, int m0, bool side) {
  constexpr int RPW = ROWS / 8;      // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cseg = lane & 15, sub = lane >> 4;
  constexpr bool USE1 = PRO != MSR3D_PRO_PLAIN;
  constexpr bool BWD = PRO == MSR3D_PRO_LNBWD || PRO == MSR3D_PRO_LN2BWD;
  constexpr int NPASS = RPW / 4;
  const bool has2 = PRO == MSR3D_PRO_LN2BWD || (PRO == MSR3D_PRO_LN && p.a2 != nullptr);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  Row4 in0[NPASS], in1[NPASS], in2[NPASS];
  float2 sv1[NPASS], sv2[NPASS];
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const int row = min(m0 + wave * RPW + 4 * j + sub, p.M - 1);
    const size_t o = (size_t)row * KD + 4 * cseg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      in0[j].q[q] = ld4(p.a0 + o + 64 * q);
      in1[j].q[q] = USE1 ? ld4(p.a1 + o + 64 * q) : z;
      in2[j].q[q] = has2 ? ld4(p.a2 + o + 64 * q) : z;
    }
    if (BWD) sv1[j] = *reinterpret_cast<const float2 *>(p.st1 + (size_t)row * 2);
    if (PRO == MSR3D_PRO_LN2BWD) sv2[j] = *reinterpret_cast<const float2 *>(p.st2 + (size_t)row * 2);
  }
  if (BWD) {                                            // the waves' column-sum slots
    for (int e = threadIdx.x; e < 4 * 8 * KD; e += 512) red[e] = 0.f;
  }
  const bool d1 = p.p1 > 0.f, d2 = p.p2 > 0.f;
  const unsigned long long sd = (d1 || d2) ? *p.seed : 0ull;
  const unsigned th1 = msr3d::drop_thresh(p.p1), th2 = msr3d::drop_thresh(p.p2);
  const float sc1 = d1 ? 1.0f / (1.0f - p.p1) : 1.0f, sc2 = d2 ? 1.0f / (1.0f - p.p2) : 1.0f;
  Row4 g1, b1, g2, b2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 64 * q + 4 * cseg;
    g1.q[q] = (PRO != MSR3D_PRO_PLAIN && p.g1) ? ld4(p.g1 + c) : z;
    b1.q[q] = (PRO != MSR3D_PRO_PLAIN && p.b1) ? ld4(p.b1 + c) : z;
    g2.q[q] = (PRO == MSR3D_PRO_LN2 || PRO == MSR3D_PRO_LN2BWD) ? ld4(p.g2 + c) : z;
    b2.q[q] = (PRO == MSR3D_PRO_LN2 && p.b2) ? ld4(p.b2 + c) : z;
  }
  if (BWD) __syncthreads();                             // red is zero before anyone adds to it
  // LayerNorm parameter gradients: the strip's column sums.  Every workgroup of the strip holds the
  // same row values, so the reduction is SHARED OUT: work item (array k, column quarter q) belongs to
  // the workgroup with blockIdx.x == (4 k + q) % gridDim.x.  A lane's partial is summed over the
  // wave's four sub-rows (two cross-row shuffles), lanes 0..15 add it to the wave's own slot
  // red[k][wave][.] (no contention); after the barrier each column costs ONE global atomicAdd per strip.
  const int gx = gridDim.x, bx = blockIdx.x;
  auto acc = [&](int k, const Row4 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((4 * k + q) % gx != bx) continue;
      float4 t = v.q[q];
      t.x += __shfl_xor(t.x, 16); t.y += __shfl_xor(t.y, 16); t.z += __shfl_xor(t.z, 16); t.w += __shfl_xor(t.w, 16);
      t.x += __shfl_xor(t.x, 32); t.y += __shfl_xor(t.y, 32); t.z += __shfl_xor(t.z, 32); t.w += __shfl_xor(t.w, 32);
      if (sub == 0) {
        float *d = red + (k * 8 + wave) * KD + 64 * q + 4 * cseg;
        st4(d, f4_add(ld4(d), t));
      }
    }
  };
  auto put = [&](float *dst, int row, const Row4 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(dst + (size_t)row * KD + 64 * q + 4 * cseg, v.q[q]);
  };
#pragma unroll
  for (int j = 0; j < NPASS; ++j) {
    const int r = wave * RPW + 4 * j + sub, row = m0 + r;
    const bool ok = row < p.M;
    const bool wr = side && ok;
    Row4 a = in0[j];
    if (PRO == MSR3D_PRO_ADD) {
      // tokens + positional term (+ the two constant embedding rows every object token receives,
      // model/ose3d_situation.py:327-365): the layer input
      a = r4_add(r4_add(r4_add(a, in1[j]), g1), b1);
      if (wr && p.o1) put(p.o1, row, a);
    } else if (PRO == MSR3D_PRO_LN) {
      // y = LN(dropout(a0) + a1) [+ a2]: a layer's closing norm, then the next layer's positional term
      const Row4 v = r4_add(r4_drop(a, d1, sd, p.salt1, th1, sc1, row, cseg), in1[j]);
      float mean, rstd;
      a = r4_add(r4_ln(v, g1, b1, p.eps1, mean, rstd), in2[j]);
      if (wr) {
        if (p.o0) put(p.o0, row, v);
        if (p.ost1 && cseg == 0) *reinterpret_cast<float2 *>(p.ost1 + (size_t)row * 2) = make_float2(mean, rstd);
        if (p.o1) put(p.o1, row, a);
      }
    } else if (PRO == MSR3D_PRO_LN2) {
      // t = LN2(dropout2(LN1(dropout1(a0) + a1)) + a1): the attention block's tail, then norm1
      // over the SAME residual (transformers.py:250-251 then :324-325)
      const Row4 v1 = r4_add(r4_drop(a, d1, sd, p.salt1, th1, sc1, row, cseg), in1[j]);
      float m1, r1, m2, r2;
      const Row4 y1 = r4_ln(v1, g1, b1, p.eps1, m1, r1);
      const Row4 v2 = r4_add(r4_drop(y1, d2, sd, p.salt2, th2, sc2, row, cseg), in1[j]);
      a = r4_ln(v2, g2, b2, p.eps2, m2, r2);
      if (wr) {
        put(p.o0, row, v1);
        put(p.o2, row, v2);
        if (cseg == 0) {
          *reinterpret_cast<float2 *>(p.ost1 + (size_t)row * 2) = make_float2(m1, r1);
          *reinterpret_cast<float2 *>(p.ost2 + (size_t)row * 2) = make_float2(m2, r2);
        }
        put(p.o1, row, a);
      }
    } else if (PRO == MSR3D_PRO_LNBWD) {
      // a0 = d y, a1 = saved pre-norm sum: dx -> o1 (the residual's gradient), dropout-bwd(dx) -> operand
      Row4 tg, tb;
      const Row4 dx = r4_ln_bwd(a, in1[j], sv1[j].x, sv1[j].y, g1, tg, tb);
      { const Row4 zr = {{z, z, z, z}}; acc(0, ok ? tg : zr); acc(1, ok ? tb : zr); }
      a = r4_drop(dx, d1, sd, p.salt1, th1, sc1, row, cseg);
      if (wr) {
        if (p.o1) put(p.o1, row, dx);
        if (p.o0) put(p.o0, row, a);
      }
    } else if (PRO == MSR3D_PRO_LN2BWD) {
      // backward of PRO_LN2: a0 = d t, a1 = v1, a2 = v2
      Row4 tg1, tb1, tg2, tb2;
      const Row4 dx2 = r4_ln_bwd(a, in2[j], sv2[j].x, sv2[j].y, g2, tg2, tb2);
      const Row4 d = r4_drop(dx2, d2, sd, p.salt2, th2, sc2, row, cseg);
      const Row4 dx1 = r4_ln_bwd(d, in1[j], sv1[j].x, sv1[j].y, g1, tg1, tb1);
      { const Row4 zr = {{z, z, z, z}};
        acc(0, ok ? tg1 : zr); acc(1, ok ? tb1 : zr); acc(2, ok ? tg2 : zr); acc(3, ok ? tb2 : zr); }
      a = r4_drop(dx1, d1, sd, p.salt1, th1, sc1, row, cseg);
      if (wr) {
        put(p.o1, row, r4_add(dx2, dx1));
        put(p.o0, row, a);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) st4(As + r * LDA + 64 * q + 4 * cseg, ok ? a.q[q] : z);
  }
  if (BWD) {
    __syncthreads();
    constexpr int NA = PRO == MSR3D_PRO_LN2BWD ? 4 : 2;
    float *const dst[4] = {p.dg1, p.db1, p.dg2, p.db2};
    for (int e = threadIdx.x; e < NA * KD; e += 512) {
      const int k = e >> 8, col = e & 255;
      if ((4 * k + (col >> 6)) % gx != bx || !dst[k]) continue;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += red[(k * 8 + w) * KD + col];
      atomicAdd(dst[k] + col, sum);
    }
  }
}

template <int PRO, int ROWS>
__global__ __launch_bounds__(512) void strip_gemm_kc_kernel(const P p) {
  constexpr int MTW = ROWS / 32;                        // 16-row tiles per wave
  extern __shared__ __attribute__((aligned(16))) float As[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.y * ROWS;
  const int N = p.N, NG = p.groups_per_wg;
  const int grp0 = blockIdx.x * NG;
  const int ngroups = (N + 63) >> 6;
  const int grp1 = min(ngroups, grp0 + NG);
  const float *__restrict__ W = p.W;

  auto wrow = [&](int grp) {                            // this lane's weight row for column group grp
    int col = grp * 64 + wn * 16 + i;
    col = col < N ? col : N - 1;                        // columns past N re-read the last row, results dropped
    return W + (size_t)col * p.ldw + 4 * g;
  };
  float4 b[16];                                         // the lane's whole K panel: 16 slabs x 4 k
  {
    const float *wp = wrow(grp0);
#pragma unroll
    for (int s = 0; s < 16; ++s) b[s] = ld4(wp + 16 * s);
  }
  stage_strip<PRO, ROWS>(p, As, As + ROWS * LDA, m0, blockIdx.x == 0);
  __syncthreads();

  const float *a_base = As + (wm * 16 * MTW + i) * LDA + 4 * g;
  const int epi = p.epi;
  const bool drop = p.p_drop > 0.f;
  const unsigned thresh = msr3d::drop_thresh(p.p_drop);
  const float dscale = drop ? 1.0f / (1.0f - p.p_drop) : 1.0f;
  const unsigned long long sd = drop ? *p.seed : 0ull;

  // epilogue of one column group: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg.  No load in
  // here (the bias arrives with the weight panel): a load next to the stores makes the compiler wait
  // vmcnt(0) -- i.e. for the previous store's completion -- before every store.
  const int rowb = m0 + wm * 16 * MTW + g * 4;
  auto finish = [&](int grp, const f32x4 (&acc)[MTW], float bv) {
    const int col = grp * 64 + wn * 16 + i;
    const bool cok = col < N;
    float v[4 * MTW];
#pragma unroll
    for (int e = 0; e < 4 * MTW; ++e) v[e] = acc[e >> 2][e & 3] + bv;
    if (epi == MSR3D_EPI_GELU) {
      if (p.Cpre) {
#pragma unroll
        for (int e = 0; e < 4 * MTW; ++e) {
          const int row = rowb + (e >> 2) * 16 + (e & 3);
          if (cok && row < p.M) p.Cpre[(size_t)row * p.ldc + col] = v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4 * MTW; ++e) {
        const int row = rowb + (e >> 2) * 16 + (e & 3);
        v[e] = msr3d::gelu_exact(v[e]);
        if (drop)
          v[e] = msr3d::keep_elem(sd, p.salt, (unsigned)((size_t)row * p.ldc + col), thresh) ? v[e] * dscale : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4 * MTW; ++e) {
      const int row = rowb + (e >> 2) * 16 + (e & 3);
      if (cok && row < p.M) p.C[(size_t)row * p.ldc + col] = v[e];
    }
  };
  auto bias_of = [&](int grp) {
    int col = grp * 64 + wn * 16 + i;
    col = col < N ? col : N - 1;
    return p.bias ? p.bias[col] : 0.f;
  };
  // Software pipeline over the column groups: [issue the NEXT group's weight panel] [this group's
  // MFMAs] [the PREVIOUS group's epilogue and stores].  The panel loads have a whole MFMA phase to
  // land, and no wait ever sits behind a freshly issued store.
  f32x4 pacc[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) pacc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv_cur = bias_of(grp0), bv_prev = 0.f;
  // one pipeline stage; the two panel buffers alternate roles (no register copy, hence no wait for
  // the freshly loaded panel -- nor for the stores issued after it -- at the end of a stage)
  auto stage = [&](float4 (&bc)[16], float4 (&bnx)[16], int grp) {
    const bool more = grp + 1 < grp1;
    const float *wnext = wrow(more ? grp + 1 : grp);
#pragma unroll
    for (int s = 0; s < 16; ++s) bnx[s] = ld4(wnext + 16 * s);   // (last group: a re-read, unused)
    const float bv_next = bias_of(more ? grp + 1 : grp);
    __builtin_amdgcn_sched_barrier(0);                  // keep the loads in front of the MFMA phase
    f32x4 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float4 a[MTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) a[mt] = ld4(a_base + mt * 16 * LDA + 16 * s);
      const float4 bb = bc[s];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, bb.x, acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, bb.y, acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, bb.z, acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, bb.w, acc[mt], 0, 0, 0);
    }
    if (grp > grp0) finish(grp - 1, pacc, bv_prev);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) pacc[mt] = acc[mt];
    bv_prev = bv_cur; bv_cur = bv_next;
  };
  float4 b2[16];
  for (int grp = grp0; grp < grp1; grp += 2) {
    stage(b, b2, grp);
    if (grp + 1 < grp1) stage(b2, b, grp + 1);
  }
  finish(grp1 - 1, pacc, bv_prev);
}

// ------------------------------------------------------------------------------------------------
// b(n, k) = W[k * ldw + n] (backward products dx = dy W: the reduction runs over W's rows).  A lane
// cannot read four consecutive k of one column with one load here, so it reads consecutive COLUMNS
// instead: the 64-column group is covered by four MFMA column tiles with tile t, lane i <-> column
// 4 i + t; wave (wr, wc) owns the 16-row tile wr and tiles {2 wc, 2 wc + 1}, i.e. one 8-byte load
// per k fetches both of its B values and one 8-byte store per row writes both results.
// ------------------------------------------------------------------------------------------------
template <int PRO, int ROWS>
__global__ __launch_bounds__(512) void strip_gemm_kr_kernel(const P p) {
  // 64-row strips: wave (wr, wc) owns the 16-row tile wr (of four) and column tiles {2 wc, 2 wc + 1} -- CT = 2
  // values per 8-byte load / store; 32-row strips: row tile wr (of two) and the single column tile wc -- CT = 1.
  constexpr int CT = ROWS / 32;
  extern __shared__ __attribute__((aligned(16))) float As[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wr = CT == 2 ? (wave & 3) : (wave & 1), wc = CT == 2 ? (wave >> 2) : (wave >> 1);
  const int m0 = blockIdx.y * ROWS;
  const int N = p.N, NG = p.groups_per_wg;
  const int grp0 = blockIdx.x * NG;
  const int grp1 = min(N >> 6, grp0 + NG);              // N % 64 == 0 (checked by the host entry)
  const float *__restrict__ W = p.W;
  const int ldw = p.ldw;
  auto ldc = [&](const float *q, float (&v)[CT]) {      // CT consecutive floats
    if (CT == 2) { const float2 t = *reinterpret_cast<const float2 *>(q); v[0] = t.x; v[CT - 1] = t.y; }
    else v[0] = *q;
  };

  // element (s, q) = W[(16 s + 4 g + q)][n0 + 4 i + CT wc + {0 .. CT-1}].  Eight slabs (half a panel) live in
  // registers: slab s sits in slot s % 8 and is replaced by slab s + 8 right after its use, i.e. the
  // weight stream runs half a column group (~1 us of MFMA) ahead of the product.
  const float *wbase = W + (size_t)(4 * g) * ldw + 4 * i + CT * wc;
  float b[8][4][CT];
  auto load_ring = [&]() {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) ldc(wbase + (size_t)(16 * s + q) * ldw + grp0 * 64, b[s][q]);
  };
  // (the two-LayerNorm backward prologue needs the registers: its weight ring is fetched afterwards)
  constexpr bool LATE_B = PRO == MSR3D_PRO_LN2BWD;
  if (!LATE_B) load_ring();
  stage_strip<PRO, ROWS>(p, As, As + ROWS * LDA, m0, blockIdx.x == 0);
  if (LATE_B) load_ring();
  __syncthreads();

  const float *a_base = As + (wr * 16 + i) * LDA + 4 * g;
  const int epi = p.epi;
  const bool drop = p.p_drop > 0.f;
  const unsigned thresh = msr3d::drop_thresh(p.p_drop);
  const float dscale = drop ? 1.0f / (1.0f - p.p_drop) : 1.0f;
  const unsigned long long sd = drop ? *p.seed : 0ull;

  // epilogue of one column group (its saved pre-activations were fetched a whole group earlier)
  auto finish = [&](int grp, const f32x4 (&acc)[CT], const float (&pre)[4][CT]) {
    const int colb = grp * 64 + 4 * i + CT * wc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 16 + g * 4 + r;
      if (row >= p.M) continue;
      float v[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) v[c] = acc[c][r];
      const size_t o = (size_t)row * p.ldc + colb;
      if (epi == MSR3D_EPI_GELUBWD) {
        // d pre = dropout-bwd(d h) * gelu'(pre); the forward's mask index is row * N + col
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (drop) {
            const unsigned mi = (unsigned)((size_t)row * N + colb) + c;
            v[c] = msr3d::keep_elem(sd, p.salt, mi, thresh) ? v[c] * dscale : 0.f;
          }
          v[c] *= msr3d::gelu_exact_grad(pre[r][c]);
        }
      }
      if (CT == 2) *reinterpret_cast<float2 *>(p.C + o) = make_float2(v[0], v[CT - 1]);
      else p.C[o] = v[0];
    }
  };
  // same software pipeline as the forward kernel: [refills + this group's pre-activations]
  // [MFMAs] [the previous group's epilogue]
  f32x4 pacc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) pacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ppre[4][CT] = {};
  for (int grp = grp0; grp < grp1; ++grp) {
    const bool more = grp + 1 < grp1;
    const int n0 = grp * 64, nn = (more ? grp + 1 : grp) * 64;
    float pre[4][CT] = {};
    if (epi == MSR3D_EPI_GELUBWD) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(m0 + wr * 16 + g * 4 + r, p.M - 1);
        ldc(p.pre_in + (size_t)row * N + n0 + 4 * i + CT * wc, pre[r]);
      }
    }
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float4 a = ld4(a_base + 16 * s);
      const float av[4] = {a.x, a.y, a.z, a.w};
      float bb[4][CT];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int c = 0; c < CT; ++c) bb[q][c] = b[s & 7][q][c];
        if (s < 8)
          ldc(wbase + (size_t)(16 * (s + 8) + q) * ldw + n0, b[s][q]);
        else      // (last group: nn == n0, a harmless re-read)
          ldc(wbase + (size_t)(16 * (s - 8) + q) * ldw + nn, b[s - 8][q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bb[q][c], acc[c], 0, 0, 0);
    }
    if (grp > grp0) finish(grp - 1, pacc, ppre);
#pragma unroll
    for (int c = 0; c < CT; ++c) pacc[c] = acc[c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < CT; ++c) ppre[r][c] = pre[r][c];
  }
  finish(grp1 - 1, pacc, ppre);
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" int msr3d_strip_gemm_f32(const msr3d_strip_gemm_t *pp, msr3d_stream_t stream) {
  if (!pp) return MSR3D_EINVAL;
  P p = *pp;
  if (p.M < 0 || p.N < 0) return MSR3D_EINVAL;
  if (p.M == 0 || p.N == 0) return 0;
  if (!p.a0 || !p.W || !p.C || p.ldw <= 0 || p.ldc < p.N) return MSR3D_EINVAL;
  if (!al16(p.a0) || !al16(p.a1) || !al16(p.a2) || !al16(p.W) || !al16(p.o0) || !al16(p.o1) || !al16(p.o2) ||
      !al16(p.g1) || !al16(p.b1) || !al16(p.g2) || !al16(p.b2))
    return MSR3D_EINVAL;
  if (p.p1 < 0.f || p.p1 >= 1.f || p.p2 < 0.f || p.p2 >= 1.f || p.p_drop < 0.f || p.p_drop >= 1.f)
    return MSR3D_EINVAL;
  if ((p.p1 > 0.f || p.p2 > 0.f || p.p_drop > 0.f) && !p.seed) return MSR3D_EINVAL;
  switch (p.pro) {
    case MSR3D_PRO_PLAIN: break;
    case MSR3D_PRO_ADD: if (!p.a1) return MSR3D_EINVAL; break;
    case MSR3D_PRO_LN: if (!p.a1 || !p.g1 || !p.b1) return MSR3D_EINVAL; break;
    case MSR3D_PRO_LN2:
      if (!p.a1 || !p.g1 || !p.b1 || !p.g2 || !p.b2 || !p.o0 || !p.o1 || !p.o2 || !p.ost1 || !p.ost2)
        return MSR3D_EINVAL;
      break;
    case MSR3D_PRO_LNBWD: if (!p.a1 || !p.st1 || !p.g1) return MSR3D_EINVAL; break;
    case MSR3D_PRO_LN2BWD:
      if (!p.a1 || !p.a2 || !p.st1 || !p.st2 || !p.g1 || !p.g2 || !p.o0 || !p.o1) return MSR3D_EINVAL;
      break;
    default: return MSR3D_EINVAL;
  }
  if (p.epi != MSR3D_EPI_BIAS && p.epi != MSR3D_EPI_GELU && p.epi != MSR3D_EPI_GELUBWD) return MSR3D_EINVAL;
  if (p.epi == MSR3D_EPI_GELU && p.p_drop > 0.f && p.ldc != p.N) return MSR3D_EINVAL;   // mask index = row * N + col
  const int ngroups = (p.N + 63) / 64;
  // 32-row strips where 64-row ones would leave more than half of the chip without a workgroup
  const bool half = (long long)ngroups * ((p.M + 63) / 64) <= 128 && p.M > 32;
  const int ROWS = half ? 32 : 64;
  const int strips = (p.M + ROWS - 1) / ROWS;
  int ng = p.groups_per_wg;
  if (ng <= 0) {                     // about one workgroup per CU
    ng = (ngroups * strips + 128) / 256;
    if (ng < 1) ng = 1;
  }
  p.groups_per_wg = ng;
  const dim3 grid((ngroups + ng - 1) / ng, strips);
  const bool bwd = p.pro == MSR3D_PRO_LNBWD || p.pro == MSR3D_PRO_LN2BWD;
  const size_t lds = sizeof(float) * (ROWS * LDA + (bwd ? 4 * 8 * KD : 0));
  hipStream_t st = (hipStream_t)stream;
  if (p.b_kc) {
    if (p.epi == MSR3D_EPI_GELUBWD || bwd) return MSR3D_EINVAL;
    if (p.ldw % 4) return MSR3D_EINVAL;
  } else {
    if (p.N % 64 || p.ldw % 2 || p.ldc % 2 || p.epi == MSR3D_EPI_GELU || p.bias) return MSR3D_EINVAL;
    if (p.epi == MSR3D_EPI_GELUBWD && !p.pre_in) return MSR3D_EINVAL;
    if (p.pro != MSR3D_PRO_PLAIN && !bwd) return MSR3D_EINVAL;      // the backward products' prologues
  }
#define LAUNCH1(KERN, PROC, R)                                                                              \
  do {                                                                                                \
    static bool attr_done = false;                                                                    \
    if (!attr_done) {                                                                                 \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERN<PROC, R>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                               (int)(sizeof(float) * (R * LDA + 4 * 8 * KD)));        \
      if (e != hipSuccess) return (int)e;                                                             \
      attr_done = true;                                                                               \
    }                                                                                                 \
    KERN<PROC, R><<<grid, 512, lds, st>>>(p);                                                         \
  } while (0)
#define LAUNCH(KERN, PROC)                                                                                  \
  do {                                                                                                \
    if (half) LAUNCH1(KERN, PROC, 32); else LAUNCH1(KERN, PROC, 64);                                  \
  } while (0)
  switch (p.pro) {
    case MSR3D_PRO_PLAIN:
      if (p.b_kc) LAUNCH(strip_gemm_kc_kernel, MSR3D_PRO_PLAIN); else LAUNCH(strip_gemm_kr_kernel, MSR3D_PRO_PLAIN);
      break;
    case MSR3D_PRO_ADD: LAUNCH(strip_gemm_kc_kernel, MSR3D_PRO_ADD); break;
    case MSR3D_PRO_LN: LAUNCH(strip_gemm_kc_kernel, MSR3D_PRO_LN); break;
    case MSR3D_PRO_LN2: LAUNCH(strip_gemm_kc_kernel, MSR3D_PRO_LN2); break;
    case MSR3D_PRO_LNBWD: LAUNCH(strip_gemm_kr_kernel, MSR3D_PRO_LNBWD); break;
    default: LAUNCH(strip_gemm_kr_kernel, MSR3D_PRO_LN2BWD); break;
  }
#undef LAUNCH
#undef LAUNCH1
  return (int)hipGetLastError();
}
