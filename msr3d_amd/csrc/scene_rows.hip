// scene_rows.hip -- the row-local half of the scene blocks (include/msr3d_hip.h: msr3d_scene_rows).
//
// A scene block (scene_block.hip) ends with `slices` partial products of a (tokens x 256) result, one
// per workgroup of a scene.  Meeting them by fp32 atomicAdd costs 24 us per block on this part: every
// float atomic executes at the memory side of the fabric (TCC_EA0_ATOMIC == TCC_ATOMIC, whatever scope
// the source names -- the L2s of the eight XCDs are not coherent), one 64-byte request per ~47 cycles
// and CU.  And applying the row-local chain that FOLLOWS the sum (dropout + residual + LayerNorm, once
// or twice, or their backward) inside the consuming block repeats it in each of that block's 8 / 16
// slice workgroups: 10 us of redundant VALU work per block.
//
// So the partials are stored as plain slabs and this kernel does, ONCE per token row and with the whole
// chip (one wave per row, 960 waves):
//
//     a0   = sum_s part[s][row] (+ extra[row]) (+ bias)          fixed order: bit-reproducible
//     A    = prologue(a0, a1, a2, ..)                            MSR3D_PRO_* exactly as msr3d_strip_gemm_f32
//     xp   = A split exactly into three bf16 planes              the next block's matrix operand
//
// plus the row outputs the backward needs (pre-norm sums, statistics, layer inputs) and, in the backward
// chains, the LayerNorm parameter gradients (column sums: one atomicAdd per column and 4 rows).
// Arithmetic per element: rowmath.h, i.e. the operation order of rowops.hip's stand-alone row kernels
// (/root/reference/modules/layers/transformers.py:250-251,324-328).
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "colsum.h"
#include "rowmath.h"
#include "split_mma.h"

namespace {

using namespace msr3d;
using SR = msr3d_scene_rows_t;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

constexpr int MAXS = 20;

template <int PRO>
__global__ __launch_bounds__(256) void scene_rows_kernel(const SR p) {
  constexpr bool BWD = PRO == MSR3D_PRO_LNBWD || PRO == MSR3D_PRO_LN2BWD;
  __shared__ __attribute__((aligned(16))) float red[BWD ? 4 : 1][4][ROW_D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  const bool ok = row < p.M;
  const int rc = ok ? row : p.M - 1;
  const size_t o = (size_t)rc * ROW_D + 4 * lane;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- every load of the row is issued before anything waits ----
  float4 v[MAXS];
  const int ns = p.nslab;
  if (ns == 0) {
    v[0] = ld4(p.a0 + o);
  } else {
#pragma unroll
    for (int s = 0; s < MAXS; ++s) v[s] = s < ns ? ld4(p.part + (size_t)s * p.part_stride + o) : z;
  }
  const float4 ex = p.extra ? ld4(p.extra + o) : z;
  const float4 bs = p.a0_bias ? ld4(p.a0_bias + 4 * lane) : z;
  constexpr bool USE1 = PRO != MSR3D_PRO_PLAIN;
  const bool has2 = PRO == MSR3D_PRO_LN2BWD || (PRO == MSR3D_PRO_LN && p.a2 != nullptr);
  const float4 in1 = USE1 ? ld4(p.a1 + o) : z;
  const float4 in2 = has2 ? ld4(p.a2 + o) : z;
  float2 sv1 = make_float2(0.f, 0.f), sv2 = make_float2(0.f, 0.f);
  if (BWD) sv1 = *reinterpret_cast<const float2 *>(p.st1 + (size_t)rc * 2);
  if (PRO == MSR3D_PRO_LN2BWD) sv2 = *reinterpret_cast<const float2 *>(p.st2 + (size_t)rc * 2);
  const float4 g1 = (PRO != MSR3D_PRO_PLAIN && p.g1) ? ld4(p.g1 + 4 * lane) : z;
  const float4 b1 = (PRO != MSR3D_PRO_PLAIN && p.b1) ? ld4(p.b1 + 4 * lane) : z;
  const float4 g2 = (PRO == MSR3D_PRO_LN2 || PRO == MSR3D_PRO_LN2BWD) ? ld4(p.g2 + 4 * lane) : z;
  const float4 b2 = (PRO == MSR3D_PRO_LN2 && p.b2) ? ld4(p.b2 + 4 * lane) : z;
  const bool d1 = p.p1 > 0.f, d2 = p.p2 > 0.f;
  const unsigned long long sd = (d1 || d2) ? *p.seed : 0ull;
  const unsigned th1 = drop_thresh(p.p1), th2 = drop_thresh(p.p2);
  const float sc1 = d1 ? 1.0f / (1.0f - p.p1) : 1.0f, sc2 = d2 ? 1.0f / (1.0f - p.p2) : 1.0f;

  float4 a = v[0];
  if (ns > 0) {
#pragma unroll
    for (int s = 1; s < MAXS; ++s)
      if (s < ns) a = f4_add(a, v[s]);
    if (p.extra) a = f4_add(a, ex);
    if (p.a0_bias) a = f4_add(a, bs);
  }
  if (p.sum_out && ok) st4(p.sum_out + o, a);

  float4 tg1 = z, tb1 = z, tg2 = z, tb2 = z;
  if (PRO == MSR3D_PRO_ADD) {
    a = f4_add(f4_add(f4_add(a, in1), g1), b1);
    if (ok && p.o1) st4(p.o1 + o, a);
  } else if (PRO == MSR3D_PRO_LN) {
    const float4 s = f4_add(row_drop(a, d1, sd, p.salt1, th1, sc1, row, lane), in1);
    float mean, rstd;
    a = f4_add(row_ln(s, g1, b1, p.eps1, mean, rstd), in2);
    if (ok) {
      if (p.o0) st4(p.o0 + o, s);
      if (p.ost1 && lane == 0) *reinterpret_cast<float2 *>(p.ost1 + (size_t)row * 2) = make_float2(mean, rstd);
      if (p.o1) st4(p.o1 + o, a);
    }
  } else if (PRO == MSR3D_PRO_LN2) {
    const float4 v1 = f4_add(row_drop(a, d1, sd, p.salt1, th1, sc1, row, lane), in1);
    float m1, r1, m2, r2;
    const float4 y1 = row_ln(v1, g1, b1, p.eps1, m1, r1);
    const float4 v2 = f4_add(row_drop(y1, d2, sd, p.salt2, th2, sc2, row, lane), in1);
    a = row_ln(v2, g2, b2, p.eps2, m2, r2);
    if (ok) {
      st4(p.o0 + o, v1);
      st4(p.o2 + o, v2);
      if (lane == 0) {
        *reinterpret_cast<float2 *>(p.ost1 + (size_t)row * 2) = make_float2(m1, r1);
        *reinterpret_cast<float2 *>(p.ost2 + (size_t)row * 2) = make_float2(m2, r2);
      }
      st4(p.o1 + o, a);
    }
  } else if (PRO == MSR3D_PRO_LNBWD) {
    const float4 dx = row_ln_bwd(a, in1, sv1.x, sv1.y, g1, tg1, tb1);
    a = row_drop(dx, d1, sd, p.salt1, th1, sc1, row, lane);
    if (ok) {
      if (p.o1) st4(p.o1 + o, dx);
      if (p.o0) st4(p.o0 + o, a);
    }
  } else if (PRO == MSR3D_PRO_LN2BWD) {
    const float4 dx2 = row_ln_bwd(a, in2, sv2.x, sv2.y, g2, tg2, tb2);
    const float4 d = row_drop(dx2, d2, sd, p.salt2, th2, sc2, row, lane);
    const float4 dx1 = row_ln_bwd(d, in1, sv1.x, sv1.y, g1, tg1, tb1);
    a = row_drop(dx1, d1, sd, p.salt1, th1, sc1, row, lane);
    if (ok) {
      st4(p.o1 + o, f4_add(dx2, dx1));
      st4(p.o0 + o, a);
    }
  }
  // the next block's operand: three bf16 planes, (scene, plane, 64 rows, 256) -- rows past L stay zero
  if (p.xp && ok) {
    const int b = row / p.L, r = row - b * p.L;
    const float f[4] = {a.x, a.y, a.z, a.w};
    uint2 pl[3];
    sm_split4(f, pl);
    unsigned short *d = p.xp + ((size_t)b * 3 * 64 + r) * ROW_D + 4 * lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + (size_t)k * 64 * ROW_D) = pl[k];
  }
  if (BWD) {
    // LayerNorm parameter gradients: the four rows of the workgroup meet in LDS, one atomicAdd per column
    st4(&red[0][wave][4 * lane], ok ? tg1 : z);
    st4(&red[1][wave][4 * lane], ok ? tb1 : z);
    if (PRO == MSR3D_PRO_LN2BWD) {
      st4(&red[2][wave][4 * lane], ok ? tg2 : z);
      st4(&red[3][wave][4 * lane], ok ? tb2 : z);
    }
    __syncthreads();
    float *const dst[4] = {p.dg1, p.db1, p.dg2, p.db2};
    constexpr int NA = PRO == MSR3D_PRO_LN2BWD ? 4 : 2;
    const int col = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      if (!dst[k]) continue;
      const float s4 = (red[k][0][col] + red[k][1][col]) + (red[k][2][col] + red[k][3][col]);
      // 240 workgroups x 256 columns x 2-4 vectors of memory-side atomics cost ~4 us of a 12 us launch: as plain
      // stores to the workgroup's row of a partial buffer they are free, and the sum becomes ordered
      if (p.grad_partials) dst[k][(size_t)blockIdx.x * ROW_D + col] = s4;
      else atomicAdd(dst[k] + col, s4);
    }
  }
}

template <int PRO>
int launch_rows(const SR &p, hipStream_t s) {
  scene_rows_kernel<PRO><<<(p.M + 3) / 4, 256, 0, s>>>(p);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void colsum_partials_kernel(const msr3d_colsum_job_t *__restrict__ jobs) {
  __shared__ __attribute__((aligned(16))) float red[4 * ROW_D];
  colsum_job(jobs[blockIdx.x], red);
}

}  // namespace

extern "C" int msr3d_colsum_partials(int n_jobs, const msr3d_colsum_job_t *jobs, msr3d_stream_t stream) {
  if (n_jobs < 0) return MSR3D_EINVAL;
  if (n_jobs == 0) return 0;
  if (!jobs) return MSR3D_EINVAL;
  colsum_partials_kernel<<<n_jobs, 256, 0, (hipStream_t)stream>>>(jobs);
  return (int)hipGetLastError();
}

extern "C" int msr3d_scene_rows(const msr3d_scene_rows_t *pp, msr3d_stream_t stream) {
  if (!pp) return MSR3D_EINVAL;
  const SR &p = *pp;
  if (p.M < 0 || p.L <= 0 || p.L > 64 || p.nslab < 0 || p.nslab > MAXS) return MSR3D_EINVAL;
  if (p.M == 0) return 0;
  if (p.nslab == 0 ? !p.a0 : !p.part) return MSR3D_EINVAL;
  if ((p.p1 > 0.f || p.p2 > 0.f) && !p.seed) return MSR3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  switch (p.pro) {
    case MSR3D_PRO_PLAIN: return launch_rows<MSR3D_PRO_PLAIN>(p, s);
    case MSR3D_PRO_ADD:
      if (!p.a1) return MSR3D_EINVAL;
      return launch_rows<MSR3D_PRO_ADD>(p, s);
    case MSR3D_PRO_LN:
      if (!p.a1 || !p.g1) return MSR3D_EINVAL;
      return launch_rows<MSR3D_PRO_LN>(p, s);
    case MSR3D_PRO_LN2:
      if (!p.a1 || !p.g1 || !p.g2 || !p.o0 || !p.o1 || !p.o2 || !p.ost1 || !p.ost2) return MSR3D_EINVAL;
      return launch_rows<MSR3D_PRO_LN2>(p, s);
    case MSR3D_PRO_LNBWD:
      if (!p.a1 || !p.st1 || !p.g1) return MSR3D_EINVAL;
      return launch_rows<MSR3D_PRO_LNBWD>(p, s);
    case MSR3D_PRO_LN2BWD:
      if (!p.a1 || !p.a2 || !p.st1 || !p.st2 || !p.g1 || !p.g2 || !p.o0 || !p.o1) return MSR3D_EINVAL;
      return launch_rows<MSR3D_PRO_LN2BWD>(p, s);
    default:
      return MSR3D_EINVAL;
  }
}
