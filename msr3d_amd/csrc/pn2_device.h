// pn2_device.h -- device helpers shared by pn2_ops.hip and sa_fused.hip: the pinned
// distance chain, the wave64 DPP max, the reference's block-size rule and the FPS kernel.
// Both translation units are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

namespace msr3d {

constexpr int kWave = 64;

// (a*a + b*b) + c*c under the floating-point contract the library is built with (oracle/pn2_oracle.c: sq3 has
// the same table and a run-time switch).  0, the default, is how an LLVM-based device compiler contracts the
// reference's expression; `python -m msr3d_amd.build --sqdist-contract N` rebuilds the index kernels under
// another one (msr3d_sqdist_contract() reports the value compiled in).  The files that include this header
// are built with -ffp-contract=off, so nothing is fused beyond what is written here.
#ifndef MSR3D_SQDIST_CONTRACT
#define MSR3D_SQDIST_CONTRACT 0
#endif
__device__ __forceinline__ float sq3(float a, float b, float c) {
#if MSR3D_SQDIST_CONTRACT == 1
  const float aa = a * a, bb = b * b, cc = c * c;
  const float s = aa + bb;
  return s + cc;
#elif MSR3D_SQDIST_CONTRACT == 2
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#elif MSR3D_SQDIST_CONTRACT == 3
  return __builtin_fmaf(a, a, __builtin_fmaf(b, b, c * c));
#else
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#endif
}

// Two points a lane at once: the same chain on both halves of a register pair (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: every half is the IEEE operation of the scalar instruction, so sq3x2(a, b, c)[h] == sq3(a[h], b[h], c[h])
// bit for bit under every contract above).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 sq3x2(f32x2 a, f32x2 b, f32x2 c) {
#if MSR3D_SQDIST_CONTRACT == 1
  const f32x2 aa = a * a, bb = b * b, cc = c * c;
  const f32x2 s = aa + bb;
  return s + cc;
#elif MSR3D_SQDIST_CONTRACT == 2
  return __builtin_elementwise_fma(c, c, __builtin_elementwise_fma(b, b, a * a));
#elif MSR3D_SQDIST_CONTRACT == 3
  return __builtin_elementwise_fma(a, a, __builtin_elementwise_fma(b, b, c * c));
#else
  return __builtin_elementwise_fma(c, c, __builtin_elementwise_fma(a, a, b * b));
#endif
}

// fminf without the canonicalising v_max the compiler puts in front of it: both operands here are results of
// arithmetic instructions or constants (never a signalling NaN), for which v_min_f32 IS fminf.
__device__ __forceinline__ float min_arith(float a, float b) {
  float r;
  asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// ---- wave64 integer max, all lanes -> uniform ---------------------------------
// One step: max(v, v of the partner lane) as ONE v_max_i32 with the DPP control on its first operand (through
// __builtin_amdgcn_update_dpp the compiler emits v_mov + s_nop + v_mov_dpp + v_max: twice the dependent latency, and
// this reduction sits on the sampling chain once per pick).  The s_nop covers "VALU writes a VGPR, DPP reads it".
#define MSR3D_DPP_MAX(name, ctrl)                                                                     \
  __device__ __forceinline__ int name(int v) {                                                        \
    int r;                                                                                            \
    asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
    return r;                                                                                         \
  }
MSR3D_DPP_MAX(dpp_max_xor1, "quad_perm:[1,0,3,2]")
MSR3D_DPP_MAX(dpp_max_xor2, "quad_perm:[2,3,0,1]")
MSR3D_DPP_MAX(dpp_max_half_mirror, "row_half_mirror")
MSR3D_DPP_MAX(dpp_max_mirror, "row_mirror")
#undef MSR3D_DPP_MAX

__device__ __forceinline__ int wave_max_i32(int v) {
  v = dpp_max_xor1(v);
  v = dpp_max_xor2(v);
  v = dpp_max_half_mirror(v);
  v = dpp_max_mirror(v);      // every lane holds its 16-lane row max
  const int r0 = __builtin_amdgcn_readlane(v, 0);
  const int r1 = __builtin_amdgcn_readlane(v, 16);
  const int r2 = __builtin_amdgcn_readlane(v, 32);
  const int r3 = __builtin_amdgcn_readlane(v, 48);
  const int a = r0 > r1 ? r0 : r1;
  const int b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// include/cuda_utils.h:13-19 of the reference (its block size enters the FPS tie-break)
inline int ref_opt_n_threads(int work_size) {
  const int pow_2 = (int)(std::log((double)work_size) / std::log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

struct FpsShape {   // host-computed description of the reference's launch for n points
  int n, bs, log2bs, q;
  long long slots;
};

inline FpsShape fps_shape(int n) {
  FpsShape s;
  s.n = n;
  s.bs = ref_opt_n_threads(n);
  s.log2bs = 0;
  while ((1 << s.log2bs) < s.bs) ++s.log2bs;
  s.q = (n + s.bs - 1) / s.bs;            // points per reference thread (upper bound)
  s.slots = (long long)s.bs * s.q;        // rank slots incl. holes (< 2n)
  return s;
}

// =================================================================================
// Furthest point sampling.
//
// The reference (sampling_gpu.cu:69-173) runs one block of bs = opt_n_threads(n)
// threads per cloud: thread t scans k = t, t+bs, ... with a strict '>' (lowest k
// wins inside a thread), then a shared-memory halving tree keeps the LOWER slot on
// ties.  Two tied threads meet at the stride equal to their lowest differing tid
// bit and the one with that bit clear survives, so among equal maxima the winner
// minimises (bitrev(k mod bs), k / bs).
//
// Here each cloud is owned by NW waves and every lane keeps PPT points in
// registers (coordinates + running min distance), laid out in exactly that rank
// order: slot p = tid*PPT + i holds the point of rank p.  One iteration is then
//   per-lane strict-'>' scan (first max in rank order inside the lane)
//   -> DPP wave max of the f32 bit pattern (values are >= +0 or the -1 sentinel)
//   -> ballot of the lanes that hold the max, lowest set lane = lowest rank
//   -> v_readlane of that lane's point index, LDS broadcast read of its coordinates.
// No LDS traffic or barrier inside an iteration when NW == 1.
//
// `src` holds the cloud with `ps` floats per point (3 = packed xyz; 6 = the
// dataset's xyz+rgb rows read in place).  Winners are appended to out_idx /
// out_xyz (global, optional) and to `keep` (LDS, optional, packed xyz).
// =================================================================================
template <int PPT, int NW>
__device__ __forceinline__ void fps_level(const float *src, int ps, int n, int m, int bs,
                                          int log2bs, int q, int *red_bits, int *red_k,
                                          int *__restrict__ out_idx, float *__restrict__ out_xyz,
                                          float *keep, int *progress = nullptr) {
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
  int kk[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid * PPT + i;       // rank slot
    const int tr = p / q;              // bit-reversed reference thread id
    const int j = p - tr * q;          // that thread's j-th point
    const int t = log2bs ? (int)(__brev((unsigned)tr) >> (32 - log2bs)) : 0;
    const int k = t + j * bs;
    const bool valid = (tr < bs) && (k < n);
    float x = 0.f, y = 0.f, z = 0.f;
    bool live = false;
    if (valid) {
      x = src[k * ps + 0];
      y = src[k * ps + 1];
      z = src[k * ps + 2];
      const float mag = sq3(x, y, z);
      live = !((double)mag <= 1e-3);   // sampling_gpu.cu:100-101 (double compare)
    }
    px[i] = x; py[i] = y; pz[i] = z;
    kk[i] = valid ? k : 0;
    // skipped / padding slots: min(d, -inf) = -inf never beats the -1 sentinel
    tmp[i] = live ? 1e10f : -INFINITY;
  }

  // A pick is a dependent chain: scan -> wave maximum -> winner's lane -> its coordinates (LDS).  What is written
  // about winner j - 1 (out_idx / out_xyz, `keep`, the counter the query waves poll) is ISSUED before pick j's scan
  // and nothing waits for it there: the counter's release store comes after the scan, when those writes have long
  // retired (measured: equal to publishing in front of the scan -- the launch is issue-bound, not bound by this wait).
  int old = 0;
  float ox = src[0], oy = src[1], oz = src[2];
  int par = 0;
  for (int jj = 1;; ++jj) {
    if (tid == 0) {
      if (out_idx) out_idx[jj - 1] = old;
      if (out_xyz) { out_xyz[(jj - 1) * 3 + 0] = ox; out_xyz[(jj - 1) * 3 + 1] = oy; out_xyz[(jj - 1) * 3 + 2] = oz; }
      if (keep) { keep[(jj - 1) * 3 + 0] = ox; keep[(jj - 1) * 3 + 1] = oy; keep[(jj - 1) * 3 + 2] = oz; }
    }
    if (jj >= m) break;
    float best = -1.0f;
    int bk = 0;
    if (PPT % 2 == 0) {
      const f32x2 o_x = {ox, ox}, o_y = {oy, oy}, o_z = {oz, oz};
#pragma unroll
      for (int i = 0; i + 1 < PPT; i += 2) {
        const f32x2 x2 = {px[i], px[i + 1]}, y2 = {py[i], py[i + 1]}, z2 = {pz[i], pz[i + 1]};
        const f32x2 d = sq3x2(x2 - o_x, y2 - o_y, z2 - o_z);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float d2 = min_arith(d[h], tmp[i + h]);
          tmp[i + h] = d2;
          const bool gt = d2 > best;
          bk = gt ? kk[i + h] : bk;
          best = gt ? d2 : best;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float d = sq3(px[i] - ox, py[i] - oy, pz[i] - oz);
        const float d2 = min_arith(d, tmp[i]);
        tmp[i] = d2;
        const bool gt = d2 > best;
        bk = gt ? kk[i] : bk;
        best = gt ? d2 : best;
      }
    }
    const int bits = __float_as_int(best);   // >= +0.0 or -1.0f: int order == float order
    // (fps_query_kernel: the query waves of this workgroup start on a winner as soon as it is in `keep`)
    if (tid == 0 && progress) __hip_atomic_store(progress, jj, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    int vmax = wave_max_i32(bits);
    const unsigned long long hit = __ballot(bits == vmax);
    const int first = __ffsll((long long)hit) - 1;
    int kw = __builtin_amdgcn_readlane(bk, first);
    if (NW > 1) {
      if (lane == 0) {
        red_bits[par * NW + wave] = vmax;
        red_k[par * NW + wave] = kw;
      }
      __syncthreads();
      vmax = red_bits[par * NW];
      kw = red_k[par * NW];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const int vb = red_bits[par * NW + w];
        const int vk = red_k[par * NW + w];
        const bool gt = vb > vmax;   // strict: the lower wave (= lower rank) wins ties
        kw = gt ? vk : kw;
        vmax = gt ? vb : vmax;
      }
      par ^= 1;
    }
    old = vmax < 0 ? 0 : kw;   // every candidate skipped: all threads report (-1, 0)
    const int o = __mul24(old, ps);   // (indices and strides < 2^23: v_mul_i32_i24 is full rate, v_mul_lo_u32 a quarter)
    ox = src[o + 0];
    oy = src[o + 1];
    oz = src[o + 2];
  }
  if (tid == 0 && progress) __hip_atomic_store(progress, m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Cloud -> LDS by the whole workgroup (ends on a barrier); returns non-zero (to every thread) unless ALL points of the
// cloud are bit-identical to its first point -- a CONSTANT cloud, the dataset's padding slot
// (dataset_wrapper.py:156-158): the distinct-row kernels of sa_split.hip then multiply one row for it.  Compared as bit
// patterns: +0 / -0 or two NaN payloads are different points.
__device__ __forceinline__ int stage_cloud(const float *__restrict__ P, float *sx, int count, int ps, int tid, int threads) {
  int differs = 0;
  if ((reinterpret_cast<uintptr_t>(P) & 15u) == 0 && (count & 3) == 0 && ps <= 12 && 12 % ps == 0) {
    // 16 bytes a lane, up to eight fetches in flight.  The first point repeated is a pattern of period 12 floats
    // (ps divides 12): three different float4, chosen by the vector's index mod 3.
    const float4 *P4 = reinterpret_cast<const float4 *>(P);
    float4 *s4 = reinterpret_cast<float4 *>(sx);          // (16-byte aligned: kFpsLdsFixed and the fused kernel's header)
    const int c4 = count >> 2;
    float f[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) f[e] = P[e % ps];        // (e % ps < count: count >= ps)
    for (int i0 = tid; i0 < c4; i0 += 8 * threads) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * threads < c4) v[u] = P4[i0 + u * threads];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * threads;
        if (i < c4) {
          s4[i] = v[u];
          const int ph = i % 3;
          const float a0 = ph == 0 ? f[0] : ph == 1 ? f[4] : f[8], a1 = ph == 0 ? f[1] : ph == 1 ? f[5] : f[9];
          const float a2 = ph == 0 ? f[2] : ph == 1 ? f[6] : f[10], a3 = ph == 0 ? f[3] : ph == 1 ? f[7] : f[11];
          differs |= (__float_as_int(v[u].x) ^ __float_as_int(a0)) | (__float_as_int(v[u].y) ^ __float_as_int(a1)) |
                     (__float_as_int(v[u].z) ^ __float_as_int(a2)) | (__float_as_int(v[u].w) ^ __float_as_int(a3));
        }
      }
    }
    return __syncthreads_or(differs);
  }
  for (int i = tid; i < count; i += threads) {
    const float v = P[i];
    sx[i] = v;
    differs |= __float_as_int(v) ^ __float_as_int(P[i % ps]);
  }
  return __syncthreads_or(differs);
}

// One cloud per block.  pts: (b, n, ps) f32.  Optional second level (m2 > 0): FPS over the
// m winners of the first level (what the next set-abstraction level does), same launch,
// wave 0 only; requires m <= 64.
template <int PPT, int NW, bool STAGE>
__global__ __launch_bounds__(kWave * NW) void fps_kernel(
    int n, int ps, int m, int bs, int log2bs, int q, const float *__restrict__ pts,
    int *__restrict__ idxs, float *__restrict__ new_xyz,
    int m2, int bs2, int log2bs2, int q2, int *__restrict__ idxs2, float *__restrict__ new_xyz2,
    const unsigned char *__restrict__ valid, unsigned char *__restrict__ constant_out) {
  if (valid && !valid[blockIdx.x]) return;                // padding object: nothing downstream reads it
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *red_bits = reinterpret_cast<int *>(smem);          // [2][NW]
  int *red_k = red_bits + 2 * NW;                         // [2][NW]
  float *keep = reinterpret_cast<float *>(red_k + 2 * NW);  // [64*3] winners (level-2 input)
  float *sx = keep + 64 * 3;                              // [n*ps] when STAGE

  const int obj = blockIdx.x;
  const int tid = threadIdx.x;
  const float *P = pts + (size_t)obj * n * ps;
  if (STAGE) {
    const int differs = stage_cloud(P, sx, n * ps, ps, tid, kWave * NW);
    if (tid == 0 && constant_out) constant_out[obj] = differs ? 0 : 1;
  } else if (tid == 0 && constant_out) {
    constant_out[obj] = 0;                                // (a cloud too large to stage is not examined: "not constant")
  }
  const float *src = STAGE ? sx : P;
  fps_level<PPT, NW>(src, ps, n, m, bs, log2bs, q, red_bits, red_k,
                     idxs ? idxs + (size_t)obj * m : nullptr,
                     new_xyz ? new_xyz + (size_t)obj * m * 3 : nullptr, m2 > 0 ? keep : nullptr);
  if (m2 > 0) {
    __syncthreads();                 // `keep` written by thread 0
    if (tid < kWave) {
      fps_level<1, 1>(keep, 3, m, m2, bs2, log2bs2, q2, red_bits, red_k,
                      idxs2 ? idxs2 + (size_t)obj * m2 : nullptr,
                      new_xyz2 ? new_xyz2 + (size_t)obj * m2 * 3 : nullptr, nullptr);
    }
  }
}

constexpr size_t kFpsLdsFixed(int NW) { return sizeof(int) * 4 * NW + sizeof(float) * 64 * 3; }

template <int PPT, int NW>
inline hipError_t launch_fps(int b, const FpsShape &s, int ps, int m, const float *pts, int *idx,
                             float *new_xyz, int m2, int *idx2, float *new_xyz2,
                             hipStream_t st, const unsigned char *valid, unsigned char *constant_out) {
  const size_t cloud = (size_t)s.n * ps * sizeof(float);
  const bool stage = cloud <= 64 * 1024;
  int bs2 = 1, log2bs2 = 0, q2 = 1;      // the second level's launch of the reference: m points, its own block size
  if (m2 > 0) {
    const FpsShape s2 = fps_shape(m);
    bs2 = s2.bs;
    log2bs2 = s2.log2bs;
    q2 = s2.q;
  }
  if (stage) {
    fps_kernel<PPT, NW, true><<<b, kWave * NW, kFpsLdsFixed(NW) + cloud, st>>>(
        s.n, ps, m, s.bs, s.log2bs, s.q, pts, idx, new_xyz, m2, bs2, log2bs2, q2, idx2, new_xyz2, valid, constant_out);
  } else {
    fps_kernel<PPT, NW, false><<<b, kWave * NW, kFpsLdsFixed(NW), st>>>(
        s.n, ps, m, s.bs, s.log2bs, s.q, pts, idx, new_xyz, m2, bs2, log2bs2, q2, idx2, new_xyz2, valid, constant_out);
  }
  return hipGetLastError();
}

// One cloud per block, FPS by wave 0 as above (one wave, <= 1024 rank slots, cloud staged) -- and BESIDE it the ball query
// of the level whose centres the FPS picks (ball_query_gpu.cu:9-44, the loop of ball_query_kernel below): QW more waves
// take the winners as wave 0 publishes them (a counter in LDS, release / acquire at workgroup scope) and each scans the
// staged cloud for its centre.  FPS is a dependent chain per pick (~0.5 us: 31 us for the launch of its own) that leaves
// the CU's other wave slots idle; the query fits under it.
// Round 6 (49.8 -> 35.0 us for 960 clouds of 1024 points; stamps of the FPS wave, 2.15 GHz: staging 8.0 -> 5.2 k cycles,
// level 1 36 k = 31 picks of ~1.15 k, level 2 7.5 k = 15 picks of ~0.5 k -- a pick of ONE point a lane is ~500 cycles of
// reduce -> readlane -> LDS latency): the launch as a whole was instruction-issue-bound on the query waves, which issued
// four times the FPS wave's instructions.  (a) the query in rounds of 256 points, packed distances, slots by v_mbcnt into
// a row in LDS and one row store: -12.6 us; (b) the FPS scan packed, fminf without its canonicalising v_max: in the
// same step; (c) the cloud staged with 16-byte loads, eight in flight: -1.8; (d) point pairs fetched as pairs
// (ds_read2st64_b32): -1.0.  Measured and without effect here: the winner's bookkeeping deferred past the scan, one
// or two query waves (60 / 43 us), five (49).
// What a caller may hang on the sampling launch (sa_split.hip: the level-1 / level-2 PLANS of the distinct-row kernels):
//   kOn            all m rows of the object stay in LDS (m * nsample words) instead of one row a query wave
//   kLdsInts       words of LDS behind the rows that are the plan's own
//   skipped(obj, lane)                                   wave 0 of an object the valid mask skips
//   after_queries(obj, lane, m, nsample, rows, is_const)   ONE query wave, once every row of the object is complete
//                  (rows: LDS; beside the FPS wave's second level)
//   after_sampling(obj, lane, m, m2, keep, keep2, scratch, is_const)   the FPS wave, after the second level (keep / keep2:
//                  the two levels' winners in LDS, packed xyz; scratch: kLdsInts words)
struct FpsNoPlan {
  static constexpr bool kOn = false;
  static constexpr int kLdsInts = 0;
  __device__ void skipped(int, int) const {}
  __device__ void after_queries(int, int, int, int, const int *, bool) const {}
  __device__ void after_sampling(int, int, int, int, const float *, const float *, int *, bool) const {}
};

template <int PPT, int QW, int PS, class Plan>      // PS: the point stride when it is 3 or 6 (constant LDS offsets), 0 = `ps_arg`
__device__ __forceinline__ void fps_query_body(
    int n, int ps_arg, int m, int bs, int log2bs, int q, const float *__restrict__ pts,
    int *__restrict__ idxs, float *__restrict__ new_xyz,
    int m2, int bs2, int log2bs2, int q2, int *__restrict__ idxs2, float *__restrict__ new_xyz2,
    const unsigned char *__restrict__ valid, float radius2, int nsample, int *__restrict__ ball_idx,
    unsigned char *__restrict__ constant_out, char *smem, const Plan &plan) {
  const int obj = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  if (valid && !valid[obj]) {
    if (Plan::kOn && wave == 0) plan.skipped(obj, lane);
    return;
  }
  const int ps = PS ? PS : ps_arg;
  int *red_bits = reinterpret_cast<int *>(smem);          // [2] (unused with one FPS wave)
  int *red_k = red_bits + 2;                              // [2]
  int *progress = red_k + 2;                              // winners published so far
  float *keep = reinterpret_cast<float *>(progress + 4);  // [64 * 3] winners
  float *sx = keep + 64 * 3;                              // [n * ps]
  int *rows = reinterpret_cast<int *>(sx + (size_t)n * ps);            // [QW][nsample], or [m][nsample] with a plan
  int *extra = rows + (size_t)(Plan::kOn ? m : QW) * nsample;          // plan: done | - | - | - | keep2 [16 * 3] | scratch
  float *keep2 = reinterpret_cast<float *>(extra + 4);
  int *scratch = extra + 4 + 48;
  const float *P = pts + (size_t)obj * n * ps;
  if (tid == 0) {
    *progress = 0;
    if (Plan::kOn) extra[0] = 0;
  }
  const int differs = stage_cloud(P, sx, n * ps, ps, tid, kWave * (1 + QW));
  if (tid == 0 && constant_out) constant_out[obj] = differs ? 0 : 1;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);                        // the chain everything waits for
    fps_level<PPT, 1>(sx, ps, n, m, bs, log2bs, q, red_bits, red_k, idxs ? idxs + (size_t)obj * m : nullptr,
                      new_xyz ? new_xyz + (size_t)obj * m * 3 : nullptr, keep, progress);
    if (m2 > 0)                                           // (one wave: its own LDS writes are in order, no barrier)
      fps_level<1, 1>(keep, 3, m, m2, bs2, log2bs2, q2, red_bits, red_k, idxs2 ? idxs2 + (size_t)obj * m2 : nullptr,
                      new_xyz2 ? new_xyz2 + (size_t)obj * m2 * 3 : nullptr, Plan::kOn ? keep2 : nullptr);
    if (Plan::kOn) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      plan.after_sampling(obj, lane, m, m2, keep, keep2, scratch, differs == 0);
    }
    return;
  }
  // A query wave's centre: 256 points a round (four chunks of 64 in index order; the lane's four points are fetched
  // together and their distances are two packed chains), a ballot per chunk, the hits' slots from the lanes below
  // (v_mbcnt) into the centre's row in LDS; the row leaves as ONE store of nsample consecutive words with the fill
  // (its slot 0 is the first hit: ball_query_gpu.cu:35-39).
  const unsigned lds_sx = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)sx;   // sx as an LDS byte address
  int at[4];                                              // the lane's four points of round 0 (float offsets in sx)
#pragma unroll
  for (int c = 0; c < 4; ++c) at[c] = (c * kWave + lane) * ps;
  for (int j = wave - 1; j < m; j += QW) {
    int *lrow = rows + (size_t)(Plan::kOn ? j : wave - 1) * nsample;
    while (__hip_atomic_load(progress, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(2);
    const float cx = keep[j * 3 + 0], cy = keep[j * 3 + 1], cz = keep[j * 3 + 2];
    const f32x2 c_x = {cx, cx}, c_y = {cy, cy}, c_z = {cz, cz};
    int cnt = 0;
    auto round = [&](int base, auto whole_t) {
      constexpr bool WHOLE = decltype(whole_t)::value;      // every lane's four points exist
      const int off = base * ps;
      float d[4];
      if constexpr (WHOLE && PS != 0) {
        // x of points k and k + 64 as ONE register pair (ds_read2st64_b32: two dwords 64 * PS dwords apart), so the
        // packed chains take the fetched pairs as they arrive (through C the compiler fetches (x, y) of a point
        // together and spends three v_mov a pair of points on re-pairing them).  The waits name what they release.
        const unsigned a0 = lds_sx + 4u * (unsigned)(at[0] + off);
        f32x2 x01, y01, z01, x23, y23, z23;
        asm volatile("ds_read2st64_b32 %0, %1 offset1:%2" : "=v"(x01) : "v"(a0), "n"(PS));
        asm volatile("ds_read2st64_b32 %0, %1 offset1:%2" : "=v"(y01) : "v"(a0 + 4u), "n"(PS));
        asm volatile("ds_read2st64_b32 %0, %1 offset1:%2" : "=v"(z01) : "v"(a0 + 8u), "n"(PS));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(x23) : "v"(a0), "n"(2 * PS), "n"(3 * PS));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(y23) : "v"(a0 + 4u), "n"(2 * PS), "n"(3 * PS));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(z23) : "v"(a0 + 8u), "n"(2 * PS), "n"(3 * PS));
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x01), "+v"(y01), "+v"(z01));
        const f32x2 d01 = sq3x2(c_x - x01, c_y - y01, c_z - z01);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x23), "+v"(y23), "+v"(z23));
        const f32x2 d23 = sq3x2(c_x - x23, c_y - y23, c_z - z23);
        d[0] = d01[0]; d[1] = d01[1]; d[2] = d23[0]; d[3] = d23[1];
      } else {
        float x[4], y[4], z[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int o = WHOLE ? at[c] + off : min(base + c * kWave + lane, n - 1) * ps;
          x[c] = sx[o + 0]; y[c] = sx[o + 1]; z[c] = sx[o + 2];
        }
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const f32x2 x2 = {x[c], x[c + 1]}, y2 = {y[c], y[c + 1]}, z2 = {z[c], z[c + 1]};
          const f32x2 dd = sq3x2(c_x - x2, c_y - y2, c_z - z2);
          d[c] = dd[0]; d[c + 1] = dd[1];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = base + c * kWave + lane;
        const bool hit = WHOLE ? d[c] < radius2 : (k < n && d[c] < radius2);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, cnt));
        if (hit && slot < nsample) lrow[slot] = k;
        cnt += __popcll(mask);
      }
    };
    int base = 0;
    for (; base + 4 * kWave <= n && cnt < nsample; base += 4 * kWave) round(base, std::true_type{});
    if (base < n && cnt < nsample) round(base, std::false_type{});
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int filled = cnt < nsample ? cnt : nsample;
    const int fill = cnt > 0 ? lrow[0] : 0;
    int *row = ball_idx + ((size_t)obj * m + j) * nsample;
    for (int l = lane; l < nsample; l += kWave) {
      const int v = l < filled ? lrow[l] : fill;
      row[l] = v;
      if (Plan::kOn) lrow[l] = v;                          // (the plan reads whole rows)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (the next centre's hits overwrite the row)
    __builtin_amdgcn_wave_barrier();
  }
  if (Plan::kOn) {
    // the last query wave to finish plans over the object's rows (beside the FPS wave's second level)
    int last = 0;
    if (lane == 0) last = __hip_atomic_fetch_add(extra, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == QW - 1;
    if (__builtin_amdgcn_readfirstlane(last)) plan.after_queries(obj, lane, m, nsample, rows, differs == 0);
  }
}

template <int PPT, int QW, int PS>
__global__ __launch_bounds__(kWave * (1 + QW)) void fps_query_kernel(
    int n, int ps_arg, int m, int bs, int log2bs, int q, const float *__restrict__ pts,
    int *__restrict__ idxs, float *__restrict__ new_xyz,
    int m2, int bs2, int log2bs2, int q2, int *__restrict__ idxs2, float *__restrict__ new_xyz2,
    const unsigned char *__restrict__ valid, float radius2, int nsample, int *__restrict__ ball_idx,
    unsigned char *__restrict__ constant_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fps_query_body<PPT, QW, PS>(n, ps_arg, m, bs, log2bs, q, pts, idxs, new_xyz, m2, bs2, log2bs2, q2, idxs2, new_xyz2, valid,
                              radius2, nsample, ball_idx, constant_out, smem, FpsNoPlan{});
}

// LDS of the sampling launch: header + winners + cloud + rows (+ the plan's words)
inline size_t fps_query_lds(int n, int ps, int m, int nsample, int qw, bool plan, int plan_ints) {
  return sizeof(int) * 8 + sizeof(float) * 64 * 3 + (size_t)n * ps * sizeof(float) +
         sizeof(int) * (size_t)(plan ? m : qw) * nsample + (plan ? sizeof(int) * (4 + 48 + (size_t)plan_ints) : 0);
}

// -> hipErrorInvalidValue for a shape the fused kernel does not take (the caller then runs the two launches)
inline hipError_t launch_fps_query(int b, int n, int ps, int m, const float *pts, int *idx, float *new_xyz, int m2,
                                   int *idx2, float *new_xyz2, float radius2, int nsample, int *ball_idx,
                                   hipStream_t st, const unsigned char *valid, unsigned char *constant_out = nullptr) {
  const FpsShape s = fps_shape(n);
  const size_t cloud = (size_t)n * ps * sizeof(float);
  if (s.slots > 1024 || s.slots <= 256 || cloud > 48 * 1024 || m > 64 || (m2 > 0 && m2 > m) || !ball_idx || nsample <= 0 || nsample > 256)
    return hipErrorInvalidValue;
  int bs2 = 1, log2bs2 = 0, q2 = 1;      // the second level's launch of the reference: m points, its own block size
  if (m2 > 0) {
    const FpsShape s2 = fps_shape(m);
    bs2 = s2.bs;
    log2bs2 = s2.log2bs;
    q2 = s2.q;
  }
  constexpr int QW = 3;
  const size_t lds = fps_query_lds(n, ps, m, nsample, QW, false, 0);
#define MSR3D_FQ(PS)                                                                                              \
  fps_query_kernel<16, QW, PS><<<b, kWave * (1 + QW), lds, st>>>(s.n, ps, m, s.bs, s.log2bs, s.q, pts, idx, new_xyz, m2, \
                                                                bs2, log2bs2, q2, idx2, new_xyz2, valid, radius2, nsample, \
                                                                ball_idx, constant_out)
  if (ps == 6) MSR3D_FQ(6);
  else if (ps == 3) MSR3D_FQ(3);
  else MSR3D_FQ(0);
#undef MSR3D_FQ
  return hipGetLastError();
}

// Dispatch on the number of rank slots.  Returns hipErrorInvalidValue above 32768 slots.
inline hipError_t dispatch_fps(int b, int n, int ps, int m, const float *pts, int *idx,
                               float *new_xyz, int m2, int *idx2, float *new_xyz2,
                               hipStream_t st, const unsigned char *valid = nullptr,
                               unsigned char *constant_out = nullptr) {
  const FpsShape s = fps_shape(n);
  if (m2 > 0 && (m > 64 || m2 > m)) return hipErrorInvalidValue;
#define MSR3D_FPS(PPT, NW) \
  return launch_fps<PPT, NW>(b, s, ps, m, pts, idx, new_xyz, m2, idx2, new_xyz2, st, valid, constant_out)
  // (measured: 4 waves x 4 points/lane per cloud times the same as 1 wave x 16 at 960 clouds --
  // the iteration is a dependent chain scan -> reduce -> readlane -> LDS read, not VALU-bound)
  if (s.slots <= 64) MSR3D_FPS(1, 1);
  if (s.slots <= 256) MSR3D_FPS(4, 1);
  if (s.slots <= 1024) MSR3D_FPS(16, 1);
  if (s.slots <= 2048) MSR3D_FPS(16, 2);
  if (s.slots <= 4096) MSR3D_FPS(16, 4);
  if (s.slots <= 8192) MSR3D_FPS(16, 8);
  if (s.slots <= 16384) MSR3D_FPS(16, 16);
  if (s.slots <= 32768) MSR3D_FPS(32, 16);
#undef MSR3D_FPS
  return hipErrorInvalidValue;   // > 32768 rank slots per cloud: not a configuration of this path
}

// =================================================================================
// Ball query (ball_query_gpu.cu:9-44).  The reference gives each centre ONE thread
// that walks all n points.  Here a wave owns a centre and tests 64 points per step:
// ballot -> prefix popcount gives every hit its output slot in index order, and the
// wave stops as soon as nsample hits are placed.  The cloud is staged in LDS once
// per block and shared by the block's centres.
// =================================================================================
template <int NW, bool STAGE>
__global__ __launch_bounds__(kWave * NW) void ball_query_kernel(int n, int m, float radius2,
                                                                int nsample,
                                                                const float *__restrict__ new_xyz,
                                                                const float *__restrict__ xyz,
                                                                int ps, int *__restrict__ idx,
                                                                const unsigned char *__restrict__ valid) {
  if (valid && !valid[blockIdx.x]) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *sx = reinterpret_cast<float *>(smem);
  const int obj = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const float *P = xyz + (size_t)obj * n * ps;   // ps floats per point, xyz first
  if (STAGE) {
    for (int i = tid; i < n * 3; i += kWave * NW) {
      const int p = i / 3;
      sx[i] = P[p * ps + (i - p * 3)];
    }
    __syncthreads();
  }
  const float *src = STAGE ? sx : P;
  const int st = STAGE ? 3 : ps;
  const unsigned long long lt = (1ull << lane) - 1ull;

  for (int j = blockIdx.y * NW + wave; j < m; j += NW * gridDim.y) {
    const float *c = new_xyz + ((size_t)obj * m + j) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    int *row = idx + ((size_t)obj * m + j) * nsample;
    int cnt = 0, first = 0;
    for (int base = 0; base < n && cnt < nsample; base += kWave) {
      const int k = base + lane;
      bool hit = false;
      if (k < n) {
        const float d2 = sq3(cx - src[k * st + 0], cy - src[k * st + 1], cz - src[k * st + 2]);
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
    // slots never reached: the first hit (pre-fill at :32-36) or 0 (host zero-init)
    const int filled = cnt < nsample ? cnt : nsample;
    const int fill = cnt > 0 ? first : 0;
    for (int l = filled + lane; l < nsample; l += kWave) row[l] = fill;
  }
}


inline hipError_t launch_ball_query(int b, int n, int ps, int m, float radius2, int nsample,
                                    const float *new_xyz, const float *xyz, int *idx,
                                    hipStream_t st, const unsigned char *valid = nullptr) {
  constexpr int NW = 4;
  int ysplit = (1024 + b - 1) / b;          // enough blocks to cover 256 CUs at small b
  const int ymax = (m + NW - 1) / NW;
  if (ysplit > ymax) ysplit = ymax;
  if (ysplit < 1) ysplit = 1;
  dim3 grid(b, ysplit);
  const bool stage = (size_t)n * 12 <= 64 * 1024;
  if (stage) {
    ball_query_kernel<NW, true><<<grid, kWave * NW, (size_t)n * 12, st>>>(n, m, radius2, nsample,
                                                                         new_xyz, xyz, ps, idx, valid);
  } else {
    ball_query_kernel<NW, false><<<grid, kWave * NW, 0, st>>>(n, m, radius2, nsample, new_xyz, xyz,
                                                              ps, idx, valid);
  }
  return hipGetLastError();
}

}  // namespace msr3d
