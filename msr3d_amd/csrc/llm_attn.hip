// llm_attn.hip -- the causal self-attention of a LoRA-Llama decoder layer as three fused kernels (forward, dQ, dK / dV)
// (/root/reference/model/msr3d/msr3d.py:409-415: the LLM forward under bf16 autocast, whose
// attention is transformers' LlamaAttention with a causal + key-padding mask):
//
//     P = softmax(scale q k^T + causal + padding)      o = P v            per (sequence, head), D = 64 or 128
//
// Round 4 ran this as seven batched GEMM launches, two softmax launches and six transposes per layer through 85 MB
// of fp32 scores (473 us forward + backward at 4 x 576 tokens: 22 % of the full step).  Here the scores never leave the
// registers: a workgroup owns 64 query rows (forward, dQ) or 64 keys (dK / dV), one wave 16 of them, and walks the
// 64-wide blocks of the other side that the causal mask leaves; the forward keeps the row's log-sum-exp, the two
// backward kernels recompute P from it (no dQ atomics: each output row has one owner, bit-reproducible).
//
// Every product is on v_mfma_f32_16x16x32_bf16 with the ROLES chosen so that a lane's accumulator elements belong to
// ONE row of the side the workgroup owns (D = A B^T with the owned side as B): the row's running maximum / sum / delta
// are then per-lane scalars, and the probabilities a lane holds (four consecutive positions of the other side per
// 16-wide tile) ARE its share of the next product's B fragment if that product's contraction index is numbered
// (32 s + 16 (e >> 2) + 4 g + (e & 3)) -- the A side (v^T, k^T, dO^T, q^T) is read from a row-major LDS tile with the
// LDS transpose read (ds_read_b64_tr_b16: a 16-lane group hands in a [4 positions][16 columns] block, lane i receives
// column i), which delivers exactly that numbering.  No transposes in memory, no P / dS round trip through LDS.
//
// q, k, v, o, dO, dq, dk, dv: (B, T, H, D) bf16, token-major (row stride ld = H D, head h at column h D) -- the
// projections' own output layout; RoPE already applied to q and k.  lse, delta: (B, H, T) fp32 (log2 domain).
// keep: (B, T) bytes, 0 = padded key (or NULL).  A query row with no visible key gives o = 0 and lse = +inf (P = 0).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/msr3d_hip.h"

namespace {

using u16 = unsigned short;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using v4s = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ u16 f2bf(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ v4s tr_read(const u16 *p) {                       // ds_read_b64_tr_b16
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)p);
}
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
// the four lanes (g = 0..3) that hold one row's values: all-reduce
__device__ __forceinline__ float row_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float row_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

struct FA {
  int B, T, H, ld;
  const u16 *q, *k, *v;
  u16 *o;
  float *lse;
  const unsigned char *keep;
  float c;                                  // scale * log2(e)
  // backward
  const u16 *dout;
  float *delta;
  u16 *dq, *dk, *dv;
  float scale;
};

constexpr int BLK = 64;                     // rows of a block (both sides)

// a 64 x D bf16 tile, rows `ld` apart in memory -> registers (D / 32 16-byte pieces per thread of 256)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;   // (a native vector: HIP's uint4 is a struct, whose copies
                                                               // through an array reference stay memcpys into scratch)
template <int D>
using Tile = u32x4[D / 32];
template <int D>
__device__ __forceinline__ void tile_fetch(u32x4 (&t)[D / 32], const u16 *base, int ld, int tid) {
  constexpr int CPR = D / 8;                // pieces per row
#pragma unroll
  for (int u = 0; u < D / 32; ++u) {
    const int c = tid + 256 * u, row = c / CPR, col = (c % CPR) * 8;
    t[u] = *reinterpret_cast<const u32x4 *>(base + (size_t)row * ld + col);
  }
}
template <int D>
__device__ __forceinline__ void tile_store(const u32x4 (&t)[D / 32], u16 *lds, int tid) {
  constexpr int CPR = D / 8, P = D + 8;
#pragma unroll
  for (int u = 0; u < D / 32; ++u) {
    const int c = tid + 256 * u, row = c / CPR, col = (c % CPR) * 8;
    *reinterpret_cast<u32x4 *>(lds + row * P + col) = t[u];
  }
}

// pack (a[0..3], b[0..3]) -> eight bf16 in fragment order e = 0..7
__device__ __forceinline__ bf16x8 pack_frag(const float (&a)[4], const float (&b)[4]) {
  union { unsigned w[4]; bf16x8 v; } r;
  r.w[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
  r.w[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
  r.w[2] = (unsigned)f2bf(b[0]) | ((unsigned)f2bf(b[1]) << 16);
  r.w[3] = (unsigned)f2bf(b[2]) | ((unsigned)f2bf(b[3]) << 16);
  return r.v;
}

// acc[dt] += X^T (D x 64 positions, from the row-major LDS tile xs) * frag[s]: the transposed A side by tr reads
template <int D>
__device__ __forceinline__ void mma_transposed(f32x4 (&acc)[D / 16], const u16 *xs, const bf16x8 (&frag)[2], int lane) {
  constexpr int P = D + 8;
  const int i = lane & 15, g = lane >> 4;
  const u16 *base = xs + (4 * g + (i >> 2)) * P + 4 * (i & 3);
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { v4s h[2]; bf16x8 v; } a;
      a.h[0] = tr_read(base + (32 * s) * P + 16 * dt);
      a.h[1] = tr_read(base + (32 * s + 16) * P + 16 * dt);
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, frag[s], acc[dt], 0, 0, 0);
    }
}

// s[t] = X (64 positions x D, row-major LDS tile) * own^T: s[t][r] = <X[16 t + 4 g + r], own row of this lane>
template <int D>
__device__ __forceinline__ void mma_rows(f32x4 (&s)[4], const u16 *xs, const bf16x8 (&own)[D / 32], int lane) {
  constexpr int P = D + 8;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8 *>(xs + (16 * t + i) * P + 32 * ks + 8 * g);
      s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, own[ks], s[t], 0, 0, 0);
    }
  }
}

// this lane's own row (position `row` of the block at `base`) as B fragments: D / 32 x 16 bytes
template <int D>
__device__ __forceinline__ void own_fetch(bf16x8 (&f)[D / 32], const u16 *rowp, int g) {
#pragma unroll
  for (int ks = 0; ks < D / 32; ++ks) f[ks] = *reinterpret_cast<const bf16x8 *>(rowp + 32 * ks + 8 * g);
}

// write acc (x mul) as bf16: lane (j, g) holds columns 16 dt + 4 g + r of its row
template <int D>
__device__ __forceinline__ void own_store(const f32x4 (&acc)[D / 16], float mul, u16 *rowp, int g) {
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
    *reinterpret_cast<uint2 *>(rowp + 16 * dt + 4 * g) =
        make_uint2((unsigned)f2bf(acc[dt][0] * mul) | ((unsigned)f2bf(acc[dt][1] * mul) << 16),
                   (unsigned)f2bf(acc[dt][2] * mul) | ((unsigned)f2bf(acc[dt][3] * mul) << 16));
}

// workgroup -> (sequence, head, block): the blocks with the most work (causal: the LAST query block, the FIRST key
// block) are handed out first
__device__ __forceinline__ void decode(int B, int T, int H, bool heavy_last, int &b, int &h, int &blk) {
  const int nb = T / BLK;
  const int x = blockIdx.x / (B * H), bh = blockIdx.x % (B * H);
  blk = heavy_last ? nb - 1 - x : x;
  b = bh / H;
  h = bh % H;
}

// =====================================================================================================
// forward: workgroup = 64 query rows; online softmax over the key blocks 0 .. own
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel(const FA a) {
  constexpr int P = D + 8, NDT = D / 16;
  __shared__ __attribute__((aligned(16))) u16 ks_[BLK * P];
  __shared__ __attribute__((aligned(16))) u16 vs_[BLK * P];
  int b, h, qb;
  decode(a.B, a.T, a.H, true, b, h, qb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int qrow = qb * BLK + 16 * wave + j;                   // this lane's query (position in the sequence)
  const size_t seq = (size_t)b * a.T;
  const u16 *kbase = a.k + seq * a.ld + (size_t)h * D, *vbase = a.v + seq * a.ld + (size_t)h * D;
  bf16x8 qf[D / 32];
  own_fetch<D>(qf, a.q + (seq + qrow) * a.ld + (size_t)h * D, g);
  const unsigned char *keep = a.keep ? a.keep + seq : nullptr;
  f32x4 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  for (int kb = 0; kb <= qb; ++kb) {
    const int k0 = kb * BLK;
    {   // no register prefetch: 128 registers a lane = four workgroups per CU, whose round trips hide one another's
      Tile<D> kt, vt;
      tile_fetch<D>(kt, kbase + (size_t)k0 * a.ld, a.ld, tid);
      tile_fetch<D>(vt, vbase + (size_t)k0 * a.ld, a.ld, tid);
      __syncthreads();                                         // the previous block's readers are done
      tile_store<D>(kt, ks_, tid);
      tile_store<D>(vt, vs_, tid);
    }
    __syncthreads();
    f32x4 s[4];
    mma_rows<D>(s, ks_, qf, lane);                             // s[t][r] = <k[k0 + 16 t + 4 g + r], q[qrow]>
    float x[4][4], mloc = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned kp = keep ? *reinterpret_cast<const unsigned *>(keep + k0 + 16 * t + 4 * g) : 0x01010101u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * g + r;
        const bool vis = key <= qrow && ((kp >> (8 * r)) & 0xffu);
        x[t][r] = vis ? s[t][r] * a.c : -INFINITY;
        mloc = fmaxf(mloc, x[t][r]);
      }
    }
    mloc = row_max(mloc);
    const float m_new = fmaxf(m_run, mloc);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;      // (nothing visible yet: every p below is exp2(-inf) = 0)
    const float alpha = ex2(m_run - m_use);
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { x[t][r] = ex2(x[t][r] - m_use); lsum += x[t][r]; }
    l_run = l_run * alpha + row_sum(lsum);
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha; }
    const bf16x8 pf[2] = {pack_frag(x[0], x[1]), pack_frag(x[2], x[3])};
    mma_transposed<D>(oacc, vs_, pf, lane);                    // o[qrow][d] += sum_key p[key] v[key][d]
  }
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  own_store<D>(oacc, inv, a.o + (seq + qrow) * a.ld + (size_t)h * D, g);
  if (g == 0) a.lse[((size_t)b * a.H + h) * a.T + qrow] = l_run > 0.f ? m_run + __log2f(l_run) : INFINITY;
}

// =====================================================================================================
// dQ: workgroup = 64 query rows, key blocks 0 .. own; also writes delta = rowsum(dO o) for the dK / dV kernel
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(256, 3) void attn_dq_kernel(const FA a) {
  constexpr int P = D + 8, NDT = D / 16;
  __shared__ __attribute__((aligned(16))) u16 ks_[BLK * P];
  __shared__ __attribute__((aligned(16))) u16 vs_[BLK * P];
  int b, h, qb;
  decode(a.B, a.T, a.H, true, b, h, qb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int qrow = qb * BLK + 16 * wave + j;
  const size_t seq = (size_t)b * a.T;
  const u16 *kbase = a.k + seq * a.ld + (size_t)h * D, *vbase = a.v + seq * a.ld + (size_t)h * D;
  const size_t rowoff = (seq + qrow) * a.ld + (size_t)h * D;
  bf16x8 qf[D / 32], dof[D / 32];
  own_fetch<D>(qf, a.q + rowoff, g);
  own_fetch<D>(dof, a.dout + rowoff, g);
  float delta;
  {
    bf16x8 of[D / 32];
    own_fetch<D>(of, a.o + rowoff, g);
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
      union { bf16x8 v; unsigned w[4]; } x, y;
      x.v = of[ks]; y.v = dof[ks];
#pragma unroll
      for (int e = 0; e < 4; ++e) d += bf_lo(x.w[e]) * bf_lo(y.w[e]) + bf_hi(x.w[e]) * bf_hi(y.w[e]);
    }
    delta = row_sum(d);
  }
  const size_t stat = ((size_t)b * a.H + h) * a.T + qrow;
  if (g == 0) a.delta[stat] = delta;
  const float lse = a.lse[stat];
  const unsigned char *keep = a.keep ? a.keep + seq : nullptr;
  f32x4 acc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb <= qb; ++kb) {
    const int k0 = kb * BLK;
    {   // (no register prefetch: see the forward kernel)
      Tile<D> kt, vt;
      tile_fetch<D>(kt, kbase + (size_t)k0 * a.ld, a.ld, tid);
      tile_fetch<D>(vt, vbase + (size_t)k0 * a.ld, a.ld, tid);
      __syncthreads();
      tile_store<D>(kt, ks_, tid);
      tile_store<D>(vt, vs_, tid);
    }
    __syncthreads();
    f32x4 s[4], dp[4];
    mma_rows<D>(s, ks_, qf, lane);                             // <k[key], q[qrow]>
    mma_rows<D>(dp, vs_, dof, lane);                           // <v[key], dO[qrow]> = dP[qrow][key]
    float ds[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned kp = keep ? *reinterpret_cast<const unsigned *>(keep + k0 + 16 * t + 4 * g) : 0x01010101u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * t + 4 * g + r;
        const bool vis = key <= qrow && ((kp >> (8 * r)) & 0xffu);
        const float p = vis ? ex2(s[t][r] * a.c - lse) : 0.f;
        ds[t][r] = p * (dp[t][r] - delta);
      }
    }
    const bf16x8 df[2] = {pack_frag(ds[0], ds[1]), pack_frag(ds[2], ds[3])};
    mma_transposed<D>(acc, ks_, df, lane);                     // dq[qrow][d] += sum_key dS[key] k[key][d]
  }
  own_store<D>(acc, a.scale, a.dq + rowoff, g);
}

// =====================================================================================================
// dK, dV: workgroup = 64 keys, query blocks own .. last
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_dkv_kernel(const FA a) {
  constexpr int P = D + 8, NDT = D / 16;
  __shared__ __attribute__((aligned(16))) u16 qs_[BLK * P];
  __shared__ __attribute__((aligned(16))) u16 os_[BLK * P];     // dO
  __shared__ __attribute__((aligned(16))) float st_[2][2 * BLK];   // [buffer][lse | delta]
  int b, h, kb;
  decode(a.B, a.T, a.H, false, b, h, kb);
  const int nb = a.T / BLK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int key = kb * BLK + 16 * wave + j;                     // this lane's key
  const size_t seq = (size_t)b * a.T;
  const u16 *qbase = a.q + seq * a.ld + (size_t)h * D, *dobase = a.dout + seq * a.ld + (size_t)h * D;
  const size_t rowoff = (seq + key) * a.ld + (size_t)h * D;
  const size_t stat0 = ((size_t)b * a.H + h) * a.T;
  bf16x8 kf[D / 32], vf[D / 32];
  own_fetch<D>(kf, a.k + rowoff, g);
  own_fetch<D>(vf, a.v + rowoff, g);
  const bool kept = a.keep ? a.keep[seq + key] != 0 : true;
  f32x4 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int qb = kb; qb < nb; ++qb) {
    const int q0 = qb * BLK;
    float *st = st_[(qb - kb) & 1];
    {   // (no register prefetch: see the forward kernel)
      Tile<D> qt, ot;
      float stv = 0.f;                                          // threads 0..127: one lse / delta value of the block
      tile_fetch<D>(qt, qbase + (size_t)q0 * a.ld, a.ld, tid);
      tile_fetch<D>(ot, dobase + (size_t)q0 * a.ld, a.ld, tid);
      if (tid < BLK) stv = a.lse[stat0 + q0 + tid];
      else if (tid < 2 * BLK) stv = a.delta[stat0 + q0 + tid - BLK];
      __syncthreads();
      tile_store<D>(qt, qs_, tid);
      tile_store<D>(ot, os_, tid);
      if (tid < 2 * BLK) st[tid] = stv;
    }
    __syncthreads();
    f32x4 s[4], dp[4];
    mma_rows<D>(s, qs_, kf, lane);                              // s[t][r] = <q[q0 + 16 t + 4 g + r], k[key]>
    mma_rows<D>(dp, os_, vf, lane);                             // <dO[query], v[key]> = dP[query][key]
    float p[4][4], ds[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 l4 = *reinterpret_cast<const float4 *>(st + 16 * t + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4 *>(st + BLK + 16 * t + 4 * g);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int query = q0 + 16 * t + 4 * g + r;
        const bool vis = kept && key <= query;
        p[t][r] = vis ? ex2(s[t][r] * a.c - ls[r]) : 0.f;
        ds[t][r] = p[t][r] * (dp[t][r] - dl[r]);
      }
    }
    const bf16x8 pf[2] = {pack_frag(p[0], p[1]), pack_frag(p[2], p[3])};
    const bf16x8 df[2] = {pack_frag(ds[0], ds[1]), pack_frag(ds[2], ds[3])};
    mma_transposed<D>(dv, os_, pf, lane);                       // dv[key][d] += sum_query p[query] dO[query][d]
    mma_transposed<D>(dk, qs_, df, lane);                       // dk[key][d] += sum_query dS[query] q[query][d]
  }
  own_store<D>(dv, 1.0f, a.dv + rowoff, g);
  own_store<D>(dk, a.scale, a.dk + rowoff, g);
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

bool shape_ok(int B, int T, int H, int D, int ld) {
  return B >= 0 && T > 0 && H > 0 && (D == 64 || D == 128) && (T % BLK) == 0 && ld >= H * D && (ld % 8) == 0 &&
         (long long)B * H * (T / BLK) <= 0x7fffffffLL;
}

}  // namespace

extern "C" {

int msr3d_attn_fwd(int B, int T, int H, int D, const void *q, const void *k, const void *v, int ld,
                   const unsigned char *key_keep, float scale, void *out, float *lse, msr3d_stream_t stream) {
  if (!shape_ok(B, T, H, D, ld)) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!q || !k || !v || !out || !lse || !al16(q) || !al16(k) || !al16(v) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
      (key_keep && (reinterpret_cast<uintptr_t>(key_keep) & 3u)))
    return MSR3D_EINVAL;
  FA a{};
  a.B = B; a.T = T; a.H = H; a.ld = ld;
  a.q = (const u16 *)q; a.k = (const u16 *)k; a.v = (const u16 *)v; a.o = (u16 *)out; a.lse = lse; a.keep = key_keep;
  a.c = scale * 1.4426950408889634f;
  const unsigned grid = (unsigned)(B * H * (T / BLK));
  if (D == 128) attn_fwd_kernel<128><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else attn_fwd_kernel<64><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return (int)hipGetLastError();
}

int msr3d_attn_bwd(int B, int T, int H, int D, const void *q, const void *k, const void *v, const void *out,
                   const void *dout, int ld, const unsigned char *key_keep, float scale, const float *lse, float *delta,
                   void *dq, void *dk, void *dv, msr3d_stream_t stream) {
  if (!shape_ok(B, T, H, D, ld)) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!q || !k || !v || !out || !dout || !lse || !delta || !dq || !dk || !dv) return MSR3D_EINVAL;
  if (!al16(q) || !al16(k) || !al16(v) || !al16(out) || !al16(dout) || (reinterpret_cast<uintptr_t>(dq) & 7u) ||
      (reinterpret_cast<uintptr_t>(dk) & 7u) || (reinterpret_cast<uintptr_t>(dv) & 7u) ||
      (key_keep && (reinterpret_cast<uintptr_t>(key_keep) & 3u)))
    return MSR3D_EINVAL;
  FA a{};
  a.B = B; a.T = T; a.H = H; a.ld = ld;
  a.q = (const u16 *)q; a.k = (const u16 *)k; a.v = (const u16 *)v; a.o = (u16 *)const_cast<void *>(out);
  a.lse = const_cast<float *>(lse); a.keep = key_keep;
  a.c = scale * 1.4426950408889634f;
  a.scale = scale;
  a.dout = (const u16 *)dout; a.delta = delta; a.dq = (u16 *)dq; a.dk = (u16 *)dk; a.dv = (u16 *)dv;
  const unsigned grid = (unsigned)(B * H * (T / BLK));
  hipStream_t st = (hipStream_t)stream;
  if (D == 128) {
    attn_dq_kernel<128><<<grid, 256, 0, st>>>(a);               // (writes delta, which the second kernel reads)
    attn_dkv_kernel<128><<<grid, 256, 0, st>>>(a);
  } else {
    attn_dq_kernel<64><<<grid, 256, 0, st>>>(a);
    attn_dkv_kernel<64><<<grid, 256, 0, st>>>(a);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
