// anchor_front.hip -- the front of the situated encoder for situation_type 'as_object' / 'as_object_add_loc'
// without positional Fourier encoders (configs/leo_3_dataset_pure_txt.yaml's prompter), one launch each way
// (include/msr3d_hip.h: msr3d_anchor_front_fwd / _bwd).  The agent is a token of its own in front of the
// scene's objects (/root/reference/model/ose3d_situation.py:334-353):
//
//     row (b, 0)      v = (anchor_feat + orientation_encoder(fourier(quaternion_b))) + type_embedding[1]
//     row (b, r > 0)  v = (obj_linear_projection(feature) + object_orientation_feat) + type_embedding[0]
//     every row       pos = LayerNorm(loc_layers[0][0](loc6))      (:384-386 with obj_loc_encoding same_0 / same_all)
//                     xin0 = v + pos
//
// The two linear outputs arrive from the GEMM launch in front of this one (the projection over all B L rows of a
// feature matrix whose agent rows are zero; the orientation encoder over the B quaternion rows); the 6-wide
// location layer is computed here (six fmas a channel).  One wave per row, lane l owns channels 4 l .. 4 l + 3
// (rowmath.h); the row also leaves as the first attention block's operand planes (split_mma.h).
// Backward: d v = d0, d pos = d0 (+ d1 + d2: the later layers' inputs, 'same_all'); LayerNorm backward -> the
// location layer's output gradient (its weight gradient is a problem of the weight-gradient launch), and the
// column sums that ARE the gradients of the constant vectors -- over the object rows: type_embedding[0],
// object_orientation_feat and obj_linear_projection.bias; over the agent rows: type_embedding[1] and anchor_feat.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/msr3d_hip.h"
#include "rowmath.h"
#include "split_mma.h"

namespace {

using namespace msr3d;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__global__ __launch_bounds__(256) void anchor_front_fwd_kernel(
    int M, int L, const float *__restrict__ x0, const float *__restrict__ a_ori, const float *__restrict__ anchor_feat,
    const float *__restrict__ type_emb, const float *__restrict__ ori_feat, const float *__restrict__ loc6,
    const float *__restrict__ Wl, const float *__restrict__ bl, const float *__restrict__ gamma,
    const float *__restrict__ beta, float eps, float *__restrict__ pos, float *__restrict__ s_lin,
    float *__restrict__ stats, float *__restrict__ xin0, unsigned short *__restrict__ xp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const int b = row / L, r = row - b * L;
  const int c = 4 * lane;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  // every load of the row before anything waits
  const bool agent = r == 0;
  const float4 base = agent ? ld4(anchor_feat + c) : ld4(x0 + (size_t)row * ROW_D + c);
  const float4 ori = agent ? ld4(a_ori + (size_t)b * ROW_D + c) : (ori_feat ? ld4(ori_feat + c) : z);
  const float4 ty = ld4(type_emb + (agent ? ROW_D : 0) + c);
  float w[4][6];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float2 *wp = reinterpret_cast<const float2 *>(Wl + (size_t)(c + q) * 6);
    const float2 w0 = wp[0], w1 = wp[1], w2 = wp[2];
    w[q][0] = w0.x; w[q][1] = w0.y; w[q][2] = w1.x; w[q][3] = w1.y; w[q][4] = w2.x; w[q][5] = w2.y;
  }
  const float4 bias = ld4(bl + c), g = ld4(gamma + c), be = ld4(beta + c);
  float lc[6];
  {
    const float2 *lp = reinterpret_cast<const float2 *>(loc6 + (size_t)row * 6);
    const float2 l0 = lp[0], l1 = lp[1], l2 = lp[2];
    lc[0] = l0.x; lc[1] = l0.y; lc[2] = l1.x; lc[3] = l1.y; lc[4] = l2.x; lc[5] = l2.y;
  }
  float lin[4] = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < 6; ++k) lin[q] = fmaf(lc[k], w[q][k], lin[q]);
  const float4 s = make_float4(lin[0], lin[1], lin[2], lin[3]);
  float mean, rstd;
  const float4 p = row_ln(s, g, be, eps, mean, rstd);
  const float4 v = f4_add(f4_add(base, ori), ty);       // (feat + orientation) + type, as ose3d_situation.py:352-353
  const float4 a = f4_add(v, p);
  const size_t o = (size_t)row * ROW_D + c;
  st4(pos + o, p);
  st4(s_lin + o, s);
  if (lane == 0) *reinterpret_cast<float2 *>(stats + (size_t)row * 2) = make_float2(mean, rstd);
  st4(xin0 + o, a);
  if (xp) {
    const float f[4] = {a.x, a.y, a.z, a.w};
    uint2 pl[3];
    sm_split4(f, pl);
    unsigned short *d = xp + ((size_t)b * 3 * 64 + r) * ROW_D + c;
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + (size_t)k * 64 * ROW_D) = pl[k];
  }
}

// 16 rows per workgroup, 4 per wave; the waves' column partials meet in LDS, one atomicAdd per column and vector.
__global__ __launch_bounds__(256) void anchor_front_bwd_kernel(
    int M, int L, const float *__restrict__ d0, const float *__restrict__ d1, const float *__restrict__ d2,
    const float *__restrict__ s_lin, const float *__restrict__ stats, const float *__restrict__ gamma,
    float *__restrict__ d_lin, float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ obj_a,
    float *__restrict__ obj_b, float *__restrict__ obj_c, float *__restrict__ agent_a, float *__restrict__ agent_b) {
  __shared__ __attribute__((aligned(16))) float red[4][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r0 = blockIdx.x * 16 + wave * 4;
  float4 x0[4], x1[4], x2[4], sv[4];
  float2 st[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = min(r0 + j, M - 1);
    const size_t o = (size_t)row * ROW_D + c;
    x0[j] = ld4(d0 + o);
    x1[j] = d1 ? ld4(d1 + o) : z;
    x2[j] = d2 ? ld4(d2 + o) : z;
    sv[j] = ld4(s_lin + o);
    st[j] = *reinterpret_cast<const float2 *>(stats + (size_t)row * 2);
  }
  const float4 g = ld4(gamma + c);
  float4 ag = z, ab = z, co = z, ca = z;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = r0 + j;
    if (row >= M) continue;
    const float4 d = f4_add(f4_add(x0[j], x1[j]), x2[j]);
    const float4 dl = row_ln_bwd(d, sv[j], st[j].x, st[j].y, g, ag, ab);
    st4(d_lin + (size_t)row * ROW_D + c, dl);
    if (row % L == 0) ca = f4_add(ca, x0[j]);
    else co = f4_add(co, x0[j]);
  }
  st4(&red[0][wave][c], ag); st4(&red[1][wave][c], ab); st4(&red[2][wave][c], co); st4(&red[3][wave][c], ca);
  __syncthreads();
  const int col = threadIdx.x;
  float sum[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sum[k] = (red[k][0][col] + red[k][1][col]) + (red[k][2][col] + red[k][3][col]);
  if (dgamma) atomicAdd(dgamma + col, sum[0]);
  if (dbeta) atomicAdd(dbeta + col, sum[1]);
  if (obj_a) atomicAdd(obj_a + col, sum[2]);
  if (obj_b) atomicAdd(obj_b + col, sum[2]);
  if (obj_c) atomicAdd(obj_c + col, sum[2]);
  // (a workgroup of 16 rows holds at most one agent row when L >= 16; zero sums are not sent)
  if (sum[3] != 0.f) {
    if (agent_a) atomicAdd(agent_a + col, sum[3]);
    if (agent_b) atomicAdd(agent_b + col, sum[3]);
  }
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
inline bool al8(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 7u) == 0; }

}  // namespace

extern "C" {

int msr3d_anchor_front_fwd(int B, int L, const float *x0, const float *a_ori, const float *anchor_feat,
                           const float *type_emb, const float *ori_feat, const float *loc6, const float *W_loc,
                           const float *b_loc, const float *gamma, const float *beta, float eps, float *pos,
                           float *s_lin, float *stats, float *xin0, unsigned short *planes, msr3d_stream_t stream) {
  if (B < 0 || L < 2 || (planes && L > 64)) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!x0 || !a_ori || !anchor_feat || !type_emb || !loc6 || !W_loc || !b_loc || !gamma || !beta || !pos || !s_lin ||
      !stats || !xin0)
    return MSR3D_EINVAL;
  if (!al16(x0) || !al16(a_ori) || !al16(anchor_feat) || !al16(type_emb) || !al16(ori_feat) || !al8(loc6) ||
      !al8(W_loc) || !al16(b_loc) || !al16(gamma) || !al16(beta) || !al16(pos) || !al16(s_lin) || !al8(stats) ||
      !al16(xin0) || !al8(planes))
    return MSR3D_EINVAL;
  const int M = B * L;
  anchor_front_fwd_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(M, L, x0, a_ori, anchor_feat, type_emb, ori_feat,
                                                                     loc6, W_loc, b_loc, gamma, beta, eps, pos, s_lin,
                                                                     stats, xin0, planes);
  return (int)hipGetLastError();
}

int msr3d_anchor_front_bwd(int B, int L, const float *d0, const float *d1, const float *d2, const float *s_lin,
                           const float *stats, const float *gamma, float *d_lin, float *dgamma, float *dbeta,
                           float *obj_sum_a, float *obj_sum_b, float *obj_sum_c, float *agent_sum_a,
                           float *agent_sum_b, msr3d_stream_t stream) {
  if (B < 0 || L < 2) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!d0 || !s_lin || !stats || !gamma || !d_lin) return MSR3D_EINVAL;
  if (!al16(d0) || !al16(d1) || !al16(d2) || !al16(s_lin) || !al8(stats) || !al16(gamma) || !al16(d_lin))
    return MSR3D_EINVAL;
  const int M = B * L;
  anchor_front_bwd_kernel<<<(M + 15) / 16, 256, 0, (hipStream_t)stream>>>(M, L, d0, d1, d2, s_lin, stats, gamma, d_lin,
                                                                       dgamma, dbeta, obj_sum_a, obj_sum_b, obj_sum_c,
                                                                       agent_sum_a, agent_sum_b);
  return (int)hipGetLastError();
}

}  // extern "C"
