// prompter_rows.hip -- the small row-wise pieces around the spatial encoder layers, each ONE launch
// where the module formulation issues a handful (each launch inside the captured step costs ~4.6 us
// whatever it does, profiles/r02_base_trace):
//
//   msr3d_scene_prologue   everything that depends only on the batch's data tensors: inverted key
//                          mask, pairwise spatial features (/root/reference/modules/utils.py:88-137),
//                          agent-frame transform + Fourier features (modules/utils.py:60-82,
//                          model/ose3d_situation.py:31-59), and a copy of obj_locs -- read from the
//                          batch where it lies, written to the step's static buffers
//   msr3d_pos_embed_fwd    pos = LN(Linear(fourier)) + LN(Linear(size))
//                          (loc_embedding_encoder + size_embedding_encoder, ose3d_situation.py:399-404)
//   msr3d_pos_embed_bwd    its row-wise backward: sums the positional term's upstream gradients (one
//                          per layer, 'same_all'), both LayerNorm backwards, LayerNorm parameter
//                          gradients, and the column sums that are the gradients of the two constant
//                          token embeddings (object_type_embedding row 0, object_orientation_feat)
//   msr3d_step_begin       one fill for every split-K meeting point of the step + the dropout seed bump
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/msr3d_hip.h"
#include "rowmath.h"
#include "split_mma.h"

namespace {

using msr3d::f4_add;
using msr3d::row_ln;
using msr3d::row_ln_bwd;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// one block per sample (same arithmetic, in the same order, as prologue.hip's two kernels)
__global__ __launch_bounds__(256) void scene_prologue_kernel(
    int L, const float *__restrict__ loc, const unsigned char *__restrict__ valid,
    const float *__restrict__ anchor_loc, const float *__restrict__ anchor_ori,
    const float *__restrict__ freqs, int nb, int transform, float eps, float *__restrict__ pw,
    float *__restrict__ ff, float *__restrict__ loc_out, unsigned char *__restrict__ pad,
    unsigned char *__restrict__ valid_out, float *__restrict__ anchor_loc_out, float *__restrict__ anchor_ori_out,
    const float *__restrict__ agent_size, float *__restrict__ quat_ff) {
  // agent_size != NULL (msr3d_scene_prologue_agent): token 0 of every scene is the AGENT -- box = [anchor_loc | agent_size],
  // always valid -- and tokens 1 .. L-1 are the scene's L-1 objects (loc, valid: (B, L-1, ..)); quat_ff (B, 4 + 8 nb)
  // receives the Fourier rows of the orientation quaternion (generate_fourier_features of a (B,1,4) input)
  __shared__ float cx[128], cy[128], cz[128];
  __shared__ float wmax[4];
  // grid (B, parts): the blocks of a sample each find the sample's largest distance (cheap, all pairs; a maximum,
  // so the same bits however it is split) and write their share of the pair features and of the Fourier features;
  // block 0 also writes the per-token outputs.  parts = 16 (round 5; 4 before): the launch is 64 -> 256 workgroups and
  // a thread's chain of accurate sinf / cosf calls -- the kernel's longest -- shrinks from four to one
  const int b = blockIdx.x, part = blockIdx.y, parts = gridDim.y, tid = threadIdx.x;
  if (part == 0 && tid < 7) {          // the anchor pose, copied into the step's static buffers in the same launch
    if (tid < 3) { if (anchor_loc_out && anchor_loc) anchor_loc_out[b * 3 + tid] = anchor_loc[b * 3 + tid]; }
    else if (anchor_ori_out && anchor_ori) anchor_ori_out[b * 4 + tid - 3] = anchor_ori[b * 4 + tid - 3];
  }
  const int LO = agent_size ? L - 1 : L, off = agent_size ? 1 : 0;       // objects per scene; first object token
  for (int i = tid; i < L; i += 256) {
    float bx[6];
    unsigned char ok = 1;
    if (i < off) {
      bx[0] = anchor_loc[b * 3]; bx[1] = anchor_loc[b * 3 + 1]; bx[2] = anchor_loc[b * 3 + 2];
      bx[3] = agent_size[0]; bx[4] = agent_size[1]; bx[5] = agent_size[2];
    } else {
      const float *p = loc + ((size_t)b * LO + i - off) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) bx[k] = p[k];
      ok = valid[(size_t)b * LO + i - off] ? 1 : 0;
    }
    cx[i] = bx[0]; cy[i] = bx[1]; cz[i] = bx[2];
    if (part == 0) {
      float *q = loc_out + ((size_t)b * L + i) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) q[k] = bx[k];
      pad[(size_t)b * L + i] = ok ^ 1;
      if (valid_out) valid_out[(size_t)b * L + i] = ok;
    }
  }
  if (quat_ff && part == 1 % parts) {
    // [q (4) | sin(pi q_c f_k), (c, k) flattened (4 nb) | cos(..) (4 nb)]: ose3d_situation.py:31-59 on a (B,1,4) input
    const float pi = 3.14159265358979323846f;
    const int Wq = 4 + 8 * nb;
    for (int e = tid; e < 4 * nb; e += 256) {
      const int c = e / nb, k = e - c * nb;
      const float v = anchor_ori[b * 4 + c];
      float *o = quat_ff + (size_t)b * Wq;
      if (k == 0) o[c] = v;
      const float sarg = pi * (v * freqs[k]);
      o[4 + c * nb + k] = sinf(sarg);
      o[4 + 4 * nb + c * nb + k] = cosf(sarg);
    }
  }
  __syncthreads();
  // Fourier features of the (agent-frame) centres: one (token, coordinate, frequency) per thread and pass over all the
  // sample's blocks -- a thread's chain is one sinf + one cosf (accurate forms, ~100 instructions each), not the 20 it
  // was with one (token, coordinate) per thread
  {
    const int W = 3 + 6 * nb;
    const float pi = 3.14159265358979323846f;
    for (int e = part * 256 + tid; e < L * 3 * nb; e += parts * 256) {
      const int tok = e / (3 * nb), rem = e - tok * 3 * nb, c = rem / nb, k = rem - c * nb;
      float v[3] = {cx[tok], cy[tok], cz[tok]};
      if (transform) {
        const float *a = anchor_loc + (size_t)b * 3;
        const float *q = anchor_ori + (size_t)b * 4;
        const float r0 = v[0] - a[0], r1 = v[1] - a[1], r2 = v[2] - a[2];
        const float x = -q[0], y = -q[1], z = -q[2], w = q[3];
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, xz = x * z, xw = x * w, yz = y * z, yw = y * w, zw = z * w;
        const float R00 = 1.f - 2.f * (yy + zz), R01 = 2.f * (xy + zw), R02 = 2.f * (xz - yw);
        const float R10 = 2.f * (xy - zw), R11 = 1.f - 2.f * (xx + zz), R12 = 2.f * (yz + xw);
        const float R20 = 2.f * (xz + yw), R21 = 2.f * (yz - xw), R22 = 1.f - 2.f * (xx + yy);
        v[0] = (r0 * R00 + r1 * R10) + r2 * R20;
        v[1] = (r0 * R01 + r1 * R11) + r2 * R21;
        v[2] = (r0 * R02 + r1 * R12) + r2 * R22;
      }
      const float vc = c == 0 ? v[0] : c == 1 ? v[1] : v[2];
      float *o = ff + ((size_t)b * L + tok) * W;
      if (k == 0) o[c] = vc;
      const float s = pi * (vc * freqs[k]);
      o[3 + c * nb + k] = sinf(s);
      o[3 + 3 * nb + c * nb + k] = cosf(s);
    }
  }
  float m = 0.f;
  for (int e = tid; e < L * L; e += 256) {
    const int l = e / L, t = e - l * L;
    const float dx = cx[l] - cx[t], dy = cy[l] - cy[t], dz = cz[l] - cz[t];
    m = fmaxf(m, sqrtf(((dx * dx + dy * dy) + dz * dz) + eps));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) wmax[tid >> 6] = m;
  __syncthreads();
  const float dmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  float *O = pw + (size_t)b * L * L * 5;
  const int per = (L * L + parts - 1) / parts, e1 = min(L * L, (part + 1) * per);
  for (int e = part * per + tid; e < e1; e += 256) {
    const int l = e / L, t = e - l * L;
    const float dx = cx[l] - cx[t], dy = cy[l] - cy[t], dz = cz[l] - cz[t];
    const float d = sqrtf(((dx * dx + dy * dy) + dz * dz) + eps);
    const float d2 = sqrtf((dx * dx + dy * dy) + eps);
    float *o = O + (size_t)e * 5;
    o[0] = d / dmax; o[1] = dz / d; o[2] = d2 / d; o[3] = dy / d2; o[4] = dx / d2;
  }
}

__global__ void step_begin_kernel(float4 *__restrict__ z, long long n4, unsigned long long *seed) {
  if (seed && blockIdx.x == 0 && threadIdx.x == 0)
    *seed = *seed * 6364136223846793005ull + 1442695040888963407ull;     // = msr3d_bump_seed
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x)
    z[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// pos = LN_a(f Wa^T + ba) + LN_b(s Wb^T + bb),  f (M, KF <= 64) Fourier features, s = loc[:, 3:6].
// 16 rows per workgroup: thread c accumulates column c of all 16 rows (weights transposed in LDS,
// features broadcast from LDS), then the rows are normalised wave-per-row.
// ------------------------------------------------------------------------------------------------
// PR rows per workgroup, chosen so that ~240 workgroups exist (a workgroup's fixed cost is staging Wa,
// 64 KB out of L2; at 16 rows there were 60 of them doing 4 rows of LayerNorm per wave in sequence).
template <int PR>
__global__ __launch_bounds__(256) void pos_embed_fwd_kernel(
    int M, int KF, const float *__restrict__ ff, const float *__restrict__ loc,
    const float *__restrict__ Wa, const float *__restrict__ ba, const float *__restrict__ ga,
    const float *__restrict__ bta, float epsa, const float *__restrict__ Wb,
    const float *__restrict__ bb, const float *__restrict__ gb, const float *__restrict__ btb,
    float epsb, float *__restrict__ pos, float *__restrict__ sa, float *__restrict__ sta,
    float *__restrict__ sb, float *__restrict__ stb, const float *__restrict__ x0, const float *__restrict__ c0,
    const float *__restrict__ c1, float *__restrict__ xin0, unsigned short *__restrict__ xp, int L) {
  // x0 != NULL (msr3d_pos_embed_tokens_fwd): the first layer's input in the same launch -- xin0 = ((x0 + pos) + c0) + c1
  // (c0 / c1: the constant type / orientation rows, c1 optional), also as the first attention block's operand planes
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *WT = sm;                    // [64][257]  k-major copy of Wa (row stride 257: the transposing
                                     //            stores of consecutive k land in consecutive banks)
  float *F = sm + 64 * 257;          // [PR][64]   the rows' features, zero-padded (64 * 257 % 4 == 0)
  float *T = F + PR * 64;            // [2][PR][256] pre-norm results, both encoders
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * PR;
  // Wa (256, KF) row by row: a wave takes rows wave, wave + 4, ...; lane = k.  Eight row loads in flight,
  // no index arithmetic beyond an add (the flat e -> (row, k) form cost a division per element).
  for (int c0 = wave; c0 < 256; c0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = lane < KF ? Wa[(size_t)(c0 + 4 * u) * KF + lane] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) WT[lane * 257 + c0 + 4 * u] = v[u];
  }
  for (int e = tid; e < PR * 64; e += 256) {
    const int r = e >> 6, k = e & 63;
    F[e] = (r0 + r < M && k < KF) ? ff[(size_t)(r0 + r) * KF + k] : 0.f;
  }
  __syncthreads();
  float acc[PR];
  const float bias = ba[tid];
#pragma unroll
  for (int r = 0; r < PR; ++r) acc[r] = bias;
  for (int k = 0; k < 64; k += 4) {
    const float w0 = WT[(k + 0) * 257 + tid], w1 = WT[(k + 1) * 257 + tid];
    const float w2 = WT[(k + 2) * 257 + tid], w3 = WT[(k + 3) * 257 + tid];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const float4 f = ld4(F + r * 64 + k);
      acc[r] = fmaf(f.w, w3, fmaf(f.z, w2, fmaf(f.y, w1, fmaf(f.x, w0, acc[r]))));
    }
  }
  const float wb0 = Wb[tid * 3], wb1 = Wb[tid * 3 + 1], wb2 = Wb[tid * 3 + 2], bbias = bb[tid];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    T[r * 256 + tid] = acc[r];
    float v = bbias;
    if (r0 + r < M) {
      const float *s = loc + (size_t)(r0 + r) * 6 + 3;
      v = fmaf(s[2], wb2, fmaf(s[1], wb1, fmaf(s[0], wb0, v)));
    }
    T[(PR + r) * 256 + tid] = v;
  }
  __syncthreads();
  const int c = lane * 4;
  const float4 g1 = ld4(ga + c), b1 = ld4(bta + c), g2 = ld4(gb + c), b2 = ld4(btb + c);
  for (int r = wave; r < PR; r += 4) {
    const int row = r0 + r;
    if (row >= M) break;
    const float4 va = ld4(T + r * 256 + c), vb = ld4(T + (PR + r) * 256 + c);
    float m1, s1, m2, s2;
    const float4 ya = row_ln(va, g1, b1, epsa, m1, s1);
    const float4 yb = row_ln(vb, g2, b2, epsb, m2, s2);
    const size_t o = (size_t)row * 256 + c;
    const float4 pv = f4_add(ya, yb);
    st4(pos + o, pv);
    st4(sa + o, va);
    st4(sb + o, vb);
    if (lane == 0) { sta[row * 2] = m1; sta[row * 2 + 1] = s1; stb[row * 2] = m2; stb[row * 2 + 1] = s2; }
    if (x0) {
      // the operation order of msr3d_scene_rows' MSR3D_PRO_ADD: ((a0 + a1) + g1) + b1
      float4 a = f4_add(f4_add(ld4(x0 + o), pv), ld4(c0 + c));
      if (c1) a = f4_add(a, ld4(c1 + c));
      st4(xin0 + o, a);
      if (xp) {
        const int b = row / L, rr = row - b * L;
        const float f[4] = {a.x, a.y, a.z, a.w};
        uint2 pl[3];
        msr3d::sm_split4(f, pl);
        unsigned short *d = xp + ((size_t)b * 3 * 64 + rr) * 256 + c;
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + (size_t)k * 64 * 256) = pl[k];
      }
    }
  }
}

// d pos = d0 + d1 + d2 (the layers' input gradients; NULL = absent); da = LN_a-bwd(d pos),
// db = LN_b-bwd(d pos); colsum(d0) is added to cs1 and cs2 (both optional).  16 rows per workgroup,
// 4 per wave, every load issued before the first use; the waves' column partials meet in LDS.
__global__ __launch_bounds__(256) void pos_embed_bwd_kernel(
    int M, const float *__restrict__ d0, const float *__restrict__ d1, const float *__restrict__ d2,
    const float *__restrict__ sa, const float *__restrict__ sta, const float *__restrict__ ga,
    const float *__restrict__ sb, const float *__restrict__ stb, const float *__restrict__ gb,
    float *__restrict__ da, float *__restrict__ db, float *__restrict__ dga, float *__restrict__ dba,
    float *__restrict__ dgb, float *__restrict__ dbb, float *__restrict__ cs1,
    float *__restrict__ cs2) {
  __shared__ __attribute__((aligned(16))) float red[5][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int r0 = blockIdx.x * 16 + wave * 4;
  float4 x0[4], x1[4], x2[4], va[4], vb[4];
  float2 s1[4], s2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = min(r0 + j, M - 1);
    const size_t o = (size_t)row * 256 + c;
    x0[j] = ld4(d0 + o);
    x1[j] = d1 ? ld4(d1 + o) : z;
    x2[j] = d2 ? ld4(d2 + o) : z;
    va[j] = ld4(sa + o);
    vb[j] = ld4(sb + o);
    s1[j] = *reinterpret_cast<const float2 *>(sta + (size_t)row * 2);
    s2[j] = *reinterpret_cast<const float2 *>(stb + (size_t)row * 2);
  }
  const float4 g1 = ld4(ga + c), g2 = ld4(gb + c);
  float4 ag1 = z, ab1 = z, ag2 = z, ab2 = z, cs = z;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = r0 + j;
    const bool ok = row < M;
    const size_t o = (size_t)row * 256 + c;
    const float4 d = f4_add(f4_add(x0[j], x1[j]), x2[j]);
    float4 t1 = z, t2 = z, t3 = z, t4 = z;
    const float4 ra = row_ln_bwd(d, va[j], s1[j].x, s1[j].y, g1, t1, t2);
    const float4 rb = row_ln_bwd(d, vb[j], s2[j].x, s2[j].y, g2, t3, t4);
    if (ok) {
      st4(da + o, ra);
      st4(db + o, rb);
      ag1 = f4_add(ag1, t1); ab1 = f4_add(ab1, t2); ag2 = f4_add(ag2, t3); ab2 = f4_add(ab2, t4);
      cs = f4_add(cs, x0[j]);
    }
  }
  st4(&red[0][wave][c], ag1); st4(&red[1][wave][c], ab1); st4(&red[2][wave][c], ag2);
  st4(&red[3][wave][c], ab2); st4(&red[4][wave][c], cs);
  __syncthreads();
  float *const dst[6] = {dga, dba, dgb, dbb, cs1, cs2};
  const int col = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float s = (red[k][0][col] + red[k][1][col]) + (red[k][2][col] + red[k][3][col]);
    if (dst[k]) atomicAdd(dst[k] + col, s);
    if (k == 4 && dst[5]) atomicAdd(dst[5] + col, s);
  }
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" {

int msr3d_scene_prologue(int B, int L, const float *obj_locs, const unsigned char *obj_valid,
                         const float *anchor_loc, const float *anchor_ori, const float *freqs,
                         int num_bands, int transform, float eps, float *pairwise_out,
                         float *fourier_out, float *locs_out, unsigned char *pad_out,
                         unsigned char *valid_out, float *anchor_loc_out, float *anchor_ori_out,
                         msr3d_stream_t stream) {
  if (B < 0 || L <= 0 || L > 128 || num_bands <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!obj_locs || !obj_valid || !freqs || !pairwise_out || !fourier_out || !locs_out || !pad_out ||
      (transform && (!anchor_loc || !anchor_ori)))
    return MSR3D_EINVAL;
  scene_prologue_kernel<<<dim3(B, 16), 256, 0, (hipStream_t)stream>>>(L, obj_locs, obj_valid, anchor_loc, anchor_ori,
                                                           freqs, num_bands, transform, eps, pairwise_out,
                                                           fourier_out, locs_out, pad_out, valid_out, anchor_loc_out,
                                                           anchor_ori_out, nullptr, nullptr);
  return (int)hipGetLastError();
}

int msr3d_scene_prologue_agent(int B, int O, const float *obj_locs, const unsigned char *obj_valid,
                               const float *anchor_loc, const float *anchor_ori, const float *agent_size,
                               const float *freqs, int num_bands, float eps, float *pairwise_out, float *fourier_out,
                               float *locs_out, unsigned char *pad_out, unsigned char *valid_out, float *quat_fourier_out,
                               float *anchor_loc_out, float *anchor_ori_out, msr3d_stream_t stream) {
  if (B < 0 || O <= 0 || O + 1 > 128 || num_bands <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!obj_locs || !obj_valid || !anchor_loc || !anchor_ori || !agent_size || !freqs || !pairwise_out || !fourier_out ||
      !locs_out || !pad_out)
    return MSR3D_EINVAL;
  scene_prologue_kernel<<<dim3(B, 16), 256, 0, (hipStream_t)stream>>>(O + 1, obj_locs, obj_valid, anchor_loc, anchor_ori,
                                                           freqs, num_bands, 0, eps, pairwise_out, fourier_out, locs_out,
                                                           pad_out, valid_out, anchor_loc_out, anchor_ori_out, agent_size,
                                                           quat_fourier_out);
  return (int)hipGetLastError();
}

int msr3d_step_begin(float *zero_region, long long n_floats, unsigned long long *seed,
                     msr3d_stream_t stream) {
  if (n_floats < 0 || (n_floats % 4) != 0 || (n_floats > 0 && (!zero_region || !al16(zero_region))))
    return MSR3D_EINVAL;
  if (n_floats == 0 && !seed) return 0;
  const long long n4 = n_floats / 4;
  long long g = (n4 + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  step_begin_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<float4 *>(zero_region), n4, seed);
  return (int)hipGetLastError();
}

static int pos_embed_fwd_launch(int M, int KF, const float *fourier, const float *locs, const float *Wa,
                                const float *ba, const float *gamma_a, const float *beta_a, float eps_a,
                                const float *Wb, const float *bb, const float *gamma_b, const float *beta_b,
                                float eps_b, float *pos, float *s_a, float *stats_a, float *s_b,
                                float *stats_b, const float *x0, const float *c0, const float *c1, float *xin0,
                                unsigned short *xp, int L, msr3d_stream_t stream) {
  if (M < 0 || KF <= 0 || KF > 64) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!fourier || !locs || !Wa || !ba || !gamma_a || !beta_a || !Wb || !bb || !gamma_b || !beta_b || !pos ||
      !s_a || !stats_a || !s_b || !stats_b)
    return MSR3D_EINVAL;
  if (!al16(gamma_a) || !al16(beta_a) || !al16(gamma_b) || !al16(beta_b) || !al16(pos) || !al16(s_a) || !al16(s_b))
    return MSR3D_EINVAL;
#define LAUNCH_POS(PR)                                                                                   \
  do {                                                                                                   \
    const size_t lds = sizeof(float) * (64 * 257 + PR * 64 + 2 * PR * 256);                              \
    static bool done = false;                                                                            \
    if (!done) {                                                                                         \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pos_embed_fwd_kernel<PR>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
      if (e != hipSuccess) return (int)e;                                                                \
      done = true;                                                                                       \
    }                                                                                                    \
    pos_embed_fwd_kernel<PR><<<(M + PR - 1) / PR, 256, lds, (hipStream_t)stream>>>(                      \
        M, KF, fourier, locs, Wa, ba, gamma_a, beta_a, eps_a, Wb, bb, gamma_b, beta_b, eps_b, pos, s_a,  \
        stats_a, s_b, stats_b, x0, c0, c1, xin0, xp, L);                                                 \
  } while (0)
  if (M > 16 * 192) LAUNCH_POS(16); else if (M > 8 * 192) LAUNCH_POS(8); else LAUNCH_POS(4);
#undef LAUNCH_POS
  return (int)hipGetLastError();
}

int msr3d_pos_embed_fwd(int M, int KF, const float *fourier, const float *locs, const float *Wa,
                        const float *ba, const float *gamma_a, const float *beta_a, float eps_a,
                        const float *Wb, const float *bb, const float *gamma_b, const float *beta_b,
                        float eps_b, float *pos, float *s_a, float *stats_a, float *s_b,
                        float *stats_b, msr3d_stream_t stream) {
  return pos_embed_fwd_launch(M, KF, fourier, locs, Wa, ba, gamma_a, beta_a, eps_a, Wb, bb, gamma_b, beta_b, eps_b, pos,
                              s_a, stats_a, s_b, stats_b, nullptr, nullptr, nullptr, nullptr, nullptr, 1, stream);
}

int msr3d_pos_embed_tokens_fwd(int M, int L, int KF, const float *fourier, const float *locs, const float *Wa,
                               const float *ba, const float *gamma_a, const float *beta_a, float eps_a,
                               const float *Wb, const float *bb, const float *gamma_b, const float *beta_b,
                               float eps_b, float *pos, float *s_a, float *stats_a, float *s_b, float *stats_b,
                               const float *x0, const float *type_row, const float *orientation_row, float *xin0,
                               unsigned short *planes, msr3d_stream_t stream) {
  if (!x0 || !type_row || !xin0 || L <= 0 || (planes && L > 64)) return MSR3D_EINVAL;
  if (!al16(x0) || !al16(type_row) || !al16(orientation_row) || !al16(xin0) || (reinterpret_cast<uintptr_t>(planes) & 7u))
    return MSR3D_EINVAL;
  return pos_embed_fwd_launch(M, KF, fourier, locs, Wa, ba, gamma_a, beta_a, eps_a, Wb, bb, gamma_b, beta_b, eps_b, pos,
                              s_a, stats_a, s_b, stats_b, x0, type_row, orientation_row, xin0, planes, L, stream);
}

int msr3d_pos_embed_bwd(int M, const float *d0, const float *d1, const float *d2, const float *s_a,
                        const float *stats_a, const float *gamma_a, const float *s_b,
                        const float *stats_b, const float *gamma_b, float *d_lin_a, float *d_lin_b,
                        float *dgamma_a, float *dbeta_a, float *dgamma_b, float *dbeta_b,
                        float *colsum1, float *colsum2, msr3d_stream_t stream) {
  if (M < 0) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!d0 || !s_a || !stats_a || !gamma_a || !s_b || !stats_b || !gamma_b || !d_lin_a || !d_lin_b)
    return MSR3D_EINVAL;
  if (!al16(d0) || !al16(d1) || !al16(d2) || !al16(s_a) || !al16(s_b) || !al16(d_lin_a) || !al16(d_lin_b) ||
      !al16(gamma_a) || !al16(gamma_b))
    return MSR3D_EINVAL;
  pos_embed_bwd_kernel<<<(M + 15) / 16, 256, 0, (hipStream_t)stream>>>(
      M, d0, d1, d2, s_a, stats_a, gamma_a, s_b, stats_b, gamma_b, d_lin_a, d_lin_b, dgamma_a, dbeta_a,
      dgamma_b, dbeta_b, colsum1, colsum2);
  return (int)hipGetLastError();
}

}  // extern "C"
