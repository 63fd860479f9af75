// bn_train.hip -- BatchNorm in TRAINING mode fused with ReLU, forward and backward, over a
// token-major (rows, C) tensor: the normalisation of the unfrozen backbone's SharedMLP layers
// (/root/reference/modules/third_party/pointnet2/pytorch_utils.py:39-66 BatchNorm2d over
// (b, C, npoint, nsample) == statistics per channel over b*npoint*nsample rows; SURVEY.md §8(f)
// rank 3).  The 1x1 convolutions around it are the token GEMMs of gemm_f32.hip, so in training the
// SharedMLP runs on this build's kernels end to end.
//
// Statistics are two-stage and ordered: every workgroup reduces a chunk of rows into a partial per
// channel (fp32 inside a chunk of MSR3D_BN_CHUNK_ROWS rows), a second stage adds the partials
// in a fixed order in double -- no float atomics, run-to-run bit-identical, and
// the variance E[x^2] - mean^2 is formed in double.
//
//   forward : mean, var (biased) -> y = relu((x - mean) * rstd * gamma + beta);
//             running_mean / running_var updated with `momentum` (unbiased variance, as torch)
//   backward: g = dy * [gamma * xhat + beta > 0];  dbeta = sum g;  dgamma = sum g * xhat;
//             dx = gamma * rstd * (g - dbeta / R - xhat * dgamma / R)
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"

namespace {

constexpr int kChunk = MSR3D_BN_CHUNK_ROWS;     // rows per workgroup of the reduction passes

// grid (chunks); 256 threads = RL row-lanes x C/4 float4 columns (RL = 256 / (C/4), C <= 1024): every
// thread streams float4s of its 4 channels down the chunk's rows, the row-lanes then meet in LDS and
// are added in lane order.  partial[chunk][2][C].
// Upstream gradient of the fused max-pool variant: only the arg-max row of a (group, channel) carries
// one, and only if the pooled value is positive (ReLU).  ns rows per group.
struct Pooled {
  int ns;
  const float4 *dpooled;     // (G, C)
  const float4 *pooled;      // (G, C)
  const int4 *argmax;        // (G, C): row inside the group
};
__device__ __forceinline__ float4 pooled_grad(const Pooled &pg, long long r, int C4, int c4) {
  const long long grp = r / pg.ns;
  const int rloc = (int)(r - grp * pg.ns);
  const int4 a = pg.argmax[grp * C4 + c4];
  const float4 d = pg.dpooled[grp * C4 + c4], v = pg.pooled[grp * C4 + c4];
  float4 g;
  g.x = (a.x == rloc && v.x > 0.f) ? d.x : 0.f;
  g.y = (a.y == rloc && v.y > 0.f) ? d.y : 0.f;
  g.z = (a.z == rloc && v.z > 0.f) ? d.z : 0.f;
  g.w = (a.w == rloc && v.w > 0.f) ? d.w : 0.f;
  return g;
}

template <bool BWD, bool POOLED = false>
__global__ __launch_bounds__(256) void bn_partial_kernel(long long R, int C, const float *__restrict__ x,
                                                         const float *__restrict__ dy,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ rstd,
                                                         float *__restrict__ partial, Pooled pg = Pooled()) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [2][RL][C]
  const int C4 = C >> 2, RL = 256 / C4;
  const int rl = threadIdx.x / C4, c4 = threadIdx.x - rl * C4;
  const long long r0 = (long long)blockIdx.x * kChunk;
  const long long r1 = r0 + kChunk < R ? r0 + kChunk : R;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < RL) {
    float4 mu = s1, rs = s1, ga = s1, be = s1;
    if (BWD) {
      mu = reinterpret_cast<const float4 *>(mean)[c4];
      rs = reinterpret_cast<const float4 *>(rstd)[c4];
      ga = reinterpret_cast<const float4 *>(gamma)[c4];
      be = reinterpret_cast<const float4 *>(beta)[c4];
    }
    const float4 *X = reinterpret_cast<const float4 *>(x);
    const float4 *D = reinterpret_cast<const float4 *>(dy);
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += RL) {
      const float4 v = X[r * C4 + c4];
      if (BWD && POOLED) {
        const float4 g = pooled_grad(pg, r, C4, c4);
#define BNQ(k)                                              \
        {                                                   \
          const float xh = (v.k - mu.k) * rs.k;             \
          s1.k += g.k;                                      \
          s2.k = __builtin_fmaf(g.k, xh, s2.k);             \
        }
        BNQ(x) BNQ(y) BNQ(z) BNQ(w)
#undef BNQ
      } else if (BWD) {
        const float4 d = D[r * C4 + c4];
#define BNP(k)                                                             \
        {                                                                  \
          const float xh = (v.k - mu.k) * rs.k;                            \
          const float g = (__builtin_fmaf(ga.k, xh, be.k) > 0.f) ? d.k : 0.f; \
          s1.k += g;                                                       \
          s2.k = __builtin_fmaf(g, xh, s2.k);                              \
        }
        BNP(x) BNP(y) BNP(z) BNP(w)
#undef BNP
      } else {
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x = __builtin_fmaf(v.x, v.x, s2.x); s2.y = __builtin_fmaf(v.y, v.y, s2.y);
        s2.z = __builtin_fmaf(v.z, v.z, s2.z); s2.w = __builtin_fmaf(v.w, v.w, s2.w);
      }
    }
    reinterpret_cast<float4 *>(red)[rl * C4 + c4] = s1;
    reinterpret_cast<float4 *>(red)[(RL + rl) * C4 + c4] = s2;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < RL; ++l) {            // fixed order
      a += red[l * C + c];
      q += red[(RL + l) * C + c];
    }
    float *p = partial + ((size_t)blockIdx.x * 2) * C + c;
    p[0] = a;
    p[C] = q;
  }
}

// Second stage: 16 channels per workgroup, 16 chunk-lanes per channel (lane l adds chunks l, l + 16, ... in order,
// in double; the sixteen lane sums are then added pairwise in a fixed tree).  (Four lanes per channel and 64
// channels per workgroup left the pass at 17 us for 1,920 partials -- eighteen such launches per step.)
constexpr int kFinCh = 16, kFinLanes = 16;
__device__ __forceinline__ void bn_sum_partials(int C, int chunks, const float *__restrict__ partial, int c,
                                                int l, double (*red)[kFinLanes][kFinCh], double &s, double &q) {
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int k = l; k < chunks; k += kFinLanes) {
      a += (double)partial[((size_t)k * 2) * C + c];
      b += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  const int j = threadIdx.x % kFinCh;
  red[0][l][j] = a;
  red[1][l][j] = b;
  __syncthreads();
  double t[2];
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    double v[kFinLanes];
#pragma unroll
    for (int u = 0; u < kFinLanes; ++u) v[u] = red[w][u][j];
#pragma unroll
    for (int step = 1; step < kFinLanes; step *= 2)
#pragma unroll
      for (int u = 0; u < kFinLanes; u += 2 * step) v[u] += v[u + step];
    t[w] = v[0];
  }
  s = t[0];
  q = t[1];
}

__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(
    long long R, int C, int chunks, const float *__restrict__ partial, float eps, float momentum,
    float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ save_mean,
    float *__restrict__ save_rstd, const float *__restrict__ gamma = nullptr, const float *__restrict__ beta = nullptr,
    float *__restrict__ gamma_out = nullptr, float *__restrict__ beta_out = nullptr) {
  __shared__ double red[2][kFinLanes][kFinCh];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, l = threadIdx.x / kFinCh;
  double s, q;
  bn_sum_partials(C, chunks, partial, c, l, red, s, q);
  if (l != 0 || c >= C) return;
  const double mu = s / (double)R;
  double var = q / (double)R - mu * mu;
  var = var > 0.0 ? var : 0.0;
  save_mean[c] = (float)mu;
  save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (gamma_out) {                 // (the operand-prologue block [gamma | beta | mean | rstd] in one launch)
    gamma_out[c] = gamma[c];
    beta_out[c] = beta[c];
  }
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) {
    const double unbiased = R > 1 ? var * ((double)R / (double)(R - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(int C, int chunks,
                                                              const float *__restrict__ partial,
                                                              float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta,
                                                              float *__restrict__ dgamma_acc,
                                                              float *__restrict__ dbeta_acc) {
  __shared__ double red[2][kFinLanes][kFinCh];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, l = threadIdx.x / kFinCh;
  double s, q;
  bn_sum_partials(C, chunks, partial, c, l, red, s, q);
  if (l != 0 || c >= C) return;
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
  if (dgamma_acc) {                // (the parameters' gradient buffers: added to, no AccumulateGrad launch each)
    dbeta_acc[c] += (float)s;
    dgamma_acc[c] += (float)q;
  }
}

// elementwise, float4 over channels (C % 4 == 0)
__global__ void bn_relu_apply_kernel(long long n4, int C4, const float4 *__restrict__ x,
                                     const float4 *__restrict__ gamma, const float4 *__restrict__ beta,
                                     const float4 *__restrict__ mean, const float4 *__restrict__ rstd,
                                     float4 *__restrict__ y) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % C4);
    const float4 v = x[t], ga = gamma[c4], be = beta[c4], mu = mean[c4], rs = rstd[c4];
    float4 o;
#define BNF(k) o.k = fmaxf(__builtin_fmaf(ga.k, (v.k - mu.k) * rs.k, be.k), 0.f)
    BNF(x); BNF(y); BNF(z); BNF(w);
#undef BNF
    y[t] = o;
  }
}

template <bool POOLED>
__global__ void bn_relu_bwd_apply_kernel(long long n4, int C4, float inv_rows,
                                         const float4 *__restrict__ x, const float4 *__restrict__ dy,
                                         const float4 *__restrict__ gamma, const float4 *__restrict__ beta,
                                         const float4 *__restrict__ mean, const float4 *__restrict__ rstd,
                                         const float4 *__restrict__ dgamma, const float4 *__restrict__ dbeta,
                                         float4 *__restrict__ dx, Pooled pg) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(t % C4);
    const float4 v = x[t], ga = gamma[c4], be = beta[c4], mu = mean[c4], rs = rstd[c4];
    const float4 dg = dgamma[c4], db = dbeta[c4];
    float4 o;
    if (POOLED) {
      const float4 g = pooled_grad(pg, t / C4, C4, c4);
#define BNC(k)                                                                        \
      {                                                                               \
        const float xh = (v.k - mu.k) * rs.k;                                         \
        o.k = ga.k * rs.k * (g.k - db.k * inv_rows - xh * (dg.k * inv_rows));         \
      }
      BNC(x) BNC(y) BNC(z) BNC(w)
#undef BNC
      dx[t] = o;
      continue;
    }
    const float4 d = dy[t];
#define BNB(k)                                                                        \
    {                                                                                 \
      const float xh = (v.k - mu.k) * rs.k;                                           \
      const float g = (__builtin_fmaf(ga.k, xh, be.k) > 0.f) ? d.k : 0.f;             \
      o.k = ga.k * rs.k * (g - db.k * inv_rows - xh * (dg.k * inv_rows));             \
    }
    BNB(x) BNB(y) BNB(z) BNB(w)
#undef BNB
    dx[t] = o;
  }
}

// y = relu(bn(x)) and max over the ns rows of a group in one pass: pooled (G, C), argmax (G, C) =
// FIRST row holding the maximum (F.max_pool2d's choice, pointnet2_modules.py:66-68).  The (R, C)
// activation of the last layer is never written.  256 threads = GL groups x C/4 float4 columns.
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(long long G, int ns, int C4,
                                                              const float4 *__restrict__ x,
                                                              const float4 *__restrict__ gamma,
                                                              const float4 *__restrict__ beta,
                                                              const float4 *__restrict__ mean,
                                                              const float4 *__restrict__ rstd,
                                                              float4 *__restrict__ pooled,
                                                              int4 *__restrict__ argmax,
                                                              float4 *__restrict__ xsel) {
  const int GL = 256 / C4;
  const int gl = threadIdx.x / C4, c4 = threadIdx.x - gl * C4;
  const long long grp = (long long)blockIdx.x * GL + gl;
  if (gl >= GL || grp >= G) return;
  const float4 ga = gamma[c4], be = beta[c4], mu = mean[c4], rs = rstd[c4];
  float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);       // relu(...) >= 0 > -1: row 0 always enters
  float4 sel = best;                                       // x at the arg-max row (the backward's statistics)
  int4 arg = make_int4(0, 0, 0, 0);
  const float4 *X = x + grp * ns * C4 + c4;
#pragma unroll 4
  for (int r = 0; r < ns; ++r) {
    const float4 v = X[(long long)r * C4];
#define BNM(k)                                                                     \
    {                                                                              \
      const float y = fmaxf(__builtin_fmaf(ga.k, (v.k - mu.k) * rs.k, be.k), 0.f); \
      if (y > best.k) { best.k = y; arg.k = r; sel.k = v.k; }                      \
    }
    BNM(x) BNM(y) BNM(z) BNM(w)
#undef BNM
  }
  pooled[grp * C4 + c4] = best;
  argmax[grp * C4 + c4] = arg;
  xsel[grp * C4 + c4] = sel;
}

// Backward statistics of the pooled layer.  Only the arg-max row of a (group, channel) carries a gradient, so
//   dbeta = sum_groups [pooled > 0] dpooled        dgamma = sum_groups [pooled > 0] dpooled * xhat(arg-max row)
// need the (G, C) arrays only -- xsel is x at that row, kept by the forward -- not a pass over the (R, C)
// activation with three group lookups per element (bn_partial_kernel<true, true>: 205 us a layer at 983 k rows).
// Same two-stage ordered scheme: a workgroup reduces gpc groups (the groups of kChunk rows: as many partials as
// the row pass had), the finalize adds the partials in double.
__global__ __launch_bounds__(256) void bn_pooled_partial_kernel(long long G, int C, int gpc,
                                                                const float4 *__restrict__ dpooled,
                                                                const float4 *__restrict__ pooled,
                                                                const float4 *__restrict__ xsel,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ rstd,
                                                                float *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float red[];      // [2][RL][C]
  const int C4 = C >> 2, RL = 256 / C4;
  const int rl = threadIdx.x / C4, c4 = threadIdx.x - rl * C4;
  const long long g0 = (long long)blockIdx.x * gpc;
  const long long g1 = g0 + gpc < G ? g0 + gpc : G;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (rl < RL) {
    const float4 mu = reinterpret_cast<const float4 *>(mean)[c4], rs = reinterpret_cast<const float4 *>(rstd)[c4];
#pragma unroll 4
    for (long long r = g0 + rl; r < g1; r += RL) {
      const float4 d = dpooled[r * C4 + c4], v = pooled[r * C4 + c4], xs = xsel[r * C4 + c4];
#define BNS(k)                                              \
      {                                                     \
        const float g = v.k > 0.f ? d.k : 0.f;              \
        s1.k += g;                                          \
        s2.k = __builtin_fmaf(g, (xs.k - mu.k) * rs.k, s2.k); \
      }
      BNS(x) BNS(y) BNS(z) BNS(w)
#undef BNS
    }
    reinterpret_cast<float4 *>(red)[rl * C4 + c4] = s1;
    reinterpret_cast<float4 *>(red)[(RL + rl) * C4 + c4] = s2;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int l = 0; l < RL; ++l) {            // fixed order
      a += red[l * C + c];
      q += red[(RL + l) * C + c];
    }
    float *p = partial + ((size_t)blockIdx.x * 2) * C + c;
    p[0] = a;
    p[C] = q;
  }
}

inline int chunks_of(long long R) { return (int)((R + kChunk - 1) / kChunk); }
inline size_t partial_lds(int C) { return sizeof(float) * 2 * (size_t)(256 / (C / 4)) * C; }
inline int ew_grid(long long n4) {
  long long g = (n4 + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" {

int msr3d_bn_relu_train_fwd(long long rows, int C, const float *x, const float *gamma,
                            const float *beta, float eps, float momentum, float *running_mean,
                            float *running_var, float *y, float *save_mean, float *save_rstd,
                            float *partial_ws, int partial_chunks, msr3d_stream_t stream) {
  if (rows < 0 || C <= 0 || (C % 4) != 0 || C > 1024 || partial_chunks < 0) return MSR3D_EINVAL;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !y || !save_mean || !save_rstd || !partial_ws) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = partial_chunks ? partial_chunks : chunks_of(rows);
  if (!partial_chunks)
    bn_partial_kernel<false><<<chunks, 256, partial_lds(C), st>>>(rows, C, x, nullptr, nullptr, nullptr, nullptr,
                                                                  nullptr, partial_ws);
  bn_fwd_finalize_kernel<<<(C + kFinCh - 1) / kFinCh, 256, 0, st>>>(rows, C, chunks, partial_ws, eps, momentum,
                                                       running_mean, running_var, save_mean, save_rstd);
  const long long n4 = rows * (C / 4);
  bn_relu_apply_kernel<<<ew_grid(n4), 256, 0, st>>>(
      n4, C / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(gamma),
      reinterpret_cast<const float4 *>(beta), reinterpret_cast<const float4 *>(save_mean),
      reinterpret_cast<const float4 *>(save_rstd), reinterpret_cast<float4 *>(y));
  return (int)hipGetLastError();
}

int msr3d_bn_train_stats(long long rows, int C, const float *partial_ws, int partial_chunks, float eps, float momentum,
                         float *running_mean, float *running_var, const float *gamma, const float *beta, float *bn_block,
                         msr3d_stream_t stream) {
  if (rows <= 0 || C <= 0 || (C % 4) != 0 || C > 1024 || partial_chunks <= 0) return MSR3D_EINVAL;
  if (!partial_ws || !gamma || !beta || !bn_block) return MSR3D_EINVAL;
  bn_fwd_finalize_kernel<<<(C + kFinCh - 1) / kFinCh, 256, 0, (hipStream_t)stream>>>(
      rows, C, partial_chunks, partial_ws, eps, momentum, running_mean, running_var, bn_block + 2 * C, bn_block + 3 * C,
      gamma, beta, bn_block, bn_block + C);
  return (int)hipGetLastError();
}

int msr3d_bn_relu_train_bwd(long long rows, int C, const float *x, const float *dy, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_rstd,
                            float *dx, float *dgamma, float *dbeta, float *partial_ws,
                            float *dgamma_acc, float *dbeta_acc, msr3d_stream_t stream) {
  if (rows < 0 || C <= 0 || (C % 4) != 0 || C > 1024) return MSR3D_EINVAL;
  if (rows == 0) return 0;
  if (!x || !dy || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !partial_ws ||
      (dgamma_acc == nullptr) != (dbeta_acc == nullptr))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = chunks_of(rows);
  bn_partial_kernel<true><<<chunks, 256, partial_lds(C), st>>>(rows, C, x, dy, gamma, beta, save_mean, save_rstd,
                                                               partial_ws);
  bn_bwd_finalize_kernel<<<(C + kFinCh - 1) / kFinCh, 256, 0, st>>>(C, chunks, partial_ws, dgamma, dbeta, dgamma_acc,
                                                                    dbeta_acc);
  const long long n4 = rows * (C / 4);
  bn_relu_bwd_apply_kernel<false><<<ew_grid(n4), 256, 0, st>>>(
      n4, C / 4, 1.0f / (float)rows, reinterpret_cast<const float4 *>(x),
      reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(gamma),
      reinterpret_cast<const float4 *>(beta), reinterpret_cast<const float4 *>(save_mean),
      reinterpret_cast<const float4 *>(save_rstd), reinterpret_cast<const float4 *>(dgamma),
      reinterpret_cast<const float4 *>(dbeta), reinterpret_cast<float4 *>(dx), Pooled());
  return (int)hipGetLastError();
}

int msr3d_bn_relu_maxpool_train_fwd(long long rows, int C, int nsample, const float *x,
                                    const float *gamma, const float *beta, float eps, float momentum,
                                    float *running_mean, float *running_var, float *pooled, int *argmax,
                                    float *xsel, float *save_mean, float *save_rstd, float *partial_ws,
                                    int partial_chunks, msr3d_stream_t stream) {
  if (rows < 0 || C <= 0 || (C % 4) != 0 || C > 1024 || nsample <= 0 || rows % nsample || partial_chunks < 0)
    return MSR3D_EINVAL;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !pooled || !argmax || !xsel || !save_mean || !save_rstd || !partial_ws)
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = partial_chunks ? partial_chunks : chunks_of(rows);
  if (!partial_chunks)
    bn_partial_kernel<false><<<chunks, 256, partial_lds(C), st>>>(rows, C, x, nullptr, nullptr, nullptr, nullptr,
                                                                  nullptr, partial_ws);
  bn_fwd_finalize_kernel<<<(C + kFinCh - 1) / kFinCh, 256, 0, st>>>(rows, C, chunks, partial_ws, eps, momentum,
                                                        running_mean, running_var, save_mean, save_rstd);
  const long long G = rows / nsample;
  const int GL = 256 / (C / 4);
  bn_relu_maxpool_kernel<<<(unsigned)((G + GL - 1) / GL), 256, 0, st>>>(
      G, nsample, C / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(gamma),
      reinterpret_cast<const float4 *>(beta), reinterpret_cast<const float4 *>(save_mean),
      reinterpret_cast<const float4 *>(save_rstd), reinterpret_cast<float4 *>(pooled),
      reinterpret_cast<int4 *>(argmax), reinterpret_cast<float4 *>(xsel));
  return (int)hipGetLastError();
}

int msr3d_bn_relu_maxpool_train_bwd(long long rows, int C, int nsample, const float *x,
                                    const float *dpooled, const float *pooled, const int *argmax,
                                    const float *xsel, const float *gamma, const float *save_mean,
                                    const float *save_rstd,
                                    float *dx, float *dgamma, float *dbeta, float *partial_ws,
                                    float *dgamma_acc, float *dbeta_acc, msr3d_stream_t stream) {
  if (rows < 0 || C <= 0 || (C % 4) != 0 || C > 1024 || nsample <= 0 || rows % nsample) return MSR3D_EINVAL;
  if (rows == 0) return 0;
  if (!x || !dpooled || !pooled || !argmax || !xsel || !gamma || !save_mean || !save_rstd || !dx || !dgamma ||
      !dbeta || !partial_ws || (dgamma_acc == nullptr) != (dbeta_acc == nullptr))
    return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long G = rows / nsample;
  const int gpc = (kChunk + nsample - 1) / nsample;          // groups per partial: >= kChunk rows' worth
  const int chunks = (int)((G + gpc - 1) / gpc);             // (<= ceil(rows / kChunk): partial_ws holds them)
  Pooled pg;
  pg.ns = nsample;
  pg.dpooled = reinterpret_cast<const float4 *>(dpooled);
  pg.pooled = reinterpret_cast<const float4 *>(pooled);
  pg.argmax = reinterpret_cast<const int4 *>(argmax);
  bn_pooled_partial_kernel<<<chunks, 256, partial_lds(C), st>>>(
      G, C, gpc, reinterpret_cast<const float4 *>(dpooled), reinterpret_cast<const float4 *>(pooled),
      reinterpret_cast<const float4 *>(xsel), save_mean, save_rstd, partial_ws);
  bn_bwd_finalize_kernel<<<(C + kFinCh - 1) / kFinCh, 256, 0, st>>>(C, chunks, partial_ws, dgamma, dbeta, dgamma_acc,
                                                                    dbeta_acc);
  const long long n4 = rows * (C / 4);
  bn_relu_bwd_apply_kernel<true><<<ew_grid(n4), 256, 0, st>>>(
      n4, C / 4, 1.0f / (float)rows, reinterpret_cast<const float4 *>(x), nullptr,
      reinterpret_cast<const float4 *>(gamma), reinterpret_cast<const float4 *>(gamma),
      reinterpret_cast<const float4 *>(save_mean), reinterpret_cast<const float4 *>(save_rstd),
      reinterpret_cast<const float4 *>(dgamma), reinterpret_cast<const float4 *>(dbeta),
      reinterpret_cast<float4 *>(dx), pg);
  return (int)hipGetLastError();
}

}  // extern "C"
