/*
 * oracle/pn2_oracle.c -- CPU restatement of the reference's pointnet2._ext ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under msr3d_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * Reference (read-only, /root/reference, paths relative to
 * modules/third_party/pointnet2/_ext_src/):
 *   src/sampling_gpu.cu, src/ball_query_gpu.cu, src/group_points_gpu.cu,
 *   src/interpolate_gpu.cu, include/cuda_utils.h and the host allocators in
 *   src/sampling.cpp, src/ball_query.cpp, src/group_points.cpp,
 *   src/interpolate.cpp.  Each function below cites the lines it follows.
 *
 * PINNING STATUS: the reference's native ops are CUDA-only (every host entry
 * asserts "CPU not supported") and there is no nvcc here, so the reference
 * itself cannot be run.  The reference's only test for this path
 * (pointnet2_test.py:18-33, three_interpolate gradcheck input) is reproduced in
 * tests/; everything else is pinned by known-answer cases derived from the
 * kernel text (tests/test_oracle_kat.py).  For the native ops parity is
 * therefore "unpinned by reference outputs": this restatement DEFINES the
 * contract.  The FPS below is a literal thread-by-thread simulation of the CUDA
 * block (per-thread strided scan + shared-memory halving tree), NOT the
 * closed-form tie-break rule the HIP kernels use, so HIP == oracle also checks
 * that rule.
 *
 * Floating-point contract.  nvcc's default (-fmad=true) contracts
 *     a*a + b*b + c*c         (a,b,c = coordinate differences)
 * An LLVM-based device compiler contracts the parse tree ((a*a + b*b) + c*c) to
 *     fma(c, c, fma(a, a, b*b))
 * (checked here with hipcc on the reference's literal expression: v_mul b*b,
 * v_fma a, v_fma c).  We pin exactly that chain with explicit fmaf() and build
 * with -ffp-contract=off, so CPU and GPU agree bit-for-bit by construction.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* (a*a + b*b) + c*c under the floating-point contract in force.  The reference's source spells the sum
 * without parentheses (sampling_gpu.cu:99-104, ball_query_gpu.cu:32-35); which products its compiler fused
 * into fma is not recoverable from the source.  Contract 0 is what an LLVM-based device compiler emits
 * and what the HIP kernels are built with by default; the others are the remaining plausible choices, kept
 * so that (a) tools/fma_contract_risk.py can measure how much the choice matters and (b) a vector from an
 * NVIDIA build, should one appear, can be matched by flipping a switch (and MSR3D_SQDIST_CONTRACT in
 * msr3d_amd/csrc/pn2_device.h):
 *   0  fma(c,c, fma(a,a, b*b))      LLVM contraction of ((a*a + b*b) + c*c)          [default]
 *   1  (a*a + b*b) + c*c            no contraction (-fmad=false)
 *   2  fma(c,c, fma(b,b, a*a))      left product kept, the other two fused
 *   3  fma(a,a, fma(b,b, c*c))      right-to-left chain
 *   4, 5  contract 0 pushed one ulp up / down: NOT a contract -- the envelope of every possible rounding
 *         difference, for the exposure figure of tools/fma_contract_risk.py */
static int g_contract = 0;
void pn2o_set_contract(int c) { g_contract = (c >= 0 && c <= 5) ? c : 0; }
int pn2o_get_contract(void) { return g_contract; }
static inline float sq3(float a, float b, float c) {
  switch (g_contract) {
    case 1: { const float aa = a * a, bb = b * b, cc = c * c; const float s = aa + bb; return s + cc; }
    case 2: return fmaf(c, c, fmaf(b, b, a * a));
    case 3: return fmaf(a, a, fmaf(b, b, c * c));
    case 4: return nextafterf(fmaf(c, c, fmaf(a, a, b * b)), INFINITY);
    case 5: return nextafterf(fmaf(c, c, fmaf(a, a, b * b)), -INFINITY);
    default: return fmaf(c, c, fmaf(a, a, b * b));
  }
}

/* the squared distance under the contract in force (tests, tools/fma_contract_risk.py) */
float pn2o_sq3(float a, float b, float c) { return sq3(a, b, c); }

/* include/cuda_utils.h:13-19 -- 2^floor(log2 work) clamped to [1, 512]; the
 * double log/log and the int truncation are kept as written. */
int pn2o_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

/* include/cuda_utils.h:21-28 */
void pn2o_opt_block_config(int x, int y, int *bx, int *by) {
  const int xt = pn2o_opt_n_threads(x);
  int yt = pn2o_opt_n_threads(y);
  if (yt > 512 / xt) yt = 512 / xt;
  if (yt < 1) yt = 1;
  *bx = xt;
  *by = yt;
}

/* sampling_gpu.cu:59-65 (__update) */
static inline void fps_update(float *dists, int *dists_i, int i1, int i2) {
  const float v1 = dists[i1], v2 = dists[i2];
  const int a1 = dists_i[i1], a2 = dists_i[i2];
  dists[i1] = v1 > v2 ? v1 : v2; /* max(v1, v2) */
  dists_i[i1] = v2 > v1 ? a2 : a1;
}

/* sampling_gpu.cu:69-173 + host sampling.cpp:66-87 (temp = 1e10, idx zeroed).
 * One simulated block of `bs` threads per batch element. */
void pn2o_furthest_point_sampling(int b, int n, int m, const float *xyz, int *idxs) {
  if (m <= 0) return;
  const int bs = pn2o_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *dataset = xyz + (size_t)bi * n * 3;
    int *out = idxs + (size_t)bi * m;
    float *temp = (float *)malloc(sizeof(float) * (size_t)n);
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    for (int k = 0; k < n; ++k) temp[k] = 1e10f;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = dataset[old * 3 + 0];
      const float y1 = dataset[old * 3 + 1];
      const float z1 = dataset[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) { /* :89-111, one thread at a time */
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = dataset[k * 3 + 0];
          const float y2 = dataset[k * 3 + 1];
          const float z2 = dataset[k * 3 + 2];
          const float mag = sq3(x2, y2, z2);
          if ((double)mag <= 1e-3) continue; /* :100-101, double literal */
          const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
          const float d2 = fminf(d, temp[k]);
          temp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      /* :114-168 halving tree, strides 256..1 limited to the block size */
      for (int s = 256; s >= 1; s >>= 1) {
        if (bs >= 2 * s) {
          for (int tid = 0; tid < s; ++tid) fps_update(dists, dists_i, tid, tid + s);
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
    free(temp);
    free(dists);
    free(dists_i);
  }
}

/* sampling_gpu.cu:8-20; out pre-zeroed by sampling.cpp:25-27 */
void pn2o_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                        float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* sampling_gpu.cu:34-47; grad_points zero-init (sampling.cpp:50-52).  The GPU
 * uses atomicAdd (order undefined); the oracle accumulates in (j) order. */
void pn2o_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                             float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ball_query_gpu.cu:9-44; idx zero-init (ball_query.cpp:19-21).
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample). */
void pn2o_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx) {
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
  const float radius2 = radius * radius;
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *P = xyz + (size_t)bi * n * 3;
    const float *Q = new_xyz + (size_t)bi * m * 3;
    int *I = idx + (size_t)bi * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float nx = Q[j * 3 + 0], ny = Q[j * 3 + 1], nz = Q[j * 3 + 2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = P[k * 3 + 0], y = P[k * 3 + 1], z = P[k * 3 + 2];
        const float d2 = sq3(nx - x, ny - y, nz - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) I[j * nsample + l] = k;
          I[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* group_points_gpu.cu:8-28: out[b,c,j,k] = points[b,c,idx[b,j,k]] */
void pn2o_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *P = points + (size_t)bi * n * c;
    const int *I = idx + (size_t)bi * npoints * nsample;
    float *O = out + (size_t)bi * npoints * nsample * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          O[((size_t)l * npoints + j) * nsample + k] = P[(size_t)l * n + I[j * nsample + k]];
  }
}

/* group_points_gpu.cu:43-64; zero-init by group_points.cpp:48-50; sums in (j,k) order */
void pn2o_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi) {
    const float *G = grad_out + (size_t)bi * npoints * nsample * c;
    const int *I = idx + (size_t)bi * npoints * nsample;
    float *O = grad_points + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          O[(size_t)l * n + I[j * nsample + k]] += G[((size_t)l * npoints + j) * nsample + k];
  }
}

/* interpolate_gpu.cu:9-59: unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3), idx (b,n,3).
 * f32 distance, double bests initialised to 1e40, strict '<'. */
void pn2o_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx) {
  for (int bi = 0; bi < b; ++bi) {
    const float *U = unknown + (size_t)bi * n * 3;
    const float *K = known + (size_t)bi * m * 3;
    float *D = dist2 + (size_t)bi * n * 3;
    int *I = idx + (size_t)bi * n * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = U[j * 3 + 0], uy = U[j * 3 + 1], uz = U[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = K[k * 3 + 0], y = K[k * 3 + 1], z = K[k * 3 + 2];
        const float d = sq3(ux - x, uy - y, uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      D[j * 3 + 0] = (float)best1; D[j * 3 + 1] = (float)best2; D[j * 3 + 2] = (float)best3;
      I[j * 3 + 0] = besti1; I[j * 3 + 1] = besti2; I[j * 3 + 2] = besti3;
    }
  }
}

/* interpolate_gpu.cu:72-101: points (b,c,m), idx/weight (b,n,3) -> out (b,c,n).
 * (p1*w1 + p2*w2) + p3*w3 contracted like sq3: fma(p3,w3, fma(p1,w1, p2*w2)). */
void pn2o_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                            const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi) {
    const float *P = points + (size_t)bi * m * c;
    const int *I = idx + (size_t)bi * n * 3;
    const float *W = weight + (size_t)bi * n * 3;
    float *O = out + (size_t)bi * n * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float w1 = W[j * 3 + 0], w2 = W[j * 3 + 1], w3 = W[j * 3 + 2];
        const float p1 = P[(size_t)l * m + I[j * 3 + 0]];
        const float p2 = P[(size_t)l * m + I[j * 3 + 1]];
        const float p3 = P[(size_t)l * m + I[j * 3 + 2]];
        O[(size_t)l * n + j] = fmaf(p3, w3, fmaf(p1, w1, p2 * w2));
      }
  }
}

/* interpolate_gpu.cu:116-143; zero-init by interpolate.cpp:86-88; sums in (j, 1..3) order */
void pn2o_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi) {
    const float *G = grad_out + (size_t)bi * n * c;
    const int *I = idx + (size_t)bi * n * 3;
    const float *W = weight + (size_t)bi * n * 3;
    float *O = grad_points + (size_t)bi * m * c;
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < n; ++j) {
        const float g = G[(size_t)l * n + j];
        O[(size_t)l * m + I[j * 3 + 0]] += g * W[j * 3 + 0];
        O[(size_t)l * m + I[j * 3 + 1]] += g * W[j * 3 + 1];
        O[(size_t)l * m + I[j * 3 + 2]] += g * W[j * 3 + 2];
      }
  }
}

int pn2o_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void pn2o_set_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}
