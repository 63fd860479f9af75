// pn2_ops.hip -- the nine pointnet2 ops for gfx950 (MI355X), behind the C ABI of
// include/msr3d_hip.h.  Written for CDNA4 wave64: no block-size emulation of the
// reference's CUDA launch shapes, only of its RESULTS.
//
// Reference semantics (paths relative to
// /root/reference/modules/third_party/pointnet2/_ext_src/):
//   src/sampling_gpu.cu, src/ball_query_gpu.cu, src/group_points_gpu.cu,
//   src/interpolate_gpu.cu, include/cuda_utils.h
//
// Compiled with -ffp-contract=off: the distance is the explicit chain
// fma(dz,dz, fma(dx,dx, dy*dy)) shared with oracle/pn2_oracle.c (see its header).
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/msr3d_hip.h"
#include "pn2_device.h"

namespace {

using namespace msr3d;

// ---- gather / group: exact copies, one thread per output element ----------------
__global__ void gather_points_kernel(long long total, int c, int n, int m,
                                     const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % m);
    const long long bc = t / m;       // b*c + l
    const long long bi = bc / c;
    out[t] = points[bc * n + idx[bi * m + j]];
  }
}

__global__ void gather_points_grad_kernel(long long total, int c, int n, int m,
                                          const float *__restrict__ grad_out,
                                          const int *__restrict__ idx,
                                          float *__restrict__ grad_points) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % m);
    const long long bc = t / m;
    const long long bi = bc / c;
    atomicAdd(grad_points + bc * n + idx[bi * m + j], grad_out[t]);
  }
}

// out (b,c,npoints,nsample); thread = one (b,l,j) row segment of VEC samples
template <int VEC>
__global__ void group_points_kernel(long long total, int c, int n, int npoints, int nsample,
                                    const float *__restrict__ points,
                                    const int *__restrict__ idx, float *__restrict__ out) {
  const int per_row = nsample / VEC;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int kq = (int)(t % per_row);
    long long r = t / per_row;
    const int j = (int)(r % npoints);
    r /= npoints;                      // b*c + l
    const long long bi = r / c;
    const int *ip = idx + (bi * npoints + j) * nsample + kq * VEC;
    const float *pp = points + r * n;
    float *op = out + (r * npoints + j) * (long long)nsample + kq * VEC;
    if (VEC == 4) {
      const int4 ii = *reinterpret_cast<const int4 *>(ip);
      float4 v;
      v.x = pp[ii.x]; v.y = pp[ii.y]; v.z = pp[ii.z]; v.w = pp[ii.w];
      *reinterpret_cast<float4 *>(op) = v;
    } else {
      op[0] = pp[ip[0]];
    }
  }
}

__global__ void group_points_grad_kernel(long long total, int c, int n, int npoints, int nsample,
                                         const float *__restrict__ grad_out,
                                         const int *__restrict__ idx,
                                         float *__restrict__ grad_points) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t % nsample);
    long long r = t / nsample;
    const int j = (int)(r % npoints);
    r /= npoints;
    const long long bi = r / c;
    atomicAdd(grad_points + r * n + idx[(bi * npoints + j) * nsample + k], grad_out[t]);
  }
}

// ---- three_nn (interpolate_gpu.cu:9-59): thread per unknown, known staged in LDS tiles
constexpr int kNNTile = 1024;
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m,
                                                       const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2,
                                                       int *__restrict__ idx) {
  __shared__ float sk[kNNTile * 3];
  const int obj = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const float *U = unknown + (size_t)obj * n * 3;
  const float *K = known + (size_t)obj * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) { ux = U[j * 3 + 0]; uy = U[j * 3 + 1]; uz = U[j * 3 + 2]; }
  // the reference keeps doubles initialised to 1e40; every comparison is against an f32 d,
  // so f32 bests initialised to +inf order identically and (float)1e40 == +inf on output.
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += kNNTile) {
    const int cnt = (m - base) < kNNTile ? (m - base) : kNNTile;
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) sk[i] = K[(size_t)base * 3 + i];
    __syncthreads();
    if (j < n) {
      for (int k = 0; k < cnt; ++k) {
        const float d = sq3(ux - sk[k * 3 + 0], uy - sk[k * 3 + 1], uz - sk[k * 3 + 2]);
        const int kg = base + k;
        if (d < b1) {
          b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kg;
        } else if (d < b2) {
          b3 = b2; i3 = i2; b2 = d; i2 = kg;
        } else if (d < b3) {
          b3 = d; i3 = kg;
        }
      }
    }
  }
  if (j < n) {
    float *D = dist2 + ((size_t)obj * n + j) * 3;
    int *I = idx + ((size_t)obj * n + j) * 3;
    D[0] = b1; D[1] = b2; D[2] = b3;
    I[0] = i1; I[1] = i2; I[2] = i3;
  }
}

// out (b,c,n): (p1*w1 + p2*w2) + p3*w3 contracted as fma(p3,w3, fma(p1,w1, p2*w2))
__global__ void three_interpolate_kernel(long long total, int c, int m, int n,
                                         const float *__restrict__ points,
                                         const int *__restrict__ idx,
                                         const float *__restrict__ weight,
                                         float *__restrict__ out) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % n);
    const long long bc = t / n;
    const long long bi = bc / c;
    const long long o = (bi * n + j) * 3;
    const float *pp = points + bc * m;
    const float p1 = pp[idx[o + 0]], p2 = pp[idx[o + 1]], p3 = pp[idx[o + 2]];
    out[t] = __builtin_fmaf(p3, weight[o + 2], __builtin_fmaf(p1, weight[o + 0], p2 * weight[o + 1]));
  }
}

__global__ void three_interpolate_grad_kernel(long long total, int c, int n, int m,
                                              const float *__restrict__ grad_out,
                                              const int *__restrict__ idx,
                                              const float *__restrict__ weight,
                                              float *__restrict__ grad_points) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % n);
    const long long bc = t / n;
    const long long bi = bc / c;
    const long long o = (bi * n + j) * 3;
    const float g = grad_out[t];
    float *gp = grad_points + bc * m;
    atomicAdd(gp + idx[o + 0], g * weight[o + 0]);
    atomicAdd(gp + idx[o + 1], g * weight[o + 1]);
    atomicAdd(gp + idx[o + 2], g * weight[o + 2]);
  }
}

inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 8;   // 256 CUs x 8 blocks, grid-stride beyond that
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- deterministic backward of gather / group / interpolate ---------------------------
// The reference scatters with atomicAdd (sampling_gpu.cu:36-57, group_points_gpu.cu:42-75,
// interpolate_gpu.cu:117-154): the sum order, hence the low bits of the gradient, changes from
// run to run.  Here every (batch, row tile) workgroup first inverts the index in LDS -- a STABLE
// counting sort of the E source entries by destination -- and then each destination sums its
// entries in ascending entry order, the order of the sequential oracle: bit-reproducible, equal
// to oracle/pn2_oracle.c bit for bit, no atomics on floats and no memset (every output is
// written).  out[b][r][d] = sum_{e: idx[b][e] == d} src[b][r][e / DIV] * (w ? w[b][e] : 1).
constexpr int kDetRows = 32;             // rows (channels) per workgroup
constexpr int kDetMaxInts = 36 * 1024;   // LDS ints available to the inverted index

// Stable counting sort of the E entries of one batch item by destination, in LDS:
// list[start[d] .. start[d+1]) = the entries e with I[e] == d, in ascending e.
__device__ __forceinline__ void build_inverted_index(const int *__restrict__ I, int E, int n_dst, int *start,
                                                     int *cursor, int *list, int *chunk_dst, int *partial) {
  const int tid = threadIdx.x;
  for (int d = tid; d < n_dst; d += 256) cursor[d] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    const int d = I[e];
    if (d >= 0 && d < n_dst) atomicAdd(&cursor[d], 1);          // integer counts: order-free
  }
  __syncthreads();
  // exclusive scan of the counts: contiguous strip per thread, then a scan of the 256 strip sums
  const int per = (n_dst + 255) / 256, d0 = min(n_dst, tid * per), d1 = min(n_dst, d0 + per);
  int sum = 0;
  for (int d = d0; d < d1; ++d) sum += cursor[d];
  partial[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = partial[t];
      partial[t] = run;
      run += v;
    }
    start[n_dst] = run;
  }
  __syncthreads();
  int run = partial[tid];
  for (int d = d0; d < d1; ++d) {
    const int v = cursor[d];
    start[d] = run;
    cursor[d] = run;
    run += v;
  }
  __syncthreads();
  // stable placement, 256 entries at a time
  for (int e0 = 0; e0 < E; e0 += 256) {
    const int e = e0 + tid;
    int d = -1;
    if (e < E) {
      d = I[e];
      if (d < 0 || d >= n_dst) d = -1;
    }
    chunk_dst[tid] = d;
    __syncthreads();
    if (d >= 0) {
      int rank = 0;
      for (int t = 0; t < tid; ++t) rank += (chunk_dst[t] == d);
      list[cursor[d] + rank] = e;
    }
    __syncthreads();
    if (d >= 0) atomicAdd(&cursor[d], 1);
    __syncthreads();
  }
}

template <int DIV>
__global__ __launch_bounds__(256) void scatter_rows_det_kernel(
    int c, int n_dst, int E, const float *__restrict__ src, const int *__restrict__ idx,
    const float *__restrict__ w, float *__restrict__ out) {
  extern __shared__ int det_lds[];
  int *start = det_lds;                  // n_dst + 1
  int *cursor = start + n_dst + 1;       // n_dst
  int *list = cursor + n_dst;            // E
  __shared__ int chunk_dst[256];
  __shared__ int partial[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  build_inverted_index(idx + (size_t)b * E, E, n_dst, start, cursor, list, chunk_dst, partial);
  // ordered sums
  const int S = E / DIV;
  const int r_end = min(c, ((int)blockIdx.y + 1) * kDetRows);
  for (int r = blockIdx.y * kDetRows; r < r_end; ++r) {
    const float *G = src + ((size_t)b * c + r) * S;
    float *O = out + ((size_t)b * c + r) * n_dst;
    for (int d = tid; d < n_dst; d += 256) {
      float acc = 0.f;
      for (int q = start[d]; q < start[d + 1]; ++q) {
        const int e = list[q];
        acc += w ? G[e / DIV] * w[(size_t)b * E + e] : G[e / DIV];
      }
      O[d] = acc;
    }
  }
}

// true when the inverted index of one batch item fits in LDS
inline bool det_fits(long long n_dst, long long E) { return 2 * n_dst + 1 + E <= kDetMaxInts; }

template <int DIV>
hipError_t launch_scatter_det(int b, int c, int n_dst, int E, const float *src, const int *idx,
                              const float *w, float *out, hipStream_t stream) {
  const size_t lds = sizeof(int) * (size_t)(2 * n_dst + 1 + E);
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void *>(&scatter_rows_det_kernel<DIV>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * kDetMaxInts));
  if (attr != hipSuccess) return attr;
  scatter_rows_det_kernel<DIV><<<dim3(b, (c + kDetRows - 1) / kDetRows), 256, lds, stream>>>(
      c, n_dst, E, src, idx, w, out);
  return hipGetLastError();
}

// ---- token-major neighbourhood rows (training-mode SharedMLP operand) -------------------------
// rows[((b*m + j)*ns + r)][k] = k < 3 ? xyz[b][idx][k] - new_xyz[b][j][k] : k < 3+C ? feats[b][k-3][idx] : 0
// with idx = idx[b][j][r]: QueryAndGroup.forward (pointnet2_utils.py:314-373; K order [xyz, features],
// use_xyz) written straight in the layout the token GEMM reads, KP >= 3 + C columns (zero padded).
__global__ __launch_bounds__(256) void group_rows_kernel(int n, int m, int ns, int C, int KP,
                                                         const float *__restrict__ xyz,
                                                         const float *__restrict__ new_xyz,
                                                         const float *__restrict__ feats,
                                                         const int *__restrict__ idx,
                                                         float *__restrict__ rows) {
  const int j = blockIdx.x, b = blockIdx.y;
  const int *I = idx + ((size_t)b * m + j) * ns;
  const float *ctr = new_xyz + ((size_t)b * m + j) * 3;
  const float *X = xyz + (size_t)b * n * 3;
  const float *F = feats ? feats + (size_t)b * C * n : nullptr;
  float *O = rows + ((size_t)b * m + j) * ns * KP;
  for (int e = threadIdx.x; e < ns * KP; e += 256) {
    const int r = e / KP, k = e - r * KP;
    const int p = I[r];
    float v = 0.f;
    if (k < 3) v = X[(size_t)p * 3 + k] - ctr[k];
    else if (k < 3 + C) v = F[(size_t)(k - 3) * n + p];
    O[e] = v;
  }
}

// The same rows for levels whose source object is small enough to sit in LDS (n * (C + 1) floats: the second
// level's 32 points x 128 channels): the channel-major features are transposed into LDS once per workgroup
// (coalesced over the points), and a thread then writes 16 bytes of a row at a time -- the element-per-thread
// kernel above reads features with a stride of n floats and pays an integer division per element (206 us for the
// second level's 260 MB at 16 scenes: 1.3 TB/s).
__global__ __launch_bounds__(256) void group_rows_lds_kernel(int n, int m, int ns, int C, int KP,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             const float *__restrict__ feats,
                                                             const int *__restrict__ idx,
                                                             float *__restrict__ rows) {
  extern __shared__ float gr_lds[];              // [n][C + 1] features, then [n][3] coordinates
  const int j = blockIdx.x, b = blockIdx.y;
  const int pitch = C + 1;
  float *Ft = gr_lds, *Xs = gr_lds + (size_t)n * pitch;
  const float *F = feats + (size_t)b * C * n;
  for (int e = threadIdx.x; e < C * n; e += 256) {
    const int c = e / n, p = e - c * n;
    Ft[p * pitch + c] = F[e];
  }
  for (int e = threadIdx.x; e < n * 3; e += 256) Xs[e] = xyz[(size_t)b * n * 3 + e];
  __syncthreads();
  const int *I = idx + ((size_t)b * m + j) * ns;
  const float cx = new_xyz[((size_t)b * m + j) * 3], cy = new_xyz[((size_t)b * m + j) * 3 + 1],
              cz = new_xyz[((size_t)b * m + j) * 3 + 2];
  float4 *O = reinterpret_cast<float4 *>(rows + ((size_t)b * m + j) * ns * KP);
  const int Q = KP >> 2;                         // float4s per row (KP % 4 == 0)
  for (int e = threadIdx.x; e < ns * Q; e += 256) {
    const int r = e / Q, q = e - r * Q;
    const int p = I[r];
    const float *fp = Ft + p * pitch;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * q + u;
      v[u] = k < 3 ? Xs[p * 3 + k] - (k == 0 ? cx : k == 1 ? cy : cz) : k < 3 + C ? fp[k - 3] : 0.f;
    }
    O[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// d_feats[b][c][p] = sum over the entries e (ascending) with idx[b][e] == p of d_rows[b*E + e][3 + c]:
// the order of the sequential oracle, as in scatter_rows_det_kernel; no atomics, every output written.
__global__ __launch_bounds__(256) void group_rows_grad_kernel(int n, int E, int C, int KP,
                                                              const float *__restrict__ d_rows,
                                                              const int *__restrict__ idx,
                                                              float *__restrict__ d_feats, int stage) {
  extern __shared__ int det_lds[];
  int *start = det_lds;
  int *cursor = start + n + 1;
  int *list = cursor + n;
  __shared__ int chunk_dst[256];
  __shared__ int partial[256];
  const int b = blockIdx.x;
  build_inverted_index(idx + (size_t)b * E, E, n, start, cursor, list, chunk_dst, partial);
  const float *G = d_rows + (size_t)b * E * KP + 3;
  float *O = d_feats + (size_t)b * C * n;
  const int c0 = blockIdx.y * 64;                 // 64 channels per workgroup
  const int cw = min(64, C - c0);
  // (four gathers in flight per thread -- the entries are added in list order all the same --, and the (64, n) result
  //  goes out through LDS so that the global writes run along the points: a wave's direct stores were 64 four-byte
  //  writes n floats apart)
  float *ot = reinterpret_cast<float *>(list + E);           // [64][n + 1]
  for (int item = threadIdx.x; item < n * cw; item += 256) {
    const int p = item / cw, cl = item - p * cw, c = c0 + cl;
    float acc = 0.f;
    int q = start[p];
    const int qe = start[p + 1];
    for (; q + 4 <= qe; q += 4) {
      const float g0 = G[(size_t)list[q] * KP + c], g1 = G[(size_t)list[q + 1] * KP + c],
                  g2 = G[(size_t)list[q + 2] * KP + c], g3 = G[(size_t)list[q + 3] * KP + c];
      acc += g0; acc += g1; acc += g2; acc += g3;
    }
    for (; q < qe; ++q) acc += G[(size_t)list[q] * KP + c];
    if (stage) ot[cl * (n + 1) + p] = acc;
    else O[(size_t)c * n + p] = acc;
  }
  if (!stage) return;
  __syncthreads();
  for (int e = threadIdx.x; e < n * cw; e += 256) {
    const int cl = e / n, p = e - cl * n;
    O[(size_t)(c0 + cl) * n + p] = ot[cl * (n + 1) + p];
  }
}

}  // namespace

// =================================================================================
// C ABI
// =================================================================================
extern "C" {

int msr3d_abi_version(void) { return MSR3D_ABI_VERSION; }
int msr3d_sqdist_contract(void) { return MSR3D_SQDIST_CONTRACT; }

const char *msr3d_status_string(int status) {
  if (status == 0) return "ok";
  if (status == MSR3D_EINVAL) return "invalid argument";
  return hipGetErrorString((hipError_t)status);
}

int msr3d_furthest_point_sampling(int b, int n, int m, const float *xyz, int *idx,
                                  float *new_xyz, msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0) return MSR3D_EINVAL;
  if (b == 0 || m == 0) return 0;              // empty problem: nothing to write
  if (!xyz || !idx) return MSR3D_EINVAL;
  const hipError_t e = dispatch_fps(b, n, 3, m, xyz, idx, new_xyz, 0, nullptr, nullptr,
                                     (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MSR3D_EINVAL : (int)e;
}

int msr3d_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                        float *out, msr3d_stream_t stream) {
  if (b < 0 || c < 0 || n <= 0 || m < 0) return MSR3D_EINVAL;
  const long long total = (long long)b * c * m;
  if (total == 0) return 0;
  if (!points || !idx || !out) return MSR3D_EINVAL;
  gather_points_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(total, c, n, m,
                                                                              points, idx, out);
  return (int)hipGetLastError();
}

int msr3d_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                             float *grad_points, msr3d_stream_t stream) {
  if (b < 0 || c < 0 || n <= 0 || m < 0) return MSR3D_EINVAL;
  const long long total = (long long)b * c * m;
  const size_t obytes = sizeof(float) * (size_t)b * c * n;
  if (obytes == 0) return 0;
  if (!grad_points) return MSR3D_EINVAL;
  if (total > 0 && det_fits(n, m)) {
    if (!grad_out || !idx) return MSR3D_EINVAL;
    return (int)launch_scatter_det<1>(b, c, n, m, grad_out, idx, nullptr, grad_points,
                                      (hipStream_t)stream);
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, obytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (total == 0) return 0;
  if (!grad_out || !idx) return MSR3D_EINVAL;
  gather_points_grad_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      total, c, n, m, grad_out, idx, grad_points);
  return (int)hipGetLastError();
}

int msr3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx, msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample < 0) return MSR3D_EINVAL;
  if ((long long)b * m * nsample == 0) return 0;
  if (!new_xyz || !xyz || !idx) return MSR3D_EINVAL;
  const float radius2 = radius * radius;   // ball_query_gpu.cu:22, f32 product
  return (int)launch_ball_query(b, n, 3, m, radius2, nsample, new_xyz, xyz, idx, (hipStream_t)stream);
}

int msr3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, msr3d_stream_t stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0) return MSR3D_EINVAL;
  const long long elems = (long long)b * c * npoints * nsample;
  if (elems == 0) return 0;
  if (!points || !idx || !out) return MSR3D_EINVAL;
  if (nsample % 4 == 0 && aligned16(idx) && aligned16(out)) {
    const long long total = elems / 4;
    group_points_kernel<4><<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
        total, c, n, npoints, nsample, points, idx, out);
  } else {
    group_points_kernel<1><<<grid_for(elems, 256), 256, 0, (hipStream_t)stream>>>(
        elems, c, n, npoints, nsample, points, idx, out);
  }
  return (int)hipGetLastError();
}

int msr3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points, msr3d_stream_t stream) {
  if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0) return MSR3D_EINVAL;
  const size_t obytes = sizeof(float) * (size_t)b * c * n;
  if (obytes == 0) return 0;
  if (!grad_points) return MSR3D_EINVAL;
  const long long total = (long long)b * c * npoints * nsample;
  if (total > 0 && det_fits(n, (long long)npoints * nsample)) {
    if (!grad_out || !idx) return MSR3D_EINVAL;
    return (int)launch_scatter_det<1>(b, c, n, npoints * nsample, grad_out, idx, nullptr,
                                      grad_points, (hipStream_t)stream);
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, obytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (total == 0) return 0;
  if (!grad_out || !idx) return MSR3D_EINVAL;
  group_points_grad_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      total, c, n, npoints, nsample, grad_out, idx, grad_points);
  return (int)hipGetLastError();
}

int msr3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx, msr3d_stream_t stream) {
  if (b < 0 || n < 0 || m < 0) return MSR3D_EINVAL;
  if ((long long)b * n == 0) return 0;
  if (!unknown || (!known && m > 0) || !dist2 || !idx) return MSR3D_EINVAL;
  dim3 grid((n + 255) / 256, b);
  three_nn_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(n, m, unknown, known, dist2, idx);
  return (int)hipGetLastError();
}

int msr3d_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                            const float *weight, float *out, msr3d_stream_t stream) {
  if (b < 0 || c < 0 || m <= 0 || n < 0) return MSR3D_EINVAL;
  const long long total = (long long)b * c * n;
  if (total == 0) return 0;
  if (!points || !idx || !weight || !out) return MSR3D_EINVAL;
  three_interpolate_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      total, c, m, n, points, idx, weight, out);
  return (int)hipGetLastError();
}

int msr3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points,
                                 msr3d_stream_t stream) {
  if (b < 0 || c < 0 || m <= 0 || n < 0) return MSR3D_EINVAL;
  const size_t obytes = sizeof(float) * (size_t)b * c * m;
  if (obytes == 0) return 0;
  if (!grad_points) return MSR3D_EINVAL;
  const long long total = (long long)b * c * n;
  if (total > 0 && det_fits(m, 3ll * n)) {
    if (!grad_out || !idx || !weight) return MSR3D_EINVAL;
    return (int)launch_scatter_det<3>(b, c, m, 3 * n, grad_out, idx, weight, grad_points,
                                      (hipStream_t)stream);
  }
  hipError_t e = hipMemsetAsync(grad_points, 0, obytes, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (total == 0) return 0;
  if (!grad_out || !idx || !weight) return MSR3D_EINVAL;
  three_interpolate_grad_kernel<<<grid_for(total, 256), 256, 0, (hipStream_t)stream>>>(
      total, c, n, m, grad_out, idx, weight, grad_points);
  return (int)hipGetLastError();
}

int msr3d_group_rows(int b, int n, int m, int nsample, int C, int KP, const float *xyz,
                     const float *new_xyz, const float *feats, const int *idx, float *rows,
                     msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample < 0 || C < 0 || KP < 3 + C) return MSR3D_EINVAL;
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (!xyz || !new_xyz || !idx || !rows || (C > 0 && !feats)) return MSR3D_EINVAL;
  const size_t lds = sizeof(float) * ((size_t)n * (C + 1) + (size_t)n * 3);
  if (C > 0 && (KP & 3) == 0 && lds <= 64 * 1024 && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0)
    group_rows_lds_kernel<<<dim3(m, b), 256, lds, (hipStream_t)stream>>>(n, m, nsample, C, KP, xyz, new_xyz, feats,
                                                                         idx, rows);
  else
    group_rows_kernel<<<dim3(m, b), 256, 0, (hipStream_t)stream>>>(n, m, nsample, C, KP, xyz, new_xyz, feats,
                                                                    idx, rows);
  return (int)hipGetLastError();
}

int msr3d_group_rows_grad(int b, int n, int m, int nsample, int C, int KP, const float *d_rows,
                          const int *idx, float *d_feats, msr3d_stream_t stream) {
  if (b < 0 || n <= 0 || m < 0 || nsample < 0 || C <= 0 || KP < 3 + C) return MSR3D_EINVAL;
  if (b == 0) return 0;
  if (!d_feats) return MSR3D_EINVAL;
  const int E = m * nsample;
  if (E == 0) return (int)hipMemsetAsync(d_feats, 0, sizeof(float) * (size_t)b * C * n, (hipStream_t)stream);
  if (!d_rows || !idx) return MSR3D_EINVAL;
  if (!det_fits(n, E)) return MSR3D_EINVAL;          // the caller keeps the composite path
  size_t lds = sizeof(int) * (size_t)(2 * n + 1 + E);
  const size_t staged = lds + sizeof(float) * 64 * (size_t)(n + 1);
  const int stage = staged <= 160 * 1024 - 4096;             // (the output tile beside the inverted index)
  if (stage) lds = staged;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void *>(&group_rows_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
      160 * 1024 - 4096);
  if (attr != hipSuccess) return (int)attr;
  group_rows_grad_kernel<<<dim3(b, (C + 63) / 64), 256, lds, (hipStream_t)stream>>>(n, E, C, KP, d_rows, idx,
                                                                                   d_feats, stage);
  return (int)hipGetLastError();
}

}  // extern "C"
