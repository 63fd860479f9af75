// preprocess.hip -- scene-sample construction on the device.
//
// The reference builds every training sample on the host, single process (`num_workers: 0`,
// configs/msr3d.yaml:165): per scan a per-instance boolean mask over all points
// (data/datasets/scannet_base.py:65-67, "time consuming"), per sample and object a rotate /
// box / subsample / normalise chain of numpy calls (data/datasets/msr3d.py:181-216) and the
// wrapper's padding (data/datasets/dataset_wrapper.py:141-158), then 1.47 MB/sample over PCIe.
// Here the scans live in HBM (a whole ScanNet is ~3.4 GB of 288), segmented ONCE into
// instance-contiguous order, and a sample is one launch that writes `obj_fts`, `obj_locs`,
// `obj_masks` where the encoder reads them:
//
//   segment_scan      labels -> stable counting sort by instance slot (3 small kernels):
//                     instance i's points, in ascending original order, become the contiguous
//                     range [inst_offsets[i], inst_offsets[i+1]) of points_sorted/colors_sorted
//                     (== pcds[instance_labels == i]).
//   preprocess_pcd    one workgroup per (object slot, sample):  pass 1 streams the object's
//                     points (contiguous, coalesced) for centre / box -> obj_locs;  pass 2
//                     gathers the P-point subsample into LDS, centres it on its own mean,
//                     scales by the largest norm, writes (P,6) fp32.  Padding slots are filled
//                     with 1.0 / 0 / mask 0 by the same launch.
//
// Arithmetic is float64 like the reference's (the `colors / 127.5 - 1` at scannet_base.py:60
// promotes the concatenated array to float64; obj_fts/obj_locs are cast with `.float()` at the
// end, dataset_wrapper.py:156-158).  Sums are tree-reduced here and sequential in numpy, so
// results agree to float64 rounding and, after the cast, to <= 1 fp32 ulp (tests state it).
// HBM-bound byte work: no MFMA.  Built with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/msr3d_hip.h"

namespace {

constexpr int kSegChunk = MSR3D_SEG_CHUNK;   // points per segmentation block (= block size)
constexpr int kMaxSlots = MSR3D_SEG_MAX_SLOTS;   // LDS histogram bins

// ------------------------------------------------------------------ segmentation
__device__ __forceinline__ int slot_of(const long long *labels, const int *slot_of_label,
                                       int n_labels, int i) {
  const long long l = labels[i];
  return (l >= 0 && l < n_labels) ? slot_of_label[l] : -1;
}

__global__ __launch_bounds__(kSegChunk) void seg_hist_kernel(int n, const long long *__restrict__ labels,
                                                             const int *__restrict__ slot_of_label,
                                                             int n_labels, int n_slots,
                                                             int *__restrict__ chunk_hist) {
  extern __shared__ int hist[];
  for (int s = threadIdx.x; s < n_slots; s += kSegChunk) hist[s] = 0;
  __syncthreads();
  const int i = blockIdx.x * kSegChunk + threadIdx.x;
  if (i < n) {
    const int s = slot_of(labels, slot_of_label, n_labels, i);
    if (s >= 0 && s < n_slots) atomicAdd(&hist[s], 1);
  }
  __syncthreads();
  for (int s = threadIdx.x; s < n_slots; s += kSegChunk)
    chunk_hist[(size_t)blockIdx.x * n_slots + s] = hist[s];
}

// chunk_hist[c][s] -> number of slot-s points in chunks < c;  inst_offsets = exclusive scan of totals
__global__ __launch_bounds__(1024) void seg_scan_kernel(int n_chunks, int n_slots,
                                                        int *__restrict__ chunk_hist,
                                                        int *__restrict__ inst_offsets) {
  for (int s = threadIdx.x; s < n_slots; s += 1024) {
    int run = 0;
    for (int c = 0; c < n_chunks; ++c) {
      const size_t at = (size_t)c * n_slots + s;
      const int t = chunk_hist[at];
      chunk_hist[at] = run;
      run += t;
    }
    inst_offsets[s + 1] = run;           // totals, scanned below
  }
  __syncthreads();
  if (threadIdx.x == 0) {                // <= 8192 adds, once per scan
    int run = 0;
    inst_offsets[0] = 0;
    for (int s = 0; s < n_slots; ++s) {
      run += inst_offsets[s + 1];
      inst_offsets[s + 1] = run;
    }
  }
}

__global__ __launch_bounds__(kSegChunk) void seg_scatter_kernel(
    int n, const long long *__restrict__ labels, const int *__restrict__ slot_of_label, int n_labels,
    int n_slots, const int *__restrict__ chunk_base, const int *__restrict__ inst_offsets,
    const float *__restrict__ points, const unsigned char *__restrict__ colors,
    float *__restrict__ points_sorted, unsigned char *__restrict__ colors_sorted,
    int *__restrict__ order) {
  __shared__ int slots[kSegChunk];
  const int i = blockIdx.x * kSegChunk + threadIdx.x;
  int s = -1;
  if (i < n) {
    s = slot_of(labels, slot_of_label, n_labels, i);
    if (s >= n_slots) s = -1;
  }
  slots[threadIdx.x] = s;
  __syncthreads();
  if (s < 0) return;
  int rank = 0;                          // earlier points of this chunk with the same slot: stable
  for (int t = 0; t < (int)threadIdx.x; ++t) rank += (slots[t] == s);
  const int dst = inst_offsets[s] + chunk_base[(size_t)blockIdx.x * n_slots + s] + rank;
  if (order) order[dst] = i;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    points_sorted[(size_t)dst * 3 + d] = points[(size_t)i * 3 + d];
    colors_sorted[(size_t)dst * 3 + d] = colors[(size_t)i * 3 + d];
  }
}

// ------------------------------------------------------------------ subsample draw
// Restated bit for bit in oracle/sample_input.py (draw_indices).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint32_t object_key(unsigned long long seed, int b, int o) {
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9u);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k ^ ((uint32_t)b * 0x85EBCA6Bu));
  k = mix32(k ^ ((uint32_t)o * 0xC2B2AE35u));
  return k;
}

// 6-round alternating Feistel permutation of [0, 2^bits): high bits/2 and low bits - bits/2 bits
// take turns being whitened by a hash of the other half (every round is a bijection, so the
// composition is a permutation for any split).
__device__ __forceinline__ uint32_t feistel(uint32_t x, uint32_t key, int bits) {
  const int lb = bits >> 1, rb = bits - lb;
  const uint32_t lmask = (1u << lb) - 1u, rmask = (1u << rb) - 1u;
  uint32_t left = (x >> rb) & lmask, right = x & rmask;
#pragma unroll
  for (uint32_t r = 0; r < 6; ++r) {
    if ((r & 1u) == 0)
      left ^= mix32(right ^ key ^ (r * 0x9E3779B1u)) & lmask;
    else
      right ^= mix32(left ^ key ^ (r * 0x9E3779B1u)) & rmask;
  }
  return (left << rb) | right;
}

// n >= P: image of j under a keyed permutation of [0, n) (distinct for distinct j);
// n <  P: an independent uniform draw.
__device__ __forceinline__ int draw_index(uint32_t key, uint32_t j, int n, int P, int bits) {
  if (n < P) {
    const uint32_t u = mix32(mix32(j ^ key) + 0x68E31DA4u);
    return (int)(((unsigned long long)u * (unsigned long long)n) >> 32);
  }
  uint32_t y = feistel(j, key, bits);
  while (y >= (uint32_t)n) y = feistel(y, key, bits);      // cycle walk back into range
  return (int)y;
}

// ------------------------------------------------------------------ block reductions (f64)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// reduces K values per thread over the block; op: 0 sum, 1 min, 2 max.  Result in every thread.
template <int K, int kNT>
__device__ __forceinline__ void block_reduce(double (&v)[K], const int (&op)[K], double *scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double r = op[k] == 0 ? wave_sum(v[k]) : op[k] == 1 ? wave_min(v[k]) : wave_max(v[k]);
    if (lane == 0) scratch[wave * K + k] = r;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double r = scratch[k];
#pragma unroll
    for (int w = 1; w < kNT / 64; ++w) {
      const double t = scratch[w * K + k];
      r = op[k] == 0 ? r + t : op[k] == 1 ? fmin(r, t) : fmax(r, t);
    }
    v[k] = r;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ preprocess_pcd
// grid (O, B), kNT threads; dynamic LDS = P * (3 doubles + 3 floats) + 160 doubles + 256 floats.  Loads are
// issued four deep (independent points per thread) to keep enough HBM requests in flight.
template <int kNT>
__global__ __launch_bounds__(kNT) void preprocess_pcd_kernel(
    int O, int P, const float *__restrict__ points, const unsigned char *__restrict__ colors,
    const long long *__restrict__ obj_begin, const int *__restrict__ obj_count,
    const float *__restrict__ rot, const int *__restrict__ pcd_idxs, unsigned long long seed,
    float *__restrict__ obj_fts, float *__restrict__ obj_locs, unsigned char *__restrict__ obj_masks,
    int *__restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *scratch = lds;                       // 160 >= 16 waves x 9
  double *sub = lds + 160;                     // P x 3, the rotated subsample
  float *rgb = reinterpret_cast<float *>(sub + (size_t)P * 3);   // P x 3
  float *lut = rgb + (size_t)P * 3;            // 256: (float)(c / 127.5 - 1) for every byte value
  const int o = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t slot = (size_t)b * O + o;
  const int n = obj_count[slot];
  float *out = obj_fts + slot * (size_t)P * 6;
  float *loc = obj_locs + slot * 6;

  if (n <= 0) {                                // padding slot (dataset_wrapper.py:156-158)
    for (int e = tid; e < P * 6; e += kNT) out[e] = 1.0f;
    if (tid < 6) loc[tid] = 0.0f;
    if (tid == 0) obj_masks[slot] = 0;
    if (idx_out)
      for (int j = tid; j < P; j += kNT) idx_out[slot * P + j] = -1;
    return;
  }
  for (int c = tid; c < 256; c += kNT) lut[c] = (float)((double)c / 127.5 - 1.0);   // scannet_base.py:60
  const float *pts = points + (size_t)obj_begin[slot] * 3;
  const unsigned char *col = colors + (size_t)obj_begin[slot] * 3;
  double r[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const bool rotate = rot != nullptr;
  if (rotate)
#pragma unroll
    for (int e = 0; e < 9; ++e) r[e] = (double)rot[(size_t)b * 9 + e];

  // xyz @ rot^T in float64 (msr3d.py:189-190)
  auto load_point = [&](int i, double &x, double &y, double &z) {
    const double px = (double)pts[(size_t)i * 3], py = (double)pts[(size_t)i * 3 + 1],
                 pz = (double)pts[(size_t)i * 3 + 2];
    if (rotate) {
      x = (px * r[0] + py * r[1]) + pz * r[2];
      y = (px * r[3] + py * r[4]) + pz * r[5];
      z = (px * r[6] + py * r[7]) + pz * r[8];
    } else {
      x = px; y = py; z = pz;
    }
  };

  // pass 1: centre and box of the whole object (msr3d.py:192-194)
  {
    double v[9] = {0, 0, 0, INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int i0 = tid; i0 < n; i0 += 4 * kNT) {
      double x[4], y[4], z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kNT;
        load_point(i < n ? i : i0, x[u], y[u], z[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * kNT < n) {
          v[0] += x[u]; v[1] += y[u]; v[2] += z[u];
          v[3] = fmin(v[3], x[u]); v[4] = fmin(v[4], y[u]); v[5] = fmin(v[5], z[u]);
          v[6] = fmax(v[6], x[u]); v[7] = fmax(v[7], y[u]); v[8] = fmax(v[8], z[u]);
        }
    }
    const int op[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2};
    block_reduce<9, kNT>(v, op, scratch);
    if (tid < 3) {
      loc[tid] = (float)(v[tid] / (double)n);
      loc[3 + tid] = (float)(v[6 + tid] - v[3 + tid]);
    }
    if (tid == 0) obj_masks[slot] = 1;
  }

  // pass 2: gather the subsample (msr3d.py:200-202), its mean (:205)
  int bits = 1;
  while ((1ll << bits) < (long long)n) ++bits;
  const uint32_t key = object_key(seed, b, o);
  double m[3] = {0, 0, 0};
  for (int j0 = tid; j0 < P; j0 += 4 * kNT) {
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * kNT;
      idx[u] = 0;
      if (j < P) {
        if (pcd_idxs) {
          const int t = pcd_idxs[slot * P + j];
          idx[u] = t < 0 ? 0 : (t >= n ? n - 1 : t);        // memory safety only
        } else {
          idx[u] = draw_index(key, (uint32_t)j, n, P, bits);
        }
        if (idx_out) idx_out[slot * P + j] = idx[u];
      }
    }
    double x[4], y[4], z[4];
    unsigned char c[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load_point(idx[u], x[u], y[u], z[u]);
#pragma unroll
      for (int d = 0; d < 3; ++d) c[u][d] = col[(size_t)idx[u] * 3 + d];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * kNT;
      if (j < P) {
        sub[j * 3] = x[u]; sub[j * 3 + 1] = y[u]; sub[j * 3 + 2] = z[u];
        m[0] += x[u]; m[1] += y[u]; m[2] += z[u];
#pragma unroll
        for (int d = 0; d < 3; ++d) rgb[j * 3 + d] = lut[c[u][d]];
      }
    }
  }
  const int op3[3] = {0, 0, 0};
  block_reduce<3, kNT>(m, op3, scratch);
#pragma unroll
  for (int d = 0; d < 3; ++d) m[d] = m[d] / (double)P;

  // pass 3: centre, largest norm (:205-208)
  double d2[1] = {0};
  for (int j = tid; j < P; j += kNT) {
    const double x = sub[j * 3] - m[0], y = sub[j * 3 + 1] - m[1], z = sub[j * 3 + 2] - m[2];
    sub[j * 3] = x; sub[j * 3 + 1] = y; sub[j * 3 + 2] = z;
    d2[0] = fmax(d2[0], (x * x + y * y) + z * z);
  }
  const int op1[1] = {2};
  block_reduce<1, kNT>(d2, op1, scratch);
  double max_dist = sqrt(d2[0]);           // sqrt is monotone: max of sqrt == sqrt of max
  if (max_dist < 1e-6) max_dist = 1.0;

  // pass 4: scale (:209), cast, write whole points
  for (int j = tid; j < P; j += kNT) {
    float2 *dst = reinterpret_cast<float2 *>(out + (size_t)j * 6);
    dst[0] = make_float2((float)(sub[j * 3] / max_dist), (float)(sub[j * 3 + 1] / max_dist));
    dst[1] = make_float2((float)(sub[j * 3 + 2] / max_dist), rgb[j * 3]);
    dst[2] = make_float2(rgb[j * 3 + 1], rgb[j * 3 + 2]);
  }
}

}  // namespace

extern "C" {

int msr3d_segment_scan(int n_points, const long long *instance_labels, const int *slot_of_label,
                       int n_labels, int n_slots, const float *points, const unsigned char *colors,
                       float *points_sorted, unsigned char *colors_sorted, int *order,
                       int *inst_offsets, int *workspace, msr3d_stream_t stream) {
  if (n_points < 0 || n_labels < 0 || n_slots <= 0 || n_slots > kMaxSlots) return MSR3D_EINVAL;
  if (!inst_offsets) return MSR3D_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (n_points == 0) return (int)hipMemsetAsync(inst_offsets, 0, sizeof(int) * (n_slots + 1), s);
  if (!instance_labels || !slot_of_label || !points || !colors || !points_sorted || !colors_sorted ||
      !workspace)
    return MSR3D_EINVAL;
  const int n_chunks = (n_points + kSegChunk - 1) / kSegChunk;
  seg_hist_kernel<<<n_chunks, kSegChunk, sizeof(int) * n_slots, s>>>(
      n_points, instance_labels, slot_of_label, n_labels, n_slots, workspace);
  seg_scan_kernel<<<1, 1024, 0, s>>>(n_chunks, n_slots, workspace, inst_offsets);
  seg_scatter_kernel<<<n_chunks, kSegChunk, 0, s>>>(n_points, instance_labels, slot_of_label, n_labels,
                                                    n_slots, workspace, inst_offsets, points, colors,
                                                    points_sorted, colors_sorted, order);
  return (int)hipGetLastError();
}

int msr3d_preprocess_pcd(int B, int O, int P, const float *points, const unsigned char *colors,
                         const long long *obj_begin, const int *obj_count, const float *rot,
                         const int *pcd_idxs, unsigned long long seed, float *obj_fts,
                         float *obj_locs, unsigned char *obj_masks, int *idx_out,
                         msr3d_stream_t stream) {
  if (B < 0 || O < 0 || P <= 0 || (P & 1) || P > 4096) return MSR3D_EINVAL;
  if (B == 0 || O == 0) return 0;
  if (!points || !colors || !obj_begin || !obj_count || !obj_fts || !obj_locs || !obj_masks)
    return MSR3D_EINVAL;
  const size_t lds = sizeof(double) * (160 + (size_t)P * 3) + sizeof(float) * ((size_t)P * 3 + 256);
#define LAUNCH_PRE(NT)                                                                              \
  {                                                                                                 \
    static const hipError_t attr = hipFuncSetAttribute(                                             \
        reinterpret_cast<const void *>(&preprocess_pcd_kernel<NT>),                                 \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                    \
    if (attr != hipSuccess) return (int)attr;                                                       \
    preprocess_pcd_kernel<NT><<<dim3(O, B), NT, lds, (hipStream_t)stream>>>(                        \
        O, P, points, colors, obj_begin, obj_count, rot, pcd_idxs, seed, obj_fts, obj_locs,         \
        obj_masks, idx_out);                                                                        \
  }
  LAUNCH_PRE(256)
  return (int)hipGetLastError();
}

}  // extern "C"
