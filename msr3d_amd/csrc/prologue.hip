// prologue.hip -- the data-only front of the situated encoder (no gradients flow here:
// inputs are dataset tensors), one launch each instead of ~10-15 torch ops:
//
//   msr3d_pairwise_locs   calc_pairwise_locs(center, spatial_dim 5, spatial_dist_norm)
//                         (/root/reference/modules/utils.py:88-137)
//   msr3d_agent_fourier   transform_to_agent_coor (modules/utils.py:60-82) followed by
//                         generate_fourier_features (model/ose3d_situation.py:31-59)
//
// Compiled with -ffp-contract=off and written in the reference's operation order (squares
// rounded separately, true divisions) so results agree with the torch formulation to the
// last bits except for the 3-term matmul order.
#include <hip/hip_runtime.h>

#include <cstdint>

#include <cmath>

#include "../../include/msr3d_hip.h"

namespace {

// one block per sample.  out[b][l][t][0..4] = [d/max d, dz/d, d2/d, dy/d2, dx/d2],
// diff = c_l - c_t, d = sqrt(dx^2+dy^2+dz^2 + eps), d2 = sqrt(dx^2+dy^2 + eps); max over ALL
// (l,t) of the sample, padded objects included.
__global__ __launch_bounds__(256) void pairwise_locs_kernel(int L, const float *__restrict__ loc,
                                                            int ld, float eps,
                                                            float *__restrict__ out) {
  __shared__ float cx[128], cy[128], cz[128];
  __shared__ float wmax[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < L; i += 256) {
    const float *p = loc + ((size_t)b * L + i) * ld;
    cx[i] = p[0]; cy[i] = p[1]; cz[i] = p[2];
  }
  __syncthreads();
  float m = 0.f;
  for (int e = tid; e < L * L; e += 256) {
    const int l = e / L, t = e - l * L;
    const float dx = cx[l] - cx[t], dy = cy[l] - cy[t], dz = cz[l] - cz[t];
    const float d = sqrtf(((dx * dx + dy * dy) + dz * dz) + eps);
    m = fmaxf(m, d);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((tid & 63) == 0) wmax[tid >> 6] = m;
  __syncthreads();
  const float dmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  float *O = out + (size_t)b * L * L * 5;
  for (int e = tid; e < L * L; e += 256) {
    const int l = e / L, t = e - l * L;
    const float dx = cx[l] - cx[t], dy = cy[l] - cy[t], dz = cz[l] - cz[t];
    const float d = sqrtf(((dx * dx + dy * dy) + dz * dz) + eps);
    const float d2 = sqrtf((dx * dx + dy * dy) + eps);
    float *o = O + (size_t)e * 5;
    o[0] = d / dmax;
    o[1] = dz / d;
    o[2] = d2 / d;
    o[3] = dy / d2;
    o[4] = dx / d2;
  }
}

// out[m][c] = x[m][c] + v1[c] (+ v2[c]): the constant type / orientation embeddings broadcast onto
// every object token (model/ose3d_situation.py:327-365), float4 per thread
__global__ void add_row_vectors_kernel(long long n4, int D4, const float4 *__restrict__ x,
                                       const float4 *__restrict__ v1, const float4 *__restrict__ v2,
                                       float4 *__restrict__ out) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % D4);
    float4 a = x[t];
    const float4 p = v1[c];
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    if (v2) {
      const float4 q = v2[c];
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    out[t] = a;
  }
}

// one thread per token: p' = (p - anchor) @ R(q) with R built from the INVERSE orientation
// (x, y, z negated), then [p', sin(pi p' f_k), cos(pi p' f_k)] in (coordinate, band) order.
__global__ void agent_fourier_kernel(int B, int L, const float *__restrict__ loc, int ld,
                                     const float *__restrict__ anchor_loc,
                                     const float *__restrict__ anchor_ori,
                                     const float *__restrict__ freqs, int nb, int transform,
                                     float *__restrict__ out) {
  const int tok = blockIdx.x * blockDim.x + threadIdx.x;
  if (tok >= B * L) return;
  const int b = tok / L;
  const float *p = loc + (size_t)tok * ld;
  float v[3] = {p[0], p[1], p[2]};
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
  const int W = 3 + 6 * nb;
  float *o = out + (size_t)tok * W;
  const float pi = 3.14159265358979323846f;
  for (int c = 0; c < 3; ++c) {
    o[c] = v[c];
    for (int k = 0; k < nb; ++k) {
      const float s = pi * (v[c] * freqs[k]);
      o[3 + c * nb + k] = sinf(s);
      o[3 + 3 * nb + c * nb + k] = cosf(s);
    }
  }
}

}  // namespace

extern "C" {

int msr3d_pairwise_locs(int B, int L, const float *loc, int ld_loc, float eps, float *out,
                        msr3d_stream_t stream) {
  if (B < 0 || L <= 0 || L > 128 || ld_loc < 3) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!loc || !out) return MSR3D_EINVAL;
  pairwise_locs_kernel<<<B, 256, 0, (hipStream_t)stream>>>(L, loc, ld_loc, eps, out);
  return (int)hipGetLastError();
}

int msr3d_agent_fourier(int B, int L, const float *loc, int ld_loc, const float *anchor_loc,
                        const float *anchor_ori, const float *freqs, int num_bands, int transform,
                        float *out, msr3d_stream_t stream) {
  if (B < 0 || L <= 0 || ld_loc < 3 || num_bands <= 0) return MSR3D_EINVAL;
  if (B == 0) return 0;
  if (!loc || !freqs || !out || (transform && (!anchor_loc || !anchor_ori))) return MSR3D_EINVAL;
  const int n = B * L;
  agent_fourier_kernel<<<(n + 127) / 128, 128, 0, (hipStream_t)stream>>>(
      B, L, loc, ld_loc, anchor_loc, anchor_ori, freqs, num_bands, transform, out);
  return (int)hipGetLastError();
}

int msr3d_add_row_vectors(int M, int D, const float *x, const float *v1, const float *v2,
                          float *out, msr3d_stream_t stream) {
  if (M < 0 || D <= 0 || D % 4 != 0) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!x || !v1 || !out) return MSR3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v1) | reinterpret_cast<uintptr_t>(v2) |
       reinterpret_cast<uintptr_t>(out)) & 15u)
    return MSR3D_EINVAL;
  const long long n4 = (long long)M * (D / 4);
  long long g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  add_row_vectors_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(
      n4, D / 4, reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(v1),
      reinterpret_cast<const float4 *>(v2), reinterpret_cast<float4 *>(out));
  return (int)hipGetLastError();
}

}  // extern "C"
