// rows_linear.hip -- C = A W^T + b for a few hundred token / object rows and a LONG reduction, on the bf16
// matrix pipe at fp32 accuracy (split_mma.h: three exact bf16 terms per operand, six products).
//
// Two products of the hot path have this shape at 960 rows and were the last ones on the f32-input MFMA
// (panel_gemm.hip, 32 cycles per 16 x 16 x 4 block against 96 cycles per 16 x 16 x 32 here):
//   the encoder's `fc`, 768 -> 768 (/root/reference/modules/layers/pointnet.py:52-63), frozen: 22.8 us a launch there,
//   18.5 us here;
//   `obj_linear_projection`, 768 -> 256 (/root/reference/model/ose3d_situation.py:284-290): 12.7 us there, 12.6 here --
//   measured, and left on the panel kernel (the step would pay for one more pack job).
//
// A workgroup (4 waves) owns 64 rows x 32 output columns: wave (rh, ct) holds row tiles 2 rh, 2 rh + 1 of column
// tile ct.  The reduction runs in chunks of 128: the chunk's rows are read as fp32 (16-byte loads, two chunks
// ahead, in registers), split into three bf16 planes on their way into LDS (ROWS layout of split_mma.h, pitch
// 136), and multiplied against the chunk's four weight pieces, which come pre-split and fragment-packed
// (msr3d_split_pack's layout: [K / 32 slabs][N / 16 tiles][3 planes][64 lanes][16 B]) through a buffer descriptor.
// 52 KB of LDS: three workgroups share a CU and hide one another's round trips; 15 x 24 = 360 workgroups for
// 960 x 768.  No K split, no atomics: a row's result does not depend on which other rows share the launch, and it
// is bit-reproducible.
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "split_mma.h"

namespace {

using namespace msr3d;

constexpr int RT = 64;                    // rows of a tile
constexpr int KC = 128;                   // reduction chunk
constexpr int RPITCH = KC + 8;            // bf16 units: 272-byte rows, conflict-free ds_read_b128 fragments
constexpr int RPLANE = RT * RPITCH;
constexpr int RL_LDS = 3 * RPLANE * 2;    // 52,224 B

__device__ __forceinline__ float4 rl_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

struct Chunk { float4 v[4][2]; };         // thread (wave, sub, cseg): rows 16 wave + 4 j + sub, columns 64 q + 4 cseg ..

__device__ __forceinline__ void fetch_chunk(Chunk &c, const float *__restrict__ a, int lda, int row0, int rows, int k0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cseg = lane & 15, sub = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 16 + 4 * j + sub;
    const float *src = a + (size_t)(row0 + min(r, rows - 1)) * lda + k0 + 4 * cseg;
#pragma unroll
    for (int q = 0; q < 2; ++q) c.v[j][q] = rl_ld4(src + 64 * q);
  }
}

__device__ __forceinline__ void split_chunk(const Chunk &c, unsigned short *xs, int rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cseg = lane & 15, sub = lane >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 16 + 4 * j + sub;
    const bool ok = r < rows;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 in = c.v[j][q];
      const float v[4] = {ok ? in.x : 0.f, ok ? in.y : 0.f, ok ? in.z : 0.f, ok ? in.w : 0.f};
      uint2 pl[3];
      sm_split4(v, pl);
      unsigned short *d = xs + r * RPITCH + 64 * q + 4 * cseg;
#pragma unroll
      for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2 *>(d + k * RPLANE) = pl[k];
    }
  }
}

__global__ __launch_bounds__(256) void rows_linear_kernel(int M, int N, int K, const float *__restrict__ a, int lda,
                                                          const unsigned short *__restrict__ w, unsigned w_bytes,
                                                          const float *__restrict__ bias, float *__restrict__ C, int ldc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short *xs = reinterpret_cast<unsigned short *>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int rh = wave >> 1, ct = wave & 1;
  // (Measured and not kept: an XCD-aware tile order -- XCD x owning one row half and one column quarter, 19 MB over the
  //  fabric instead of 51 -- 18.1 us against 18.5: the launch is not bound by what crosses the fabric.)
  const int rt = blockIdx.y, cp = blockIdx.x;
  const int row0 = RT * rt, rows = min(RT, M - row0);
  const int tile = 2 * cp + ct, nt = N / 16, nc = K / KC;
  const XRows xr = make_xrows(xs, RPITCH, RT, lane);
  const int n0 = 16 * tile + 4 * g;
  const float4 bv = bias ? rl_ld4(bias + n0) : make_float4(0.f, 0.f, 0.f, 0.f);

  f32x4 acc[1][2];
  zero_acc3(acc);
  Chunk c0, c1;
  fetch_chunk(c0, a, lda, row0, rows, 0);
  if (nc > 1) fetch_chunk(c1, a, lda, row0, rows, KC);
  // (Measured and not kept: the next chunk's weight pieces in a second register ring -- 196 registers instead of 120,
  //  two workgroups a CU instead of three: 18.5 -> 27.5 us for 768 -> 768.)
  for (int c = 0; c < nc; c += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (c + u >= nc) break;
      // this chunk's four weight pieces fly under the split
      const WStream ws = make_wstream(w, w_bytes, nt, 4 * (c + u), tile, lane);
      WPiece ring[4];
      preload_wring<1, 4>(ring, ws);
      if (c + u > 0) __syncthreads();                 // every wave is done with the previous chunk's planes
      split_chunk(u == 0 ? c0 : c1, xs, rows);
      if (c + u + 2 < nc) fetch_chunk(u == 0 ? c0 : c1, a, lda, row0, rows, (c + u + 2) * KC);
      __syncthreads();
      gemm_split3<true, 1, 2, 4, 4>(xr, 2 * rh, ws, acc, ring);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int r = 16 * (2 * rh + mt) + j;
    if (r < rows)
      *reinterpret_cast<float4 *>(C + (size_t)(row0 + r) * ldc + n0) =
          make_float4(acc[0][mt][0] + bv.x, acc[0][mt][1] + bv.y, acc[0][mt][2] + bv.z, acc[0][mt][3] + bv.w);
  }
}

}  // namespace

extern "C" int msr3d_rows_linear_split(int M, int N, int K, const float *a, int lda, const unsigned short *w_pack,
                                       unsigned w_bytes, const float *bias, float *C, int ldc, msr3d_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0 || N % 32 || K % KC || lda < K || lda % 4 || ldc < N || ldc % 4) return MSR3D_EINVAL;
  if (M == 0) return 0;
  if (!a || !w_pack || !C) return MSR3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(w_pack)) & 15)
    return MSR3D_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return MSR3D_EINVAL;
  if ((unsigned long long)w_bytes < (unsigned long long)(K / 32) * (N / 16) * kPieceBytes) return MSR3D_EINVAL;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&rows_linear_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
  if (attr != hipSuccess) return (int)attr;
  rows_linear_kernel<<<dim3(N / 32, (M + RT - 1) / RT), 256, RL_LDS, (hipStream_t)stream>>>(M, N, K, a, lda, w_pack, w_bytes,
                                                                                          bias, C, ldc);
  return (int)hipGetLastError();
}
