// wgrad_split.hip -- every weight gradient of a training step in ONE launch, fp32-accurate on the bf16
// matrix pipe (include/msr3d_hip.h: msr3d_wgrad_split).
//
//     dW[n][k] += sum_m dy[m][n] x[m][k]          db[n] += sum_m dy[m][n]          m = token rows
//
// The reduction runs over the TOKENS, i.e. across all scenes -- the one part of the backward that is not
// scene-local -- so the weight gradients are deferred to the end of the backward (the scene blocks leave
// dy and x of every linear layer behind as dense fp32 side outputs) and computed together: ~650 tiles of
// 128 x 64 in one grid instead of ten launches of 2-3 products whose K-splits met by atomics.  A
// workgroup owns a tile over the WHOLE reduction: no split-K, no atomics, bit-reproducible.
//
// Both operands are contracted over their ROW index, so a fragment (8 consecutive tokens of one column)
// is a strided read: each thread loads 8 tokens x 1 column (a wave: 64 consecutive columns = 256
// contiguous bytes per token), splits the eight values exactly into three bf16 terms (split_mma.h) and
// writes its 16 bytes of each plane straight into fragment order in LDS -- the transpose costs nothing.
// Register-prefetched, double-buffered over slabs of 32 tokens: one barrier per slab.
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "split_mma.h"

namespace {

using namespace msr3d;
using WP = msr3d_wgrad_problem_t;

constexpr int TN = 128, TK = 64;                   // tile: 128 outputs x 64 inputs
constexpr int A_BYTES = (TN / 16) * 3 * 1024;      // 24,576 per slab
constexpr int B_BYTES = (TK / 16) * 3 * 1024;      // 12,288
constexpr int BUF = A_BYTES + B_BYTES;

__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(int nprob, const WP *__restrict__ probs,
                                                            const int *__restrict__ prefix) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 x BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lo = 0, hi = nprob - 1;
  const int t = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
  }
  const WP pr = probs[lo];
  // A problem owns a multiple of 8 workgroups; workgroup b runs on XCD b % 8 (observed dispatch order:
  // speed only), so XCD x takes the CONTIGUOUS run of tiles [x nb/8, (x+1) nb/8): tiles that share a dy
  // block (same output tile, consecutive input tiles) read it through one L2.
  const int lb = t - prefix[lo], nb = prefix[lo + 1] - prefix[lo];
  const int local = (lb & 7) * (nb >> 3) + (lb >> 3);
  const int nkt = (pr.k_in + TK - 1) / TK;
  if (local >= ((pr.n_out + TN - 1) / TN) * nkt) return;
  const int ntile = local / nkt, ktile = local - ntile * nkt;
  const int n0 = ntile * TN, k0 = ktile * TK;
  const int M = pr.M;
  const int nslab = (M + 31) >> 5;

  // loader role: column c (and c + 64 of the dy tile), tokens 8 wave .. 8 wave + 7 of the slab
  const int c = lane;
  const bool a0_ok = n0 + c < pr.n_out, a1_ok = n0 + 64 + c < pr.n_out, b_ok = k0 + c < pr.k_in;
  const float *pa0 = pr.dy + n0 + c, *pa1 = pa0 + 64, *pb = pr.x + k0 + c;
  float va0[8], va1[8], vb[8];
  float cs0 = 0.f, cs1 = 0.f;                      // running column sums of dy (bias gradient)
  auto fetch = [&](int s) {
    const int m0 = 32 * s + 8 * wave;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = m0 + e;
      const bool ok = m < M;
      va0[e] = (ok && a0_ok) ? pa0[(size_t)m * pr.ldy] : 0.f;
      va1[e] = (ok && a1_ok) ? pa1[(size_t)m * pr.ldy] : 0.f;
      vb[e] = (ok && b_ok) ? pb[(size_t)m * pr.ldx] : 0.f;
    }
  };
  auto stash = [&](int buf) {
    unsigned char *A = smem + buf * BUF, *Bs = A + A_BYTES;
    uint4 pl[3];
    const int slot = ((c & 15) + 16 * wave) * 16;   // lane' = column-in-tile + 16 x token group
    sm_split8(va0, pl);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(A + ((c >> 4) * 3 + k) * 1024 + slot) = pl[k];
    sm_split8(va1, pl);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(A + ((4 + (c >> 4)) * 3 + k) * 1024 + slot) = pl[k];
    sm_split8(vb, pl);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(Bs + ((c >> 4) * 3 + k) * 1024 + slot) = pl[k];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs0 += va0[e]; cs1 += va1[e]; }
  };

  // MMA role: wave (wr, wc): output rows 64 wr .. (4 tiles), input columns 32 wc .. (2 tiles)
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(0);
  for (int s = 0; s < nslab; ++s) {
    stash(s & 1);
    __syncthreads();
    if (s + 1 < nslab) fetch(s + 1);
    const unsigned char *A = smem + (s & 1) * BUF, *Bs = A + A_BYTES;
    bf16x8 fa[4][3], fb[2][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        fa[a][k] = *reinterpret_cast<const bf16x8 *>(A + ((4 * wr + a) * 3 + k) * 1024 + lane * 16);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        fb[b][k] = *reinterpret_cast<const bf16x8 *>(Bs + ((2 * wc + b) * 3 + k) * 1024 + lane * 16);
#define MSR3D_TERM(PA, PB)                                                                       \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                \
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][PA], fb[b][PB], acc[a][b], 0, 0, 0);
    MSR3D_TERM(2, 0)
    MSR3D_TERM(0, 2)
    MSR3D_TERM(1, 1)
    MSR3D_TERM(1, 0)
    MSR3D_TERM(0, 1)
    MSR3D_TERM(0, 0)
#undef MSR3D_TERM
  }

  // D[n][k]: lane (j = k column, g): rows n = 4 g + r.  dW holds the value to add to; this workgroup is
  // the tile's only writer.
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kk = k0 + 32 * wc + 16 * b + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * wr + 16 * a + 4 * g + r;
        if (n < pr.n_out && kk < pr.k_in) {
          float *d = pr.dW + (size_t)n * pr.ldw + kk;
          *d += acc[a][b][r];
        }
      }
    }
  if (pr.db && ktile == 0) {                        // bias gradient: the four token groups meet in LDS
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);   // [4 waves][128]
    red[wave * TN + c] = cs0;
    red[wave * TN + 64 + c] = cs1;
    __syncthreads();
    if (tid < TN && n0 + tid < pr.n_out)
      pr.db[n0 + tid] += (red[tid] + red[TN + tid]) + (red[2 * TN + tid] + red[3 * TN + tid]);
  }
}

}  // namespace

extern "C" int msr3d_wgrad_split(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix,
                                 int total_tiles, msr3d_stream_t stream) {
  if (n < 0 || total_tiles < 0) return MSR3D_EINVAL;
  if (n == 0 || total_tiles == 0) return 0;
  if (!problems || !tile_prefix) return MSR3D_EINVAL;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_split_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
  if (attr != hipSuccess) return (int)attr;
  wgrad_split_kernel<<<total_tiles, 256, 2 * BUF, (hipStream_t)stream>>>(n, problems, tile_prefix);
  return (int)hipGetLastError();
}
