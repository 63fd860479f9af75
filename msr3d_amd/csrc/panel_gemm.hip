// panel_gemm.hip -- the LONG-reduction products of the trainable part and every weight gradient:
//
//     forward   ffn_out = h W2^T                 (K = 2048)      obj_proj = embeds Wp^T        (K = 768)
//     dx        d_t += d_pre W1                  (K = 2048)      d_tok += g W_llm              (K = 4096)
//               d_xin += d_qkvc W_qkvc           (K = 816)
//     dW        dW[n_out][k_in] += sum over the 960 tokens of dy[m][n_out] x[m][k_in],  db += colsum(dy)
//
// (/root/reference/modules/layers/transformers.py:200-252,314-329 and their autograd).  All of them
// are "small output, long K" at M = 960 tokens.  Two things bound them on this part, both measured
// (tools/prof_panel.py, tools/bench_multi.py):
//   * K-splits meet by float atomicAdd, and the chip retires only ~1.1 TB/s of them (64-byte requests:
//     16 consecutive floats per row is the best a 16x16 MFMA tile can offer) -- a 960 x 256 output
//     split 16 ways spends 14 us in atomics alone.  So: as FEW K-splits as the workgroup count
//     allows, and none at all where a problem has 240+ tiles of its own (then beta = 1 is a plain
//     read-modify-write by the tile's only owner);
//   * gemm_f32.hip's 64x64 / BK = 32 kernel crosses a barrier and a global -> register -> LDS staging
//     step every 32 k, which is what holds it at 45-55 TFLOP/s on these shapes.
//
// Here a workgroup (256 threads, 4 waves as 2 x 2) owns a 64 x 64 tile of C and a run of K STAGES of
// 128: per stage the A panel (64 x 128) goes through LDS (double-buffered, one barrier per stage) and
// the B panel straight to registers (64 VGPRs per lane, ping-pong), both fetched a whole stage --
// 128 MFMAs per wave -- ahead of their use.  Two such workgroups fit a CU and run out of phase.
//
// Operand layouts (as gemm_f32.hip):  a(m,k) = a_kc ? A[m*lda + k] : A[k*lda + m]
//                                     b(n,k) = b_kc ? B[n*ldb + k] : B[k*ldb + n]
//   a_kc = 0 (dW: A = dy, token-major) is staged as it lies, [k][64 m], and its fragments are read
//   with four ds_read_b32 (conflict-free) instead of one ds_read_b128.
//   b_kc = 0 reads two consecutive COLUMNS per load: MFMA column tile t, lane i <-> column 2 i + t.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/msr3d_hip.h"
#include "panel_gemm.h"

namespace msr3d {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int TM = 64, TN = 64, KS = 128;           // tile rows, tile columns, K stage
constexpr int LDA_K = KS + 8;                       // a_kc = 1: As[m][k], ds_read_b128 conflict-free
constexpr int LDA_M = TM + 4;                       // a_kc = 0: As[k][m], ds_read_b32 conflict-free
constexpr int ABUF = TM * LDA_K > KS * LDA_M ? TM * LDA_K : KS * LDA_M;     // floats per LDS buffer

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// phase marks: empty here; tools/prof/panel_gemm_stamped.hip defines STAMP and includes this file
#ifndef STAMP
#define STAMP(k) do {} while (0)
#endif

// One operand panel of a stage: 64 "output" indices x 128 k.
//   KC layout (index-major in memory, k contiguous):  LDS [64][LDA_K], fragment = one ds_read_b128
//   KR layout (k-major in memory, index contiguous):  LDS [128][LDA_M], fragment = four ds_read_b32
// Both are fetched with row-contiguous 16-byte loads by all 256 threads (512 B / 256 B runs): the
// address path moves whole cache lines, which is what lets two operands stream at the rate the MFMA
// pipe consumes them -- B fragments fetched straight into registers (16 rows x 64 B per instruction,
// and every wave pair fetching the same columns twice) held this kernel at 57 % of the MFMA rate.
template <bool KC>
__device__ __forceinline__ void panel_load(float4 (&r)[8], const float *__restrict__ P, int ld, int o0, int O,
                                           int kbeg, int klen, bool full) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {                  // 64 rows x 128 k: thread t reads (row = t / 32 + 8 j, k4 = t % 32)
      const int row = min(o0 + (tid >> 5) + 8 * j, O - 1), k4 = (tid & 31) * 4;
      if (k4 < klen) v = ld4(P + (size_t)row * ld + kbeg + k4);
    } else {                   // 128 k rows x 64 columns: thread t reads (k = t / 16 + 16 j, c4 = t % 16)
      const int k = (tid >> 4) + 16 * j, c4 = (tid & 15) * 4;
      if (k < klen) {
        const float *src = P + (size_t)(kbeg + k) * ld + o0 + c4;
        if (full) {
          v = ld4(src);
        } else {
          if (o0 + c4 + 0 < O) v.x = src[0];
          if (o0 + c4 + 1 < O) v.y = src[1];
          if (o0 + c4 + 2 < O) v.z = src[2];
          if (o0 + c4 + 3 < O) v.w = src[3];
        }
      }
    }
    r[j] = v;
  }
}
template <bool KC>
__device__ __forceinline__ void panel_store(float *S, const float4 (&r)[8]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (KC) *reinterpret_cast<float4 *>(S + ((tid >> 5) + 8 * j) * LDA_K + (tid & 31) * 4) = r[j];
    else    *reinterpret_cast<float4 *>(S + ((tid >> 4) + 16 * j) * LDA_M + (tid & 15) * 4) = r[j];
  }
}
// fragment of index row `r` for slab s: k = 16 s + 4 g .. + 3
template <bool KC>
__device__ __forceinline__ void panel_frag(const float *S, int r, int s, int g, float (&f)[4]) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4 *>(S + r * LDA_K + 16 * s + 4 * g);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = S[(16 * s + 4 * g + q) * LDA_M + r];
  }
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void panel_body(const PanelP &p, int bx, int by, int bz, float *smem) {
  STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * TM, n0 = bx * TN;
  const int nstage_all = (p.K + KS - 1) / KS;
  const int sbeg = bz * p.spw, send = min(nstage_all, sbeg + p.spw);
  const bool full_m = m0 + TM <= p.M, full_n = n0 + TN <= p.N;
  float *const As = smem, *const Bs = smem + ABUF;

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bias[2] = {0.f, 0.f};
  if (p.bias && bz == 0) {
#pragma unroll
    for (int rn = 0; rn < 2; ++rn) {
      const int col = n0 + 32 * wn + 16 * rn + i;
      bias[rn] = p.bias[col < p.N ? col : p.N - 1];
    }
  }
  float csum = 0.f;                                  // bias gradient partial: thread t < 64 <-> row m0 + t
  const bool do_colsum = !A_KC && p.colsum != nullptr && bx == 0;

  // ---- the stages: operands of stage s + 1 travel (registers) while stage s multiplies (LDS) --------
  float4 ra[8], rb[8];
  auto fetch = [&](int st) {
    const int kbeg = st * KS, klen = min(KS, p.K - kbeg);
    panel_load<A_KC>(ra, p.A, p.lda, m0, p.M, kbeg, klen, full_m);
    panel_load<B_KC>(rb, p.B, p.ldb, n0, p.N, kbeg, klen, full_n);
  };
  fetch(sbeg);
  STAMP(1);
  for (int st = sbeg; st < send; ++st) {
    panel_store<A_KC>(As, ra);
    panel_store<B_KC>(Bs, rb);
    __syncthreads();
    if (st == sbeg) STAMP(2);
    fetch(st + 1 < send ? st + 1 : st);               // (past the end: a re-read nobody uses)
    const int nslab = (min(KS, p.K - st * KS)) >> 4;
    if (do_colsum && tid < TM) {                     // colsum[m] += sum_k a(m, k), from the staged panel
      for (int k = 0; k < nslab * 16; ++k) csum += As[k * LDA_M + tid];
    }
    auto slab = [&](int s) {
      float fa[2][4], fb[2][4];
#pragma unroll
      for (int rm = 0; rm < 2; ++rm) panel_frag<A_KC>(As, wm * 32 + rm * 16 + i, s, g, fa[rm]);
#pragma unroll
      for (int rn = 0; rn < 2; ++rn) panel_frag<B_KC>(Bs, wn * 32 + rn * 16 + i, s, g, fb[rn]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rn = 0; rn < 2; ++rn)
#pragma unroll
          for (int rm = 0; rm < 2; ++rm)
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[rm][q], fb[rn][q], acc[rm][rn], 0, 0, 0);
    };
    if (nslab == 8) {                                  // a whole stage: one straight-line block
#pragma unroll
      for (int s = 0; s < 8; ++s) slab(s);
    } else {                                           // the reduction's tail
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < nslab) slab(s);
    }
    __syncthreads();                                   // everyone is done reading before the next stores
  }
  STAMP(3);

  // ---- epilogue: C/D map col = lane & 15 (tile column), row = (lane >> 4) * 4 + reg ---------------
  // one K run (gz == 1): the tile has a single owner -- plain store (beta 0) or read-modify-write
  // (beta 1); several: atomicAdd onto zeros / the value to add to.  The atomic branch contains no
  // load (the bias was fetched at the start): a load among them makes the compiler drain vmcnt before
  // every atomic, i.e. serialises them on the round trip (14k cycles per workgroup, tools/prof_panel.py)
  const bool owner = p.gz == 1;
  int cols[2];
  bool cok[2];
#pragma unroll
  for (int rn = 0; rn < 2; ++rn) {
    cols[rn] = n0 + 32 * wn + 16 * rn + i;
    cok[rn] = cols[rn] < p.N;
  }
  const int rowb = m0 + wm * 32 + g * 4;
  if (owner) {
    float old[2][2][4];
#pragma unroll
    for (int rn = 0; rn < 2; ++rn)
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb + rm * 16 + r;
          old[rm][rn][r] = (p.beta != 0.f && cok[rn] && row < p.M) ? p.C[(size_t)row * p.ldc + cols[rn]] : 0.f;
        }
#pragma unroll
    for (int rn = 0; rn < 2; ++rn)
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb + rm * 16 + r;
          if (cok[rn] && row < p.M) p.C[(size_t)row * p.ldc + cols[rn]] = old[rm][rn][r] + acc[rm][rn][r] + bias[rn];
        }
  } else {
#pragma unroll
    for (int rn = 0; rn < 2; ++rn)
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb + rm * 16 + r;
          if (cok[rn] && row < p.M) atomicAdd(p.C + (size_t)row * p.ldc + cols[rn], acc[rm][rn][r] + bias[rn]);
        }
  }
  if (do_colsum && tid < TM && m0 + tid < p.M) atomicAdd(p.colsum + m0 + tid, csum);
  STAMP(4);
}

__global__ __launch_bounds__(256, 2) void panel_multi_kernel(const PanelBatch pb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int id = blockIdx.x, q = 0;
#pragma unroll
  for (int j = 1; j < MSR3D_GEMM_MULTI_MAX; ++j)
    if (j < pb.n && id >= pb.first[j]) q = j;
  id -= pb.first[q];
  const PanelP &p = pb.p[q];
  // XCD-aware order (workgroup b runs on XCD b % 8, each XCD has its own L2): the tiles that read the
  // SAME streamed panel -- the `nm` tiles along the output's short side, for one K run -- get ids that
  // are equal mod 8 and 8 apart, i.e. run on one XCD at about the same time, so the panel crosses the
  // fabric once instead of nm times (at 64 x 64 tiles these products are otherwise L2-miss bound:
  // 16 FLOP per byte)
  const bool share_a = p.gx <= p.gy;                   // members differ in bx and share the A panel
  const int nm = share_a ? p.gx : p.gy;
  const int ng = (share_a ? p.gy : p.gx) * p.gz;
  int member, grp;
  const int tail = (ng / 8) * 8 * nm;
  if (id < tail) {
    const int rem = id % (8 * nm);
    member = rem / 8;
    grp = (id / (8 * nm)) * 8 + (rem & 7);
  } else {
    const int r = id - tail, nl = ng % 8;
    member = r / nl;
    grp = (ng / 8) * 8 + r % nl;
  }
  const int bz = grp % p.gz, outer = grp / p.gz;
  const int bx = share_a ? member : outer, by = share_a ? outer : member;
  switch (p.kind) {
    case 3: panel_body<true, true>(p, bx, by, bz, smem); break;
    case 2: panel_body<true, false>(p, bx, by, bz, smem); break;
    case 1: panel_body<false, true>(p, bx, by, bz, smem); break;
    default: panel_body<false, false>(p, bx, by, bz, smem); break;
  }
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

bool panel_eligible(const msr3d_gemm_problem_t &q) {
  if (q.M <= 0 || q.N <= 0 || q.K < 16 || (q.K % 16) != 0) return false;
  if (q.N % 4 || q.lda % 4 || q.ldb % 4 || !al16(q.A) || !al16(q.B)) return false;
  if (!q.b_kc && (q.ldb < q.N)) return false;
  if (!q.a_kc && q.lda < q.M) return false;
  if (q.beta != 0.f && q.beta != 1.f) return false;
  if (q.colsum && (q.a_kc || q.beta != 1.f)) return false;
  // worth it only where the reduction is long enough to stage -- unless the caller asked for the one-run
  // schedule (results must not depend on how many rows share the launch)
  return q.single_run || (long long)q.M * q.N * q.K >= (1ll << 24);
}

void panel_shape(const msr3d_gemm_problem_t &q, int *tiles, int *stages) {
  *tiles = ((q.N + TN - 1) / TN) * ((q.M + TM - 1) / TM);
  *stages = (q.K + KS - 1) / KS;
}

int panel_plan(const msr3d_gemm_problem_t &q, int stages_per_run, PanelP *out, hipStream_t st) {
  PanelP p;
  p.M = q.M; p.N = q.N; p.K = q.K;
  p.A = q.A; p.lda = q.lda; p.B = q.B; p.ldb = q.ldb; p.C = q.C; p.ldc = q.ldc;
  p.bias = q.bias; p.colsum = q.colsum; p.beta = q.beta;
  p.kind = (q.a_kc ? 2 : 0) + (q.b_kc ? 1 : 0);
  p.gx = (q.N + TN - 1) / TN; p.gy = (q.M + TM - 1) / TM;
  const int stages = (q.K + KS - 1) / KS;
  // K runs of ~stages_per_run stages (the caller balances the problems of a launch): every extra run
  // costs the tile's size again in atomics, so never shorter than two stages
  int spw = stages_per_run < 2 ? 2 : stages_per_run;
  if (spw > stages || q.single_run) spw = stages;
  int gz = (stages + spw - 1) / spw;
  spw = (stages + gz - 1) / gz;                      // even the runs out
  p.spw = spw;
  p.gz = (stages + spw - 1) / spw;
  if (p.gz > 1 && q.beta == 0.f) {
    // atomic meeting point on a C that is not known to be zero: clear it first (callers on the hot
    // path pass beta = 1 over a region msr3d_step_begin zeroed, and never get here)
    if (q.ldc != q.N) return MSR3D_EINVAL;
    const hipError_t e = hipMemsetAsync(q.C, 0, sizeof(float) * (size_t)q.M * q.N, st);
    if (e != hipSuccess) return (int)e;
    p.beta = 1.f;
  }
  *out = p;
  return 0;
}

int panel_launch(const PanelBatch &pb, int blocks, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * 2 * ABUF;
  static bool done = false;
  if (!done) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_multi_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  panel_multi_kernel<<<blocks, 256, lds, st>>>(pb);
  return (int)hipGetLastError();
}

}  // namespace msr3d
