// gemm_f32.hip -- token GEMMs of the trainable part (situated encoder + projector) on
// f32-input MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products/sums, like the reference,
// which runs this part with autocast disabled (model/ose3d_situation.py:377).
//
// One kernel, three operand layouts, so that forward AND both backward products read the
// tensors where they already are (no transposed copies):
//     C[m][n] (+)= sum_k a(m,k) * b(n,k)
//     a(m,k) = A_KC ? A[m*lda + k] : A[k*lda + m]        (KC = reduction index contiguous)
//     b(n,k) = B_KC ? B[n*ldb + k] : B[k*ldb + n]
//   forward  y  = x W^T + b      : A = x  (KC),  B = W (KC)         (nn.Linear layout W[N][K])
//   backward dx = dy W           : A = dy (KC),  B = W (not KC)     reduction over N
//   backward dW = dy^T x         : A = dy (not KC), B = x (not KC)  reduction over tokens
//
// These are SMALL problems (M = 960 tokens at 16 scenes/GPU; 0.1 - 2 GFLOP each): what
// matters is that every launch fills the chip for its few microseconds: 64x64 tiles (4 waves
// x 32x32) with split-K for the 256-wide outputs, 128-row / 128-column tiles where a side is
// long; BK = 32, LDS double-buffered, operands staged global -> registers -> LDS with 16-byte
// accesses.  Forward linears with K <= 256 and N >= 512 take gemm_nt_ares_kernel instead (x strip
// resident in LDS, W fragments straight from L2, no per-slab barrier).  Split-K partial sums meet in C by atomicAdd (default: fastest, rounding depends on
// arrival order) or, when the caller passes a workspace, in an ORDERED hand-over: every split
// stores its accumulators, takes a ticket, and the last workgroup to arrive adds the partials in
// split order and runs the epilogue -- no zero-fill of C, no float atomics, bit-reproducible
// results, at +5 % step time (the records cross the 8 L2s through device-coherent accesses).
// hipBLASLt's heuristic picks a
// 256x256 macro-tile = ONE workgroup for the (960 x 256 x 256) fp32 linears of this path
// (215 us each, profiles/r01_v2_bench_kernel_stats.csv); this kernel is the replacement.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/msr3d_hip.h"
#include "dropout_rng.h"
#include "panel_gemm.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 32;
// [TR][36]: 16 consecutive rows start 36 floats apart = banks {0,36,8,44,...} mod 64, sixteen distinct
// multiples of 4 -> a 16-lane ds_read_b128 phase covers all 64 banks once.  (A stride of 40 is
// conflict-free too but makes the 64x64 double buffer exactly 40 KB, i.e. 4 x 40 KB = ALL of the
// CU's LDS: with the kernel's few static bytes only 3 workgroups fitted; at 36 KB four do, and the
// kernel is held to 128 registers so that four waves per SIMD are resident: 2.45 -> 2.35 ms/step.)
constexpr int LD_KC = BK + 4;

__host__ __device__ constexpr int ld_mc(int tr) { return tr + 4; }   // [32][TR+4]: half-waves on disjoint banks
__host__ __device__ constexpr int tile_floats(int tr) { return tr * LD_KC; }   // >= 32 * (tr + 4) for tr >= 64

__device__ __forceinline__ float gelu_f(float x) {     // exact erf GELU (F.gelu default)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// ---- split-K workspace accesses: relaxed, agent scope (coherent across the XCDs' L2s) -----
__device__ __forceinline__ void ws_store(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ws_load(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ws_store2(float *p, float a, float b) {     // p 8-byte aligned
  const unsigned long long bits =
      (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ws_load2(const float *p, float &a, float &b) {
  const unsigned long long bits = __hip_atomic_load(
      reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = __uint_as_float((unsigned)bits);
  b = __uint_as_float((unsigned)(bits >> 32));
}

// ---- global -> registers: one TR x 32 operand tile = TR/32 float4 per thread ------------
// KC tile: rows = output index (TR), cols = k (32).   thread t: row = t/8 + 32p, k4 = (t%8)*4
// MC tile: rows = k (32), cols = output index (TR).   thread t: krow = t/(TR/4) + (1024/TR) p,
//                                                                 c4 = (t % (TR/4)) * 4
template <bool KC, int TR>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int ld, int o0, int k0,
                                          int O, int K, bool vec_ok, float4 (&r)[TR / 32]) {
  const int t = threadIdx.x;
  constexpr int TPR = TR / 4;            // threads per k-row in the MC layout
#pragma unroll
  for (int p = 0; p < TR / 32; ++p) {
    int row, col;          // row/col in GLOBAL matrix terms
    if (KC) { row = o0 + (t >> 3) + 32 * p; col = k0 + (t & 7) * 4; }
    else    { row = k0 + t / TPR + (256 / TPR) * p; col = o0 + (t % TPR) * 4; }
    const int R = KC ? O : K, Cn = KC ? K : O;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < R) {
      const float *src = P + (size_t)row * ld + col;
      if (vec_ok && col + 3 < Cn) {
        v = *reinterpret_cast<const float4 *>(src);
      } else {
        if (col + 0 < Cn) v.x = src[0];
        if (col + 1 < Cn) v.y = src[1];
        if (col + 2 < Cn) v.z = src[2];
        if (col + 3 < Cn) v.w = src[3];
      }
    }
    r[p] = v;
  }
}

template <bool KC, int TR>
__device__ __forceinline__ void store_tile(float *s, const float4 (&r)[TR / 32]) {
  const int t = threadIdx.x;
  constexpr int TPR = TR / 4;
#pragma unroll
  for (int p = 0; p < TR / 32; ++p) {
    if (KC) *reinterpret_cast<float4 *>(s + ((t >> 3) + 32 * p) * LD_KC + (t & 7) * 4) = r[p];
    else    *reinterpret_cast<float4 *>(s + (t / TPR + (256 / TPR) * p) * ld_mc(TR) + (t % TPR) * 4) = r[p];
  }
}

// fragment for MFMA step s of 16-wide sub-slab `sub`: element (row = base + i, k = 16 sub + 4g + s)
template <bool KC, int TR>
__device__ __forceinline__ void read_frag(const float *s, int base, int sub, int i, int g,
                                          float (&f)[4]) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4 *>(s + (base + i) * LD_KC + sub * 16 + 4 * g);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = s[(sub * 16 + 4 * g + q) * ld_mc(TR) + base + i];
  }
}

// One problem of a launch: operands, epilogue switches and the (virtual) grid it was sized for.
struct GemmP {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float *C; int ldc;
  const float *bias;
  float *Cpre;
  int flags;            // bit0 GELU, bit1 column sums accumulate, bit2 dropout
  float beta;
  int a_vec, b_vec, slabs_per_split;
  float *a_colsum;
  float *ws_part; int *ws_count;
  float p_drop; const unsigned long long *seed; unsigned salt;
  int gx, gy, gz;       // tiles along N, tiles along M, K-splits
};

// Workgroup tile (32 RM) x (32 RN): 2 x 2 waves of (16 RM) x (16 RN) each.  (bx, by, bz) is the
// workgroup's place in the problem's own grid, so several problems can share one launch.
template <bool A_KC, bool B_KC, int RM, int RN>
__device__ __forceinline__ void gemm_body(const GemmP &p, const int bx, const int by, const int bz) {
  const int M = p.M, N = p.N, K = p.K, lda = p.lda, ldb = p.ldb, ldc = p.ldc;
  const float *__restrict__ A = p.A;
  const float *__restrict__ B = p.B;
  float *__restrict__ C = p.C;
  const float *__restrict__ bias = p.bias;
  float *__restrict__ Cpre = p.Cpre;
  const int flags = p.flags, a_vec = p.a_vec, b_vec = p.b_vec, slabs_per_split = p.slabs_per_split;
  const float beta = p.beta, p_drop = p.p_drop;
  float *__restrict__ a_colsum = p.a_colsum;
  float *__restrict__ ws_part = p.ws_part;
  int *__restrict__ ws_count = p.ws_count;
  const unsigned long long *__restrict__ seed = p.seed;
  const unsigned salt = p.salt;
  constexpr int BM = 32 * RM, BN = 32 * RN;
  constexpr int TA = tile_floats(BM), TB = tile_floats(BN);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const As = smem;                 // [2][TA]
  float *const Bs = smem + 2 * TA;        // [2][TB]

  const int m0 = by * BM, n0 = bx * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 15, g = lane >> 4;

  f32x4 acc[RM][RN];
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split-K: bz owns slabs [kbeg, kend) (never empty: the host sizes p.gz so)
  const int nk_all = (K + BK - 1) / BK;
  const int kbeg = bz * slabs_per_split;
  const int kend = min(nk_all, kbeg + slabs_per_split);
  const int nk = kend - kbeg;

  // bias gradient for free: in the dW product (A = dy, k-strided) the A tiles of the first
  // column of workgroups stream every dy element exactly once; thread t owns 4 columns
  // (t % (BM/4)) * 4 .. +3 of the tile for BM/32 of the 32 k-rows per slab.
  const bool do_colsum = (!A_KC) && a_colsum != nullptr && bx == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

  auto fetch = [&](int kt, float4 (&ra)[BM / 32], float4 (&rb)[BN / 32]) {
    load_tile<A_KC, BM>(A, lda, m0, (kbeg + kt) * BK, M, K, a_vec, ra);
    load_tile<B_KC, BN>(B, ldb, n0, (kbeg + kt) * BK, N, K, b_vec, rb);
  };
  auto stage = [&](int buf, const float4 (&ra)[BM / 32], const float4 (&rb)[BN / 32]) {
    if (do_colsum) {
#pragma unroll
      for (int p2 = 0; p2 < BM / 32; ++p2) { csum.x += ra[p2].x; csum.y += ra[p2].y; csum.z += ra[p2].z; csum.w += ra[p2].w; }
    }
    store_tile<A_KC, BM>(As + buf * TA, ra);
    store_tile<B_KC, BN>(Bs + buf * TB, rb);
  };
  auto compute = [&](int cur) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float fa[RM][4], fb[RN][4];
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
        read_frag<A_KC, BM>(As + cur * TA, wm * (16 * RM) + rm * 16, sub, i, g, fa[rm]);
#pragma unroll
      for (int rn = 0; rn < RN; ++rn)
        read_frag<B_KC, BN>(Bs + cur * TB, wn * (16 * RN) + rn * 16, sub, i, g, fb[rn]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn)
            acc[rm][rn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[rm][s], fb[rn][s], acc[rm][rn], 0, 0, 0);
    }
  };

  // Two register stages: while slab kt is multiplied out of LDS, slab kt+1 waits in registers and
  // slab kt+2 is in flight.  Measured against the one-stage loop: steady-state 285 -> 267 us
  // (M 960, N 2048, K 4096), +1 % on the full step.  Two cleaner-looking variants were slower and
  // are not kept: a branch-free load path for interior tiles (304-318 us) and a single-exit loop.
  float4 ra0[BM / 32], rb0[BN / 32], ra1[BM / 32], rb1[BN / 32];
  fetch(0, ra0, rb0);
  if (nk > 1) fetch(1, ra1, rb1);
  stage(0, ra0, rb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) fetch(kt + 2, ra0, rb0);
    compute(0);
    if (kt + 1 < nk) stage(1, ra1, rb1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) fetch(kt + 3, ra1, rb1);
    compute(1);
    if (kt + 2 < nk) stage(0, ra0, rb0);
    __syncthreads();
  }

  const bool ordered = p.gz > 1 && ws_part != nullptr;   // workspace meeting point
  const int tile = by * p.gx + bx;
  constexpr int REC = BM * BN + BM;                  // one split's record: accumulators + column sums
  float *const rec0 = ordered ? ws_part + (size_t)tile * p.gz * REC : nullptr;
  float col_tot = 0.f;                               // thread c < BM: sum of column m0 + c
  if (do_colsum) {
    // 256 / (BM/4) threads share each group of 4 columns: reduce their partial sums through LDS
    constexpr int TPR = BM / 4;
    float *red = smem;                               // all MFMA reads are behind the last barrier
    *reinterpret_cast<float4 *>(red + threadIdx.x * 4) = csum;
    __syncthreads();
    if (threadIdx.x < BM) {
      const int c = threadIdx.x;                     // column of the tile
#pragma unroll
      for (int j = 0; j < 256 / TPR; ++j) col_tot += red[(j * TPR + (c >> 2)) * 4 + (c & 3)];
      if (ordered) ws_store(rec0 + (size_t)bz * REC + BM * BN + c, col_tot);
      else if (p.gz > 1) { if (m0 + c < M) atomicAdd(a_colsum + m0 + c, col_tot); }
    }
  }

  if (ordered) {
    // The record travels through device-coherent (agent-scope, write-through / cache-bypassing)
    // accesses, so the hand-over needs no L2 write-back + invalidate -- a __threadfence() pair per
    // workgroup costs more than the whole GEMM on this 8-L2 part.
    float *mine = rec0 + (size_t)bz * REC;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const f32x4 a = acc[rm][rn];
        float *d = mine + ((rm * RN + rn) * 256 + threadIdx.x) * 4;
        ws_store2(d, a[0], a[1]);
        ws_store2(d + 2, a[2], a[3]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // own stores complete (vmcnt 0)
    __syncthreads();
    __shared__ int ticket;
    if (threadIdx.x == 0)
      ticket = __hip_atomic_fetch_add(ws_count + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != p.gz - 1) return;        // not the last split of this tile
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x == 0)                            // ready for the next launch
      __hip_atomic_store(ws_count + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) acc[rm][rn] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.gz; ++z) {       // fixed order: bit-reproducible
      const float *part = rec0 + (size_t)z * REC;
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          const float *q = part + ((rm * RN + rn) * 256 + threadIdx.x) * 4;
          float v0, v1, v2, v3;
          ws_load2(q, v0, v1);
          ws_load2(q + 2, v2, v3);
          acc[rm][rn][0] += v0; acc[rm][rn][1] += v1; acc[rm][rn][2] += v2; acc[rm][rn][3] += v3;
        }
    }
    if (do_colsum && threadIdx.x < BM) {
      col_tot = 0.f;
      for (int z = 0; z < p.gz; ++z) col_tot += ws_load(rec0 + (size_t)z * REC + BM * BN + threadIdx.x);
    }
  }
  // single writer per column from here on (one split, or the last arriver of the ordered path)
  if (do_colsum && threadIdx.x < BM && (ordered || p.gz == 1) && m0 + (int)threadIdx.x < M) {
    float *dst = a_colsum + m0 + threadIdx.x;
    *dst = (flags & 2) ? *dst + col_tot : col_tot;   // flags bit 1: accumulate onto db
  }

  // epilogue: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg
  const bool drop = (flags & 4) != 0;                 // dropout on the final value (after GELU)
  const unsigned thresh = msr3d::drop_thresh(drop ? p_drop : 0.f);
  const float dscale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned long long sd = drop ? *seed : 0ull;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const int col = n0 + wn * (16 * RN) + rn * 16 + i;
    if (col >= N) continue;
    const float bv = (bias && (ordered || bz == 0)) ? bias[col] : 0.f;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * (16 * RM) + rm * 16 + g * 4 + r;
        if (row >= M) continue;
        float v = acc[rm][rn][r] + bv;
        const size_t o = (size_t)row * ldc + col;
        if (p.gz > 1 && !ordered) {   // C was zeroed (beta == 0) or holds the value to add to
          atomicAdd(C + o, v);
          continue;
        }
        if (beta != 0.f) v += beta * C[o];
        if (flags & 1) {
          if (Cpre) Cpre[o] = v;
          v = gelu_f(v);
        }
        if (drop) v = msr3d::keep_elem(sd, salt, (unsigned)o, thresh) ? v * dscale : 0.f;
        C[o] = v;
      }
  }
}

template <bool A_KC, bool B_KC, int RM, int RN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_f32_kernel(const GemmP p) {
  gemm_body<A_KC, B_KC, RM, RN>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------
// Forward linears with a short reduction (K <= 256: q/k/v/cond, fc, linear1, llm_proj), A-resident:
// the workgroup's 64-row strip of x is staged in LDS ONCE (all its loads in flight together, one
// barrier), then the K loop runs barrier-free with the W fragments read straight from L2 into
// registers -- the structure of the set-abstraction kernels (sa_fused.hip).  No per-slab staging of
// W through LDS, no per-slab barrier: what these launches pay for is latency, not bandwidth.
// Waves are laid out 1 x 4: each owns all 64 rows (RM = 4) and 16 RN of the workgroup's 64 RN columns.
// ---------------------------------------------------------------------------------------------
template <int RN>
__global__ __launch_bounds__(256) void gemm_nt_ares_kernel(const GemmP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RM = 4;
  const int M = p.M, N = p.N, K = p.K, LDA = K + 8;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * (64 * RN);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const float *__restrict__ A = p.A;
  const float *__restrict__ W = p.B;

  // this lane's W rows (column tiles of the output); columns past N re-read row N-1, results dropped
  const int colbase = n0 + wave * (16 * RN);
  const float *wp[RN];
  float4 bcur[RN];
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    int col = colbase + rn * 16 + i;
    col = col < N ? col : N - 1;
    wp[rn] = W + (size_t)col * p.ldb + 4 * g;
    bcur[rn] = *reinterpret_cast<const float4 *>(wp[rn]);
  }
  // A strip: 64 x K floats, K/4 float4 per row, batches of 8 loads per thread
  const int k4 = K >> 2, total = 64 * k4;
  for (int e0 = 0; e0 < total; e0 += 256 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256 + tid;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < total) {
        const int row = e / k4, c4 = e - row * k4;
        if (m0 + row < M) v[u] = *reinterpret_cast<const float4 *>(A + (size_t)(m0 + row) * p.lda + c4 * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256 + tid;
      if (e < total) {
        const int row = e / k4, c4 = e - row * k4;
        *reinterpret_cast<float4 *>(smem + row * LDA + c4 * 4) = v[u];
      }
    }
  }
  __syncthreads();

  f32x4 acc[RM][RN];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) acc[rm][rn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float *xp = smem + i * LDA + 4 * g;
  for (int k0 = 0; k0 < K; k0 += 16) {
    float4 bnext[RN], a[RM];
    const int kn = (k0 + 16 < K) ? k0 + 16 : k0;
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) bnext[rn] = *reinterpret_cast<const float4 *>(wp[rn] + kn);
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) a[rm] = *reinterpret_cast<const float4 *>(xp + rm * 16 * LDA + k0);
#define MSR3D_STEP(c)                                                                           \
    _Pragma("unroll") for (int rm = 0; rm < RM; ++rm)                                           \
    _Pragma("unroll") for (int rn = 0; rn < RN; ++rn)                                           \
        acc[rm][rn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rm].c, bcur[rn].c, acc[rm][rn], 0, 0, 0);
    MSR3D_STEP(x)
    MSR3D_STEP(y)
    MSR3D_STEP(z)
    MSR3D_STEP(w)
#undef MSR3D_STEP
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) bcur[rn] = bnext[rn];
  }

  // epilogue (same order of operations as gemm_body's): bias, beta * C, GELU (+ pre-activation), dropout
  const int flags = p.flags;
  const bool drop = (flags & 4) != 0;
  const unsigned thresh = msr3d::drop_thresh(drop ? p.p_drop : 0.f);
  const float dscale = drop ? 1.0f / (1.0f - p.p_drop) : 1.0f;
  const unsigned long long sd = drop ? *p.seed : 0ull;
  float *__restrict__ C = p.C;
  float *__restrict__ Cpre = p.Cpre;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const int col = colbase + rn * 16 + i;
    if (col >= N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + rm * 16 + g * 4 + r;
        if (row >= M) continue;
        float v = acc[rm][rn][r] + bv;
        const size_t o = (size_t)row * p.ldc + col;
        if (p.beta != 0.f) v += p.beta * C[o];
        if (flags & 1) {
          if (Cpre) Cpre[o] = v;
          v = gelu_f(v);
        }
        if (drop) v = msr3d::keep_elem(sd, p.salt, (unsigned)o, thresh) ? v * dscale : 0.f;
        C[o] = v;
      }
  }
}

template <int RN>
static hipError_t launch_ares(const GemmP &p, hipStream_t st) {
  const size_t lds = sizeof(float) * 64 * (size_t)(p.K + 8);
  static size_t granted = 0;
  if (lds > 64 * 1024 && lds > granted) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_nt_ares_kernel<RN>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    granted = lds;
  }
  dim3 grid((p.N + 64 * RN - 1) / (64 * RN), (p.M + 63) / 64);
  gemm_nt_ares_kernel<RN><<<grid, 256, lds, st>>>(p);
  return hipGetLastError();
}

// Both backward products of a linear layer in ONE launch: dx = dy W (first p1's workgroups) and
// dW += dy^T x, db += colsum(dy) (then p2's).  They are independent, each too small to fill the
// chip, and every launch inside the captured graph costs ~5 us whatever it does.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_linear_bwd_kernel(const GemmP p1, const GemmP p2) {
  int id = blockIdx.x;
  const int n1 = p1.gx * p1.gy * p1.gz;
  if (id < n1) {
    gemm_body<true, false, 2, 2>(p1, id % p1.gx, (id / p1.gx) % p1.gy, id / (p1.gx * p1.gy));
  } else {
    id -= n1;
    gemm_body<false, false, 2, 2>(p2, id % p2.gx, (id / p2.gx) % p2.gy, id / (p2.gx * p2.gy));
  }
}

// Up to MSR3D_GEMM_MULTI_MAX independent problems in ONE launch (any mix of the three operand
// layouts): the backward of a linear layer -- dx, dW + db -- together with whatever other weight
// gradient is ready at that point of the schedule.  Each problem alone is too small to fill the
// chip; the workgroup index picks the problem, then its tile and K-split.
struct GemmBatch {
  GemmP p[MSR3D_GEMM_MULTI_MAX];
  int first[MSR3D_GEMM_MULTI_MAX + 1];   // first workgroup of each problem, then the total
  int kind[MSR3D_GEMM_MULTI_MAX];        // a_kc * 2 + b_kc
  int n;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_multi_kernel(const GemmBatch gb) {
  int id = blockIdx.x, q = 0;
#pragma unroll
  for (int j = 1; j < MSR3D_GEMM_MULTI_MAX; ++j)
    if (j < gb.n && id >= gb.first[j]) q = j;
  id -= gb.first[q];
  const GemmP &p = gb.p[q];
  const int bx = id % p.gx, by = (id / p.gx) % p.gy, bz = id / (p.gx * p.gy);
  switch (gb.kind[q]) {
    case 3: gemm_body<true, true, 2, 2>(p, bx, by, bz); break;
    case 2: gemm_body<true, false, 2, 2>(p, bx, by, bz); break;
    case 1: gemm_body<false, true, 2, 2>(p, bx, by, bz); break;
    default: gemm_body<false, false, 2, 2>(p, bx, by, bz); break;
  }
}

// out[n] (+)= sum_m X[m][n]   (bias gradients).  One wave per 64 columns, rows split over
// gridDim.y with one atomicAdd per (column, row-chunk) unless a single chunk covers M.
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float *__restrict__ X,
                                                     int ldx, float *__restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + wave; r < r1; r += 4) s += X[(size_t)r * ldx + col];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && col < N) {
    const float v = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    atomicAdd(out + col, v);
  }
}

// dpre = dy * gelu'(pre)   (exact erf form), elementwise, float4
__global__ void gelu_bwd_kernel(long long n4, const float4 *__restrict__ dy,
                                const float4 *__restrict__ pre, float4 *__restrict__ out,
                                float p_drop, const unsigned long long *__restrict__ seed,
                                unsigned salt) {
  const bool drop = p_drop > 0.f;                    // the forward's epilogue dropout, regenerated
  const unsigned thresh = msr3d::drop_thresh(p_drop);
  const float dscale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const unsigned long long sd = drop ? *seed : 0ull;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    float4 d = dy[t];
    const float4 x = pre[t];
    if (drop) {
      const unsigned base = (unsigned)(t * 4);
      d.x = msr3d::keep_elem(sd, salt, base + 0, thresh) ? d.x * dscale : 0.f;
      d.y = msr3d::keep_elem(sd, salt, base + 1, thresh) ? d.y * dscale : 0.f;
      d.z = msr3d::keep_elem(sd, salt, base + 2, thresh) ? d.z * dscale : 0.f;
      d.w = msr3d::keep_elem(sd, salt, base + 3, thresh) ? d.w * dscale : 0.f;
    }
    float4 o;
    const float k0 = 0.70710678118654752440f, k1 = 0.39894228040143267794f;   // 1/sqrt2, 1/sqrt(2pi)
#define GB(c) o.c = d.c * (0.5f * (1.0f + erff(x.c * k0)) + x.c * k1 * __expf(-0.5f * x.c * x.c))
    GB(x); GB(y); GB(z); GB(w);
#undef GB
    out[t] = o;
  }
}

inline bool vec_ok(const float *p, int ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (ld % 4 == 0);
}

template <bool AK, bool BKc, int RM, int RN>
static hipError_t launch_gemm(const GemmP &p, hipStream_t st) {
  constexpr size_t lds = sizeof(float) * 2 * (tile_floats(32 * RM) + tile_floats(32 * RN));
  auto kern = gemm_f32_kernel<AK, BKc, RM, RN>;
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      done = true;
    }
  }
  kern<<<dim3(p.gx, p.gy, p.gz), 256, lds, st>>>(p);
  return hipGetLastError();
}

// Sizes one problem: tile shape, K-splits, meeting point, and the zero-fill the atomic meeting
// point needs (issued here).  Returns 0, MSR3D_EINVAL or a hipError_t; *empty = nothing to launch.
static int plan_gemm(int a_kc, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                     float *C, int ldc, const float *bias, float *C_pre, int flags, float beta,
                     float *a_colsum, void *workspace, size_t workspace_bytes, float p_drop,
                     const unsigned long long *seed, unsigned salt, bool allow_big_tiles,
                     hipStream_t st, GemmP *out, int *rm_out, int *rn_out, bool *empty, int target_override = 0) {
  *empty = true;
  if (M < 0 || N < 0 || K < 0 || lda <= 0 || ldb <= 0 || ldc <= 0) return MSR3D_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!A || !B || !C) return MSR3D_EINVAL;
  *empty = false;
  // Tile choice.  Measured on the path's shapes (tools/bench_gemm.py, tools/ablate_gemm.py): the
  // staging path sustains ~10 B/clk/CU, so a 64x64 tile (16 FLOP/B) is load-bound at ~60 TFLOP/s
  // in steady state; 128-wide tiles halve the traffic but these problems (M = 960 tokens, 0.1-2
  // GFLOP) then have too few workgroups to cover the load latency and time WORSE (ffn1 forward
  // 27 -> 51 us).  So: 64x64 everywhere, split-K only when there are at most 256 tiles.  The
  // larger tiles are not selected at these sizes.
  auto ntiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  int rm = 2, rn = 2;
  // ~2 workgroups per CU; the 2-GFLOP problems (llm_proj's dx and dW) time better with 4 per CU
  // (54.6 -> 47.8 us, 52.8 -> 46.6 us), the 1-GFLOP ones do not care, more than that is worse for all
  const int target_wgs = target_override > 0 ? target_override : ((double)M * N * K >= 0.75e9 ? 1024 : 512);
  (void)allow_big_tiles;           // (the 128-wide tile instantiations remain for callers that ask for them by shape)
  const int BM = 32 * rm, BN = 32 * rn;
  const int tiles = ntiles(BM, BN);
  const int slabs = (K + BK - 1) / BK;
  // Split K until ~2 workgroups per CU exist, keeping >= 2 slabs per split; beta must be 0 or 1
  // for the atomic meeting point.
  int splits = 1;
  // (measured in the full step: requiring >= 9 / 17 / 33 slabs before splitting costs 1 / 2 / 16 %)
  if ((beta == 0.f || beta == 1.f) && tiles <= 256 && slabs >= 4) {
    splits = (target_wgs + tiles - 1) / tiles;
    if (splits > slabs / 2) splits = slabs / 2;
    if (splits < 1) splits = 1;
  }
  // Meeting point of the splits.  With a workspace ([MSR3D_GEMM_WS_COUNTERS ints, zero before the
  // first use and left zero by every launch][records]) the last workgroup of a tile adds the
  // partials in order and runs the epilogue; without one the splits atomicAdd into a zeroed C.
  float *ws_part = nullptr;
  int *ws_count = nullptr;
  const size_t rec_bytes = sizeof(float) * ((size_t)BM * BN + BM);
  const size_t ws_head = sizeof(int) * MSR3D_GEMM_WS_COUNTERS;
  if (splits > 1 && workspace && workspace_bytes > ws_head && tiles <= MSR3D_GEMM_WS_COUNTERS) {
    const size_t fit = (workspace_bytes - ws_head) / (rec_bytes * (size_t)tiles);
    if (fit >= 2) {
      if ((size_t)splits > fit) splits = (int)fit;
      ws_count = reinterpret_cast<int *>(workspace);
      ws_part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + ws_head);
    }
  }
  if (p_drop > 0.f) {
    if (p_drop >= 1.f || !seed || ldc != N) return MSR3D_EINVAL;   // mask index = row * N + col
    flags |= 4;
  } else {
    flags &= ~4;
  }
  if (!ws_part && (flags & 5)) splits = 1;          // fused GELU / dropout need the complete sum
  int per = (slabs + splits - 1) / splits;
  if (per < 1) per = 1;                             // K == 0: C = bias + beta * C
  splits = slabs > 0 ? (slabs + per - 1) / per : 1;
  if (splits == 1) ws_part = nullptr;
  if (a_colsum) {
    // dW + db in one launch.  beta == 0: C (M x N, dense) is immediately followed by the M
    // column sums; beta == 1: both destinations already hold the values to add to (e.g. slices of
    // a zeroed flat gradient buffer) and may live anywhere.
    if (a_kc || ldc != N) return MSR3D_EINVAL;
    if (beta == 0.f) {
      if (a_colsum != C + (size_t)M * N) return MSR3D_EINVAL;
    } else if (beta != 1.f) {
      return MSR3D_EINVAL;
    }
    if (beta == 1.f) flags |= 2;                    // column sums accumulate as well
    if (splits > 1 && !ws_part) {                   // atomic meeting point
      if (beta == 0.f) {
        hipError_t e = hipMemsetAsync(C, 0, sizeof(float) * ((size_t)M * N + M), st);
        if (e != hipSuccess) return (int)e;
      }
    }
  } else if (splits > 1 && !ws_part && beta == 0.f) {
    if (ldc != N) return MSR3D_EINVAL;          // atomic split path zeroes a dense C
    hipError_t e = hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, st);
    if (e != hipSuccess) return (int)e;
  }
  GemmP g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.Cpre = C_pre; g.flags = flags; g.beta = beta;
  g.a_vec = vec_ok(A, lda); g.b_vec = vec_ok(B, ldb); g.slabs_per_split = per;
  g.a_colsum = a_colsum; g.ws_part = ws_part; g.ws_count = ws_count;
  g.p_drop = p_drop; g.seed = seed; g.salt = salt;
  g.gx = (N + BN - 1) / BN; g.gy = (M + BM - 1) / BM; g.gz = splits;
  *out = g; *rm_out = rm; *rn_out = rn;
  return 0;
}

static int gemm_f32_impl(int a_kc, int b_kc, int M, int N, int K, const float *A, int lda,
                         const float *B, int ldb, float *C, int ldc, const float *bias,
                         float *C_pre, int flags, float beta, float *a_colsum, void *workspace,
                         size_t workspace_bytes, float p_drop, const unsigned long long *seed,
                         unsigned salt, msr3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  GemmP g;
  int rm, rn;
  bool empty;
  // short-reduction forward linears: A-resident kernel (no split-K, no meeting point, no zero-fill)
  constexpr int ares_min_n = 512;
  // (wide outputs, or tall problems -- the unfrozen backbone's SharedMLP layers, up to 983 k rows -- whose
  // row strips alone fill the chip)
  if (a_kc && b_kc && !a_colsum && M > 0 && (N >= ares_min_n || (M >= 8192 && N >= 64)) && K >= 16 && K <= 256 &&
      K % 16 == 0 && A && B && C &&
      vec_ok(A, lda) && vec_ok(B, ldb)) {
    if (p_drop > 0.f) {
      if (p_drop >= 1.f || !seed || ldc != N) return MSR3D_EINVAL;
      flags |= 4;
    } else {
      flags &= ~4;
    }
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.bias = bias; g.Cpre = C_pre; g.flags = flags; g.beta = beta; g.a_vec = 1; g.b_vec = 1;
    g.slabs_per_split = 0; g.a_colsum = nullptr; g.ws_part = nullptr; g.ws_count = nullptr;
    g.p_drop = p_drop; g.seed = seed; g.salt = salt; g.gx = g.gy = g.gz = 1;
    const int strips = (M + 63) / 64;
    const bool wide = strips * ((N + 127) / 128) >= 384;      // enough workgroups with 128 columns each
    return (int)(wide ? launch_ares<2>(g, st) : launch_ares<1>(g, st));
  }
  const int rc = plan_gemm(a_kc, M, N, K, A, lda, B, ldb, C, ldc, bias, C_pre, flags, beta, a_colsum,
                           workspace, workspace_bytes, p_drop, seed, salt, true, st, &g, &rm, &rn, &empty);
  if (rc != 0 || empty) return rc;
  hipError_t e = hipErrorInvalidValue;
#define PICK(AK, BKc)                                                     \
  do {                                                                    \
    if (rm == 2 && rn == 2) e = launch_gemm<AK, BKc, 2, 2>(g, st);        \
    else if (rm == 4 && rn == 2) e = launch_gemm<AK, BKc, 4, 2>(g, st);   \
    else if (rm == 2 && rn == 4) e = launch_gemm<AK, BKc, 2, 4>(g, st);   \
    else e = launch_gemm<AK, BKc, 4, 4>(g, st);                           \
  } while (0)
  if (a_kc && b_kc) PICK(true, true);
  else if (a_kc && !b_kc) PICK(true, false);
  else if (!a_kc && !b_kc) PICK(false, false);
  else PICK(false, true);
#undef PICK
  return (int)e;
}

}  // namespace

extern "C" {

int msr3d_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float *A, int lda,
                   const float *B, int ldb, float *C, int ldc, const float *bias, float *C_pre,
                   int flags, float beta, float p_drop, const unsigned long long *seed,
                   unsigned salt, void *workspace, size_t workspace_bytes, msr3d_stream_t stream) {
  return gemm_f32_impl(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, bias, C_pre, flags & ~6, beta,
                       nullptr, workspace, workspace_bytes, p_drop, seed, salt, stream);
}

int msr3d_linear_wgrad_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                           float *dw_db, void *workspace, size_t workspace_bytes,
                           msr3d_stream_t stream) {
  if (!dw_db) return MSR3D_EINVAL;
  return gemm_f32_impl(0, 0, N_out, K_in, M_tokens, dy, N_out, x, K_in, dw_db, K_in, nullptr,
                       nullptr, 0, 0.f, dw_db + (size_t)N_out * K_in, workspace, workspace_bytes, 0.f,
                       nullptr, 0, stream);
}

int msr3d_linear_wgrad_acc_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                               float *dw, float *db, void *workspace, size_t workspace_bytes,
                               msr3d_stream_t stream) {
  if (!dw) return MSR3D_EINVAL;
  return gemm_f32_impl(0, 0, N_out, K_in, M_tokens, dy, N_out, x, K_in, dw, K_in, nullptr,
                       nullptr, 0, 1.f, db, workspace, workspace_bytes, 0.f, nullptr, 0, stream);
}

int msr3d_linear_bwd_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                         const float *w, float *dx, float dx_beta, float *dw, float *db,
                         void *workspace, size_t workspace_bytes, msr3d_stream_t stream) {
  if (!dx || !dw) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // the two problems share the workspace: halves, each with its own counters
  void *ws1 = nullptr, *ws2 = nullptr;
  size_t wsb = 0;
  if (workspace && workspace_bytes >= 2 * (sizeof(int) * MSR3D_GEMM_WS_COUNTERS + (1u << 20))) {
    wsb = (workspace_bytes / 2) & ~(size_t)255;
    ws1 = workspace;
    ws2 = reinterpret_cast<char *>(workspace) + wsb;
  }
  GemmP p1, p2;
  int rm, rn;
  bool e1, e2;
  // dx (M x K_in) = dy (M x N_out) W (N_out x K_in): reduction over N_out
  int rc = plan_gemm(1, M_tokens, K_in, N_out, dy, N_out, w, K_in, dx, K_in, nullptr, nullptr, 0,
                     dx_beta, nullptr, ws1, wsb, 0.f, nullptr, 0, false, st, &p1, &rm, &rn, &e1);
  if (rc != 0) return rc;
  // dW (N_out x K_in) += dy^T x, db += colsum(dy): reduction over tokens
  rc = plan_gemm(0, N_out, K_in, M_tokens, dy, N_out, x, K_in, dw, K_in, nullptr, nullptr, 0, 1.f, db,
                 ws2, wsb, 0.f, nullptr, 0, false, st, &p2, &rm, &rn, &e2);
  if (rc != 0) return rc;
  if (e1 || e2) return 0;                            // M_tokens, N_out or K_in is zero
  constexpr size_t lds = sizeof(float) * 2 * (tile_floats(64) + tile_floats(64));
  const int blocks = p1.gx * p1.gy * p1.gz + p2.gx * p2.gy * p2.gz;
  gemm_linear_bwd_kernel<<<blocks, 256, lds, st>>>(p1, p2);
  return (int)hipGetLastError();
}

int msr3d_gemm_multi_f32(int n, const msr3d_gemm_problem_t *pr, msr3d_stream_t stream) {
  if (n < 0 || n > MSR3D_GEMM_MULTI_MAX || (n > 0 && !pr)) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // The problems of one launch share the chip.  Panel problems (panel_gemm.hip) are cut into K runs
  // of equal LENGTH across the launch -- total (tiles x stages) over ~1.5 workgroups per CU, at least
  // 4 stages -- so that a problem with few tiles and a long reduction is split and one with many tiles
  // is not (each K run beyond the first costs the output's size in atomic traffic, and the chip
  // retires only ~1.1 TB/s of float atomics: tools/bench_multi.py, tools/prof_panel.py).  Tiled
  // problems (gemm_f32.hip's kernel) get their share of ~3 workgroups per CU.
  int live = 0;
  long long units = 0;
  for (int j = 0; j < n; ++j) live += (pr[j].M > 0 && pr[j].N > 0) ? 1 : 0;
  // Measured (tools/bench_multi.py, 960 tokens): a problem that has the chip to itself is 30-40 %
  // faster on the panel kernel (ffn2 27.9 -> 17.3 us, proj 15.3 -> 12.3 us); launches that mix a dx
  // with weight gradients are not (B4 55.5 vs 60.8 us, B2 24.3 vs 28.1 us) -- co-resident panel
  // workgroups gain nothing from each other -- and stay on the tiled kernel with a per-problem share
  // of the workgroup budget.
  const bool use_panel = live == 1;
  for (int j = 0; j < n; ++j) {
    if (pr[j].M <= 0 || pr[j].N <= 0) continue;
    if ((use_panel || pr[j].single_run) && msr3d::panel_eligible(pr[j])) {
      int tiles, stages;
      msr3d::panel_shape(pr[j], &tiles, &stages);
      units += (long long)tiles * stages;
    }
  }
  const int target = live > 1 ? (768 / live < 192 ? 192 : 768 / live) : 0;
  int run_stages = (int)((units + 383) / 384);
  if (run_stages < 4) run_stages = 4;
  GemmBatch gb;
  msr3d::PanelBatch pb;
  gb.n = pb.n = 0;
  int blocks = 0, pblocks = 0;
  for (int j = 0; j < n; ++j) {
    const msr3d_gemm_problem_t &q = pr[j];
    // atomic meeting point only: C (and the column-sum destination) hold the values to add to, or
    // zeros -- the caller's one zero-fill per step covers them; beta = 0 is allowed where no K-split
    // happens, and the planners zero-fill otherwise
    if (q.beta != 0.f && q.beta != 1.f) return MSR3D_EINVAL;
    if (q.colsum && (q.a_kc || q.beta != 1.f)) return MSR3D_EINVAL;
    if (q.M < 0 || q.N < 0 || q.K < 0 || q.lda <= 0 || q.ldb <= 0 || q.ldc <= 0) return MSR3D_EINVAL;
    if (q.M == 0 || q.N == 0) continue;
    if (!q.A || !q.B || !q.C) return MSR3D_EINVAL;
    if ((use_panel || q.single_run) && msr3d::panel_eligible(q)) {
      // long reductions and weight gradients: one K chunk per workgroup, operands staged once
      // (panel_gemm.hip)
      msr3d::PanelP pp;
      const int rc = msr3d::panel_plan(q, run_stages, &pp, st);
      if (rc != 0) return rc;
      pb.p[pb.n] = pp;
      pb.first[pb.n] = pblocks;
      pblocks += pp.gx * pp.gy * pp.gz;
      ++pb.n;
      continue;
    }
    GemmP g;
    int rm, rn;
    bool empty;
    const int rc = plan_gemm(q.a_kc, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.bias, nullptr, 0,
                             q.beta, q.colsum, nullptr, 0, 0.f, nullptr, 0, false, st, &g, &rm, &rn, &empty, target);
    if (rc != 0) return rc;
    if (empty) continue;
    gb.p[gb.n] = g;
    gb.kind[gb.n] = (q.a_kc ? 2 : 0) + (q.b_kc ? 1 : 0);
    gb.first[gb.n] = blocks;
    blocks += g.gx * g.gy * g.gz;
    ++gb.n;
  }
  if (pb.n > 0) {
    for (int j = pb.n; j <= MSR3D_GEMM_MULTI_MAX; ++j) pb.first[j] = pblocks;
    const int rc = msr3d::panel_launch(pb, pblocks, st);
    if (rc != 0) return rc;
  }
  if (gb.n == 0) return 0;
  for (int j = gb.n; j <= MSR3D_GEMM_MULTI_MAX; ++j) gb.first[j] = blocks;
  constexpr size_t lds = sizeof(float) * 2 * (tile_floats(64) + tile_floats(64));
  gemm_multi_kernel<<<blocks, 256, lds, st>>>(gb);
  return (int)hipGetLastError();
}

int msr3d_colsum_f32(int M, int N, const float *X, int ldx, float *out, int accumulate,
                     msr3d_stream_t stream) {
  if (M < 0 || N <= 0 || !out || (M > 0 && !X)) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (!(accumulate & 1)) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, st);
    if (e != hipSuccess) return (int)e;
  }
  if (M == 0) return 0;
  int chunks = (M + 127) / 128;
  if (chunks > 64) chunks = 64;
  if (accumulate & 2) chunks = 1;       // ordered: one workgroup owns a column's whole sum
  dim3 grid((N + 63) / 64, chunks);
  colsum_kernel<<<grid, 256, 0, st>>>(M, N, X, ldx, out);
  return (int)hipGetLastError();
}

int msr3d_gelu_bwd_f32(long long n, const float *dy, const float *pre, float *out, float p_drop,
                       const unsigned long long *seed, unsigned salt, msr3d_stream_t stream) {
  if (n < 0 || (n % 4) != 0 || p_drop >= 1.f || (p_drop > 0.f && !seed)) return MSR3D_EINVAL;
  if (n == 0) return 0;
  if (!dy || !pre || !out) return MSR3D_EINVAL;
  const long long n4 = n / 4;
  long long g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  gelu_bwd_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(
      n4, reinterpret_cast<const float4 *>(dy), reinterpret_cast<const float4 *>(pre),
      reinterpret_cast<float4 *>(out), p_drop, seed, salt);
  return (int)hipGetLastError();
}

}  // extern "C"
