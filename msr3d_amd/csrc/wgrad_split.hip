// wgrad_split.hip -- every weight gradient of a training step in ONE launch, fp32-accurate on the bf16
// matrix pipe (include/msr3d_hip.h: msr3d_wgrad_split).
//
//     dW[n][k] += sum_m dy[m][n] x[m][k]          db[n] += sum_m dy[m][n]          m = token rows
//
// The reduction runs over the TOKENS, i.e. across all scenes -- the one part of the backward that is not
// scene-local -- so the weight gradients are deferred to the end of the backward (the scene blocks leave
// dy and x of every linear layer behind as dense fp32 side outputs) and computed together: ~330 tiles of
// 128 x 128 in one grid instead of ten launches of 2-3 products whose K-splits met by atomics.  A
// workgroup owns a tile over the WHOLE reduction: no split-K, no atomics, bit-reproducible.
//
// Both operands are contracted over their ROW index, so a fragment (8 consecutive tokens of one column)
// is a strided read: each thread loads 8 tokens x 1 column (a wave: 64 consecutive columns = 256
// contiguous bytes per token), splits the eight values exactly into three bf16 terms (split_mma.h) and
// writes its 16 bytes of each plane straight into fragment order in LDS -- the transpose costs nothing.
// Register-prefetched three slabs ahead, LDS double-buffered over slabs of 32 tokens: one barrier per slab.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "../../include/msr3d_hip.h"
#include "colsum.h"
#include "split_mma.h"

// phase marks: empty here; tools/prof/wgrad_stamped.hip defines WG_STAMP and includes this file
#ifndef WG_STAMP
#define WG_STAMP(i)
#define WG_CLOCK() 0ull
#define WG_PUT(i, v)
#endif

namespace {

using namespace msr3d;
using WP = msr3d_wgrad_problem_t;

constexpr int TN = 128, TK = 128;                  // tile: 128 outputs x 128 inputs
// LDS fragment tiles: [3 planes][64 slots][16 B] + 64 B so that consecutive tiles start half a bank row apart,
// and slot(j, g) = perm(j) + 16 g with perm(j) = 4 (j % 4) + j / 4.  A loader thread holds 4 CONSECUTIVE
// columns (one 16-byte load per token): with the identity layout its 8-lane store groups would hit two
// 16-byte positions per bank row (4-way conflicts); transposing the 4 x 4 slot grid and offsetting
// alternate tiles makes the 8 stores of a group tile one whole bank row, and the multipliers'
// ds_read_b128 lane groups still cover 16 distinct positions.
constexpr int TILE_BYTES = 3 * 1024 + 64;
constexpr int OP_BYTES = (TN / 16) * TILE_BYTES;   // one operand of one slab: 25,088
constexpr int STAGE = 2 * OP_BYTES;                // dy^T tile + x tile
constexpr int LDS_BYTES = 2 * STAGE;               // 100,352: one workgroup per CU
__device__ __forceinline__ int frag_slot(int j, int g) { return (4 * (j & 3) + (j >> 2) + 16 * g) * 16; }

// 16 waves, two roles (round 4; round 3 ran 4 + 4 waves).  Splitting an operand element costs ~6 VALU instructions,
// so the work is divided: waves 8-15 LOAD -- waves 8-11 the dy tile, 12-15 the x tile; a thread takes 8 tokens x 2
// consecutive columns with eight 8-byte loads through a buffer descriptor (out-of-range tokens read as zero), splits
// each column's eight values exactly into three bf16 terms and writes 16 bytes per plane straight into fragment
// order, two slabs ahead -- and waves 0-7 MULTIPLY: a 64 x 32 piece each, 48 MFMAs per slab from 18 ds_read_b128,
// two multiplier waves per SIMD so that one's fragment reads fly under the other's MFMAs.  One barrier per slab.
// Round 3's stamps (4 multipliers of 64 x 64, 4 loaders of 8 x 4 units) per slab and wave: multiplier 1.1 k cycles
// of exposed fragment reads + 2.0 k of MFMAs, loader 3.2 k -- both twice the matrix pipe's 1.5 k.
// One 128 x 128 tile of one problem, whole reduction.  OVERWRITE: dW is set, not added to.
// FOLD (n_out, k_in <= 64: half the tile's columns would be padding, and so would half of every loader
// instruction's bytes): the tile carries TWO row chunks
// side by side -- columns 0..63 of both operands from the rows of chunk A, columns 64..127 from those of chunk B
// (fold_rows further down, fold_mb of them) -- so D's diagonal 64 x 64 blocks are the two chunks' products
// (written to pr.dW and pr.dW + n_out * k_in) and the off-diagonal ones are never multiplied.
// XPRO: the x operand is the PRE-normalisation activation of a BatchNorm + ReLU layer and the product wants its output:
// x <- max(gamma (x - mean) rstd + beta, 0) per column, applied by the loaders before the split (xpro = [gamma | beta |
// mean | rstd], k_in floats each) -- the normalised activation is never written (hipops._BNReLULinear).
constexpr int NMULT = 8, NLOAD = 8, NTHREADS = (NMULT + NLOAD) * 64;
// HALVES (msr3d_wgrad_split_halves): a tile's reduction is cut in two units of whole slab pairs so that the ~1.4
// tiles per CU of a step spread evenly (360 one-per-CU tiles = two rounds, the second 41 % full).  A unit takes a
// ticket when it STARTS: the first starter (role 0) parks its accumulators in the tile's workspace slot and raises a
// flag, the second (role 1) adds  first-half + second-half  in that fixed order -- whichever of them it is -- onto dW:
// bit-reproducible, no float atomics; the flag's owner is resident when anyone waits for it.
struct Half {
  int s0, s1;               // slab range [s0, s1), both even
  int role;                 // -1: whole reduction, 0: park the partial, 1: finish
  int second;               // this unit holds the reduction's second half
  float *ws;                // the tile's slot: 128 x 128 accumulator image + 128 column sums
  int *flag;                // raised by role 0 after its image is visible; cleared by role 1
};
constexpr int kHalfSlot = MSR3D_WGRAD_HALF_SLOT_FLOATS;      // floats per tile slot: image + column sums
static_assert(kHalfSlot == TN * TK + TN, "slot size");

template <bool OVERWRITE, bool FOLD = false, bool XPRO = false, bool HALVES = false>
__device__ __forceinline__ void wgrad_tile(const WP &pr, int ntile, int ktile, unsigned char *smem, int fold_rows = 0,
                                           int fold_mb = 0, const float *xpro = nullptr, Half hf = Half{0, 0, -1, 0, nullptr, nullptr}) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = ntile * TN, k0 = ktile * TK;
  const int M = pr.M;
  // slabs of 32 tokens, padded to an even count: a slab past M reads zeros (out-of-range buffer loads) and
  // adds nothing, and the loader's loop body stays straight-line (see there)
  const int s_lo = HALVES ? hf.s0 : 0;
  const int nslab = HALVES ? hf.s1 : ((M + 63) >> 6) << 1;
  WG_STAMP(0);

  if (wave >= NMULT) {
    // ------------------------------------------------------------------ loader
    const int lw = wave - NMULT;
    const bool isx = lw >= NLOAD / 2;              // which operand this wave feeds (wave-uniform)
    const float *src = isx ? pr.x : pr.dy;
    const int ld = isx ? pr.ldx : pr.ldy, ncol = isx ? pr.k_in : pr.n_out, c0 = isx ? k0 : n0;
    const int u = (lw & 3) * 64 + lane;            // unit: column pair P (of 64), token octet o (of 4)
    const int P = u & 63, o = u >> 6;
    const int col = FOLD ? 2 * (P & 31) : c0 + 2 * P;
    const int fold_off = FOLD ? (P >> 5) * fold_rows * ld * 4 : 0;     // (chunk B's rows)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(src), 0, (int)((size_t)(FOLD && fold_mb ? fold_rows + fold_mb : M) * ld * 4), 0x00020000);
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 7u) == 0 && (ld & 1) == 0;   // wave-uniform
    float cm[2];
    int cb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { cm[i] = col + i < ncol ? 1.f : 0.f; cb[i] = 4 * min(col + i, ncol - 1); }
    const bool edge = cm[1] == 0.f;
    float xg[2] = {0.f, 0.f}, xb[2] = {0.f, 0.f}, xm[2] = {0.f, 0.f}, xr[2] = {0.f, 0.f};
    if (XPRO && isx) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cc = min(col + i, ncol - 1);
        xg[i] = xpro[cc]; xb[i] = xpro[ncol + cc]; xm[i] = xpro[2 * ncol + cc]; xr[i] = xpro[3 * ncol + cc];
      }
    }
    struct Slab { float v[8][2]; };
    float cs[2] = {0.f, 0.f};
    // VEC is a compile-time property of the whole loop: a per-load `if (vec)` makes every load its own
    // basic block, and the compiler then waits vmcnt(0) at each join -- the fetch serialises (measured:
    // the loader waves at 3.7k cycles per slab, the multipliers waiting for them a third of their life)
    unsigned long long tw = 0;                     // (profiling build: cycles spent waiting at the barriers)
    (void)tw;
    auto run = [&](auto vec_tag) {
      constexpr bool VEC = decltype(vec_tag)::value;
      auto fetch = [&](Slab &s_, int s) {
        // the whole byte offset goes through the lane offset (the range check is defined on it): a token
        // past M starts exactly at the buffer's size, so every out-of-range load returns 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = ((32 * s + e) + 8 * o) * ld * 4 + fold_off;
          if constexpr (VEC) {
#ifdef WG_ABLATE_LOAD
            s_.v[e][0] = (float)row; s_.v[e][1] = 1.f;
#else
            const sm_f32x2 r = __builtin_bit_cast(sm_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, cb[0] + row, 0, 0));
            s_.v[e][0] = r.x; s_.v[e][1] = r.y;
#endif
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
              s_.v[e][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cb[i] + row, 0, 0));
          }
        }
      };
      auto stash = [&](Slab &s_, int buf) {
        unsigned char *T = smem + buf * STAGE + (isx ? OP_BYTES : 0) + (P >> 3) * TILE_BYTES;
        if (XPRO && isx) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              s_.v[e][i] = fmaxf(__builtin_fmaf(xg[i], (s_.v[e][i] - xm[i]) * xr[i], xb[i]), 0.f);
        }
        if (edge) {                                // columns past the matrix (an 8-byte load reads on into the row)
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i) s_.v[e][i] *= cm[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float c8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { c8[e] = s_.v[e][i]; cs[i] += c8[e]; }
          uint4 pl[3];
#ifdef WG_ABLATE_SPLIT
          pl[0] = make_uint4(__float_as_uint(c8[0]), __float_as_uint(c8[1]), __float_as_uint(c8[2]), __float_as_uint(c8[3]));
          pl[1] = make_uint4(__float_as_uint(c8[4]), __float_as_uint(c8[5]), __float_as_uint(c8[6]), __float_as_uint(c8[7]));
          pl[2] = pl[0];
#else
          sm_split8(c8, pl);
#endif
          // (lanes P = 0..7 of a store group hold the even columns of one 16-column tile: slots 0, 8, 1, 9, 2, 10, 3, 11
          // -- eight distinct 16-byte positions of one bank row)
          const int slot = frag_slot(2 * (P & 7) + i, o);
#pragma unroll
          for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(T + k * 1024 + slot) = pl[k];
        }
      };
      // Two slabs in flight per thread, fetched unconditionally (past the end: zeros).
      Slab v0, v1;
      fetch(v0, s_lo);
      fetch(v1, s_lo + 1);
      for (int s = s_lo; s < nslab; s += 2) {
        stash(v0, 0);
        fetch(v0, s + 2);
        { const unsigned long long c0_ = WG_CLOCK(); __syncthreads(); tw += WG_CLOCK() - c0_; }   // slab s is in stage 0
        stash(v1, 1);
        fetch(v1, s + 3);
        { const unsigned long long c0_ = WG_CLOCK(); __syncthreads(); tw += WG_CLOCK() - c0_; }
      }
    };
    if (vec) run(std::true_type{});
    else run(std::false_type{});
    WG_PUT(1, tw);
    __syncthreads();                               // the last slab has been multiplied: LDS is free
    float *red = reinterpret_cast<float *>(smem);  // [4 octets][128]: bias gradient = column sums of dy
    if (!isx) {
#pragma unroll
      for (int i = 0; i < 2; ++i) red[o * TN + 2 * P + i] = cs[i];
    }
    __syncthreads();
    if (pr.db && ktile == 0 && lw < 2) {
      const int c = lw * 64 + lane;
      float t = (red[c] + red[TN + c]) + (red[2 * TN + c] + red[3 * TN + c]);
      if (HALVES && hf.role == 0) {
        hf.ws[TN * TK + c] = t;
      } else {
        if (HALVES && hf.role == 1) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          const float o = hf.ws[TN * TK + c];
          t = hf.second ? o + t : t + o;           // first half + second half
        }
        if (n0 + c < pr.n_out) pr.db[n0 + c] += t;
      }
    }
    if (HALVES) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                             // (the hand-over barrier: see the multipliers' tail)
    }
  } else {
    // ------------------------------------------------------------------ multiplier: wave (wr, wc) owns
    // output rows 64 wr .. (4 tiles) x input columns 32 wc .. (2 tiles)
    const int wr = wave >> 2, wc = wave & 3;
    const int slot = frag_slot(lane & 15, lane >> 4);
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long tw = 0;
    (void)tw;
    for (int s = s_lo; s < nslab; ++s) {
      { const unsigned long long c0_ = WG_CLOCK(); __syncthreads(); tw += WG_CLOCK() - c0_; }
#ifdef WG_ABLATE_MMA
      continue;
#endif
      if (FOLD && wr != (wc >> 1)) continue;        // (an off-diagonal block: chunk A's dy against chunk B's x)
      const unsigned char *A = smem + (s & 1) * STAGE + slot, *Bs = A + OP_BYTES;
      // all 18 fragment reads first, in the order the products consume them (planes A2 B0 | A0 B2 | A1 B1), then the
      // MFMAs behind counted waits: left to itself hipcc re-used one fragment register serially -- read, wait
      // lgkmcnt(0), two MFMAs, read ... -- and the first third of every slab ran at LDS latency
      bf16x8 fa[4][3], fb[2][3];
      constexpr int ORD_A[3] = {2, 0, 1}, ORD_B[3] = {0, 2, 1};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          fa[a][ORD_A[q]] = *reinterpret_cast<const bf16x8 *>(A + (4 * wr + a) * TILE_BYTES + ORD_A[q] * 1024);
#pragma unroll
        for (int b = 0; b < 2; ++b)
          fb[b][ORD_B[q]] = *reinterpret_cast<const bf16x8 *>(Bs + (2 * wc + b) * TILE_BYTES + ORD_B[q] * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#define MSR3D_TERM(PA, PB)                                                                       \
      _Pragma("unroll") for (int a = 0; a < 4; ++a)                                              \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][PA], fb[b][PB], acc[a][b], 0, 0, 0);
#if MSR3D_TRAIN_PLANES == 3
      MSR3D_TERM(2, 0)
      MSR3D_TERM(0, 2)
      MSR3D_TERM(1, 1)
      MSR3D_TERM(1, 0)
      MSR3D_TERM(0, 1)
#endif
      MSR3D_TERM(0, 0)
#undef MSR3D_TERM
    }
    WG_STAMP(5);
    WG_PUT(1, tw);
    if (HALVES && hf.role == 1) {
      // the other half's image must be there: its owner took its ticket before this unit did, so it is running (or
      // done) -- one lane polls, everyone else waits at the barrier below
      if (tid == 0)
        while (__hip_atomic_load(hf.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    __syncthreads();
    if (HALVES && hf.role == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // D[n][k]: lane (j = k column, g): rows n = 4 g + r.  dW holds the value to add to; this workgroup is
    // the tile's only writer.
    const int j = lane & 15, g = lane >> 4;
    if (HALVES && hf.role == 0) {
      // park the accumulators: image[(wave, a, b)][lane] float4 -- plain 16-byte stores (4-byte device-coherent ones
      // are partial-line writes at the memory side: the first version of this hand-over cost 40 us a launch); the
      // release fence below writes the L2 lines back before the flag goes up
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          *reinterpret_cast<f32x4 *>(hf.ws + (((wave * 4 + a) * 2 + b) * 64 + lane) * 4) = acc[a][b];
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (HALVES && hf.role == 1) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(hf.ws + (((wave * 4 + a) * 2 + b) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[a][b][r] = hf.second ? o[r] + acc[a][b][r] : acc[a][b][r] + o[r];     // first half + second half
          }
          const int kk = FOLD ? 32 * (wc & 1) + 16 * b + j : k0 + 32 * wc + 16 * b + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = FOLD ? 16 * a + 4 * g + r : n0 + 64 * wr + 16 * a + 4 * g + r;
            if (n < pr.n_out && kk < pr.k_in && (!FOLD || (wr == (wc >> 1) && (wr == 0 || fold_mb > 0)))) {
              float *d = pr.dW + (FOLD ? (size_t)wr * pr.n_out * pr.k_in : 0) + (size_t)n * pr.ldw + kk;
              *d = OVERWRITE ? acc[a][b][r] : *d + acc[a][b][r];
            }
          }
        }
    }
    if (HALVES) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                             // every wave's image stores (loaders: column sums) have left the CU
      if (tid == 0) {
        if (hf.role == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_store(hf.flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (hf.role == 1) {
          __hip_atomic_store(hf.flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // clean for the next launch
        }
      }
    }
  }
  WG_STAMP(6);
}

// =====================================================================================================
// PIPE (round 6): the same 128 x 128 tile, whole reduction, by EIGHT waves (two per SIMD, 256-register budget) that
// each load, split, stash AND multiply -- no loader waves -- with the fragment reads of the NEXT half-slab issued
// under the CURRENT half-slab's MFMAs inside every wave.
//
// What this replaces: 8 loader + 8 multiplier waves met at one barrier per slab, so all eight multipliers read their 18
// fragments (147 KB of LDS traffic a slab: 1.1 k cycles of the CU's 128 B / clk) and THEN issued their 48 MFMAs (1.5 k
// cycles of the SIMD's pipe), one after the other in every wave at the same time: 3.1 k cycles a slab, 59 us a tile
// against 23 us of MFMA issue (profiles/r05_v2_wgrad_forms.txt; 16 waves = 128 registers a wave left no room for a
// second fragment set).  Here a slab is two UNITS of 24 MFMAs (output row tiles 0-1 | 2-3 of the wave's 64 x 32 piece):
//
//     slab s, stage s & 1:      read A(2,3)(s) | MFMA unit (s, 0) | lgkmcnt(0) + barrier |
//                               read A(0,1)(s+1), B(s+1) from the other stage | MFMA unit (s, 1) |
//                               split + stash slab s+2 into THIS stage | fetch slab s+4
//
// one barrier per slab as before: it says both that every wave is done reading stage s & 1 (free for slab s+2) and
// that slab s+1's planes, stashed a slab earlier, are complete.  Registers: two B fragment sets + two A half sets (96),
// 32 accumulators, two slabs of fetched rows (32).  Loads / split / stash are the loader waves' code (same units: wave w
// feeds operand w >> 2, token octet w & 3; 8 tokens x 2 columns per lane) and the products and their order are the
// multiplier waves': the SAME bits as wgrad_tile (tests/test_scene_blocks_gpu.py compares the two forms).
// =====================================================================================================
constexpr int NTHREADS_PIPE = 512;
#ifndef WGP_ABLATE
#define WGP_ABLATE 0        // tools/ablate_wgrad.sh builds 1..5: the pipe tile without its MFMAs / split / loads / fragment reads / stash
#endif

template <bool HALVES = false>
__device__ __forceinline__ void wgrad_tile_pipe(const WP &pr, int ntile, int ktile, unsigned char *smem,
                                                Half hf = Half{0, 0, -1, 0, nullptr, nullptr}) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = ntile * TN, k0 = ktile * TK;
  const int M = pr.M;
  const int s_lo = HALVES ? hf.s0 : 0;
  const int nslab = HALVES ? hf.s1 : ((M + 63) >> 6) << 1;          // even, padded (slabs past M read zeros)
  WG_STAMP(0);

  // ---- loader role: operand isx, token octet o, column pair P ----
  const bool isx = wave >= 4;
  const float *src = isx ? pr.x : pr.dy;
  const int ld = isx ? pr.ldx : pr.ldy, ncol = isx ? pr.k_in : pr.n_out, c0 = isx ? k0 : n0;
  const int P = lane, o = wave & 3;
  const int col = c0 + 2 * P;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)((size_t)M * ld * 4),
                                                                       0x00020000);
  const bool vec = (reinterpret_cast<uintptr_t>(src) & 7u) == 0 && (ld & 1) == 0;   // wave-uniform
  float cm[2];
  int cb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { cm[i] = col + i < ncol ? 1.f : 0.f; cb[i] = 4 * min(col + i, ncol - 1); }
  float cs[2] = {0.f, 0.f};
  struct Slab { float v[8][2]; };
  unsigned char *const stash_base = smem + (isx ? OP_BYTES : 0) + (P >> 3) * TILE_BYTES;
  int slot_w[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) slot_w[i] = frag_slot(2 * (P & 7) + i, o);

  // ---- multiplier role: wave (wr, wc) owns output rows 64 wr .. x input columns 32 wc .. ----
  const int wr = (wave >> 2) & 1, wc = wave & 3;
  const int slot_r = frag_slot(lane & 15, lane >> 4);
  f32x4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto run = [&](auto vec_tag) {
    constexpr bool VEC = decltype(vec_tag)::value;
    auto fetch = [&](Slab &s_, int s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = ((32 * s + e) + 8 * o) * ld * 4;
        if constexpr (VEC) {
#if WGP_ABLATE == 3
          s_.v[e][0] = (float)row; s_.v[e][1] = 1.f;
#else
          const sm_f32x2 r = __builtin_bit_cast(sm_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, cb[0] + row, 0, 0));
          s_.v[e][0] = r.x; s_.v[e][1] = r.y;
#endif
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            s_.v[e][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cb[i] + row, 0, 0));
        }
      }
    };
    // split: a fetched slab's 8 tokens x 2 columns -> three bf16 planes per column, kept in registers (+ the column sums);
    // put: their six 16-byte stores, fragment order.  `live`: false for a slab past the reduction (nothing may enter db)
    auto split = [&](Slab &s_, uint4 (&pl)[2][3], bool live) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // (columns past the matrix -- an 8-byte load reads on into the row -- times 0: a multiply, not a branch per lane)
        float c8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { c8[e] = s_.v[e][i] * cm[i]; cs[i] += live ? c8[e] : 0.f; }
#if WGP_ABLATE == 2
        pl[i][0] = make_uint4(__float_as_uint(c8[0]), __float_as_uint(c8[1]), __float_as_uint(c8[2]), __float_as_uint(c8[3]));
        pl[i][1] = make_uint4(__float_as_uint(c8[4]), __float_as_uint(c8[5]), __float_as_uint(c8[6]), __float_as_uint(c8[7]));
        pl[i][2] = pl[i][0];
#else
        sm_split8(c8, pl[i]);
#endif
      }
    };
    auto put = [&](const uint4 (&pl)[2][3], int buf) {
      unsigned char *T = stash_base + buf * STAGE;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#if WGP_ABLATE == 5
        asm volatile("" ::"v"(pl[i][0].x), "v"(pl[i][1].y), "v"(pl[i][2].z));
#else
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint4 *>(T + k * 1024 + slot_w[i]) = pl[i][k];
#endif
      }
    };
    // fragments: A = dy^T tile (output rows), B = x tile (input columns)
    auto read_a = [&](bf16x8 (&fa)[2][3], int buf, int half) {
#if WGP_ABLATE == 4
      return;
#endif
      const unsigned char *A = smem + buf * STAGE + slot_r;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < kPlanes; ++q)
          fa[a][q] = *reinterpret_cast<const bf16x8 *>(A + (4 * wr + 2 * half + a) * TILE_BYTES + q * 1024);
    };
    auto read_b = [&](bf16x8 (&fb)[2][3], int buf) {
#if WGP_ABLATE == 4
      return;
#endif
      const unsigned char *Bs = smem + buf * STAGE + OP_BYTES + slot_r;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < kPlanes; ++q)
          fb[b][q] = *reinterpret_cast<const bf16x8 *>(Bs + (2 * wc + b) * TILE_BYTES + q * 1024);
    };
    auto unit = [&](const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3], int half) {
#if WGP_ABLATE == 1
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("" ::"v"(fa[a][q]), "v"(fb[a][q]));
      return;
#endif
#define MSR3D_TERM(PA, PB)                                                                       \
      _Pragma("unroll") for (int a = 0; a < 2; ++a)                                              \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
          acc[2 * half + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][PA], fb[b][PB], acc[2 * half + a][b], 0, 0, 0);
#if MSR3D_TRAIN_PLANES == 3
      MSR3D_TERM(2, 0)
      MSR3D_TERM(0, 2)
      MSR3D_TERM(1, 1)
      MSR3D_TERM(1, 0)
      MSR3D_TERM(0, 1)
#endif
      MSR3D_TERM(0, 0)
#undef MSR3D_TERM
    };

    Slab g0, g1;                                   // fetched rows of the slabs two and three ahead of the multiply
    bf16x8 fa0[2][3], fa1[2][3], fb0[2][3], fb1[2][3];
    uint4 pl[2][3];
#if WGP_ABLATE == 4
    for (int a = 0; a < 2; ++a)
      for (int q = 0; q < 3; ++q) {
        const bf16x8 c = {(short)(lane + a), (short)q, 1, 2, 3, 4, 5, 6};
        fa0[a][q] = fa1[a][q] = fb0[a][q] = fb1[a][q] = c;
      }
#endif
    // ---- prologue: slabs s_lo, s_lo + 1 stashed, s_lo + 2, s_lo + 3 in flight, slab s_lo's first fragments read ----
    fetch(g0, s_lo);
    fetch(g1, s_lo + 1);
    split(g0, pl, true);
    put(pl, 0);
    fetch(g0, s_lo + 2);
    split(g1, pl, true);
    put(pl, 1);
    fetch(g1, s_lo + 3);
    __syncthreads();
    read_a(fa0, 0, 0);
    read_b(fb0, 0);
    // One slab; `cur` = its stage, fbc / fbn = the B fragments of this / the next slab, g = the fetched rows of slab s + 2.
    // No branch inside (a junk slab stashed or read past the end is never multiplied).
    auto slab_body = [&](int s, int cur, bf16x8 (&fbc)[2][3], bf16x8 (&fbn)[2][3], Slab &g) {
      read_a(fa1, cur, 1);                         // this slab's second half, under unit 0
      __builtin_amdgcn_sched_barrier(0);
      unit(fa0, fbc, 0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                             // stage `cur` read by everyone; slab s + 1 complete in the other stage
      read_a(fa0, cur ^ 1, 0);                     // the next slab's first fragments ..
      read_b(fbn, cur ^ 1);
      split(g, pl, s + 2 < nslab);                 // .. and slab s + 2's planes (registers), under unit 1
      unit(fa1, fbc, 1);
#if defined(WGP_GROUPS)
      // (measured and NOT kept: forcing one MFMA : five VALU with sched_group_barrier so that the split rides between the
      //  wave's own products -- 120 us a launch against 106 with the compiler's order: profiles/r06_wgrad_ablation.txt)
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      put(pl, cur);                                // slab s + 2 into the stage this slab has left
      fetch(g, s + 4);                             // (past the end: zeros, or rows nobody multiplies)
    };
    for (int s = s_lo; s < nslab; s += 2) {
      slab_body(s, 0, fb0, fb1, g0);
      slab_body(s + 1, 1, fb1, fb0, g1);
    }
  };
  if (vec) run(std::true_type{});
  else run(std::false_type{});
  WG_STAMP(5);

  __syncthreads();                                 // the last slab has been multiplied: LDS is free
  float *red = reinterpret_cast<float *>(smem);    // [4 octets][128]: bias gradient = column sums of dy
  if (!isx) {
#pragma unroll
    for (int i = 0; i < 2; ++i) red[o * TN + 2 * P + i] = cs[i];
  }
  if (HALVES && hf.role == 1) {
    // the other half's image must be there: its owner took its ticket before this unit did, so it is running (or done)
    if (tid == 0)
      while (__hip_atomic_load(hf.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  if (HALVES && hf.role == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (pr.db && ktile == 0 && tid < TN) {
    const int c = tid;
    float t = (red[c] + red[TN + c]) + (red[2 * TN + c] + red[3 * TN + c]);
    if (HALVES && hf.role == 0) {
      hf.ws[TN * TK + c] = t;
    } else {
      if (HALVES && hf.role == 1) {
        const float o2 = hf.ws[TN * TK + c];
        t = hf.second ? o2 + t : t + o2;           // first half + second half
      }
      if (n0 + c < pr.n_out) pr.db[n0 + c] += t;
    }
  }
  // D[n][k]: lane (j = k column, g): rows n = 4 g + r.  dW holds the value to add to; this workgroup is the tile's
  // only writer.
  const int j = lane & 15, g = lane >> 4;
  if (HALVES && hf.role == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        *reinterpret_cast<f32x4 *>(hf.ws + (((wave * 4 + a) * 2 + b) * 64 + lane) * 4) = acc[a][b];
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (HALVES && hf.role == 1) {
          const f32x4 o2 = *reinterpret_cast<const f32x4 *>(hf.ws + (((wave * 4 + a) * 2 + b) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[a][b][r] = hf.second ? o2[r] + acc[a][b][r] : acc[a][b][r] + o2[r];     // first half + second half
        }
      }
    // dW += acc: all 32 reads of the lane in flight together, then the stores (a read-add-write per element waits for
    // each read in turn: 32 dependent round trips at the end of every tile)
    float old[4][2][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int kk = k0 + 32 * wc + 16 * b + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 64 * wr + 16 * a + 4 * g + r;
          const bool ok = n < pr.n_out && kk < pr.k_in;
          old[a][b][r] = ok ? pr.dW[(size_t)n * pr.ldw + kk] : 0.f;
        }
      }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int kk = k0 + 32 * wc + 16 * b + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 64 * wr + 16 * a + 4 * g + r;
          if (n < pr.n_out && kk < pr.k_in) pr.dW[(size_t)n * pr.ldw + kk] = old[a][b][r] + acc[a][b][r];
        }
      }
  }
  if (HALVES) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                               // every wave's image stores (and column sums) have left the CU
    if (tid == 0 && hf.flag) {                     // (flag == nullptr: the stream launch -- nobody waits inside the launch)
      if (hf.role == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(hf.flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (hf.role == 1) {
        __hip_atomic_store(hf.flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // clean for the next launch
      }
    }
  }
  WG_STAMP(6);
}

// workgroups past `tiles` (msr3d_wgrad_split_colsum): one column-sum job each -- the LayerNorm parameter gradients'
// second stage rides in the weight-gradient launch's second round instead of a launch of its own
template <bool PIPE>
__global__ __launch_bounds__(PIPE ? NTHREADS_PIPE : NTHREADS) void wgrad_split_kernel(int nprob, const WP *__restrict__ probs,
                                                          const int *__restrict__ prefix, int tiles,
                                                          const msr3d_colsum_job_t *__restrict__ cjobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x >= tiles) {
    colsum_job(cjobs[blockIdx.x - tiles], reinterpret_cast<float *>(smem));
    return;
  }
  int lo = 0, hi = nprob - 1;
  const int t = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
  }
  const WP pr = probs[lo];
  // A problem owns a multiple of 8 workgroups; workgroup b runs on XCD b % 8 (observed dispatch order:
  // speed only), so XCD x takes the CONTIGUOUS run of tiles [x' nb/8, (x'+1) nb/8), x' = (x - xcd_rot) % 8: tiles that
  // share a dy block (same output tile, consecutive input tiles) read it through one L2.  xcd_rot (the host's, per
  // problem) spreads the problems' PADDING workgroups over the XCDs: unrotated, the step's 230 real tiles fell
  // 33 / 33 / 31 / 31 / 28 / 28 / 26 / 20 on the eight XCDs of 32 CUs -- one tile too many on two of them, i.e. a
  // second round of one tile: the launch lasted TWO tile times (118 us) for 0.9 rounds of work.
  const int lb = t - prefix[lo], nb = prefix[lo + 1] - prefix[lo];
  const int local = ((lb - pr.xcd_rot) & 7) * (nb >> 3) + (lb >> 3);
  const int nkt = (pr.k_in + TK - 1) / TK;
  if (local >= ((pr.n_out + TN - 1) / TN) * nkt) return;
  if constexpr (PIPE) wgrad_tile_pipe<false>(pr, local / nkt, local % nkt, smem);
  else wgrad_tile<false>(pr, local / nkt, local % nkt, smem);
}

// The same tiles as wgrad_split_kernel, each as TWO units (gridDim.x = 2 x padded tiles): workgroups [0, T) hold the
// first halves, [T, 2 T) the second halves -- T is a multiple of 8, so both halves of a tile run on one XCD
// (workgroup b -> XCD b % 8: speed only) and the parked image travels through that XCD's L2.
// sync: [T] tickets | [T] flags (all zero between launches); ws: T slots of kHalfSlot floats.
__global__ __launch_bounds__(NTHREADS) void wgrad_halves_kernel(int nprob, const WP *__restrict__ probs,
                                                                const int *__restrict__ prefix, float *__restrict__ ws,
                                                                int *__restrict__ sync) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int role_s;
  const int T = gridDim.x >> 1;
  const int second = blockIdx.x >= T ? 1 : 0;
  const int t = blockIdx.x - second * T;
  int lo = 0, hi = nprob - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
  }
  const WP pr = probs[lo];
  const int lb = t - prefix[lo], nb = prefix[lo + 1] - prefix[lo];
  const int local = ((lb - pr.xcd_rot) & 7) * (nb >> 3) + (lb >> 3);
  const int nkt = (pr.k_in + TK - 1) / TK;
  if (local >= ((pr.n_out + TN - 1) / TN) * nkt) return;
  const int nslab = ((pr.M + 63) >> 6) << 1;
  const int cut = ((nslab >> 1) + 1) & ~1;            // whole slab pairs: 30 slabs -> 16 + 14
  Half hf;
  hf.second = second;
  hf.s0 = second ? cut : 0;
  hf.s1 = second ? nslab : cut;
  hf.ws = ws + (size_t)t * kHalfSlot;
  hf.flag = sync + T + t;
  if (cut >= nslab) {                                  // a reduction of one slab pair is not cut
    if (second) return;
    hf.s1 = nslab;
    hf.role = -1;
  } else {
    if (threadIdx.x == 0) {
      const int k = __hip_atomic_fetch_add(sync + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == 1) __hip_atomic_store(sync + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (both tickets taken)
      role_s = k;
    }
    __syncthreads();
    hf.role = role_s;
  }
  wgrad_tile<false, false, false, true>(pr, local / nkt, local % nkt, smem, 0, 0, nullptr, hf);
}

// MIXED (round 5): the step's ~330 real tiles on 256 CUs are one full round and a second one that is 29 % full -- the
// launch lasts two tile times.  Cutting EVERY tile in two (wgrad_halves_kernel) does not change that ratio and pays the
// hand-over everywhere (measured slower).  Here only the tiles of the partial round are cut: workgroups [0, W) take whole
// tiles, [W, W + H) and [W + H, W + 2 H) the first and second halves of tiles W .. W + H - 1 (H = tiles - W, a multiple
// of 8: both halves on one XCD), so the second round lasts half a tile time.  Same ticket protocol as above (the unit
// that STARTS first parks, the second adds first + second in that order: bit-reproducible); workgroups past W + 2 H
// run the column-sum jobs.  sync: [H] tickets | [H] flags, zero between launches; ws: H slots.
template <bool PIPE>
__global__ __launch_bounds__(PIPE ? NTHREADS_PIPE : NTHREADS) void wgrad_mixed_kernel(int nprob, const WP *__restrict__ probs,
                                                               const int *__restrict__ prefix, int tiles, int W,
                                                               float *__restrict__ ws, int *__restrict__ sync,
                                                               const msr3d_colsum_job_t *__restrict__ cjobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int role_s;
  const int H = tiles - W, b = blockIdx.x;
  if (b >= W + 2 * H) {
    colsum_job(cjobs[b - (W + 2 * H)], reinterpret_cast<float *>(smem));
    return;
  }
  const bool halved = b >= W;
  const int second = b >= W + H ? 1 : 0;
  const int t = halved ? b - second * H : b;
  int lo = 0, hi = nprob - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= t) lo = mid; else hi = mid - 1;
  }
  const WP pr = probs[lo];
  const int lb = t - prefix[lo], nb = prefix[lo + 1] - prefix[lo];
  const int local = ((lb - pr.xcd_rot) & 7) * (nb >> 3) + (lb >> 3);
  const int nkt = (pr.k_in + TK - 1) / TK;
  if (local >= ((pr.n_out + TN - 1) / TN) * nkt) return;
  if (!halved) {
    if constexpr (PIPE) wgrad_tile_pipe<false>(pr, local / nkt, local % nkt, smem);
    else wgrad_tile<false>(pr, local / nkt, local % nkt, smem);
    return;
  }
  const int nslab = ((pr.M + 63) >> 6) << 1;
  const int cut = ((nslab >> 1) + 1) & ~1;            // whole slab pairs: 30 slabs -> 16 + 14
  const int h = t - W;
  Half hf;
  hf.second = second;
  hf.s0 = second ? cut : 0;
  hf.s1 = second ? nslab : cut;
  hf.ws = ws + (size_t)h * kHalfSlot;
  hf.flag = sync + H + h;
  if (cut >= nslab) {                                  // a reduction of one slab pair is not cut
    if (second) return;
    hf.s1 = nslab;
    hf.role = -1;
  } else {
    if (threadIdx.x == 0) {
      const int k = __hip_atomic_fetch_add(sync + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == 1) __hip_atomic_store(sync + h, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (both tickets taken)
      role_s = k;
    }
    __syncthreads();
    hf.role = role_s;
  }
  if constexpr (PIPE) wgrad_tile_pipe<true>(pr, local / nkt, local % nkt, smem, hf);
  else wgrad_tile<false, false, false, true>(pr, local / nkt, local % nkt, smem, 0, 0, nullptr, hf);
}

// STREAM (round 6): the launch's tiles as ONE sequence of slab pairs dealt evenly to a persistent grid (one workgroup per
// CU).  ~330 tiles of 30 slabs on 256 CUs are 1.3 rounds: whole tiles last two tile times, the mixed launch 1.5 -- here
// every workgroup gets 1/256 of the slab pairs (+ a fixed charge per piece), i.e. up to three PIECES: the tail of one
// tile, whole tiles, the head of another.  A tile is cut at most once (the host deals at least one tile's worth to every
// workgroup).  The two parts of a cut tile do NOT meet inside the launch: an in-kernel hand-over between CUs costs an
// agent-scope release + acquire -- 23-27 k cycles, profiles/r06_last_arriver_probe.txt; every tile halved with tickets is
// the slowest form of all -- while a kernel boundary is ~2 us.  So the part with the EARLIER slabs parks its accumulator
// image (+ column sums) in the tile's workspace slot with plain stores, the part with the later slabs adds onto dW like
// a whole tile, and wgrad_fixup_kernel, launched behind, adds the parked images: dW = (dW + later) + earlier, a fixed
// order: bit-reproducible, no atomics, no flags, no spinning.  The column-sum jobs ride as pieces of the least loaded
// workgroups.  pieces / wg_first / slot_piece: msr3d_amd/scene_blocks.py.
using PC = msr3d_wgrad_piece_t;
__global__ __launch_bounds__(NTHREADS_PIPE) void wgrad_stream_kernel(const WP *__restrict__ probs, const PC *__restrict__ pieces,
                                                                     const int *__restrict__ wg_first, float *__restrict__ ws,
                                                                     const msr3d_colsum_job_t *__restrict__ cjobs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int p0 = wg_first[blockIdx.x], p1 = wg_first[blockIdx.x + 1];
  for (int i = p0; i < p1; ++i) {
    const PC pc = pieces[i];
    if (i > p0) __syncthreads();                   // the previous piece's tail is done with the LDS
    if (pc.kind == 1) {
      colsum_job(cjobs[pc.prob], reinterpret_cast<float *>(smem));
      continue;
    }
    const WP pr = probs[pc.prob];
    if (pc.slot < 0) {
      wgrad_tile_pipe<false>(pr, pc.ntile, pc.ktile, smem);
      continue;
    }
    Half hf;
    hf.second = pc.second;
    hf.s0 = pc.s0;
    hf.s1 = pc.s1;
    hf.ws = ws + (size_t)pc.slot * kHalfSlot;
    hf.flag = nullptr;                             // no in-kernel hand-over: the parked image waits for the fixup launch
    hf.role = pc.second ? -1 : 0;                  // later slabs: onto dW; earlier slabs: parked
    wgrad_tile_pipe<true>(pr, pc.ntile, pc.ktile, smem, hf);
  }
}

// dW += parked image, db += parked column sums, one workgroup per cut tile; the image is the multiplier waves' register
// layout (wgrad_tile_pipe's epilogue): image[(wave, a, b)][lane] float4 = rows 64 wr + 16 a + 4 g + r, column 32 wc + 16 b + j
__global__ __launch_bounds__(NTHREADS_PIPE) void wgrad_fixup_kernel(const WP *__restrict__ probs, const PC *__restrict__ pieces,
                                                                    const int *__restrict__ slot_piece,
                                                                    const float *__restrict__ ws) {
  const PC pc = pieces[slot_piece[blockIdx.x]];
  const WP pr = probs[pc.prob];
  const float *img = ws + (size_t)pc.slot * kHalfSlot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) & 1, wc = wave & 3, j = lane & 15, g = lane >> 4;
  const int n0 = pc.ntile * TN, k0 = pc.ktile * TK;
  f32x4 v[4][2];
  float old[4][2][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      v[a][b] = *reinterpret_cast<const f32x4 *>(img + (((wave * 4 + a) * 2 + b) * 64 + lane) * 4);
      const int kk = k0 + 32 * wc + 16 * b + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * wr + 16 * a + 4 * g + r;
        old[a][b][r] = (n < pr.n_out && kk < pr.k_in) ? pr.dW[(size_t)n * pr.ldw + kk] : 0.f;
      }
    }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kk = k0 + 32 * wc + 16 * b + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * wr + 16 * a + 4 * g + r;
        if (n < pr.n_out && kk < pr.k_in) pr.dW[(size_t)n * pr.ldw + kk] = old[a][b][r] + v[a][b][r];
      }
    }
  if (pr.db && pc.ktile == 0 && tid < TN && n0 + tid < pr.n_out) pr.db[n0 + tid] += img[TN * TK + tid];
}

// TALL problems (an unfrozen backbone's SharedMLP layers: dW (<= 256 x <= 256) over 10^5 .. 10^6 rows): the rows are
// cut into chunks, workgroup (chunk, tile) reduces its chunk into ws[chunk] (n_out, k_in) -- plain stores, every
// workgroup the only writer of its slab -- and wgrad_rows_reduce_kernel adds the slabs in chunk order.
// n_out, k_in <= 64: workgroup b takes the chunk PAIR (2 b, 2 b + 1) in one folded tile (wgrad_tile<.., FOLD>)
__global__ __launch_bounds__(NTHREADS) void wgrad_rows_fold_kernel(WP base, int chunk_rows, float *ws, const float *xpro) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WP pr = base;
  const long long r0 = (long long)(2 * blockIdx.x) * chunk_rows;
  pr.dy = base.dy + r0 * base.ldy;
  pr.x = base.x + r0 * base.ldx;
  const long long left = (long long)base.M - r0;
  pr.M = (int)min((long long)chunk_rows, left);
  const int mb = (int)max(0ll, min((long long)chunk_rows, left - chunk_rows));
  pr.dW = ws + (size_t)(2 * blockIdx.x) * base.n_out * base.k_in;
  pr.ldw = base.k_in;
  pr.db = nullptr;
  if (xpro) wgrad_tile<true, true, true>(pr, 0, 0, smem, chunk_rows, mb, xpro);
  else wgrad_tile<true, true>(pr, 0, 0, smem, chunk_rows, mb);
}

__global__ __launch_bounds__(NTHREADS) void wgrad_rows_kernel(WP base, int chunk_rows, int tiles, float *ws,
                                                            const float *xpro) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (chunk, tile) of this workgroup.  With several tiles per chunk the tiles of ONE chunk read the same rows of one
  // operand: they are placed on the same XCD (workgroup b runs on XCD b % 8) and dispatched together -- b and b + 8 --
  // so that the second read is served by that XCD's L2 instead of HBM (491 k rows, 256 x 128: 281 -> 254 us).
  int chunk, tile;
  const int nchunks = gridDim.x / tiles;
  if (tiles > 1 && (nchunks & 7) == 0) {
    const int grp = blockIdx.x / (8 * tiles), r = blockIdx.x - grp * 8 * tiles;
    chunk = grp * 8 + (r & 7);
    tile = r >> 3;
  } else {
    chunk = blockIdx.x / tiles;
    tile = blockIdx.x - chunk * tiles;
  }
  const int nkt = (base.k_in + TK - 1) / TK;
  WP pr = base;
  const long long r0 = (long long)chunk * chunk_rows;
  pr.dy = base.dy + r0 * base.ldy;
  pr.x = base.x + r0 * base.ldx;
  pr.M = (int)min((long long)chunk_rows, (long long)base.M - r0);
  pr.dW = ws + (size_t)chunk * base.n_out * base.k_in;
  pr.ldw = base.k_in;
  pr.db = nullptr;
  if (xpro) wgrad_tile<true, false, true>(pr, tile / nkt, tile % nkt, smem, 0, 0, xpro);
  else wgrad_tile<true>(pr, tile / nkt, tile % nkt, smem);
}

__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(int n_out, int k_in, int chunks, const float *__restrict__ ws,
                                                                float *__restrict__ dW, int ldw, int accumulate) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_out * k_in) return;
  const size_t slab = (size_t)n_out * k_in;
  float t = 0.f;
  for (int c0 = 0; c0 < chunks; c0 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = c0 + u < chunks ? ws[(size_t)(c0 + u) * slab + e] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += v[u];
  }
  float *d = dW + (size_t)(e / k_in) * ldw + e % k_in;
  *d = accumulate ? *d + t : t;
}

// MSR3D_WGRAD_PIPE=0: the 16-wave loader / multiplier form of rounds 4-5 (wgrad_tile); default: wgrad_tile_pipe
int g_pipe_form = -1;             // -1: not chosen yet (environment at first use)
bool pipe_form() {
  if (g_pipe_form < 0) {
    const char *v = getenv("MSR3D_WGRAD_PIPE");
    g_pipe_form = (v && v[0] == '0') ? 0 : 1;
  }
  return g_pipe_form != 0;
}

int launch_split(int n, const WP *problems, const int *tile_prefix, int total_tiles, int n_jobs,
                 const msr3d_colsum_job_t *jobs, hipStream_t stream) {
  if (pipe_form()) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_split_kernel<true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return (int)attr;
    wgrad_split_kernel<true><<<total_tiles + n_jobs, NTHREADS_PIPE, LDS_BYTES, stream>>>(n, problems, tile_prefix, total_tiles, jobs);
    return (int)hipGetLastError();
  }
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_split_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return (int)attr;
  wgrad_split_kernel<false><<<total_tiles + n_jobs, NTHREADS, LDS_BYTES, stream>>>(n, problems, tile_prefix, total_tiles, jobs);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int msr3d_wgrad_form(int form) {
  if (form == 0 || form == 1) g_pipe_form = form;
  else if (form != -1) return MSR3D_EINVAL;
  return pipe_form() ? 1 : 0;
}

extern "C" int msr3d_wgrad_split(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix,
                                 int total_tiles, msr3d_stream_t stream) {
  if (n < 0 || total_tiles < 0) return MSR3D_EINVAL;
  if (n == 0 || total_tiles == 0) return 0;
  if (!problems || !tile_prefix) return MSR3D_EINVAL;
  return launch_split(n, problems, tile_prefix, total_tiles, 0, nullptr, (hipStream_t)stream);
}

extern "C" int msr3d_wgrad_split_colsum(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix,
                                        int total_tiles, int n_jobs, const msr3d_colsum_job_t *jobs,
                                        msr3d_stream_t stream) {
  if (n < 0 || total_tiles < 0 || n_jobs < 0) return MSR3D_EINVAL;
  if ((n == 0 || total_tiles == 0) && n_jobs == 0) return 0;
  if ((total_tiles > 0 && (!problems || !tile_prefix)) || (n_jobs > 0 && !jobs)) return MSR3D_EINVAL;
  return launch_split(n, problems, tile_prefix, total_tiles, n_jobs, jobs, (hipStream_t)stream);
}

extern "C" int msr3d_wgrad_split_halves(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix,
                                        int total_tiles, float *workspace, long long workspace_floats, int *sync,
                                        msr3d_stream_t stream) {
  if (n < 0 || total_tiles < 0) return MSR3D_EINVAL;
  if (n == 0 || total_tiles == 0) return 0;
  if (!problems || !tile_prefix || !workspace || !sync || (total_tiles & 7)) return MSR3D_EINVAL;
  if (workspace_floats < (long long)total_tiles * MSR3D_WGRAD_HALF_SLOT_FLOATS) return MSR3D_EINVAL;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_halves_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return (int)attr;
  wgrad_halves_kernel<<<2 * total_tiles, NTHREADS, LDS_BYTES, (hipStream_t)stream>>>(n, problems, tile_prefix, workspace,
                                                                                    sync);
  return (int)hipGetLastError();
}

extern "C" int msr3d_wgrad_split_mixed(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix,
                                       int total_tiles, int whole_tiles, int n_jobs, const msr3d_colsum_job_t *jobs,
                                       float *workspace, long long workspace_floats, int *sync, msr3d_stream_t stream) {
  if (n < 0 || total_tiles < 0 || n_jobs < 0 || whole_tiles < 0 || whole_tiles > total_tiles) return MSR3D_EINVAL;
  if ((n == 0 || total_tiles == 0) && n_jobs == 0) return 0;
  const int H = total_tiles - whole_tiles;
  if ((total_tiles > 0 && (!problems || !tile_prefix)) || (n_jobs > 0 && !jobs) || (H & 7) || (whole_tiles & 7)) return MSR3D_EINVAL;
  if (H > 0 && (!workspace || !sync || workspace_floats < (long long)H * MSR3D_WGRAD_HALF_SLOT_FLOATS)) return MSR3D_EINVAL;
  if (pipe_form()) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mixed_kernel<true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) return (int)attr;
    wgrad_mixed_kernel<true><<<whole_tiles + 2 * H + n_jobs, NTHREADS_PIPE, LDS_BYTES, (hipStream_t)stream>>>(
        n, problems, tile_prefix, total_tiles, whole_tiles, workspace, sync, jobs);
    return (int)hipGetLastError();
  }
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_mixed_kernel<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return (int)attr;
  wgrad_mixed_kernel<false><<<whole_tiles + 2 * H + n_jobs, NTHREADS, LDS_BYTES, (hipStream_t)stream>>>(
      n, problems, tile_prefix, total_tiles, whole_tiles, workspace, sync, jobs);
  return (int)hipGetLastError();
}

extern "C" int msr3d_wgrad_stream(int n, const msr3d_wgrad_problem_t *problems, int n_pieces,
                                  const msr3d_wgrad_piece_t *pieces, const int *wg_first, int n_wgs, int n_slots,
                                  const int *slot_piece, float *workspace, long long workspace_floats,
                                  const msr3d_colsum_job_t *jobs, msr3d_stream_t stream) {
  if (n < 0 || n_pieces < 0 || n_wgs < 0 || n_slots < 0) return MSR3D_EINVAL;
  if (n_pieces == 0 || n_wgs == 0) return 0;
  if (!pieces || !wg_first || (n > 0 && !problems)) return MSR3D_EINVAL;
  if (n_slots > 0 && (!workspace || !slot_piece || workspace_floats < (long long)n_slots * MSR3D_WGRAD_HALF_SLOT_FLOATS))
    return MSR3D_EINVAL;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_stream_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return (int)attr;
  wgrad_stream_kernel<<<n_wgs, NTHREADS_PIPE, LDS_BYTES, (hipStream_t)stream>>>(problems, pieces, wg_first, workspace, jobs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || n_slots == 0) return (int)e;
  wgrad_fixup_kernel<<<n_slots, NTHREADS_PIPE, 0, (hipStream_t)stream>>>(problems, pieces, slot_piece, workspace);
  return (int)hipGetLastError();
}

extern "C" int msr3d_wgrad_rows_split(int M, int n_out, int k_in, const float *dy, int ldy, const float *x, int ldx,
                                      float *dW, int ldw, int accumulate, float *workspace, long long workspace_floats,
                                      const float *x_bn, msr3d_stream_t stream) {
  if (M <= 0 || n_out <= 0 || k_in <= 0 || !dy || !x || !dW || !workspace || ldy < n_out || ldx < k_in || ldw < k_in)
    return MSR3D_EINVAL;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_rows_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return (int)attr;
  const int tiles = ((n_out + TN - 1) / TN) * ((k_in + TK - 1) / TK);
  const long long slab = (long long)n_out * k_in;
  const bool fold = n_out <= 64 && k_in <= 64;               // two chunks per workgroup (see wgrad_tile)
  // chunks: one round of the 256 CUs, at least 512 rows each, as many as the workspace holds
  long long chunks = (fold ? 2 : 1) * MSR3D_WGRAD_ROWS_CHUNKS / tiles;
  chunks = chunks < 1 ? 1 : chunks;
  chunks = chunks < (M + 511) / 512 ? chunks : (M + 511) / 512;
  chunks = chunks < workspace_floats / slab ? chunks : workspace_floats / slab;
  if (chunks < 1) return MSR3D_EINVAL;
  int chunk_rows = (int)((M + chunks - 1) / chunks);
  chunk_rows = (chunk_rows + 63) / 64 * 64;                  // whole slab pairs
  chunks = (M + chunk_rows - 1) / chunk_rows;
  WP base;
  base.dy = dy; base.ldy = ldy; base.n_out = n_out;
  base.x = x; base.ldx = ldx; base.k_in = k_in;
  base.M = M;
  base.dW = nullptr; base.ldw = k_in; base.db = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (fold) {
    static const hipError_t fattr = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_rows_fold_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (fattr != hipSuccess) return (int)fattr;
    wgrad_rows_fold_kernel<<<(unsigned)((chunks + 1) / 2), NTHREADS, LDS_BYTES, st>>>(base, chunk_rows, workspace, x_bn);
  } else {
    wgrad_rows_kernel<<<(unsigned)(chunks * tiles), NTHREADS, LDS_BYTES, st>>>(base, chunk_rows, tiles, workspace, x_bn);
  }
  wgrad_rows_reduce_kernel<<<(unsigned)((slab + 255) / 256), 256, 0, st>>>(n_out, k_in, (int)chunks, workspace, dW, ldw,
                                                                          accumulate);
  return (int)hipGetLastError();
}
