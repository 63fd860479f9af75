// rows_gemm_split.hip -- C (M, N) = A (M, K) op(B)^T for TALL operands (hundreds of thousands of rows, K and N
// at most a few hundred), fp32-accurate on the bf16 matrix pipe (include/msr3d_hip.h: msr3d_rows_gemm_split).
//
// The SharedMLP layers of an UNFROZEN PointNet++ backbone in training mode
// (/root/reference/model/pointnet2/pytorch_utils.py:9-60 as token GEMMs, hipops.py::_mlp_rows): 983 k / 491 k
// rows x (3..131 -> 64..256) channels at 16 scenes x 60 objects.  On the fp32 matrix instructions
// (gemm_nt_ares_kernel) they ran at ~58 TFLOP/s, 250 us a layer; the rows themselves are only 2 x 250 MB of
// traffic (~100 us).  Here every operand is split exactly into three bf16 terms and a product is the six bf16
// MFMA products above 2^-24 of it (split_mma.h, the arithmetic of the frozen encoder's sa_split.hip), which
// makes the product HBM-bound:
//
//   * op(B) -- the layer's weight, up to 144 x 160 or 64 x 256 -- is split by the workgroup itself on its way into
//     LDS, in MFMA fragment order ([k/32][n/16][3 planes][64 lanes][8]), once per workgroup: no pack launch.  Wider
//     outputs: column groups over blockIdx.y; wider reductions (the last level's 15 k x 256..768 layers): 128-wide
//     super-slabs of op(B) re-filled per row block, the accumulators carried across them;
//   * a wave owns 32 rows per pass: its lanes load the rows' fp32 values straight in fragment shape (row i,
//     k = 32 s + 8 g .. + 7: two 16-byte buffer loads; the four g-lanes of a row cover 128 contiguous bytes), split
//     them in registers, and multiply against every column tile of B from LDS (three ds_read_b128 per 12 MFMAs).
//     The NEXT unit's values are loaded into the registers of each slab right after it has been split, so the loads
//     fly under this unit's MFMAs at no register cost;
//   * 8 waves = 256 rows per pass and workgroup, one workgroup per CU walking the row blocks grid-stride.
// Two things ride on it for the BatchNorm that surrounds every product of the backbone (bn_train.hip):
//   * col_stats: the column sums and sums of squares of each 256-row block of C from the accumulators (DPP row
//     reduction, the eight waves added in order) -- the normalisation's first-stage statistics without a pass over C;
//   * a_bn: A <- max(gamma (A - mean) rstd + beta, 0) per column as the values are consumed -- the PREVIOUS layer's
//     normalisation + ReLU, whose output is then never written (hipops._BNReLULinear).
// A lane of D holds four consecutive columns of one row: 16-byte stores.
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"
#include "split_mma.h"

namespace {

using namespace msr3d;

struct RowsGemm {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb; int b_trans;
  float *C; int ldc;
  int nblk;                       // row blocks of 256
  float *stats;                   // optional: [nblk][2][N] column sums / sums of squares of each block's rows of C
  const float *apro;              // optional: [gamma | beta | mean | rstd], K floats each: A <- relu(batch_norm(A)) on load
};

// sum over the 16 lanes of a DPP row (the 16 rows of an MFMA tile a lane group holds), result in every lane;
// a fixed tree: quad, half row, row
__device__ __forceinline__ float row16_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));   // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

template <int NT, int KS>
__global__ __launch_bounds__(512, 1) void rows_gemm_split_kernel(const RowsGemm p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [KS][NT][3][64][16 B], then the statistics'
  float *sred = reinterpret_cast<float *>(smem + KS * NT * 3 * 1024);    // [2 parities][8 waves][2][NT * 16]
  float *ppro = sred + 2 * 8 * 2 * NT * 16;                             // [4][kpad]: the operand's BatchNorm parameters
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (NT * 16);

  // op(B) -> LDS planes: unit = (slab s, tile t, lane l): row n0 + 16 t + (l & 15), k = kb + 32 s + 8 (l >> 4) .. + 7
  auto load_b = [&](int kb) {
    for (int u = tid; u < KS * NT * 64; u += 512) {
      const int l = u & 63, t = (u >> 6) % NT, s = (u >> 6) / NT;
      const int n = n0 + 16 * t + (l & 15), k = kb + 32 * s + 8 * (l >> 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // (a B narrower than the product -- ldb below the width op(B) is read at -- ends there: the operand's zero
        //  padding needs no padded copy of the weight)
        const bool ok = n < p.N && k + e < p.K && (p.b_trans ? n : k + e) < p.ldb;
        v[e] = ok ? (p.b_trans ? p.B[(size_t)(k + e) * p.ldb + n] : p.B[(size_t)n * p.ldb + k + e]) : 0.f;
      }
      uint4 pl[3];
      sm_split8(v, pl);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4 *>(smem + (((s * NT + t) * 3 + q) * 64 + l) * 16) = pl[q];
    }
  };
  // K wider than one LDS image of op(B) (KS slabs): the image is re-filled per super-slab inside the row-block loop
  // (the short, wide products of the last set-abstraction level: a workgroup has one or two row blocks)
  const int nsup = (p.K + KS * 32 - 1) / (KS * 32);
  const int kpad = nsup * KS * 32;
  if (p.apro) {                   // (columns past K: all four zero -> relu(0) = 0, like the zero padding they replace)
    for (int e = tid; e < 4 * kpad; e += 512) {
      const int a = e / kpad, k = e - a * kpad;
      ppro[e] = k < p.K ? p.apro[a * p.K + k] : 0.f;
    }
  }
  if (nsup == 1) load_b(0);
  if (nsup == 1 || p.apro) __syncthreads();

  const unsigned char *bl = smem + lane * 16;
  // Units of work = (row block, super-slab), walked in order.  The fp32 values of unit u + 1 are loaded INTO the
  // registers of unit u slab by slab, each right after its slab has been split (the raw values are dead then): the
  // next unit's loads fly under this unit's MFMAs and epilogue at no register cost.
  // (buffer loads: one descriptor in SGPRs, one 32-bit lane offset per row tile, the slab's offset as an immediate --
  //  64-bit pointers per load cost 2 x 2 x KS address registers, which is what spilled)
  const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.A), 0, (int)((size_t)p.M * p.lda * 4), 0x00020000);
  auto row_off = [&](int r0, int rt, int kb) {
    return (unsigned)(min(r0 + 16 * rt + i, p.M - 1) * p.lda + kb + 8 * g) * 4u;
  };
  // (a row's tail past K is zeroed where the values are CONSUMED: a select here would wait for the load)
  auto load_slab = [&](f32x4 (&raw)[2][KS][2], int s, unsigned off0, unsigned off1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const unsigned off = rt ? off1 : off0;
      raw[rt][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, off, 128 * s, 0));
      raw[rt][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, off, 128 * s + 16, 0));
    }
  };
  f32x4 raw[2][KS][2];
  if (blockIdx.x < p.nblk && blockIdx.x * 256 + wave * 32 < p.M) {
    const int r0 = blockIdx.x * 256 + wave * 32;
    const unsigned o0 = row_off(r0, 0, 0), o1 = row_off(r0, 1, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) load_slab(raw, s, o0, o1);
  }
  int parity = 0;
  for (int blk = blockIdx.x; blk < p.nblk; blk += gridDim.x, parity ^= 1) {
    const int r0 = blk * 256 + wave * 32;
    const bool live = r0 < p.M;     // (a wave past M -- last block only -- still meets the barriers)
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sup = 0; sup < nsup; ++sup) {
      const int kb = sup * KS * 32;
      if (nsup > 1) {
        __syncthreads();            // everyone is done with the previous image
        load_b(kb);
        __syncthreads();
      }
      // the unit after this one
      const bool last_sup = sup + 1 == nsup;
      const int nblk2 = last_sup ? blk + (int)gridDim.x : blk, nkb = last_sup ? 0 : kb + KS * 32;
      const int nr0 = nblk2 * 256 + wave * 32;
      const bool next_live = nblk2 < p.nblk && nr0 < p.M;
      if (!live) continue;          // (then no later unit of this wave is live either: blocks ascend)
      const unsigned no0 = row_off(nr0, 0, nkb), no1 = row_off(nr0, 1, nkb);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 fa[2][3];
        const float in0 = kb + 32 * s + 8 * g < p.K ? 1.f : 0.f, in1 = kb + 32 * s + 8 * g + 4 < p.K ? 1.f : 0.f;
        const bool tail = kb + 32 * s + 32 > p.K;     // (wave-uniform: only the slab that straddles K pays the select)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          float v[8] = {raw[rt][s][0][0], raw[rt][s][0][1], raw[rt][s][0][2], raw[rt][s][0][3],
                        raw[rt][s][1][0], raw[rt][s][1][1], raw[rt][s][1][2], raw[rt][s][1][3]};
          if (p.apro) {            // A <- max(gamma (A - mean) rstd + beta, 0): the layer's normalisation, never stored
            const float *pp = ppro + kb + 32 * s + 8 * g;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 ga = *reinterpret_cast<const f32x4 *>(pp + 4 * h), be = *reinterpret_cast<const f32x4 *>(pp + kpad + 4 * h),
                          mu = *reinterpret_cast<const f32x4 *>(pp + 2 * kpad + 4 * h),
                          rs = *reinterpret_cast<const f32x4 *>(pp + 3 * kpad + 4 * h);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[4 * h + e] = fmaxf(__builtin_fmaf(ga[e], (v[4 * h + e] - mu[e]) * rs[e], be[e]), 0.f);
            }
          }
          if (tail) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (e < 4 ? in0 : in1) != 0.f ? v[e] : 0.f;
          }
          uint4 pl[3];
          sm_split8(v, pl);
#pragma unroll
          for (int q = 0; q < 3; ++q) fa[rt][q] = __builtin_bit_cast(bf16x8, pl[q]);
        }
        __builtin_amdgcn_sched_barrier(0);          // (the re-load of raw[s] stays below its last use ...
        if (next_live) load_slab(raw, s, no0, no1);
        __builtin_amdgcn_sched_barrier(0);          //  ... and later slabs' splits stay below this slab's MFMAs)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          bf16x8 fb[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bf16x8 *>(bl + ((s * NT + t) * 3 + q) * 1024);
#define MSR3D_TERM(PB, PA)                                                                            \
          _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                            \
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[PB], fa[rt][PA], acc[rt][t], 0, 0, 0);
          MSR3D_TERM(2, 0)
          MSR3D_TERM(0, 2)
          MSR3D_TERM(1, 1)
          MSR3D_TERM(1, 0)
          MSR3D_TERM(0, 1)
          MSR3D_TERM(0, 0)
#undef MSR3D_TERM
        }
      }
    }
    if (p.stats) {
      // column sums / sums of squares of this block's rows (rows past M: nothing), wave by wave, then the eight
      // waves in order: the first stage of the BatchNorm statistics that follow the product (bn_train.hip)
      const float m0 = r0 + i < p.M ? 1.f : 0.f, m1 = r0 + 16 + i < p.M ? 1.f : 0.f;
      float *mine = sred + ((parity * 8 + wave) * 2) * (NT * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc[0][t][e] * m0, b = acc[1][t][e] * m1;
          const float s1 = row16_sum(a + b), s2 = row16_sum(__builtin_fmaf(a, a, b * b));
          if (i == 0) {
            mine[16 * t + 4 * g + e] = s1;
            mine[NT * 16 + 16 * t + 4 * g + e] = s2;
          }
        }
      }
      __syncthreads();
      for (int c = tid; c < 2 * NT * 16; c += 512) {
        const int which = c / (NT * 16), col = c - which * (NT * 16);
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += sred[((parity * 8 + w) * 2 + which) * (NT * 16) + col];
        if (n0 + col < p.N) p.stats[((size_t)blk * 2 + which) * p.N + n0 + col] = t;
      }
    }
    if (!live) continue;
    // lane (i, g) holds columns 16 t + 4 g .. + 3 of row i
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int r = r0 + 16 * rt + i;
      if (r >= p.M) continue;
      float *crow = p.C + (size_t)r * p.ldc;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = n0 + 16 * t + 4 * g;
        if (c + 3 < p.N) {
          *reinterpret_cast<f32x4 *>(crow + c) = acc[rt][t];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < p.N) crow[c + e] = acc[rt][t][e];
        }
      }
    }
  }
}

template <int NT, int KS>
int launch(const RowsGemm &p, int ny, hipStream_t st) {
  constexpr int lds0 = KS * NT * 3 * 1024 + 2 * 8 * 2 * NT * 16 * 4;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&rows_gemm_split_kernel<NT, KS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     lds0 + 16 * 1024 < 160 * 1024 ? lds0 + 16 * 1024 : 160 * 1024);
  if (attr != hipSuccess) return (int)attr;
  const int lds = lds0 + (p.apro ? 16 * ((p.K + KS * 32 - 1) / (KS * 32)) * KS * 32 : 0);
  if (lds > 160 * 1024) return MSR3D_EINVAL;
  const int gx = min(p.nblk, max(1, 256 / ny));
  rows_gemm_split_kernel<NT, KS><<<dim3(gx, ny), 512, lds, st>>>(p);
  return (int)hipGetLastError();
}

template <int NT>
int pick_ks(const RowsGemm &p, int ny, int ks, hipStream_t st) {
  switch (ks) {
    case 1: return launch<NT, 1>(p, ny, st);
    case 2: return launch<NT, 2>(p, ny, st);
    case 3: return launch<NT, 3>(p, ny, st);
    case 4: return launch<NT, 4>(p, ny, st);
    case 5: return launch<NT, 5>(p, ny, st);
    default: return launch<NT, 4>(p, ny, st);      // wider: super-slabs of 128
  }
}

}  // namespace

extern "C" int msr3d_rows_gemm_split(int M, int N, int K, const float *A, int lda, const float *B, int ldb, int b_trans,
                                     float *C, int ldc, float *col_stats, const float *a_bn, msr3d_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || K > MSR3D_ROWS_GEMM_MAX_K || N > MSR3D_ROWS_GEMM_MAX_N) return MSR3D_EINVAL;
  if (!A || !B || !C || (K & 3) || (lda & 3) || (ldc & 3) || lda < K || ldc < N) return MSR3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(A) & 15u) || (reinterpret_cast<uintptr_t>(C) & 15u)) return MSR3D_EINVAL;
  if ((long long)M * lda * 4 >= (1ll << 31)) return MSR3D_EINVAL;       // (32-bit byte offsets into A)
  RowsGemm p{M, N, K, A, lda, B, ldb, b_trans, C, ldc, (M + 255) / 256, col_stats, a_bn};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ks = (K + 31) / 32;
  // column tiles per workgroup: all of them up to 144 columns (9 tiles), else halves of <= 128
  const int tiles = (N + 15) / 16;
  // tall with a 161..256-wide reduction (d t of a 256-channel layer): the whole of op(B) for 64 columns stays in LDS
  // and the rows are read once per 64-column group -- cheaper than re-filling LDS twice per row block
  if (ks > 5 && ks <= 8 && M > 65536) return launch<4, 8>(p, (tiles + 3) / 4, st);
  if (tiles <= 4) return pick_ks<4>(p, 1, ks, st);
  if (tiles <= 8) return pick_ks<8>(p, 1, ks, st);
  if (tiles == 9 && ks <= 4) return pick_ks<9>(p, 1, ks, st);
  return pick_ks<8>(p, (tiles + 7) / 8, ks, st);
}
