// panel_gemm.h -- internal interface between gemm_f32.hip (msr3d_gemm_multi_f32) and panel_gemm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/msr3d_hip.h"

namespace msr3d {

struct PanelP {
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float *C; int ldc;
  const float *bias;
  float *colsum;
  float beta;
  int kind;            // a_kc * 2 + b_kc
  int gx, gy, gz;      // column tiles (64), row tiles (64), K runs
  int spw;             // K stages (of 128) per run
};

struct PanelBatch {
  PanelP p[MSR3D_GEMM_MULTI_MAX];
  int first[MSR3D_GEMM_MULTI_MAX + 1];
  int n;
};

bool panel_eligible(const msr3d_gemm_problem_t &q);
void panel_shape(const msr3d_gemm_problem_t &q, int *tiles, int *stages);
int panel_plan(const msr3d_gemm_problem_t &q, int stages_per_run, PanelP *out, hipStream_t st);
int panel_launch(const PanelBatch &pb, int blocks, hipStream_t st);

}  // namespace msr3d
