/*
 * msr3d_hip.h -- C ABI of libmsr3d_hip.so, the MI355X (gfx950) implementation of
 * MSR3D's point-cloud hot path.
 *
 * This is the drop-in boundary: plain device pointers and sizes, a HIP stream,
 * an int status.  No torch types, no hidden allocation, no host sync, never
 * exit().  Every entry point replaces one `*_kernel_wrapper` the reference's
 * pybind layer calls (paths relative to
 * /root/reference/modules/third_party/pointnet2/_ext_src/); the binding a
 * maintainer adds on the reference side is in INTEGRATION.md.
 *
 * Conventions
 *   - all tensors are dense, row-major, f32 or i32, resident on the current device;
 *   - outputs are caller-allocated and FULLY written by the call (the reference
 *     relies on torch::zeros for ball_query no-hit rows and for the *_grad
 *     accumulators; here the callee establishes that state itself);
 *   - `stream` is a hipStream_t (NULL = default stream); launches are asynchronous;
 *   - return 0 on success, MSR3D_EINVAL for bad arguments, otherwise the
 *     hipError_t of the failed launch (the reference prints and exit(-1)s,
 *     include/cuda_utils.h:30-39).
 */
#ifndef MSR3D_HIP_H
#define MSR3D_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSR3D_ABI_VERSION 29
#define MSR3D_EINVAL (-22)

typedef void *msr3d_stream_t; /* hipStream_t */

int msr3d_abi_version(void);
/* Floating-point contract of the squared distance a*a + b*b + c*c in the index ops this library was built
 * with (csrc/pn2_device.h): 0 = fma(c,c,fma(a,a,b*b)), the default; 1 = no fma; 2 = fma(c,c,fma(b,b,a*a));
 * 3 = fma(a,a,fma(b,b,c*c)).  Reference: sampling_gpu.cu:99-104, ball_query_gpu.cu:32-35. */
int msr3d_sqdist_contract(void);
/* Static string for a status returned by any entry point. */
const char *msr3d_status_string(int status);

/* ---------------------------------------------------------------------------
 * B1: the nine ops of `pointnet2._ext` (src/bindings.cpp:6-19)
 * ------------------------------------------------------------------------- */

/* furthest_point_sampling_kernel_wrapper (src/sampling_gpu.cu:175-229; host
 * src/sampling.cpp:66-87).  xyz (b,n,3) f32 -> idx (b,m) i32.  idx[.,0] = 0.
 * Bit-exact with the reference's block reduction, including its tie-break
 * (which depends on the REFERENCE's block size opt_n_threads(n)), the
 * |p|^2 <= 1e-3 skip and the all-skipped -> 0 case.  The reference's `temp`
 * (b,n) scratch is not needed: distances live in registers.
 * new_xyz (b,m,3) is optional (NULL to skip): the gathered centroids, i.e. what
 * gather_points(xyz^T, idx)^T returns, produced in the same launch. */
int msr3d_furthest_point_sampling(int b, int n, int m, const float *xyz, int *idx,
                                  float *new_xyz, msr3d_stream_t stream);

/* gather_points_kernel_wrapper (src/sampling_gpu.cu:22-30).
 * points (b,c,n), idx (b,m) -> out (b,c,m). */
int msr3d_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                        float *out, msr3d_stream_t stream);

/* gather_points_grad_kernel_wrapper (src/sampling_gpu.cu:49-57).
 * grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n), zeroed then accumulated. */
int msr3d_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                             float *grad_points, msr3d_stream_t stream);

/* query_ball_point_kernel_wrapper (src/ball_query_gpu.cu:46-54).
 * new_xyz (b,m,3) centres, xyz (b,n,3) -> idx (b,m,nsample): the first nsample
 * points in index order with d^2 < radius^2 (f32, strict); remaining slots hold
 * the first hit; rows without a hit are zero. */
int msr3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                     const float *xyz, int *idx, msr3d_stream_t stream);

/* group_points_kernel_wrapper (src/group_points_gpu.cu:30-39).
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
int msr3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, msr3d_stream_t stream);

/* group_points_grad_kernel_wrapper (src/group_points_gpu.cu:66-75).
 * grad_out (b,c,npoints,nsample) -> grad_points (b,c,n); every element is written.  Unlike the
 * reference's atomicAdd scatter the three *_grad entries sum each destination's sources in
 * ascending source order (bit-reproducible) whenever 2*n_dst + n_src + 1 <= 36864. */
int msr3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points, msr3d_stream_t stream);

/* three_nn_kernel_wrapper (src/interpolate_gpu.cu:61-68).
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 (SQUARED), idx (b,n,3) i32. */
int msr3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx, msr3d_stream_t stream);

/* three_interpolate_kernel_wrapper (src/interpolate_gpu.cu:103-111).
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n). */
int msr3d_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                            const float *weight, float *out, msr3d_stream_t stream);

/* three_interpolate_grad_kernel_wrapper (src/interpolate_gpu.cu:145-154).
 * grad_out (b,c,n) -> grad_points (b,c,m), zeroed then accumulated. */
int msr3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                 const int *idx, const float *weight, float *grad_points,
                                 msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Fused set-abstraction levels (frozen / eval-mode backbone).  They replace, per level,
 * the launch sequence of _PointnetSAModuleBase.forward
 * (/root/reference/modules/third_party/pointnet2/pointnet2_modules.py:34-75) and
 * QueryAndGroup.forward (pointnet2_utils.py:314-373): FPS -> gather -> ball_query ->
 * group(xyz) -> recentre -> group(feat) -> cat -> 3 x [conv1x1, BN(eval), ReLU] -> max.
 * ------------------------------------------------------------------------- */

/* FPS of two consecutive levels in one launch: level 1 picks m1 of the n points of
 * pts (b, n, point_stride) (xyz = the first 3 floats of each point row; stride 6 reads the
 * dataset's xyz+rgb rows in place), level 2 picks m2 (<= m1 <= 64; 0 = skip) of those m1.
 * idx1 (b,m1) / idx2 (b,m2) are optional (NULL); new_xyz1 (b,m1,3) / new_xyz2 (b,m2,3) are
 * the gathered centroids.  Same bit-exact semantics as msr3d_furthest_point_sampling.
 * valid (b bytes, may be NULL): objects with valid[i] == 0 are skipped -- their outputs are left
 * untouched here and in msr3d_sa_level (the dataset pads scenes to 60 objects with a constant
 * cloud, /root/reference/data/datasets/dataset_wrapper.py:156-158, whose feature the caller
 * computes once instead of once per padding slot). */
int msr3d_sa_fps2(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                  float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid,
                  msr3d_stream_t stream);
/* msr3d_sa_fps2 AND, beside it, the ball query of level 1 on the centres it picks (ball_query_gpu.cu:9-44 with
 * radius1, nsample1; same index order, strict '<', first-hit fill): the FPS is one wave per cloud and a dependent chain
 * per pick, so three more waves of the same workgroup query each winner as soon as it is published.  ball_idx1
 * (b, m1, nsample1) is then what msr3d_sa_level / msr3d_sa_level_split take for level 1 with radius <= 0 (= "already
 * queried").  Clouds of 257..1024 rank slots that fit 48 KB of LDS; MSR3D_EINVAL otherwise (run the two launches). */
int msr3d_sa_fps2_query(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                        float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid, float radius1,
                        int nsample1, int *ball_idx1, msr3d_stream_t stream);

/* The two entries above with one more output, constant_out (b bytes, may be NULL): 1 for an object whose n points are
 * all bit-identical to its first point (coordinates and the other point_stride - 3 channels), 0 otherwise -- the
 * dataset's padding slots (dataset_wrapper.py:156-158).  Found while the cloud is staged in LDS for the sampling (no
 * extra pass over memory); a cloud too large to stage (> 64 KB) is reported 0; objects skipped by `valid` are not
 * written.  The distinct-row kernels (msr3d_sa_level*_rows) multiply ONE row per level for such an object. */
int msr3d_sa_fps2_flags(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                        float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid,
                        unsigned char *constant_out, msr3d_stream_t stream);
int msr3d_sa_fps2_query_flags(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1,
                              float *new_xyz1, int *idx2, float *new_xyz2, const unsigned char *valid, float radius1,
                              int nsample1, int *ball_idx1, unsigned char *constant_out, msr3d_stream_t stream);

/* msr3d_sa_fps2_query_flags AND msr3d_sa_plan12 (below) as ONE launch (round 6, ABI v29): what the planners of the
 * distinct-row levels read -- the object's ball rows, its two sets of centres -- is in the sampling launch's LDS when it
 * ends.  Level 1's task list is written by the last query wave of the workgroup to finish (beside the FPS wave's second
 * level), level 2's row lists by the FPS wave when its chain ends; tasks are appended with one atomic an object (the
 * order of the list is not reproducible; no result depends on it).  Arguments: those of msr3d_sa_fps2_query_flags
 * (constant_out REQUIRED), then task_ws1 (msr3d_sa_level1_rows_ws_bytes), level 2's radius, its output out2 (b, m2, 256:
 * rows of objects cut over several workgroups are zeroed here), dbg_ball_idx2 (may be NULL) and plan_ws2
 * (msr3d_sa_level2_rows_ws_bytes).  The caller then passes planned = 1 to msr3d_sa_level1_rows / msr3d_sa_level2_rows
 * on the same stream.  nsample1 = 32, m1 <= 64, m2 <= 16, n * point_stride a multiple of 4, and the shapes of
 * msr3d_sa_fps2_query; MSR3D_EINVAL otherwise (make the two calls). */
int msr3d_sa_fps2_query_plan(int b, int n, int point_stride, int m1, int m2, const float *pts, int *idx1, float *new_xyz1,
                             int *idx2, float *new_xyz2, const unsigned char *valid, float radius1, int nsample1,
                             int *ball_idx1, unsigned char *constant_out, void *task_ws1, float radius_l2, float *out2,
                             int *dbg_ball_idx2, void *plan_ws2, msr3d_stream_t stream);

/* One fused level.  `dims` = {C_in(+3), C1, C2, C3} must be one of the shipped
 * configurations (configs/msr3d.yaml:198-201) else MSR3D_EINVAL:
 *   level 1: dims {6,64,64,128};    pts (b,n,6) [xyz,rgb], feat = NULL, new_xyz (b,m,3),
 *            nsample 32; out (b,m,128) point-major; ball_idx (b,m,32) REQUIRED (workspace and
 *            output of the level's ball query, which runs as a first launch)
 *   level 2: dims {131,128,128,256}; pts = xyz (b,n,3) n<=64, feat (b,n,128) point-major,
 *            new_xyz (b,m,3), nsample 32; out (b,m,256)
 *   level 3: dims {259,256,512,768}; group-all over n = 16 points: pts = xyz (b,16,3),
 *            feat (b,16,256); new_xyz unused; out (b,768)
 * paramsL: layer L packed by the host as W'[KP/16][N/16][64][4], scale[N], shift[N] (floats):
 *   - K order: level 1 [dxyz, rgb], levels 2/3 [feat, (d)xyz]; zero-padded to KP = 16 / 144 / 272
 *     for the first layer, KP = K otherwise;
 *   - W' is the weight matrix in MFMA-fragment order: block (s, t) holds the 16 k x 16 columns of
 *     K-slab s and column tile t as 64 lanes x 4 floats,
 *         W'[s][t][lane][j] = W[16 t + (lane & 15)][16 s + 4 (lane >> 4) + j],
 *     i.e. exactly the 16 bytes lane `lane` feeds the matrix pipe, so a wave's operand load is one
 *     contiguous 1 KB read;
 *   - scale / shift: the eval-mode BN affine (conv bias folded into shift).
 * ball_idx (b,m,32): optional output at level 2.
 * valid: as in msr3d_sa_fps2 (level 3 handles two objects per workgroup: a padding object sharing
 * a workgroup with a valid one is computed too). */
int msr3d_sa_level(int level, int b, int n, int m, float radius, const float *pts,
                   const float *feat, const float *new_xyz, const int *dims,
                   const float *params1, const float *params2, const float *params3, float *out,
                   int *dbg_ball_idx, const unsigned char *valid, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Token GEMMs of the trainable part (situated encoder, projector): the nn.Linear calls of
 * /root/reference/modules/layers/transformers.py:200-252,314-329, model/ose3d_situation.py
 * :192,399-404 and model/msr3d/msr3d.py:84-86,277, forward and both backward products.
 * fp32 on f32-input MFMA.
 * ------------------------------------------------------------------------- */

/* C[m][n] = beta*C[m][n] + bias[n] + sum_k a(m,k)*b(n,k),  a(m,k) = a_kc ? A[m*lda+k] : A[k*lda+m],
 * b(n,k) = b_kc ? B[n*ldb+k] : B[k*ldb+n].  flags bit0: C = gelu(.) (exact erf form) and, if
 * C_pre != NULL, C_pre = the pre-activation (saved for backward).  bias / C_pre may be NULL.
 * p_drop > 0: inverted dropout on the final value (after GELU), mask = hash(*seed, salt, m*N+n)
 * like msr3d_dropout_add_ln_fwd (needs ldc == N); msr3d_gelu_bwd_f32 regenerates it.
 *
 * workspace (may be NULL): device memory for the split-K meeting point, used by launches that are
 * ordered on one stream.  Layout: MSR3D_GEMM_WS_COUNTERS ints that must be ZERO before the first
 * use (every launch leaves them zero), followed by the partial-sum records; 16 MiB covers every
 * shape of the path.  With a workspace the last workgroup of each tile adds the splits in a fixed
 * order and applies the epilogue: no zero-fill of C, no float atomics, results bit-reproducible
 * from run to run.  Without one the splits meet by memset + atomicAdd (order-dependent rounding). */
#define MSR3D_GEMM_WS_COUNTERS 1024
int msr3d_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float *A, int lda,
                   const float *B, int ldb, float *C, int ldc, const float *bias, float *C_pre,
                   int flags, float beta, float p_drop, const unsigned long long *seed,
                   unsigned salt, void *workspace, size_t workspace_bytes, msr3d_stream_t stream);

/* Weight AND bias gradient of y = x W^T + b in one launch: dw_db is a dense buffer of
 * N_out*K_in + N_out floats; on return dw_db[0 .. N_out*K_in) = dy^T x  (N_out x K_in) and
 * dw_db[N_out*K_in ..) = column sums of dy.  dy (M_tokens x N_out), x (M_tokens x K_in). */
int msr3d_linear_wgrad_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                           float *dw_db, void *workspace, size_t workspace_bytes,
                           msr3d_stream_t stream);

/* Same products ACCUMULATED onto dw (N_out x K_in, dense) and db (N_out; may be NULL): for
 * gradient buffers that were zeroed once for the whole step (the flat buffer of the
 * data-parallel engine), no memset, no temporary. */
int msr3d_linear_wgrad_acc_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                               float *dw, float *db, void *workspace, size_t workspace_bytes,
                               msr3d_stream_t stream);

/* Both backward products of y = x W^T + b in ONE launch (they are independent and each too small
 * to fill the chip): dx (M_tokens x K_in) = dx_beta * dx + dy W with dx_beta 0 or 1, and the
 * accumulation dw += dy^T x, db += colsum(dy) (db may be NULL) of msr3d_linear_wgrad_acc_f32. */
int msr3d_linear_bwd_f32(int M_tokens, int N_out, int K_in, const float *dy, const float *x,
                         const float *w, float *dx, float dx_beta, float *dw, float *db,
                         void *workspace, size_t workspace_bytes, msr3d_stream_t stream);

/* out[n] (+)= sum_m X[m*ldx + n]  (bias gradient).  accumulate bit0: add onto out; bit1: sum every
 * column in one workgroup (fixed order, bit-reproducible) instead of row chunks meeting by atomicAdd. */
int msr3d_colsum_f32(int M, int N, const float *X, int ldx, float *out, int accumulate,
                     msr3d_stream_t stream);

/* out = dropmask(dy) * gelu'(pre), n % 4 == 0, 16-byte aligned.  p_drop > 0 re-applies the
 * epilogue dropout of the forward msr3d_gemm_f32 call (same seed word, same salt). */
int msr3d_gelu_bwd_f32(long long n, const float *dy, const float *pre, float *out, float p_drop,
                       const unsigned long long *seed, unsigned salt, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Core of MultiHeadAttentionSpatial, 'cond' fusion
 * (/root/reference/modules/layers/transformers.py:205-248): scores, spatial term
 * sigmoid(w . pairwise_locs + bias), log-clamp, key-padding mask, softmax, P V.
 * q, k, v: token-major (B*L, ld_qkv) with head h in columns [h*dh, (h+1)*dh);
 * cond (B*L, ld_cond >= H*(spatial_dim+1)) = lang_cond_fc(x) as [bias, w_0..w_4] per head
 * (q, k, v and cond may be column blocks of ONE packed projection output);
 * pairwise_locs (B, L, L, spatial_dim); key_padding_mask (B, L) bytes, non-zero = padded;
 * ctx (B*L, H*dh); probs (B, H, L, L) (may be NULL in forward-only use).
 * Supported: L <= 128 (a 64-token tile for L <= 64, a 128-token tile above), dh = 32,
 * spatial_dim = 5; anything else -> MSR3D_EINVAL.
 *
 * mma: operand precision of the matrix products (QK^T, PV and the four backward products).
 * Softmax, the spatial term and all accumulation are fp32 in every mode; inputs/outputs stay fp32.
 *   MSR3D_MMA_F32   f32-input MFMA: the reference's arithmetic (autocast is disabled around the
 *                   encoder, model/ose3d_situation.py:377).  Default of the host side.
 *   MSR3D_MMA_BF16  operands rounded to bf16 (RNE) as they are fetched, bf16 MFMA, fp32 accumulate.
 *   MSR3D_MMA_FP8   operands rounded to OCP e4m3, fp8 MFMA, fp32 accumulate; forward only
 *                   (msr3d_spatial_attn_bwd returns MSR3D_EINVAL).
 * ------------------------------------------------------------------------- */
#define MSR3D_MMA_F32 0
#define MSR3D_MMA_BF16 1
#define MSR3D_MMA_FP8 2

int msr3d_spatial_attn_fwd(int B, int L, int H, int dh, int spatial_dim, const float *q,
                           const float *k, const float *v, int ld_qkv, const float *cond,
                           int ld_cond, const float *pairwise_locs,
                           const unsigned char *key_padding_mask, float *ctx, float *probs,
                           int mma, msr3d_stream_t stream);

/* Gradients w.r.t. q, k, v (token-major, ld_grad) and cond, given dctx (B*L, H*dh). */
int msr3d_spatial_attn_bwd(int B, int L, int H, int dh, int spatial_dim, const float *q,
                           const float *k, const float *v, int ld_qkv, const float *cond,
                           int ld_cond, const float *pairwise_locs,
                           const unsigned char *key_padding_mask, const float *probs,
                           const float *dctx, float *dq, float *dk, float *dv, int ld_grad,
                           float *dcond, int ld_dcond, int mma, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * QueryAndGroup.forward (pointnet2_utils.py:314-373, use_xyz) written token-major, the operand layout of
 * the training-mode SharedMLP: rows (b*m*nsample, KP), row ((b*m + j)*nsample + r) =
 * [xyz[b][p] - new_xyz[b][j] (3), feats[b][:, p] (C), zeros (KP - 3 - C)] with p = idx[b][j][r].
 * xyz (b,n,3), new_xyz (b,m,3), feats (b,C,n) channel-major as the reference keeps them (NULL when
 * C == 0), idx (b,m,nsample) from msr3d_ball_query.
 * _grad: d_feats (b,C,n) from d_rows, every element written, each a sum in ascending (j, r) order
 * (deterministic, the order of the sequential oracle); MSR3D_EINVAL when the inverted index of one
 * batch item (2n + 1 + m*nsample ints) exceeds 144 KB of LDS.  No gradient reaches xyz / new_xyz here.
 * ------------------------------------------------------------------------- */
int msr3d_group_rows(int b, int n, int m, int nsample, int C, int KP, const float *xyz,
                     const float *new_xyz, const float *feats, const int *idx, float *rows,
                     msr3d_stream_t stream);
int msr3d_group_rows_grad(int b, int n, int m, int nsample, int C, int KP, const float *d_rows,
                          const int *idx, float *d_feats, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * BatchNorm (training mode) + ReLU over a token-major (rows, C) tensor, forward and backward:
 * the normalisation of the unfrozen backbone's SharedMLP layers
 * (/root/reference/modules/third_party/pointnet2/pytorch_utils.py:39-66, nn.BatchNorm2d over
 * (b, C, npoint, nsample) = per-channel statistics over rows = b * npoint * nsample).
 *   fwd: y = relu((x - mean) * rstd * gamma + beta), mean / biased var over the rows; running_mean /
 *        running_var (may be NULL) updated as torch does (momentum, unbiased variance); save_mean,
 *        save_rstd (C) are kept for the backward.
 *   bwd: dx, and dgamma / dbeta (C, WRITTEN).
 * C % 4 == 0, C <= 1024.  partial_ws: 2 * C * ceil(rows / MSR3D_BN_CHUNK_ROWS) floats of scratch; the reductions
 * are two-stage and ordered (no float atomics: results are run-to-run bit-identical).
 * dgamma_acc / dbeta_acc (the _bwd entries; both or neither): buffers the parameter gradients are ADDED to as well
 * (the parameters' views of the flat gradient buffer: no AccumulateGrad launch per parameter).
 * partial_chunks (the _fwd entries): 0 = compute the first stage here; n > 0 = partial_ws already holds n partials
 * [n][2][C] (column sums, sums of squares) covering all rows -- written by the producer of x
 * (msr3d_rows_gemm_split's col_stats) -- and the pass over x that would form them is skipped.
 * ------------------------------------------------------------------------- */
#define MSR3D_BN_CHUNK_ROWS 512
int msr3d_bn_relu_train_fwd(long long rows, int C, const float *x, const float *gamma,
                            const float *beta, float eps, float momentum, float *running_mean,
                            float *running_var, float *y, float *save_mean, float *save_rstd,
                            float *partial_ws, int partial_chunks, msr3d_stream_t stream);
int msr3d_bn_relu_train_bwd(long long rows, int C, const float *x, const float *dy, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_rstd,
                            float *dx, float *dgamma, float *dbeta, float *partial_ws,
                            float *dgamma_acc, float *dbeta_acc, msr3d_stream_t stream);
/* The statistics alone (second stage over partial_chunks > 0 partials left by the producer of x; running statistics
 * updated as in _fwd): for a layer whose normalised activation is never written -- the next product applies
 * relu(batch_norm(.)) to its operand on load (msr3d_rows_gemm_split's a_bn, msr3d_wgrad_rows_split's x_bn).
 * bn_block (4, C) receives [gamma | beta | mean | rstd], the block those arguments take. */
int msr3d_bn_train_stats(long long rows, int C, const float *partial_ws, int partial_chunks, float eps, float momentum,
                         float *running_mean, float *running_var, const float *gamma, const float *beta, float *bn_block,
                         msr3d_stream_t stream);

/* The LAST layer of a SharedMLP fused with the neighbourhood max-pool that follows it
 * (pointnet2_modules.py:66-68, F.max_pool2d over nsample): rows = G * nsample consecutive rows per group;
 * pooled (G, C) = max over a group's rows of relu(bn(x)), argmax (G, C) = the FIRST row (0..nsample-1)
 * holding it, as max_pool2d chooses.  The (rows, C) activation is never materialised, and neither is
 * its gradient: _bwd takes dpooled (G, C) and routes it to the arg-max rows itself (rows whose pooled
 * value is 0 receive none: ReLU).  xsel (G, C) = x at the arg-max row: _fwd writes it and _bwd forms dgamma / dbeta
 * from the (G, C) arrays alone (only that row of a (group, channel) carries a gradient) instead of a pass over
 * the (rows, C) activation.  Other arguments as above. */
int msr3d_bn_relu_maxpool_train_fwd(long long rows, int C, int nsample, const float *x,
                                    const float *gamma, const float *beta, float eps, float momentum,
                                    float *running_mean, float *running_var, float *pooled, int *argmax,
                                    float *xsel, float *save_mean, float *save_rstd, float *partial_ws,
                                    int partial_chunks, msr3d_stream_t stream);
int msr3d_bn_relu_maxpool_train_bwd(long long rows, int C, int nsample, const float *x,
                                    const float *dpooled, const float *pooled, const int *argmax,
                                    const float *xsel, const float *gamma, const float *save_mean,
                                    const float *save_rstd, float *dx, float *dgamma, float *dbeta,
                                    float *partial_ws, float *dgamma_acc, float *dbeta_acc, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Data-only front of the situated encoder (inputs are dataset tensors, no gradients).
 * ------------------------------------------------------------------------- */

/* calc_pairwise_locs, 'center' / spatial_dim 5 / spatial_dist_norm
 * (/root/reference/modules/utils.py:88-137).  loc (B, L, ld_loc): object centres in the first
 * three floats of each row (ld_loc = 6 reads obj_locs in place); out (B, L, L, 5) =
 * [d/max d, dz/d, d_xy/d, dy/d_xy, dx/d_xy] for c_l - c_t, d = sqrt(sum^2 + eps), the max over
 * ALL L*L pairs of the sample.  L <= 128. */
int msr3d_pairwise_locs(int B, int L, const float *loc, int ld_loc, float eps, float *out,
                        msr3d_stream_t stream);

/* transform_to_agent_coor (modules/utils.py:60-82; skipped when transform == 0) followed by
 * generate_fourier_features (model/ose3d_situation.py:31-59): out (B, L, 3 + 6*num_bands) =
 * [p', sin(pi p' f), cos(pi p' f)], (coordinate, band) order inside each block; freqs
 * (num_bands) on the device (torch.linspace(1, max_freq, num_bands)); anchor_ori xyzw. */
int msr3d_agent_fourier(int B, int L, const float *loc, int ld_loc, const float *anchor_loc,
                        const float *anchor_ori, const float *freqs, int num_bands, int transform,
                        float *out, msr3d_stream_t stream);

/* out (M,D) = x (M,D) + v1 (D) + v2 (D; may be NULL): the type / orientation embeddings that every
 * object token receives (model/ose3d_situation.py:327-365).  D % 4 == 0, 16-byte aligned. */
int msr3d_add_row_vectors(int M, int D, const float *x, const float *v1, const float *v2,
                          float *out, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Row-wise tails of the spatial encoder layer: y = LayerNorm(dropout(a) + r) * gamma + beta
 * (/root/reference/modules/layers/transformers.py:250-251,324-328; r may be NULL and
 * p_drop 0 for the plain Linear->LayerNorm encoders of model/ose3d_situation.py:399-404).
 * D in {256,512,768,1024}.  s_out (M,D) = the pre-norm sum and stats (M,2) = {mean, rstd}
 * are what backward needs (may be NULL in inference).  The dropout mask is a hash of
 * (*seed, salt, element index): pass the same (seed, salt) to backward.
 * ------------------------------------------------------------------------- */
int msr3d_dropout_add_ln_fwd(int M, int D, const float *a, const float *r, const float *gamma,
                             const float *beta, float eps, float p_drop,
                             const unsigned long long *seed, unsigned salt, float *y, float *s_out,
                             float *stats, msr3d_stream_t stream);

/* da (M,D; NULL if a needs no grad) is written; dr (M,D; NULL if r was NULL) is written, or
 * added to when dr_accumulate != 0 (a residual consumed by several blocks: the gradients meet in
 * one buffer without a separate add); dgamma_acc / dbeta_acc (D) are ACCUMULATED into: by atomicAdd,
 * or -- partial_ws != NULL, ceil(M / MSR3D_LN_BWD_ROWS) * 2 * D floats (4 * D for the ln2 variant) --
 * through per-workgroup partials summed in a fixed order (bit-reproducible, one more tiny launch). */
#define MSR3D_LN_BWD_ROWS 16
int msr3d_dropout_add_ln_bwd(int M, int D, const float *dy, const float *s, const float *stats,
                             const float *gamma, float p_drop, const unsigned long long *seed,
                             unsigned salt, float *da, float *dr, int dr_accumulate,
                             float *dgamma_acc, float *dbeta_acc, float *partial_ws,
                             msr3d_stream_t stream);

/* Two chained tails sharing one residual, t = LN2(drop2(LN1(drop1(a) + r)) + r): the attention
 * block's residual + LayerNorm followed by the encoder layer's first one
 * (/root/reference/modules/layers/transformers.py:250-251 then :324-325), one launch each way.
 * s1 / stats1 and s2 / stats2 are the pre-norm sums and {mean, rstd} of the two LayerNorms; the
 * backward writes da and dr (the residual's total gradient) and accumulates both LayerNorms'
 * dgamma / dbeta.  D in {256, 512}. */
int msr3d_dropout_add_ln2_fwd(int M, int D, const float *a, const float *r, const float *gamma1,
                              const float *beta1, float eps1, float p1, unsigned salt1,
                              const float *gamma2, const float *beta2, float eps2, float p2,
                              unsigned salt2, const unsigned long long *seed, float *y, float *s1,
                              float *stats1, float *s2, float *stats2, msr3d_stream_t stream);
int msr3d_dropout_add_ln2_bwd(int M, int D, const float *dy, const float *s1, const float *stats1,
                              const float *gamma1, float p1, unsigned salt1, const float *s2,
                              const float *stats2, const float *gamma2, float p2, unsigned salt2,
                              const unsigned long long *seed, float *da, float *dr,
                              float *dgamma1_acc, float *dbeta1_acc, float *dgamma2_acc,
                              float *dbeta2_acc, float *partial_ws, msr3d_stream_t stream);

/* Advance the device-resident dropout seed word (once per training step, inside the graph). */
int msr3d_bump_seed(unsigned long long *seed, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Hand-off to the LLM: write the projected scene tokens into `inputs_embeds` / `attention_mask`
 * at the scene placeholders (/root/reference/model/msr3d/msr3d.py:279-287), without the host
 * sync of torch.where.  input_ids (B,T) int64; scene_embeds (n_scene, E) f32 (= llm_proj
 * output, B*L rows); scene_mask (n_scene) bytes (obj_masks, may be NULL with attention_mask);
 * inputs_embeds (B*T, E) of out_dtype 0 = f32, 1 = f16, 2 = bf16, updated IN PLACE;
 * attention_mask (B,T) int64 updated in place (may be NULL).  The k-th placeholder in
 * row-major order receives scene token k (k < n_scene).  map_ws: n_scene ints of workspace;
 * count_out: 1 int, receives the number of placeholders found (the reference errors out when
 * it differs from n_scene; the caller may check it without stalling the stream).
 * ------------------------------------------------------------------------- */
int msr3d_scene_scatter(int B, int T, int n_scene, int E, const long long *input_ids,
                        long long scene_token, const float *scene_embeds,
                        const unsigned char *scene_mask, int out_dtype, void *inputs_embeds,
                        long long *attention_mask, int *map_ws, int *count_out,
                        msr3d_stream_t stream);

/* The same hand-off with the projector fused in: for the k-th placeholder
 * inputs_embeds[pos_k] = cast(tokens[k] . weight^T + bias), tokens (n_scene, K) f32 (obj_tokens),
 * weight (E, K) / bias (E) f32 = llm_proj (/root/reference/model/msr3d/msr3d.py:84-86,277); the
 * fp32 (n_scene, E) projector output is never written.  The product runs on bf16 MFMA with fp32
 * accumulation (operands rounded to bf16 on the way in): use it where the result is consumed in
 * a 16-bit embedding dtype.  E % 128 == 0, K % 32 == 0, 16-byte aligned pointers. */
int msr3d_project_scatter_bf16(int B, int T, int n_scene, int E, int K, const long long *input_ids,
                               long long scene_token, const float *tokens, const float *weight,
                               const float *bias, const unsigned char *scene_mask, int out_dtype,
                               void *inputs_embeds, long long *attention_mask, int *map_ws,
                               int *count_out, msr3d_stream_t stream);

/* CUs the persistent set-abstraction kernels (msr3d_sa_level_split levels 1 and 2) leave to OTHER kernels:
 * their grids are sized to the chip, so a kernel that holds CUs beside them -- RCCL's gradient all-reduce,
 * issued beside the next batch's frozen encoder in the data-parallel step -- would push the last blocks of
 * a launch into a second round.  0 (default) on a GPU the step has to itself; msr3d_amd/dp.py sets it to
 * the number of RCCL channels when world > 1.  0 <= n <= 128; process-wide, takes effect at the next launch. */
int msr3d_set_reserved_cus(int n);

/* out[0] = sum_i a[i] b[i] (fp32, n % 4 == 0, 16-byte aligned), one launch, bit-reproducible (per-block
 * partials added in block order by the last block).  scratch: MSR3D_ADAMW_SCRATCH_FLOATS + 1 floats that
 * were ZERO at first use (the trailing word is a counter the kernel leaves at zero).  Stands for a scalar
 * loss head `(y * g).sum()` on the projector output -- what a trainer's reduction of the language-model
 * loss would be (trainer/leo_trainer.py:180-186); bench.py's synthetic loss uses it. */
int msr3d_dot_f32(long long n, const float *a, const float *b, float *scratch, float *out, msr3d_stream_t stream);

/* msr3d_sa_level on the bf16 matrix pipe at fp32 accuracy (csrc/sa_split.hip): every fp32 operand is
 * split exactly into three bf16 terms and a product is the six bf16 MFMA products above 2^-24 of it,
 * summed in the fp32 accumulator -- error per product below one fp32 rounding.  Same semantics and
 * outputs as msr3d_sa_level (same index ops); the parameters arrive pre-split:
 *   wK      [K/32][N/16][3][64][8] bf16: plane p, lane 16 g + i, element j = W_p[16 t + i][32 s + 8 g + j]
 *           (K padded with zeros to a multiple of 32; level 1: 32, level 2: 160, level 3: 288; K order [features, xyz] for
 *           levels 2 and 3; level 1 keeps its activations in registers, so the K axis of ITS layers 2 and 3 is numbered
 *           the way the previous layer's accumulators lie in the lanes: fragment position (g, e) of slab s holds channel
 *           32 s + 16 (e >> 2) + 4 g + (e & 3) -- pointnet2/fused.py::_pack_layer_split(kperm=True)),
 *           W = W_0 + W_1 + W_2 with W_0 = bf16(W), W_1 = bf16(W - W_0), W_2 = bf16(W - W_0 - W_1);
 *   affineK [2][N] f32: scale, shift (BN(eval) folded).
 * level: 1 (pts (b, n, 6) rows [xyz, rgb], feat unused, K padded 6 -> 32 in the order [xyz, rgb];
 * dbg_ball_idx is REQUIRED: the (b, m, 32) workspace the ball-query launch fills; m % 4 == 0),
 * 2 (the dominant kernel) or 3 (group-all: n = 16 points per object, m = 1, pts = the level's
 * xyz (b, 16, 3), feat (b, 16, 256), new_xyz / radius unused, out (b, 768)). */
int msr3d_sa_level_split(int level, int b, int n, int m, float radius, const float *pts, const float *feat,
                         const float *new_xyz, const void *w1, const float *affine1, const void *w2,
                         const float *affine2, const void *w3, const float *affine3, float *out,
                         int *dbg_ball_idx, const unsigned char *valid, msr3d_stream_t stream);

/* Level 3 of msr3d_sa_level_split (group-all: xyz (b, 16, 3), feat (b, 16, 256) -> out (b, 768); the same packed
 * weights and affines) dealt over the objects' FLAGS instead of four consecutive objects a workgroup (round 6, ABI v29;
 * pointnet2_modules.py:34-75 with the GroupAll grouper, pytorch_utils.py:11-36).  `constant` (b bytes, may be NULL) is
 * what msr3d_sa_fps2*_flags wrote: 1 for an object whose cloud is one repeated point (the dataset's padding slot,
 * dataset_wrapper.py:156-158) -- its sixteen level-3 rows are one row.  Decided on the device, no read-back: with R real
 * objects (valid, not constant) and C constant ones, if ceil(R / 3) + ceil(C / 48) workgroups fit the launch, a
 * workgroup takes THREE real objects (48 rows) or 48 constant objects (one row each, no maximum); otherwise four
 * consecutive objects, as msr3d_sa_level_split.  Every row's arithmetic is that entry's and max over sixteen
 * identical rows is the row: the same bits (tests/test_sa_rows_gpu.py).  Objects with valid[o] == 0 are not written.
 * Not in the reduced variant library (MSR3D_EINVAL there). */
int msr3d_sa_level3_tiles(int b, const float *xyz, const float *feat, const void *w1, const float *affine1, const void *w2,
                          const float *affine2, const void *w3, const float *affine3, float *out,
                          const unsigned char *valid, const unsigned char *constant, msr3d_stream_t stream);

/* Level 2 of msr3d_sa_level_split over the DISTINCT rows of each neighbourhood (round 5).  ball_query fills a
 * neighbourhood with fewer than nsample hits by repeating its first hit (ball_query_gpu.cu:35-39), the SharedMLP
 * acts on every row by itself and max is idempotent: multiplying only the min(hits, nsample) different (centre,
 * point) rows of a centre (one row -- index 0 -- for a centre without hits) gives the SAME BITS as multiplying all
 * nsample = 32.  Rows are packed densely across the centres of an object into 16-row MFMA tiles, up to 64 rows per
 * pass; each row's arithmetic is msr3d_sa_level_split's to the letter; the maximum is taken per centre (segmented).
 * Same arguments, layouts and outputs as msr3d_sa_level_split(level = 2, ...) (xyz = that call's `pts`), n <= 64
 * points, m <= 16 centres per object (MSR3D_EINVAL otherwise: use msr3d_sa_level_split), plus plan_ws (below) and
 *   constant (b bytes, may be NULL): objects whose cloud is ONE repeated point, as msr3d_sa_fps2* report them (the
 *   dataset's padding slots, dataset_wrapper.py:156-158): every row of such an object is the same row; one is
 *   multiplied and written to all m centres.  Flagging an object that is not constant is the caller's error. */
int msr3d_sa_level2_rows(int b, int n, int m, float radius, const float *xyz, const float *feat,
                         const float *new_xyz, const void *w1, const float *affine1, const void *w2,
                         const float *affine2, const void *w3, const float *affine3, float *out,
                         int *dbg_ball_idx, const unsigned char *valid, const unsigned char *constant,
                         void *plan_ws, int planned, msr3d_stream_t stream);
/* (planned != 0: plan_ws was filled by msr3d_sa_plan12 on this stream -- the call is then the products' launch alone.)
 * Bytes of plan_ws for b objects (device memory, 16-byte aligned, the caller's; contents need not survive the call's
 * work on the stream).  Two launches: one wave per object runs its m ball queries and writes the list of its distinct
 * rows there; the multiplying workgroups deal the objects among themselves by work (a balanced static deal, the same
 * in every workgroup) and read each of their objects' lists one object ahead. */
size_t msr3d_sa_level2_rows_ws_bytes(int b);

/* Level 1 of msr3d_sa_level_split over distinct rows: a centre with at most 16 different neighbours (slot 16 of its
 * ball-query row repeats slot 0) needs one 16-row MFMA tile, so two such centres share a wave's 32 rows; a constant
 * object (see msr3d_sa_fps2_flags) is one such task whose result is written to all m centres.  Same bits as
 * msr3d_sa_level_split(level = 1, ..., radius <= 0) on the same ball_idx (b, m, 32) -- which must hold the level's
 * neighbour lists already (msr3d_sa_fps2_query* or msr3d_ball_query).  m <= 64; task_ws: msr3d_sa_level1_rows_ws_bytes(b, m)
 * bytes of device memory, 16-byte aligned (two launches: the task list, then the products). */
int msr3d_sa_level1_rows(int b, int n, int m, const float *pts, const float *new_xyz, const int *ball_idx,
                         const void *w1, const float *affine1, const void *w2, const float *affine2,
                         const void *w3, const float *affine3, float *out, const unsigned char *valid,
                         const unsigned char *constant, void *task_ws, int planned, msr3d_stream_t stream);
size_t msr3d_sa_level1_rows_ws_bytes(int b, int m);

/* The planning launches of msr3d_sa_level1_rows and msr3d_sa_level2_rows as ONE launch: both read only what the sampling
 * launch wrote (level 1: ball_idx1 (b, m1, 32); level 2: its points xyz2 (b, n2, 3) = level 1's centres and its centres
 * new_xyz2 (b, m2, 3)), and a dependent launch costs ~5 us on this part before it does anything.  Arguments as the two
 * calls take them (radius2: level 2's radius; out2: level 2's output, whose rows of multi-chunk objects are zeroed here);
 * afterwards call both with planned = 1 on the same stream, level 1 first.  MSR3D_EINVAL for a planned = 1 call without
 * a plan; a plan that is never consumed is discarded by the next call. */
int msr3d_sa_plan12(int b, int m1, const int *ball_idx1, void *task_ws1, int n2, int m2, float radius2,
                    const float *xyz2, const float *new_xyz2, float *out2, int *dbg_ball_idx2, void *plan_ws2,
                    const unsigned char *valid, const unsigned char *constant, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * The trainable part as a fixed schedule of fused launches (msr3d_amd/fused_model.py):
 * strip GEMMs with the row-local work of /root/reference/modules/layers/transformers.py:250-251,
 * 324-328 (dropout + residual + LayerNorm chains and their backward) applied while the operand is
 * staged, multi-problem split-K launches for the long reductions and the weight gradients, and the
 * row kernels around the layers (model/ose3d_situation.py:327-365,399-404; modules/utils.py:60-137).
 * ------------------------------------------------------------------------- */

/* prologue applied to the 64-row operand strip (K = 256 = row width) */
#define MSR3D_PRO_PLAIN 0   /* A = a0 */
#define MSR3D_PRO_ADD 1     /* A = a0 + a1 (+ column vectors g1, b1 if given);            o1 = A */
#define MSR3D_PRO_LN 2      /* v = drop(a0; p1, salt1) + a1; A = LN(v; g1, b1, eps1) (+ a2);
                               o0 = v, ost1 = (mean, rstd) per row, o1 = A (each optional)  */
#define MSR3D_PRO_LN2 3     /* v1 = drop(a0; p1) + a1; y = LN(v1; g1, b1); v2 = drop(y; p2) + a1;
                               A = LN(v2; g2, b2); o0 = v1, ost1, o2 = v2, ost2, o1 = A     */
#define MSR3D_PRO_LNBWD 4   /* a0 = dy, a1 = v, st1: dx = LN-bwd; o1 = dx; A = drop-bwd(dx; p1, salt1);
                               o0 = A; dg1 / db1 += the strip's column sums                */
#define MSR3D_PRO_LN2BWD 5  /* backward of PRO_LN2: a0 = dt, a1 = v1, a2 = v2, st1, st2;
                               o1 = d(a1) (both tails), A = o0 = d(a0); dg1, db1, dg2, db2 */
/* epilogue */
#define MSR3D_EPI_BIAS 0    /* C = acc + bias (bias may be NULL) */
#define MSR3D_EPI_GELU 1    /* Cpre = acc + bias; C = drop(gelu(Cpre); p_drop, salt), mask index row*N+col */
#define MSR3D_EPI_GELUBWD 2 /* C = drop-bwd(acc; p_drop, salt) * gelu'(pre_in[row*N+col])  (b_kc = 0 only) */

typedef struct msr3d_strip_gemm {
  int M, N;                     /* token rows, output columns; the reduction length is 256 */
  int pro, epi;
  int b_kc;                     /* 1: b(n,k) = W[n*ldw + k] (forward);  0: b(n,k) = W[k*ldw + n] (dx = dy W) */
  int groups_per_wg;            /* 64-column groups per workgroup; <= 0: chosen for ~1 workgroup per CU */
  const float *a0, *a1, *a2;    /* (M,256) inputs of the prologue */
  const float *st1, *st2;       /* (M,2) saved (mean, rstd) for the backward prologues */
  const float *g1, *b1, *g2, *b2;
  float eps1, eps2, p1, p2;
  unsigned salt1, salt2;
  const unsigned long long *seed;   /* device dropout seed word (msr3d_bump_seed) */
  float *o0, *o1, *o2, *ost1, *ost2;    /* row results, written once per strip */
  float *dg1, *db1, *dg2, *db2;         /* LayerNorm parameter gradients: accumulated into (atomics) */
  const float *W; int ldw;
  const float *bias;
  float *C; int ldc;
  float *Cpre;
  const float *pre_in;
  float p_drop; unsigned salt;
} msr3d_strip_gemm_t;

/* C (M,N) = epilogue(prologue(a0, a1, a2) . b): see the MSR3D_PRO_ / MSR3D_EPI_ codes.  All row
 * tensors are dense (M,256) f32, 16-byte aligned.  b_kc = 0 needs N % 64 == 0, even ldw / ldc. */
int msr3d_strip_gemm_f32(const msr3d_strip_gemm_t *p, msr3d_stream_t stream);

#define MSR3D_GEMM_MULTI_MAX 4
typedef struct msr3d_gemm_problem {
  int a_kc, b_kc;               /* operand layouts as msr3d_gemm_f32 */
  int M, N, K;
  const float *A; int lda;
  const float *B; int ldb;
  float *C; int ldc;
  const float *bias;            /* added once (forward products) */
  float beta;                   /* 0 or 1; K-splits meet by atomicAdd, so with beta = 1 C holds the value
                                   to add to (e.g. zeros from msr3d_step_begin, or a residual gradient) */
  float *colsum;                /* a_kc = 0, beta = 1 only: colsum[m] += sum_k a(m,k)  (bias gradient) */
  int single_run;               /* 1: never split K (no atomics: bit-reproducible, every output row
                                   independent of the others in the launch) */
} msr3d_gemm_problem_t;

/* Up to MSR3D_GEMM_MULTI_MAX independent products in one launch (dx, dW + db of a linear layer and
 * whatever other weight gradient is ready at that point of the schedule). */
int msr3d_gemm_multi_f32(int n, const msr3d_gemm_problem_t *problems, msr3d_stream_t stream);

/* Data-only front of the prompter for one batch, ONE launch: pad_out = !obj_valid, valid_out =
 * obj_valid (optional copy into a step's static buffer), locs_out = obj_locs (B,L,6), pairwise_out
 * (B,L,L,5) = msr3d_pairwise_locs, fourier_out (B,L,3+6*num_bands) = msr3d_agent_fourier(transform).
 * anchor_loc_out (B,3) / anchor_ori_out (B,4): optional copies of the anchor pose (a training step's static
 * buffers).  L <= 128. */
int msr3d_scene_prologue(int B, int L, const float *obj_locs, const unsigned char *obj_valid,
                         const float *anchor_loc, const float *anchor_ori, const float *freqs,
                         int num_bands, int transform, float eps, float *pairwise_out,
                         float *fourier_out, float *locs_out, unsigned char *pad_out,
                         unsigned char *valid_out, float *anchor_loc_out, float *anchor_ori_out,
                         msr3d_stream_t stream);

/* msr3d_scene_prologue for situation_type 'as_object' (/root/reference/model/ose3d_situation.py:334-349): every scene gets
 * the AGENT as token 0 -- box [anchor_loc | agent_size (3 floats: the module's constant anchor_size)], always valid --
 * in front of its O objects; outputs are over L = O + 1 tokens (pairwise (B,L,L,5), locs (B,L,6), pad / valid (B,L);
 * fourier_out (B,L,3 + 6 nb): Fourier rows of the raw centres, no agent-frame transform -- this configuration does not read
 * them), and quat_fourier_out (B, 4 + 8 nb), optional, receives generate_fourier_features of the orientation quaternion:
 * the orientation encoder's input. */
int msr3d_scene_prologue_agent(int B, int O, const float *obj_locs, const unsigned char *obj_valid,
                               const float *anchor_loc, const float *anchor_ori, const float *agent_size,
                               const float *freqs, int num_bands, float eps, float *pairwise_out, float *fourier_out,
                               float *locs_out, unsigned char *pad_out, unsigned char *valid_out, float *quat_fourier_out,
                               float *anchor_loc_out, float *anchor_ori_out, msr3d_stream_t stream);

/* zero_region[0..n_floats) = 0 (n_floats % 4 == 0) and, if seed != NULL, the dropout seed bump of
 * msr3d_bump_seed -- the step's one fill for all split-K meeting points. */
int msr3d_step_begin(float *zero_region, long long n_floats, unsigned long long *seed,
                     msr3d_stream_t stream);

/* pos (M,256) = LN_a(fourier (M,KF) Wa^T + ba) + LN_b(locs[:,3:6] Wb^T + bb): loc_embedding_encoder +
 * size_embedding_encoder (model/ose3d_situation.py:399-404); locs (M,6); KF <= 64; saves the
 * pre-norm values s_a, s_b (M,256) and statistics (M,2) for the backward. */
int msr3d_pos_embed_fwd(int M, int KF, const float *fourier, const float *locs, const float *Wa,
                        const float *ba, const float *gamma_a, const float *beta_a, float eps_a,
                        const float *Wb, const float *bb, const float *gamma_b, const float *beta_b,
                        float eps_b, float *pos, float *s_a, float *stats_a, float *s_b,
                        float *stats_b, msr3d_stream_t stream);
/* msr3d_pos_embed_fwd + the first layer's input in the same launch (round 6; what msr3d_scene_rows' MSR3D_PRO_ADD launch
 * did behind it): xin0 (M,256) = ((x0 + pos) + type_row) + orientation_row -- x0 = obj_linear_projection's output, the two
 * constant 256-vectors of ose3d_situation.py:356-360 (orientation_row optional) -- also written as the first attention
 * block's operand planes (planes: (M / L scenes, 3, 64, 256) bf16 or NULL; L tokens a scene, L <= 64 with planes). */
int msr3d_pos_embed_tokens_fwd(int M, int L, int KF, const float *fourier, const float *locs, const float *Wa,
                               const float *ba, const float *gamma_a, const float *beta_a, float eps_a,
                               const float *Wb, const float *bb, const float *gamma_b, const float *beta_b,
                               float eps_b, float *pos, float *s_a, float *stats_a, float *s_b, float *stats_b,
                               const float *x0, const float *type_row, const float *orientation_row, float *xin0,
                               unsigned short *planes, msr3d_stream_t stream);

/* Row-wise backward of msr3d_pos_embed_fwd: d pos = d0 + d1 + d2 (d1, d2 optional);
 * d_lin_a / d_lin_b (M,256) = gradients of the two linear outputs; the LayerNorm parameter
 * gradients and colsum(d0) (added to colsum1 and colsum2, each optional) are accumulated into. */
int msr3d_pos_embed_bwd(int M, const float *d0, const float *d1, const float *d2, const float *s_a,
                        const float *stats_a, const float *gamma_a, const float *s_b,
                        const float *stats_b, const float *gamma_b, float *d_lin_a, float *d_lin_b,
                        float *dgamma_a, float *dbeta_a, float *dgamma_b, float *dbeta_b,
                        float *colsum1, float *colsum2, msr3d_stream_t stream);

/* The front of the situated encoder when the AGENT IS A TOKEN (situation_type 'as_object', /root/reference/model/
 * ose3d_situation.py:334-353,384-386: configs/leo_3_dataset_pure_txt.yaml's prompter), rows (B, L) with row (b, 0) the
 * agent and rows (b, 1..L-1) the scene's objects:
 *     v    = (anchor_feat + a_ori[b]) + type_emb[1]                 agent row   (a_ori (B,256) = orientation_encoder output)
 *          = (x0[row] + ori_feat) + type_emb[0]                     object rows (x0 (B L,256) = obj_linear_projection
 *                                                                   output; its agent rows are not read; ori_feat may be NULL)
 *     pos  = LayerNorm(loc6[row] W_loc^T + b_loc)                   loc_layers[0]: W_loc (256,6), loc6 (B L,6)
 *     xin0 = v + pos, also written as the first attention block's operand planes (planes: (B,3,64,256) bf16, or NULL;
 *            L <= 64 with planes)
 * pos, s_lin (the pre-norm linear output), stats (B L,2) are kept for the later layers ('same_all') and the backward. */
int msr3d_anchor_front_fwd(int B, int L, const float *x0, const float *a_ori, const float *anchor_feat,
                           const float *type_emb, const float *ori_feat, const float *loc6, const float *W_loc,
                           const float *b_loc, const float *gamma, const float *beta, float eps, float *pos,
                           float *s_lin, float *stats, float *xin0, unsigned short *planes, msr3d_stream_t stream);
/* Its row-wise backward.  d pos = d0 + d1 + d2 (d1, d2: the later layers' input gradients, optional), d v = d0.
 * d_lin (B L,256) = gradient of the location layer's output (LayerNorm backward of d pos); dgamma / dbeta accumulated;
 * the column sum of d0 over the OBJECT rows is added to obj_sum_a / _b / _c (type_embedding[0], object_orientation_feat,
 * obj_linear_projection.bias) and over the AGENT rows to agent_sum_a / _b (type_embedding[1], anchor_feat); each optional.
 * (The orientation encoder's and the projection's weight gradients read d0 itself: agent rows at stride L * 256, and all
 * rows against a feature matrix whose agent rows are zero.)  Float atomics: not bit-reproducible. */
int msr3d_anchor_front_bwd(int B, int L, const float *d0, const float *d1, const float *d2, const float *s_lin,
                           const float *stats, const float *gamma, float *d_lin, float *dgamma, float *dbeta,
                           float *obj_sum_a, float *obj_sum_b, float *obj_sum_c, float *agent_sum_a,
                           float *agent_sum_b, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * The trainable part as SCENE-LOCAL fused blocks on the bf16 matrix pipe at fp32 accuracy
 * (round 3; /root/reference/modules/layers/transformers.py:200-252,314-329 and their autograd).
 *
 * Every fp32 operand is split exactly into three bf16 terms and a product is six bf16 MFMA products
 * accumulated in fp32 (csrc/split_mma.h; the arithmetic of msr3d_sa_level_split).  Weights are split
 * and packed in MFMA fragment order once per optimiser step (msr3d_split_pack); activations are split
 * where they are produced and never leave the chip between the two products of a block.
 *
 * A block's unit of work is one scene (L <= 64 token rows = one 64-row tile) x one slice; its input rows
 * arrive as three bf16 planes `xp` (B, 3, 64, 256) written by msr3d_scene_rows (rows past L zero):
 *
 *   ATTN_FWD   (scene, head)      [q|k|v|cond]_h = rows W_h^T + b;  ctx_h = spatial attention (attn_core.h)
 *                                 part[h]  = ctx_h Wfc[:, 32h:32h+32]^T
 *   FFN_FWD    (scene, 128 hidden) pre = rows W1_s^T + b1; h = dropout(gelu(pre));   part[s] = h W2[:, s]^T
 *                                 (the (M, 2048) activation feeds the second product from LDS)
 *   FFN_BWD    (scene, 128 hidden) d_h = rows W2[:, s]; d_pre = gelu-bwd;               part[s] = d_pre W1[s, :]
 *   ATTN_BWD   (scene, head)      d_ctx_h = rows Wfc[:, 32h:..]; d[q|k|v|cond]_h = attention backward
 *                                 part[h]  = d[q|k|v|cond]_h W_h
 *   LINEAR     (scene, 256 cols)  C = rows W^T + b                                     (llm_proj)
 *   LINEAR_KSPLIT (scene, 256 k)  part[s] = a0[:, 256 s : 256 s + 256] W[.., s]^T      (d tokens = d scene . W_llm;
 *                                 a0 (M, lda0) f32 is split on the way in)
 *
 * The slices' partial products land in slabs part[slice] (M, 256), `part_stride` floats apart; the next
 * msr3d_scene_rows launch sums them in slab order (no atomics: the step is bit-reproducible), adds the
 * bias / residual, applies the row-local chain that follows (dropout + residual + LayerNorm once or
 * twice, or their backward: the MSR3D_PRO_* codes of msr3d_strip_gemm_f32) ONCE per row, and writes the
 * result as the next block's planes.  Side outputs (q|k|v|cond, ctx, probabilities, pre, h, and in
 * backward d_pre, d[q|k|v|cond]) are written by the slice that owns them, dense f32, for the backward
 * blocks and the weight-gradient launch (msr3d_wgrad_split).
 * ------------------------------------------------------------------------- */
typedef struct msr3d_pack_job {
  const float *src; int ld;     /* source matrix (row-major f32), row stride */
  int transposed;               /* 0: Op(n, k) = src[map(n) * ld + k];  1: Op(n, k) = src[map(k) * ld + n] */
  int rows, k;                  /* operand shape: rows % 16 == 0, k % 32 == 0; unmapped positions are zero */
  int nseg;                     /* the map over the mapped axis: up to 4 runs dst -> src */
  int seg_dst[4], seg_len[4], seg_src[4];
  unsigned short *dst;          /* [k/32][rows/16][3 planes][64 lanes][8] bf16 */
} msr3d_pack_job_t;

/* jobs, piece_prefix (njobs + 1 ints: first (slab, tile) piece of each job; [njobs] = total): DEVICE memory. */
int msr3d_split_pack(int njobs, const msr3d_pack_job_t *jobs, const int *piece_prefix, int total_pieces,
                     msr3d_stream_t stream);
/* msr3d_split_pack + msr3d_step_begin in ONE launch (extra workgroups zero-fill `zero_region` and bump the dropout seed
 * word): the two are independent and open every step of the scene-block schedule. */
int msr3d_split_pack_begin(int njobs, const msr3d_pack_job_t *jobs, const int *piece_prefix, int total_pieces,
                           float *zero_region, long long n_floats, unsigned long long *seed, msr3d_stream_t stream);

/* C (M, ldc) = A (M, lda) . W^T + bias for a few hundred rows and a long reduction (K % 128 == 0, N % 32 == 0) on the
 * bf16 matrix pipe at fp32 accuracy; W pre-split in msr3d_split_pack's layout ([K / 32][N / 16] pieces, transposed = 0).
 * No K split: row-independent and bit-reproducible.  Replaces the f32-MFMA launch of the encoder's `fc`
 * (/root/reference/modules/layers/pointnet.py:52-63).  a, C, w_pack, bias: 16-byte aligned. */
int msr3d_rows_linear_split(int M, int N, int K, const float *a, int lda, const unsigned short *w_pack, unsigned w_bytes,
                            const float *bias, float *C, int ldc, msr3d_stream_t stream);

#define MSR3D_BLK_ATTN_FWD 0
#define MSR3D_BLK_FFN_FWD 1
#define MSR3D_BLK_FFN_BWD 2
#define MSR3D_BLK_ATTN_BWD 3
#define MSR3D_BLK_LINEAR 4
#define MSR3D_BLK_LINEAR_KSPLIT 5

typedef struct msr3d_scene_block {
  int kind, B, L;               /* scenes, token rows per scene (L <= 64) */
  const unsigned short *xp;     /* (B, 3, 64, 256) bf16 planes of the input rows */
  const float *a0; int lda0;    /* LINEAR_KSPLIT only: (M, lda0) f32, lda0 % 256 == 0, lda0 <= 5120 */
  const unsigned short *w1; unsigned w1_bytes;   /* packed operand of product 1 (see the table above) */
  const float *bias1;           /* product 1's bias: ATTN_FWD the packed [q|k|v|cond] bias, FFN_FWD b1, LINEAR b */
  const unsigned short *w2; unsigned w2_bytes;   /* packed operand of product 2 */
  float *part; long long part_stride;            /* (slices, M, 256) partial products, slabs part_stride floats apart */
  /* FFN */
  float *pre;                   /* (M, ff): FFN_FWD writes, FFN_BWD reads */
  float *h;                     /* (M, ff): FFN_FWD writes dropout(gelu(pre)); FFN_BWD writes d_pre */
  int ff;                       /* ff % 128 == 0, ff <= 2048 */
  float p_drop; unsigned salt;
  const unsigned long long *seed;
  /* ATTN */
  float *qkvc; int ldq;         /* (M, ldq) [q 256 | k 256 | v 256 | cond 6 H]: ATTN_FWD writes, ATTN_BWD reads */
  float *dqkvc;                 /* ATTN_BWD writes, same layout */
  const float *ploc;            /* (B, L, L, 5) */
  const unsigned char *pad;     /* (B, L) key padding mask */
  float *probs;                 /* (B, H, L, L): ATTN_FWD writes, ATTN_BWD reads */
  float *ctx;                   /* (M, 256): ATTN_FWD writes */
  int H;                        /* 8 */
  /* LINEAR */
  float *C; int ldc; int N;     /* N % 256 == 0 */
  /* Row tiles that ignore scene boundaries (round 6; the FFN_* / LINEAR* kinds, whose work is row-local): rows_total > 0
   * says the token matrix has that many rows in all, cut into B tiles of L rows (L = 64, B = ceil(rows_total / L)), the
   * last one shorter -- a scene of more than 64 tokens (BASELINE's stress configuration: 121) then still runs its
   * feed-forward and projector halves on these kernels.  0: B scenes of exactly L rows. */
  int rows_total;
} msr3d_scene_block_t;

int msr3d_scene_block(const msr3d_scene_block_t *p, msr3d_stream_t stream);
/* The attention forward block's launch form: 1 (the library's default; MSR3D_ATTN_FWD_SPLIT=0 in the environment selects 0
 * at first use) = two workgroups per (scene, head), each 32 query rows, keys split over eight waves: the faster form when
 * the step has the chip to itself; 0 = one workgroup per (scene, head) -- 128 workgroups at the bench shape: slower alone
 * (19.3 against 16 us), FASTER in the pipelined schedule, where the next batch's frozen encoder runs beside the trainable
 * part and takes the CUs this form leaves (round 6: 0.851 against 0.863 ms a step).  Same values either way.
 * form = 0 / 1 selects, -1 only queries; returns the form in force, MSR3D_EINVAL for another value. */
int msr3d_attn_fwd_form(int form);

/* One wave per token row: a0 = sum_s part[s] (+ extra) (+ a0_bias) in slab order (nslab == 0: a0 itself),
 * optionally stored whole (sum_out); then the MSR3D_PRO_* chain with the operands / outputs of
 * the msr3d_strip_gemm_t struct -- in the backward codes o1 = the residual gradient, STORED, to be passed
 * as the next sum's `extra`; the chain's result goes to `xp` (B, 3, 64, 256) bf16 as three exactly-split planes
 * (optional).  MSR3D_PRO_PLAIN: sum only.  LayerNorm parameter gradients are accumulated (atomicAdd,
 * one per column and four rows) or stored as per-workgroup partials (grad_partials).  nslab <= 20. */
typedef struct msr3d_scene_rows {
  int M, L, pro;
  const float *a0;
  const float *part; int nslab; long long part_stride;
  const float *extra;           /* (M, 256) or NULL */
  const float *a0_bias;         /* (256) or NULL */
  float *sum_out;               /* (M, 256) or NULL */
  const float *a1, *a2;
  const float *st1, *st2;
  const float *g1, *b1, *g2, *b2;
  float eps1, eps2, p1, p2;
  unsigned salt1, salt2;
  const unsigned long long *seed;
  float *o0, *o1, *o2, *ost1, *ost2;
  float *dg1, *db1, *dg2, *db2;
  unsigned short *xp;
  int grad_partials;            /* != 0: dg1 .. db2 are PARTIAL buffers ((M + 3) / 4, 256) each -- every workgroup stores
                                   its four rows' column sums to row blockIdx.x (no atomics); msr3d_colsum_partials adds
                                   them up in row order */
} msr3d_scene_rows_t;
int msr3d_scene_rows(const msr3d_scene_rows_t *p, msr3d_stream_t stream);

/* dst[c] += sum_{i < n} part[i * 256 + c], c < 256, for every job, in row order (bit-reproducible): the second half
 * of the LayerNorm parameter gradients (norm1 / norm2 / the attention tail's LayerNorm of every layer,
 * /root/reference/modules/layers/transformers.py:250-251,324-328) when msr3d_scene_rows wrote partials.
 * jobs: device memory. */
typedef struct {
  const float *part;
  float *dst;
  int n, reserved;
} msr3d_colsum_job_t;
int msr3d_colsum_partials(int n_jobs, const msr3d_colsum_job_t *jobs, msr3d_stream_t stream);

/* All weight gradients of a step in ONE launch: for every problem dW (n_out, k_in) += dy^T x over the
 * M token rows (dy (M, n_out), x (M, k_in), dense f32) and, optionally, db (n_out) += colsum(dy).
 * fp32-accurate on the bf16 pipe (operands split on the way into LDS); one workgroup owns a 128 x 128
 * tile of dW over the WHOLE reduction: no split-K, no atomics, bit-reproducible; `dW` holds the value to
 * add to (the flat gradient buffer).  problems, tile_prefix (n + 1 ints; problem i owns workgroups
 * [tile_prefix[i], tile_prefix[i+1]), a multiple of 8 >= its ceil(n_out/128) * ceil(k_in/128) tiles): DEVICE memory. */
typedef struct msr3d_wgrad_problem {
  const float *dy; int ldy; int n_out;
  const float *x; int ldx; int k_in;
  int M;
  int xcd_rot;   /* 0..7: which XCD takes the problem's FIRST run of tiles (workgroup b of the launch runs on XCD b % 8;
                    a problem's padding workgroups -- it owns a multiple of 8 -- then fall on other XCDs for every problem) */
  float *dW; int ldw;
  float *db;
} msr3d_wgrad_problem_t;
int msr3d_wgrad_split(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix, int total_tiles,
                      msr3d_stream_t stream);
/* msr3d_wgrad_split with the jobs of msr3d_colsum_partials as n_jobs EXTRA workgroups of the same launch (they start in
 * the launch's second, partly empty round: the LayerNorm parameter gradients cost no launch of their own).  Same sums,
 * same order, as the two separate calls. */
int msr3d_wgrad_split_colsum(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix, int total_tiles,
                             int n_jobs, const msr3d_colsum_job_t *jobs, msr3d_stream_t stream);

/* The same launch with every tile's token reduction cut in TWO units (2 x total_tiles workgroups) so that a step's
 * ~1.4 tiles per CU spread evenly over the chip.  The unit that STARTS first parks its 128 x 128 partial in the
 * tile's workspace slot, the one that starts second adds first half + second half -- in that order, whichever of
 * them it is -- onto dW: still no float atomics, still bit-reproducible, identical sums for both launch forms only
 * up to fp32 association.  workspace: total_tiles x MSR3D_WGRAD_HALF_SLOT_FLOATS floats; sync: 2 x total_tiles
 * ints, ZERO before the first launch (every launch leaves them zero).  total_tiles % 8 == 0. */
#define MSR3D_WGRAD_HALF_SLOT_FLOATS (128 * 128 + 128)
int msr3d_wgrad_split_halves(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix, int total_tiles,
                             float *workspace, long long workspace_floats, int *sync, msr3d_stream_t stream);
/* MIXED form (round 5): only the tiles of the launch's PARTIAL round are cut.  Workgroups [0, whole_tiles) take whole
 * tiles, the other H = total_tiles - whole_tiles tiles run as two half-reductions each (the protocol above), n_jobs
 * column-sum workgroups follow (msr3d_wgrad_split_colsum).  The caller picks whole_tiles so that the halves fill what the
 * whole tiles leave of the chip's second round (msr3d_amd/scene_blocks.py: real tiles beyond one per CU, when they are at
 * most half a round).  whole_tiles % 8 == 0 and H % 8 == 0; workspace: H x MSR3D_WGRAD_HALF_SLOT_FLOATS floats; sync:
 * 2 H ints, zero before the first launch.  H == 0 is msr3d_wgrad_split_colsum. */
int msr3d_wgrad_split_mixed(int n, const msr3d_wgrad_problem_t *problems, const int *tile_prefix, int total_tiles,
                            int whole_tiles, int n_jobs, const msr3d_colsum_job_t *jobs, float *workspace,
                            long long workspace_floats, int *sync, msr3d_stream_t stream);

/* STREAM form (round 6): the launch's tiles as one sequence of slab pairs dealt evenly to n_wgs PERSISTENT workgroups
 * (one per CU): workgroup w runs pieces [wg_first[w], wg_first[w + 1]) one after the other.  A piece is a whole tile
 * (slot < 0), one of the two parts of a tile cut at slab s0 / s1 (even; slot = the tile's workspace slot), or a column-sum
 * job (kind = 1, prob = its index in `jobs`).  The parts of a cut tile do not meet inside the launch: the part with the
 * EARLIER slabs (second = 0) parks its 128 x 128 partial + column sums in the slot, the part with the later slabs
 * (second = 1) adds onto dW like a whole tile, and a second, small launch adds the parked partials --
 * dW = (dW + later) + earlier, a fixed order: bit-reproducible, no atomics, no flags (an in-kernel hand-over between CUs
 * costs more than the kernel boundary: profiles/r06_last_arriver_probe.txt).  slot_piece[s] = index of slot s's parking
 * piece.  The planner (msr3d_amd/scene_blocks.py::WgradTable) cuts a tile at most once.  workspace: n_slots x
 * MSR3D_WGRAD_HALF_SLOT_FLOATS floats.  problems, pieces, wg_first (n_wgs + 1 ints), slot_piece, jobs: DEVICE memory. */
typedef struct msr3d_wgrad_piece {
  int kind;                 /* 0: (part of) a tile, 1: column-sum job */
  int prob;                 /* problem index, or job index */
  int ntile, ktile;         /* the 128 x 128 tile */
  int s0, s1;               /* slab range [s0, s1) of a cut tile's part (32 tokens a slab, both even) */
  int second;               /* the part with the later slabs */
  int slot;                 /* workspace slot of a cut tile; -1: whole tile */
} msr3d_wgrad_piece_t;
int msr3d_wgrad_stream(int n, const msr3d_wgrad_problem_t *problems, int n_pieces, const msr3d_wgrad_piece_t *pieces,
                       const int *wg_first, int n_wgs, int n_slots, const int *slot_piece, float *workspace,
                       long long workspace_floats, const msr3d_colsum_job_t *jobs, msr3d_stream_t stream);

/* Which tile kernel the three launches above run: 1 (default; MSR3D_WGRAD_PIPE=0 in the environment selects 0 at first
 * use) = eight waves that each load, split, stash and multiply, the next half-slab's fragment reads under the current
 * half-slab's MFMAs (round 6); 0 = rounds 4-5's eight loader + eight multiplier waves.  Same sums in the same order:
 * bit-identical results.  form = 0 / 1 selects, -1 only queries; returns the form in force (process-wide, takes effect at
 * the next launch), MSR3D_EINVAL for another value. */
int msr3d_wgrad_form(int form);

/* dW (n_out, k_in) (+)= dy^T x over M rows for TALL operands (the SharedMLP weight gradients of an unfrozen
 * backbone: up to ~10^6 rows), the arithmetic and tile kernel of msr3d_wgrad_split: the rows are cut into up to
 * MSR3D_WGRAD_ROWS_CHUNKS / tiles chunks, one workgroup per (chunk, 128 x 128 tile) stores its partial product to
 * workspace[chunk] (n_out * k_in floats each), and a second launch adds the partials in chunk order -- no atomics,
 * bit-reproducible.  Layers with n_out, k_in <= 64 take two chunks per workgroup, side by side in one tile (twice
 * the chunks, the same launch size).  accumulate != 0: dW holds the value to add to.  workspace_floats >= n_out * k_in (more
 * chunks, up to the cap, when there is room). */
#define MSR3D_WGRAD_ROWS_CHUNKS 256
int msr3d_wgrad_rows_split(int M, int n_out, int k_in, const float *dy, int ldy, const float *x, int ldx,
                           float *dW, int ldw, int accumulate, float *workspace, long long workspace_floats,
                           const float *x_bn, msr3d_stream_t stream);

/* C (M, N) = A (M, K) op(B)^T for TALL fp32 operands on the bf16 matrix pipe at fp32 accuracy (three exact bf16
 * terms per operand, six MFMA products per product: csrc/split_mma.h): the SharedMLP layers of an UNFROZEN
 * PointNet++ backbone as token GEMMs over the grouped rows (/root/reference/model/pointnet2/pytorch_utils.py:9-60;
 * hipops.py::_mlp_rows) -- forward z = t W^T (b_trans = 0, B = W (N, K)) and d t = d z W (b_trans = 1, B = W
 * (K, N): op(B)[n][k] = B[k * ldb + n]).  HBM-bound: the weight is split into LDS by each workgroup, the rows are
 * read once in MFMA fragment shape and split in registers.  B may be NARROWER than the product reads it (ldb < K
 * with b_trans = 0, ldb < N with b_trans = 1): it then ends at column ldb and counts as zero beyond -- the weight of
 * a layer whose input rows are zero-padded to whole 16-wide slabs needs no padded copy.  K % 4 == 0, K <= MSR3D_ROWS_GEMM_MAX_K,
 * N <= MSR3D_ROWS_GEMM_MAX_N, lda % 4 == 0, ldc % 4 == 0, A and C 16-byte aligned; no split-K, no atomics:
 * bit-reproducible, every output row independent of the others.
 * col_stats (optional): [ceil(M / MSR3D_ROWS_GEMM_BLOCK)][2][N] floats -- the column sums and sums of squares of C
 * over each block of MSR3D_ROWS_GEMM_BLOCK rows, summed in a fixed order: the first stage of the BatchNorm
 * statistics that follow the product (msr3d_bn_relu_train_fwd with partial_chunks).
 * a_bn (optional): [gamma | beta | mean | rstd], K floats each -- A holds the PRE-normalisation output of a
 * BatchNorm + ReLU layer and the product is taken of max(gamma (A - mean) rstd + beta, 0), formed on the way into the
 * matrix pipe: the normalised activation is never written or re-read (msr3d_wgrad_rows_split takes the same for x). */
#define MSR3D_ROWS_GEMM_BLOCK 256
#define MSR3D_ROWS_GEMM_MAX_K 1024
#define MSR3D_ROWS_GEMM_MAX_N 1024
int msr3d_rows_gemm_split(int M, int N, int K, const float *A, int lda, const float *B, int ldb, int b_trans,
                          float *C, int ldc, float *col_stats, const float *a_bn, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * The language-model side of the training step, first two pieces:
 * the per-sequence mean cross-entropy of /root/reference/model/msr3d/msr3d.py:426-441 and the
 * LoRA-augmented linear layer of the frozen-bf16 LLM (msr3d.py:103-112, peft LoraConfig r = 16).
 * dtype codes: 0 = f32, 1 = f16, 2 = bf16.
 * ------------------------------------------------------------------------- */

/* loss[b] = sum_{t < T-1, targets[b][t+1] >= 0} (logsumexp(logits[b][t]) - logits[b][t][targets[b][t+1]])
 *           / count[b],   count[b] = #{t: targets[b][t+1] >= 0}          (0 / 0 = NaN, as the reference).
 * logits (B,T,V) in `dtype`, read in place (the shift is an index, no fp32 copy); targets (B,T) int64.
 * Outputs: lse, tok_loss (B,T-1) f32 (saved for the backward), loss (B) f32, count (B) i32.  The
 * sequence sums run in index order: bit-reproducible.  V * sizeof(dtype) % 16 == 0. */
int msr3d_seq_ce_fwd(int B, int T, int V, const void *logits, int dtype, const long long *targets,
                     float *lse, float *tok_loss, float *loss, int *count, msr3d_stream_t stream);

/* dlogits (B,T,V) in `dtype`, fully written: (softmax - onehot) * grad_loss[b] / count[b] on the
 * supervised rows, zeros elsewhere (ignored labels and the last position of every sequence). */
int msr3d_seq_ce_bwd(int B, int T, int V, const void *logits, int dtype, const long long *targets,
                     const float *lse, const int *count, const float *grad_loss, void *dlogits,
                     msr3d_stream_t stream);

/* C (M,N) = scale * (P Q^T + P2 Q2^T): bf16 operands, all k-contiguous (P (M,K), Q (N,K), P2 (M,R),
 * Q2 (N,R); R = 0: no second pair), fp32 accumulate on v_mfma_f32_16x16x32_bf16, C bf16 (c_f32 = 0)
 * or f32.  The LoRA forward y = x W^T + (s x A^T) B^T and its dx = dy W + (s dy B) A are this call with
 * Q = W resp. W^T (the frozen weight is stored in both orientations).  K % 64 == 0, R % 8 == 0,
 * leading dimensions % 8 == 0, 16-byte aligned operands. */
int msr3d_bf16_gemm_lowrank(int M, int N, int K, int R, const void *P, int ldp, const void *Q, int ldq,
                            const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                            int c_f32, float scale, msr3d_stream_t stream);
/* C += the same product, C bf16 (the sum rounded to bf16 again) -- the wide-tile kernel's domain only (M >= 128, N >= 256,
 * R % 64 == 0, N % 4 == 0): the d-input of projections that read one tensor, as msr3d_fp8_gemm_lowrank_acc. */
int msr3d_bf16_gemm_lowrank_acc(int M, int N, int K, int R, const void *P, int ldp, const void *Q, int ldq,
                                const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc, float scale,
                                msr3d_stream_t stream);

/* LoRA weight gradients' token reduction: out (R, C) f32 (+)= scale * sum_m P[m][r] Q[m][c]
 * (transpose_out: out is (C, R)); P (M, R) and Q (M, C) bf16; R in {16, 32}; `out` holds the value to
 * add to (row splits meet by atomicAdd).  dA = s (dy B)^T x: P = dy B, Q = x;  dB = s dy^T (x A^T):
 * P = x A^T, Q = dy, transpose_out = 1. */
/* Batched C[o][i] (M, N) = scale * P[o][i] (M, K) Q[o][i] (N, K)^T over outer x inner problems (e.g.
 * sequences x heads; strides in elements: *_outer, *_inner, % 8 == 0 for P / Q, % 4 for C), bf16 operands with
 * k-contiguous rows (ld % 8 == 0), fp32 accumulate, C bf16 or fp32 (ldc % 4 == 0): the per-(sequence, head)
 * products of the decoder layer's attention (Q K^T, P V, and the four of its backward) on the same kernel as the
 * projections.  K % 64 == 0, outer * inner <= 65535. */
int msr3d_bf16_gemm_batched(int outer, int inner, int M, int N, int K, const void *P, int ldp, long long p_outer,
                            long long p_inner, const void *Q, int ldq, long long q_outer, long long q_inner, void *C,
                            int ldc, long long c_outer, long long c_inner, int c_f32, float scale,
                            msr3d_stream_t stream);

/* The row-local / element-wise kernels of one Llama decoder layer (csrc/llm_layer.hip; bf16 storage, fp32
 * arithmetic; transformers.models.llama.modeling_llama: LlamaRMSNorm, apply_rotary_pos_emb, eager attention,
 * LlamaMLP; /root/reference/model/msr3d/msr3d.py:409-415 runs them under bf16 autocast):
 *   rmsnorm_fwd   s = x (+ delta) [-> sum_out]; y = w * bf16(s * rsqrt(mean(s^2) + eps)); rstd (M) f32.
 *                 D in {512, 1024, 2048, 4096, 5120, 8192}.
 *   rmsnorm_bwd   dx = rstd * (dy w - xh mean(dy w xh)) (+ dres), xh = s rstd   (w is frozen: no dw)
 *   rope_inplace  x (B, T, H, D) bf16 *= rotation by (cos, sin) (T, D) f32; transpose != 0: the backward
 *   causal_softmax_fwd   probs (B H, T, T) bf16 = softmax over keys t' <= t with key_keep[b][t'] != 0 of fp32 scores
 *   causal_softmax_bwd   dscores bf16 = probs * (dprobs - rowsum(dprobs probs)), dprobs fp32
 *                        (both: T % 4 == 0, T <= 2048, 16-byte aligned fp32 rows)
 *   swiglu_fwd / bwd     h = silu(gate) * up and its two gradients; n % 8 == 0
 *   transpose_bf16       dst[o][i] (cols, rows) = src[o][i] (rows, cols)^T, two-level batch strides (elements) */
int msr3d_rmsnorm_fwd(int M, int D, const void *x, const void *delta, const void *w, float eps, void *sum_out,
                      void *y, float *rstd, msr3d_stream_t stream);
int msr3d_rmsnorm_bwd(int M, int D, const void *dy, const void *s, const void *w, const float *rstd,
                      const void *dres, void *dx, msr3d_stream_t stream);
int msr3d_rope_inplace(int B, int T, int H, int D, void *x, const float *cos_td, const float *sin_td, int transpose,
                       msr3d_stream_t stream);
/* the same rotation on TWO tensors of one shape (q and k of a decoder layer) in one launch */
int msr3d_rope_inplace2(int B, int T, int H, int D, void *x0, void *x1, const float *cos_td, const float *sin_td,
                        int transpose, msr3d_stream_t stream);
int msr3d_causal_softmax_fwd(int B, int H, int T, const float *scores, const unsigned char *key_keep, void *probs,
                             msr3d_stream_t stream);
int msr3d_causal_softmax_bwd(int B, int H, int T, const float *dprobs, const void *probs, void *dscores,
                             msr3d_stream_t stream);
int msr3d_swiglu_fwd(long long n, const void *gate, const void *up, void *out, msr3d_stream_t stream);
int msr3d_swiglu_bwd(long long n, const void *gate, const void *up, const void *dh, void *dgate, void *dup,
                     msr3d_stream_t stream);
int msr3d_transpose_bf16(int outer, int inner, int rows, int cols, const void *src, int ld_src, long long src_outer,
                         long long src_inner, void *dst, int ld_dst, long long dst_outer, long long dst_inner,
                         msr3d_stream_t stream);

/* Causal self-attention of a decoder layer, fused (csrc/llm_attn.hip; transformers' LlamaAttention as the LLM of
 * model/msr3d/msr3d.py:409-415 runs it: softmax(scale q k^T + causal + key padding) v per sequence and head).
 * q, k, v, out, dout, dq, dk, dv: (B, T, H, D) bf16 token-major, row stride ld >= H D (ld % 8 == 0), RoPE already
 * applied; D = 64 or 128; T % 64 == 0.  key_keep (B, T) bytes, 0 = padded key, or NULL.  lse, delta: (B, H, T) fp32 --
 * the rows' log2-sum-exp2 (forward output; +inf for a row with no visible key, whose output is 0) and rowsum(dout * out)
 * (written by the backward).  The scores stay in registers (online softmax forward; the backward recomputes the
 * probabilities from lse: one kernel for dq, one for dk / dv, every output row has one owner -- no atomics). */
int msr3d_attn_fwd(int B, int T, int H, int D, const void *q, const void *k, const void *v, int ld,
                   const unsigned char *key_keep, float scale, void *out, float *lse, msr3d_stream_t stream);
int msr3d_attn_bwd(int B, int T, int H, int D, const void *q, const void *k, const void *v, const void *out,
                   const void *dout, int ld, const unsigned char *key_keep, float scale, const float *lse, float *delta,
                   void *dq, void *dk, void *dv, msr3d_stream_t stream);

/* C (M, N) = scale * P Q^T for a Q of N <= 64 rows (N % 16 == 0; the LoRA down-projections x A^T and dy B), and
 * C[:, N:zero_to] = 0 (the padding the low-rank K step of msr3d_bf16_gemm_lowrank reads).  P (M, K), Q (N, K)
 * k-contiguous bf16, K % 32 == 0; C bf16.  (peft's lora_A / lora_B applications, model/msr3d/msr3d.py:103-112.) */
int msr3d_bf16_gemm_skinny(int M, int N, int K, const void *P, int ldp, const void *Q, int ldq, void *C, int ldc,
                           int zero_to, float scale, msr3d_stream_t stream);
/* ... and, in the same pass over P, its OCP e4m3 image with one scale per row -- q8 (M, K) bytes, row_scale (M) -- bit
 * for bit what msr3d_quant_rows_fp8 writes (the frozen-weight product msr3d_fp8_gemm_lowrank reads exactly this tensor
 * next: one launch and one read of P instead of two).  ldq8 >= K, ldq8 % 8 == 0, q8 8-byte aligned. */
int msr3d_bf16_gemm_skinny_quant(int M, int N, int K, const void *P, int ldp, const void *Q, int ldq, void *C, int ldc,
                                 int zero_to, float scale, void *q8, int ldq8, float *row_scale, msr3d_stream_t stream);

/* fp8 (OCP e4m3) operands for the frozen projections of the LoRA-Llama layers (model/msr3d/msr3d.py:103-112,409-415;
 * csrc/lora_fp8.hip).
 * msr3d_quant_rows_fp8: q[m][k] = rne_e4m3(x[m][k] / scale[m]), scale[m] = max_k |x[m][k]| / 448 (1 for a zero row);
 *   x (M, K) bf16, q (M, K) bytes; K % 8 == 0, K <= 12288.  Used per call for activations / upstream gradients and
 *   once per checkpoint for the frozen weight (a row = an output channel; both orientations).
 * msr3d_fp8_gemm_lowrank: C (M, N) bf16 = diag(sp) (Pq Qq^T) diag(sq) + P2 Q2^T with Pq (M, K), Qq (N, K) e4m3
 *   k-contiguous, their row scales sp (M), sq (N), and the LoRA pair P2 (M, 64), Q2 (N, 64) bf16 zero-padded (both
 *   NULL: no low-rank term).  K % 128 == 0, N % 4 == 0, M >= 128, N >= 256.  v_mfma_scale_f32_16x16x128_f8f6f4 with
 *   unit block scales, fp32 accumulate; the LoRA term on the bf16 instruction into the same accumulators. */
int msr3d_quant_rows_fp8(int M, int K, const void *x, int ldx, void *q, int ldq, float *scale, msr3d_stream_t stream);
int msr3d_fp8_gemm_lowrank(int M, int N, int K, const void *Pq, int ldp, const float *sp, const void *Qq, int ldq,
                           const float *sq, const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                           msr3d_stream_t stream);
/* C += the same product (C read as bf16, the sum rounded to bf16 again): the input gradients of projections that read
 * ONE tensor (q / k / v, gate / up) meet in one buffer instead of three tensors and two add launches. */
int msr3d_fp8_gemm_lowrank_acc(int M, int N, int K, const void *Pq, int ldp, const float *sp, const void *Qq, int ldq,
                               const float *sq, const void *P2, int ldp2, const void *Q2, int ldq2, void *C, int ldc,
                               msr3d_stream_t stream);

/* out (R, C) fp32 -- or its transpose (C, R) -- += scale * sum_m P[m][r] Q[m][c]: the LoRA weight gradients
 * dA = (s dy B)^T x and dB = dy^T (s x A^T) (peft's lora_A / lora_B, model/msr3d/msr3d.py:103-112; R = 16 or 32).
 * P (M, R), Q (M, C) bf16.  workspace (64 * R * C floats, 16-byte aligned; may be NULL): the row chunks' partial sums
 * are added up in a fixed order by a second launch -- bit-reproducible; without it 16 chunks meet by atomicAdd.
 * accumulate == 0 (workspace only): `out` is overwritten instead of added to -- no zero-fill by the caller. */
#define MSR3D_LORA_GRAD_CHUNKS 64
int msr3d_lora_grad(int M, int R, int C, const void *P, int ldp, const void *Q, int ldq, float *out,
                    int transpose_out, float scale, int accumulate, float *workspace, long long workspace_floats,
                    msr3d_stream_t stream);

/* dA and dB of one LoRA pair (or one of them: njobs = 1) in ONE launch: job z computes out_z (R, C_z) -- or its
 * transpose (C_z, R) -- (+)= scale * sum_m P_z[m][r] Q_z[m][c] as msr3d_lora_grad does, each workgroup owning 64 output
 * columns over ALL rows (no partial sums, no workspace, bit-reproducible by construction; 236 workgroups for a
 * 4096 / 11008 pair).  accumulate == 0: `out` is overwritten.  C % 8 == 0, ldp / ldq % 8 == 0, P / Q / out 16-byte
 * aligned.  (peft's lora_A / lora_B gradients, model/msr3d/msr3d.py:103-112.) */
typedef struct {
  int C;
  const void *P; int ldp;      /* (M, R) bf16 */
  const void *Q; int ldq;      /* (M, C) bf16 */
  float *out; int transpose_out;
} msr3d_lora_grad_job_t;
int msr3d_lora_grad_pair(int M, int R, int njobs, const msr3d_lora_grad_job_t *jobs, float scale, int accumulate,
                         msr3d_stream_t stream);

/* The bf16 images of the LoRA pairs of a whole stack, in the orientations the products read, in ONE launch (once per
 * optimiser step): per job  a_pad (r, K) = A,  at2 (K, 64)[:, :r] = A^T,  b2 (N, 64)[:, :r] = B,  bt_pad (r, N) = B^T
 * (A (r, K), B (N, r) fp32; r = 16 or 32; columns r..63 of at2 / b2 are not written: the caller zeroes them once).
 * `jobs_device`: the table in DEVICE memory.  (peft LoraConfig targets, model/msr3d/msr3d.py:103-112.) */
typedef struct {
  const float *A, *B;
  unsigned short *a_pad, *b2, *bt_pad, *at2;
  int r, K, N, pad_;
} msr3d_lora_shadow_job_t;
int msr3d_lora_shadows(int njobs, const msr3d_lora_shadow_job_t *jobs_device, msr3d_stream_t stream);


/* ---------------------------------------------------------------------------
 * Optimiser step of the hot path: global-norm clip + AdamW over flat buffers
 * (/root/reference/optim/build.py:7-17, trainer/leo_trainer.py:189-195,
 * optim/scheduler.py:17-25).  All buffers hold n floats (n % 4 == 0, 16-byte aligned).
 * sumsq_scratch (MSR3D_ADAMW_SCRATCH_FLOATS floats: per-block partial sums, reduced in a fixed
 * order so every data-parallel rank derives the identical clip coefficient) and step_counter
 * (1 int, zero-initialised once, incremented by the call) are device memory owned by the caller.
 * schedule: 0 = constant lr, 1 = warmup_cosine_instructblip(warmup_steps, total_steps).
 * max_grad_norm <= 0 disables clipping; zero_grad != 0 clears grads after use.
 * ------------------------------------------------------------------------- */
#define MSR3D_ADAMW_SCRATCH_FLOATS 1024
int msr3d_adamw_flat(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                     float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                     float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                     int warmup_steps, int total_steps, int zero_grad, msr3d_stream_t stream);
/* The same with a mask: active4[i] == 0 leaves floats 4 i .. 4 i + 3 of params / exp_avg / exp_avg_sq
 * untouched (no weight decay either) -- a parameter that receives no gradient, which torch.optim.AdamW
 * skips (`p.grad is None`; the reference trains with find_unused_parameters=True,
 * trainer/leo_trainer.py:50-52).  NULL = every element active.  n / 4 bytes of device memory. */
int msr3d_adamw_flat_masked(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                            float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                            float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                            int warmup_steps, int total_steps, int zero_grad, const unsigned char *active4,
                            msr3d_stream_t stream);
/* The same reading every gradient as grads[i] * grad_scale (the rounded fp32 product, in the norm and in
 * the update -- bit-identical to a separate `grads *= grad_scale` pass before the call).  The data-parallel
 * engine leaves the all-reduced SUM in the buffer and passes 1 / world here: the averaging
 * (/root/reference/trainer/leo_trainer.py:50-52, DDP's gradient mean) costs no extra pass over 21 MB. */
int msr3d_adamw_flat_scaled(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                            float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                            float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                            int warmup_steps, int total_steps, int zero_grad, const unsigned char *active4,
                            float grad_scale, msr3d_stream_t stream);

/* ---------------------------------------------------------------------------
 * Scene-sample construction on the device: replaces the host-side
 * per-instance masks of /root/reference/data/datasets/scannet_base.py:57-67 (and
 * scan_data_loader.py:83-94), `MSR3DBase.preprocess_pcd` (data/datasets/msr3d.py:181-216) and the
 * wrapper's padding (data/datasets/dataset_wrapper.py:141-158).
 *
 * msr3d_segment_scan -- once per scan.  instance_labels (n_points) int64 as stored in
 * <scan>.pth; slot_of_label (n_labels) int32 maps a label to its object slot (-1 / out of range:
 * point dropped), n_slots <= MSR3D_SEG_MAX_SLOTS.  Writes points_sorted (.,3) f32 and
 * colors_sorted (.,3) u8 so that slot s owns the contiguous rows
 * [inst_offsets[s], inst_offsets[s+1]) in ascending original order (== pcds[labels == id]);
 * order (optional) receives the original row of every sorted row; inst_offsets has n_slots+1
 * ints.  workspace: ceil(n_points / MSR3D_SEG_CHUNK) * n_slots ints.
 *
 * msr3d_preprocess_pcd -- once per batch, grid (O, B).  Object slot (b,o) is the rows
 * [obj_begin, obj_begin + obj_count) of points/colors (any arena of sorted scans); obj_count <= 0
 * marks a padding slot (obj_fts = 1.0, obj_locs = 0, obj_masks = 0).  rot (B,9) row-major f32
 * scene rotation or NULL; pcd_idxs (B,O,P) int32 = the subsample drawn by the caller, or NULL
 * to draw on the device from `seed` (without replacement when obj_count >= P, with replacement
 * otherwise, like np.random.choice at msr3d.py:200-201); idx_out (B,O,P) optional, receives the
 * indices used.  Outputs: obj_fts (B,O,P,6) f32 [xyz centred on the subsample mean and scaled
 * by its largest norm (1 if < 1e-6); rgb = c/127.5-1], obj_locs (B,O,6) f32 [centre, box size
 * of the whole rotated object], obj_masks (B,O) bytes.  float64 arithmetic like the reference;
 * P even, <= 4096.
 * ------------------------------------------------------------------------- */
#define MSR3D_SEG_CHUNK 256
#define MSR3D_SEG_MAX_SLOTS 8192
int msr3d_segment_scan(int n_points, const long long *instance_labels, const int *slot_of_label,
                       int n_labels, int n_slots, const float *points, const unsigned char *colors,
                       float *points_sorted, unsigned char *colors_sorted, int *order,
                       int *inst_offsets, int *workspace, msr3d_stream_t stream);
int msr3d_preprocess_pcd(int B, int O, int P, const float *points, const unsigned char *colors,
                         const long long *obj_begin, const int *obj_count, const float *rot,
                         const int *pcd_idxs, unsigned long long seed, float *obj_fts,
                         float *obj_locs, unsigned char *obj_masks, int *idx_out,
                         msr3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSR3D_HIP_H */
