// optim_flat.hip -- global-norm clipping + AdamW over ONE flat parameter / gradient / state
// buffer: two launches per step instead of the ~300 tiny kernels torch's capturable foreach
// AdamW issues for this model's 52 tensors (bias-correction math on 0-dim step tensors runs
// one elementwise kernel per parameter: 156 divisions per step in profiles/r01 traces).
//
// Semantics follow the reference's optimiser stack exactly
// (/root/reference/optim/build.py:7-17 -> torch.optim.AdamW, one param group, lr 3e-5,
// betas (0.9, 0.999), weight_decay 0.05, configs/msr3d.yaml:43-47;
// accelerator.clip_grad_norm_(5.0), trainer/leo_trainer.py:192-193;
// LambdaLR(warmup_cosine_instructblip), optim/scheduler.py:17-20,23-25):
//   coef = min(1, max_norm / (||g||_2 + 1e-6));  g *= coef
//   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v)/sqrt(1 - b2^t) + eps)
// The step counter and the squared norm live on the device so the pair can be replayed
// from a HIP graph.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/msr3d_hip.h"

namespace {

constexpr int kMaxBlocks = 1024;   // == MSR3D_ADAMW_SCRATCH_FLOATS

// Deterministic global norm: every block writes ITS partial sum of squares to partial[block]
// (no atomics); adamw_kernel then adds the partials in a fixed order.  All data-parallel
// ranks hold identical gradients after the all-reduce and must derive the identical clip
// coefficient, or their weights drift apart.
__global__ __launch_bounds__(256) void sumsq_kernel(long long n4, const float4 *__restrict__ g,
                                                    float *__restrict__ partial, int *__restrict__ step_ctr,
                                                    float gscale) {
  // (also advances the step counter: it runs before adamw_kernel, which then reads the NEW value --
  // one launch fewer than a separate tick)
  if (step_ctr && blockIdx.x == 0 && threadIdx.x == 0) *step_ctr += 1;
  float s = 0.f;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    float4 v = g[t];
    v.x *= gscale; v.y *= gscale; v.z *= gscale; v.w *= gscale;      // (the rounded product, as a separate g *= s leaves it)
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[0] = sum a[i] b[i] in ONE launch, bit-reproducible: per-block partials travel through device-
// coherent stores, the last block to arrive (ticket) adds them in block order.  scratch: kMaxBlocks
// floats + one int counter (left at zero for the next call).
constexpr int kDotThreads = 1024;  // 16 waves a CU on ONE block per CU: the cost that grows with the block count is the
                                   // agent-scope release (an L2 write-back per block) -- 1024 blocks of 256: 40 us, 256: 12 us
__global__ __launch_bounds__(kDotThreads) void dot_kernel(long long n4, const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                  float *__restrict__ partial, int *__restrict__ counter,
                                                  float *__restrict__ out) {
  float s = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  for (; t < n4; t += 4 * stride) {                        // eight 16-byte loads in flight per thread; an index past
    float4 x[4], y[4];                                     // the end reads the last element and contributes zero
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = t + u * stride;
      ok[u] = i < n4;
      const long long j = ok[u] ? i : n4 - 1;
      x[u] = a[j]; y[u] = b[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float p = 0.f;
      p = fmaf(x[u].x, y[u].x, p); p = fmaf(x[u].y, y[u].y, p); p = fmaf(x[u].z, y[u].z, p); p = fmaf(x[u].w, y[u].w, p);
      s += ok[u] ? p : 0.f;
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  constexpr int NWAVE = kDotThreads / 64;
  __shared__ float part[NWAVE];
  __shared__ int ticket;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float blk = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) blk += part[w];          // fixed order
    // The partial must be visible to the LAST block, possibly on another XCD, before the ticket is.  An agent-scope
    // release fence does that by writing the XCD's whole L2 back (buffer_wbl2): 23-27 k cycles a block
    // (profiles/r06_last_arriver_probe.txt), most of this kernel's 12 us.  Instead the partial itself travels by a
    // RETURNING read-modify-write -- device-scope atomics execute at the memory side of the fabric on this part
    // (TCC_EA0_ATOMIC == TCC_ATOMIC), and its return value says it has -- the ticket is taken only then, and the last
    // block reads the partials back the same way (an atomic OR of 0): no cache is written back or invalidated.
    const int prev = __hip_atomic_exchange(reinterpret_cast<int *>(partial) + blockIdx.x, __float_as_int(blk), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(prev) : "memory");
    ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (ticket != (int)gridDim.x - 1) return;
  float tot = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kDotThreads)
    tot += __int_as_float(__hip_atomic_fetch_or(reinterpret_cast<int *>(partial) + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float all = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) all += part[w];
    out[0] = all;
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// `sched` = schedule id | (scheduler steps per optimiser update << 8): accelerate's
// AcceleratedScheduler steps the LambdaLR num_processes times per update, so with N ranks the
// reference's lambda sees step * N (0 in the high bits = 1).
__device__ __forceinline__ float lr_lambda(int sched, int step, int warmup, int total) {
  const int mult = (sched >> 8) > 0 ? (sched >> 8) : 1;
  sched &= 0xff;
  step *= mult;
  if (sched == 1) {   // warmup_cosine_instructblip (optim/scheduler.py:17-20)
    if (step <= warmup) return 1e-3f + (float)step / (float)warmup * (1.0f - 1e-3f);
    const double x = (double)(step - warmup) / (double)(total - warmup) * 3.14159265358979323846;
    return (float)(0.5 * (1.0 + cos(x)));
  }
  return 1.0f;
}

__global__ __launch_bounds__(256) void adamw_kernel(long long n4, float4 *__restrict__ p,
                                                    float4 *__restrict__ g, float4 *__restrict__ m,
                                                    float4 *__restrict__ v,
                                                    const float *__restrict__ partial,
                                                    int n_partial,
                                                    int *__restrict__ step_ctr, float base_lr,
                                                    float beta1, float beta2, float eps, float wd,
                                                    float max_norm, int sched, int warmup, int total,
                                                    int zero_grad, const unsigned char *__restrict__ active,
                                                    int ticked, float gscale) {
  __shared__ float sh[4];
  __shared__ float red[256];
  if (max_norm > 0.f) {        // fixed-order sum of the block partials (same in every block)
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    const int t = *step_ctr + (ticked ? 0 : 1);          // this update's 1-based index
    const float lr = base_lr * lr_lambda(sched, t - 1, warmup, total);   // LambdaLR: lambda(t-1)
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    float coef = 1.0f;
    if (max_norm > 0.f) {
      const float nrm = sqrtf(red[0]);
      coef = fminf(max_norm / (nrm + 1e-6f), 1.0f);
    }
    sh[0] = lr;
    sh[1] = (float)((double)lr / bc1);                   // step_size
    sh[2] = (float)sqrt(bc2);                            // bias_correction2_sqrt
    sh[3] = coef;
  }
  __syncthreads();
  const float lr = sh[0], step_size = sh[1], bc2s = sh[2], coef = sh[3];
  const float decay = 1.0f - lr * wd, w1 = 1.0f - beta1, w2 = 1.0f - beta2;
  // (measured and not kept, round 6: four float4 of every buffer in flight per thread before the first update -- 16
  //  concurrent streams a thread -- 48-55 us against 38-43 for this one-at-a-time loop, same box)
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4;
       t += (long long)gridDim.x * blockDim.x) {
    if (active && !active[t]) {      // a parameter that receives no gradient: untouched, as torch.optim.AdamW
      if (zero_grad) g[t] = make_float4(0.f, 0.f, 0.f, 0.f);      // leaves one whose .grad is None
      continue;
    }
    float4 pp = p[t], gg = g[t], mm = m[t], vv = v[t];
#define UPD(c)                                                  \
    {                                                           \
      const float gr = (gg.c * gscale) * coef;                  \
      float pv = pp.c * decay;                                  \
      mm.c = mm.c + w1 * (gr - mm.c);                           \
      vv.c = vv.c * beta2 + w2 * gr * gr;                       \
      const float denom = sqrtf(vv.c) / bc2s + eps;             \
      pp.c = pv - step_size * (mm.c / denom);                   \
    }
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    p[t] = pp; m[t] = mm; v[t] = vv;
    if (zero_grad) g[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// runs after adamw_kernel on the same stream: bump the counter
__global__ void adamw_tick_kernel(int *step_ctr) { *step_ctr += 1; }

}  // namespace

extern "C" {

int msr3d_adamw_flat(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                     float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                     float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                     int warmup_steps, int total_steps, int zero_grad, msr3d_stream_t stream) {
  return msr3d_adamw_flat_masked(n, params, grads, exp_avg, exp_avg_sq, sumsq_scratch, step_counter, base_lr, beta1, beta2,
                                 eps, weight_decay, max_grad_norm, schedule, warmup_steps, total_steps, zero_grad, nullptr,
                                 stream);
}

int msr3d_adamw_flat_masked(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                            float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                            float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                            int warmup_steps, int total_steps, int zero_grad, const unsigned char *active4,
                            msr3d_stream_t stream) {
  return msr3d_adamw_flat_scaled(n, params, grads, exp_avg, exp_avg_sq, sumsq_scratch, step_counter, base_lr, beta1, beta2,
                                 eps, weight_decay, max_grad_norm, schedule, warmup_steps, total_steps, zero_grad, active4,
                                 1.0f, stream);
}

int msr3d_adamw_flat_scaled(long long n, float *params, float *grads, float *exp_avg, float *exp_avg_sq,
                            float *sumsq_scratch, int *step_counter, float base_lr, float beta1,
                            float beta2, float eps, float weight_decay, float max_grad_norm, int schedule,
                            int warmup_steps, int total_steps, int zero_grad, const unsigned char *active4,
                            float grad_scale, msr3d_stream_t stream) {
  if (n < 0 || (n % 4) != 0) return MSR3D_EINVAL;
  if (n == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !sumsq_scratch || !step_counter)
    return MSR3D_EINVAL;
  if ((schedule & 0xff) == 1 && !(total_steps > warmup_steps && warmup_steps >= 1)) return MSR3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long n4 = n / 4;
  long long gsz = (n4 + 255) / 256;
  if (gsz > kMaxBlocks) gsz = kMaxBlocks;
  const int ticked = max_grad_norm > 0.f ? 1 : 0;
  if (ticked)
    sumsq_kernel<<<(int)gsz, 256, 0, st>>>(n4, reinterpret_cast<const float4 *>(grads), sumsq_scratch, step_counter,
                                              grad_scale);
  adamw_kernel<<<(int)gsz, 256, 0, st>>>(
      n4, reinterpret_cast<float4 *>(params), reinterpret_cast<float4 *>(grads),
      reinterpret_cast<float4 *>(exp_avg), reinterpret_cast<float4 *>(exp_avg_sq), sumsq_scratch,
      (int)gsz, step_counter, base_lr, beta1, beta2, eps, weight_decay, max_grad_norm, schedule, warmup_steps,
      total_steps, zero_grad, active4, ticked, grad_scale);
  if (!ticked) adamw_tick_kernel<<<1, 1, 0, st>>>(step_counter);
  return (int)hipGetLastError();
}

int msr3d_dot_f32(long long n, const float *a, const float *b, float *scratch, float *out, msr3d_stream_t stream) {
  if (n < 0 || (n % 4) != 0) return MSR3D_EINVAL;
  if (!a || !b || !scratch || !out) return MSR3D_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return MSR3D_EINVAL;
  const long long n4 = n / 4;
  long long gsz = (n4 + kDotThreads - 1) / kDotThreads;
  if (gsz > 256) gsz = 256;                 // one block per CU (see kDotThreads); the last block adds the partials
  if (gsz < 1) gsz = 1;
  dot_kernel<<<(int)gsz, kDotThreads, 0, (hipStream_t)stream>>>(n4, reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b),
                                                        scratch, reinterpret_cast<int *>(scratch + kMaxBlocks), out);
  return (int)hipGetLastError();
}

}  // extern "C"
