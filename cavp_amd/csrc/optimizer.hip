// Fused multi-tensor optimiser step for gfx950: torch.optim.SGD(momentum, weight_decay) on the visual parameter groups and
// torch.optim.Adam on the audio encoder (reference main_vpo_mono.py:118-125), every parameter tensor of the model in ONE
// launch.  The job table (one entry per tensor: parameter, gradient view in the flat gradient arena, state buffers,
// group hyper-parameters) lives in device memory and is built once; per-step scalars (learning rate of the poly
// schedule, Adam bias corrections) are kernel arguments, so the whole step stays hipGraph-capturable.
//
//   SGD  (torch.optim.SGD, dampening 0, nesterov False):  d = g + wd p;  buf = first ? d : mu buf + d;  p -= lr buf
//   Adam (torch.optim.Adam, amsgrad False, L2 weight decay):  d = g + wd p;  m = b1 m + (1-b1) d;  v = b2 v + (1-b2) d^2;
//        p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// HBM-bound streaming: 3 reads + 2 writes (SGD) / 4 reads + 3 writes (Adam) of 4 bytes per parameter.
#include "common.h"

namespace {

constexpr int kElemsPerBlock = 4096;   // 256 threads x 4 float4

__global__ __launch_bounds__(256) void optimizer_step_kernel(const cavp_opt_job* __restrict__ jobs, int njobs, float lr_sgd,
                                                             float lr_adam, float momentum, float beta1, float beta2,
                                                             float eps, float bc1, float bc2_sqrt, int first_step) {
  // binary search: last job with blk0 <= blockIdx.x (uniform per workgroup)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const cavp_opt_job J = jobs[lo];
  const long long base = (long long)(blockIdx.x - J.blk0) * kElemsPerBlock;
  const float wd = J.weight_decay;
  if (J.kind == 0) {
    const float lr = lr_sgd * J.lr_mult;
    for (int q = 0; q < 4; ++q) {
      const long long i = base + (q * 256 + threadIdx.x) * 4ll;
      if (i >= J.n) break;
      if (i + 4 <= J.n && J.vec) {
        float4 p = *(const float4*)(J.p + i);
        const float4 g = *(const float4*)(J.g + i);
        float4 b = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(J.m + i);
        const float d0 = g.x + wd * p.x, d1 = g.y + wd * p.y, d2 = g.z + wd * p.z, d3 = g.w + wd * p.w;
        b.x = first_step ? d0 : momentum * b.x + d0; b.y = first_step ? d1 : momentum * b.y + d1;
        b.z = first_step ? d2 : momentum * b.z + d2; b.w = first_step ? d3 : momentum * b.w + d3;
        p.x -= lr * b.x; p.y -= lr * b.y; p.z -= lr * b.z; p.w -= lr * b.w;
        *(float4*)(J.m + i) = b;
        *(float4*)(J.p + i) = p;
      } else {
        for (long long e = i; e < i + 4 && e < J.n; ++e) {
          const float d = J.g[e] + wd * J.p[e];
          const float b = first_step ? d : momentum * J.m[e] + d;
          J.m[e] = b;
          J.p[e] -= lr * b;
        }
      }
    }
  } else {
    const float step = lr_adam * J.lr_mult / bc1;
    for (int q = 0; q < 4; ++q) {
      const long long i = base + (q * 256 + threadIdx.x) * 4ll;
      if (i >= J.n) break;
      for (long long e = i; e < i + 4 && e < J.n; ++e) {   // (the compiler vectorises the aligned case)
        const float d = J.g[e] + wd * J.p[e];
        const float m = beta1 * J.m[e] + (1.f - beta1) * d;
        const float v = beta2 * J.v[e] + (1.f - beta2) * d * d;
        J.m[e] = m;
        J.v[e] = v;
        J.p[e] -= step * m / (sqrtf(v) / bc2_sqrt + eps);
      }
    }
  }
}

}  // namespace

extern "C" int32_t cavp_optimizer_blocks(int64_t n) { return (int32_t)((n + kElemsPerBlock - 1) / kElemsPerBlock); }

extern "C" int cavp_optimizer_step(const cavp_opt_job* jobs_device, int32_t njobs, int32_t total_blocks, float lr_sgd,
                                   float lr_adam, float momentum, float beta1, float beta2, float eps, int64_t step,
                                   void* stream) {
  if (!jobs_device || njobs <= 0 || total_blocks <= 0 || step < 1) return CAVP_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  optimizer_step_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(jobs_device, njobs, lr_sgd, lr_adam, momentum, beta1,
                                                                      beta2, eps, (float)bc1, (float)sqrt(bc2), step == 1);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}
