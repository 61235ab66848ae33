// How fast are device-scope f32 atomics when many workgroups add into the SAME small array (the per-channel sums of the
// BatchNorm reductions: 2 x C floats), and does spreading them over R replicas at distant addresses help?
// Each of G workgroups (256 threads) issues NA atomicAdd's per thread into replica (block % R) of a [C] array; replicas are
// `stride` floats apart.  Prints microseconds per launch.   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_spread.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_atomic(float* out, int C, int R, size_t stride, int NA) {
  float* dst = out + (size_t)(blockIdx.x % R) * stride;
  for (int i = 0; i < NA; ++i) atomicAdd(dst + ((threadIdx.x + 256 * i) % C), 1.0f);
}
// reference: the same number of plain (non-atomic) stores
__global__ __launch_bounds__(256) void k_store(float* out, int C, int R, size_t stride, int NA) {
  float* dst = out + (size_t)blockIdx.x * 4096;
  for (int i = 0; i < NA; ++i) dst[(threadIdx.x + 256 * i) % C] = 1.0f;
}

int main() {
  float* buf;
  const size_t bytes = (size_t)1 << 30;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int Gs[] = {392, 784, 2048, 8192};
  const int Cs[] = {128, 512, 2048};
  const int Rs[] = {1, 4, 16, 64, 256};
  const size_t strides[] = {4096 / 4, 65536 / 4, (1 << 20) / 4};
  printf("%6s %6s %5s %9s %4s %10s %12s\n", "G", "C", "R", "stride_B", "NA", "us", "Gatomics/s");
  for (int G : Gs)
    for (int C : Cs)
      for (int R : Rs)
        for (size_t st : strides) {
          if (R == 1 && st != strides[0]) continue;
          const int NA = (2 * C + 255) / 256;   // 2 sums per channel, as the BN reductions
          for (int w = 0; w < 3; ++w) k_atomic<<<G, 256>>>(buf, 2 * C, R, st, NA);
          hipEventRecord(e0);
          const int reps = 20;
          for (int r = 0; r < reps; ++r) k_atomic<<<G, 256>>>(buf, 2 * C, R, st, NA);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          const double us = ms * 1e3 / reps;
          const double n = (double)G * 256 * NA;
          printf("%6d %6d %5d %9zu %4d %10.2f %12.2f\n", G, C, R, st * 4, NA, us, n / us * 1e-3);
        }
  for (int G : Gs) {
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k_store<<<G, 256>>>(buf, 1024, 1, 0, 4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("plain stores G=%d: %.2f us\n", G, ms * 1e3 / 20);
  }
  return 0;
}
