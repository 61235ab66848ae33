// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds b16 value == its element index; lane l passes byte
// address addr[l]; we dump the 4 b16 each lane receives.  Build: hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void tr_probe_kernel(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  s16x4 v;
  unsigned ldsaddr = (unsigned)(uintptr_t)lds + a;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ldsaddr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int tr_probe(const int* addr, unsigned short* out, void* stream) {
  tr_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(addr, out);
  return (int)hipGetLastError();
}
