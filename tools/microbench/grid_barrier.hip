// Review item 3 of round 5 ("one persistent launch per residual stage for layer3 / layer4 / ASPP"): what does a conv -> conv seam cost
// INSIDE one launch (device-scope barrier between the stages) against the same seam as a kernel boundary of a replayed hipGraph?
//
// A stage = every workgroup produces `bytes_per_wg` of a shared tensor (16-byte stores) and, in the next stage, reads the slice another
// workgroup (on another XCD: blockIdx + 1 lands there) produced - the dependency pattern of a 1x1 conv chain on the 14 x 14 layers
// (12.8 MB activations at B = 32: 6272 pixels x 1024 channels, bf16).  Every word is checked, so a barrier that is not a barrier shows.
//   chain-graph : S launches of the stage kernel captured in a hipGraph (what the product does today)
//   chain-flat  : ONE launch, S stages, one monotonic device-scope counter (lane-0 release fence -> atomic arrive -> relaxed poll -> acquire)
//   chain-xcd   : ONE launch, S stages, XCD-hierarchical barrier (per-XCD counters, leaders meet on a top counter)
// Grids: 256 / 512 / 1024 workgroups of 256 threads (1 / 2 / 4 per CU).   Prints microseconds per stage.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bar {
  unsigned top;            // arrivals of XCD leaders (xcd) / of every workgroup (flat), monotonic
  unsigned pad0[31];
  unsigned xcd_cnt[8][32]; // arrivals per XCD, monotonic (one 128-byte line each)
  unsigned xcd_gen[8][32]; // generation published by the XCD's last arriver
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat barrier: every workgroup arrives on one counter
__device__ __forceinline__ void barrier_flat(Bar* b, unsigned nwg, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (ld_relaxed(&b->top) < nwg * gen) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// hierarchical: workgroups of an XCD (blockIdx % 8 under round-robin dispatch) arrive on their XCD's counter; its last arriver goes to the top
// counter, waits for the 8 leaders and publishes the generation to its XCD
__device__ __forceinline__ void barrier_xcd(Bar* b, unsigned nwg, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7, per = nwg >> 3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&b->xcd_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == per * gen - 1) {   // last of this XCD for this generation
      __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_relaxed(&b->top) < 8u * gen) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(&b->xcd_gen[x][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (ld_relaxed(&b->xcd_gen[x][0]) < gen) __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// one stage: read the neighbour's slice of `src` (check it), write my slice of `dst`
__device__ __forceinline__ void stage_body(const uint4* src, uint4* dst, int vec_per_wg, unsigned stage, unsigned* err) {
  const unsigned nwg = gridDim.x, me = blockIdx.x, nb = (me + 1) % nwg;
  unsigned bad = 0;
  uint4 acc = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < vec_per_wg; i += 256) {
    const uint4 v = src[(size_t)nb * vec_per_wg + i];
    const unsigned want = stage == 0 ? 0u : (stage - 1) * 1000003u + nb * 4099u + (unsigned)i;
    if (stage > 0 && (v.x != want || v.w != want + 3)) ++bad;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (int i = threadIdx.x; i < vec_per_wg; i += 256) {
    const unsigned t = stage * 1000003u + me * 4099u + (unsigned)i;
    dst[(size_t)me * vec_per_wg + i] = make_uint4(t, t + 1, t + 2 + (acc.x == 0xdeadbeefu), t + 3);
  }
  if (bad) atomicAdd(err, bad);
}

__global__ __launch_bounds__(256) void stage_kernel(const uint4* src, uint4* dst, int vec_per_wg, unsigned stage, unsigned* err) {
  stage_body(src, dst, vec_per_wg, stage, err);
}

template <int MODE>
__global__ __launch_bounds__(256) void chain_kernel(uint4* a, uint4* b, int vec_per_wg, int stages, Bar* bar, unsigned gen0, unsigned* err) {
  for (int s = 0; s < stages; ++s) {
    stage_body((s & 1) ? b : a, (s & 1) ? a : b, vec_per_wg, (unsigned)s, err);
    if (s + 1 < stages) {
      if (MODE == 0) barrier_flat(bar, gridDim.x, gen0 + s + 1);
      else barrier_xcd(bar, gridDim.x, gen0 + s + 1);
    }
  }
}

int main() {
  const int S = 16;
  uint4 *a, *b;
  const size_t cap = (size_t)64 << 20;
  CK(hipMalloc(&a, cap)); CK(hipMalloc(&b, cap));
  Bar* bar; CK(hipMalloc(&bar, sizeof(Bar)));
  unsigned* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%6s %10s %14s %14s %14s   (us per stage, %d-stage chain; stage = read neighbour's slice + write own)\n", "WGs", "KB/WG", "graph-launches", "flat-barrier", "xcd-barrier", S);
  const int grids[] = {256, 512, 1024};
  const int kbs[] = {0, 16, 48};   // 0 = one vector per workgroup (pure synchronisation); 48 KB x 256 = the 12.8 MB layer3 tensor
  for (int g : grids)
    for (int kb : kbs) {
      const int kb_eff = kb * 256 / g;   // same TENSOR size at every grid
      const int vec = kb == 0 ? 16 : kb_eff * 1024 / 16;
      float us[3];
      // graph of S launches
      {
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int s = 0; s < S; ++s) stage_kernel<<<g, 256, 0, st>>>((s & 1) ? b : a, (s & 1) ? a : b, vec, (unsigned)s, err);
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e0, st));
        const int reps = 20;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[0] = ms * 1e3f / reps / S;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
      }
      for (int mode = 0; mode < 2; ++mode) {
        CK(hipMemsetAsync(bar, 0, sizeof(Bar), st));
        unsigned gen0 = 0;
        const int reps = 20;
        for (int r = -3; r < reps; ++r) {
          if (r == 0) CK(hipEventRecord(e0, st));
          if (mode == 0) chain_kernel<0><<<g, 256, 0, st>>>(a, b, vec, S, bar, gen0, err);
          else chain_kernel<1><<<g, 256, 0, st>>>(a, b, vec, S, bar, gen0, err);
          gen0 += S - 1;
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[1 + mode] = ms * 1e3f / reps / S;
      }
      unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("%6d %10.1f %14.2f %14.2f %14.2f   stale words: %u\n", g, kb == 0 ? 0.25 : (double)kb_eff, us[0], us[1], us[2], herr);
      CK(hipMemset(err, 0, 4));
    }
  return 0;
}
