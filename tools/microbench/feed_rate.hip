// What feeds a CU fastest on MI355X?  Every MFMA kernel of this repo (igemm, weight gradients) is bound by the rate at which
// operand tiles reach the CU (profiles/r01_notes.md: ~60 GB/s per CU through `buffer_load ... lds`), so this measures that
// rate in isolation, per CU, for the candidate paths:
//   dma      buffer_load_dwordx4 ... lds (LDS-DMA), 2-stage ring, counted vmcnt + barrier per 16 KiB tile
//   dma_oob  the same instructions with every lane out of range (zero fill: pure issue / LDS-write cost)
//   reg      raw_buffer_load_b128 into VGPRs (xor-reduced), 8 loads in flight per lane
//   regst    raw_buffer_load_b128 -> VGPR -> ds_write_b128 (register staging), barrier per tile
//   mfma     32 x v_mfma_f32_16x16x32_bf16 per wave and tile (the MFMA work of a 128x128x64 K step), no loads
//   dma+mfma both
// for 1 / 2 / 4 resident 256-thread workgroups per CU, sources that live in the XCD's L2 (2 MiB per XCD), in L1 (every
// workgroup re-reads one 16 KiB tile) or stream from HBM (1 GiB), with rows that are contiguous (128-byte row stride: a wave
// instruction reads 1 KiB of consecutive memory) or strided like an NHWC tensor / weight matrix (row stride 2 KiB).
//   hipcc --offload-arch=gfx950 -O3 feed_rate.hip -o feed_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

enum { M_DMA = 0, M_DMA_OOB = 1, M_REG = 2, M_REGST = 3, M_MFMA = 4, M_DMA_MFMA = 5 };

template <int MODE>
__global__ __launch_bounds__(256) void feed(const char* src, unsigned region_bytes, unsigned row_stride, int iters, int same_tile,
                                            unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)xcd * region_bytes), 0, region_bytes, 0x00020000);
  // a 16 KiB tile = 128 rows x 128 B; wave-instruction i of wave w covers rows 8 (w + 4 i) .. + 7, lane l -> row + (l >> 3), 16-byte slot l & 7
  const unsigned kpr = row_stride / 128;   // K tiles per row
  unsigned lane_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lane_off[i] = (unsigned)(8 * (wave + 4 * i) + (lane >> 3)) * row_stride + (unsigned)(lane & 7) * 16;
  const unsigned tiles_in_region = region_bytes / 16384;
  unsigned t = same_tile ? 0u : ((blockIdx.x >> 3) * 37u) % tiles_in_region;
  auto tile_off = [&](unsigned tt) { return (tt % kpr) * 128u + (tt / kpr) * (128u * row_stride); };
  u32x4_t acc = {0u, 0u, 0u, 0u};
  f32x4_t macc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) macc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa = __builtin_bit_cast(bf16x8_t, (u32x4_t){(unsigned)lane, 1u, 2u, 3u});
  bf16x8_t fb = __builtin_bit_cast(bf16x8_t, (u32x4_t){4u, (unsigned)wave, 6u, 7u});
  auto mfma32 = [&]() {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, macc[i], 0, 0, 0);
  };
  auto dma = [&](int buf, unsigned tt, bool oob) {
    const unsigned to = tile_off(tt) | (oob ? 0x80000000u : 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + buf * 16384 + (wave + 4 * i) * 1024),
                                               16, (int)(lane_off[i] + to), 0, 0, 0);
  };
  auto next = [&]() {
    if (!same_tile) { t += 1; if (t >= tiles_in_region) t = 0; }
  };
  if constexpr (MODE == M_DMA || MODE == M_DMA_OOB || MODE == M_DMA_MFMA) {
    dma(0, t, MODE == M_DMA_OOB);
    next();
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
      dma(buf ^ 1, t, MODE == M_DMA_OOB);
      next();
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(4));
      __builtin_amdgcn_s_barrier();
      if constexpr (MODE == M_DMA_MFMA) mfma32();
      buf ^= 1;
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
  } else if constexpr (MODE == M_REG) {
    u32x4_t v[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off[i] + tile_off(t)), 0, 0);
    next();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[1][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off[i] + tile_off(t)), 0, 0);
      next();
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[0][i];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off[i] + tile_off(t)), 0, 0);
      next();
#pragma unroll
      for (int i = 0; i < 4; ++i) acc ^= v[1][i];
    }
  } else if constexpr (MODE == M_REGST) {
    u32x4_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off[i] + tile_off(t)), 0, 0);
    next();
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4_t*)(smem + buf * 16384 + (wave + 4 * i) * 1024 + lane * 16) = v[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off[i] + tile_off(t)), 0, 0);
      next();
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only
      __builtin_amdgcn_s_barrier();
      buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= v[i];
  } else {
    for (int it = 0; it < iters; ++it) {
      mfma32();
      __builtin_amdgcn_s_barrier();
    }
  }
  unsigned s = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) s ^= __float_as_uint(macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3]);
  if (s == 0x12345678u) sink[blockIdx.x] = s + smem[tid];
}

template <int MODE>
static double run(const char* src, unsigned region, unsigned row_stride, int iters, int same, int wpc, unsigned* sink) {
  const int lds = 160 * 1024 / wpc - (wpc > 1 ? 1024 : 0);
  hipFuncSetAttribute((const void*)feed<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  feed<MODE><<<256 * wpc, 256, lds>>>(src, region, row_stride, iters / 4, same, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  feed<MODE><<<256 * wpc, 256, lds>>>(src, region, row_stride, iters, same, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
  return ms * 1e-3;
}

int main() {
  char* src;
  const size_t total = (size_t)1 << 30;
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  unsigned* sink;
  hipMalloc(&sink, 1 << 20);
  struct Src { const char* name; unsigned region; int same; } srcs[] = {
      {"L2 (2 MiB / XCD)", 2u << 20, 0}, {"L1 (one tile)", 2u << 20, 1}, {"HBM (128 MiB / XCD)", 128u << 20, 0}};
  const unsigned strides[] = {128, 2048};
  const int iters = 2000;
  printf("%-22s %6s %4s %12s %12s %12s %12s %12s %12s   (GB/s per CU; mfma: TF/s chip)\n", "source", "stride", "wpc", "dma", "dma_oob", "reg",
         "regst", "mfma", "dma+mfma");
  for (const Src& s : srcs)
    for (unsigned rs : strides)
      for (int wpc : {1, 2, 4}) {
        const double bytes_cu = (double)iters * 16384.0 * wpc;
        const double t0 = run<M_DMA>(src, s.region, rs, iters, s.same, wpc, sink);
        const double t1 = run<M_DMA_OOB>(src, s.region, rs, iters, s.same, wpc, sink);
        const double t2 = run<M_REG>(src, s.region, rs, iters, s.same, wpc, sink);
        const double t3 = run<M_REGST>(src, s.region, rs, iters, s.same, wpc, sink);
        const double t4 = run<M_MFMA>(src, s.region, rs, iters, s.same, wpc, sink);
        const double t5 = run<M_DMA_MFMA>(src, s.region, rs, iters, s.same, wpc, sink);
        const double mf = (double)iters * 32 * 16384.0 * 4 * wpc * 256;   // flops of the whole chip
        printf("%-22s %6u %4d %12.1f %12.1f %12.1f %12.1f %12.1f %7.1f/%6.1f\n", s.name, rs, wpc, bytes_cu / t0 * 1e-9, bytes_cu / t1 * 1e-9,
               bytes_cu / t2 * 1e-9, bytes_cu / t3 * 1e-9, mf / t4 * 1e-12, bytes_cu / t5 * 1e-9, mf / t5 * 1e-12);
        fflush(stdout);
      }
  return 0;
}
