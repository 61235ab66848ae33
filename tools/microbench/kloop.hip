// The igemm K loop in isolation (no address set-up, no epilogue): which tile / LDS-row width / ring depth / residency feeds
// the MFMAs best on MI355X?  4 waves (2 x 2), weights = A operand, pixels = B operand, both LDS-DMA'd from an L2-resident
// source with 2 KiB row stride, fragments by swizzled ds_read_b128, v_mfma_f32_16x16x32_bf16.
//   RB = bytes of K per LDS row and stage: 128 (BK = 64, 8 rows per DMA instruction) or 64 (BK = 32, 16 half-lines per
//   DMA instruction);  NS = LDS stages (NS - 1 tiles in flight, one barrier per stage)
//   hipcc --offload-arch=gfx950 -O3 kloop.hip -o kloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

template <int BC, int BP, int RB, int NS, int WC = 2, int WP = 2>
__global__ __launch_bounds__(64 * WC * WP, WC * WP == 4 ? 4 : 2) void kloop(const char* src, unsigned region_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WC * WP;
  constexpr int MC = BC / WC / 16, MP = BP / WP / 16;          // 16x16 blocks per wave
  constexpr int STAGE = (BC + BP) * RB;
  constexpr int RPI = 1024 / RB;                      // rows per DMA wave-instruction
  constexpr int LW = BC / (NW * RPI), LX = BP / (NW * RPI);
  constexpr int LD = LW + LX;
  constexpr int KJ = RB / 64;                         // MFMA K steps per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)xcd * region_bytes), 0, region_bytes, 0x00020000);
  constexpr unsigned RS = 2048;   // source row stride (K = 1024 bf16)
  auto swz = [](int row) { return RB == 128 ? ((row >> 1) & 7) : ((0x1320 >> (4 * ((row >> 2) & 3))) & 3); };
  // DMA lane geometry
  const int lrow = RB == 128 ? (lane >> 3) : (lane >> 2), lslot = RB == 128 ? (lane & 7) : (lane & 3);
  unsigned w_off[LW], x_off[LX];
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int row = RPI * (wave + NW * i) + lrow;
    w_off[i] = (unsigned)row * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  const unsigned pix0 = (1u << 19) + ((blockIdx.x >> 3) % 5u) * (unsigned)BP * RS;   // weights: first 512 KiB; pixels behind
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int row = RPI * (wave + NW * i) + lrow;
    x_off[i] = pix0 + (unsigned)row * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  unsigned koff = 0;
  auto gdma = [&](int buf) {
    char* base = smem + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + i * NW * 1024), 16, (int)(w_off[i] + koff), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < LX; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + BC * RB + i * NW * 1024), 16,
                                               (int)(x_off[i] + koff), 0, 0, 0);
    koff += RB;
    if (koff >= RS) koff = 0;
  };
  const int wc0 = (wave % WC) * (BC / WC), wp0 = (wave / WC) * (BP / WP);
  const int fr = lane & 15, fg = lane >> 4;
  f32x4_t acc[MC][MP];
#pragma unroll
  for (int a = 0; a < MC; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const char* wb = smem + buf * STAGE;
    const char* xb = wb + BC * RB;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      u32x4_t af[MC], bf[MP];
#pragma unroll
      for (int a = 0; a < MC; ++a) {
        const int r = wc0 + a * 16 + fr;
        af[a] = *(const u32x4_t*)(wb + r * RB + (((j * 4 + fg) ^ swz(r)) << 4));
      }
#pragma unroll
      for (int b = 0; b < MP; ++b) {
        const int r = wp0 + b * 16 + fr;
        bf[b] = *(const u32x4_t*)(xb + r * RB + (((j * 4 + fg) ^ swz(r)) << 4));
      }
#pragma unroll
      for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MP; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc[a][b], 0, 0, 0);
    }
  };
  if constexpr (NS == 1) {
    for (int it = 0; it < iters; ++it) {
      gdma(0);
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
      __builtin_amdgcn_s_barrier();
      compute(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
    }
  } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) gdma(s);
    int buf = 0, fill = NS - 1;
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm((NS - 2) * LD));
      __builtin_amdgcn_s_barrier();
      gdma(fill);
      compute(buf);
      buf = buf + 1 == NS ? 0 : buf + 1;
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < MC; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (__float_as_uint(s) == 0x12345678u) sink[blockIdx.x] = 1;
}

template <int BC, int BP, int RB, int NS, int WC = 2, int WP = 2>
static void run(const char* src, unsigned* sink) {
  constexpr int lds_need = NS * (BC + BP) * RB;
  const int iters = 4096 * 64 / RB / (BC * BP / 4096 > 1 ? 2 : 1);
  printf("tile %3dx%-3d RB %3d NS %d waves %dx%d (%3d KiB):", BC, BP, RB, NS, WC, WP, lds_need / 1024);
  for (int wpc : {1, 2, 3, 4, 5, 6, 8}) {
    int lds = 160 * 1024 / wpc;
    lds -= lds % 1024;
    if (wpc > 1) lds -= 1024;
    if (lds < lds_need) { printf("  %d/CU: -            ", wpc); continue; }
    hipFuncSetAttribute((const void*)kloop<BC, BP, RB, NS, WC, WP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kloop<BC, BP, RB, NS, WC, WP>, 64 * WC * WP, lds);
    if (occ < wpc) { printf("  %d/CU: occ %d         ", wpc, occ); continue; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kloop<BC, BP, RB, NS, WC, WP><<<256 * wpc, 64 * WC * WP, lds>>>(src, 2u << 20, iters / 4, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kloop<BC, BP, RB, NS, WC, WP><<<256 * wpc, 64 * WC * WP, lds>>>(src, 2u << 20, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
    const double t = ms * 1e-3;
    const double bytes_cu = (double)iters * (BC + BP) * RB * wpc;
    const double flops = (double)iters * 2.0 * BC * BP * (RB / 2) * wpc * 256;
    printf("  %d/CU: %5.1f GB/s %6.0f TF", wpc, bytes_cu / t * 1e-9, flops / t * 1e-12);
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  char* src;
  hipMalloc(&src, (size_t)16 << 20);
  hipMemset(src, 0x3c, (size_t)16 << 20);   // bf16 0x3c3c = 0.011: finite, non-zero operands
  unsigned* sink;
  hipMalloc(&sink, 1 << 20);
  run<128, 128, 128, 2>(src, sink);
  run<128, 128, 128, 1>(src, sink);
  run<128, 128, 128, 2, 2, 4>(src, sink);
  run<128, 128, 128, 2, 4, 2>(src, sink);
  run<128, 128, 128, 1, 2, 4>(src, sink);
  run<128, 128, 128, 3, 2, 4>(src, sink);
  run<256, 128, 128, 2, 4, 2>(src, sink);
  run<128, 256, 128, 2, 2, 4>(src, sink);
  run<256, 128, 128, 1, 4, 2>(src, sink);
  run<128, 64, 128, 2, 4, 2>(src, sink);
  run<64, 128, 128, 2, 2, 4>(src, sink);
  return 0;
}
