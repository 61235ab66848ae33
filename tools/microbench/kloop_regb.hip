// K loop with the PIXEL operand loaded straight into MFMA B-fragment registers (no LDS for it) and only the weights staged through the
// LDS-DMA ring - against the product layout (both operands through LDS-DMA).  Round 5 found the K loop of the 64 x 64 / 128 x 128 tiles
// latency-bound at the LDS capacity (profiles/r05_igemm_tile_timeline.txt): ~100 KiB in flight per CU is all 160 KiB of LDS can hold,
// while the register file (512 KiB per CU) is mostly idle.  Here a wave keeps D - 1 K tiles of ITS pixel fragments in flight in VGPRs.
//   lane (r = lane & 15, g = lane >> 4) of a wave loads, per 16-pixel block and 32-wide K step, 16 bytes: row r, K bytes 16 (4 j + g)
//   (the B-fragment layout of v_mfma_f32_16x16x32_bf16) - 16 rows x 64 contiguous bytes per wave instruction.
// The two channel halves of a 2 x 2 wave layout load the same pixels twice (L1 / TA bandwidth); WCx1 layouts do not.
// Source: weights L2-resident (512 KiB), pixels either L2-resident (mode 0) or streamed once from a 256 MiB buffer (mode 1: every
// 128-byte line is a first touch, like the activation a conv reads behind its producer).  Also: "MFMA alone" with 16x16x32 and 32x32x16.
//   hipcc --offload-arch=gfx950 -O3 kloop_regb.hip -o kloop_regb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }
constexpr unsigned RS = 2048;   // source row stride: K = 1024 bf16

// REGB = false: both operands through the LDS-DMA ring (the product K loop, D stages); true: weights through the ring, pixels in registers
template <int BC, int BP, int WC, int WP, int D, bool REGB>
__global__ __launch_bounds__(64 * WC * WP) void kloop(const char* wsrc, const char* xsrc, unsigned x_bytes, int iters, int far_mode, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WC * WP;
  constexpr int MC = BC / WC / 16, MP = BP / WP / 16;
  constexpr int STAGE = (REGB ? BC : BC + BP) * 128;
  constexpr int LW = BC / (NW * 8), LX = REGB ? 0 : BP / (NW * 8);
  constexpr int LB = REGB ? 2 * MP : 0;            // register loads per K tile and lane
  constexpr int L = LW + LX + LB;                  // vector-memory operations per K tile and thread
  static_assert((D - 1) * L < 64, "vmcnt");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 1u << 19, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xsrc, 0, x_bytes, 0x00020000);
  auto swz = [](int row) { return (row >> 1) & 7; };
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned w_off[LW], x_off[LX > 0 ? LX : 1];
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    w_off[i] = (unsigned)(row % 256) * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  // pixel rows of this workgroup: far mode = its own BP rows of the big buffer, else a small shared region
  const unsigned pix0 = far_mode ? (unsigned)((blockIdx.x * BP) % (x_bytes / RS - BP)) * RS : ((blockIdx.x >> 3) % 5u) * (unsigned)BP * RS;
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    x_off[i] = pix0 + (unsigned)row * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  const int wc0 = (wave % WC) * (BC / WC), wp0 = (wave / WC) * (BP / WP);
  const int fr = lane & 15, fg = lane >> 4;
  unsigned b_off[MP];
#pragma unroll
  for (int b = 0; b < MP; ++b) b_off[b] = pix0 + (unsigned)(wp0 + b * 16 + fr) * RS + (unsigned)(fg * 16);
  unsigned koff = 0;
  u32x4_t breg[REGB ? D : 1][2][MP];
  auto issue = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    char* base = smem + slot * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + i * NW * 1024), 16, (int)(w_off[i] + koff), 0, 0, 0);
    if constexpr (!REGB) {
#pragma unroll
      for (int i = 0; i < LX; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + BC * 128 + i * NW * 1024), 16,
                                                 (int)(x_off[i] + koff), 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int b = 0; b < MP; ++b)
          breg[slot][j][b] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(b_off[b] + koff + j * 64), 0, 0);
    }
    koff += 128;
    if (koff >= RS) koff = 0;
  };
  f32x4_t acc[MC][MP];
#pragma unroll
  for (int a = 0; a < MC; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const char* wb = smem + slot * STAGE;
    const char* xb = wb + BC * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4_t af[MC], bf[MP];
#pragma unroll
      for (int a = 0; a < MC; ++a) {
        const int r = wc0 + a * 16 + fr;
        af[a] = *(const u32x4_t*)(wb + r * 128 + (((j * 4 + fg) ^ swz(r)) << 4));
      }
#pragma unroll
      for (int b = 0; b < MP; ++b) {
        if constexpr (REGB) {
          bf[b] = breg[slot][j][b];
        } else {
          const int r = wp0 + b * 16 + fr;
          bf[b] = *(const u32x4_t*)(xb + r * 128 + (((j * 4 + fg) ^ swz(r)) << 4));
        }
      }
#pragma unroll
      for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MP; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc[a][b], 0, 0, 0);
    }
  };
  // prologue: D - 1 tiles in flight; then groups of D iterations with static slots
  auto step = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    __builtin_amdgcn_s_waitcnt(vmcnt_imm((D - 2) * L));
    __builtin_amdgcn_s_barrier();
    issue(std::integral_constant<int, (slot + D - 1) % D>{});
    compute(slot_c);
  };
  [&]<int... S>(std::integer_sequence<int, S...>) { (issue(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, D - 1>{});
  for (int it = 0; it < iters; it += D)
    [&]<int... S>(std::integer_sequence<int, S...>) { (step(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, D>{});
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < MC; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (__float_as_uint(s) == 0x12345678u) sink[blockIdx.x] = 1;
}

template <int BC, int BP, int WC, int WP, int D, bool REGB>
static void run(const char* wsrc, const char* xsrc, unsigned x_bytes, unsigned* sink) {
  constexpr int lds_need = D * (REGB ? BC : BC + BP) * 128;
  const int iters = 2048;
  for (int far_mode = 0; far_mode < 2; ++far_mode) {
    printf("tile %3dx%-3d waves %dx%d D %d %-22s %s (%3d KiB):", BC, BP, WC, WP, D, REGB ? "pixels in registers" : "both through LDS-DMA",
           far_mode ? "pixels first-touch" : "pixels L2-resident", lds_need / 1024);
    for (int wpc : {1, 2, 3, 4}) {
      int lds = 160 * 1024 / wpc;
      lds -= lds % 1024;
      if (wpc > 1) lds -= 1024;
      if (lds < lds_need) { printf("  %d/CU: -              ", wpc); continue; }
      auto kfn = kloop<BC, BP, WC, WP, D, REGB>;
      hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      int occ = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, 64 * WC * WP, lds);
      if (occ < wpc) { printf("  %d/CU: occ %d          ", wpc, occ); continue; }
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      kfn<<<256 * wpc, 64 * WC * WP, lds>>>(wsrc, xsrc, x_bytes, iters / 4, far_mode, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      kfn<<<256 * wpc, 64 * WC * WP, lds>>>(wsrc, xsrc, x_bytes, iters, far_mode, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
      const double t = ms * 1e-3;
      const double flops = (double)iters * 2.0 * BC * BP * 64 * wpc * 256;
      const double cyc = t * 2.4e9 / iters;
      printf("  %d/CU: %6.0f TF %5.0f cyc/step", wpc, flops / t * 1e-12, cyc);
    }
    printf("\n");
    fflush(stdout);
  }
}

// MFMA alone: independent accumulators, no operands loaded (register-resident fragments)
template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_alone(int iters, unsigned* sink) {
  bf16x8_t a = __builtin_bit_cast(bf16x8_t, (u32x4_t){0x3c3c3c3cu, 0x3c3c3c3cu, 0x3c3c3c3cu, 0x3c3c3c3cu}), b = a;
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
  } else {
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  }
  if (__float_as_uint(s) == 0x12345678u) sink[blockIdx.x] = 1;
}

template <int SHAPE>
static void run_mfma(unsigned* sink) {
  for (int wpc : {1, 2, 4}) {
    const int iters = 100000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mfma_alone<SHAPE><<<256 * wpc, 256>>>(1000, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_alone<SHAPE><<<256 * wpc, 256>>>(iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = SHAPE == 16 ? 8.0 * 2 * 16 * 16 * 32 : 4.0 * 2 * 32 * 32 * 16;
    printf("MFMA alone %s, %d waves per SIMD: %6.0f TF/s\n", SHAPE == 16 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", wpc,
           (double)iters * per * 4 * 256 * wpc / (ms * 1e-3) * 1e-12);
  }
}

int main() {
  char *wsrc, *xsrc;
  const unsigned x_bytes = 1u << 28;   // 256 MiB of pixel rows (2 KiB each)
  hipMalloc(&wsrc, (size_t)1 << 20);
  hipMalloc(&xsrc, (size_t)x_bytes);
  hipMemset(wsrc, 0x3c, (size_t)1 << 20);
  hipMemset(xsrc, 0x3c, (size_t)x_bytes);
  unsigned* sink;
  hipMalloc(&sink, 1 << 20);
  run_mfma<16>(sink);
  run_mfma<32>(sink);
  run<64, 64, 2, 2, 2, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 2, 2, 4, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 2, 2, 4, true>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 2, 2, 8, true>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 4, 1, 4, true>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 4, 1, 6, true>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 2, false>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 3, false>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 4, false>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 5, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 128, 2, 2, 2, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 128, 2, 2, 4, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 128, 2, 2, 6, false>(wsrc, xsrc, x_bytes, sink);
  run<64, 64, 2, 2, 8, false>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 3, true>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 2, 2, 4, true>(wsrc, xsrc, x_bytes, sink);
  run<128, 128, 4, 1, 3, true>(wsrc, xsrc, x_bytes, sink);
  run<128, 64, 4, 1, 4, true>(wsrc, xsrc, x_bytes, sink);
  run<128, 64, 4, 1, 6, true>(wsrc, xsrc, x_bytes, sink);
  return 0;
}
