// K loop with the WEIGHT operand loaded straight from FRAGMENT-PACKED global memory into MFMA A-fragment registers (no LDS for it) and
// only the pixels staged through the LDS-DMA ring - against the product layout (both operands through LDS-DMA).  Round-5 review, item 1:
// the weights are the operand that is contiguous, small, L2-resident and already re-packed every step; packed in fragment order a wave's
// A fragment of one K step is ONE contiguous 1 KiB global_load_dwordx4 (lane l -> bytes 16 l), the access shape the texture path likes
// (kloop_regb.hip showed that 16 rows x 64 B per instruction - the pixel operand - is 4x slower than the LDS-DMA ring).
//
//   packed image: [K tile][16- or 32-channel block][K step][lane][16 B]
//     16x16x32: block = 16 channels, 2 K steps of 32;  lane (r = l & 15, g = l >> 4) = channel r, K elements 8 g .. 8 g + 7 of the step
//     32x32x16: block = 32 channels, 4 K steps of 16;  lane (r = l & 31, h = l >> 5) = channel r, K elements 8 h .. 8 h + 7 of the step
//   either way one K tile (64 deep) of one 16-channel block is 2 KiB, and consecutive blocks / K tiles are contiguous.
//
// Families:  AREG = 0  both operands through a D-stage LDS-DMA ring, one barrier per K tile (the product's 4-wave structure; for the
//                      256 x 256 tile the same simple ring on 8 waves - NOT the product's ping-pong schedule, so compare it with the
//                      AREG = 1 row of the same tile, not with bench_conv)
//            AREG = 1  pixels through the D-stage ring, weights D - 1 K tiles ahead in registers
//   M32: v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16 (same LDS image, same XOR swizzle: conv_igemm_big.hip M32)
// Operands are pseudo-random bf16 in [-1, 1) (zero / constant fills clock up to 19 % higher: MI355X_MICROARCH.md "DVFS give-back").
// Pixels: L2-resident (mode 0: five shared BP-row regions) or one region per workgroup of a 256 MiB buffer (mode 1: Infinity-Cache /
// HBM resident on the first pass over K, as an activation behind its producer).  Weights: 256 channels x K 1024 = 512 KiB, shared.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 kloop_rega.hip -o kloop_rega
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }
constexpr unsigned RS = 2048;        // pixel source row stride: K = 1024 bf16
constexpr unsigned KT_BYTES = 2048;  // packed weights: bytes of one 16-channel block per K tile
constexpr int KTILES = 16;           // K = 1024

template <int BC, int BP, int WC, int WP, int D, bool AREG, bool M32>
__global__ __launch_bounds__(64 * WC * WP) void kloop(const char* wsrc, const char* wpk, const char* xsrc, unsigned x_bytes, int iters, int far_mode,
                                                      unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WC * WP;
  constexpr int TC = BC / WC, TP = BP / WP;
  constexpr int MB = M32 ? 32 : 16;            // MFMA block edge
  constexpr int KS = M32 ? 4 : 2;              // K steps per 64-deep K tile
  constexpr int MC = TC / MB, MP = TP / MB;
  static_assert(TC % MB == 0 && TP % MB == 0, "wave tile vs MFMA block");
  constexpr int STAGE = (AREG ? BP : BC + BP) * 128;
  constexpr int LW = AREG ? 0 : BC / (NW * 8), LX = BP / (NW * 8);
  static_assert(AREG || BC % (NW * 8) == 0, "DMA rows");
  static_assert(BP % (NW * 8) == 0, "DMA rows");
  constexpr int LA = AREG ? MC * KS : 0;       // register loads per K tile and lane (16 B each)
  constexpr int L = LW + LX + LA;              // vector-memory operations per K tile and thread
  static_assert((D - 1) * L < 64, "vmcnt");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 1u << 19, 0x00020000);
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, 1u << 19, 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xsrc, 0, x_bytes, 0x00020000);
  auto swz = [](int row) { return (row >> 1) & 7; };
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned w_off[LW > 0 ? LW : 1], x_off[LX];
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    w_off[i] = (unsigned)(row % 256) * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  const unsigned pix0 = far_mode ? (unsigned)((blockIdx.x * BP) % (x_bytes / RS - BP)) * RS : ((blockIdx.x >> 3) % 5u) * (unsigned)BP * RS;
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    x_off[i] = pix0 + (unsigned)row * RS + (unsigned)((lslot ^ swz(row)) * 16);
  }
  const int wc0 = (wave % WC) * TC, wp0 = (wave / WC) * TP;
  // packed-weight offset of this wave's first block in K tile 0: [K tile][256 / 16 blocks][2 KiB]; the K steps of a block are contiguous
  const unsigned a_off = (unsigned)(wc0 / 16) * KT_BYTES + (unsigned)lane * 16;
  constexpr unsigned KT_STRIDE = 16 * KT_BYTES;   // 256 channels per K tile
  unsigned koff = 0, kt = 0;
  u32x4_t areg[AREG ? D : 1][KS][MC];
  auto issue = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    char* base = smem + slot * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + i * NW * 1024), 16, (int)(w_off[i] + koff), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < LX; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + (AREG ? 0 : BC * 128) + i * NW * 1024), 16,
                                               (int)(x_off[i] + koff), 0, 0, 0);
    if constexpr (AREG) {
      // block a of the wave = MB channels = MB / 16 packed 2 KiB units; K step q of it: 16x16x32 -> 1 KiB per step inside one unit;
      // 32x32x16 -> the 32-channel block's 4 KiB hold 4 steps of 1 KiB
#pragma unroll
      for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int q = 0; q < KS; ++q)
          areg[slot][q][a] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)(a_off + kt * KT_STRIDE + (unsigned)a * (MB / 16) * KT_BYTES + (unsigned)q * 1024), 0, 0);
    }
    koff += 128;
    if (koff >= RS) koff = 0;
    kt = (kt + 1) & (KTILES - 1);
  };
  f32x4_t acc[M32 ? 1 : MC][M32 ? 1 : MP];
  f32x16_t acc32[M32 ? MC : 1][M32 ? MP : 1];
  if constexpr (M32) {
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[a][b][e] = 0.f;
  } else {
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const int fr = M32 ? (lane & 31) : (lane & 15), fg = M32 ? (lane >> 5) : (lane >> 4);
  auto compute = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    const char* wb = smem + slot * STAGE;
    const char* xb = wb + (AREG ? 0 : BC * 128);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      // 16-byte slot of K step q for this lane: 16x16x32: 4 q + g;  32x32x16: 2 q + h
      const int s = M32 ? 2 * q + fg : 4 * q + fg;
      u32x4_t af[MC], bf[MP];
#pragma unroll
      for (int a = 0; a < MC; ++a) {
        if constexpr (AREG) {
          af[a] = areg[slot][q][a];
        } else {
          const int r = wc0 + a * MB + fr;
          af[a] = *(const u32x4_t*)(wb + r * 128 + ((s ^ swz(r)) << 4));
        }
      }
#pragma unroll
      for (int b = 0; b < MP; ++b) {
        const int r = wp0 + b * MB + fr;
        bf[b] = *(const u32x4_t*)(xb + r * 128 + ((s ^ swz(r)) << 4));
      }
#pragma unroll
      for (int a = 0; a < MC; ++a)
#pragma unroll
        for (int b = 0; b < MP; ++b) {
          if constexpr (M32)
            acc32[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc32[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]), __builtin_bit_cast(bf16x8_t, bf[b]), acc[a][b], 0, 0, 0);
        }
    }
  };
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
  if constexpr (M32) {
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc32[a][b][e];
  } else {
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  }
  if (__float_as_uint(s) == 0x12345678u) sink[blockIdx.x] = 1;
}

__global__ void fill_random(unsigned* p, size_t n_words, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    // two bf16 in [-1, 1): sign + exponent 0x3f0.. (0.5 .. 1) or smaller, mantissa random
    const unsigned lo = (x & 0x807fu) | (0x3e80u + ((x >> 8) & 0x0100u)), hi = ((x >> 16) & 0x807fu) | (0x3e80u + ((x >> 24) & 0x0100u));
    p[i] = lo | (hi << 16);
  }
}

template <int BC, int BP, int WC, int WP, int D, bool AREG, bool M32>
static void run(const char* wsrc, const char* wpk, const char* xsrc, unsigned x_bytes, unsigned* sink) {
  constexpr int lds_need = D * (AREG ? BP : BC + BP) * 128;
  const int iters = 2048 - 2048 % D;
  auto kfn = kloop<BC, BP, WC, WP, D, AREG, M32>;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)kfn);
  for (int far_mode = 0; far_mode < 2; ++far_mode) {
    printf("tile %3dx%-3d waves %dx%d D %d %-12s %-9s %s (%3d KiB LDS, %3d VGPR%s):", BC, BP, WC, WP, D, AREG ? "weights->VGPR" : "both LDS-DMA",
           M32 ? "32x32x16" : "16x16x32", far_mode ? "px first-touch" : "px L2-resident", lds_need / 1024, fa.numRegs,
           fa.localSizeBytes ? " SPILL" : "");
    for (int wpc : {1, 2, 3, 4}) {
      int lds = 160 * 1024 / wpc;
      lds -= lds % 1024;
      if (wpc > 1) lds -= 1024;
      if (lds < lds_need) { printf("  %d/CU: -              ", wpc); continue; }
      hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      int occ = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kfn, 64 * WC * WP, lds);
      if (occ < wpc) { printf("  %d/CU: occ %d          ", wpc, occ); continue; }
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      kfn<<<256 * wpc, 64 * WC * WP, lds>>>(wsrc, wpk, xsrc, x_bytes, iters / 4 - (iters / 4) % D, far_mode, sink);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      kfn<<<256 * wpc, 64 * WC * WP, lds>>>(wsrc, wpk, xsrc, x_bytes, iters, far_mode, sink);
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

int main(int argc, char** argv) {
  char *wsrc, *wpk, *xsrc;
  const unsigned x_bytes = 1u << 28;   // 256 MiB of pixel rows (2 KiB each)
  hipMalloc(&wsrc, (size_t)1 << 20);
  hipMalloc(&wpk, (size_t)1 << 20);
  hipMalloc(&xsrc, (size_t)x_bytes);
  fill_random<<<1024, 256>>>((unsigned*)wsrc, ((size_t)1 << 20) / 4, 1u);
  fill_random<<<1024, 256>>>((unsigned*)wpk, ((size_t)1 << 20) / 4, 2u);
  fill_random<<<4096, 256>>>((unsigned*)xsrc, (size_t)x_bytes / 4, 3u);
  hipDeviceSynchronize();
  unsigned* sink;
  hipMalloc(&sink, 1 << 20);
  const int sel = argc > 1 ? atoi(argv[1]) : 0;   // 0 = everything, 1 = 64 x 64, 2 = 128 x 128, 3 = 256 x 256
  if (sel == 0 || sel == 1) {
    run<64, 64, 2, 2, 2, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 4, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 2, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 4, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 8, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 4, 1, 4, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 4, 1, 8, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 4, false, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 4, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<64, 64, 2, 2, 8, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
  }
  if (sel == 0 || sel == 2) {
    run<128, 128, 2, 2, 2, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 3, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 2, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 3, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 4, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 4, 1, 3, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 2, false, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 2, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 2, 2, 3, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<128, 128, 4, 1, 3, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
  }
  if (sel == 0 || sel == 3) {
    run<256, 256, 4, 2, 2, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 4, 2, 2, false, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 2, 4, 2, false, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 4, 4, 2, false, false>(wsrc, wpk, xsrc, x_bytes, sink);   // 16 waves: 4 per SIMD, 64 x 64 wave tiles
    run<256, 256, 4, 4, 2, false, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 4, 2, 2, true, false>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 4, 2, 2, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 4, 2, 3, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 2, 4, 2, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 256, 8, 1, 2, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 128, 4, 2, 3, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
    run<256, 128, 4, 1, 3, true, true>(wsrc, wpk, xsrc, x_bytes, sink);
  }
  return 0;
}
