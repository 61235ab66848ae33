// ARCHIVED (round 6): built, parity-green (tests/test_gpu_igemm_big.py at commit time: bit-identical to the 8-wave tile), 10 .. 19 % SLOWER than the
// 8-wave ring as a product kernel although its K loop is 18 % faster in tools/microbench/kloop_rega.hip - profiles/r06_notes.md section 1.
// Not part of the build (cavp_amd/build.py); kept as the evidence behind that section.
// 256 x 256 implicit-GEMM tile on SIXTEEN waves (bf16, gfx950): one 1024-thread workgroup per CU, four waves per SIMD, a plain 2-stage
// LDS-DMA ring with ONE barrier per K tile, epilogue straight from the accumulator registers.
//
// Why (round 6, tools/microbench/kloop_rega.hip -> profiles/r06_kloop_rega.txt, same box, random operands, K loop only):
//   256 x 256, 8 waves (64 x 128 wave tiles), 2-stage ring          1275 TF/s   (the ping-pong schedule of conv_igemm_big.hip measures ~3 % below it)
//   256 x 256, 8 waves, weights straight into VGPRs (packed)         1383 .. 1421
//   256 x 256, 16 waves (64 x 64 wave tiles), 2-stage ring           1506
// Four waves per SIMD hide each other's fragment reads, DMA issue and MFMA latencies without any hand-built phase structure (the regime
// the 128 x 128 tile gets from its co-resident workgroups and the weight-gradient kernel from its 16-wave tile, profiles/r05_notes.md 1);
// a wave issues 4 LDS-DMA pieces per K tile instead of 8, holds 64 accumulator registers (<= 128 VGPRs), and the two SIMD-mates of the
// 8-wave kernel that had to meet at 8 barriers per K tile are replaced by four that meet at one.
//
//  * 16 waves = 4 (channel) x 4 (pixel); wave tile 64 channels x 64 pixels = 4 x 4 MFMA 16x16x32 blocks, 32 MFMAs per wave and K tile.
//  * LDS: 2 stages x (256 weight rows + 256 pixel rows) x 128 B = 128 KiB; same row image as conv_igemm.hip (16-byte slot index
//    XOR-swizzled by (row >> 1) & 7 on the SOURCE address of the DMA, out-of-range pieces zero-filled by the buffer descriptor).
//  * iteration u: wait for MY pieces of K tile u, barrier (everybody's landed, everybody is done with K tile u - 1), issue K tile u + 1
//    into the stage K tile u - 1 vacated, multiply K tile u.  The K-tile stream runs across the output tiles of the persistent
//    workgroup: the first K tile of the next output tile is in flight during the epilogue.
//  * epilogue from registers (v_permlane16_swap, conv_igemm.hip's register epilogue): a lane ends with 8 consecutive channels of one
//    pixel = one 16-byte store, 64 contiguous bytes per pixel and instruction; no LDS round trip, no scratch.  BatchNorm statistics
//    from the accumulators: per-wave (mean, M2) over its 64 rows, pairs of pixel waves combined through 8 KiB of LDS into the same
//    128-row statistics tiles conv_igemm_big.hip writes (cavp_conv2d_tile_stats_layout is unchanged).
#include <type_traits>

#include "igemm_params.h"

namespace {

constexpr int BC = 256, BP = 256, NT = 1024;
constexpr int STAGE_BYTES = (BC + BP) * 128;   // one K tile: 256 weight rows, 256 pixel rows
constexpr int RING_BYTES = 2 * STAGE_BYTES;
constexpr int STAT_BYTES = 4 * BC * 8;         // [pixel wave][channel] (mean, M2)
constexpr int LDS_BYTES = RING_BYTES + STAT_BYTES;
constexpr unsigned kOOB = 0x80000000u;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");

typedef __attribute__((address_space(3))) void* lds_ptr_t;

}  // namespace

// DBG: compile-time profiling switches (-DCAVP_PROFILE builds only): 8 no DMA, 16 no MFMA, 32 no fragment reads, 64 no epilogue.
template <int DBG>
__global__ __launch_bounds__(1024) void igemm_big16_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cavp_prefetch_kernargs<(int)sizeof(IgemmParams)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, wp = wave >> 2;
  const int lrow = lane & 15, lgrp = lane >> 4;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int my_tiles = ((int)blockIdx.x < p.nblk) ? (p.nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_tiles == 0) return;
  const int HoWo = p.Ho * p.Wo;

  // ---------------------------------------------------------------------------------------------------------------
  // issue side: a wave DMA instruction writes 8 LDS rows (1 KiB) linearly; thread -> rows r0 and r0 + 128 of each operand
  // ---------------------------------------------------------------------------------------------------------------
  const int r0 = 8 * wave + (lane >> 3);
  const int kslot = ((lane & 7) ^ ((r0 >> 1) & 7)) * 8;   // ((r0 + 128) >> 1) & 7 == (r0 >> 1) & 7
  const int kbyte = kslot * 2;
  unsigned w_off[2], x_off[2], x_mask[2];
  int iss_tile = 0, iss_ti = 0, iss_cc = 0, iss_buf = 0;
  int iss_woff = p.tap_woff[0], iss_xoff = p.tap_xoff[0];

  auto setup_tile = [&](int ord) {
    const int vb = (int)blockIdx.x + ord * (int)gridDim.x;
    const int sid = xcd_remap(vb, p.nblk);
    const int tp = fast_div(sid, p.div_tc_m, p.div_tc_s), tc = sid - tp * p.tiles_c;
    const int c_base = tc * BC, p_base = tp * BP;
    int h0[2], w0[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int c = c_base + r0 + 128 * m;
      w_off[m] = c < p.Cout ? (unsigned)(((size_t)c * p.K + kslot) * 2) : kOOB;
      const int pix = p_base + r0 + 128 * m;
      const bool ok = pix < p.M;
      const int pp = ok ? pix : 0;
      const int n = fast_div(pp, p.div_hw_m, p.div_hw_s), rr = pp - n * HoWo;
      const int ho = fast_div(rr, p.div_w_m, p.div_w_s), wo = rr - ho * p.Wo;
      h0[m] = ok ? ho * p.stride - p.pad : -0x10000000;   // a dead row fails every bounds test
      w0[m] = wo * p.stride_w - p.pad;
      x_mask[m] = 0;
      x_off[m] = (unsigned)((n * p.H + h0[m]) * p.W + w0[m]) * (unsigned)(p.ldx * 2) + (unsigned)kbyte;
    }
    for (int t = 0; t < p.ntaps; ++t) {
      const int dh = p.tap_dh[t], dw = p.tap_dw[t];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        x_mask[m] |= ((unsigned)(h0[m] + dh) < (unsigned)p.H && (unsigned)(w0[m] + dw) < (unsigned)p.W) ? (1u << t) : 0u;
    }
  };

  // the four pieces of the K tile at the issue position; past the last tile the same instructions are issued with every lane out of
  // range (zero fill into a stage nobody reads any more): the loop stays uniform
  auto issue_ktile = [&]() {
    const bool live = iss_tile < my_tiles;
    const int c0 = iss_cc * 64;
    const unsigned oobm = (live && (c0 + kslot) < p.Cin) ? 0u : kOOB;
    char* base = smem + iss_buf * STAGE_BYTES + wave * 1024;
    if constexpr ((DBG & 8) == 0) {
      const unsigned wk = (unsigned)(iss_woff + c0 * 2), xk = (unsigned)(iss_xoff + c0 * 2);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(base + m * 16384), 16, (int)((w_off[m] + wk) | oobm), 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const unsigned tapm = ((~(x_mask[m] >> iss_ti)) & 1u) << 31;   // tap outside the image for this pixel
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(base + BC * 128 + m * 16384), 16, (int)((x_off[m] + xk) | tapm | oobm), 0,
                                                 0, 0);
      }
    }
    iss_buf ^= 1;
    if (live) {
      // taps INNERMOST: the 9 shifted windows of one 64-channel slice are fetched back to back, so 8 of the 9 reads of an input line
      // hit the XCD's L2 (conv_igemm_big.hip)
      if (++iss_ti == p.ntaps) {
        iss_ti = 0;
        if (++iss_cc == p.cpt) {
          iss_cc = 0;
          if (++iss_tile < my_tiles) setup_tile(iss_tile);
        }
      }
      iss_woff = p.tap_woff[iss_ti];
      iss_xoff = p.tap_xoff[iss_ti];
    }
  };

  // ---------------------------------------------------------------------------------------------------------------
  // compute side
  // ---------------------------------------------------------------------------------------------------------------
  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int key = (lrow >> 1) & 7;   // == (row >> 1) & 7 for every row this lane reads (rows differ by multiples of 16)
  const int a_off = (wc * 64 + lrow) * 128 + ((lgrp ^ key) << 4);             // K sub-step 0; sub-step 1 flips slot bit 2 (byte 64)
  const int b_off = BC * 128 + (wp * 64 + lrow) * 128 + ((lgrp ^ key) << 4);
  int cmp_buf = 0;

  auto multiply_ktile = [&]() {
    const char* base = smem + cmp_buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      u32x4_t af[4], bf[4];
      if constexpr ((DBG & 32) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = *(const u32x4_t*)(base + (a_off ^ (j << 6)) + a * 2048);
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = *(const u32x4_t*)(base + (b_off ^ (j << 6)) + b * 2048);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) { af[a] = (u32x4_t){(unsigned)lane, 0u, 0u, 0u}; bf[a] = af[a]; }
      }
      if constexpr ((DBG & 16) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) Mma<bf16_t>::run(acc[a][b], af[a], bf[b]);
      } else {
        asm volatile("" ::"v"(af[0]), "v"(bf[0]));
      }
      // one fragment set (32 registers) live at a time: with both sub-steps' reads hoisted over the first MFMA block the kernel needs
      // 64 + 64 registers for accumulators + fragments alone and spills its DMA descriptors into the K loop (whose scratch reloads
      // then share the vmcnt counter with the ring); the other three waves of the SIMD cover this wave's read latency
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue of one wave: acc (64 channels x 64 pixels) -> y, from registers ----
  auto epilogue = [&](int ord) {
    const int vb = (int)blockIdx.x + ord * (int)gridDim.x;
    const int sid = xcd_remap(vb, p.nblk);
    const int tp = fast_div(sid, p.div_tc_m, p.div_tc_s), tc = sid - tp * p.tiles_c;
    const int c_base = tc * BC, p_base = tp * BP;
    const int c_wave = c_base + wc * 64, p_wave = p_base + wp * 64;
    const int nvw = p.M - p_wave;   // valid pixel rows of this wave's slab (<= 0: none)
    if (p.tile_stats) {
      // per-channel (mean, M2) of this wave's <= 64 rows straight from the accumulators (sums about the slab's first row, DPP row sums),
      // then the two waves of a 128-row statistics tile through LDS (Chan): the layout of conv_igemm_big.hip
      float2* wstat = (float2*)(smem + RING_BYTES);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float s1[4], s2[4], x0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x0[i] = __shfl(acc[a][0][i], lane & 48, 64);
          s1[i] = 0.f; s2[i] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool ok = b * 16 + lrow < nvw;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d = ok ? acc[a][b][i] - x0[i] : 0.f;
            s1[i] += d;
            s2[i] = fmaf(d, d, s2[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          s1[i] = row16_sum(s1[i]);
          s2[i] = row16_sum(s2[i]);
        }
        if (lrow == 0) {
          const float n = (float)(nvw < 64 ? (nvw > 0 ? nvw : 1) : 64);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float m = s1[i] / n;
            wstat[wp * BC + wc * 64 + a * 16 + lgrp * 4 + i] = make_float2(x0[i] + m, fmaxf(s2[i] - s1[i] * m, 0.f));
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0) only: the next K tile's DMA stays in flight across this barrier
      __builtin_amdgcn_s_barrier();
      if (tid < 2 * BC) {
        const int h = tid >> 8, ch = tid & (BC - 1);
        int nb0 = p.M - (p_base + h * 128), nb1 = nb0 - 64;
        nb0 = nb0 < 64 ? nb0 : 64;
        nb1 = nb1 < 64 ? nb1 : 64;
        if (nb0 > 0 && c_base + ch < p.Cout) {
          const float2 q0 = wstat[(2 * h) * BC + ch];
          float mean = q0.x, m2 = q0.y;
          if (nb1 > 0) {
            const float2 q1 = wstat[(2 * h + 1) * BC + ch];
            const float f0 = (float)nb0, f1 = (float)nb1, nt = f0 + f1, dlt = q1.x - mean;
            mean += dlt * (f1 / nt);
            m2 += q1.y + dlt * dlt * (f0 * f1 / nt);
          }
          *(float2*)(p.tile_stats + ((size_t)(tp * 2 + h) * p.Cout + c_base + ch) * 2) = make_float2(mean, m2);
        }
      }
    }
    // v_permlane16_swap of the register pair of two neighbouring 16-channel blocks leaves lane (pixel lrow, g = lgrp) with 8 CONSECUTIVE
    // channels of block 2 q + (g & 1): channels 8 (g >> 1) .. + 7 (conv_igemm.hip, register epilogue; probed on hardware)
    const bool has_ss = p.scale != nullptr || p.shift != nullptr;
    // (res_rows is a multiple of 256: a tile never straddles the wrap)
    const int res_base = (p.res_rows ? p_base % p.res_rows : p_base) + wp * 64;
    // block pair outermost: one set of per-channel coefficients (16 registers) is live at a time
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cch = c_wave + (2 * q + (lgrp & 1)) * 16 + 8 * (lgrp >> 1);
      const bool cok = cch < p.Cout;
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
      if (cok) {
        if (p.scale) {
          const float4 t0 = *(const float4*)(p.scale + cch), t1 = *(const float4*)(p.scale + cch + 4);
          sc[0] = t0.x; sc[1] = t0.y; sc[2] = t0.z; sc[3] = t0.w; sc[4] = t1.x; sc[5] = t1.y; sc[6] = t1.z; sc[7] = t1.w;
        }
        if (p.shift) {
          const float4 t0 = *(const float4*)(p.shift + cch), t1 = *(const float4*)(p.shift + cch + 4);
          sh[0] = t0.x; sh[1] = t0.y; sh[2] = t0.z; sh[3] = t0.w; sh[4] = t1.x; sh[5] = t1.y; sh[6] = t1.z; sh[7] = t1.w;
        }
      }
      u32x4_t rr[4], mm[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // the residual / multiplier rows of two 16-pixel blocks are requested together, ahead of the arithmetic
        if ((b & 1) == 0) {
          if (p.res) {
#pragma unroll
            for (int bb = b; bb < b + 2; ++bb) {
              rr[bb] = (u32x4_t){0u, 0u, 0u, 0u};
              if (bb * 16 + lrow < nvw && cok) rr[bb] = *(const u32x4_t*)((const bf16_t*)p.res + (size_t)(res_base + bb * 16 + lrow) * p.ldr + cch);
            }
          }
          if (p.aux_mode == 2) {
#pragma unroll
            for (int bb = b; bb < b + 2; ++bb) {
              mm[bb] = (u32x4_t){0u, 0u, 0u, 0u};
              if (bb * 16 + lrow < nvw && cok) mm[bb] = *(const u32x4_t*)((const bf16_t*)p.aux + (size_t)(p_wave + bb * 16 + lrow) * p.ld_aux + cch);
            }
          }
        }
        const int prow = b * 16 + lrow, pix = p_wave + prow;
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * q][b][i]), __float_as_uint(acc[2 * q + 1][b][i]), false, false);
          v[i] = __uint_as_float(sw[0]);
          v[4 + i] = __uint_as_float(sw[1]);
        }
        if (!(prow < nvw && cok)) continue;
        if (p.nbias) {
          const float* nb = p.nbias + (size_t)fast_div(pix, p.div_hw_m, p.div_hw_s) * p.Cout + cch;
          const float4 n0 = *(const float4*)nb, n1 = *(const float4*)(nb + 4);
          v[0] += n0.x; v[1] += n0.y; v[2] += n0.z; v[3] += n0.w; v[4] += n1.x; v[5] += n1.y; v[6] += n1.z; v[7] += n1.w;
        }
        if (has_ss) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __fadd_rn(__fmul_rn(v[e], sc[e]), sh[e]);   // two roundings, as every other epilogue path
        }
        if (p.aux_mode == 2) {   // d(pre) = d(hidden) * gelu'(pre)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= __uint_as_float(mm[b][e] << 16);
            v[2 * e + 1] *= __uint_as_float(mm[b][e] & 0xffff0000u);
          }
        }
        if (p.res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(rr[b][e] << 16);
            v[2 * e + 1] += __uint_as_float(rr[b][e] & 0xffff0000u);
          }
        }
        if (p.aux_mode == 1) {   // GELU forward: gelu'(t) goes to aux, the pre-activation is never stored
          float dgv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_and_grad(v[e], v[e], dgv[e]);
          u32x4_t d;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = pack2bf(dgv[2 * e], dgv[2 * e + 1]);
          __builtin_nontemporal_store(d, (u32x4_t*)((bf16_t*)p.aux + (size_t)pix * p.ld_aux + cch));
        } else {
          apply_act_vec<8>(v, p.act);
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        __builtin_nontemporal_store(o, (u32x4_t*)((bf16_t*)p.y + (size_t)pix * p.ldy + cch));
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };

  // ---------------------------------------------------------------------------------------------------------------
  // the K-tile stream
  // ---------------------------------------------------------------------------------------------------------------
  setup_tile(0);
  issue_ktile();
  // (the K loop of a tile as an INNER loop: whatever the register allocator has to park around the epilogue - the issue side's
  // descriptors of the next tile are live across it - is then spilled and reloaded once per output tile, not once per K tile)
  for (int t = 0; t < my_tiles; ++t) {
    for (int k = 0; k < p.iters; ++k) {
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
      __builtin_amdgcn_s_barrier();
      issue_ktile();
      multiply_ktile();
      cmp_buf ^= 1;
    }
    if constexpr ((DBG & 64) == 0) epilogue(t);
  }
}

template <int DBG>
static hipError_t launch_big16(const IgemmParams& p, int nblk, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_big16_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  IgemmParams q = p;
  q.nblk = nblk;
  const int grid = nblk > 256 ? 256 : nblk;
  igemm_big16_kernel<DBG><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
  return hipGetLastError();
}

hipError_t cavp_launch_igemm_big16(const IgemmParams& p, int nblk, hipStream_t s) {
  switch (p.dbg) {
    case 0: return launch_big16<0>(p, nblk, s);
#ifdef CAVP_PROFILE
    case 8: return launch_big16<8>(p, nblk, s);
    case 16: return launch_big16<16>(p, nblk, s);
    case 32: return launch_big16<32>(p, nblk, s);
    case 64: return launch_big16<64>(p, nblk, s);
    case 72: return launch_big16<72>(p, nblk, s);
    case 120: return launch_big16<120>(p, nblk, s);
#endif
    default: return hipErrorInvalidValue;
  }
}
