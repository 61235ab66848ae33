// 256 x 256 implicit-GEMM tile for the large CAVP layers (bf16, gfx950): one 8-wave workgroup per CU, 2-stage LDS-DMA ring.
//
// Why a second kernel: the 4-wave 128x128 tile of conv_igemm.hip moves 64 flop per operand byte through the LDS-DMA path
// (~60 GB/s per CU), which caps it near 1 PF/s however its K loop is scheduled (profiles/r01_notes.md).  A 256x256 tile
// needs half the operand bytes per flop; its 128 accumulator registers per lane leave room for only ONE workgroup per CU.
//
//  * 8 waves = 4 (channel) x 2 (pixel); wave tile 64 channels x 128 pixels = 4 x 8 MFMA 16x16x32 blocks, multiplied as four
//    quadrants (c0,p0) (c1,p0) (c1,p1) (c0,p1) of 16 MFMAs whose fragment reads the compiler interleaves with the MFMAs.
//  * a K tile (64 channels of one tap) = four 16 KiB half tiles in LDS: H0 = weight rows of the low channel halves, H1 = pixels of
//    the low pixel halves, H2 / H3 = the high halves; every thread issues the two 16-byte LDS-DMA pieces of each.
//  * the K loop is a plain 2-stage ring (128 KiB): iteration u waits for ITS pieces of K tile u (vmcnt(0)), ONE s_barrier
//    (everybody's pieces landed, everybody is done multiplying K tile u - 1), issues the 8 pieces of K tile u + 1 into the stage
//    K tile u - 1 vacated, multiplies K tile u.  The two waves of a SIMD cover each other's fragment reads and MFMAs by themselves.
//    Rounds 2-5 ran this tile as a hand-built ping-pong of two wave groups (4 phases and 8 barriers per K tile, DMA 6 phases ahead
//    behind counted vmcnt); round 6's micro-benchmark (tools/microbench/kloop_rega.hip, profiles/r06_notes.md 1) showed the plain
//    ring at or above it and the product A/B agreed (bit-identical outputs; 1x1 / linear launches -5 .. -7 %, 3x3 -0 .. -2 %, step
//    -0.04 ms): the ping-pong schedule, its 32x32x16 variant (4 .. 6 % slower in both schedules) and a 16-wave variant of the ring
//    (faster in the micro-benchmark, 10 .. 19 % slower as a product kernel: no registers for double-buffered fragments next to the
//    conv's DMA descriptors at 128 VGPRs) were measured and removed.
//  * the K-tile stream runs ACROSS the output tiles of the persistent workgroup: the loads of the next tile's first K
//    tile are in flight while the epilogue of the current tile runs.
//  * epilogue per WAVE, no workgroup barrier: 16-pixel x 64-channel f32 blocks go through a private 4 KiB LDS scratch (the
//    last 32 KiB of the 160 KiB) and leave as 16-byte stores, 128 contiguous bytes per pixel; BatchNorm statistics come
//    from the accumulators in registers (one 128-row statistics tile per wave slab).
//
// Same operand layout as conv_igemm.hip: weights = MFMA A (rows = output channels), gathered activations = MFMA B, 128-byte
// LDS rows with the 16-byte slot index XOR-swizzled by (row >> 1) & 7 (applied to the SOURCE address of the DMA),
// padding / K tails / M and Cout tails zero-filled by the buffer descriptor's bounds check.
#include <type_traits>

#include "igemm_params.h"

// A loaded value is "used" in the block that loads it: hipcc's wait-count pass then retires the load there.  A load still
// pending (from the compiler's point of view) on ANY path into the phase loop costs an s_waitcnt vmcnt(0) at the loop
// header, i.e. drains the LDS-DMA ring in every K tile.
#define CAVP_USE8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

namespace {

constexpr int BC = 256, BP = 256, NT = 512;
constexpr int HALF_BYTES = 128 * 128;       // 128 rows x 128 B
constexpr int BUF_BYTES = 4 * HALF_BYTES;   // one K tile: H0 H1 H2 H3
constexpr int RING_BYTES = 2 * BUF_BYTES;
constexpr int SCR_BYTES = 4096;             // per-wave epilogue scratch: 16 pixels x 64 channels f32
constexpr int LDS_BYTES = RING_BYTES + 8 * SCR_BYTES;
constexpr unsigned kOOB = 0x80000000u;
static_assert(LDS_BYTES == 160 * 1024, "the whole LDS of a CU");

typedef __attribute__((address_space(3))) void* lds_ptr_t;

}  // namespace

// DBG: compile-time profiling switches (instantiated only under -DCAVP_PROFILE; a RUN-time test of such a flag inside the
// K loop splits its basic blocks): 1 taps outermost, 8 no DMA, 16 no MFMA, 32 no fragment reads, 64 no epilogue.
template <int DBG>
__global__ __launch_bounds__(512, 2) void igemm_big_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cavp_prefetch_kernargs<(int)sizeof(IgemmParams)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, wp = wave >> 2;   // wp = wave group
  const int lrow = lane & 15, lgrp = lane >> 4;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int my_tiles = ((int)blockIdx.x < p.nblk) ? (p.nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_tiles == 0) return;
  const int HoWo = p.Ho * p.Wo;

  // ---------------------------------------------------------------------------------------------------------------
  // issue side: per-thread DMA descriptors of the tile whose K tiles are being fetched
  // ---------------------------------------------------------------------------------------------------------------
  // a wave DMA instruction writes 8 LDS rows (1 KiB) linearly; thread -> LDS row r0 (+64 for its second piece) of a half tile
  const int r0 = 8 * wave + (lane >> 3);
  const int kslot = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * 8;   // (r0 >> 1) & 7 == ((r0 + 64) >> 1) & 7
  const int kbyte = kslot * 2;
  unsigned w_off[4];   // [channel half * 2 + piece]
  unsigned x_off[4];   // [pixel half * 2 + piece]
  unsigned x_mask[4];
  int iss_tile = 0;            // ordinal of the tile being fetched (my_tiles: past the end)
  int iss_ti = 0, iss_cc = 0;  // tap index, channel tile
  int iss_buf = 0;
  // byte offsets of the current tap in a weight row / in x: re-loaded (scalar loads from the kernel arguments) when the tap
  // advances, i.e. a whole phase before their next use - a scalar load waited for inside a phase would also drain the
  // fragment reads issued before it (one lgkmcnt for both)
  int iss_woff = p.tap_woff[0], iss_xoff = p.tap_xoff[0];

  auto setup_tile = [&](int ord) {
    const int vb = (int)blockIdx.x + ord * (int)gridDim.x;
    const int sid = xcd_remap(vb, p.nblk);
    const int tp = fast_div(sid, p.div_tc_m, p.div_tc_s), tc = sid - tp * p.tiles_c;
    const int c_base = tc * BC, p_base = tp * BP;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = r0 + 64 * (m & 1), half = m >> 1;
      const int c = c_base + (r >> 5) * 64 + half * 32 + (r & 31);
      w_off[m] = c < p.Cout ? (unsigned)(((size_t)c * p.K + kslot) * 2) : kOOB;
    }
    int h0[4], w0[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = r0 + 64 * (m & 1), half = m >> 1;
      const int pix = p_base + (r >> 6) * 128 + half * 64 + (r & 63);
      const bool ok = pix < p.M;
      const int pp = ok ? pix : 0;
      const int n = fast_div(pp, p.div_hw_m, p.div_hw_s), rr = pp - n * HoWo;
      const int ho = fast_div(rr, p.div_w_m, p.div_w_s), wo = rr - ho * p.Wo;
      h0[m] = ok ? ho * p.stride - p.pad : -0x10000000;
      w0[m] = wo * p.stride_w - p.pad;
      x_mask[m] = 0;
      x_off[m] = (unsigned)((n * p.H + h0[m]) * p.W + w0[m]) * (unsigned)(p.ldx * 2) + (unsigned)kbyte;
    }
    for (int t = 0; t < p.ntaps; ++t) {
      const int dh = p.tap_dh[t], dw = p.tap_dw[t];
#pragma unroll
      for (int m = 0; m < 4; ++m)
        x_mask[m] |= ((unsigned)(h0[m] + dh) < (unsigned)p.H && (unsigned)(w0[m] + dw) < (unsigned)p.W) ? (1u << t) : 0u;
    }
  };

  // the two DMA pieces of half tile K (0 = H0 weights low, 1 = H1 pixels low, 2 = H2 weights high, 3 = H3 pixels high) of
  // the current issue position; past the last tile the same instructions are issued with every lane out of range (zero
  // fill into a region nobody reads any more), so the vmcnt arithmetic stays uniform
  auto issue_half = [&](auto kc) {
    constexpr int K = decltype(kc)::value;
    const bool live = iss_tile < my_tiles;
    const int ti = iss_ti, c0 = iss_cc * 64;
    const unsigned oobm = (live && (c0 + kslot) < p.Cin) ? 0u : kOOB;
    char* base = smem + iss_buf * BUF_BYTES + K * HALF_BYTES + wave * 1024;
    if constexpr ((DBG & 8) != 0) {
    } else if constexpr ((K & 1) == 0) {
      const unsigned wk = (unsigned)(iss_woff + c0 * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned off = (w_off[(K >> 1) * 2 + j] + wk) | oobm;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(base + j * 8192), 16, (int)off, 0, 0, 0);
      }
    } else {
      const unsigned xk = (unsigned)(iss_xoff + c0 * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = (K >> 1) * 2 + j;
        const unsigned tapm = ((~(x_mask[m] >> ti)) & 1u) << 31;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(base + j * 8192), 16, (int)((x_off[m] + xk) | tapm | oobm), 0, 0,
                                                 0);
      }
    }
    if constexpr (K == 3) {   // next K tile of the stream
      iss_buf ^= 1;
      if (live) {
        if constexpr ((DBG & 1) != 0) {   // A/B: taps outermost (the 2-stage kernel's order)
          if (++iss_cc == p.cpt) {
            iss_cc = 0;
            if (++iss_ti == p.ntaps) {
              iss_ti = 0;
              if (++iss_tile < my_tiles) setup_tile(iss_tile);
            }
            iss_woff = p.tap_woff[iss_ti];
            iss_xoff = p.tap_xoff[iss_ti];
          }
        } else {
          // taps INNERMOST: the 9 shifted windows of one 64-channel slice are fetched back to back, so 8 of the 9 reads of
          // an input line hit the XCD's L2 (with taps outermost a line is re-read after a whole pass over the channels,
          // by which time the 32 tiles in flight on an XCD have pushed it out of the 4 MiB L2)
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
      }
    }
  };

  // ---------------------------------------------------------------------------------------------------------------
  // compute side
  // ---------------------------------------------------------------------------------------------------------------
  // acc[a][b] = 16-channel block a (4) x 16-pixel block b (8), 4 channels per lane
  f32x4_t acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t fa0[2][2], fa1[2][2], fb[4][2];   // [block][k sub-step]

  const int key = (lrow >> 1) & 7;
  // byte offset of (row, slot) inside a half tile for k sub-step 0; sub-step 1 flips slot bit 2
  const int a_off = (wc * 32 + lrow) * 128 + ((lgrp ^ key) << 4);
  const int b_off = (wp * 64 + lrow) * 128 + ((lgrp ^ key) << 4);
  int cmp_buf = 0, cmp_k = 0, cmp_tile = 0;

  auto read_a = [&](u32x4_t (&f)[2][2], int half) {
    const char* base = smem + cmp_buf * BUF_BYTES + (half ? 2 : 0) * HALF_BYTES + a_off;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f[a][0] = *(const u32x4_t*)(base + a * 2048);
      f[a][1] = *(const u32x4_t*)(base + a * 2048 + ((((lgrp ^ key) ^ 4) - (lgrp ^ key)) << 4));
    }
  };
  auto read_b = [&](int half) {
    const char* base = smem + cmp_buf * BUF_BYTES + (half ? 3 : 1) * HALF_BYTES + b_off;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      fb[b][0] = *(const u32x4_t*)(base + b * 2048);
      fb[b][1] = *(const u32x4_t*)(base + b * 2048 + ((((lgrp ^ key) ^ 4) - (lgrp ^ key)) << 4));
    }
  };
  auto mma_quadrant = [&](const u32x4_t (&f)[2][2], auto hc, auto hq) {
    constexpr int HC = decltype(hc)::value, HQ = decltype(hq)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Mma<bf16_t>::run(acc[HC * 2 + a][HQ * 4 + b], f[a][j], fb[b][j]);
  };

  // ---- epilogue of one wave: acc (64 channels x 128 pixels) -> y ----
  auto epilogue = [&](int ord) {
    const int vb = (int)blockIdx.x + ord * (int)gridDim.x;
    const int sid = xcd_remap(vb, p.nblk);
    const int tp = fast_div(sid, p.div_tc_m, p.div_tc_s), tc = sid - tp * p.tiles_c;
    const int c_wave = tc * BC + wc * 64, p_wave = tp * BP + wp * 128;
    const int nvw = p.M - p_wave;   // valid pixel rows of this wave's slab (<= 0: none)
    if (p.tile_stats) {
      // per-channel (mean, M2) of this wave's <= 128 rows: one statistics tile per wave slab (see conv_igemm.hip)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float s1[4], s2[4], x0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x0[i] = row16_first(acc[a][0][i]);
          s1[i] = 0.f; s2[i] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
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
        const int c = c_wave + a * 16 + lgrp * 4;
        if (lrow == 0 && nvw > 0 && c < p.Cout) {
          const float inv_n = __frcp_rn((float)(nvw < 128 ? nvw : 128));
          float4 o;
          float m = s1[0] * inv_n; o.x = x0[0] + m; o.y = fmaxf(s2[0] - s1[0] * m, 0.f);
          m = s1[1] * inv_n; o.z = x0[1] + m; o.w = fmaxf(s2[1] - s1[1] * m, 0.f);
          float4 o2;
          m = s1[2] * inv_n; o2.x = x0[2] + m; o2.y = fmaxf(s2[2] - s1[2] * m, 0.f);
          m = s1[3] * inv_n; o2.z = x0[3] + m; o2.w = fmaxf(s2[3] - s1[3] * m, 0.f);
          float* dst = p.tile_stats + ((size_t)(tp * 2 + wp) * p.Cout + c) * 2;
          *(float4*)dst = o;
          *(float4*)(dst + 4) = o2;
        }
      }
    }
    // the lane finishes channels ec .. ec+7 of pixels (lane >> 3) and (lane >> 3) + 8 of every 16-pixel block
    float* scr = (float*)(smem + RING_BYTES + wave * SCR_BYTES);
    const int cg = lane & 7, ec = c_wave + cg * 8, pr = lane >> 3;
    const bool ec_ok = ec < p.Cout;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (ec_ok) {
      if (p.scale) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 t = *(const float4*)(p.scale + ec + 4 * q);
          sc[4 * q] = t.x; sc[4 * q + 1] = t.y; sc[4 * q + 2] = t.z; sc[4 * q + 3] = t.w;
        }
        CAVP_USE8(sc);
      }
      if (p.shift) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 t = *(const float4*)(p.shift + ec + 4 * q);
          sh[4 * q] = t.x; sh[4 * q + 1] = t.y; sh[4 * q + 2] = t.z; sh[4 * q + 3] = t.w;
        }
        CAVP_USE8(sh);
      }
    }
    const bool has_ss = p.scale != nullptr || p.shift != nullptr;
    bf16_t* yp = (bf16_t*)p.y + (size_t)(p_wave + pr) * p.ldy + ec;
    // (res_rows is a multiple of 256: a wave slab never straddles the wrap)
    const bf16_t* rp = p.res ? (const bf16_t*)p.res + (size_t)((p.res_rows ? p_wave % p.res_rows : p_wave) + pr) * p.ldr + ec : nullptr;
    bf16_t* xp = p.aux_mode ? (bf16_t*)p.aux + (size_t)(p_wave + pr) * p.ld_aux + ec : nullptr;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      u32x4_t rr[2], mm[2];
      if (p.aux_mode == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          mm[k] = (u32x4_t){0u, 0u, 0u, 0u};
          if (ec_ok && b * 16 + pr + 8 * k < nvw) {
            mm[k] = *(const u32x4_t*)(xp + (size_t)(b * 16 + 8 * k) * p.ld_aux);
            asm volatile("" : "+v"(mm[k]));   // (see CAVP_USE8)
          }
        }
      }
      if (rp) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          rr[k] = (u32x4_t){0u, 0u, 0u, 0u};
          if (ec_ok && b * 16 + pr + 8 * k < nvw) {
            rr[k] = *(const u32x4_t*)(rp + (size_t)(b * 16 + 8 * k) * p.ldr);
            asm volatile("" : "+v"(rr[k]));   // (see CAVP_USE8)
          }
        }
      }
      // scratch [16 pixels][16 slots of 4 channels], slot index XOR pixel: conflict-free 16-byte writes and reads
#pragma unroll
      for (int a = 0; a < 4; ++a) *(f32x4_t*)(scr + lrow * 64 + (((a * 4 + lgrp) ^ lrow) << 2)) = acc[a][b];
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are done before its reads
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = pr + 8 * k;
        float v[8];
        {
          const f32x4_t t0 = *(const f32x4_t*)(scr + px * 64 + (((2 * cg) ^ px) << 2));
          const f32x4_t t1 = *(const f32x4_t*)(scr + px * 64 + (((2 * cg + 1) ^ px) << 2));
          v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];
          v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
        }
        const int row = b * 16 + px;
        if (!(ec_ok && row < nvw)) continue;
        if (p.nbias) {
          const float* nb = p.nbias + (size_t)fast_div(p_wave + row, p.div_hw_m, p.div_hw_s) * p.Cout + ec;
          float nbv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) nbv[e] = nb[e];
          CAVP_USE8(nbv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += nbv[e];
        }
        if (has_ss) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __fadd_rn(__fmul_rn(v[e], sc[e]), sh[e]);
        }
        if (p.aux_mode == 2) {   // d(pre) = d(hidden) * gelu'(pre)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= __uint_as_float(mm[k][e] << 16);
            v[2 * e + 1] *= __uint_as_float(mm[k][e] & 0xffff0000u);
          }
        }
        if (rp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(rr[k][e] << 16);
            v[2 * e + 1] += __uint_as_float(rr[k][e] & 0xffff0000u);
          }
        }
        if (p.aux_mode == 1) {   // GELU forward: gelu'(t) goes to aux, the pre-activation is never stored
          float dgv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_and_grad(v[e], v[e], dgv[e]);
          u32x4_t d;
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = pack2bf(dgv[2 * e], dgv[2 * e + 1]);
          __builtin_nontemporal_store(d, (u32x4_t*)(xp + (size_t)(b * 16 + 8 * k) * p.ld_aux));
        } else {
          apply_act_vec<8, true>(v, p.act);
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        // non-temporal: the tile's 128 KiB per output tensor leave the CU as a stream (same-box A/B 15.467 -> 15.417 ms per step;
        // restricting it to outputs beyond the 256 MiB Infinity Cache - the token MLP's hidden tensors - measured slower)
        __builtin_nontemporal_store(o, (u32x4_t*)(yp + (size_t)(b * 16 + 8 * k) * p.ldy));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);   // reads of this block retired before the next block overwrites the scratch
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };

  // ---------------------------------------------------------------------------------------------------------------
  // the K-tile stream
  // ---------------------------------------------------------------------------------------------------------------
  setup_tile(0);
  const int total_k = my_tiles * p.iters;
  issue_half(std::integral_constant<int, 0>{});
  issue_half(std::integral_constant<int, 1>{});
  issue_half(std::integral_constant<int, 2>{});
  issue_half(std::integral_constant<int, 3>{});
  for (int u = 0; u < total_k; ++u) {
    // my pieces of K tile u landed; behind the barrier everybody's did, and everybody has finished multiplying K tile u - 1, whose stage the
    // pieces of K tile u + 1 (the next output tile's first one behind a tile's last) now overwrite
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    __builtin_amdgcn_s_barrier();
    issue_half(std::integral_constant<int, 0>{});
    issue_half(std::integral_constant<int, 1>{});
    issue_half(std::integral_constant<int, 2>{});
    issue_half(std::integral_constant<int, 3>{});
    if constexpr (!(DBG & 32)) { read_b(0); read_a(fa0, 0); }
    if constexpr (!(DBG & 16)) mma_quadrant(fa0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});   // (c0, p0): H0 and H1
    if constexpr (!(DBG & 32)) read_a(fa1, 1);
    if constexpr (!(DBG & 16)) mma_quadrant(fa1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});   // (c1, p0): H2
    if constexpr (!(DBG & 32)) read_b(1);
    if constexpr (!(DBG & 16)) {
      mma_quadrant(fa1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});                            // (c1, p1): H3
      mma_quadrant(fa0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});                            // (c0, p1)
    }
    cmp_buf ^= 1;
    if (++cmp_k == p.iters) {
      cmp_k = 0;
      if constexpr (!(DBG & 64)) epilogue(cmp_tile);
      ++cmp_tile;
    }
  }
}

template <int DBG>
static hipError_t launch_big(const IgemmParams& p, int nblk, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_big_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  IgemmParams q = p;
  q.nblk = nblk;
  const int grid = nblk > 256 ? 256 : nblk;
  igemm_big_kernel<DBG><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
  return hipGetLastError();
}

hipError_t cavp_launch_igemm_big(const IgemmParams& p, int nblk, hipStream_t s) {
  switch (p.dbg) {
    case 0: return launch_big<0>(p, nblk, s);
#ifdef CAVP_PROFILE
    case 1: return launch_big<1>(p, nblk, s);
    case 8: return launch_big<8>(p, nblk, s);
    case 16: return launch_big<16>(p, nblk, s);
    case 32: return launch_big<32>(p, nblk, s);
    case 64: return launch_big<64>(p, nblk, s);
    case 72: return launch_big<72>(p, nblk, s);
    case 120: return launch_big<120>(p, nblk, s);
#endif
    default: return hipErrorInvalidValue;
  }
}
