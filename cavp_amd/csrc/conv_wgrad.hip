// Weight gradient of conv / linear for gfx950:  dW[co][tap][ci] += sum_pix dY[pix][co] * X[pix @ tap][ci]
//
// A "TN" GEMM whose reduction dimension (pixels) is the SLOW dimension of both NHWC operands.  Design:
// * One workgroup = (ci tile x co tile) of one live tap and one slice of the pixel range (split-K).  With one slice
//   the workgroup owns its tile and adds it to dW directly; otherwise every slice stores a private f32 slab and a
//   second kernel reduces the slabs in a fixed order (deterministic; f32 atomics on a 32 K-element dW from 512
//   workgroups were 2.3x slower - profiles/r01_notes.md).
// * Both operands are streamed global -> LDS with LDS-DMA (buffer_load ... lds) exactly as they lie in memory:
//   32 pixel rows x 256 bytes of channels per stage (128 bf16 / 64 f32 channels), two stages = 32 KiB per workgroup,
//   four workgroups per CU (64-row stages are kept behind CAVP_WGRAD_BK=64 for A/B runs).  Rows that fall
//   outside the image (padding taps), beyond the pixel range or beyond Cin/Cout are zero-filled by the buffer
//   descriptor's bounds check.
// * bf16: the MFMA 16x16x32 fragment (8 consecutive k = pixels for one channel) is gathered with the gfx950 LDS
//   transpose read ds_read_b64_tr_b16: within a 16-lane group lane s passes the address of row r0 + (s >> 2),
//   channels c0 + 4 (s & 3) .. +3 and lane i receives channel c0 + i of rows r0 .. r0+3 (probed on hardware:
//   profiles/r01_ds_read_b64_tr_b16_probe.txt).  Two reads give the 8 k values.  The 32-byte chunk index of a row
//   is XOR-swizzled with (row & 3) | ((row >> 3) & 1) << 2 (on the DMA source side) so the 8 rows of a 32-lane
//   group fall into 8 different bank groups.
// * f32: v_mfma_f32_16x16x4_f32 fragments are plain ds_read_b32 (lanes 0-15 = 16 consecutive channels of one row).
// * X (rows = ci) is the MFMA A operand and dY (cols = co) the B operand, so a lane's 4 accumulators are 4
//   consecutive ci of one co = 16 contiguous bytes of dW.
#include <stdlib.h>

#include "wgrad_params.h"

// One logical workgroup `bid` of one weight gradient (shared by the single and the grouped launch).
// BK = pixel rows per stage; BIAS: also the bias gradient (column sums of dY); NSTG = LDS stages: 2 (BK = 32: four 32 KiB
// workgroups per CU, BK = 64: two 64 KiB ones) or 1 (BK = 64: ONE 32 KiB stage, four workgroups per CU - load, wait, barrier,
// multiply two k steps, barrier; nothing overlaps inside a workgroup, the co-resident ones fill the gaps.  In the K-loop
// micro-benchmark tools/microbench/kloop.hip that shape beats two double-buffered workgroups by 26 %)
template <typename T, int BK, bool BIAS, int NSTG = 2>
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, const int bid, char* smem) {
  constexpr int ES = (int)sizeof(T);
  constexpr int TCH = 256 / ES;      // channels per tile row (256 bytes)
  constexpr int NI = BK / 16;        // DMA instructions per operand per thread and stage
  constexpr int STAGE = 2 * BK * 256;  // X tile + dY tile
  constexpr int WT = TCH / 2;        // per-wave tile edge (2 x 2 waves)
  constexpr int MB = WT / 16;        // 16x16 blocks per wave edge
  constexpr unsigned kOOB = 0x80000000u;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: LDS-DMA bases via SALU)

  // (multiply + shift: integer division is a ~40-instruction VALU sequence even for wave-uniform values)
  const int b1 = fast_div(bid, p.dv_co[0], p.dv_co[1]), tco = bid - b1 * p.tiles_co;
  const int b2 = fast_div(b1, p.dv_ci[0], p.dv_ci[1]), tci = b1 - b2 * p.tiles_ci;
  const int z = fast_div(b2, p.dv_nt[0], p.dv_nt[1]), ti = b2 - z * p.ntaps;
  const int tap = (int)((p.taps >> (4 * ti)) & 15ull);
  const int kh = fast_div(tap, p.dv_kw[0], p.dv_kw[1]), kw = tap - kh * p.KW;
  const int co_base = tco * TCH, ci_base = tci * TCH;
  const int r_begin = z * p.rows_per_split;
  int r_end = r_begin + p.rows_per_split;
  if (r_end > p.M) r_end = p.M;

  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);

  // DMA geometry: wave-instruction g (= wave + 4 i, i < 4) fills rows 4g .. 4g+3 of a tile; lane l lands at byte
  // 16 l of that 1 KiB, i.e. row 4g + (l >> 4), physical 16-byte slot l & 15.  The swizzle key of row drow0 + 16 i
  // does not depend on i, so the lane's channel offset is fixed; the pixel coordinates of its 4 rows are advanced
  // incrementally (+64 pixels per K tile) instead of being re-derived with integer divisions every iteration.
  const int drow0 = 4 * wave + (lane >> 4);  // + 16 i
  const int HoWo = p.Ho * p.Wo;
  auto swz_key = [](int row) { return (row & 3) | (((row >> 3) & 1) << 2); };
  const int cel = ((((lane & 15) >> 1) ^ swz_key(drow0)) * 32 + (lane & 1) * 16) / ES;  // logical channel of this lane
  const bool ci_ok = ci_base + cel < p.Cin, co_ok = co_base + cel < p.Cout;
  const unsigned xcb = (unsigned)((ci_base + cel) * ES), ycb = (unsigned)((co_base + cel) * ES);
  const bool pointwise = (p.ntaps_all == 1) && p.stride == 1 && p.pad == 0;
  // Per DMA row: output pixel index, its (ho, wo), the input coordinates of the current tap and the byte offsets into x / dY.
  // Everything is advanced INCREMENTALLY by one stage (BK pixels): the issue path of a stage is adds, compares and selects - no
  // integer multiply (quarter rate; re-deriving the offsets cost 6 v_mul_lo per stage and more: a multiply-shift re-split of
  // the pixel index, tried instead of the wrap loops below, made the training step 0.5 ms SLOWER) and no scalar load: `p` may
  // live in the kernel-argument segment behind a run-time job index (the grouped launch), and left to itself hipcc re-loaded
  // every field inside the stage loop (9 s_load + s_waitcnt lgkmcnt(0) pairs per stage).  readfirstlane results stay in SGPRs.
  const int dh = kh * p.dil - p.pad, dw = kw * p.dil - p.pad;
  const int gH = __builtin_amdgcn_readfirstlane(p.H), gW = __builtin_amdgcn_readfirstlane(p.W);
  const int gWo = __builtin_amdgcn_readfirstlane(p.Wo), gHo = __builtin_amdgcn_readfirstlane(p.Ho);
  const int gStride = __builtin_amdgcn_readfirstlane(p.stride);
  const unsigned gLdx = (unsigned)__builtin_amdgcn_readfirstlane(p.ldx * ES), gLdy = (unsigned)__builtin_amdgcn_readfirstlane(p.ldy * ES);
  const int gWoS = gWo * gStride, gHoS = gHo * gStride;
  const unsigned stepY = (unsigned)BK * gLdy;
  const unsigned stepX = (unsigned)(BK * (pointwise ? 1 : gStride)) * gLdx;          // BK pixels to the right
  const unsigned stepRow = (unsigned)((gW - gWo) * gStride) * gLdx;                   // wrap to the next output row
  const unsigned stepImg = (unsigned)((gH - gHoS) * gW) * gLdx;                       // wrap to the next image
  // state per row: input coordinates of the tap (which also tell when the output row / image wraps) and the two byte offsets;
  // the pixel-range end is compared on the dY offset (yEnd), so the pixel index itself is not carried
  int hin[NI], win[NI];
  unsigned xo[NI], yo[NI];
  const unsigned yEnd = (unsigned)r_end * gLdy + ycb;
  const int hWrap = gHoS + dh, wWrap = gWoS + dw;   // first input row / column past the last output row / column
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = r_begin + drow0 + 16 * i;
    yo[i] = (unsigned)pix * gLdy + ycb;
    const int pp = pix < p.M ? pix : 0;
    const int n = fast_div(pp, p.dv_hw[0], p.dv_hw[1]);
    const int rr = pp - n * HoWo;
    const int ho = fast_div(rr, p.dv_w[0], p.dv_w[1]);
    hin[i] = ho * gStride + dh;
    win[i] = (rr - ho * p.Wo) * gStride + dw;
    xo[i] = pointwise ? (unsigned)pix * gLdx + xcb : (unsigned)((n * gH + hin[i]) * gW + win[i]) * gLdx + xcb;
  }
  const bool dbg_nodma = CAVP_DBG(p, 1);
  auto gdma = [&](int buf) {
    char* base = smem + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool pok = yo[i] < yEnd && !dbg_nodma;
      bool xok = pok && ci_ok;
      if (!pointwise) xok = xok && ((unsigned)hin[i] < (unsigned)gH) && ((unsigned)win[i] < (unsigned)gW);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + i * 4096), 16,
                                               (int)(xok ? xo[i] : kOOB), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc,
                                               (__attribute__((address_space(3))) void*)(base + BK * 256 + i * 4096), 16,
                                               (int)((pok && co_ok) ? yo[i] : kOOB), 0, 0, 0);
      // advance this row by one stage
      yo[i] += stepY;
      xo[i] += stepX;
      if (!pointwise) {
        win[i] += BK * gStride;
        while (win[i] >= wWrap) { win[i] -= gWoS; xo[i] += stepRow; hin[i] += gStride; }
        while (hin[i] >= hWrap) { hin[i] -= gHoS; xo[i] += stepImg; }
      }
    }
  };

  const int wci0 = (wave & 1) * WT, wco0 = (wave >> 1) * WT;  // wave's sub-tile origin (channels)
  const int lrow = lane & 15, lgrp = lane >> 4;

  f32x4_t acc[MB][MB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // bias gradient: the workgroups of the first ci tile and first live tap see every dY element exactly once; their
  // waves with wci0 == 0 add up the dY fragments they load for the MFMAs anyway (no extra pass over dY).
  const bool do_bias = BIAS && p.dbias != nullptr && tci == 0 && ti == 0 && (wave & 1) == 0;
  float bsum[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) bsum[b] = 0.f;

  // 16 x 16 blocks of this wave's 64 x 64 sub-tile that hold real channels: a 304-channel operand fills 2.375 tiles of 128, and the
  // last tile's zero-filled blocks (5 of its 8 block rows) were multiplied like any other - 21 % of the MFMAs of a 304-wide
  // dimension, 37 % of a 304 x 304 weight.  Dead blocks are skipped (wave-uniform), their accumulators stay zero.
  const int na = __builtin_amdgcn_readfirstlane(min(MB, max(0, (p.Cin - ci_base - wci0 + 15) >> 4)));
  const int nb = __builtin_amdgcn_readfirstlane(min(MB, max(0, (p.Cout - co_base - wco0 + 15) >> 4)));

  auto compute = [&](int buf) {
    const char* xb = smem + buf * STAGE;
    const char* yb = xb + BK * 256;
    if constexpr (ES == 2) {
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        // rows this lane ADDRESSES: r0 + (s >> 2) with s = lane & 15; r0 = ks*32 + 8*lgrp (+4 for the second read)
        const int ra = ks * 32 + 8 * lgrp + (lrow >> 2);
        const int rb = ra + 4;
        const int ka = swz_key(ra), kb = swz_key(rb);
        const int sub = (lrow & 3) * 8;
        u32x4_t af[MB], bfv[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) {
          const int ch = (wci0 >> 4) + a;  // 32-byte chunk = 16 bf16 channels
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xb + ra * 256 + ((ch ^ ka) << 5) + sub));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xb + rb * 256 + ((ch ^ kb) << 5) + sub));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          af[a] = (u32x4_t){l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          const int ch = (wco0 >> 4) + b;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yb + ra * 256 + ((ch ^ ka) << 5) + sub));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yb + rb * 256 + ((ch ^ kb) << 5) + sub));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          bfv[b] = (u32x4_t){l2.x, l2.y, h2.x, h2.y};
          if (BIAS && do_bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              bsum[b] += __uint_as_float(bfv[b][e] << 16) + __uint_as_float(bfv[b][e] & 0xffff0000u);
          }
        }
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < MB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]),
                                                                __builtin_bit_cast(bf16x8_t, bfv[b]), acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < BK / 4; ++ks) {
        const int r = ks * 4 + lgrp;  // k index of this lane
        const int key = swz_key(r);
        float af[MB], bfv[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) {
          const int cb = (wci0 + a * 16 + lrow) * 4;  // logical byte offset in the row
          af[a] = *(const float*)(xb + r * 256 + ((((cb >> 5) ^ key) << 5) | (cb & 31)));
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          const int cb = (wco0 + b * 16 + lrow) * 4;
          bfv[b] = *(const float*)(yb + r * 256 + ((((cb >> 5) ^ key) << 5) | (cb & 31)));
          if (BIAS && do_bias) bsum[b] += bfv[b];
        }
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < MB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bfv[b], acc[a][b], 0, 0, 0);
      }
    }
  };

  if constexpr (NSTG == 1) {
    for (int r0 = r_begin; r0 < r_end; r0 += BK) {
      if (r0 != r_begin) __builtin_amdgcn_s_barrier();   // everybody finished multiplying the previous stage
      gdma(0);
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
      __builtin_amdgcn_s_barrier();
      compute(0);
    }
  } else if (r_begin < r_end) {
    gdma(0);
    __syncthreads();
    int buf = 0;
    if constexpr (ES == 2 && BK == 32) {
      // (A 3-stage ring behind a counted vmcnt - transposed reads as inline asm so that hipcc does not drain it - was built twice
      // and measured: three 48 KiB workgroups per CU lose 0.5 .. 0.7 ms per step against four 32 KiB ones, profiles/r03_notes.md.)
      // One stage = one 32-row k step.  The fragments of the CURRENT stage are read before the next stage's DMA is issued:
      // hipcc cannot tell the ring's two buffers apart and puts an s_waitcnt vmcnt(0) in front of every LDS read that follows
      // a pending LDS-DMA in program order - with the DMA issued first (the generic loop below) the next tile was waited for
      // BEFORE the current one was multiplied, i.e. a workgroup never overlapped its own loads with its own MFMAs
      // (SQ_WAIT_ANY 58 % of the wave cycles, profiles/r03_pmc_wait_train_bf16.txt).
      for (int r0 = r_begin; r0 < r_end; r0 += BK) {
        const char* xb = smem + buf * STAGE;
        const char* yb = xb + BK * 256;
        const int ra = 8 * lgrp + (lrow >> 2), rb = ra + 4;
        const int ka = swz_key(ra), kb = swz_key(rb);
        const int sub = (lrow & 3) * 8;
        u32x4_t af[MB], bfv[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) {
          const int ch = (wci0 >> 4) + a;  // 32-byte chunk = 16 bf16 channels
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xb + ra * 256 + ((ch ^ ka) << 5) + sub));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xb + rb * 256 + ((ch ^ kb) << 5) + sub));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          af[a] = (u32x4_t){l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) {
          const int ch = (wco0 >> 4) + b;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yb + ra * 256 + ((ch ^ ka) << 5) + sub));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(yb + rb * 256 + ((ch ^ kb) << 5) + sub));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          bfv[b] = (u32x4_t){l2.x, l2.y, h2.x, h2.y};
        }
        if (r0 + BK < r_end) gdma(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (BIAS && do_bias) {
#pragma unroll
          for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              bsum[b] += __uint_as_float(bfv[b][e] << 16) + __uint_as_float(bfv[b][e] & 0xffff0000u);
        }
#pragma unroll
        for (int a = 0; a < MB; ++a) {
          if (a >= na) break;   // (wave-uniform)
#pragma unroll
          for (int b = 0; b < MB; ++b) {
            if (b >= nb) break;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[a]),
                                                                __builtin_bit_cast(bf16x8_t, bfv[b]), acc[a][b], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise hoists the wait + barrier above the MFMAs)
        // the next stage's LDS-DMA must have LANDED before any wave crosses the barrier and reads it: hipcc (ROCm 7.2) emits this
        // vmcnt(0) itself in front of the barrier, but nothing in the source required it - gfx950's workgroup release fence does
        // not - so it is stated here (same instruction, no timing change; round-3 advisor note)
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        __syncthreads();
        buf ^= 1;
      }
    } else {
      for (int r0 = r_begin; r0 < r_end; r0 += BK) {
        if (r0 + BK < r_end && !CAVP_DBG(p, 4)) gdma(buf ^ 1);
        if (!CAVP_DBG(p, 2)) compute(buf);
        __syncthreads();
        buf ^= 1;
      }
    }
  }

  // D[i = ci][j = co]: lane holds ci = 4*lgrp + {0..3} (rows) of co = lrow (col) in each 16x16 block = 16 contiguous
  // bytes of dW.  ksplit == 1: this workgroup owns the tile -> plain read-modify-write; otherwise plain stores into
  // this split's slab (reduced afterwards, deterministic, no atomics).
  if (CAVP_DBG(p, 8)) return;
  if (BIAS && do_bias) {   // lanes (lrow, lgrp) hold 4 disjoint pixel subsets of column co = wco0 + 16 b + lrow
    float* bo = p.ksplit > 1 ? p.bias_slabs + (size_t)z * p.Cout : p.dbias;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      float v = bsum[b];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int co = co_base + wco0 + b * 16 + lrow;
      if (lgrp == 0 && co < p.Cout) bo[co] = p.ksplit > 1 ? v : bo[co] + v;   // one writer per (split, co)
    }
  }
  float* out = p.ksplit > 1 ? p.slabs + (size_t)z * p.Cout * p.ntaps_all * p.Cin : p.dw;
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const int ci = ci_base + wci0 + a * 16 + lgrp * 4;
      const int co = co_base + wco0 + b * 16 + lrow;
      if (co < p.Cout && ci < p.Cin) {   // Cin % 4 == 0: a quad is in range as a whole
        if (p.ksplit == 1 && p.oihw) {   // straight into the torch-layout gradient: 4 strided read-modify-writes
          float* dst = out + ((size_t)co * p.Cin + ci) * p.ntaps_all + tap;
          if (p.overwrite) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * p.ntaps_all] = acc[a][b][e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * p.ntaps_all] += acc[a][b][e];
          }
        } else {
          float4* dst = (float4*)(out + ((size_t)co * p.ntaps_all + tap) * p.Cin + ci);
          float4 v = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
          if (p.ksplit == 1 && !p.overwrite) {
            const float4 o = *dst;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *dst = v;
        }
      }
    }
  }
}

template <typename T, int BK, bool BIAS, int NSTG = 2>
__global__ __launch_bounds__(256, (BK == 32 || NSTG == 1 ? 4 : 2)) void wgrad_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wgrad_tile<T, BK, BIAS, NSTG>(p, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Grouped launch: the logical workgroups of up to 16 weight gradients in one grid.  The backward of a stage of small layers
// (layer3: 19 convs on 6272 pixels) used to be 19 launches that each split their 98 pixel chunks ~8 ways to find 1024
// workgroups - 19 x (slab write + slab read + reduce launch); together the same layers fill the chip with 2 splits.
template <typename T, bool BIAS, int BK = 32, int NSTG = 2>
__global__ __launch_bounds__(256, 4) void wgrad_group_kernel(const WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cavp_prefetch_kernargs<(int)offsetof(WgradGroupArgs, job)>();
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int j = 0;
  while (j + 1 < g.njobs && bid >= g.blk_end[j]) ++j;
  cavp_prefetch_kernargs_at<(int)sizeof(WgradParams)>((int)offsetof(WgradGroupArgs, job) + j * (int)sizeof(WgradParams));
  wgrad_tile<T, BK, BIAS, NSTG>(g.job[j], bid - (j ? g.blk_end[j - 1] : 0), smem);
}

// (bf16: two 32-row stages per workgroup.  The one-64-row-stage variant of rounds 3-5 - bit-identical, never faster - and its switch
// cavp_set_wgrad_variant were removed in round 6.)
// The 256 x 256 tile of conv_wgrad_big.hip (bf16).  mode 0 (default): the jobs big enough for it (wgrad_big_eligible), 1: never
// (the 128 x 128 tile everywhere: A/B baseline), 2: every bf16 job (tests: tiny shapes through the big tile).  pipelined: its
// software-pipelined schedule (conv_wgrad_big.hip; 0 = read, barrier, multiply).  The choice depends on the job alone, never on the group it travels in, so a grouped
// and a single launch of one job with the same split count stay bit-identical.
static int g_wgrad_big_mode = 0;
static int g_wgrad_big_pipe = 2;
extern "C" int cavp_set_wgrad_big(int mode, int pipelined) {
  if (mode < 0 || mode > 2 || pipelined < 0 || pipelined > 2) return CAVP_ERR_BAD_ARG;
  g_wgrad_big_mode = mode;
  g_wgrad_big_pipe = pipelined;
  return CAVP_OK;
}

// dw += sum_z slabs[z] over the live taps only (dead-tap regions of the slabs are never written).
// 256 threads = QPB output quads x ZG split groups: group zg sums the splits zg, zg + ZG, ... and the groups are
// combined through LDS in a fixed order (deterministic).  Small dW (16 K elements from 392 splits) needs the split
// dimension spread over threads: one thread per quad walking all splits left 16 workgroups each chasing 392
// dependent-latency loads (50 us of an 83 us launch).
template <int ZG>
__device__ __forceinline__ void wgrad_reduce_body(const WgradParams& p, const int blk, const int nblk, float4* part) {
  constexpr int QPB = 256 / ZG;   // part: [ZG][QPB]
  const int cq = p.Cin >> 2;
  const long long total = (long long)p.Cout * p.ntaps * cq;
  const size_t slab = (size_t)p.Cout * p.ntaps_all * p.Cin;
  const int ql = threadIdx.x % QPB, zg = threadIdx.x / QPB;
  for (long long i0 = (long long)blk * QPB; i0 < total; i0 += (long long)nblk * QPB) {
    const long long i = i0 + ql;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t off = 0;
    int co = 0, tap = 0, c4 = 0;
    if (i < total) {   // (total = Cout * ntaps * Cin / 4 < 2^31: checked by the planner)
      const int r = fast_div((int)i, p.dv_cq[0], p.dv_cq[1]);
      c4 = (int)i - r * cq;
      co = fast_div(r, p.dv_nt[0], p.dv_nt[1]);
      const int ti = r - co * p.ntaps;
      tap = (int)((p.taps >> (4 * ti)) & 15ull);
      off = ((size_t)co * p.ntaps_all + tap) * p.Cin + (size_t)c4 * 4;
      // the splits are summed in a fixed order (z ascending) with the loads of 8 splits in flight
      const float* sp = p.slabs + off + (size_t)zg * slab;
      const size_t zstep = (size_t)ZG * slab;
      int z = zg;
      for (; z + 7 * ZG < p.ksplit; z += 8 * ZG) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(sp + (size_t)u * zstep);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        sp += 8 * zstep;
      }
      for (; z < p.ksplit; z += ZG) {
        const float4 v = *(const float4*)sp;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        sp += zstep;
      }
    }
    if constexpr (ZG > 1) {
      part[zg * QPB + ql] = s;
      __syncthreads();
      if (zg == 0 && i < total) {
#pragma unroll
        for (int g = 1; g < ZG; ++g) {
          const float4 v = part[g * QPB + ql];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
      __syncthreads();
    }
    if (zg == 0 && i < total) {
      if (p.oihw) {   // off = (co * taps_all + tap) * Cin + ci  ->  (co * Cin + ci) * taps_all + tap
        float* d = p.dw + ((size_t)co * p.Cin + (size_t)c4 * 4) * p.ntaps_all + tap;
        if (p.overwrite) {
          d[0] = s.x; d[p.ntaps_all] = s.y; d[2 * (size_t)p.ntaps_all] = s.z; d[3 * (size_t)p.ntaps_all] = s.w;
        } else {
          d[0] += s.x; d[p.ntaps_all] += s.y; d[2 * (size_t)p.ntaps_all] += s.z; d[3 * (size_t)p.ntaps_all] += s.w;
        }
      } else if (p.overwrite) {
        *(float4*)(p.dw + off) = s;
      } else {
        const float4 o = *(const float4*)(p.dw + off);
        *(float4*)(p.dw + off) = make_float4(o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w);
      }
    }
  }
  if (p.dbias) {   // bias partials: one thread per output channel, splits in order
    for (int c = blk * 256 + threadIdx.x; c < p.Cout; c += nblk * 256) {
      float s = 0.f;
      for (int z = 0; z < p.ksplit; ++z) s += p.bias_slabs[(size_t)z * p.Cout + c];
      p.dbias[c] += s;
    }
  }
}

template <int ZG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradParams p) {
  __shared__ float4 part[256];
  wgrad_reduce_body<ZG>(p, blockIdx.x, gridDim.x, part);
}

// the slab reduces of a grouped launch, one grid
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const WgradGroupArgs g) {
  __shared__ float4 part[256];
  const int bid = blockIdx.x;
  int j = 0;
  while (j + 1 < g.njobs && bid >= g.red_end[j]) ++j;
  const int b0 = j ? g.red_end[j - 1] : 0;
  const WgradParams& p = g.job[j];
  const int blk = bid - b0, nblk = g.red_end[j] - b0;
  switch (p.red_zg) {   // (wave-uniform)
    case 1: wgrad_reduce_body<1>(p, blk, nblk, part); break;
    case 2: wgrad_reduce_body<2>(p, blk, nblk, part); break;
    case 4: wgrad_reduce_body<4>(p, blk, nblk, part); break;
    case 8: wgrad_reduce_body<8>(p, blk, nblk, part); break;
    default: wgrad_reduce_body<16>(p, blk, nblk, part); break;
  }
}

namespace {
// time model of the 256 x 256 tile (conv_wgrad_big.hip): seconds per 128-row ring trip of one workgroup and per workgroup
// (ring fill + 256 KiB accumulator store)
static const double kBigUnitSec = cavp_knob_double("CAVP_WGRAD_BIG_UNIT_US", 2.0) * 1e-6;
static const double kBigFixSec = cavp_knob_double("CAVP_WGRAD_BIG_FIX_US", 4.0) * 1e-6;
struct WgradPlan { WgradParams p; int nblk; size_t ws_bytes; int status; int base, chunks; };

// Jobs that go to the 256 x 256 tile: bf16, both channel counts fill most of a 256-wide tile edge and the pixel range is long
// enough for >= 256 workgroups of a few ring trips each (the head / token / projector layers at 2B x 56 x 56 pixels; the 14 x 14
// and 28 x 28 layers of the backbone stay on the 128 x 128 tile: four co-resident workgroups per CU fill the chip with fewer
// pixel splits there).
bool wgrad_big_eligible(const cavp_conv_desc* d) {
  if (!d || d->dtype != CAVP_BF16 || g_wgrad_big_mode == 1) return false;
  if (g_wgrad_big_mode == 2) return true;
  const long long Ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const long long Wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  static const int min_rows = cavp_knob_int("CAVP_WGRAD_BIG_MIN_ROWS", 16384), min_ch = cavp_knob_int("CAVP_WGRAD_BIG_MIN_CH", 192);
  return (long long)d->N * Ho * Wo >= min_rows && d->Cin >= min_ch && d->Cout >= min_ch;
}

// force_ks > 0: the group planner's split count (cavp_conv_desc.splitk still wins).  big: plan for the 256 x 256 tile (256-channel
// tiles, pixel slices in units of 128 rows = one trip of its 4-stage ring, 256 resident workgroups)
WgradPlan make_wgrad_plan(const cavp_conv_desc* d, int force_ks = 0, bool big = false) {
  WgradPlan pl{};
  pl.status = CAVP_OK;
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
      d->dil <= 0 || d->pad < 0 || d->ldx < d->Cin || d->ldy < d->Cout) { pl.status = CAVP_ERR_BAD_ARG; return pl; }
  if (d->dtype != CAVP_F32 && d->dtype != CAVP_BF16) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  const int es = d->dtype == CAVP_F32 ? 4 : 2;
  const int VE = 16 / es;
  if (d->Cin % VE || d->Cout % VE || d->ldx % VE || d->ldy % VE || d->KH * d->KW > 9) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  WgradParams& p = pl.p;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Cout = d->Cout; p.ldy = d->ldy;
  p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.Ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  p.Wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) { pl.status = CAVP_ERR_BAD_ARG; return pl; }
  const long long M = (long long)d->N * p.Ho * p.Wo;
  const size_t xb = ((size_t)d->N * d->H * d->W - 1) * d->ldx * es + (size_t)d->Cin * es;
  const size_t yb = ((size_t)M - 1) * d->ldy * es + (size_t)d->Cout * es;
  if (M > 0x3fffffff || xb >= 0x7fffffffull || yb >= 0x7fffffffull) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  p.M = (int)M; p.x_bytes = (int)xb; p.dy_bytes = (int)yb;
  p.ntaps_all = d->KH * d->KW;
  p.ntaps = 0; p.taps = 0;
  for (int kh = 0; kh < d->KH; ++kh) {
    bool hl = false;
    for (int ho = 0; ho < p.Ho && !hl; ++ho) { const int hi = ho * d->stride - d->pad + kh * d->dil; hl = hi >= 0 && hi < d->H; }
    for (int kw = 0; kw < d->KW; ++kw) {
      bool wl = false;
      for (int wo = 0; wo < p.Wo && !wl; ++wo) { const int wi = wo * d->stride - d->pad + kw * d->dil; wl = wi >= 0 && wi < d->W; }
      if (hl && wl) { p.taps |= (unsigned long long)(kh * d->KW + kw) << (4 * p.ntaps); ++p.ntaps; }
    }
  }
  if (p.ntaps == 0) { pl.nblk = 0; return pl; }
  const int TCH = big ? 256 : 256 / es;
  const int unit = big ? 128 : 64;   // pixel rows per planning chunk
  p.tiles_co = (d->Cout + TCH - 1) / TCH;
  p.tiles_ci = (d->Cin + TCH - 1) / TCH;
  const int base = p.tiles_co * p.tiles_ci * p.ntaps;
  fast_div_prepare(p.tiles_co, &p.dv_co[0], &p.dv_co[1]);
  fast_div_prepare(p.tiles_ci, &p.dv_ci[0], &p.dv_ci[1]);
  fast_div_prepare(p.ntaps, &p.dv_nt[0], &p.dv_nt[1]);
  fast_div_prepare(p.KW, &p.dv_kw[0], &p.dv_kw[1]);
  fast_div_prepare(p.Cin >> 2, &p.dv_cq[0], &p.dv_cq[1]);
  fast_div_prepare(p.Ho * p.Wo, &p.dv_hw[0], &p.dv_hw[1]);
  fast_div_prepare(p.Wo, &p.dv_w[0], &p.dv_w[1]);
  const int chunks = (p.M + unit - 1) / unit;
  pl.base = base; pl.chunks = chunks;
  // Split count from a small time model fitted to tools/bench_wgrad.py on MI355X (profiles/r01_notes.md):
  //   t(ks) = rounds * steps * 1.08 us  +  ks * |dW| * 8 B / 5 TB/s (the slabs mostly live in the 256 MB MALL)  (+ reduce launch)
  // rounds = ceil(base * ks / 1024 resident workgroups: four 32 KiB workgroups per CU), steps = 32-row K tiles per
  // workgroup.  The first version aimed at "about 512 workgroups" with a ceil: 540 or 513 workgroups = a second, nearly
  // empty round (head conv 951 -> 600 us); 64-row stages (two workgroups per CU, 1.7 us per tile) -> 32-row stages:
  // head conv 580 -> 415 us (the K loop is bound by DMA latency per workgroup, more resident workgroups hide it).
  int ks = 1;
  if (d->splitk > 0) {
    ks = d->splitk;
  } else if (force_ks > 0) {
    ks = force_ks;
  } else {
    const double dw_bytes = (double)d->Cout * p.ntaps * d->Cin * 4.0;
    double best = 1e30;
    static const double slab_bw = cavp_knob_double("CAVP_WGRAD_SLAB_TBS", 5.0) * 1e12;   // A/B knob; whole-step sweep 1.2 / 1.8 / 2.5 / 5 / 8 / 12 / 30 -> 18.85 / 18.65 / 18.55 / 18.47 / 18.50 / 18.58 / 18.69 ms
    const int ks_max = chunks / 2 > 1 ? chunks / 2 : 1;
    for (int k = 1; k <= ks_max && k <= 512; ++k) {
      const int steps = (chunks + k - 1) / k;
      const int kk = (chunks + steps - 1) / steps;   // effective split count for this step count
      const long long rounds = big ? ((long long)base * kk + 255) / 256 : ((long long)base * kk + 1023) / 1024;
      double t = (big ? (double)rounds * (steps * kBigUnitSec + kBigFixSec) : (double)rounds * (2 * steps) * 1.08e-6) +
                 (kk > 1 ? kk * dw_bytes * 2.0 / slab_bw + 6e-6 : 0.0);
      if (t < best) { best = t; ks = kk; }
    }
  }
  if (ks > chunks) ks = chunks;
  if (ks < 1) ks = 1;
  const int cps = (chunks + ks - 1) / ks;
  ks = (chunks + cps - 1) / cps;
  p.ksplit = ks;
  p.rows_per_split = cps * unit;
  pl.nblk = base * ks;
  pl.ws_bytes = ks > 1 ? ((size_t)ks * d->Cout * p.ntaps_all * d->Cin + (size_t)ks * d->Cout) * sizeof(float) : 0;   // dW slabs + bias slabs
  return pl;
}
}  // namespace

extern "C" size_t cavp_conv2d_wgrad_workspace_bytes(const cavp_conv_desc* d) {
  if (wgrad_big_eligible(d)) {   // the 256 x 256 tile has one launch path: a group of one
    cavp_wgrad_job jb{};
    jb.desc = *d;
    return cavp_conv2d_wgrad_group_workspace_bytes(&jb, 1);
  }
  WgradPlan pl = make_wgrad_plan(d);
  return pl.status == CAVP_OK ? pl.ws_bytes : 0;
}

extern "C" int cavp_conv2d_wgrad_nhwc(const cavp_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !x || !dy || !dw) return CAVP_ERR_BAD_ARG;
  if (wgrad_big_eligible(d)) {
    cavp_wgrad_job jb{};
    jb.desc = *d; jb.x = x; jb.dy = dy; jb.dw = dw; jb.dbias = dbias;
    return cavp_conv2d_wgrad_group(&jb, 1, workspace, workspace_bytes, stream);
  }
  WgradPlan pl = make_wgrad_plan(d);
  if (pl.status != CAVP_OK) return pl.status;
  const size_t dw_bytes = (size_t)d->Cout * d->KH * d->KW * d->Cin * sizeof(float);
  if (pl.nblk == 0) {   // every tap is outside the image: the gradient is zero
    if (d->dw_overwrite && cavp_zero_f32_async(dw, dw_bytes, (hipStream_t)stream) != hipSuccess) return CAVP_ERR_LAUNCH;
    return CAVP_OK;
  }
  if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dw & 15)) return CAVP_ERR_ALIGN;
  if (pl.ws_bytes > 0 && (!workspace || workspace_bytes < pl.ws_bytes || ((uintptr_t)workspace & 15))) return CAVP_ERR_WORKSPACE;
  WgradParams& p = pl.p;
  p.x = x; p.dy = dy; p.dw = dw; p.slabs = (float*)workspace; p.dbias = dbias;
  p.oihw = d->dw_oihw != 0 && p.ntaps_all > 1;   // (1x1: the two layouts coincide)
  p.overwrite = d->dw_overwrite != 0;
  if (p.overwrite && p.ntaps < p.ntaps_all) {   // dead taps of a dilated kernel are never visited: clear, then accumulate
    if (cavp_zero_f32_async(dw, dw_bytes, (hipStream_t)stream) != hipSuccess) return CAVP_ERR_LAUNCH;
    p.overwrite = 0;
  }
  p.bias_slabs = p.slabs ? p.slabs + (size_t)p.ksplit * d->Cout * p.ntaps_all * d->Cin : nullptr;
  {
    static const int dbg = cavp_knob_int("CAVP_WGRAD_DBG", 0);
    p.dbg = dbg;
  }
  static const int bk = cavp_knob_int("CAVP_WGRAD_BK", 32);   // A/B knob (profiling)
  const int lds = 2 * 2 * bk * 256;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<float, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<bf16_t, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<float, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256);
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<bf16_t, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256);
    attr = true;
  }
#define WG_LAUNCH(T, B, BI) wgrad_kernel<T, B, BI><<<pl.nblk, 256, lds, s>>>(p)
  const bool bi = dbias != nullptr;
  if (d->dtype == CAVP_F32) {
    if (bk == 32) { if (bi) WG_LAUNCH(float, 32, true); else WG_LAUNCH(float, 32, false); }
    else { if (bi) WG_LAUNCH(float, 64, true); else WG_LAUNCH(float, 64, false); }
  } else {
    if (bk == 32) { if (bi) WG_LAUNCH(bf16_t, 32, true); else WG_LAUNCH(bf16_t, 32, false); }
    else { if (bi) WG_LAUNCH(bf16_t, 64, true); else WG_LAUNCH(bf16_t, 64, false); }
  }
#undef WG_LAUNCH
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (p.ksplit > 1) {
    const long long quads = (long long)p.Cout * p.ntaps * (p.Cin / 4);
    if (quads > 0x7fffffffll) return CAVP_ERR_UNSUPPORTED;
    int zgrp = 1;   // split groups per workgroup: spread the split dimension when there are few output quads
    while (zgrp < 16 && zgrp * 2 <= p.ksplit && quads * zgrp < 131072) zgrp *= 2;
    long long nb = (quads + (256 / zgrp) - 1) / (256 / zgrp);
    if (nb > 8192) nb = 8192;
    switch (zgrp) {
      case 1: wgrad_reduce_kernel<1><<<(int)nb, 256, 0, s>>>(p); break;
      case 2: wgrad_reduce_kernel<2><<<(int)nb, 256, 0, s>>>(p); break;
      case 4: wgrad_reduce_kernel<4><<<(int)nb, 256, 0, s>>>(p); break;
      case 8: wgrad_reduce_kernel<8><<<(int)nb, 256, 0, s>>>(p); break;
      default: wgrad_reduce_kernel<16><<<(int)nb, 256, 0, s>>>(p); break;
    }
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  return CAVP_OK;
}

// --------------------------------------------------------------------------------------------------------------------------
// Grouped launch (ABI 7)
// --------------------------------------------------------------------------------------------------------------------------
namespace {
struct GroupPlan {
  WgradPlan pl[CAVP_WGRAD_GROUP_MAX];
  bool big[CAVP_WGRAD_GROUP_MAX];          // job runs on the 256 x 256 tile (conv_wgrad_big.hip)
  size_t slab_off[CAVP_WGRAD_GROUP_MAX];   // byte offsets into the workspace
  size_t ws_bytes;
  int status;
};

// average relative cost of a job's 256 x 256 tiles: a tile with few live 32-channel blocks (304 = 256 + 48) skips the dead
// MFMAs and fetches zero-filled rows without memory traffic, but still walks its stages
double big_tile_weight(const WgradParams& p) {
  double w = 0.0;
  for (int tc = 0; tc < p.tiles_ci; ++tc)
    for (int to = 0; to < p.tiles_co; ++to) {
      const int lci = (p.Cin - tc * 256 > 256 ? 256 : p.Cin - tc * 256 + 31) / 32, lco = (p.Cout - to * 256 > 256 ? 256 : p.Cout - to * 256 + 31) / 32;
      w += 0.3 + 0.7 * (double)(lci * lco) / 64.0;
    }
  return w / (p.tiles_ci * p.tiles_co);
}

// One pixel-range length L (in planning chunks: 64 rows, 128 for the 256 x 256 tile) for all the jobs of one tile kind: job j
// splits its chunks_j into ks_j = ceil(chunks_j / L) slices, so every workgroup of the launch runs about L chunk steps.  L
// minimises the single-launch model of make_wgrad_plan applied to the sum:  rounds(L) * steps(L) * t_step  +  slab traffic
// (+ the reduce launch).
void plan_subset(GroupPlan& gp, const cavp_wgrad_job* jobs, int njobs, bool big) {
  int max_chunks = 1;
  bool any = false;
  for (int j = 0; j < njobs; ++j)
    if (gp.big[j] == big && gp.pl[j].nblk > 0) {
      any = true;
      if (gp.pl[j].chunks > max_chunks) max_chunks = gp.pl[j].chunks;
    }
  if (!any) return;
  static const double slab_bw = cavp_knob_double("CAVP_WGRAD_SLAB_TBS", 5.0) * 1e12;
  double best = 1e30;
  int bestL = max_chunks;
  for (int L = 1; L <= max_chunks; ++L) {
    long long wgs = 0;
    double wwgs = 0.0;
    int steps = 0;
    double slab = 0.0;
    bool split = false;
    for (int j = 0; j < njobs; ++j) {
      const WgradPlan& q = gp.pl[j];
      if (gp.big[j] != big || q.nblk == 0) continue;
      int ks = jobs[j].desc.splitk > 0 ? jobs[j].desc.splitk : (q.chunks + L - 1) / L;
      if (ks > 512) ks = 512;
      if (ks > q.chunks) ks = q.chunks;
      const int st = (q.chunks + ks - 1) / ks;
      ks = (q.chunks + st - 1) / st;
      wgs += (long long)q.base * ks;
      if (big) wwgs += (double)q.base * ks * big_tile_weight(q.p);
      if (st > steps) steps = st;
      if (ks > 1) {
        split = true;
        slab += (double)ks * q.p.Cout * q.p.ntaps * q.p.Cin * 4.0 * 2.0;
      }
    }
    double t;
    if (big) {   // 256 resident workgroups; the dispatcher balances tiles of different cost over the rounds
      const long long rounds = wgs <= 256 ? 1 : (long long)(wwgs / 256.0 + 0.999);
      t = (double)(rounds < 1 ? 1 : rounds) * (steps * kBigUnitSec + kBigFixSec);
    } else {
      const long long rounds = (wgs + 1023) / 1024;
      t = (double)rounds * (2 * steps) * 1.08e-6;
    }
    t += slab / slab_bw + (split ? 6e-6 : 0.0);
    if (t < best) { best = t; bestL = L; }
  }
  for (int j = 0; j < njobs; ++j) {
    WgradPlan& q = gp.pl[j];
    if (gp.big[j] != big || q.nblk == 0) continue;
    int ks = (q.chunks + bestL - 1) / bestL;
    if (ks > 512) ks = 512;
    q = make_wgrad_plan(&jobs[j].desc, ks, big);
  }
}

GroupPlan make_group_plan(const cavp_wgrad_job* jobs, int njobs) {
  GroupPlan gp{};
  gp.status = CAVP_OK;
  if (!jobs || njobs <= 0 || njobs > CAVP_WGRAD_GROUP_MAX) { gp.status = CAVP_ERR_BAD_ARG; return gp; }
  for (int j = 0; j < njobs; ++j) {
    if (jobs[j].desc.dtype != jobs[0].desc.dtype) { gp.status = CAVP_ERR_BAD_ARG; return gp; }
    gp.big[j] = wgrad_big_eligible(&jobs[j].desc);
    gp.pl[j] = make_wgrad_plan(&jobs[j].desc, 1, gp.big[j]);
    if (gp.pl[j].status != CAVP_OK) { gp.status = gp.pl[j].status; return gp; }
  }
  plan_subset(gp, jobs, njobs, false);
  plan_subset(gp, jobs, njobs, true);
  size_t off = 0;
  for (int j = 0; j < njobs; ++j) {
    if (gp.pl[j].nblk == 0) continue;
    gp.slab_off[j] = off;
    off += (gp.pl[j].ws_bytes + 255) & ~(size_t)255;
  }
  gp.ws_bytes = off;
  return gp;
}
}  // namespace

extern "C" size_t cavp_conv2d_wgrad_group_workspace_bytes(const cavp_wgrad_job* jobs, int32_t njobs) {
  const GroupPlan gp = make_group_plan(jobs, njobs);
  return gp.status == CAVP_OK ? gp.ws_bytes : 0;
}

extern "C" int cavp_conv2d_wgrad_group(const cavp_wgrad_job* jobs, int32_t njobs, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  GroupPlan gp = make_group_plan(jobs, njobs);
  if (gp.status != CAVP_OK) return gp.status;
  if (gp.ws_bytes > 0 && (!workspace || workspace_bytes < gp.ws_bytes || ((uintptr_t)workspace & 15))) return CAVP_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // three kernel-argument blocks: the jobs of the 128 x 128 tile, the jobs of the 256 x 256 tile, and ALL jobs that split their
  // pixel range for the one grouped slab reduce (only its red_end / red_zg / job fields are read there)
  WgradGroupArgs gs{}, gb{}, gr{};
  int ns = 0, nb = 0, nr = 0, sblocks = 0, bblocks = 0, rblocks = 0;
  bool sbias = false, bbias = false;
  for (int j = 0; j < njobs; ++j) {
    const cavp_wgrad_job& jb = jobs[j];
    const cavp_conv_desc* d = &jb.desc;
    if (!jb.x || !jb.dy || !jb.dw) return CAVP_ERR_BAD_ARG;
    for (int i = 0; i < j; ++i)   // two jobs adding into one gradient would race inside the launch
      if (jobs[i].dw == jb.dw || (jb.dbias && jobs[i].dbias == jb.dbias)) return CAVP_ERR_BAD_ARG;
    WgradPlan& pl = gp.pl[j];
    const size_t dw_bytes = (size_t)d->Cout * d->KH * d->KW * d->Cin * sizeof(float);
    if (pl.nblk == 0) {   // every tap is outside the image: the gradient is zero
      if (d->dw_overwrite && cavp_zero_f32_async(jb.dw, dw_bytes, s) != hipSuccess) return CAVP_ERR_LAUNCH;
      continue;
    }
    if (((uintptr_t)jb.x & 15) || ((uintptr_t)jb.dy & 15) || ((uintptr_t)jb.dw & 15)) return CAVP_ERR_ALIGN;
    WgradParams& p = pl.p;
    p.x = jb.x; p.dy = jb.dy; p.dw = jb.dw; p.dbias = jb.dbias;
    p.slabs = pl.ws_bytes ? (float*)((char*)workspace + gp.slab_off[j]) : nullptr;
    p.oihw = d->dw_oihw != 0 && p.ntaps_all > 1;
    p.overwrite = d->dw_overwrite != 0;
    if (p.overwrite && p.ntaps < p.ntaps_all) {   // dead taps of a dilated kernel are never visited: clear, then accumulate
      if (cavp_zero_f32_async(jb.dw, dw_bytes, s) != hipSuccess) return CAVP_ERR_LAUNCH;
      p.overwrite = 0;
    }
    p.bias_slabs = p.slabs ? p.slabs + (size_t)p.ksplit * d->Cout * p.ntaps_all * d->Cin : nullptr;
    {
      static const int dbg = cavp_knob_int("CAVP_WGRAD_DBG", 0);   // (profile builds; the 256 x 256 tile reads bits 1 and 2)
      p.dbg = dbg;
    }
    p.red_zg = 1;
    if (p.ksplit > 1) {
      const long long quads = (long long)p.Cout * p.ntaps * (p.Cin / 4);
      if (quads > 0x7fffffffll) return CAVP_ERR_UNSUPPORTED;
      int zgrp = 1;
      while (zgrp < 16 && zgrp * 2 <= p.ksplit && quads * zgrp < 131072) zgrp *= 2;
      long long nbk = (quads + (256 / zgrp) - 1) / (256 / zgrp);
      if (nbk > 2048) nbk = 2048;
      p.red_zg = zgrp;
      rblocks += (int)nbk;
      gr.job[nr] = p;
      gr.red_end[nr] = rblocks;
      ++nr;
    }
    if (gp.big[j]) {
      bblocks += pl.nblk;
      bbias = bbias || jb.dbias != nullptr;
      gb.job[nb] = p;
      gb.blk_end[nb] = bblocks;
      ++nb;
    } else {
      sblocks += pl.nblk;
      sbias = sbias || jb.dbias != nullptr;
      gs.job[ns] = p;
      gs.blk_end[ns] = sblocks;
      ++ns;
    }
  }
  gs.njobs = ns; gb.njobs = nb; gr.njobs = nr;
  if (nb > 0) {   // the long-running workgroups first
    if (cavp_launch_wgrad_big_group(gb, bblocks, bbias, g_wgrad_big_pipe, s) != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  if (ns > 0) {
    const int lds = 2 * 2 * 32 * 256;
    const bool f32 = jobs[0].desc.dtype == CAVP_F32;
    if (f32) {
      if (sbias) wgrad_group_kernel<float, true><<<sblocks, 256, lds, s>>>(gs);
      else wgrad_group_kernel<float, false><<<sblocks, 256, lds, s>>>(gs);
    } else {
      if (sbias) wgrad_group_kernel<bf16_t, true><<<sblocks, 256, lds, s>>>(gs);
      else wgrad_group_kernel<bf16_t, false><<<sblocks, 256, lds, s>>>(gs);
    }
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  if (nr > 0) {
    wgrad_reduce_group_kernel<<<rblocks, 256, 0, s>>>(gr);
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  return CAVP_OK;
}
