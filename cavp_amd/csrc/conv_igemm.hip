// Implicit-GEMM convolution / linear for gfx950 (MI355X), NHWC activations, OHWI weights, f32 accumulate on MFMA.
//
//   D[cout][pixel] = sum_{tap, ci} W[cout][tap][ci] * X[pixel @ tap][ci]
//
// * The weights are the MFMA "A" operand (rows = cout) and the gathered activations the "B" operand (cols = pixels),
//   so each lane's 4 accumulator registers are 4 CONSECUTIVE output channels of ONE pixel: the epilogue applies
//   the per-channel scale/shift (+ per-image bias, + residual, + activation) and stores 16 B (f32) / 8 B (bf16)
//   vectors straight into the NHWC output (optionally a channel slice of a wider tensor = free torch.cat).
// * Both operand tiles are staged in LDS as rows of 128 bytes = 8 x 16-byte slots with the slot index XOR-swizzled
//   by (row >> 1) & 7, which makes every ds_read_b128 lane group hit 16 distinct slots of the 256-byte bank row.
// * One K step consumes 4 slots: lane l reads slot j*4 + (l >> 4) of row (l & 15) for BOTH operands.  For bf16
//   that is exactly the 16x16x32 fragment; for f32 it is a K-permutation shared by A and B (sum order only), fed
//   to four exact-f32 v_mfma_f32_16x16x4_f32.
// * Global -> LDS directly (buffer_load ... lds, no VGPR staging) through a ring of NS LDS stages with ONE barrier
//   per K tile: tile k is multiplied while tiles k+1 .. k+NS-1 are in flight (counted s_waitcnt vmcnt, raw
//   s_barrier).  The 4-wave / 2-stage tiles keep two workgroups per CU; the 8-wave 256x128 / 128x256 tiles hold one
//   144 KiB 3-stage workgroup per CU (96 KiB of operands in flight per CU and 1.36x fewer operand bytes per flop).
//   Out-of-image taps, K tails (Cin not a multiple of the tile) and M / Cout tails are zero-filled by the buffer
//   descriptor's bounds check (the lane is pointed past num_records instead of branching).
// * Taps that cannot touch the image for any output pixel (dilation >= extent, e.g. ASPP d=18 on 14x14) are
//   removed on the host; split-K writes f32 slabs that a small epilogue kernel reduces deterministically.
#include <stdlib.h>

#include <stdio.h>

#include "common.h"
#include "igemm_params.h"

// Shared epilogue math for one output element.
__device__ __forceinline__ float epi_one(float v, int cc, int n_img, const IgemmParams& p) {
  if (p.nbias) v += p.nbias[(size_t)n_img * p.Cout + cc];
  if (p.scale) v *= p.scale[cc];
  if (p.shift) v += p.shift[cc];
  return v;
}

template <typename T>
__device__ __forceinline__ void epilogue_store4(const IgemmParams& p, float v0, float v1, float v2, float v3, int pix,
                                                int c) {
  const int n_img = p.nbias ? fast_div(pix, p.div_hw_m, p.div_hw_s) : 0;
  float v[4] = {v0, v1, v2, v3};
  T* yp = (T*)p.y + (size_t)pix * p.ldy + c;
  const T* rp = p.res ? (const T*)p.res + (size_t)pix * p.ldr + c : nullptr;
  if (p.vec_io) {
    if (c >= p.Cout) return;
    if (p.nbias) {
      const float4 nb = *(const float4*)(p.nbias + (size_t)n_img * p.Cout + c);
      v[0] += nb.x; v[1] += nb.y; v[2] += nb.z; v[3] += nb.w;
    }
    if (p.scale) {
      const float4 s = *(const float4*)(p.scale + c);
      v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
    }
    if (p.shift) {
      const float4 s = *(const float4*)(p.shift + c);
      v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
    }
    if constexpr (sizeof(T) == 4) {
      if (rp) {
        const float4 r = *(const float4*)rp;
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      float4 o;
      o.x = apply_act(v[0], p.act); o.y = apply_act(v[1], p.act);
      o.z = apply_act(v[2], p.act); o.w = apply_act(v[3], p.act);
      *(float4*)yp = o;
    } else {
      if (rp) {
        const uint2 r = *(const uint2*)rp;
        v[0] += bf2f((bf16_t)(r.x & 0xffff)); v[1] += bf2f((bf16_t)(r.x >> 16));
        v[2] += bf2f((bf16_t)(r.y & 0xffff)); v[3] += bf2f((bf16_t)(r.y >> 16));
      }
      uint2 o;
      o.x = pack2bf(apply_act(v[0], p.act), apply_act(v[1], p.act));
      o.y = pack2bf(apply_act(v[2], p.act), apply_act(v[3], p.act));
      *(uint2*)yp = o;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cc = c + e;
      if (cc < p.Cout) {
        float t = epi_one(v[e], cc, n_img, p);
        if (rp) t += Elem<T>::ld(rp + e);
        Elem<T>::st(yp + e, apply_act(t, p.act));
      }
    }
  }
}

#ifdef CAVP_PROFILE
// Timeline of physical workgroup 0 / wave 0 (profile builds, CAVP_IGEMM_DBG bit 256): s_memtime sums over the tiles it walks.
// [0] tiles, [1] K iterations, [2] kernel entry -> first tile, [3] tile set-up (tile ids, per-row descriptors, tap masks),
// [4] first DMA issue -> first stage landed for everybody, [5] rest of the K loop, [6] epilogue, [7] whole kernel.
__device__ unsigned long long g_igemm_tl[8];
extern "C" int cavp_prof_igemm_timeline(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_igemm_tl), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#define CAVP_TLI(stmt) do { if (tl_on) { stmt; } } while (0)
#else
#define CAVP_TLI(stmt) do { } while (0)
#endif

// BNB: the instantiation whose staged epilogue carries the BatchNorm-backward statistics (IgemmParams.bnb_*).  A template flag, not a
// run-time one: as a run-time branch the extra live ranges cost the plain launches 40 .. 230 spilled VGPRs per kernel.
template <typename T, int BC, int BP, int WC, int WP, bool UP, int NS, bool BNB = false>
__global__ __launch_bounds__(64 * WC * WP, (WC * WP == 4 ? (BC * BP <= 64 * 64 && sizeof(T) == 2 ? 4 : 2) : 1)) void igemm_kernel(const IgemmParams p) {
  constexpr int VE = Elem<T>::VE;
  constexpr int BK = 8 * VE;  // one 128-byte LDS row of K
  constexpr int TC = BC / WC, TP = BP / WP;
  constexpr int MC = TC / 16, MP = TP / 16;
  constexpr int NW = WC * WP, NT = 64 * NW;
  constexpr int NVW = BC * 8, NVX = BP * 8;
  constexpr int LW = (NVW + NT - 1) / NT, LX = (NVX + NT - 1) / NT;
  constexpr int TILE_BYTES = (BC + BP) * 128;
  constexpr int RSTEP = 8 * NW;  // LDS rows covered by one DMA instruction of the whole workgroup
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  static_assert(NS >= 2 && (NS - 2) * (LW + LX) < 64, "vmcnt is a 6-bit counter");
  static_assert(NW == 4 || NS == 3, "the 8-wave tiles are written for three stages");
  static_assert(TC % 16 == 0 && TP % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA block");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  cavp_prefetch_kernargs<(int)sizeof(IgemmParams)>();

  // (the wave index as a SCALAR: every LDS-DMA destination - M0 - then comes from SALU adds; as a vector value it cost one
  // VGPR + one v_readfirstlane per DMA instruction)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CAVP_PROFILE
  const bool tl_on = (p.dbg & 256) != 0 && blockIdx.x == 0 && wave == 0;   // (wave-uniform)
  unsigned long long tl_entry = 0, tl_t0 = 0, tl_t1 = 0, tl_t2 = 0, tl_t3 = 0, tl_s[4] = {0, 0, 0, 0}, tl_tiles = 0, tl_iters = 0, tl_first = 0;
  CAVP_TLI(tl_entry = __builtin_readcyclecounter());
#endif
  // Buffer descriptors, DMA geometry and the tile state that the K loop and the epilogue share.  The state of a tile (ids, per-thread
  // DMA descriptors, K-loop position) is set up by start_tile(), which also issues the tile's FIRST stage(s): it runs once in front of
  // the tile loop and then at the END of every tile for the next one - see the tail of the loop body.
  // Buffer descriptors (wave-uniform, from kernel arguments): the hardware bounds check returns 0 for any offset
  // >= num_records, so padding taps / K tails / M and Cout tails are "loaded" as zeros by pointing the lane at
  // kOOB instead of branching or selecting on the loaded data.
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int row0 = 8 * wave + (lane >> 3);
  const int kslot = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * VE;
  const int kbyte = kslot * (int)sizeof(T);
  unsigned w_off[LW];  // byte offset of (cout row, k = kslot) or kOOB
  unsigned x_off[LX];   // byte offset of (pixel @ tap displacement 0, channel kslot); wraps are harmless (masked)
  unsigned x_mask[LX];  // bit ti = live tap ti reads inside the image for this row
  int x_h0[LX], x_w0[LX], x_nb[LX];  // x_nb: UP only
  const int HoWo = p.Ho * p.Wo;
  const bool pointwise = !UP && p.ntaps == 1 && p.stride == 1 && p.stride_w == 1 && p.tap_dh[0] == p.pad && p.tap_dw[0] == p.pad &&
                         p.Ho == p.H && p.Wo == p.W;
  int z = 0, tc = 0, tp = 0, c_base = 0, p_base = 0, it_begin = 0, it_end = 0;   // logical workgroup: split-K slice, channel tile, pixel tile
  int pq = 0;   // UP, parity-ordered tiles (igemm_params.h): the tile's output parity class
  int par_rot = 0;   // UP, parity-ordered tiles: log2 of the quadruples a workgroup strides per trip (see start_tile)
  if constexpr (UP) {
    const int step = fast_div((int)(gridDim.x >> 5), p.div_tc_m, p.div_tc_s);   // (gridDim.x / 8 / tiles_c) / 4
    par_rot = step > 1 ? 31 - __builtin_clz((unsigned)step) : 0;
  }
  float bcf[4] = {1.f, 0.f, 0.f, 1.f};
  int g_ti = 0, g_cc = 0;

  // global -> LDS directly (buffer_load ... lds), no VGPR staging, no ds_write; out-of-range lanes land zeros.
  // `live` = false issues the same number of DMA instructions with every lane out of range (zero fill, no memory
  // traffic): the ring's tail keeps the per-iteration vmcnt arithmetic uniform.
  auto gdma = [&](int buf, bool live) {
    int ti = g_ti < p.ntaps ? g_ti : p.ntaps - 1;
    if constexpr (UP) {
      if (p.up_par) {   // the class's own tap list (4 bits per launch tap index)
        const int nt = p.par_nt[pq], gi = g_ti < nt ? g_ti : nt - 1;
        ti = (int)((p.par_taps[pq] >> (4 * gi)) & 15ull);
      }
    }
    const int c0 = g_cc * BK;
    const bool c_ok = live && (c0 + kslot) < p.Cin;
    const unsigned oobm = c_ok ? 0u : kOOB;  // OR-ed into the offset: any offset >= 2^31 is out of range (tensors < 2 GiB)
    const unsigned wk = (unsigned)(p.tap_woff[ti] + c0 * (int)sizeof(T));
    char* base = smem + buf * TILE_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      if (BC % RSTEP == 0 || 8 * wave + RSTEP * i < BC) {
        const unsigned off = (w_off[i] + wk) | oobm;  // (a kOOB row + wk stays >= 2^31)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + i * (NW * 1024)), 16,
                                                 (int)off, 0, 0, 0);
      }
    }
    if constexpr (UP) {
      const int dh = p.tap_dh[ti], dw = p.tap_dw[ti];
      const int xk = c0 * (int)sizeof(T) + kbyte;
#pragma unroll
      for (int i = 0; i < LX; ++i) {
        if (BP % RSTEP == 0 || 8 * wave + RSTEP * i < BP) {
          const int hv = x_h0[i] + dh, wv = x_w0[i] + dw;  // position in the (virtually zero-upsampled) input
          const int hi = hv >> p.up_shift, wi = wv >> p.up_shift;
          const bool ok = (((hv | wv) & p.up_mask) == 0) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
          const unsigned off = (unsigned)(x_nb[i] + hi * p.W + wi) * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)xk;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              xrsrc, (__attribute__((address_space(3))) void*)(base + BC * 128 + i * (NW * 1024)), 16, (int)((ok ? off : kOOB) | oobm), 0, 0, 0);
        }
      }
    } else {
      const unsigned xk = (unsigned)(p.tap_xoff[ti] + c0 * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < LX; ++i) {
        if (BP % RSTEP == 0 || 8 * wave + RSTEP * i < BP) {
          const unsigned tapm = ((~(x_mask[i] >> ti)) & 1u) << 31;  // tap outside the image for this pixel
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              xrsrc, (__attribute__((address_space(3))) void*)(base + BC * 128 + i * (NW * 1024)), 16,
              (int)((x_off[i] + xk) | tapm | oobm), 0, 0, 0);
        }
      }
    }
    if (++g_cc == p.cpt) { g_cc = 0; ++g_ti; }
  };

  auto start_tile = [&](int vb) {
    const int sid = xcd_remap(vb, p.nblk);
    const int tiles = p.tiles_c * p.tiles_p;
    // (wave-uniform, but integer division has no scalar instruction: each `/` below used to be a ~40-instruction VALU
    // sequence, the 64-bit ones more, paid by every wave for every tile)
    z = p.splitk == 1 ? 0 : sid / tiles;
    const int rem = sid - z * tiles;
    tp = fast_div(rem, p.div_tc_m, p.div_tc_s);
    tc = rem - tp * p.tiles_c;
    if constexpr (UP) {
      // parity-ordered tiles: consecutive tile ids cycle through the four classes.  In class-major order the XCD remap hands the one class
      // of a 1x1 stride-2 gradient that has any tap (a quarter of the tiles, all of the K loops) to two of the eight XCDs: 74 us where
      // the image-order launch took 69.
      // ... and the class of a quadruple is rotated by (quadruple index >> par_rot): a persistent workgroup strides gridDim.x / 8 / tiles_c pixel
      // tiles per trip - a multiple of four - and so met ONE class in all its tiles (a 1x1 gradient's workgroups were either all K loop or
      // none).  Any function of the quadruple index keeps the map a bijection.
      if (p.up_par) {
        const int quad = tp >> 2;
        tp = (((tp & 3) + (quad >> par_rot)) & 3) * (p.tiles_p >> 2) + quad;
      }
    }
    c_base = tc * BC;
    p_base = tp * BP;
    // BatchNorm-backward instantiations: the tile's per-channel coefficients (mask scale / shift, batch mean, rstd) are requested HERE,
    // one channel per thread, and parked in LDS behind the K loop - the epilogue then reads them with LDS latency instead of waiting
    // out two rounds of dependent global loads per tile
    bcf[0] = 1.f; bcf[1] = 0.f; bcf[2] = 0.f; bcf[3] = 1.f;
    if constexpr (BNB) {
      const int c = c_base + tid;
      if (tid < BC && c < p.Cout) {
        if (p.bnb_scale) { bcf[0] = p.bnb_scale[c]; bcf[1] = p.bnb_shift[c]; }
        bcf[2] = p.bnb_mean[c]; bcf[3] = p.bnb_rstd[c];
      }
    }
    it_begin = p.splitk == 1 ? 0 : p.iters * z / p.splitk;   // iters * splitk < 2^31
    it_end = p.splitk == 1 ? p.iters : p.iters * (z + 1) / p.splitk;
    if constexpr (UP) {
      if (p.up_par) {   // a tile lies inside one parity class and walks that class's live taps only (possibly none: the K loop is skipped)
        pq = fast_div(p_base, p.div_mq_m, p.div_mq_s);
        it_begin = 0;
        it_end = p.par_nt[pq] * p.cpt;
      }
    }

    // ---- per-thread load descriptors (fixed for the whole K loop) ----
    // LDS-DMA geometry: a wave-instruction writes 1 KiB = rows 8g..8g+7 LINEARLY (lane l -> byte 16 l), g = wave + NW i,
    // so the bank swizzle moves to the SOURCE: lane l fetches logical slot (l & 7) ^ ((row >> 1) & 7) of its row.
    // Everything that does not change over the K loop is folded into one byte offset + one tap-validity bitmask per
    // row, so a K iteration costs ~4 VALU per load (the first version redid the bounds tests and two integer
    // divisions every iteration: 80 VALU + 134 SALU per 32 MFMAs, i.e. the address math, not the MFMAs, set the pace).
  #pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int row = row0 + RSTEP * i;
      const int c = c_base + row;
      const bool ok = (row < BC) && (c < p.Cout);
      w_off[i] = ok ? (unsigned)(((size_t)c * p.K + kslot) * sizeof(T)) : kOOB;
    }
    // pointwise layers (1x1, stride 1, no padding: two thirds of the conv / linear launches of a CAVP step): output pixel = input
    // pixel, the one tap is always inside the image - no (n, ho, wo) split (two multiply-shift divisions per row) and no tap loop.
    // Short-K layers pay this set-up once per tile next to a handful of K steps.
    // (also a dilated 3x3 whose only live tap is the centre one: ASPP d = 18 on 14 x 14)
    if (pointwise) {
  #pragma unroll
      for (int i = 0; i < LX; ++i) {
        const int row = row0 + RSTEP * i;
        const int pix = p_base + row;
        const bool ok = (row < BP) && (pix < p.M);
        x_h0[i] = 0; x_w0[i] = 0; x_nb[i] = 0;
        x_mask[i] = ok ? 1u : 0u;
        x_off[i] = (unsigned)pix * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)kbyte - (unsigned)p.tap_xoff[0];   // (gdma adds tap_xoff back)
      }
    } else {
  #pragma unroll
    for (int i = 0; i < LX; ++i) {
      const int row = row0 + RSTEP * i;
      const int pix = p_base + row;
      const bool ok = (row < BP) && (pix < p.M);
      const int pp = ok ? pix : 0;
      int n, ho, wo;
      bool rok = ok;
      if (UP && p.up_par) {
        rok = ok && par_out_pixel(p, pp, n, ho, wo) >= 0;   // (padding slots of a class are dead rows)
      } else {
        n = fast_div(pp, p.div_hw_m, p.div_hw_s);
        const int r = pp - n * HoWo;
        ho = fast_div(r, p.div_w_m, p.div_w_s);
        wo = r - ho * p.Wo;
      }
      const int h0 = ho * p.stride - p.pad, w0 = wo * p.stride_w - p.pad;
      x_h0[i] = rok ? h0 : -0x10000000;  // a dead row fails every bounds test below
      x_w0[i] = w0;
      x_mask[i] = 0;
      if constexpr (UP) {
        x_nb[i] = n * p.H * p.W;
        x_off[i] = 0;
      } else {
        x_off[i] = (unsigned)((n * p.H + h0) * p.W + w0) * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)kbyte;
      }
    }
    if constexpr (!UP) {
      for (int t = 0; t < p.ntaps; ++t) {  // tap outermost: its two scalar table loads are shared by the LX rows
        const int dh = p.tap_dh[t], dw = p.tap_dw[t];
  #pragma unroll
        for (int i = 0; i < LX; ++i)
          x_mask[i] |= ((unsigned)(x_h0[i] + dh) < (unsigned)p.H && (unsigned)(x_w0[i] + dw) < (unsigned)p.W) ? (1u << t) : 0u;
      }
    }
    }

    // K-loop position (tap index, channel tile) kept incrementally: no division in the loop
    g_ti = it_begin == 0 ? 0 : it_begin / p.cpt;
    g_cc = it_begin - g_ti * p.cpt;

    // the tile's first stage(s): requested here, i.e. for every tile but a workgroup's first one right behind the previous tile's
    // epilogue - the round trip of these loads overlaps the acknowledgement of that tile's global stores (hipcc parks an
    // s_waitcnt vmcnt(0) at the tile loop's latch whatever the source does, profiles/r06_notes.md 3: before this, a workgroup sat out
    // its stores' acknowledgement and only THEN set up and requested the next tile)
    if (it_begin < it_end) {
      if constexpr (NS == 2) {
        gdma(0, true);
      } else if constexpr (NW == 4) {
        const int n = it_end - it_begin;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) gdma(s, s < n);
      } else {
        const int n = it_end - it_begin;
        gdma(0, true);
        gdma(1, 1 < n);
        gdma(2, 2 < n);
      }
    }
  };
  start_tile((int)blockIdx.x);

  // persistent workgroups: gridDim.x (a multiple of 8, so a workgroup keeps its XCD) physical workgroups walk the
  // p.nblk logical ones: launch, kernel-argument load and teardown are paid once per physical workgroup
  for (int vb = blockIdx.x; vb < p.nblk; vb += gridDim.x) {
  do {   // (`continue` below = this tile is finished: on to the loop tail)
#ifdef CAVP_PROFILE
  CAVP_TLI({
    const unsigned long long now = __builtin_readcyclecounter();
    if (tl_tiles) { tl_s[0] += tl_t1 - tl_t0; tl_s[1] += tl_t2 - tl_t1; tl_s[2] += tl_t3 - tl_t2; tl_s[3] += now - tl_t3; } else { tl_first = now - tl_entry; }
    tl_t0 = now; tl_t2 = 0;
    ++tl_tiles;
  });
#endif
  const int wc0 = (wave % WC) * TC, wp0 = (wave / WC) * TP;
  const int lrow = lane & 15, lgrp = lane >> 4;

  f32x4_t acc[MC][MP];
#pragma unroll
  for (int a = 0; a < MC; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment loads of K sub-step j (32 bf16 / 16 f32 of K) of one stage, and the MFMA block that consumes them
  auto read_frag = [&](u32x4_t (&af)[MC], u32x4_t (&bfv)[MP], int buf, int j) {
    const char* wb = smem + buf * TILE_BYTES;
    const char* xb = wb + BC * 128;
    const int s = j * 4 + lgrp;
#pragma unroll
    for (int a = 0; a < MC; ++a) {
      const int r = wc0 + a * 16 + lrow;
      af[a] = *(const u32x4_t*)(wb + r * 128 + ((s ^ ((r >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int b = 0; b < MP; ++b) {
      const int r = wp0 + b * 16 + lrow;
      bfv[b] = *(const u32x4_t*)(xb + r * 128 + ((s ^ ((r >> 1) & 7)) << 4));
    }
  };
  auto mma_block = [&](const u32x4_t (&af)[MC], const u32x4_t (&bfv)[MP]) {
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b) Mma<T>::run(acc[a][b], af[a], bfv[b]);
  };

  CAVP_TLI({ tl_t1 = __builtin_readcyclecounter(); tl_iters += (unsigned long long)(it_end - it_begin); });
  if (it_begin < it_end) {
    if constexpr (NS == 2) {
      // two stages, two workgroups per CU.  Iteration `it`: wait for MY loads of tile `it`, barrier (=> everybody's
      // tile `it` landed and everybody finished multiplying tile it-1), issue tile it+1 into the stage tile it-1
      // vacated, multiply tile `it`.  (Tile 0 of the ring was requested by start_tile.)
      int buf = 0;
      for (int it = it_begin; it < it_end; ++it) {
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        __builtin_amdgcn_s_barrier();
        CAVP_TLI(if (tl_t2 == 0) tl_t2 = __builtin_readcyclecounter());
        // (no dummy DMA behind a tile's LAST K tile: with vmcnt(0) per iteration the count need not stay uniform, and the barrier that
        // ends the K loop would wait for its zero fill - ~0.5 k cycles in front of every epilogue of the one-K-tile 1x1 layers)
        if (it + 1 < it_end && !CAVP_DBG(p, 8)) gdma(buf ^ 1, !CAVP_DBG(p, 1));
        if (!CAVP_DBG(p, 2)) {
          u32x4_t af[MC], bfv[MP];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            read_frag(af, bfv, buf, j);
            mma_block(af, bfv);
          }
        }
        buf ^= 1;
      }
    } else if constexpr (NW == 4) {
      // NS >= 3 stages, 4 waves: a deep ring for the launches that cannot fill the chip (the 14x14 / 28x28 layers: <= 2
      // workgroups per CU, nothing co-resident to hide a K tile's load latency behind).  NS-1 tiles are in flight; one
      // barrier per K tile: it proves that everybody's tile `it` landed AND that everybody finished tile it-1, whose stage
      // is refilled right after it.
      constexpr int LD = LW + LX;
      const int n = it_end - it_begin;   // (stages 0 .. NS-2 were requested by start_tile)
      int buf = 0, fill = NS - 1;
      for (int it = 0; it < n; ++it) {
        __builtin_amdgcn_s_waitcnt(vmcnt_imm((NS - 2) * LD));
        __builtin_amdgcn_s_barrier();
        CAVP_TLI(if (tl_t2 == 0) tl_t2 = __builtin_readcyclecounter());
        gdma(fill, it + NS - 1 < n);
        u32x4_t af[MC], bfv[MP];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          read_frag(af, bfv, buf, j);
          mma_block(af, bfv);
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
      }
    } else {
      // three stages, one 8-wave workgroup per CU, software-pipelined through two fragment register sets so that no
      // ds_read latency is exposed: the barrier sits in the MIDDLE of a K tile.
      //   top:   issue the j=1 fragment reads of tile `it`; multiply its j=0 fragments (loaded during tile it-1)
      //   mid:   wait: my DMA of tile it+1 landed (tile it+2 may still fly) and my j=1 reads returned; barrier
      //          => tile it+1 is complete for everybody and stage it%3 is not read any more
      //          issue the DMA of tile it+3 into stage it%3, issue the j=0 fragment reads of tile it+1
      //   tail:  multiply the j=1 fragments of tile `it`
      static_assert(NS == 3, "ring arithmetic below is written for 3 stages");
      constexpr int LD = LW + LX;
      const int n = it_end - it_begin;   // (stages 0 .. 2 were requested by start_tile)
      u32x4_t a0[MC] = {}, b0[MP] = {}, a1[MC] = {}, b1[MP] = {};
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * LD));
      __builtin_amdgcn_s_barrier();
      CAVP_TLI(tl_t2 = __builtin_readcyclecounter());
      read_frag(a0, b0, 0, 0);
      int buf = 0;
      for (int it = 0; it < n; ++it) {
        const int nb = buf == 2 ? 0 : buf + 1;
        if (!CAVP_DBG(p, 4)) read_frag(a1, b1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!CAVP_DBG(p, 2)) mma_block(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt((LD & 15) | ((LD >> 4) << 14) | (7 << 4) | (0 << 8));  // vmcnt(LD) & lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (!CAVP_DBG(p, 8)) gdma(buf, it + 3 < n && !CAVP_DBG(p, 1));
        if (!CAVP_DBG(p, 4)) read_frag(a0, b0, nb, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!CAVP_DBG(p, 2)) mma_block(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nb;
      }
    }
    __syncthreads();  // drains the (empty) tail DMAs and fences the K-loop LDS reads before the epilogue reuses LDS
  }
  CAVP_TLI({ tl_t3 = __builtin_readcyclecounter(); if (tl_t2 == 0) tl_t2 = tl_t3; });

  // ---- epilogue ----
  if (CAVP_DBG(p, 16)) continue;
  if (p.coalesced) {
    // Stage the f32 accumulators in LDS as [pixel][cout] (16-byte slots XOR-swizzled by pixel & 7), then let each
    // thread finish VE consecutive channels of one pixel: scale/shift/bias/residual are 16-byte vector loads and the
    // output leaves as full 16-byte stores, BC*sizeof(T) contiguous bytes per pixel (vs 8-byte pieces scattered
    // over 16 pixels straight from the MFMA layout).  All K-loop LDS reads are behind the loop's last barrier.
    float* st = (float*)smem;
    constexpr int SLOTS = BC / 4;  // 16-byte f32 slots per pixel row
    constexpr int SWZ = SLOTS >= 8 ? 7 : SLOTS - 1;
    constexpr int CH = BC / VE;    // output chunks (16 B of T) per pixel row
    constexpr int NCH = BP * CH;
    constexpr int ITER = NCH / NT;  // chunks per thread
    constexpr int RSTR = NT / CH;   // pixel rows between a thread's consecutive chunks
    static_assert(NCH % NT == 0 && NT % CH == 0, "a thread keeps one channel group for all its chunks");
    // the thread's channel group is the same for every chunk it finishes: fetch its scale / shift ONCE, before the
    // staging barrier hides the latency (the first version re-loaded 2 x VE scalars per chunk inside a serial loop
    // whose every iteration waited on them: 11 us per 128x128 tile, 35 us of a 230 us launch)
    const int ecc = (tid % CH) * VE, ec = c_base + ecc, eprow0 = tid / CH;
    const bool ec_ok = ec < p.Cout;
    float sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (!BNB && ec_ok) {
      if (p.scale) {
#pragma unroll
        for (int q = 0; q < VE / 4; ++q) {
          const float4 t = *(const float4*)(p.scale + ec + 4 * q);
          sc[4 * q] = t.x; sc[4 * q + 1] = t.y; sc[4 * q + 2] = t.z; sc[4 * q + 3] = t.w;
        }
      }
      if (p.shift) {
#pragma unroll
        for (int q = 0; q < VE / 4; ++q) {
          const float4 t = *(const float4*)(p.shift + ec + 4 * q);
          sh[4 * q] = t.x; sh[4 * q + 1] = t.y; sh[4 * q + 2] = t.z; sh[4 * q + 3] = t.w;
        }
      }
    }
    // The residual of the thread's first G chunks is requested HERE, before the statistics and the staging round trip: issued
    // behind the staging barrier (rounds 1-4) every tile waited out one full memory latency (~2000 cycles of the ~14000-cycle
    // epilogue of a 128 x 128 bottleneck-output tile, profiles/r05_igemm_tile_timeline.txt) with nothing else to do.
    // chunks finished per batch: all residual loads of a batch fly together (the BatchNorm-backward instantiations also fetch z
    // and the activation output per chunk: half-size batches keep them in registers)
    constexpr int G = BNB ? (ITER < 4 ? ITER : 4) : (ITER < 8 ? ITER : 8);
    static_assert(ITER % G == 0, "chunk batches");
    const T* rp = p.res ? (const T*)p.res + (size_t)((p.res_rows ? p_base % p.res_rows : p_base) + eprow0) * p.ldr + ec : nullptr;
    const size_t rstep = (size_t)RSTR * p.ldr;
    const int rows_left = p.M - (p_base + eprow0);   // chunk k is in range iff k * RSTR < rows_left
    // UP instantiations (stride-2 data gradients: four launches of a step) address every chunk row through its OUTPUT pixel index, which the
    // parity-ordered tiles (igemm_params.h) do not get by adding a constant: -1 = no such pixel (tile tail / padding slot of a class)
    auto opix_of = [&](int k) -> int {
      const int r = p_base + eprow0 + k * RSTR;
      if (r >= p.M) return -1;
      if (!p.up_par) return r;
      int n_, h_, w_;
      return par_out_pixel(p, r, n_, h_, w_);
    };
    // (not for the tiles with 64 accumulator registers per lane: hoisting a batch spilled 34 VGPRs in the 128 x 128 kernel, half a
    // batch still 7)
    constexpr int GH = (MC * MP * 4 >= 64) ? 0 : G;
    u32x4_t rr0[GH > 0 ? GH : 1];
    if (GH > 0 && rp) {
#pragma unroll
      for (int g = 0; g < GH; ++g) {
        rr0[g] = (u32x4_t){0u, 0u, 0u, 0u};
        if constexpr (UP) {
          const int op_ = opix_of(g);
          if (ec_ok && op_ >= 0) rr0[g] = *(const u32x4_t*)((const T*)p.res + (size_t)op_ * p.ldr + ec);
        } else {
          if (ec_ok && g * RSTR < rows_left) rr0[g] = *(const u32x4_t*)(rp + (size_t)g * rstep);
        }
      }
    }
    // BatchNorm-backward instantiations: z (and the activation output) of the first batch travel with the residual
    u32x4_t zz0[(BNB && GH > 0) ? GH : 1], oo0[(BNB && GH > 0) ? GH : 1];
    if constexpr (BNB && GH > 0) {
      const T* zp0 = (const T*)p.bnb_z + (size_t)(p_base + eprow0) * p.ld_bnb_z + ec;
      const T* op0 = p.bnb_out ? (const T*)p.bnb_out + (size_t)(p_base + eprow0) * p.ld_bnb_out + ec : nullptr;
#pragma unroll
      for (int g = 0; g < GH; ++g) {
        zz0[g] = (u32x4_t){0u, 0u, 0u, 0u};
        oo0[g] = (u32x4_t){0u, 0u, 0u, 0u};
        if constexpr (UP) {
          const int op_ = opix_of(g);
          if (ec_ok && op_ >= 0) {
            zz0[g] = *(const u32x4_t*)((const T*)p.bnb_z + (size_t)op_ * p.ld_bnb_z + ec);
            if (op0) oo0[g] = *(const u32x4_t*)((const T*)p.bnb_out + (size_t)op_ * p.ld_bnb_out + ec);
          }
        } else if (ec_ok && g * RSTR < rows_left) {
          zz0[g] = *(const u32x4_t*)(zp0 + (size_t)g * RSTR * p.ld_bnb_z);
          if (op0) oo0[g] = *(const u32x4_t*)(op0 + (size_t)g * RSTR * p.ld_bnb_out);
        }
      }
    }
    float2* wstat = (float2*)(smem + NS * TILE_BYTES);   // [WP][BC] per-wave (mean, M2) partials (launch_cfg adds the bytes)
    if (!BNB && p.tile_stats) {
      // BatchNorm batch statistics for free, straight from the f32 accumulators IN REGISTERS: a lane holds 4 channels x MP
      // pixels per 16-row block; sums of (x - x0) and (x - x0)^2 about the wave's first row x0 (a sample of the same
      // distribution, so the one-pass form does not cancel), reduced over the 16 row lanes with DPP adds.  The WP waves
      // of a channel are combined after the staging barrier (Chan), the tiles in cavp_bn_finalize_tiles: no extra pass
      // over the activation and no atomics.  (The first version walked the staged tile in LDS with one thread per
      // channel: 2 x BP dependent LDS reads while the other threads idled, +20 % on the 3x3 head convs.)
      const int nvw = p.M - (p_base + wp0);   // valid rows of this wave's slab (<= 0: none)
#pragma unroll
      for (int a = 0; a < MC; ++a) {
        float s1[4], s2[4], x0[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x0[i] = row16_first(acc[a][0][i]);
          s1[i] = 0.f; s2[i] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < MP; ++b) {
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
          // (one reciprocal per wave and tile instead of an IEEE division per channel: this block is VALU time in front of every
          // store of a store-bound layer, profiles/r06_skinny_epilogue_anatomy.txt)
          const float inv_n = __frcp_rn((float)(nvw < TP ? (nvw > 0 ? nvw : 1) : TP));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float m = s1[i] * inv_n;
            wstat[(wave / WC) * BC + wc0 + a * 16 + lgrp * 4 + i] = make_float2(x0[i] + m, fmaxf(s2[i] - s1[i] * m, 0.f));
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < MC; ++a)
#pragma unroll
      for (int b = 0; b < MP; ++b) {
        const int prow = wp0 + b * 16 + lrow;
        const int slot = (wc0 + a * 16) / 4 + lgrp;
        if (!CAVP_DBG(p, 64)) *(f32x4_t*)(st + (size_t)prow * BC + ((slot ^ (prow & SWZ)) << 2)) = acc[a][b];
      }
    float* bcoef = (float*)(smem + NS * TILE_BYTES + WP * BC * 8);   // BNB: [4][BC] behind the statistics partials (launch_cfg adds the bytes)
    if constexpr (BNB) {
      if (tid < BC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bcoef[q * BC + tid] = bcf[q];
      }
    }
    __syncthreads();
    if (!BNB && p.tile_stats && tid < BC && c_base + tid < p.Cout) {  // (BC <= NT for every tile)
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < WP; ++w) {
        int nb = p.M - (p_base + w * TP);
        nb = nb < TP ? nb : TP;
        if (nb > 0) {
          const float2 q = wstat[w * BC + tid];
          const float fb = (float)nb, nt = n + fb, dlt = q.x - mean, fr = fb * __frcp_rn(nt);   // (one reciprocal instead of two divisions)
          mean += dlt * fr;
          m2 += q.y + dlt * dlt * (n * fr);
          n = nt;
        }
      }
      float* o = p.tile_stats + ((size_t)tp * p.Cout + c_base + tid) * 2;
      o[0] = mean;
      o[1] = m2;
    }
    static_assert(RSTR % 8 == 0, "the LDS swizzle key (row & 7) is the same for all chunks of a thread");
    // per-thread invariants: everything that does not depend on the chunk index is computed once, the chunk loop only
    // adds constants (this loop was VALU-bound: 64-bit multiplies and the swizzle per chunk, activation dispatch and an
    // integer bf16 rounding per element)
    const int key = eprow0 & SWZ;
    const float* lds0 = st + (size_t)eprow0 * BC + (((ecc / 4) ^ key) << 2);                       // first f32x4 of chunk 0
    const float* lds1 = VE == 8 ? st + (size_t)eprow0 * BC + (((ecc / 4 + 1) ^ key) << 2) : lds0;  // second (bf16 only)
    T* yp = (T*)p.y + (size_t)(p_base + eprow0) * p.ldy + ec;
    // (res_rows: a multiple of every tile height, so a tile never straddles the wrap)
    T* xp = (!BNB && p.aux_mode) ? (T*)p.aux + (size_t)(p_base + eprow0) * p.ld_aux + ec : nullptr;
    const size_t xstep = (size_t)RSTR * p.ld_aux;
    const size_t ystep = (size_t)RSTR * p.ldy;
    const bool has_ss = !BNB && (p.scale != nullptr || p.shift != nullptr);   // (BNB launches: plain data gradients)
    // BatchNorm-backward statistics (IgemmParams.bnb_*): the thread's channel group is fixed, so its mask / zhat coefficients and
    // its two running sums live in registers over all its chunks (loaded here, behind the staging barrier: the accumulators are
    // dead, their registers free)
    constexpr bool bnb = BNB;
    const T* zp = bnb ? (const T*)p.bnb_z + (size_t)(p_base + eprow0) * p.ld_bnb_z + ec : nullptr;
    const T* op = (bnb && p.bnb_out) ? (const T*)p.bnb_out + (size_t)(p_base + eprow0) * p.ld_bnb_out + ec : nullptr;
    const size_t zstep = (size_t)RSTR * p.ld_bnb_z, ostep = (size_t)RSTR * p.ld_bnb_out;
    // (sum g * zhat = rstd * (sum g * z - mean * sum g): mean / rstd enter once per thread behind the chunk loop - a thread sums at
    // most 8 rows, so this costs no accuracy against (z - mean) per element - and are not live inside it)
    // (f32, the parity path: 4 channels per thread, so the batch mean stays in registers and g * (z - mean) is formed per element)
    float bfs[VE], bfh[VE], bs0[VE], bs1[VE], bmu[sizeof(T) == 4 ? VE : 1];
#pragma unroll
    for (int e = 0; e < VE; ++e) { bfs[e] = 1.f; bfh[e] = 0.f; bs0[e] = 0.f; bs1[e] = 0.f; }
    if constexpr (bnb) {
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < VE; ++e) bmu[e] = bcoef[2 * BC + ecc + e];
      }
      if (!op && p.bnb_act != CAVP_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < VE; ++e) { bfs[e] = bcoef[ecc + e]; bfh[e] = bcoef[BC + ecc + e]; }
      }
    }
    for (int i0 = 0; i0 < ITER; i0 += G) {
      // UP: the output pixels of this batch's chunks, mapped ONCE (three multiply-shift divisions each) and used by the residual / z /
      // output loads and the store below
      int opb[UP ? G : 1];
      if constexpr (UP) {
#pragma unroll
        for (int g = 0; g < G; ++g) opb[g] = opix_of(i0 + g);
      }
      u32x4_t rr[G];
      if (rp) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (GH > 0 && i0 == 0 && g < GH) { rr[g] = rr0[g < GH ? g : 0]; continue; }
          rr[g] = (u32x4_t){0u, 0u, 0u, 0u};
          if constexpr (UP) {
            const int op_ = opb[g];
            if (ec_ok && op_ >= 0) rr[g] = *(const u32x4_t*)((const T*)p.res + (size_t)op_ * p.ldr + ec);
          } else {
            if (ec_ok && (i0 + g) * RSTR < rows_left) rr[g] = *(const u32x4_t*)(rp + (size_t)(i0 + g) * rstep);
          }
        }
      }
      u32x4_t zz[G], oo[G];
      if constexpr (bnb) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (GH > 0 && i0 == 0 && g < GH) { zz[g] = zz0[g < GH ? g : 0]; oo[g] = oo0[g < GH ? g : 0]; continue; }
          zz[g] = (u32x4_t){0u, 0u, 0u, 0u};
          oo[g] = (u32x4_t){0u, 0u, 0u, 0u};
          if constexpr (UP) {
            const int op_ = opb[g];
            if (ec_ok && op_ >= 0) {
              zz[g] = *(const u32x4_t*)((const T*)p.bnb_z + (size_t)op_ * p.ld_bnb_z + ec);
              if (op) oo[g] = *(const u32x4_t*)((const T*)p.bnb_out + (size_t)op_ * p.ld_bnb_out + ec);
            }
          } else if (ec_ok && (i0 + g) * RSTR < rows_left) {
            zz[g] = *(const u32x4_t*)(zp + (size_t)(i0 + g) * zstep);
            if (op) oo[g] = *(const u32x4_t*)(op + (size_t)(i0 + g) * ostep);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int k = i0 + g;
        // (no `continue` for out-of-range chunks: every register a predicated load above may have filled is CONSUMED on every path -
        // a loaded value that stays unused on some static path is still "pending" at the tile loop's latch for hipcc's wait-count
        // pass, which then parks an s_waitcnt vmcnt(0) there: every tile waited for the acknowledgement of its own global stores
        // before the next tile's operands were requested.  Dead chunks compute on zeros; only their stores and sums are masked.)
        const int opk = UP ? opb[UP ? g : 0] : 0;   // (UP: this chunk's output pixel)
        const bool live = ec_ok && (UP ? opk >= 0 : k * RSTR < rows_left);
        float v[VE];
        {
          const f32x4_t t = *(const f32x4_t*)(lds0 + (size_t)k * RSTR * BC);
          v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        }
        if constexpr (VE == 8) {
          const f32x4_t t = *(const f32x4_t*)(lds1 + (size_t)k * RSTR * BC);
          v[4] = t[0]; v[5] = t[1]; v[6] = t[2]; v[7] = t[3];
        }
        if (!BNB && p.nbias && live) {
          const float* nb = p.nbias + (size_t)fast_div(p_base + eprow0 + k * RSTR, p.div_hw_m, p.div_hw_s) * p.Cout + ec;
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] += nb[e];
        }
        if (has_ss) {
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] = __fadd_rn(__fmul_rn(v[e], sc[e]), sh[e]);  // two roundings, as every other epilogue path
        }
        if (!BNB && p.aux_mode == 2 && live) {   // d(pre) = d(hidden) * gelu'(pre): the multiplier tensor of the forward (aux_mode 1)
          float m[VE];
          VecT<T>::load(xp + (size_t)k * xstep, m);
#pragma unroll
          for (int e = 0; e < VE; ++e) v[e] *= m[e];
        }
        if (rp) {
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(rr[g][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += __uint_as_float(rr[g][e] << 16);
              v[2 * e + 1] += __uint_as_float(rr[g][e] & 0xffff0000u);
            }
          }
        }
        if constexpr (bnb) {   // v = gradient of the BatchNorm + activation output: g = v * act'(.), sums of g and g * zhat
          float zf[VE], yf[VE];
          if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { zf[e] = __uint_as_float(zz[g][e]); yf[e] = __uint_as_float(oo[g][e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              zf[2 * e] = __uint_as_float(zz[g][e] << 16); zf[2 * e + 1] = __uint_as_float(zz[g][e] & 0xffff0000u);
              yf[2 * e] = __uint_as_float(oo[g][e] << 16); yf[2 * e + 1] = __uint_as_float(oo[g][e] & 0xffff0000u);
            }
          }
          if (!op) {
#pragma unroll
            for (int e = 0; e < VE; ++e) yf[e] = zf[e] * bfs[e] + bfh[e];   // same expression (and contraction) as scale_shift_act
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const float gm = p.bnb_act == CAVP_ACT_RELU ? (yf[e] > 0.f ? 1.f : 0.f) : (p.bnb_act == CAVP_ACT_LEAKY ? (yf[e] > 0.f ? 1.f : 0.01f) : 1.f);
            const float gv = live ? v[e] * gm : 0.f;
            bs0[e] += gv;
            if constexpr (sizeof(T) == 4) bs1[e] = fmaf(gv, zf[e] - bmu[e], bs1[e]);
            else bs1[e] = fmaf(gv, zf[e], bs1[e]);
            v[e] = gv;
          }
        }
        if (!BNB && p.aux_mode == 1) {   // GELU forward: the derivative goes to aux, the pre-activation is never stored
          float m[VE];
#pragma unroll
          for (int e = 0; e < VE; ++e) gelu_and_grad(v[e], v[e], m[e]);
          if (live) VecT<T>::store(xp + (size_t)k * xstep, m);
        } else {
          if constexpr (!BNB) apply_act_vec<VE, sizeof(T) == 2>(v, p.act);
        }
        u32x4_t o;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        }
        if (live && (!CAVP_DBG(p, 32) || o[0] == 0x12345678u)) {
          if constexpr (UP) *(u32x4_t*)((T*)p.y + (size_t)opk * p.ldy + ec) = o;
          else *(u32x4_t*)(yp + (size_t)k * ystep) = o;
        }
      }
    }
    if constexpr (bnb) {
      // the NT / CH threads of a channel group -> one pair per channel and pixel tile, in a fixed order (deterministic, no atomics):
      // xor-shuffles over the 64 / CH lanes of a wave that share the group, then the NW wave sums through LDS (the staging buffer is
      // free once every thread has read its chunks).  (First version: every thread's 2 VE sums to LDS and a serial loop over the
      // NT / CH rows by BC threads - 4000 cycles per tile.)
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float mu = bcoef[2 * BC + ecc + e], rs = bcoef[3 * BC + ecc + e];
        bs1[e] = sizeof(T) == 4 ? rs * bs1[e] : rs * (bs1[e] - mu * bs0[e]);
      }
      static_assert(CH == 8 || CH == 16 || CH == 32 || CH == 64, "lanes of a wave that share a channel group");
      if constexpr (MC * MP * 4 >= 64) {
        // (the 128 x 128 tile sits at 256 VGPRs: the two-register results of the lane swaps below spilled 67 of them - 14.0 -> 14.2 ms
        // per step; it keeps the LDS-crossbar shuffles)
#pragma unroll
        for (int m = CH; m < 64; m <<= 1) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            bs0[e] += __shfl_xor(bs0[e], m, 64);
            bs1[e] += __shfl_xor(bs1[e], m, 64);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          if constexpr (CH <= 8) { bs0[e] = xor_add<8>(bs0[e]); bs1[e] = xor_add<8>(bs1[e]); }
          if constexpr (CH <= 16) { bs0[e] = xor_add<16>(bs0[e]); bs1[e] = xor_add<16>(bs1[e]); }
          if constexpr (CH <= 32) { bs0[e] = xor_add<32>(bs0[e]); bs1[e] = xor_add<32>(bs1[e]); }
        }
      }
      constexpr int GPW = CH < 64 ? 1 : CH / 64;   // (CH <= 64 for every tile: one channel-group set per wave)
      static_assert(GPW == 1, "a wave covers all channel groups of the tile");
      // (raw barriers: the chunk loop's global stores need not be acknowledged before the tile sums go through LDS)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      float* red = (float*)smem;   // [NW][BC][2]
      if (lane < CH) {
#pragma unroll
        for (int e = 0; e < VE; ++e) *(float2*)(red + ((size_t)wave * BC + ecc + e) * 2) = make_float2(bs0[e], bs1[e]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      if (tid < BC && c_base + tid < p.Cout) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float2 q = *(const float2*)(red + ((size_t)w * BC + tid) * 2);
          a0 += q.x; a1 += q.y;
        }
        *(float2*)(p.bnb_part + ((size_t)tp * p.Cout + c_base + tid) * 2) = make_float2(a0, a1);
      }
    }
    continue;
  }
#pragma unroll
  for (int a = 0; a < MC; ++a) {
#pragma unroll
    for (int b = 0; b < MP; ++b) {
      const int c = c_base + wc0 + a * 16 + lgrp * 4;
      int pix = p_base + wp0 + b * 16 + lrow;
      if (pix >= p.M) continue;
      if constexpr (UP) {
        if (p.up_par) {   // (never with split-K: the planner keeps parity-ordered launches unsplit)
          int n_, h_, w_;
          pix = par_out_pixel(p, pix, n_, h_, w_);
          if (pix < 0) continue;
        }
      }
      const f32x4_t v = acc[a][b];
      if (p.splitk > 1) {
        float* dst = p.partial + ((size_t)z * p.M + pix) * p.Cout + c;
        if ((p.Cout & 3) == 0) {
          if (c < p.Cout) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < p.Cout) dst[e] = v[e];
        }
      } else {
        epilogue_store4<T>(p, v[0], v[1], v[2], v[3], pix, c);
      }
    }
  }
  } while (0);
  // tail: when everybody has finished reading the staging LDS, set up the next tile and request its first stage(s) - behind this
  // tile's global stores, which are still on their way
  if (vb + (int)gridDim.x < p.nblk) {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    start_tile(vb + (int)gridDim.x);
  }
  }  // persistent loop
#ifdef CAVP_PROFILE
  CAVP_TLI({
    const unsigned long long now = __builtin_readcyclecounter();
    if (tl_tiles) { tl_s[0] += tl_t1 - tl_t0; tl_s[1] += tl_t2 - tl_t1; tl_s[2] += tl_t3 - tl_t2; tl_s[3] += now - tl_t3; }
    if (lane == 0) {
      g_igemm_tl[0] = tl_tiles; g_igemm_tl[1] = tl_iters; g_igemm_tl[2] = tl_first;
      g_igemm_tl[3] = tl_s[0]; g_igemm_tl[4] = tl_s[1]; g_igemm_tl[5] = tl_s[2]; g_igemm_tl[6] = tl_s[3]; g_igemm_tl[7] = now - tl_entry;
    }
  });
#endif
}

// Reduce split-K slabs (deterministic order) and apply the epilogue.  One thread per 4 consecutive channels.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const IgemmParams p) {
  const int cq = (p.Cout + 3) >> 2;
  const long long total = (long long)p.M * cq;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int pix = total <= 0x7fffffffll ? fast_div((int)i, p.div_cq_m, p.div_cq_s) : (int)(i / cq);
    const int c = (int)(i - (long long)pix * cq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((p.Cout & 3) == 0) {   // slabs are 16-byte aligned rows: one vector load per split (summed in split order)
      const float* src = p.partial + (size_t)pix * p.Cout + c;
      const size_t zstep = (size_t)p.M * p.Cout;
#pragma unroll 4
      for (int z = 0; z < p.splitk; ++z) {
        const float4 t = *(const float4*)(src + (size_t)z * zstep);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
    } else {
      for (int z = 0; z < p.splitk; ++z) {
        const float* src = p.partial + ((size_t)z * p.M + pix) * p.Cout + c;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < p.Cout) v[e] += src[e];
      }
    }
    epilogue_store4<T>(p, v[0], v[1], v[2], v[3], pix, c);
  }
}

// --------------------------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------------------------
namespace {

struct TileCfg {
  int id, BC, BP;
  float eff;  // relative per-WG efficiency (bigger tiles reuse LDS operands more)
};
const TileCfg kTiles[] = {
    // eff ~ operand bytes per flop relative to 128x128 (the K loop is bound by the LDS-DMA path, profiles/r01_notes.md),
    // softened for the small tiles whose launches are dominated by fixed costs; checked against tools/bench_conv.py sweeps
    {1, 128, 128, 1.00f}, {2, 64, 128, 0.80f}, {3, 64, 64, 0.70f}, {4, 128, 64, 0.80f},   // (tools/eff_sweep.sh: end-to-end sweep)
    {5, 128, 32, 0.45f},  {6, 16, 128, 0.25f}, {7, 32, 128, 0.45f},
    {8, 256, 128, 1.00f}, {9, 128, 256, 1.00f},  // 8 waves, 3 stages (eff set to 1.0 until measured: kBigEff below)
    {10, 256, 256, 2.00f},                       // conv_igemm_big.hip: 8 waves in two ping-pong groups, one workgroup per CU
    {11, 64, 64, 0.70f}, {12, 64, 64, 0.70f}, {13, 128, 64, 0.80f}, {14, 64, 128, 0.80f},   // deep rings (4 / 8 / 4 / 4 stages) for under-filled launches
};
inline int tile_stages(int id) { return id == 12 ? 8 : id >= 11 ? 4 : id == 10 ? 2 : id >= 8 ? 3 : 2; }
inline bool tile_is_big(int id) { return id == 10; }
inline bool tile_has_bnb(int id) { return id == 1 || id == 2 || id == 3 || id == 4 || id == 11 || id == 13 || id == 14; }   // launch_tile
inline int tile_wp(int id) { return id == 5 ? 1 : (id == 6 || id == 7 || id == 9) ? 4 : 2; }   // waves along the pixel dimension (launch_tile)
// A/B knob: CAVP_IGEMM_EFF="e1,e2,...,e9" overrides the efficiency column of kTiles (time-model sweeps without a rebuild)
inline double tile_eff(const TileCfg& t) {
  static double ov[16];
  static const bool have = [] {
    const char* e = cavp_knob_str("CAVP_IGEMM_EFF");
    if (!e) return false;
    int i = 1;
    for (const char* q = e; *q && i < 16; ++i) {
      ov[i] = atof(q);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    for (; i < 16; ++i) ov[i] = 0.0;
    return true;
  }();
  return (have && t.id < 16 && ov[t.id] > 0.0) ? ov[t.id] : (double)t.eff;
}
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <typename T, int BC, int BP, int WC, int WP, bool UP, int NS = 2, bool BNB = false>
hipError_t launch_cfg(const IgemmParams& p, int nblk, hipStream_t s) {
  constexpr int lds = NS * (BC + BP) * 128 + WP * BC * 8 + (BNB ? BC * 16 : 0);   // + the per-wave BatchNorm-statistics partials (+ BNB: coefficients)
  static_assert(BP * BC * 4 <= NS * (BC + BP) * 128, "epilogue staging must fit in the K-loop LDS");
  static_assert(BC <= 64 * WC * WP, "tile_stats: one thread per output channel of the tile");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BC, BP, WC, WP, UP, NS, BNB>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  int bpc = (160 * 1024) / lds;   // resident workgroups per CU by LDS ...
  if (bpc > 4) bpc = 4;
  if (bpc < 1) bpc = 1;
  // ... and by registers: the 64 x 128 / 128 x 64 tiles hold 194 .. 233 VGPRs = two waves per SIMD = TWO workgroups per CU, while their 49 KiB of
  // LDS would admit three.  A persistent grid of 3 x 256 leaves 256 workgroups queued behind the resident 512 until those have walked ALL their
  // tiles (the surplus of a persistent grid starts when a resident workgroup EXITS): two waves of workgroups instead of 1.5.  (Grid only: the
  // planner's time model keeps its LDS-based slot count, so tile choices - and with them every summation order - stay what they were.)
  static int occ = -1;
  if (occ < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)igemm_kernel<T, BC, BP, WC, WP, UP, NS, BNB>, 64 * WC * WP, lds) != hipSuccess || n < 1)
      n = 4;
    occ = n;
  }
  if (bpc > occ) bpc = occ;
  static const int bpc_cap = cavp_knob_int("CAVP_IGEMM_BPC", 0);   // A/B knob: resident workgroups per CU
  if (bpc_cap > 0 && bpc > bpc_cap) bpc = bpc_cap;
  static const bool persistent = cavp_knob_int("CAVP_IGEMM_PERSISTENT", 1) != 0;
  const int grid = (persistent && nblk > bpc * 256) ? bpc * 256 : nblk;
  IgemmParams q = p;
  q.nblk = nblk;
  igemm_kernel<T, BC, BP, WC, WP, UP, NS, BNB><<<dim3(grid), dim3(64 * WC * WP), lds, s>>>(q);
  return hipGetLastError();
}

template <typename T, bool UP>
hipError_t launch_tile(int id, const IgemmParams& p, int nblk, hipStream_t s) {
  if (p.bnb_z) {   // BatchNorm-backward statistics: the tiles the backbone's data gradients run on (tile_has_bnb)
    switch (id) {
      case 1: return launch_cfg<T, 128, 128, 2, 2, UP, 2, true>(p, nblk, s);
      case 2: return launch_cfg<T, 64, 128, 2, 2, UP, 2, true>(p, nblk, s);
      case 3: return launch_cfg<T, 64, 64, 2, 2, UP, 2, true>(p, nblk, s);
      case 4: return launch_cfg<T, 128, 64, 2, 2, UP, 2, true>(p, nblk, s);
      case 11: return launch_cfg<T, 64, 64, 2, 2, UP, 4, true>(p, nblk, s);
      case 13: return launch_cfg<T, 128, 64, 2, 2, UP, 4, true>(p, nblk, s);
      case 14: return launch_cfg<T, 64, 128, 2, 2, UP, 4, true>(p, nblk, s);
      default: return hipErrorInvalidValue;
    }
  }
  switch (id) {
    case 1: return launch_cfg<T, 128, 128, 2, 2, UP>(p, nblk, s);
    case 2: return launch_cfg<T, 64, 128, 2, 2, UP>(p, nblk, s);
    case 3: return launch_cfg<T, 64, 64, 2, 2, UP>(p, nblk, s);
    case 4: return launch_cfg<T, 128, 64, 2, 2, UP>(p, nblk, s);
    case 5: return launch_cfg<T, 128, 32, 4, 1, UP>(p, nblk, s);
    case 6: return launch_cfg<T, 16, 128, 1, 4, UP>(p, nblk, s);
    case 7: return launch_cfg<T, 32, 128, 1, 4, UP>(p, nblk, s);
    case 8: return launch_cfg<T, 256, 128, 4, 2, UP, 3>(p, nblk, s);
    case 9: return launch_cfg<T, 128, 256, 2, 4, UP, 3>(p, nblk, s);
    case 10:
      if constexpr (sizeof(T) == 2 && !UP) return cavp_launch_igemm_big(p, nblk, s);
      return hipErrorInvalidValue;
    case 11: return launch_cfg<T, 64, 64, 2, 2, UP, 4>(p, nblk, s);
    case 12: return launch_cfg<T, 64, 64, 2, 2, UP, 8>(p, nblk, s);
    case 13: return launch_cfg<T, 128, 64, 2, 2, UP, 4>(p, nblk, s);
    case 14: return launch_cfg<T, 64, 128, 2, 2, UP, 4>(p, nblk, s);
    default: return hipErrorInvalidValue;
  }
}

struct Plan {
  IgemmParams p;
  int tile_id;
  int direct_epi;  // testing: force the direct (MFMA-layout) epilogue
  int nblk;
  size_t ws_bytes;
  int status;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

Plan make_plan(const cavp_conv_desc* d_in, bool allow_big = true, bool allow_par = true) {
  cavp_conv_desc dd{};
  if (d_in) dd = *d_in;
  if (dd.aux_mode != 0 || dd.res_rows > 0) dd.splitk = 1;   // the fused token-path epilogues live in the 16-byte epilogue only
  const cavp_conv_desc* d = d_in ? &dd : nullptr;
  Plan pl{};
  pl.status = CAVP_OK;
  if (!d) { pl.status = CAVP_ERR_BAD_ARG; return pl; }
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 ||
      d->stride <= 0 || d->dil <= 0 || d->pad < 0 || d->ldx < d->Cin || d->ldy < d->Cout) {
    pl.status = CAVP_ERR_BAD_ARG;
    return pl;
  }
  if (d->dtype != CAVP_F32 && d->dtype != CAVP_BF16) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  const int VE = d->dtype == CAVP_F32 ? 4 : 8;
  if (d->Cin % VE || d->ldx % VE) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  if (d->KH * d->KW > 9) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  IgemmParams& p = pl.p;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.ldx = d->ldx; p.Cout = d->Cout; p.ldy = d->ldy;
  p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.ldr = d->ldr; p.act = d->act;
  p.stride_w = d->stride_w > 0 ? d->stride_w : d->stride;
  const int up = d->up > 1 ? d->up : 1;
  if (up > 1) {  // transposed conv: stride-1 gather over the zero-upsampled input, explicit output extent
    if ((up & (up - 1)) || d->stride != 1 || d->Ho <= 0 || d->Wo <= 0) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
    p.Ho = d->Ho; p.Wo = d->Wo;
  } else {
    p.Ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / (d->stride_w > 0 ? d->stride_w : d->stride) + 1;
  }
  p.up_mask = up - 1;
  p.up_shift = 0;
  while ((1 << p.up_shift) < up) ++p.up_shift;
  if (p.Ho <= 0 || p.Wo <= 0) { pl.status = CAVP_ERR_BAD_ARG; return pl; }
  const long long M = (long long)d->N * p.Ho * p.Wo;
  if (M > 0x7fffffffll / 4) { pl.status = CAVP_ERR_UNSUPPORTED; return pl; }
  p.M = (int)M;
  fast_div_prepare(p.Ho * p.Wo, &p.div_hw_m, &p.div_hw_s);
  fast_div_prepare(p.Wo, &p.div_w_m, &p.div_w_s);
  fast_div_prepare((d->Cout + 3) >> 2, &p.div_cq_m, &p.div_cq_s);
  p.K = d->KH * d->KW * d->Cin;
  // live taps: tap (kh,kw) is live iff some output row/col maps it inside the image
  p.ntaps = 0;
  p.taps = 0;
  for (int kh = 0; kh < d->KH; ++kh) {
    bool hlive = false;
    for (int ho = 0; ho < p.Ho && !hlive; ++ho) {
      const int hi = ho * d->stride - d->pad + kh * d->dil;
      hlive = hi >= 0 && (hi % up) == 0 && hi / up < d->H;
    }
    for (int kw = 0; kw < d->KW; ++kw) {
      bool wlive = false;
      for (int wo = 0; wo < p.Wo && !wlive; ++wo) {
        const int wi = wo * (d->stride_w > 0 ? d->stride_w : d->stride) - d->pad + kw * d->dil;
        wlive = wi >= 0 && (wi % up) == 0 && wi / up < d->W;
      }
      if (hlive && wlive) {
        const int es_ = d->dtype == CAVP_F32 ? 4 : 2;
        p.taps |= (unsigned long long)(kh * d->KW + kw) << (4 * p.ntaps);
        p.tap_dh[p.ntaps] = kh * d->dil;
        p.tap_dw[p.ntaps] = kw * d->dil;
        p.tap_xoff[p.ntaps] = (kh * d->dil * d->W + kw * d->dil) * d->ldx * es_;
        p.tap_woff[p.ntaps] = (kh * d->KW + kw) * d->Cin * es_;
        ++p.ntaps;
      }
    }
  }
  const int BK = 8 * VE;
  p.cpt = cdiv(d->Cin, BK);
  p.iters = p.ntaps * p.cpt;
  // ---- tile + split-K choice: a small time model, fitted to tools/bench_conv.py on MI355X ----
  //   t = rounds * (flops of one padded workgroup) / (rate per resident slot)  +  split-K slab traffic
  // rate: the 2-stage LDS-DMA structure tops out at ~600 TF/s (bf16) / ~100 TF/s (f32) with 128x128 tiles, because
  // the operand fill rate of the chip (~9 TB/s global->LDS), not the MFMA pipe, is its ceiling; smaller tiles move
  // more operand bytes per flop (eff below).  Resident workgroups per CU follow from the tile's LDS footprint.
  int best = -1, best_sk = 1;
  pl.direct_epi = (d->tile / 1000) % 10;
  // d->tile = id + 1000 * (direct epilogue: testing) + profiling digits (hundreds, ten- and hundred-thousands: pieces of the
  // kernel switched off for the K-loop anatomy; honoured by -DCAVP_PROFILE builds only, the product kernels ignore p.dbg)
  p.dbg = (d->tile / 100) % 10 + ((d->tile / 10000) % 10) * 8 + ((d->tile / 100000) % 10) * 16;
  {   // -DCAVP_PROFILE builds: CAVP_IGEMM_DBG=<bits> switches the same pieces off for EVERY launch (whole-step anatomy; results are garbage)
    static const int all_dbg = cavp_knob_int("CAVP_IGEMM_DBG", 0);
    p.dbg |= all_dbg;
  }
  const int want_tile = d->tile % 100;
  const double peak = d->dtype == CAVP_F32 ? 100e12 : 600e12;
  static const double slab_bw = cavp_knob_double("CAVP_IGEMM_SLAB_TBS", 3.0) * 1e12;   // A/B knob
  const int sk_opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  double best_t = 1e30;
  for (int i = 0; i < kNumTiles; ++i) {
    const TileCfg& t = kTiles[i];
    if (want_tile > 0 && t.id != want_tile) continue;
    const long long nwg = (long long)cdiv(p.Cout, t.BC) * cdiv(p.M, t.BP);
    int bpc = (160 * 1024) / (tile_stages(t.id) * (t.BC + t.BP) * 128 + tile_wp(t.id) * t.BC * 8);
    if (bpc > 4) bpc = 4;
    if (t.id >= 8 && want_tile != t.id) continue;  // big tiles: explicit request only until the time model is refitted
    if (tile_is_big(t.id) && (d->dtype != CAVP_BF16 || up > 1 || d->splitk > 1 || d->Cout % 8 || d->ldy % 8 ||
                              (d->ldr && d->ldr % 8)))
      continue;
    if (tile_is_big(t.id)) bpc = 1;
    const double slots = 256.0 * bpc;
    for (int sk : sk_opts) {
      if (d->splitk > 0 && sk != 1) continue;
      if (tile_is_big(t.id) && sk != 1) continue;
      int use_sk = d->splitk > 0 ? d->splitk : sk;
      if (use_sk > 1 && (use_sk > p.iters / 4)) continue;
      if (use_sk > p.iters) use_sk = p.iters > 0 ? p.iters : 1;
      const double blocks = (double)nwg * use_sk;
      const double rounds = (double)((long long)((blocks + slots - 1) / slots));
      const double wg_flops = 2.0 * t.BC * t.BP * (double)BK * ((double)p.iters / use_sk);
      double tt = rounds * wg_flops / (peak * tile_eff(t) / slots) + 2e-6;
      // HBM floor (input once, output once): on bandwidth-bound layers every tile ties on it and the smaller tile (more
      // workgroups in flight, cheaper prologue / epilogue each) wins the tie
      const double es_ = d->dtype == CAVP_F32 ? 4.0 : 2.0;
      const double t_mem = ((double)d->N * d->H * d->W * d->Cin + (double)p.M * p.Cout) * es_ / 3.5e12 + 2e-6;
      if (tt < t_mem) tt = t_mem;
      tt += 2e-7 * (double)(t.BC * t.BP) / 16384.0;
      if (use_sk > 1) tt += (2.0 * use_sk + 1.0) * p.M * p.Cout * 4.0 / slab_bw + 4e-6;
      if (tt < best_t) { best_t = tt; best = i; best_sk = use_sk; }
    }
  }
  if (best < 0) { pl.status = CAVP_ERR_BAD_ARG; return pl; }
  // Under-filled launches (the 14x14 layers: <= 512 tiles of 64x64, i.e. at most two workgroups per CU whatever the
  // split) with a long K loop: the 4-stage ring WITHOUT split-K (bench_conv on MI355X: 3x3 256->256 at 32x14x14 28.6 ->
  // 18.5 us, 1x1 2048->256 23.9 -> 16.4 us; no slab pass, and the fused BatchNorm statistics stay available).  With more
  // tiles than that the 2-stage tile at four workgroups per CU is faster.
  // (not for very long K loops: the ASPP 3x3 2048 -> 256 convs, 288 K tiles, are faster split over K on 128x128 tiles:
  // 76 vs 100 us)
  // (Extending this to short K loops - 4 .. 15 K tiles: the PVTv2 kv / stage-4 linears at B = 8 - bought 0.15 ms per config-#4 step.
  // Not kept: it changes the summation order of the token linears (no split-K), and two end-to-end parity tests sit on inputs
  // where a 1e-6 change of a pre-activation flips ONE ReLU mask element under a large gradient (tools/probes/stage_grad_diff.py:
  // 1.6e-3 of the gradient norm from that single element; the tile itself agrees bit for bit with the 2-stage one).)
  if (want_tile == 0 && d->splitk <= 0 && p.iters >= 16 && p.iters <= 96 && (long long)cdiv(p.Cout, 64) * cdiv(p.M, 64) <= 512) {
    for (int i = 0; i < kNumTiles; ++i)
      if (kTiles[i].id == 11) { best = i; best_sk = 1; }
  }
  // A/B probe (profile builds: CAVP_IGEMM_UP1X1_TILE=<id>): tile of the 1x1 stride-2 data gradients, whose launches are bound by the four
  // streams of their epilogue, not by their K loop
  {
    static const int up1 = cavp_knob_int("CAVP_IGEMM_UP1X1_TILE", 0);
    if (up1 > 0 && want_tile == 0 && up == 2 && p.ntaps == 1 && d->splitk <= 0)
      for (int i = 0; i < kNumTiles; ++i)
        if (kTiles[i].id == up1) { best = i; best_sk = 1; }
  }
  // The 256x256 ping-pong tile (conv_igemm_big.hip) by rule, not by the time model: measured on MI355X (bench_conv,
  // profiles/r02_notes.md) it wins where its whole-tile quantisation is harmless - the output is (nearly) a multiple of 256
  // channels wide, there are at least four K tiles to amortise its ~6 us epilogue, and the tiles fill whole rounds of 256
  // workgroups to >= 75 % (3x3 304->256 at 2B x 56 x 56: -10 %, 304->1216 token linear: -24 %); it loses on 304- / 128- /
  // 64-channel outputs.
  if (allow_big && want_tile == 0 && d->dtype == CAVP_BF16 && up == 1 && d->splitk <= 1 && d->Cout % 8 == 0 && d->ldy % 8 == 0 &&
      (d->ldr == 0 || d->ldr % 8 == 0) && (d->Cout % 256 == 0 || d->Cout % 256 >= 192) && p.iters >= 4) {
    const long long nb = (long long)cdiv(p.Cout, 256) * cdiv(p.M, 256);
    const long long rounds = (nb + 255) / 256;
    if (nb >= 256 && (double)nb / (double)(rounds * 256) >= 0.75) {
      for (int i = 0; i < kNumTiles; ++i)
        if (tile_is_big(kTiles[i].id)) { best = i; best_sk = 1; }
    }
  }
  const TileCfg& t = kTiles[best];
  pl.tile_id = t.id;
  p.tiles_c = cdiv(p.Cout, t.BC);
  fast_div_prepare(p.tiles_c, &p.div_tc_m, &p.div_tc_s);
  // Stride-2 data gradients (up = 2): parity-ordered pixel tiles (igemm_params.h).  As one pixel stream in image order every tile holds
  // all four output parities and so walks all live taps with three quarters of its rows zero-filled: 9 of 36 tap visits of a 3x3 do
  // any work, 1 of 4 of a 1x1 down-sample (profiles/r06_notes.md: 226 us per step in four launches whose roofs add up to 22).
  static const bool par_on = cavp_knob_int("CAVP_IGEMM_UP_PARITY", 1) != 0;   // A/B knob (profile builds)
  p.up_par = 0;
  if (par_on && allow_par && up == 2 && d->splitk <= 1) {
    const int hq = (p.Ho + 1) / 2, wq = (p.Wo + 1) / 2;
    const long long mq = ((long long)d->N * hq * wq + t.BP - 1) / t.BP * t.BP;
    if (4 * mq <= 0x7fffffffll / 4) {
      p.up_par = 1;
      p.par_m = p.M;
      p.par_mq = (int)mq; p.par_hq = hq; p.par_wq = wq;
      p.M = (int)(4 * mq);
      fast_div_prepare(p.par_mq, &p.div_mq_m, &p.div_mq_s);
      fast_div_prepare(hq * wq, &p.div_hwq_m, &p.div_hwq_s);
      fast_div_prepare(wq, &p.div_wq_m, &p.div_wq_s);
      for (int q = 0; q < 4; ++q) {
        const int a = q >> 1, b = q & 1;
        p.par_nt[q] = 0;
        p.par_taps[q] = 0;
        for (int ti = 0; ti < p.ntaps; ++ti) {   // the tap reads input position (ho - pad + dh) / 2: live iff that is even in both directions
          if (((a - d->pad + p.tap_dh[ti]) & 1) == 0 && ((b - d->pad + p.tap_dw[ti]) & 1) == 0) {
            p.par_taps[q] |= (unsigned long long)ti << (4 * p.par_nt[q]);
            ++p.par_nt[q];
          }
        }
      }
      best_sk = 1;
    }
  }
  p.tiles_p = cdiv(p.M, t.BP);
  const int nwg = p.tiles_c * p.tiles_p;
  int sk = best_sk;
  if (sk > p.iters) sk = p.iters > 0 ? p.iters : 1;
  if (sk < 1) sk = 1;
  p.splitk = sk;
  pl.nblk = nwg * sk;
  pl.ws_bytes = sk > 1 ? (size_t)sk * p.M * p.Cout * sizeof(float) : 0;
  static const bool trace = cavp_knob_str("CAVP_IGEMM_TRACE") != nullptr;   // debugging: the plan of every make_plan call
  if (trace)
    fprintf(stderr, "[igemm plan] M=%d K=%d Cout=%d k%dx%d up=%d -> tile %d (%dx%d) splitk %d, %d workgroups, model %.1f us\n", p.M,
            p.K, p.Cout, d->KH, d->KW, up, t.id, t.BC, t.BP, sk, pl.nblk, best_t * 1e6);
  return pl;
}

inline bool aligned(const void* ptr, size_t a) { return ((uintptr_t)ptr % a) == 0; }

}  // namespace

// Tail split of a 256x256-tile launch.  The persistent grid runs whole rounds of 256 tiles; the head 3x3 convs of the
// training step (2B x 56 x 56 pixels, 256 channels: 784 tiles) leave a fourth round with 16 tiles - 86 us of a 335 us launch
// for 2 % of the work.  When the last round is at most a quarter full, the leading images that fill the whole rounds go to
// the big tile and the remaining images (here 2 of 64) to an ordinary launch of the small tiles (~25 us).  Returns the number
// of leading images, 0 = no split.
static bool g_tail_split = true;
extern "C" int cavp_set_tail_split(int on) { g_tail_split = on != 0; return CAVP_OK; }

static int tail_split_images(const cavp_conv_desc* d, const Plan& pl, bool with_tile_stats) {
  if (!g_tail_split) return 0;
  if (!tile_is_big(pl.tile_id) || d->tile % 100 != 0 || d->N < 2 || d->res_rows > 0) return 0;
  const long long nb = (long long)pl.p.tiles_c * pl.p.tiles_p;
  const long long rounds = (nb + 255) / 256, last = nb - (rounds - 1) * 256;
  if (rounds < 2 || last > 64) return 0;
  const long long hw = (long long)pl.p.Ho * pl.p.Wo;
  const long long ptiles = (rounds - 1) * 256 / pl.p.tiles_c;   // pixel tiles that fit the whole rounds
  long long n1 = ptiles * 256 / hw;
  if (n1 >= d->N) n1 = d->N - 1;
  // per-tile statistics: the tail's first pixel must start a 128-row statistics tile
  while (with_tile_stats && n1 > 0 && (n1 * hw) % 128) --n1;
  if (n1 < 1 || (d->N - n1) * hw > 65536) return 0;   // the tail must stay a small problem
  cavp_conv_desc da = *d;
  da.N = (int)n1;
  if (!tile_is_big(make_plan(&da).tile_id)) return 0;   // (the leading part must keep the tile the caller's layouts assume)
  return (int)n1;
}

// Fused BatchNorm statistics are available when the launch uses the LDS-staged epilogue: no split-K, vector-aligned
// Cout / ldy.  Returns 1 and the tile geometry, else 0 (caller falls back to cavp_colsum / cavp_colstats).
extern "C" int cavp_conv2d_tile_stats_layout(const cavp_conv_desc* d, int32_t* tiles, int32_t* rows_per_tile) {
  Plan pl = make_plan(d);
  if (pl.status != CAVP_OK || !tiles || !rows_per_tile) return 0;
  const int VE = d->dtype == CAVP_F32 ? 4 : 8;
  if (pl.p.splitk != 1 || pl.direct_epi || d->Cout % VE || d->ldy % VE) return 0;
  int BP = 0;
  for (int i = 0; i < kNumTiles; ++i)
    if (kTiles[i].id == pl.tile_id) BP = kTiles[i].BP;
  if (tile_is_big(pl.tile_id)) {   // one statistics tile per 128-row wave slab
    *tiles = (pl.p.M + 127) / 128;
    *rows_per_tile = 128;
    return 1;
  }
  *tiles = pl.p.tiles_p;
  *rows_per_tile = BP;
  return 1;
}

extern "C" size_t cavp_conv2d_workspace_bytes(const cavp_conv_desc* d) {
  Plan pl = make_plan(d);
  if (pl.status != CAVP_OK) return 0;
  size_t need = pl.ws_bytes;
  if (pl.p.up_par) {   // a launch with a per-image bias / auxiliary tensor re-plans without the parity-ordered tiles: that plan may split K
    Plan lin = make_plan(d, true, false);
    if (lin.status == CAVP_OK && lin.ws_bytes > need) need = lin.ws_bytes;
  }
  if (tile_is_big(pl.tile_id)) {   // the launch re-plans without the 256x256 tile when an operand is not 16-byte aligned: that
    Plan alt = make_plan(d, false);   // plan may split K
    if (alt.status == CAVP_OK && alt.ws_bytes > need) need = alt.ws_bytes;
    const int n1 = tail_split_images(d, pl, true);   // ... and the tail of a split launch runs on the small tiles
    if (n1 > 0) {
      cavp_conv_desc db = *d;
      db.N = d->N - n1;
      for (int forced = 0; forced <= 2; forced += 2) {   // (tile 2, unsplit, when the launch carries per-tile statistics)
        db.tile = forced;
        db.splitk = forced ? 1 : d->splitk;
        Plan tail = make_plan(&db);
        if (tail.status == CAVP_OK && tail.ws_bytes > need) need = tail.ws_bytes;
      }
    }
  }
  return need;
}

extern "C" int cavp_conv2d_nhwc(const cavp_conv_desc* d, const void* x, const void* w, const float* scale,
                                const float* shift, const float* nbias, const void* residual, void* y, void* workspace,
                                size_t workspace_bytes, float* tile_stats, void* stream) {
  if (d && (d->aux_mode != 0)) return CAVP_ERR_BAD_ARG;   // the auxiliary tensor travels through cavp_conv2d_nhwc_aux
  return cavp_conv2d_nhwc_aux(d, x, w, scale, shift, nbias, residual, y, nullptr, workspace, workspace_bytes, tile_stats, stream);
}

static int conv2d_launch(const cavp_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                         const float* nbias, const void* residual, void* y, void* aux, void* workspace, size_t workspace_bytes,
                         float* tile_stats, void* stream,
                         bool allow_split = true, const cavp_bnbwd_args* bnb = nullptr);

extern "C" int cavp_conv2d_nhwc_aux(const cavp_conv_desc* d, const void* x, const void* w, const float* scale,
                                    const float* shift, const float* nbias, const void* residual, void* y, void* aux,
                                    void* workspace, size_t workspace_bytes, float* tile_stats, void* stream) {
  return conv2d_launch(d, x, w, scale, shift, nbias, residual, y, aux, workspace, workspace_bytes, tile_stats, stream);
}

// Launches that can carry the fused BatchNorm-backward statistics: the LDS-staged epilogue of the 4-wave tiles, unsplit (the same
// condition as the forward's fused statistics, minus the 256 x 256 tile, whose epilogue has no registers to spare).  Returns 1 and
// the geometry of the partial sums, else 0 (caller: cavp_bn_act_bwd_reduce as before).
extern "C" int cavp_conv2d_bnbwd_layout(const cavp_conv_desc* d, int32_t* tiles, int32_t* rows_per_tile) {
  if (!d || d->aux_mode != 0 || d->res_rows != 0) return 0;
  Plan pl = make_plan(d);
  if (pl.status != CAVP_OK || !tile_has_bnb(pl.tile_id)) return 0;
  return cavp_conv2d_tile_stats_layout(d, tiles, rows_per_tile);
}

extern "C" int cavp_conv2d_nhwc_bnbwd(const cavp_conv_desc* d, const void* x, const void* w, const void* residual, void* y,
                                      const cavp_bnbwd_args* b, void* workspace, size_t workspace_bytes, void* stream) {
  if (!d || !b || !b->z || !b->mean || !b->rstd || !b->partials || d->aux_mode != 0 || d->res_rows != 0 || d->act != CAVP_ACT_NONE)
    return CAVP_ERR_BAD_ARG;
  if (b->act != CAVP_ACT_NONE && b->act != CAVP_ACT_RELU && b->act != CAVP_ACT_LEAKY) return CAVP_ERR_UNSUPPORTED;
  if (!b->out && b->act != CAVP_ACT_NONE && (!b->fwd_scale || !b->fwd_shift)) return CAVP_ERR_BAD_ARG;
  const int VE = d->dtype == CAVP_F32 ? 4 : 8;
  if (b->ld_z < d->Cout || b->ld_z % VE || !aligned(b->z, 16) || (b->out && (b->ld_out < d->Cout || b->ld_out % VE || !aligned(b->out, 16))))
    return CAVP_ERR_ALIGN;
  int32_t tiles = 0, rpt = 0;
  if (!cavp_conv2d_bnbwd_layout(d, &tiles, &rpt)) return CAVP_ERR_UNSUPPORTED;
  return conv2d_launch(d, x, w, nullptr, nullptr, nullptr, residual, y, nullptr, workspace, workspace_bytes, nullptr, stream, false, b);
}

static int conv2d_launch(const cavp_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                         const float* nbias, const void* residual, void* y, void* aux, void* workspace, size_t workspace_bytes,
                         float* tile_stats, void* stream,
                         bool allow_split, const cavp_bnbwd_args* bnb) {
  if (!d || !x || !w || !y) return CAVP_ERR_BAD_ARG;
  const bool fused = d->aux_mode != 0 || (residual && d->res_rows > 0);
  if (d->aux_mode < 0 || d->aux_mode > 2 || (d->aux_mode != 0) != (aux != nullptr) || (d->aux_mode == 1 && d->act != CAVP_ACT_GELU) ||
      (d->aux_mode && d->ld_aux < d->Cout) || d->res_rows < 0 || (d->res_rows % 256) != 0)
    return CAVP_ERR_BAD_ARG;
  if (fused && (tile_stats || d->splitk > 1)) return CAVP_ERR_UNSUPPORTED;
  if (tile_stats && (scale || shift || nbias || residual || d->act != CAVP_ACT_NONE)) return CAVP_ERR_BAD_ARG;
  // (the parity-ordered tiles of a stride-2 data gradient address rows by output pixel: not with the per-image bias, the forward's
  // tile statistics or the token-path fusions, none of which a data gradient carries)
  const bool par_ok = !(tile_stats || nbias || aux || fused);
  Plan pl = make_plan(d, true, par_ok);
  if (pl.status != CAVP_OK) return pl.status;
  if (!aligned(x, 16) || !aligned(w, 16)) return CAVP_ERR_ALIGN;
  if (tile_is_big(pl.tile_id) && d->tile % 100 == 0 &&
      !(aligned(y, 16) && (!residual || aligned(residual, 16)) && (!scale || aligned(scale, 16)) && (!shift || aligned(shift, 16)) &&
        (!nbias || aligned(nbias, 16)))) {
    if (tile_stats) return CAVP_ERR_ALIGN;   // the statistics layout was sized for the 256x256 tile
    pl = make_plan(d, false, par_ok);        // automatically chosen big tile, but an operand is not 16-byte aligned
    if (pl.status != CAVP_OK) return pl.status;
  }
  if (residual && d->ldr < d->Cout) return CAVP_ERR_BAD_ARG;
  if (allow_split && d->up <= 1) {
    const int n1 = tail_split_images(d, pl, tile_stats != nullptr);
    if (n1 > 0) {
      const size_t es_ = d->dtype == CAVP_F32 ? 4 : 2;
      const size_t xin = (size_t)n1 * d->H * d->W * d->ldx * es_;
      const size_t opix = (size_t)n1 * pl.p.Ho * pl.p.Wo;
      cavp_conv_desc da = *d, db = *d;
      da.N = n1;
      db.N = d->N - n1;
      // per-tile BatchNorm statistics: the big tile writes one pair per 128-row slab; the tail launch continues that tile
      // sequence (its first pixel is a multiple of 128, see tail_split_images) on a tile with 128 pixel rows
      if (tile_stats) { db.tile = 2; db.splitk = 1; }
      int st = conv2d_launch(&da, x, w, scale, shift, nbias, residual, y, aux, workspace, workspace_bytes, tile_stats, stream, false);
      if (st != CAVP_OK) return st;
      return conv2d_launch(&db, (const char*)x + xin, w, scale, shift, nbias ? nbias + (size_t)n1 * d->Cout : nullptr,
                           residual ? (const char*)residual + opix * d->ldr * es_ : nullptr, (char*)y + opix * d->ldy * es_,
                           aux ? (char*)aux + opix * d->ld_aux * es_ : nullptr, workspace, workspace_bytes,
                           tile_stats ? tile_stats + (opix / 128) * (size_t)d->Cout * 2 : nullptr, stream, false);
    }
  }
  if (pl.ws_bytes > 0 && (!workspace || workspace_bytes < pl.ws_bytes || !aligned(workspace, 16)))
    return CAVP_ERR_WORKSPACE;
  IgemmParams& p = pl.p;
  p.x = x; p.w = w; p.y = y; p.scale = scale; p.shift = shift; p.nbias = nbias; p.res = residual;
  p.partial = (float*)workspace;
  const size_t es = d->dtype == CAVP_F32 ? 4 : 2;
  {
    const size_t xb = ((size_t)d->N * d->H * d->W - 1) * d->ldx * es + (size_t)d->Cin * es;
    const size_t wb = (size_t)d->Cout * p.K * es;
    if (xb >= 0x7fffffffull || wb >= 0x7fffffffull) return CAVP_ERR_UNSUPPORTED;  // 32-bit buffer offsets
    p.x_bytes = (int)xb;
    p.w_bytes = (int)wb;
  }
  p.vec_io = (d->Cout % 4 == 0) && (d->ldy % 4 == 0) && aligned(y, 4 * es) &&
             (!residual || (d->ldr % 4 == 0 && aligned(residual, 4 * es))) && (!scale || aligned(scale, 16)) &&
             (!shift || aligned(shift, 16)) && (!nbias || aligned(nbias, 16));
  const int VE = d->dtype == CAVP_F32 ? 4 : 8;
  p.coalesced = !pl.direct_epi && p.splitk == 1 && (d->Cout % VE == 0) && (d->ldy % VE == 0) && aligned(y, 16) &&
                (!residual || (d->ldr % VE == 0 && aligned(residual, 16))) && (!scale || aligned(scale, 16)) &&
                (!shift || aligned(shift, 16));
  p.tile_stats = tile_stats;
  if (bnb) {
    if (!p.coalesced || !tile_has_bnb(pl.tile_id)) return CAVP_ERR_UNSUPPORTED;   // see cavp_conv2d_bnbwd_layout
    p.bnb_z = bnb->z; p.bnb_out = bnb->out; p.ld_bnb_z = bnb->ld_z; p.ld_bnb_out = bnb->ld_out;
    p.bnb_scale = bnb->fwd_scale; p.bnb_shift = bnb->fwd_shift; p.bnb_mean = bnb->mean; p.bnb_rstd = bnb->rstd;
    p.bnb_act = bnb->act; p.bnb_part = bnb->partials;
  }
  p.aux = aux; p.aux_mode = d->aux_mode; p.ld_aux = d->ld_aux; p.res_rows = residual ? d->res_rows : 0;
  if (fused && !(p.coalesced && (!aux || (d->ld_aux % VE == 0 && aligned(aux, 16))))) return CAVP_ERR_UNSUPPORTED;
  if (p.res_rows > 0 && (long long)p.res_rows > p.M) return CAVP_ERR_BAD_ARG;
  if (tile_stats && !p.coalesced) return CAVP_ERR_UNSUPPORTED;   // see cavp_conv2d_tile_stats_layout
  if (tile_is_big(pl.tile_id) && !p.coalesced) return CAVP_ERR_ALIGN;   // the 256x256 tile only has the 16-byte epilogue
  hipStream_t s = (hipStream_t)stream;
  const bool upm = p.up_mask != 0;
  hipError_t e = d->dtype == CAVP_F32
                     ? (upm ? launch_tile<float, true>(pl.tile_id, p, pl.nblk, s) : launch_tile<float, false>(pl.tile_id, p, pl.nblk, s))
                     : (upm ? launch_tile<bf16_t, true>(pl.tile_id, p, pl.nblk, s) : launch_tile<bf16_t, false>(pl.tile_id, p, pl.nblk, s));
  if (e != hipSuccess) return CAVP_ERR_LAUNCH;
  if (p.splitk > 1) {
    const long long total = (long long)p.M * ((p.Cout + 3) / 4);
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (d->dtype == CAVP_F32)
      splitk_epilogue_kernel<float><<<dim3(nb), dim3(256), 0, s>>>(p);
    else
      splitk_epilogue_kernel<bf16_t><<<dim3(nb), dim3(256), 0, s>>>(p);
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  return CAVP_OK;
}
