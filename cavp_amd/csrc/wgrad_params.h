// Kernel-argument blocks shared by the weight-gradient kernels (conv_wgrad.hip: the 4-wave 128 x 128 tile, both dtypes;
// conv_wgrad_big.hip: the 8-wave 256 x 256 bf16 tile with a four-stage LDS-DMA ring).
#pragma once
#include "common.h"

struct WgradParams {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, Cin, ldx, Cout, ldy, KW, stride, pad, dil;
  int Ho, Wo, M;
  int ntaps, ntaps_all;
  unsigned long long taps;
  int tiles_co, tiles_ci, ksplit;
  int rows_per_split;  // multiple of 64
  unsigned dv_co[2], dv_ci[2], dv_nt[2], dv_kw[2], dv_hw[2], dv_w[2], dv_cq[2];  // fast_div (multiplier, shift) of tiles_co, tiles_ci, ntaps, KW, Ho*Wo, Wo
  int x_bytes, dy_bytes;
  int overwrite;       // 1: dw = gradient (beta = 0, dw is not read); 0: dw += gradient
  int oihw;            // dw layout: 0 = [Cout][taps][Cin] (OHWI), 1 = [Cout][Cin][taps] (torch .grad layout)
  int dbg;             // -DCAVP_PROFILE builds only (CAVP_WGRAD_DBG): 1 = loads out of range, 2 = no MFMAs, 4 = no DMA, 8 = no epilogue
  float* dbias;        // optional: dbias[co] += sum_pix dY[pix][co] (bias gradient), taken from the dY tiles streamed anyway
  float* bias_slabs;   // ksplit > 1: [ksplit][Cout] partial column sums
  float* slabs;        // ksplit > 1: per-split partial gradients [ksplit][Cout][taps][Cin] (plain stores, then reduced)
  int red_zg;          // grouped launches: split groups per workgroup of this job's slab reduce (1, 2, 4, 8, 16)
  int pad_;
};

// Kernel-argument block of a grouped launch (cavp_conv2d_wgrad_group): up to CAVP_WGRAD_GROUP_MAX independent weight gradients
// walked by ONE grid.  Logical workgroup b belongs to job j with blk_end[j-1] <= b < blk_end[j]; the slab reduces of the jobs
// that split their pixel range form a second grouped launch (red_end).  The whole block travels as kernel arguments (< 4 KiB),
// so a grouped launch is hipGraph-capturable like any other and needs no device-side table.
struct WgradGroupArgs {
  int njobs;
  int blk_end[CAVP_WGRAD_GROUP_MAX];
  int red_end[CAVP_WGRAD_GROUP_MAX];
  WgradParams job[CAVP_WGRAD_GROUP_MAX];
};
static_assert(sizeof(WgradGroupArgs) <= 4096, "kernel-argument segment");

typedef short s16x4_t __attribute__((ext_vector_type(4)));

// conv_wgrad_big.hip: the grouped launch of the 256 x 256 tile (one 512-thread workgroup per CU, 128 KiB of LDS).  The job fields
// that depend on the tile (tiles_co / tiles_ci and their divisors, rows_per_split: a multiple of 128) are planned for that tile.
// schedule: 2 = sixteen waves (64 x 64 wave tiles, four waves per SIMD), 1 = eight waves with every LDS read and DMA issue in the
// gaps between the MFMAs, 0 = eight waves: read, multiply, retire, fetch.
hipError_t cavp_launch_wgrad_big_group(const WgradGroupArgs& g, int blocks, bool bias, int schedule, hipStream_t s);
