// Kernel-argument block and MFMA wrappers shared by the implicit-GEMM kernels (conv_igemm.hip: the 2-stage 4-wave tiles,
// conv_igemm_big.hip: the 256x256 tile).
#pragma once
#include "common.h"

struct IgemmParams {
  const void* x;
  const void* w;
  void* y;
  const float* scale;
  const float* shift;
  const float* nbias;
  const void* res;
  float* partial;
  int N, H, W, Cin, ldx, Cout, ldy, KW, stride, stride_w, pad, dil, ldr, act;
  int Ho, Wo, M, K;
  unsigned div_hw_m, div_hw_s, div_w_m, div_w_s, div_tc_m, div_tc_s, div_cq_m, div_cq_s;  // floor(n / (Ho*Wo)) and floor(n / Wo) as multiply + shift (fast_div)
  int ntaps;
  unsigned long long taps;  // 4 bits per live tap id (kh*KW + kw)
  int cpt;                  // K tiles per tap
  int iters;                // ntaps * cpt
  int splitk;
  int tiles_c, tiles_p;
  int vec_io;               // epilogue may use vector loads/stores (Cout, ldy, ldr multiples of 4, pointers aligned)
  int x_bytes, w_bytes;     // buffer-descriptor extents (< 2 GiB)
  int up_shift, up_mask;    // transposed-conv input upsampling (log2, mask); 0, 0 for an ordinary conv
  int tap_dh[9], tap_dw[9];  // per LIVE tap: kh*dil, kw*dil (input-space displacement)
  int tap_xoff[9];          // per live tap: byte displacement (dh*W + dw)*ldx*sizeof(T) in x (ordinary conv only)
  int tap_woff[9];          // per live tap: byte offset tap*Cin*sizeof(T) inside a weight row
  float* tile_stats;        // optional [tiles_p][Cout][2] per-tile (mean, M2) of the raw outputs (BatchNorm statistics)
  int nblk;                 // logical workgroups (tiles x split-K); the launch may use fewer, persistent, workgroups
  int dbg;                  // -DCAVP_PROFILE builds only (tile knob digits): pieces of the kernel switched off for the K-loop anatomy
  int coalesced;            // LDS-staged, fully coalesced 16-byte epilogue (needs Cout, ldy, ldr % VE == 0, 16-B aligned)
  // token-path epilogue fusions (16-byte epilogue only): see cavp_conv_desc.res_rows / aux_mode
  void* aux;                // aux_mode 1: gelu'(t) is stored here; 2: the result is multiplied by it
  int aux_mode, ld_aux;
  int res_rows;             // 0, or: output pixel p adds residual row p % res_rows (a multiple of 256)
  // BatchNorm-backward statistics fused into a data-gradient launch (cavp_conv2d_nhwc_bnbwd; staged 4-wave epilogue only): the launch
  // produces the gradient of a BatchNorm + activation OUTPUT; the epilogue multiplies it by the activation's derivative, stores
  // g = dy * act'(.) and emits per pixel tile the two sums the BatchNorm backward needs (sum g, sum g * zhat)
  const void* bnb_z;        // the BatchNorm's input z (y's shape, pixel stride ld_bnb_z); nullptr = plain launch
  const void* bnb_out;      // the activation's output (mask source), or nullptr: the mask is re-derived from z * bnb_scale + bnb_shift
  const float* bnb_scale;   // the forward's folded scale / shift (mask without bnb_out)
  const float* bnb_shift;
  const float* bnb_mean;    // batch mean / 1 / sqrt(var + eps) of z: zhat = (z - mean) * rstd
  const float* bnb_rstd;
  float* bnb_part;          // f32 [tiles_p][Cout][2]: per pixel tile (sum g, sum g * zhat)
  int ld_bnb_z, ld_bnb_out, bnb_act, pad2_;
  // Parity-ordered pixel tiles of a stride-2 data gradient (up = 2; conv_igemm.hip): the launch's logical pixel m runs over the four
  // output parity classes q = (ho & 1) * 2 + (wo & 1) one after the other, par_mq (a multiple of every tile height) logical pixels per
  // class, m = q * par_mq + (n * par_hq + i) * par_wq + j  <->  output pixel (n, 2 i + (q >> 1), 2 j + (q & 1)).  A tile then lies inside
  // ONE class and walks only the taps that can hit the zero-upsampled input for that parity (par_nt[q] of them, launch tap indices
  // 4 bits each in par_taps[q]; 3x3 stride 2: 1 / 2 / 2 / 4 instead of 9 for every pixel; 1x1 stride 2: 1 / 0 / 0 / 0).
  int up_par;               // 0: linear pixel order (M = N Ho Wo)
  int par_mq, par_hq, par_wq, par_m;   // par_m: N * Ho * Wo (p.M is 4 * par_mq in this mode)
  unsigned div_mq_m, div_mq_s, div_hwq_m, div_hwq_s, div_wq_m, div_wq_s;
  int par_nt[4];
  unsigned long long par_taps[4];
};

// parity mode: logical pixel m -> linear output pixel (n * Ho + ho) * Wo + wo, or -1 for a padding slot; (n, ho, wo) returned too
__device__ __forceinline__ int par_out_pixel(const IgemmParams& p, int m, int& n, int& ho, int& wo) {
  const int q = fast_div(m, p.div_mq_m, p.div_mq_s), idx = m - q * p.par_mq;
  n = fast_div(idx, p.div_hwq_m, p.div_hwq_s);
  const int r = idx - n * (p.par_hq * p.par_wq);
  const int i = fast_div(r, p.div_wq_m, p.div_wq_s), j = r - i * p.par_wq;
  ho = 2 * i + (q >> 1);
  wo = 2 * j + (q & 1);
  const bool ok = n < p.N && ho < p.Ho && wo < p.Wo;
  return ok ? (n * p.Ho + ho) * p.Wo + wo : -1;
}

template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[0]), __uint_as_float(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[1]), __uint_as_float(b[1]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[2]), __uint_as_float(b[2]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[3]), __uint_as_float(b[3]), acc, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                  0, 0, 0);
  }
};

// conv_igemm_big.hip
hipError_t cavp_launch_igemm_big(const IgemmParams& p, int nblk, hipStream_t s);
