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
};

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

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15): xor 1, xor 2 (quad_perm), then half-row and row mirrors, which
// swap quads / half-rows and so act as xor 4 / xor 8 once the lower levels are uniform.  Every lane ends with the total.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}



// value of lane 0 of the lane's 16-lane DPP row in every lane of the row (row_newbcast:0, gfx90a+): one VALU move where __shfl(v,
// lane & 48) is an LDS-crossbar ds_bpermute + an lgkmcnt wait
__device__ __forceinline__ float row16_first(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150, 0xf, 0xf, true));
}

// v + (v of lane ^ M) for M = 8 / 16 / 32 without the LDS crossbar: xor 8 stays inside a 16-lane DPP row (row_ror:8); xor 16 / xor 32 pair
// rows / halves, which v_permlane16_swap / v_permlane32_swap of a register WITH ITSELF produce (swap(a, a) = (rows [0 0 2 2], rows
// [1 1 3 3]) resp. (halves [lo lo], [hi hi]): their sum is v + xor).  __shfl_xor is a ds_bpermute_b32 + address arithmetic + an
// lgkmcnt wait per value; the BatchNorm-backward tile sums do 16 of them per step of the tree.
template <int M>
__device__ __forceinline__ float xor_add(float v) {
  const unsigned u = __float_as_uint(v);
  if constexpr (M == 8) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, true));   // row_ror:8
  } else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    static_assert(M == 32, "xor_add: 8, 16 or 32");
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}

// conv_igemm_big.hip
hipError_t cavp_launch_igemm_big(const IgemmParams& p, int nblk, hipStream_t s);
