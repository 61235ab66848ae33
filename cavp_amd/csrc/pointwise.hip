// Bandwidth-bound kernels of the CAVP forward path for gfx950: small-Cin direct conv, pooling, bilinear resize,
// LayerNorm, the sigmoid attention gate, BN folding and weight packing.  All NHWC, 16-byte vector accesses along
// the channel dimension, wave64 shuffles for the row reductions.
#include "common.h"

namespace {

inline int nblocks(long long total, int per_block, int cap = 1 << 20) {
  long long nb = (total + per_block - 1) / per_block;
  if (nb < 1) nb = 1;
  if (nb > cap) nb = cap;
  return (int)nb;
}

// load / store VE consecutive channels as floats
template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int VE = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<bf16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    const uint4 t = *(const uint4*)p;
    const unsigned u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(u[i] << 16);
      v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    uint4 t;
    t.x = pack2bf(v[0], v[1]);
    t.y = pack2bf(v[2], v[3]);
    t.z = pack2bf(v[4], v[5]);
    t.w = pack2bf(v[6], v[7]);
    *(uint4*)p = t;
  }
};

// ------------------------------------------------------------------------------------------------------------
// direct 3x3 conv, Cin <= 3, pad 1, NCHW f32 in -> NHWC out.
// One workgroup = a run of TW = 2 * (256 / G) output pixels of one output row (G = Cout / 16 channel groups); thread =
// (two pixels PP apart, 16 output channels).  The 3-row input patch of the run is staged in LDS with coalesced row loads
// (the first version gathered its 27 inputs per thread straight from global memory: 27 wave loads of 16 distinct
// addresses each) and every weight vector read from LDS feeds two pixels.  106 -> ~45 us for the 224x224 stem at B = 32.
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ y, int N,
                                                               int Cin, int H, int W, int Cout, int stride, int Ho,
                                                               int Wo, int act, int tiles_w) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int K = Cin * 9;
  const int G = Cout >> 4, PP = 256 / G, TW = 2 * PP;
  const int PW = (TW - 1) * stride + 3;          // patch width (input columns under the run, incl. the 1-pixel halo)
  float* ws = (float*)smem_raw;                  // [K][Cout]
  float* patch = ws + K * Cout;                  // [Cin][3][PW]
  for (int i = threadIdx.x; i < K * Cout; i += 256) {
    const int co = i / K, k = i - co * K;  // w is [Cout][Cin][3][3] -> k = ci*9 + kh*3 + kw
    ws[k * Cout + co] = w[i];
  }
  const int tw = blockIdx.x % tiles_w;
  const int row = blockIdx.x / tiles_w;          // = n * Ho + ho
  const int ho = row % Ho, n = row / Ho;
  const int wo0 = tw * TW;
  const int wi0 = wo0 * stride - 1, hi0 = ho * stride - 1;
  for (int r = threadIdx.x >> 6; r < Cin * 3; r += 4) {   // one wave per patch row, lanes along the row
    const int ci = r / 3, kh = r - ci * 3;
    const int hi = hi0 + kh;
    const float* xr = x + (((size_t)n * Cin + ci) * H + (hi >= 0 && hi < H ? hi : 0)) * W;
    for (int c = threadIdx.x & 63; c < PW; c += 64) {
      const int wi = wi0 + c;
      patch[r * PW + c] = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? xr[wi] : 0.f;
    }
  }
  __syncthreads();
  const int g = threadIdx.x % G, pp = threadIdx.x / G;
  if (pp >= PP) return;                          // (G not a power of two: a few idle threads)
  float acc0[16], acc1[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
  const int c0 = pp * stride, c1 = (pp + PP) * stride;
  for (int r = 0; r < Cin * 3; ++r) {
    const float* pr = patch + r * PW;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float x0 = pr[c0 + kw], x1 = pr[c1 + kw];
      const float* wk = ws + (r * 3 + kw) * Cout + g * 16;   // (r * 3 + kw = ci*9 + kh*3 + kw)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float wv = wk[j];
        acc0[j] = fmaf(x0, wv, acc0[j]);
        acc1[j] = fmaf(x1, wv, acc1[j]);
      }
    }
  }
  float sc[16], sh[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sc[j] = scale ? scale[g * 16 + j] : 1.f;
    sh[j] = shift ? shift[g * 16 + j] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int wo = wo0 + pp + u * PP;
    if (wo >= Wo) continue;
    float* acc = u ? acc1 : acc0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v = acc[j];
      if (scale) v *= sc[j];
      if (shift) v += sh[j];
      acc[j] = apply_act(v, act);
    }
    T* yp = y + ((size_t)row * Wo + wo) * Cout + g * 16;
#pragma unroll
    for (int j = 0; j < 16; j += Vec<T>::VE) Vec<T>::store(yp + j, acc + j);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same conv on the matrix cores, for bf16 outputs with Cout = 64 (the image stem 3 -> 64 stride 2 and the audio stem 1 -> 64):
// K = 9 Cin <= 27 is ONE v_mfma_f32_16x16x32_bf16 K step.  The VALU version above spends 864 FMAs + 108 ds_read_b128 per thread on a
// layer whose HBM floor is 14 us (19 MB in, 51 MB out at B = 32) and runs 69 .. 71 us; here a thread does 8 ds_read_b32 per 16 pixels.
// * f32 inputs and weights enter as bf16 hi + lo pairs (x = hi + lo to 2^-17): acc = Whi Xhi + Wlo Xhi + Whi Xlo with f32 accumulation, i.e.
//   the f32 products of the VALU version to ~1e-5 relative - the stem keeps reading the f32 image, not a bf16 rounding of it.
// * weights = the A operand with the rows permuted (row 4 g + i of block a = channel 16 g + 4 a + i), so a lane's four accumulators
//   of the four channel blocks are 16 CONSECUTIVE channels of one pixel: two 16-byte stores per lane and pixel.
// * persistent workgroups walk (image row, 128-pixel run) items: the weight fragments (32 VGPRs) are built once per workgroup.
// ------------------------------------------------------------------------------------------------------------
template <int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_smallcin_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    bf16_t* __restrict__ y, int N, int Cin, int H, int W, int Ho, int Wo,
                                                                    int act, int tiles_w, int total) {
  constexpr int TW = 128, PW = (TW - 1) * STRIDE + 3, COUT = 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* patch = (float*)smem_raw;   // [Cin * 3][PW]
  const int K = Cin * 9;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4, kbase = g * 8;
  // A fragments: block a, row (lane & 15) = channel 16 (row >> 2) + 4 a + (row & 3), k = kbase .. kbase + 7 (zero beyond K)
  u32x4_t ahi[4], alo[4];
  int poff[8];   // patch offset of k = kbase + j for pixel 0: row (k / 3) = ci * 3 + kh, column kw = k % 3
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kbase + j;
    poff[j] = k < K ? (k / 3) * PW + (k % 3) : 0;   // (k >= K: any finite value, its weight is zero)
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int co = 16 * (col >> 2) + 4 * a + (col & 3);
    float wv[8], wl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = (kbase + j) < K ? w[co * K + kbase + j] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ahi[a][q] = pack2bf(wv[2 * q], wv[2 * q + 1]);
      wl[2 * q] = wv[2 * q] - __uint_as_float(ahi[a][q] << 16);
      wl[2 * q + 1] = wv[2 * q + 1] - __uint_as_float(ahi[a][q] & 0xffff0000u);
      alo[a][q] = pack2bf(wl[2 * q], wl[2 * q + 1]);
    }
  }
  // The patch of the NEXT item travels in registers while the current item is multiplied: a wave owns patch rows wave, wave + 4, wave + 8
  // (<= 3 of the Cin * 3 rows) and NC = ceil(PW / 64) columns per lane; the loads are issued behind the second barrier and written to
  // LDS at the top of the next trip.  (As a load -> wait -> ds_write loop inside the item - the VALU version's form - the five
  // dependent round trips per row were the launch: 44 us.)
  constexpr int NC = (PW + 63) / 64;
  float pv[3][NC];
  auto fetch = [&](int it) {
    const int tw_ = it % tiles_w, row_ = it / tiles_w;
    const int ho_ = row_ % Ho, n_ = row_ / Ho;
    const int wi0_ = tw_ * TW * STRIDE - 1, hi0_ = ho_ * STRIDE - 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int r = wave + 4 * q;
      const int ci = r / 3, kh = r - ci * 3;
      const int hi = hi0_ + kh;
      const bool rok = r < Cin * 3 && (unsigned)hi < (unsigned)H;
      const float* xr = x + (((size_t)n_ * Cin + (rok ? ci : 0)) * H + (rok ? hi : 0)) * W;
#pragma unroll
      for (int u = 0; u < NC; ++u) {
        const int wi = wi0_ + lane + 64 * u;
        pv[q][u] = (rok && (unsigned)wi < (unsigned)W) ? xr[wi] : 0.f;
      }
    }
  };
  if ((int)blockIdx.x < total) fetch((int)blockIdx.x);
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const int tw = item % tiles_w;
    const int row = item / tiles_w;          // = n * Ho + ho
    const int wo0 = tw * TW;
    __syncthreads();                         // the previous item's fragment reads are done
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int r = wave + 4 * q;
      if (r < Cin * 3) {
#pragma unroll
        for (int u = 0; u < NC; ++u)
          if (lane + 64 * u < PW) patch[r * PW + lane + 64 * u] = pv[q][u];
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < total) fetch(item + (int)gridDim.x);
    const int nblk = (min(TW, Wo - wo0) + 15) >> 4;
    for (int blk = wave; blk < nblk; blk += 4) {
      const int px = blk * 16 + col;
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = patch[poff[j] + px * STRIDE];
      u32x4_t bhi, blo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bhi[q] = pack2bf(xv[2 * q], xv[2 * q + 1]);
        blo[q] = pack2bf(xv[2 * q] - __uint_as_float(bhi[q] << 16), xv[2 * q + 1] - __uint_as_float(bhi[q] & 0xffff0000u));
      }
      float o[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, alo[a]), __builtin_bit_cast(bf16x8_t, bhi), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ahi[a]), __builtin_bit_cast(bf16x8_t, blo), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ahi[a]), __builtin_bit_cast(bf16x8_t, bhi), acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * a + i] = acc[i];
      }
      const int wo = wo0 + px;
      if (wo < Wo) {
        // (scale / shift re-read per pixel block, 16-byte L1 hits: as 32 resident registers they put the kernel at 130 VGPRs = three
        // waves per SIMD instead of four)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 s4 = scale ? *(const float4*)(scale + g * 16 + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
          const float4 h4 = shift ? *(const float4*)(shift + g * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
          o[4 * q] = apply_act(o[4 * q] * s4.x + h4.x, act);
          o[4 * q + 1] = apply_act(o[4 * q + 1] * s4.y + h4.y, act);
          o[4 * q + 2] = apply_act(o[4 * q + 2] * s4.z + h4.z, act);
          o[4 * q + 3] = apply_act(o[4 * q + 3] * s4.w + h4.w, act);
        }
        bf16_t* yp = y + ((size_t)row * Wo + wo) * COUT + g * 16;
        Vec<bf16_t>::store(yp, o);
        Vec<bf16_t>::store(yp + 8, o + 8);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// max pool NHWC.  thread = (output pixel, 16-byte channel vector)
// ------------------------------------------------------------------------------------------------------------
// AFF: the pooled tensor is act(x * scale + shift) rounded to T - the BatchNorm apply + ReLU of the training stem folded into its max pool
// (cavp_maxpool_affine_nhwc): same expression, contraction and rounding as scale_shift_act, so values and arg-max are those of the
// two-launch route, without the [32][112][112][128] activation written and read back (206 MB per step).
template <typename T, bool AFF = false>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                      unsigned char* __restrict__ argmax, int N, int H, int W,
                                                      int C, int k, int stride, int pad, int Ho, int Wo,
                                                      const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr,
                                                      int act = 0) {
  constexpr int VE = Vec<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)N * Ho * Wo * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int cv, wo, ho, n;
    long long pix;
    split_index(idx, CV, Wo, Ho, cv, pix, wo, ho, n);
    float sc[AFF ? VE : 1], sh[AFF ? VE : 1];
    if constexpr (AFF) {
#pragma unroll
      for (int j = 0; j < VE; ++j) { sc[j] = scale[cv * VE + j]; sh[j] = shift[cv * VE + j]; }
    }
    float m[VE];
    int am[VE];   // window-relative position kh * k + kw of the FIRST maximum (strict >, scan order: as ATen)
    bool first = true;
#pragma unroll
    for (int j = 0; j < VE; ++j) { m[j] = -INFINITY; am[j] = 0; }
    for (int kh = 0; kh < k; ++kh) {
      const int hi = ho * stride - pad + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int wi = wo * stride - pad + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        float v[VE];
        Vec<T>::load(x + ((size_t)(n * H + hi) * W + wi) * C + cv * VE, v);
        if constexpr (AFF) {
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            const float t = apply_act(v[j] * sc[j] + sh[j], act);
            if constexpr (sizeof(T) == 2) v[j] = __uint_as_float(pack2bf(t, 0.f) << 16);   // the value scale_shift_act would have stored
            else v[j] = t;
          }
        }
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          const bool take = first || v[j] > m[j];
          am[j] = take ? kh * k + kw : am[j];
          m[j] = take ? v[j] : m[j];
        }
        first = false;
      }
    }
    Vec<T>::store(y + (size_t)pix * C + cv * VE, m);
    if (argmax) {
      unsigned char* ap = argmax + (size_t)pix * C + cv * VE;
      if constexpr (VE == 8) {
        uint2 o;
        o.x = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
        o.y = (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24);
        *(uint2*)ap = o;
      } else {
        *(unsigned*)ap = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// global average pool: block = (image n, 64-channel chunk); 4 waves split the pixels, lane = channel
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gap_kernel(const T* __restrict__ x, float* __restrict__ y, int HW, int C,
                                                  int ldx) {
  __shared__ float part[4][64];
  const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C) {
    const T* xp = x + (size_t)n * HW * ldx + c;
    for (int i = wv; i < HW; i += 4) s += Elem<T>::ld(xp + (size_t)i * ldx);
  }
  part[wv][threadIdx.x & 63] = s;
  __syncthreads();
  if (wv == 0 && c < C) {
    const int l = threadIdx.x;
    y[(size_t)n * C + c] = (part[0][l] + part[1][l] + part[2][l] + part[3][l]) / (float)HW;
  }
}

// ------------------------------------------------------------------------------------------------------------
// bilinear resize (PyTorch area_pixel_compute_source_index semantics, f32 index math)
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int dst, int in, int out, int align, int& i0, int& i1, float& lam) {
  float src;
  if (align) {
    const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = sc * (float)dst;
  } else {
    const float sc = (float)in / (float)out;
    src = sc * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  lam = src - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hi,
                                                            int Wi, int C, int ldx, int Ho, int Wo, int ldy, int align) {
  constexpr int VE = Vec<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)N * Ho * Wo * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int cv, wo, ho, n;
    long long pix;
    split_index(idx, CV, Wo, Ho, cv, pix, wo, ho, n);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(ho, Hi, Ho, align, h0, h1, lh);
    src_index(wo, Wi, Wo, align, w0, w1, lw);
    const T* base = x + (size_t)n * Hi * Wi * ldx + cv * VE;
    float a[VE], b[VE], c[VE], d[VE], o[VE];
    Vec<T>::load(base + ((size_t)h0 * Wi + w0) * ldx, a);
    Vec<T>::load(base + ((size_t)h0 * Wi + w1) * ldx, b);
    Vec<T>::load(base + ((size_t)h1 * Wi + w0) * ldx, c);
    Vec<T>::load(base + ((size_t)h1 * Wi + w1) * ldx, d);
    const float w00 = (1.f - lh) * (1.f - lw), w01 = (1.f - lh) * lw, w10 = lh * (1.f - lw), w11 = lh * lw;
#pragma unroll
    for (int j = 0; j < VE; ++j) o[j] = w00 * a[j] + w01 * b[j] + w10 * c[j] + w11 * d[j];
    Vec<T>::store(y + (size_t)pix * ldy + cv * VE, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N,
                                                               int Hi, int Wi, int C, int ldx, int Ho, int Wo,
                                                               int align) {
  const long long total = (long long)N * Ho * Wo;
  for (long long pix = blockIdx.x * 256ll + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    int wo, ho, n;
    split_pixel(pix, Wo, Ho, wo, ho, n);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(ho, Hi, Ho, align, h0, h1, lh);
    src_index(wo, Wi, Wo, align, w0, w1, lw);
    const T* base = x + (size_t)n * Hi * Wi * ldx;
    const T* pa = base + ((size_t)h0 * Wi + w0) * ldx;
    const T* pb = base + ((size_t)h0 * Wi + w1) * ldx;
    const T* pc = base + ((size_t)h1 * Wi + w0) * ldx;
    const T* pd = base + ((size_t)h1 * Wi + w1) * ldx;
    const float w00 = (1.f - lh) * (1.f - lw), w01 = (1.f - lh) * lw, w10 = lh * (1.f - lw), w11 = lh * lw;
    float* yp = y + (size_t)n * C * Ho * Wo + (size_t)ho * Wo + wo;
    for (int c = 0; c < C; ++c) {
      const float v = w00 * Elem<T>::ld(pa + c) + w01 * Elem<T>::ld(pb + c) + w10 * Elem<T>::ld(pc + c) +
                      w11 * Elem<T>::ld(pd + c);
      yp[(size_t)c * Ho * Wo] = v;
    }
  }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                               float* scale, float* shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
  }
}

template <typename T>
__global__ void pack_ohwi_kernel(const float* __restrict__ w, T* __restrict__ o, int Cout, int Cin, int KHW) {
  const long long total = (long long)Cout * Cin * KHW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    // destination index i = (co*KHW + t)*Cin + ci ; source = (co*Cin + ci)*KHW + t
    const int ci = (int)(i % Cin);
    const long long r = i / Cin;
    const int t = (int)(r % KHW);
    const long long co = r / KHW;
    Elem<T>::st(o + i, w[(co * Cin + ci) * KHW + t]);
  }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    Elem<D>::st(d + i, Elem<S>::ld(s + i));
}

// 8 elements per thread (16 bytes of bf16 / 2 x 16 bytes of f32) when the tensors allow it
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_vec8_kernel(const S* __restrict__ s, D* __restrict__ d, long long n8) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    if constexpr (sizeof(S) == 2) {
      VecT<bf16_t>::load((const bf16_t*)s + i * 8, v);
    } else {
      VecT<float>::load((const float*)s + i * 8, v);
      VecT<float>::load((const float*)s + i * 8 + 4, v + 4);
    }
    if constexpr (sizeof(D) == 2) {
      VecT<bf16_t>::store((bf16_t*)d + i * 8, v);
    } else {
      VecT<float>::store((float*)d + i * 8, v);
      VecT<float>::store((float*)d + i * 8 + 4, v + 4);
    }
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline bool dtype_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH

}  // namespace

extern "C" int cavp_abi_version(void) { return CAVP_ABI_VERSION; }

CavpDetState g_cavp_det = {nullptr, 0};
extern "C" int cavp_set_deterministic(void* scratch, size_t bytes) {
  if (scratch && (((uintptr_t)scratch & 15) || bytes < (1u << 20))) return CAVP_ERR_BAD_ARG;
  g_cavp_det.scratch = (float*)scratch;
  g_cavp_det.floats = scratch ? bytes / sizeof(float) : 0;
  return CAVP_OK;
}
extern "C" int cavp_get_deterministic(void) { return g_cavp_det.scratch != nullptr; }

extern "C" const char* cavp_error_string(int s) {
  switch (s) {
    case CAVP_OK: return "ok";
    case CAVP_ERR_BAD_ARG: return "bad argument";
    case CAVP_ERR_UNSUPPORTED: return "unsupported shape/dtype";
    case CAVP_ERR_ALIGN: return "misaligned pointer or leading dimension";
    case CAVP_ERR_WORKSPACE: return "workspace missing or too small";
    case CAVP_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

extern "C" int cavp_conv3x3_smallcin_nchw(int32_t dtype, const float* x, const float* w, const float* scale,
                                          const float* shift, void* y, int32_t N, int32_t Cin, int32_t H, int32_t W,
                                          int32_t Cout, int32_t stride, int32_t act, void* stream) {
  if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || stride <= 0) return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(dtype) || Cin < 1 || Cin > 3 || Cout % 16 || Cout > 128) return CAVP_ERR_UNSUPPORTED;
  if (!al16(y)) return CAVP_ERR_ALIGN;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int G = Cout / 16, TW = 2 * (256 / G);
  const int tiles_w = (Wo + TW - 1) / TW;
  const long long nb = (long long)N * Ho * tiles_w;
  if (nb > 0x7fffffffll) return CAVP_ERR_UNSUPPORTED;
  const size_t lds = ((size_t)Cin * 9 * Cout + (size_t)Cin * 3 * ((TW - 1) * stride + 3)) * sizeof(float);
  if (lds > 64 * 1024) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_BF16 && Cout == 64 && (stride == 1 || stride == 2)) {   // both stems of the model: the matrix-core version
    const int tw = (Wo + 127) / 128;
    const long long items = (long long)N * Ho * tw;
    if (items <= 0x7fffffffll) {
      const int grid = items > 1024 ? 1024 : (int)items;   // 4 resident workgroups per CU (109 .. 116 VGPRs)
      const size_t l2 = (size_t)Cin * 3 * (127 * stride + 3) * sizeof(float);
      if (stride == 2)
        conv3x3_smallcin_mfma_kernel<2><<<grid, 256, l2, s>>>(x, w, scale, shift, (bf16_t*)y, N, Cin, H, W, Ho, Wo, act, tw, (int)items);
      else
        conv3x3_smallcin_mfma_kernel<1><<<grid, 256, l2, s>>>(x, w, scale, shift, (bf16_t*)y, N, Cin, H, W, Ho, Wo, act, tw, (int)items);
      CHECK_LAUNCH();
    }
  }
  if (dtype == CAVP_F32)
    conv3x3_smallcin_kernel<float><<<(int)nb, 256, lds, s>>>(x, w, scale, shift, (float*)y, N, Cin, H, W, Cout, stride, Ho, Wo, act, tiles_w);
  else
    conv3x3_smallcin_kernel<bf16_t><<<(int)nb, 256, lds, s>>>(x, w, scale, shift, (bf16_t*)y, N, Cin, H, W, Cout, stride, Ho, Wo, act, tiles_w);
  CHECK_LAUNCH();
}

extern "C" int cavp_maxpool_nhwc(int32_t dtype, const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W,
                                 int32_t C, int32_t k, int32_t stride, int32_t pad, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return CAVP_ERR_BAD_ARG;
  if (argmax && ((uintptr_t)argmax & 7)) return CAVP_ERR_ALIGN;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(y)) return CAVP_ERR_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CAVP_ERR_BAD_ARG;
  const long long total = (long long)N * Ho * Wo * (C / VE);
  const int nb = nblocks(total, 256, 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    maxpool_kernel<float><<<nb, 256, 0, s>>>((const float*)x, (float*)y, (unsigned char*)argmax, N, H, W, C, k, stride, pad, Ho, Wo);
  else
    maxpool_kernel<bf16_t><<<nb, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, (unsigned char*)argmax, N, H, W, C, k, stride, pad, Ho, Wo);
  CHECK_LAUNCH();
}

extern "C" int cavp_maxpool_affine_nhwc(int32_t dtype, const void* x, const float* scale, const float* shift, int32_t act, void* y,
                                        uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride,
                                        int32_t pad, void* stream) {
  if (!x || !y || !scale || !shift || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return CAVP_ERR_BAD_ARG;
  if (argmax && ((uintptr_t)argmax & 7)) return CAVP_ERR_ALIGN;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(y)) return CAVP_ERR_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return CAVP_ERR_BAD_ARG;
  const long long total = (long long)N * Ho * Wo * (C / VE);
  const int nb = nblocks(total, 256, 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    maxpool_kernel<float, true><<<nb, 256, 0, s>>>((const float*)x, (float*)y, (unsigned char*)argmax, N, H, W, C, k, stride, pad, Ho, Wo, scale, shift, act);
  else
    maxpool_kernel<bf16_t, true><<<nb, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, (unsigned char*)argmax, N, H, W, C, k, stride, pad, Ho, Wo, scale, shift, act);
  CHECK_LAUNCH();
}

extern "C" int cavp_global_avgpool_nhwc(int32_t dtype, const void* x, float* y, int32_t N, int32_t HW, int32_t C,
                                        int32_t ldx, void* stream) {
  if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((C + 63) / 64, N);
  if (dtype == CAVP_F32)
    gap_kernel<float><<<grid, 256, 0, s>>>((const float*)x, y, HW, C, ldx);
  else
    gap_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, y, HW, C, ldx);
  CHECK_LAUNCH();
}

extern "C" int cavp_bilinear_nhwc(int32_t dtype, const void* x, void* y, int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                                  int32_t ldx, int32_t Ho, int32_t Wo, int32_t ldy, int32_t align_corners,
                                  void* stream) {
  if (!x || !y || N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || ldx < C || ldy < C)
    return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE || ldy % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(y)) return CAVP_ERR_ALIGN;
  const long long total = (long long)N * Ho * Wo * (C / VE);
  const int nb = nblocks(total, 256, 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    bilinear_nhwc_kernel<float><<<nb, 256, 0, s>>>((const float*)x, (float*)y, N, Hi, Wi, C, ldx, Ho, Wo, ldy, align_corners);
  else
    bilinear_nhwc_kernel<bf16_t><<<nb, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, N, Hi, Wi, C, ldx, Ho, Wo, ldy, align_corners);
  CHECK_LAUNCH();
}

extern "C" int cavp_bilinear_nhwc_to_nchw(int32_t dtype, const void* x, float* y, int32_t N, int32_t Hi, int32_t Wi,
                                          int32_t C, int32_t ldx, int32_t Ho, int32_t Wo, int32_t align_corners,
                                          void* stream) {
  if (!x || !y || N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const long long total = (long long)N * Ho * Wo;
  const int nb = nblocks(total, 256, 16384);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    bilinear_to_nchw_kernel<float><<<nb, 256, 0, s>>>((const float*)x, y, N, Hi, Wi, C, ldx, Ho, Wo, align_corners);
  else
    bilinear_to_nchw_kernel<bf16_t><<<nb, 256, 0, s>>>((const bf16_t*)x, y, N, Hi, Wi, C, ldx, Ho, Wo, align_corners);
  CHECK_LAUNCH();
}

extern "C" int cavp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            float* scale, float* shift, int32_t C, void* stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift || C <= 0) return CAVP_ERR_BAD_ARG;
  bn_fold_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(gamma, beta, mean, var, eps, scale, shift, C);
  CHECK_LAUNCH();
}

extern "C" int cavp_pack_weight_ohwi(int32_t dtype, const float* w, void* o, int32_t Cout, int32_t Cin, int32_t KH,
                                     int32_t KW, void* stream) {
  if (!w || !o || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const long long total = (long long)Cout * Cin * KH * KW;
  const int nb = nblocks(total, 256, 8192);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    pack_ohwi_kernel<float><<<nb, 256, 0, s>>>(w, (float*)o, Cout, Cin, KH * KW);
  else
    pack_ohwi_kernel<bf16_t><<<nb, 256, 0, s>>>(w, (bf16_t*)o, Cout, Cin, KH * KW);
  CHECK_LAUNCH();
}

extern "C" int cavp_cast(int32_t sdt, const void* src, int32_t ddt, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return CAVP_ERR_BAD_ARG;
  if (!dtype_ok(sdt) || !dtype_ok(ddt)) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n % 8 == 0 && al16(src) && al16(dst)) {
    const long long n8 = n / 8;
    const int nb8 = nblocks(n8, 256, 16384);
    if (sdt == CAVP_F32 && ddt == CAVP_BF16)
      cast_vec8_kernel<float, bf16_t><<<nb8, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n8);
    else if (sdt == CAVP_BF16 && ddt == CAVP_F32)
      cast_vec8_kernel<bf16_t, float><<<nb8, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n8);
    else if (sdt == CAVP_F32)
      cast_vec8_kernel<float, float><<<nb8, 256, 0, s>>>((const float*)src, (float*)dst, n8);
    else
      cast_vec8_kernel<bf16_t, bf16_t><<<nb8, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n8);
    CHECK_LAUNCH();
  }
  const int nb = nblocks(n, 256, 8192);
  if (sdt == CAVP_F32 && ddt == CAVP_BF16)
    cast_kernel<float, bf16_t><<<nb, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n);
  else if (sdt == CAVP_BF16 && ddt == CAVP_F32)
    cast_kernel<bf16_t, float><<<nb, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n);
  else if (sdt == CAVP_F32)
    cast_kernel<float, float><<<nb, 256, 0, s>>>((const float*)src, (float*)dst, n);
  else
    cast_kernel<bf16_t, bf16_t><<<nb, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n);
  CHECK_LAUNCH();
}


// clear n [start, end) element ranges of one f32 buffer in ONE launch (the gradient arena's per-step reset: ~30 ranges)
static __global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long long* __restrict__ table) {
  const long long a = table[2 * blockIdx.y], b = table[2 * blockIdx.y + 1];
  float4* p4 = (float4*)(base + a);          // ranges are 16-byte aligned and a multiple of 4 floats long
  const long long n4 = (b - a) >> 2;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// clear a 4-byte-aligned buffer of a multiple of 4 bytes (16-byte stores over the aligned middle, 4-byte ones at the ragged ends)
static __global__ __launch_bounds__(256) void zero_bytes_kernel(unsigned* __restrict__ p, long long nwords) {
  const long long head = (4 - (((uintptr_t)p >> 2) & 3)) & 3;   // words in front of the first 16-byte boundary
  const long long h = head < nwords ? head : nwords, n4 = (nwords - h) >> 2, tail0 = h + 4 * n4;
  uint4* p4 = (uint4*)(p + h);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0) {
    if ((long long)threadIdx.x < h) p[threadIdx.x] = 0u;
    if (tail0 + threadIdx.x < nwords) p[tail0 + threadIdx.x] = 0u;
  }
}

extern "C" int cavp_zero_bytes(void* p, size_t nbytes, void* stream) {
  if (!p || nbytes == 0) return nbytes == 0 ? CAVP_OK : CAVP_ERR_BAD_ARG;
  if (((uintptr_t)p & 3) || (nbytes & 3)) return CAVP_ERR_ALIGN;
  long long nb = ((long long)(nbytes / 16) + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  zero_bytes_kernel<<<dim3((unsigned)nb), 256, 0, (hipStream_t)stream>>>((unsigned*)p, (long long)(nbytes / 4));
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

// *table[i] += inc for n int64 counters scattered in device memory (nn.BatchNorm2d.num_batches_tracked of every layer: one launch)
static __global__ __launch_bounds__(256) void i64_add_table_kernel(const long long* __restrict__ table, int n, long long inc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) *(long long*)(uintptr_t)table[i] += inc;
}

extern "C" int cavp_i64_add_table(const int64_t* table_dev, int32_t n, int64_t inc, void* stream) {
  if (!table_dev || n <= 0) return CAVP_ERR_BAD_ARG;
  i64_add_table_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>((const long long*)table_dev, n, (long long)inc);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_zero_ranges_f32(float* base, const int64_t* table_dev, int32_t nranges, int64_t max_len, void* stream) {
  if (!base || !table_dev || nranges <= 0 || max_len <= 0) return CAVP_ERR_BAD_ARG;
  if ((uintptr_t)base & 15) return CAVP_ERR_ALIGN;
  long long nb = (max_len / 4 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 512 ? 512 : nb);
  zero_ranges_kernel<<<dim3((unsigned)nb, (unsigned)nranges), 256, 0, (hipStream_t)stream>>>(base, (const long long*)table_dev);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}
