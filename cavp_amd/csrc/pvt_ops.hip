// PVTv2 (models/visual/backbones/pvt/pvt.py) specific kernels for gfx950: MFMA spatial-reduction attention with a
// wavefront softmax, depth-wise 3x3 conv (+bias +GELU) and the 7x7/stride-4 overlapping patch embedding.
#include "common.h"

namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// softmax(q k^T * scale) v  for one (batch, head) and 64 queries per workgroup (16 per wave), head_dim = 64, the
// whole K / V of the head resident in LDS (Nk <= 256 after the spatial reduction: pvt.py:76-79,113-118).
//   S^T[key][query] = K Q^T  (A = K rows from LDS, B = Q fragments from registers)  -> a lane owns ONE query column
//     and 4 keys per 16-key block: the row softmax is in-register max/sum + two cross-lane shuffles (xor 16, 32);
//   O^T[d][query]   = V^T P^T (A = V^T gathered with ds_read_b64_tr_b16, B = P^T straight from the S^T
//     accumulators: the MFMA k index is an arbitrary but shared permutation of the keys, so no layout shuffle).
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sra_attention_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                            T* __restrict__ o, int Nq, int Nk, int heads, float scale, int qpw) {
  constexpr int ES = (int)sizeof(T), HD = 64, ROWB = HD * ES;  // bytes per K / V row in LDS
  constexpr int VE = 16 / ES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;
  char* vs = smem + 256 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * HD;
  // ---- stage K, V of this head: rows key, 64 channels; 16-byte slots XOR-swizzled by (key & 7) (bf16: 8 slots/row,
  //      f32: 16 slots/row -> swizzle the low 3 bits) ----
  const T* kvb = kv + (size_t)b * Nk * 2 * C + h * HD;
  constexpr int SPR = ROWB / 16;  // slots per row
  for (int i = tid; i < 256 * SPR; i += 256) {
    const int key = i / SPR, sl = i - key * SPR;
    uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
    if (key < Nk) {
      kk = *(const uint4*)(kvb + (size_t)key * 2 * C + sl * VE);
      vv = *(const uint4*)(kvb + (size_t)key * 2 * C + C + sl * VE);
    }
    *(uint4*)(ks + key * ROWB + ((sl ^ (key & 7)) << 4)) = kk;
    *(uint4*)(vs + key * ROWB + (sl << 4)) = vv;   // V is read along keys (columns): keep it linear
  }
  __syncthreads();
  const int lrow = lane & 15, lgrp = lane >> 4;
  // `qpw` blocks of 64 queries per workgroup, one after the other on the same staged K / V: the 64 KiB staging (a global ->
  // LDS round trip of every thread) is paid once per workgroup, and the launch is sized to ONE round of two workgroups per CU
  for (int qb = 0; qb < qpw; ++qb) {
  const int q0 = ((int)blockIdx.x * qpw + qb) * 64 + wave * 16;
  if (q0 >= Nq) break;   // (wave-uniform)
  const int qi = q0 + lrow;  // this lane's query (B operand column / C-D column)
  const bool qok = qi < Nq;
  const T* qp = q + ((size_t)b * Nq + (qok ? qi : 0)) * C + h * HD;

  f32x4_t s[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if constexpr (ES == 2) {
    // B operand: Q^T[k = d][j = query]: lane (query lrow, group lgrp) holds d = 8*lgrp + e (+32 for the 2nd k-step)
    u32x4_t qf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) qf[j] = qok ? *(const u32x4_t*)(qp + j * 32 + lgrp * 8) : (u32x4_t){0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int key = kb * 16 + lrow, sl = j * 4 + lgrp;
        const u32x4_t kf = *(const u32x4_t*)(ks + key * ROWB + ((sl ^ (key & 7)) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[j]),
                                                        s[kb], 0, 0, 0);
      }
    }
  } else {
    // f32: 16 k-steps of 4 channels; lane group g supplies channel 4*ks + g ... use the shared-permutation trick:
    // lane reads a float4 at slot (ks) -> component c feeds MFMA c; both operands use d = 16*ks4 + 4*lgrp + c
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4_t qv = qok ? *(const u32x4_t*)(qp + j * 16 + lgrp * 4) : (u32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const int key = kb * 16 + lrow, sl = j * 4 + lgrp;
        const u32x4_t kf = *(const u32x4_t*)(ks + key * ROWB + ((sl ^ (key & 7)) << 4));
#pragma unroll
        for (int c = 0; c < 4; ++c)
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(kf[c]), __uint_as_float(qv[c]), s[kb], 0, 0, 0);
      }
    }
  }
  // ---- softmax over the keys of this lane's query: key = 16*kb + 4*lgrp + r ----
  float m = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb * 16 + lgrp * 4 + r;
      const float v = key < Nk ? s[kb][r] * scale : -INFINITY;
      s[kb][r] = v;
      m = fmaxf(m, v);
    }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __expf(s[kb][r] - m);   // (v_exp_f32 of x * log2 e: ~1e-6 relative, the precise expf is 4x the instructions)
      s[kb][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  // ---- O^T[d][query] = sum_key V[key][d] * P[query][key] ----
  f32x4_t acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) acc[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  if constexpr (ES == 2) {
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {  // pairs of 16-key blocks = one k-step of 32 keys
      // B operand (P^T): this lane's 8 keys = {4g + r} of block 2kp and {4g + r} of block 2kp+1
      u32x4_t pf;
      pf[0] = pack2bf(s[2 * kp][0], s[2 * kp][1]);
      pf[1] = pack2bf(s[2 * kp][2], s[2 * kp][3]);
      pf[2] = pack2bf(s[2 * kp + 1][0], s[2 * kp + 1][1]);
      pf[3] = pack2bf(s[2 * kp + 1][2], s[2 * kp + 1][3]);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        // A operand (V^T): lane i = d, same key set: tr-read rows r0 = 32kp + 4g (+16), lane s addresses row
        // r0 + (s >> 2), channels db*16 + 4 (s & 3)
        const int ra = 32 * kp + 4 * lgrp + (lrow >> 2);
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(vs + ra * ROWB + db * 32 + (lrow & 3) * 8));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(vs + (ra + 16) * ROWB + db * 32 + (lrow & 3) * 8));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        const u32x4_t vf = (u32x4_t){l2.x, l2.y, h2.x, h2.y};
        acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf),
                                                          acc[db], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb * 16 + lgrp * 4 + r;  // MFMA k index = lane group, key chosen per (kb, r)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const float vv = *(const float*)(vs + key * ROWB + (db * 16 + lrow) * 4);
          acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kb][r], acc[db], 0, 0, 0);
        }
      }
  }
  if (qok) {
    T* op = o + ((size_t)b * Nq + qi) * C + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int d = db * 16 + lgrp * 4;
      if constexpr (ES == 4) {
        *(float4*)(op + d) = make_float4(acc[db][0] * inv, acc[db][1] * inv, acc[db][2] * inv, acc[db][3] * inv);
      } else {
        uint2 w;
        w.x = pack2bf(acc[db][0] * inv, acc[db][1] * inv);
        w.y = pack2bf(acc[db][2] * inv, acc[db][3] * inv);
        *(uint2*)(op + d) = w;
      }
    }
  }
  }   // query blocks of this workgroup
}

// depth-wise 3x3, pad 1, + bias, optional exact GELU.  thread = (pixel, 16-byte channel vector); weights [9][C] f32
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w9c,
                                                        const float* __restrict__ bias, T* __restrict__ y, int N, int H,
                                                        int W, int C, int act, T* __restrict__ aux, int flip) {
  constexpr int VE = 16 / (int)sizeof(T);
  const int CV = C / VE;
  const long long total = (long long)N * H * W * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cv = (int)(idx % CV);
    const long long pix = idx / CV;
    const int wi = (int)(pix % W), hi = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = bias ? bias[cv * VE + e] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h2 = hi - 1 + kh;
      if ((unsigned)h2 >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w2 = wi - 1 + kw;
        if ((unsigned)w2 >= (unsigned)W) continue;
        const T* xp = x + ((size_t)(n * H + h2) * W + w2) * C + cv * VE;
        const float* wp = w9c + (flip ? 8 - (kh * 3 + kw) : kh * 3 + kw) * C + cv * VE;
        if constexpr (sizeof(T) == 4) {
          const float4 v = *(const float4*)xp;
          acc[0] += v.x * wp[0]; acc[1] += v.y * wp[1]; acc[2] += v.z * wp[2]; acc[3] += v.w * wp[3];
        } else {
          const uint4 v = *(const uint4*)xp;
          const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[2 * i] += __uint_as_float(u[i] << 16) * wp[2 * i];
            acc[2 * i + 1] += __uint_as_float(u[i] & 0xffff0000u) * wp[2 * i + 1];
          }
        }
      }
    }
    T* yp = y + (size_t)pix * C + cv * VE;
    if (aux) {   // GELU with its derivative as second output (see dwconv3x3_strip_kernel)
      float dg[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) gelu_and_grad(acc[e], acc[e], dg[e]);
      VecT<T>::store(yp, acc);
      VecT<T>::store(aux + (size_t)pix * C + cv * VE, dg);
      continue;
    }
    if constexpr (sizeof(T) == 4) {
      *(float4*)yp = make_float4(apply_act(acc[0], act), apply_act(acc[1], act), apply_act(acc[2], act), apply_act(acc[3], act));
    } else {
      uint4 t;
      unsigned* tu = (unsigned*)&t;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        tu[i] = pack2bf(apply_act(acc[2 * i], act), apply_act(acc[2 * i + 1], act));
      *(uint4*)yp = t;
    }
  }
}

// The same conv with a thread owning one 16-byte channel vector of SW = 4 (bf16) / 8 (f32) consecutive output pixels of a row:
// 3 x (SW + 2) input vectors serve SW outputs (4.5 / 3.75 loads per output instead of 9; SW = 8 for bf16 needed 200 VGPRs), the 9 x VE weights of the vector are fetched once per row of taps
// with 16-byte loads (the pixel-per-thread kernel above issued 72 scalar weight loads and four 64-bit divisions per output),
// and the index arithmetic is paid once per strip.  Lanes run along the channel vectors, so loads and stores stay coalesced.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const T* __restrict__ x, const float* __restrict__ w9c,
                                                              const float* __restrict__ bias, T* __restrict__ y, int N, int H,
                                                              int W, int C, int act, T* __restrict__ aux, int flip) {
  constexpr int VE = 16 / (int)sizeof(T), SW = 32 / VE;   // 32 accumulators per thread: 8 pixels x 4 (f32) / 4 pixels x 8 (bf16)
  const int CV = C / VE, strips = (W + SW - 1) / SW;
  const long long total = (long long)N * H * strips * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const unsigned u = (unsigned)idx;   // (total < 2^31 checked by the launcher)
    const unsigned r1 = u / (unsigned)CV, cv = u - r1 * (unsigned)CV;
    const unsigned r2 = r1 / (unsigned)strips, st = r1 - r2 * (unsigned)strips;
    const unsigned n = r2 / (unsigned)H, hi = r2 - n * (unsigned)H;
    const int w0 = (int)st * SW, c0 = (int)cv * VE;
    float acc[SW][VE];
    {
      float b[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) b[e] = 0.f;
      if (bias) {
#pragma unroll
        for (int q = 0; q < VE; q += 4) VecT<float>::load(bias + c0 + q, b + q);
      }
#pragma unroll
      for (int p = 0; p < SW; ++p)
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[p][e] = b[e];
    }
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {   // (not unrolled: one row of taps' inputs and weights live at a time)
      const int h2 = (int)hi - 1 + kh;
      if ((unsigned)h2 >= (unsigned)H) continue;
      float wt[3][VE];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int q = 0; q < VE; q += 4) VecT<float>::load(w9c + (size_t)(flip ? 8 - (kh * 3 + kw) : kh * 3 + kw) * C + c0 + q, wt[kw] + q);
      const T* row = x + ((size_t)(n * H + h2) * W) * C + c0;
      // (the 10 input vectors stay PACKED - 4 registers each for bf16 - and are widened where they are used: with them held as
      // f32 the kernel needed 204 VGPRs, two waves per SIMD)
      uint4 xr[SW + 2];
#pragma unroll
      for (int p = 0; p < SW + 2; ++p) {
        const int w2 = w0 - 1 + p;
        xr[p] = (unsigned)w2 < (unsigned)W ? *(const uint4*)(row + (size_t)w2 * C) : make_uint4(0u, 0u, 0u, 0u);
      }
      // input-major: every input vector is widened ONCE and scattered into the (up to 3) outputs it feeds
#pragma unroll
      for (int p = 0; p < SW + 2; ++p) {
        float xv[VE];
        const unsigned q[4] = {xr[p].x, xr[p].y, xr[p].z, xr[p].w};
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) xv[e] = __uint_as_float(q[e]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            xv[2 * i] = __uint_as_float(q[i] << 16);
            xv[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u);
          }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int o = p - kw;   // output pixel of the strip that reads input p through tap kw
          if (o >= 0 && o < SW) {
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[o][e] = fmaf(xv[e], wt[kw][e], acc[o][e]);
          }
        }
      }
    }
    const size_t off = ((size_t)(n * H + hi) * W + w0) * C + c0;
    T* yp = y + off;
    if (aux) {
      // Mlp.forward (pvt.py:46-55) dwconv -> GELU with gelu'(t) as second output: the backward multiplies it into the epilogue of
      // the GEMM that produces d(hidden) (cavp_conv_desc.aux_mode 2), so neither the pre-activation nor a separate GELU /
      // GELU-backward pass exists
#pragma unroll
      for (int p = 0; p < SW; ++p) {
        if (w0 + p < W) {
          float o[VE], dg[VE];
#pragma unroll
          for (int e = 0; e < VE; ++e) gelu_and_grad(acc[p][e], o[e], dg[e]);
          VecT<T>::store(yp + (size_t)p * C, o);
          VecT<T>::store(aux + off + (size_t)p * C, dg);
        }
      }
      continue;
    }
#pragma unroll
    for (int p = 0; p < SW; ++p) {
      if (w0 + p < W) {
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] = apply_act(acc[p][e], act);
        VecT<T>::store(yp + (size_t)p * C, o);
      }
    }
  }
}

// torch depth-wise weight [C][1][3][3] -> [9][C] f32
__global__ void pack_dw_kernel(const float* __restrict__ w, float* __restrict__ o, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 9 * C) { const int t = i / C, c = i - t * C; o[i] = w[c * 9 + t]; }
}

// direct KSxKS conv for Cin <= 3 (NCHW f32 in, NHWC out) with stride / pad / bias: OverlapPatchEmbed of stage 1
template <typename T>
__global__ __launch_bounds__(256) void conv_smallcin_kxk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, T* __restrict__ y, int N,
                                                                int Cin, int H, int W, int Cout, int KS, int stride,
                                                                int pad, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* ws = (float*)smem_raw;  // [Cin*KS*KS][Cout]
  const int K = Cin * KS * KS;
  for (int i = threadIdx.x; i < K * Cout; i += 256) {
    const int co = i / K, k = i - co * K;
    ws[k * Cout + co] = w[i];
  }
  __syncthreads();
  const int G = Cout >> 4;
  const long long total = (long long)N * Ho * Wo * G;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int g = (int)(idx % G);
    const long long pix = idx / G;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho);
    const int n = (int)(pix / ((long long)Wo * Ho));
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[g * 16 + j] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + ((size_t)n * Cin + ci) * H * W;
      for (int kh = 0; kh < KS; ++kh) {
        const int hi = ho * stride - pad + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kw = 0; kw < KS; ++kw) {
          const int wi = wo * stride - pad + kw;
          if ((unsigned)wi >= (unsigned)W) continue;
          const float xv = xp[(size_t)hi * W + wi];
          const float* wk = ws + ((ci * KS + kh) * KS + kw) * Cout + g * 16;
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] = fmaf(xv, wk[j], acc[j]);
        }
      }
    }
    T* yp = y + (size_t)pix * Cout + g * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) Elem<T>::st(yp + j, acc[j]);
  }
}

inline bool dt_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
}  // namespace
#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH

extern "C" int cavp_sra_attention(int32_t dtype, const void* q, const void* kv, void* o, int32_t B, int32_t Nq, int32_t Nk,
                                  int32_t heads, int32_t head_dim, float scale, void* stream) {
  if (!q || !kv || !o || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || head_dim != 64 || Nk > 256) return CAVP_ERR_UNSUPPORTED;
  if (((uintptr_t)q & 15) || ((uintptr_t)kv & 15) || ((uintptr_t)o & 15)) return CAVP_ERR_ALIGN;
  const int es = dtype == CAVP_F32 ? 4 : 2;
  const int lds = 2 * 256 * 64 * es;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)sra_attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 64 * 4);
    (void)hipFuncSetAttribute((const void*)sra_attention_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 64 * 2);
    attr = true;
  }
  const int qpw = cavp_sra_blocks_per_wg(Nq, B * heads);
  dim3 grid(((Nq + 63) / 64 + qpw - 1) / qpw, B * heads);
  if (dtype == CAVP_F32)
    sra_attention_kernel<float><<<grid, 256, lds, s>>>((const float*)q, (const float*)kv, (float*)o, Nq, Nk, heads, scale, qpw);
  else
    sra_attention_kernel<bf16_t><<<grid, 256, lds, s>>>((const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)o, Nq, Nk, heads, scale, qpw);
  CHECK_LAUNCH();
}

extern "C" int cavp_dwconv3x3_nhwc(int32_t dtype, const void* x, const float* w9c, const float* bias, void* y, int32_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t act, void* stream) {
  return cavp_dwconv3x3_nhwc_aux(dtype, x, w9c, bias, y, nullptr, N, H, W, C, act, stream);
}

namespace {
int dwconv_launch(int32_t dtype, const void* x, const float* w9c, const float* bias, void* y, void* aux, int32_t N, int32_t H,
                  int32_t W, int32_t C, int32_t act, int flip, void* stream) {
  if (!x || !w9c || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  if (aux && act != CAVP_ACT_GELU) return CAVP_ERR_BAD_ARG;   // the second output is gelu'(t)
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)aux) & 15) return CAVP_ERR_ALIGN;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int SW = 32 / VE;
  const long long strips_total = (long long)N * H * ((W + SW - 1) / SW) * (C / VE);
  if (W >= SW && strips_total < 0x7fffffffll && (((uintptr_t)w9c | (uintptr_t)bias) & 15) == 0 && C % 4 == 0) {
    long long nb = (strips_total + 255) / 256;
    if (nb > 32768) nb = 32768;
    if (dtype == CAVP_F32)
      dwconv3x3_strip_kernel<float><<<(int)nb, 256, 0, s>>>((const float*)x, w9c, bias, (float*)y, N, H, W, C, act, (float*)aux, flip);
    else
      dwconv3x3_strip_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)x, w9c, bias, (bf16_t*)y, N, H, W, C, act, (bf16_t*)aux, flip);
    CHECK_LAUNCH();
  }
  long long nb = ((long long)N * H * W * (C / VE) + 255) / 256;
  if (nb > 32768) nb = 32768;
  if (dtype == CAVP_F32)
    dwconv3x3_kernel<float><<<(int)nb, 256, 0, s>>>((const float*)x, w9c, bias, (float*)y, N, H, W, C, act, (float*)aux, flip);
  else
    dwconv3x3_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)x, w9c, bias, (bf16_t*)y, N, H, W, C, act, (bf16_t*)aux, flip);
  CHECK_LAUNCH();
}
}  // namespace

extern "C" int cavp_dwconv3x3_nhwc_aux(int32_t dtype, const void* x, const float* w9c, const float* bias, void* y, void* aux,
                                       int32_t N, int32_t H, int32_t W, int32_t C, int32_t act, void* stream) {
  return dwconv_launch(dtype, x, w9c, bias, y, aux, N, H, W, C, act, 0, stream);
}

// data gradient of the depth-wise conv: dx = correlation of dy with the REVERSED taps (row 8 - t of the same [9][C] weights)
extern "C" int cavp_dwconv3x3_bwd_data_nhwc(int32_t dtype, const void* dy, const float* w9c, void* dx, int32_t N, int32_t H,
                                            int32_t W, int32_t C, void* stream) {
  return dwconv_launch(dtype, dy, w9c, nullptr, dx, nullptr, N, H, W, C, CAVP_ACT_NONE, 1, stream);
}

extern "C" int cavp_pack_dwconv_weight(const float* w_c133, float* w9c, int32_t C, void* stream) {
  if (!w_c133 || !w9c || C <= 0) return CAVP_ERR_BAD_ARG;
  pack_dw_kernel<<<(9 * C + 255) / 256, 256, 0, (hipStream_t)stream>>>(w_c133, w9c, C);
  CHECK_LAUNCH();
}

extern "C" int cavp_conv_smallcin_kxk_nchw(int32_t dtype, const float* x_nchw, const float* w_oihw, const float* bias,
                                           void* y_nhwc, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout,
                                           int32_t KS, int32_t stride, int32_t pad, void* stream) {
  if (!x_nchw || !w_oihw || !y_nhwc || N <= 0 || H <= 0 || W <= 0 || stride <= 0 || KS <= 0 || pad < 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || Cin < 1 || Cin > 3 || Cout % 16 || (size_t)Cin * KS * KS * Cout * 4 > 64 * 1024) return CAVP_ERR_UNSUPPORTED;
  const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
  const long long total = (long long)N * Ho * Wo * (Cout / 16);
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  const size_t lds = (size_t)Cin * KS * KS * Cout * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    conv_smallcin_kxk_kernel<float><<<(int)nb, 256, lds, s>>>(x_nchw, w_oihw, bias, (float*)y_nhwc, N, Cin, H, W, Cout, KS, stride, pad, Ho, Wo);
  else
    conv_smallcin_kxk_kernel<bf16_t><<<(int)nb, 256, lds, s>>>(x_nchw, w_oihw, bias, (bf16_t*)y_nhwc, N, Cin, H, W, Cout, KS, stride, pad, Ho, Wo);
  CHECK_LAUNCH();
}
