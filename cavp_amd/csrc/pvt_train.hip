// Backward kernels of the PVTv2-B5 backbone (models/visual/backbones/pvt/pvt.py) for gfx950: spatial-reduction attention
// (dq | dk, dv), depth-wise 3x3 conv weight gradient, the im2col behind the 7x7 / stride-4 patch embedding's weight gradient, the sr x sr
// space-to-depth rearrangement that turns the spatial-reduction conv (kernel = stride) into a token GEMM, and the
// stochastic-depth residual add.  The attention backward exists twice: on the f32 matrix pipe (v_mfma_f32_16x16x4f32, the
// f32 parity path) and on v_mfma_f32_16x16x32_bf16 for bf16 storage.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 t = *(const uint2*)p;
  return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                     __uint_as_float(t.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *(uint2*)p = make_uint2(pack2bf(a, b), pack2bf(c, d));
}

// LDS tile of f32 rows of 64 channels (256 bytes); the 16-byte slot index is XORed with (row & 7) so that 16 lanes reading
// the same slot of 16 consecutive rows spread over the banks.
__device__ __forceinline__ int tile_off(int row, int d) { return row * 256 + ((((d >> 2) ^ (row & 7))) << 4) + ((d & 3) << 2); }
__device__ __forceinline__ float4 tile_ld4(const char* base, int row, int slot) {
  return *(const float4*)(base + row * 256 + ((slot ^ (row & 7)) << 4));
}
__device__ __forceinline__ float tile_ld1(const char* base, int row, int d) { return *(const float*)(base + tile_off(row, d)); }

// stage `rows` rows (row r of the source = src + r * ld, 64 channels of T) into a swizzled f32 tile of `cap` rows (zero fill)
template <typename T>
__device__ __forceinline__ void stage_tile(char* dst, const T* src, size_t ld, int rows, int cap, int tid) {
  for (int i = tid; i < cap * 16; i += 256) {
    const int r = i >> 4, sl = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = ld4<T>(src + (size_t)r * ld + sl * 4);
    *(float4*)(dst + r * 256 + ((sl ^ (r & 7)) << 4)) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Attention backward, part 1 (pvt.py:120-126): one (batch, head) and 64 queries per workgroup, K and V of the head in LDS.
//   S^T[key][query] = K Q^T, P = softmax(scale S), dP^T = V dO^T, delta = sum_key P dP, dS = scale P (dP - delta),
//   dQ^T[d][query] = sum_key K[key][d] dS^T[key][query].  Also writes the row statistics (log-sum-exp, delta) that part 2
//   needs.  Same lane <-> (key, query) mapping as the forward kernel (pvt_ops.hip): a lane owns one query column.
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sra_bwd_q_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                        const T* __restrict__ dout, T* __restrict__ dq,
                                                        float* __restrict__ stats, int Nq, int Nk, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;
  char* vs = smem + 256 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * 64;
  const T* kvb = kv + (size_t)b * Nk * 2 * C + h * 64;
  stage_tile<T>(ks, kvb, (size_t)2 * C, Nk, 256, tid);
  stage_tile<T>(vs, kvb + C, (size_t)2 * C, Nk, 256, tid);
  __syncthreads();
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int qi = blockIdx.x * 64 + wave * 16 + lrow;
  const bool qok = qi < Nq;
  const size_t qoff = ((size_t)b * Nq + (qok ? qi : 0)) * C + h * 64;
  f32x4_t s[16], dp[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) { s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[kb] = s[kb]; }
  // MFMA k index of instruction (j, c) in lane group g <-> channel d = 16 j + 4 g + c, for both operands
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
    if (qok) { qv = ld4<T>(q + qoff + j * 16 + lgrp * 4); gv = ld4<T>(dout + qoff + j * 16 + lgrp * 4); }
    const float qa[4] = {qv.x, qv.y, qv.z, qv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const int key = kb * 16 + lrow;
      const float4 kf = tile_ld4(ks, key, j * 4 + lgrp), vf = tile_ld4(vs, key, j * 4 + lgrp);
      const float ka[4] = {kf.x, kf.y, kf.z, kf.w}, va[4] = {vf.x, vf.y, vf.z, vf.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[c], qa[c], s[kb], 0, 0, 0);
        dp[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[c], ga[c], dp[kb], 0, 0, 0);
      }
    }
  }
  // softmax over the keys of this lane's query: key = 16 kb + 4 lgrp + r
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
  float delta = 0.f;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[kb][r] *= inv;
      delta += s[kb][r] * dp[kb][r];   // padded keys: p = 0
    }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kb][r] = s[kb][r] * (dp[kb][r] - delta) * scale;   // dS^T
  // dQ^T[d][query] = sum_key K[key][d] dS^T[key][query]: MFMA (kb, r) has k index lgrp <-> key 16 kb + 4 lgrp + r
  f32x4_t acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) acc[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb * 16 + lgrp * 4 + r;
#pragma unroll
      for (int db = 0; db < 4; ++db)
        acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_ld1(ks, key, db * 16 + lrow), s[kb][r], acc[db], 0, 0, 0);
    }
  if (qok) {
#pragma unroll
    for (int db = 0; db < 4; ++db) st4<T>(dq + qoff + db * 16 + lgrp * 4, acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
    if (lgrp == 0) {
      stats[(size_t)bh * Nq + qi] = m + logf(sum);
      stats[(size_t)gridDim.y * Nq + (size_t)bh * Nq + qi] = delta;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Attention backward, part 2: dK, dV.  Workgroup = (query split, batch x head, block of 64 keys); wave w owns 16 keys whose
// K / V rows stay in registers; chunks of 64 queries (Q, dO as f32 tiles in LDS) stream past them:
//   S[query][key] = Q K^T, P = exp(scale S - lse), dP = dO V^T, dS = scale P (dP - delta),
//   dV^T[d][key] += sum_query dO^T[d][query] P[query][key],  dK^T[d][key] += sum_query Q^T[d][query] dS[query][key].
// Each query split stores its partial sums into its own f32 slab; sra_bwd_kv_reduce_kernel adds the slabs in split order
// (deterministic; a single split writes the gradient itself).
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sra_bwd_kv_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                         const T* __restrict__ dout, const float* __restrict__ stats,
                                                         float* __restrict__ dkv, int Nq, int Nk, int heads, float scale,
                                                         size_t slab_stride) {
  __shared__ __attribute__((aligned(16))) char qs[64 * 256];
  __shared__ __attribute__((aligned(16))) char gs[64 * 256];
  __shared__ float lse_s[64], del_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * 64;
  const int key = blockIdx.z * 64 + wave * 16 + lrow;
  const bool kok = key < Nk;
  float4 kr[4], vr[4];
  {
    const T* kp = kv + ((size_t)b * Nk + (kok ? key : 0)) * 2 * C + h * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kr[j] = kok ? ld4<T>(kp + j * 16 + lgrp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      vr[j] = kok ? ld4<T>(kp + C + j * 16 + lgrp * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) { dk[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[db] = dk[db]; }
  const int chunks = (Nq + 63) / 64;
  for (int ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const int q0 = ch * 64, rows = min(64, Nq - q0);
    __syncthreads();   // the previous chunk's tiles are no longer read
    stage_tile<T>(qs, q + ((size_t)b * Nq + q0) * C + h * 64, (size_t)C, rows, 64, tid);
    stage_tile<T>(gs, dout + ((size_t)b * Nq + q0) * C + h * 64, (size_t)C, rows, 64, tid);
    if (tid < 64) {
      const bool ok = tid < rows;
      lse_s[tid] = ok ? stats[(size_t)bh * Nq + q0 + tid] : 0.f;
      del_s[tid] = ok ? stats[(size_t)gridDim.y * Nq + (size_t)bh * Nq + q0 + tid] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int qb = 0; qb < 4; ++qb) {
      f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, da = sa;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 qf = tile_ld4(qs, qb * 16 + lrow, j * 4 + lgrp), gf = tile_ld4(gs, qb * 16 + lrow, j * 4 + lgrp);
        const float qa[4] = {qf.x, qf.y, qf.z, qf.w}, ga[4] = {gf.x, gf.y, gf.z, gf.w};
        const float ka[4] = {kr[j].x, kr[j].y, kr[j].z, kr[j].w}, va[4] = {vr[j].x, vr[j].y, vr[j].z, vr[j].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sa = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c], ka[c], sa, 0, 0, 0);
          da = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[c], va[c], da, 0, 0, 0);
        }
      }
      // sa[r] = S[query = 16 qb + 4 lgrp + r][key = this lane's]
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qb * 16 + lgrp * 4 + r;
        const bool ok = kok && ql < rows;
        p[r] = ok ? __expf(sa[r] * scale - lse_s[ql]) : 0.f;
        ds[r] = p[r] * (da[r] - del_s[ql]) * scale;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qb * 16 + lgrp * 4 + r;   // MFMA k index lgrp <-> query ql
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dv[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_ld1(gs, ql, db * 16 + lrow), p[r], dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(tile_ld1(qs, ql, db * 16 + lrow), ds[r], dk[db], 0, 0, 0);
        }
      }
    }
  }
  if (kok) {
    // every (key, channel) of this (batch, head, key block) is owned by ONE lane of one workgroup per query split: each split
    // stores its partial into its own slab with 16-byte stores (split 0 of a single-split launch writes dkv itself) and
    // sra_bwd_kv_reduce_kernel adds the slabs in split order.  (Round 2 added every partial with scalar f32 atomics onto a
    // zero-filled dkv: 2 M atomics per launch at 8 splits, and atomics even with one split.)
    float* op = dkv + (size_t)blockIdx.x * slab_stride + ((size_t)b * Nk + key) * 2 * C + h * 64;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      *(float4*)(op + db * 16 + lgrp * 4) = make_float4(dk[db][0], dk[db][1], dk[db][2], dk[db][3]);
      *(float4*)(op + C + db * 16 + lgrp * 4) = make_float4(dv[db][0], dv[db][1], dv[db][2], dv[db][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 storage: the same two kernels on v_mfma_f32_16x16x32_bf16 (8x the f32 matrix rate).  Tiles stay bf16 in LDS; an
// operand that is multiplied along its ROWS (K in dQ = dS K, Q / dO in dK = dS^T Q, dV = P^T dO) is read with the
// transposing LDS load (ds_read_b64_tr_b16) from a second, un-swizzled copy of the tile - the same idiom as V^T in the
// forward kernel (pvt_ops.hip); P and dS go from the f32 accumulators to the B operand as bf16 without a layout change
// (MFMA k index = an arbitrary but shared permutation of the keys / queries).
// ------------------------------------------------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4_t btile_ld(const char* base, int row, int slot) {
  return *(const u32x4_t*)(base + row * 128 + ((slot ^ (row & 7)) << 4));
}
// A operand X^T[i = channel db*16 + lrow][k <-> rows r0 + {0..3} and r0 + 16 + {0..3}] from a linear bf16 tile (128-byte rows)
__device__ __forceinline__ bf16x8_t btile_tr(const char* base, int r0, int db, int lrow) {
  const int ra = r0 + (lrow >> 2);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(base + ra * 128 + db * 32 + (lrow & 3) * 8));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(base + (ra + 16) * 128 + db * 32 + (lrow & 3) * 8));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, (u32x4_t){l2.x, l2.y, h2.x, h2.y});
}
// the same operand from the SWIZZLED tile (bstage's `sw`): a lane's 8 bytes are half of the 16-byte slot 2 db + (lrow & 3) / 2 of
// its row, which lives at slot ^ (row & 7); rows ra and ra + 16 share the swizzle
__device__ __forceinline__ bf16x8_t btile_tr_sw(const char* base, int r0, int db, int lrow) {
  const int ra = r0 + (lrow >> 2);
  const int off = ra * 128 + (((db * 2 + ((lrow & 3) >> 1)) ^ (ra & 7)) << 4) + (lrow & 1) * 8;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + off));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + off + 16 * 128));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8_t, (u32x4_t){l2.x, l2.y, h2.x, h2.y});
}
__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
  return __builtin_bit_cast(bf16x8_t, (u32x4_t){pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])});
}
// stage rows of 64 bf16 channels: swizzled copy `sw` and (optional) linear copy `lin`; zero fill up to `cap` rows
__device__ __forceinline__ void bstage(char* sw, char* lin, const bf16_t* src, size_t ld, int rows, int cap, int tid) {
  for (int i = tid; i < cap * 8; i += 256) {
    const int r = i >> 3, sl = i & 7;
    u32x4_t v = (u32x4_t){0, 0, 0, 0};
    if (r < rows) v = *(const u32x4_t*)(src + (size_t)r * ld + sl * 8);
    *(u32x4_t*)(sw + r * 128 + ((sl ^ (r & 7)) << 4)) = v;
    if (lin) *(u32x4_t*)(lin + r * 128 + (sl << 4)) = v;
  }
}

__global__ __launch_bounds__(256) void sra_bwd_q_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                             const bf16_t* __restrict__ dout, bf16_t* __restrict__ dq,
                                                             float* __restrict__ stats, int Nq, int Nk, int heads, float scale,
                                                             int qpw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;                 // K, swizzled rows (A operand of S^T = K Q^T; transposed reads for dQ^T = K^T dS^T)
  char* vs = smem + 256 * 128;     // V, swizzled rows (A operand of dP^T = V dO^T)
  // (64 KiB: two workgroups per CU.  A third, linear copy of K for the transposed reads made it 96 KiB and one per CU.)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * 64;
  const bf16_t* kvb = kv + (size_t)b * Nk * 2 * C + h * 64;
  bstage(ks, nullptr, kvb, (size_t)2 * C, Nk, 256, tid);
  bstage(vs, nullptr, kvb + C, (size_t)2 * C, Nk, 256, tid);
  __syncthreads();
  const int lrow = lane & 15, lgrp = lane >> 4;
  for (int qb = 0; qb < qpw; ++qb) {   // query blocks of this workgroup on the same staged K / V (cavp_sra_blocks_per_wg)
  const int q0 = ((int)blockIdx.x * qpw + qb) * 64 + wave * 16;
  if (q0 >= Nq) break;   // (wave-uniform)
  const int qi = q0 + lrow;
  const bool qok = qi < Nq;
  const size_t qoff = ((size_t)b * Nq + (qok ? qi : 0)) * C + h * 64;
  u32x4_t qf[2], gf[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    qf[j] = qok ? *(const u32x4_t*)(q + qoff + j * 32 + lgrp * 8) : (u32x4_t){0, 0, 0, 0};
    gf[j] = qok ? *(const u32x4_t*)(dout + qoff + j * 32 + lgrp * 8) : (u32x4_t){0, 0, 0, 0};
  }
  f32x4_t s[16], dp[16];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    s[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    dp[kb] = s[kb];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = kb * 16 + lrow;
      s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, btile_ld(ks, key, j * 4 + lgrp)),
                                                      __builtin_bit_cast(bf16x8_t, qf[j]), s[kb], 0, 0, 0);
      dp[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, btile_ld(vs, key, j * 4 + lgrp)),
                                                       __builtin_bit_cast(bf16x8_t, gf[j]), dp[kb], 0, 0, 0);
    }
  }
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
  float delta = 0.f;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[kb][r] *= inv;
      delta += s[kb][r] * dp[kb][r];
    }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) s[kb][r] = s[kb][r] * (dp[kb][r] - delta) * scale;   // dS^T
  f32x4_t acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) acc[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kp = 0; kp < 8; ++kp) {   // one k step = 32 keys = the 16-key blocks 2 kp, 2 kp + 1
    const bf16x8_t df = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
    for (int db = 0; db < 4; ++db)
      acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(btile_tr_sw(ks, 32 * kp + 4 * lgrp, db, lrow), df, acc[db], 0, 0, 0);
  }
  if (qok) {
#pragma unroll
    for (int db = 0; db < 4; ++db) st4<bf16_t>(dq + qoff + db * 16 + lgrp * 4, acc[db][0], acc[db][1], acc[db][2], acc[db][3]);
    if (lgrp == 0) {
      stats[(size_t)bh * Nq + qi] = m + logf(sum);
      stats[(size_t)gridDim.y * Nq + (size_t)bh * Nq + qi] = delta;
    }
  }
  }   // query blocks of this workgroup
}

__global__ __launch_bounds__(256) void sra_bwd_kv_bf16_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ stats,
                                                              float* __restrict__ dkv, int Nq, int Nk, int heads, float scale,
                                                              size_t slab_stride) {
  // Q and dO chunks of 64 queries, swizzled rows: row-major fragments for S = Q K^T / dP = dO V^T and (btile_tr_sw) transposed
  // ones for dK = dS^T Q / dV = P^T dO from the same copy.  The next chunk's rows are fetched into registers while this one is
  // multiplied (the loop used to stall on a global round trip per chunk: 16 chunks x ~2 us for the stage-3 shape).
  __shared__ __attribute__((aligned(16))) char qs[64 * 128];
  __shared__ __attribute__((aligned(16))) char gs[64 * 128];
  __shared__ float lse_s[64], del_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, lgrp = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * 64;
  const int key = blockIdx.z * 64 + wave * 16 + lrow;
  const bool kok = key < Nk;
  u32x4_t kr[2], vr[2];
  {
    const bf16_t* kp = kv + ((size_t)b * Nk + (kok ? key : 0)) * 2 * C + h * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      kr[j] = kok ? *(const u32x4_t*)(kp + j * 32 + lgrp * 8) : (u32x4_t){0, 0, 0, 0};
      vr[j] = kok ? *(const u32x4_t*)(kp + C + j * 32 + lgrp * 8) : (u32x4_t){0, 0, 0, 0};
    }
  }
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) { dk[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[db] = dk[db]; }
  const int chunks = (Nq + 63) / 64;
  u32x4_t pq[2], pg[2];
  float pl = 0.f, pd = 0.f;
  auto fetch = [&](int ch) {
    const int q0 = ch * 64, rows = min(64, Nq - q0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = (tid >> 3) + 32 * k, sl = tid & 7;
      const size_t at = ((size_t)b * Nq + q0 + r) * C + h * 64 + sl * 8;
      pq[k] = r < rows ? *(const u32x4_t*)(q + at) : (u32x4_t){0, 0, 0, 0};
      pg[k] = r < rows ? *(const u32x4_t*)(dout + at) : (u32x4_t){0, 0, 0, 0};
    }
    if (tid < 64) {
      const bool ok = tid < rows;
      pl = ok ? stats[(size_t)bh * Nq + q0 + tid] : 0.f;
      pd = ok ? stats[(size_t)gridDim.y * Nq + (size_t)bh * Nq + q0 + tid] : 0.f;
    }
  };
  if ((int)blockIdx.x < chunks) fetch(blockIdx.x);
  for (int ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    const int rows = min(64, Nq - ch * 64);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = (tid >> 3) + 32 * k, sl = tid & 7;
      *(u32x4_t*)(qs + r * 128 + ((sl ^ (r & 7)) << 4)) = pq[k];
      *(u32x4_t*)(gs + r * 128 + ((sl ^ (r & 7)) << 4)) = pg[k];
    }
    if (tid < 64) { lse_s[tid] = pl; del_s[tid] = pd; }
    __syncthreads();
    if (ch + (int)gridDim.x < chunks) fetch(ch + gridDim.x);
    f32x4_t p[4], ds[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      f32x4_t sa = (f32x4_t){0.f, 0.f, 0.f, 0.f}, da = sa;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, btile_ld(qs, qb * 16 + lrow, j * 4 + lgrp)),
                                                     __builtin_bit_cast(bf16x8_t, kr[j]), sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, btile_ld(gs, qb * 16 + lrow, j * 4 + lgrp)),
                                                     __builtin_bit_cast(bf16x8_t, vr[j]), da, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qlr = qb * 16 + lgrp * 4 + r;
        const bool ok = kok && qlr < rows;
        p[qb][r] = ok ? __expf(sa[r] * scale - lse_s[qlr]) : 0.f;
        ds[qb][r] = p[qb][r] * (da[r] - del_s[qlr]) * scale;
      }
    }
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {   // one k step = 32 queries = the 16-query blocks 2 qp, 2 qp + 1
      const bf16x8_t pf = pack8(p[2 * qp], p[2 * qp + 1]), df = pack8(ds[2 * qp], ds[2 * qp + 1]);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(btile_tr_sw(gs, 32 * qp + 4 * lgrp, db, lrow), pf, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(btile_tr_sw(qs, 32 * qp + 4 * lgrp, db, lrow), df, dk[db], 0, 0, 0);
      }
    }
  }
  if (kok) {
    // every (key, channel) of this (batch, head, key block) is owned by ONE lane of one workgroup per query split: each split
    // stores its partial into its own slab with 16-byte stores (split 0 of a single-split launch writes dkv itself) and
    // sra_bwd_kv_reduce_kernel adds the slabs in split order.  (Round 2 added every partial with scalar f32 atomics onto a
    // zero-filled dkv: 2 M atomics per launch at 8 splits, and atomics even with one split.)
    float* op = dkv + (size_t)blockIdx.x * slab_stride + ((size_t)b * Nk + key) * 2 * C + h * 64;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      *(float4*)(op + db * 16 + lgrp * 4) = make_float4(dk[db][0], dk[db][1], dk[db][2], dk[db][3]);
      *(float4*)(op + C + db * 16 + lgrp * 4) = make_float4(dv[db][0], dv[db][1], dv[db][2], dv[db][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Depth-wise 3x3 (pad 1) weight + bias gradient: dw[c][t] += sum_p x[p + off_t][c] g[p][c], db[c] += sum_p g[p][c].
// thread = 4 channels x one of 16 pixel lanes (a wave = 16 channel quads x 4 pixel lanes: 128 contiguous bytes per pixel);
// workgroup = 16 channel quads x `ppb` pixels; the 16 pixel lanes are reduced with two cross-lane shuffles and a pass through
// LDS, then one atomic per (channel, tap) and workgroup.
// ------------------------------------------------------------------------------------------------------------------------
// DX = true: the same walk also produces the DATA gradient.  The sum is re-indexed by the input pixel p': dw[t] = sum_p' x[p'] *
// g[p' - off_t] and dx[p'] = sum_t w[t] * g[p' - off_t] use the same nine neighbours of g (one load of x and nine of g per pixel
// instead of nine of x and one of g - the same traffic), so the separate data-gradient launch (a second pass over g) is gone.
template <typename T, bool DX>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                              float* __restrict__ dw, float* __restrict__ db, int N, int H,
                                                              int W, int C, int ppb, float* __restrict__ part_dw,
                                                              float* __restrict__ part_db, const float* __restrict__ w9c,
                                                              T* __restrict__ dx) {
  __shared__ float red[4][16][41];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int cv = blockIdx.x * 16 + cl;
  const bool cok = cv * 4 < C;
  const long long total = (long long)N * H * W;
  const long long p0 = (long long)blockIdx.y * ppb, p1 = min(total, p0 + ppb);
  float acc[40];
#pragma unroll
  for (int i = 0; i < 40; ++i) acc[i] = 0.f;
  float wt[DX ? 36 : 1];
  if constexpr (DX) {
    if (cok) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 wv = *(const float4*)(w9c + (size_t)t * C + cv * 4);
        wt[4 * t] = wv.x; wt[4 * t + 1] = wv.y; wt[4 * t + 2] = wv.z; wt[4 * t + 3] = wv.w;
      }
    }
  }
  if (cok)
    for (long long p = p0 + pl; p < p1; p += 16) {
      const int wi = (int)(p % W), hi = (int)((p / W) % H);
      if constexpr (DX) {
        const float4 xv = ld4<T>(x + (size_t)p * C + cv * 4);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int h2 = hi + 1 - kh;          // g at p - off_t, off_t = (kh - 1, kw - 1)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int w2 = wi + 1 - kw;
            const bool ok = (unsigned)h2 < (unsigned)H && (unsigned)w2 < (unsigned)W;
            const long long off = ok ? (long long)(1 - kh) * W + (1 - kw) : 0;
            const float4 gv = ld4<T>(g + ((size_t)p + off) * C + cv * 4);
            const float m = ok ? 1.f : 0.f;
            const float ga[4] = {gv.x * m, gv.y * m, gv.z * m, gv.w * m};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[(kh * 3 + kw) * 4 + e] += xa[e] * ga[e];
              d[e] += wt[(kh * 3 + kw) * 4 + e] * ga[e];
              if (kh == 1 && kw == 1) acc[36 + e] += ga[e];
            }
          }
        }
        st4<T>(dx + (size_t)p * C + cv * 4, d[0], d[1], d[2], d[3]);
      } else {
        const float4 gv = ld4<T>(g + (size_t)p * C + cv * 4);
        const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[36 + e] += ga[e];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int h2 = hi - 1 + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int w2 = wi - 1 + kw;
            const bool ok = (unsigned)h2 < (unsigned)H && (unsigned)w2 < (unsigned)W;
            // branch-free: an out-of-image tap reads the centre pixel and is multiplied by 0 (keeps the 9 loads in flight together)
            const long long off = ok ? (long long)(kh - 1) * W + (kw - 1) : 0;
            const float4 xv = ld4<T>(x + ((size_t)p + off) * C + cv * 4);
            const float m = ok ? 1.f : 0.f;
            const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[(kh * 3 + kw) * 4 + e] += xa[e] * (ga[e] * m);
          }
        }
      }
    }
#pragma unroll
  for (int i = 0; i < 40; ++i) {   // the wave's 4 pixel lanes (lane bits 4, 5)
    acc[i] += __shfl_xor(acc[i], 16, 64);
    acc[i] += __shfl_xor(acc[i], 32, 64);
  }
  if ((threadIdx.x & 63) < 16)
#pragma unroll
    for (int i = 0; i < 40; ++i) red[wave][cl][i] = acc[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * 40; i += 256) {
    const int c4 = i / 40, k = i - c4 * 40;
    const int c0 = (blockIdx.x * 16 + c4) * 4;
    if (c0 >= C) continue;
    const float v = red[0][c4][k] + red[1][c4][k] + red[2][c4][k] + red[3][c4][k];
    const int t = k >> 2, e = k & 3;
    if (part_dw) {   // deterministic mode: one writer per (pixel split, channel, tap); cavp_det_finish adds the splits in order
      if (t < 9) part_dw[(size_t)blockIdx.y * 9 * C + (c0 + e) * 9 + t] = v;
      else if (db) part_db[(size_t)blockIdx.y * C + c0 + e] = v;
    } else if (t < 9) atomicAdd(dw + (c0 + e) * 9 + t, v);
    else if (db) atomicAdd(db + c0 + e, v);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// im2col of the KS x KS / Cin <= 3 patch embedding input (NCHW f32) into [pixels][Kpad] rows of T, column order (ci, kh, kw)
// = the OIHW weight's; columns >= Cin*KS*KS are zero.  The weight gradient is then the token-GEMM weight gradient
// (cavp_conv2d_wgrad_nhwc, 1x1) of these rows against dy: 2.5 GFLOP on the matrix pipe instead of 1.2 G scalar-fed FMAs.
// thread = one pixel x 8 consecutive columns (one 16-byte store for bf16).
// ------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void smallcin_kxk_im2col_kernel(const float* __restrict__ x, T* __restrict__ cols, int N, int Cin,
                                                                  int H, int W, int KS, int stride, int pad, int Ho, int Wo,
                                                                  int Kpad) {
  const int K = Cin * KS * KS, G = Kpad / 8;
  const long long total = (long long)N * Ho * Wo * G;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int g = (int)(idx % G);
    const long long p = idx / G;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((long long)Wo * Ho));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = g * 8 + e;
      const int kk = k < K ? k : 0;
      const int ci = kk / (KS * KS), rem = kk - ci * KS * KS, kh = rem / KS, kw = rem - kh * KS;
      const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
      const bool ok = k < K && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const float t = x[(((size_t)n * Cin + ci) * H + (ok ? hi : 0)) * W + (ok ? wi : 0)];
      v[e] = ok ? t : 0.f;
    }
    T* dst = cols + (size_t)p * Kpad + g * 8;
    if constexpr (sizeof(T) == 4) {
      *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      *(uint4*)dst = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
  }
}

// [B][H][W][C] <-> [B][H/s][W/s][s*s*C] (patch element order (row, column, channel) = OHWI weight order); 16-byte vectors
template <typename T>
__global__ __launch_bounds__(256) void space_to_depth_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H,
                                                             int W, int C, int s, int inverse) {
  constexpr int VE = 16 / (int)sizeof(T);
  const int CV = C / VE, Ho = H / s, Wo = W / s;
  const long long total = (long long)B * H * W * CV;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    // i enumerates the patch-major tensor: (b, yo, xo, r, c2, cv)
    long long t = i;
    const int cv = (int)(t % CV); t /= CV;
    const int c2 = (int)(t % s); t /= s;
    const int r = (int)(t % s); t /= s;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const size_t img = (((size_t)b * H + yo * s + r) * W + xo * s + c2) * C + (size_t)cv * VE;
    if (inverse) *(uint4*)(dst + img) = *(const uint4*)(src + i * VE);
    else *(uint4*)(dst + i * VE) = *(const uint4*)(src + img);
  }
}

// stochastic depth (timm DropPath, pvt.py:167-168): out = x + s[sample] * branch  (x == nullptr: out = s[sample] * branch)
template <typename T>
__global__ __launch_bounds__(256) void row_scale_add_kernel(const T* __restrict__ x, const T* __restrict__ br,
                                                            const float* __restrict__ s, T* __restrict__ out,
                                                            long long per_sample_vec, long long total_vec) {
  constexpr int VE = VecT<T>::VE;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total_vec; i += (long long)gridDim.x * 256) {
    const float sc = s[i / per_sample_vec];
    float a[VE], v[VE];
    VecT<T>::load(br + i * VE, v);
    if (x) {
      VecT<T>::load(x + i * VE, a);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] = a[e] + sc * v[e];
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] *= sc;
    }
    VecT<T>::store(out + i * VE, v);
  }
}

// dkv[i] = sum_s slabs[s][i] (split order), 16-byte vectors; TO = float or bf16_t (the compute dtype of the consumer: the cast pass
// that used to follow is folded in)
template <typename TO>
__global__ __launch_bounds__(256) void sra_bwd_kv_reduce_kernel(const float* __restrict__ slabs, TO* __restrict__ dkv,
                                                                long long n4, int splits, size_t slab_stride) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 a = *(const float4*)(slabs + 4 * i);
    for (int s = 1; s < splits; ++s) {
      const float4 v = *(const float4*)(slabs + (size_t)s * slab_stride + 4 * i);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4<TO>(dkv + 4 * i, a.x, a.y, a.z, a.w);
  }
}

inline bool dt_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
}  // namespace
#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH

// row statistics (log-sum-exp, delta) + the dK / dV slabs of the query splits: splits * B * Nk * 2C floats with
// splits <= 256 / (B * heads) and Nk <= 256, i.e. at most 256 * 256 * 128 floats whatever the shape
extern "C" size_t cavp_sra_attention_bwd_workspace_bytes(int32_t B, int32_t Nq, int32_t heads) {
  return ((size_t)2 * B * heads * Nq + (size_t)256 * 256 * 128 + 64) * sizeof(float);
}

extern "C" int cavp_sra_attention_bwd(int32_t dtype, const void* q, const void* kv, const void* dout, void* dq, float* dkv,
                                      int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t head_dim, float scale,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return cavp_sra_attention_bwd_to(dtype, q, kv, dout, dq, dkv, CAVP_F32, B, Nq, Nk, heads, head_dim, scale, workspace,
                                   workspace_bytes, stream);
}

extern "C" int cavp_sra_attention_bwd_to(int32_t dtype, const void* q, const void* kv, const void* dout, void* dq, void* dkv,
                                         int32_t dkv_dtype, int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t head_dim,
                                         float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (!q || !kv || !dout || !dq || !dkv || !workspace || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || !dt_ok(dkv_dtype) || head_dim != 64 || Nk > 256) return CAVP_ERR_UNSUPPORTED;
  if (workspace_bytes < cavp_sra_attention_bwd_workspace_bytes(B, Nq, heads)) return CAVP_ERR_WORKSPACE;
  if (((uintptr_t)q & 15) || ((uintptr_t)kv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 15) || ((uintptr_t)dkv & 15) ||
      ((uintptr_t)workspace & 15))
    return CAVP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)sra_bwd_q_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 256);
    (void)hipFuncSetAttribute((const void*)sra_bwd_q_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128);
    attr = true;
  }
  float* stats = (float*)workspace;
  const int C = heads * 64;
  const dim3 ga((Nq + 63) / 64, B * heads);
  const int chunks = (Nq + 63) / 64, kblocks = (Nk + 63) / 64;
  // query splits of the dK / dV pass: ~4 workgroups per CU (a workgroup is 4 waves walking its query chunks one after the
  // other; with 256 workgroups the stage-3 shape ran one wave per SIMD, 70 us per call); splits * B * heads <= 256 keeps the slabs
  // inside the workspace bound above
  int splits = 1024 / (B * heads * kblocks);
  splits = splits < 1 ? 1 : (splits > chunks ? chunks : splits);
  const size_t slab = (size_t)B * Nk * 2 * C;                     // floats per split
  float* slabs = stats + (((size_t)2 * B * heads * Nq + 3) & ~(size_t)3);
  const bool via_slabs = splits > 1 || dkv_dtype != CAVP_F32;   // a single split writes an f32 gradient itself
  if (via_slabs && (size_t)((slabs - stats) + splits * slab) * sizeof(float) > workspace_bytes) return CAVP_ERR_WORKSPACE;
  float* kv_out = via_slabs ? slabs : (float*)dkv;
  const dim3 gb(splits, B * heads, kblocks);
  if (dtype == CAVP_F32) {
    sra_bwd_q_kernel<float><<<ga, 256, 2 * 256 * 256, s>>>((const float*)q, (const float*)kv, (const float*)dout, (float*)dq, stats,
                                                          Nq, Nk, heads, scale);
    sra_bwd_kv_kernel<float><<<gb, 256, 0, s>>>((const float*)q, (const float*)kv, (const float*)dout, stats, kv_out, Nq, Nk, heads,
                                                scale, slab);
  } else {
    const int qpw = cavp_sra_blocks_per_wg(Nq, B * heads);
    const dim3 gq((chunks + qpw - 1) / qpw, B * heads);
    sra_bwd_q_bf16_kernel<<<gq, 256, 2 * 256 * 128, s>>>((const bf16_t*)q, (const bf16_t*)kv, (const bf16_t*)dout, (bf16_t*)dq, stats,
                                                        Nq, Nk, heads, scale, qpw);
    sra_bwd_kv_bf16_kernel<<<gb, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)kv, (const bf16_t*)dout, stats, kv_out, Nq, Nk, heads,
                                              scale, slab);
  }
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (via_slabs) {
    const long long n4 = (long long)(slab / 4);
    long long nb = (n4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (dkv_dtype == CAVP_F32) sra_bwd_kv_reduce_kernel<float><<<(int)nb, 256, 0, s>>>(slabs, (float*)dkv, n4, splits, slab);
    else sra_bwd_kv_reduce_kernel<bf16_t><<<(int)nb, 256, 0, s>>>(slabs, (bf16_t*)dkv, n4, splits, slab);
  }
  CHECK_LAUNCH();
}

extern "C" int cavp_dwconv3x3_wgrad(int32_t dtype, const void* x, const void* dy, float* dw_c133, float* dbias, int32_t N,
                                    int32_t H, int32_t W, int32_t C, void* stream) {
  return cavp_dwconv3x3_bwd(dtype, x, dy, nullptr, nullptr, dw_c133, dbias, N, H, W, C, stream);
}

extern "C" int cavp_dwconv3x3_bwd(int32_t dtype, const void* x, const void* dy, const float* w9c, void* dx, float* dw_c133,
                                  float* dbias, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !dy || !dw_c133 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || ((w9c == nullptr) != (dx == nullptr))) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || C % 8) return CAVP_ERR_UNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w9c) & 15) return CAVP_ERR_ALIGN;
  const long long total = (long long)N * H * W;
  const int gx = (C / 4 + 15) / 16;
  long long ppb = 256;
  while ((total + ppb - 1) / ppb * gx > 8192) ppb *= 2;
  // deterministic mode (cavp_set_deterministic): the pixel splits store their partials (as many splits as the scratch holds) and
  // cavp_det_finish adds them in split order.  (The first version ran ONE split - C / 64 workgroups - and made the PVTv2-B5
  // step 3.1 x slower in that mode: 98.6 vs 31.4 ms.)
  float *part_dw = nullptr, *part_db = nullptr;
  if (g_cavp_det.scratch) {
    const long long cap = (long long)(g_cavp_det.floats / ((size_t)10 * C));
    if (cap < 1) return CAVP_ERR_WORKSPACE;
    while ((total + ppb - 1) / ppb > cap) ppb *= 2;
    part_dw = g_cavp_det.scratch;
    part_db = part_dw + (size_t)((total + ppb - 1) / ppb) * 9 * C;
  }
  if (ppb > 0x7fffffffll) return CAVP_ERR_UNSUPPORTED;
  const int nsplit = (int)((total + ppb - 1) / ppb);
  const dim3 grid(gx, (unsigned)nsplit);
  hipStream_t s = (hipStream_t)stream;
#define DW_LAUNCH(T, DX) dwconv3x3_wgrad_kernel<T, DX><<<grid, 256, 0, s>>>((const T*)x, (const T*)dy, dw_c133, dbias, N, H, W, C, \
                                                                            (int)ppb, part_dw, part_db, w9c, (T*)dx)
  if (dtype == CAVP_F32) { if (dx) DW_LAUNCH(float, true); else DW_LAUNCH(float, false); }
  else { if (dx) DW_LAUNCH(bf16_t, true); else DW_LAUNCH(bf16_t, false); }
#undef DW_LAUNCH
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (part_dw) {
    if (cavp_det_finish(part_dw, nsplit, 9 * C, dw_c133, nullptr, s) != hipSuccess) return CAVP_ERR_LAUNCH;
    if (dbias && cavp_det_finish(part_db, nsplit, C, dbias, nullptr, s) != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  return CAVP_OK;
}

extern "C" int cavp_smallcin_kxk_im2col(int32_t dtype, const float* x_nchw, void* cols, int32_t N, int32_t Cin, int32_t H,
                                        int32_t W, int32_t KS, int32_t stride, int32_t pad, int32_t Kpad, void* stream) {
  if (!x_nchw || !cols || N <= 0 || H <= 0 || W <= 0 || stride <= 0 || KS <= 0 || pad < 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || Cin < 1 || Cin > 3 || Kpad % 8 || Kpad < Cin * KS * KS) return CAVP_ERR_UNSUPPORTED;
  if ((uintptr_t)cols & 15) return CAVP_ERR_ALIGN;
  const int Ho = (H + 2 * pad - KS) / stride + 1, Wo = (W + 2 * pad - KS) / stride + 1;
  const long long total = (long long)N * Ho * Wo * (Kpad / 8);
  long long nb = (total + 255) / 256;
  if (nb > 32768) nb = 32768;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    smallcin_kxk_im2col_kernel<float><<<(int)nb, 256, 0, s>>>(x_nchw, (float*)cols, N, Cin, H, W, KS, stride, pad, Ho, Wo, Kpad);
  else
    smallcin_kxk_im2col_kernel<bf16_t><<<(int)nb, 256, 0, s>>>(x_nchw, (bf16_t*)cols, N, Cin, H, W, KS, stride, pad, Ho, Wo, Kpad);
  CHECK_LAUNCH();
}

extern "C" int cavp_space_to_depth(int32_t dtype, const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C,
                                   int32_t s, int32_t inverse, void* stream) {
  if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || s <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || H % s || W % s || C % (dtype == CAVP_F32 ? 4 : 8)) return CAVP_ERR_UNSUPPORTED;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return CAVP_ERR_ALIGN;
  const long long total = (long long)B * H * W * (C / (dtype == CAVP_F32 ? 4 : 8));
  long long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    space_to_depth_kernel<float><<<(int)nb, 256, 0, st>>>((const float*)src, (float*)dst, B, H, W, C, s, inverse);
  else
    space_to_depth_kernel<bf16_t><<<(int)nb, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, B, H, W, C, s, inverse);
  CHECK_LAUNCH();
}

extern "C" int cavp_row_scale_add(int32_t dtype, const void* x, const void* branch, const float* sample_scale, void* out,
                                  int32_t B, int64_t per_sample, void* stream) {
  if (!branch || !sample_scale || !out || B <= 0 || per_sample <= 0) return CAVP_ERR_BAD_ARG;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (!dt_ok(dtype) || per_sample % VE) return CAVP_ERR_UNSUPPORTED;
  if (((uintptr_t)branch & 15) || ((uintptr_t)out & 15) || ((uintptr_t)x & 15)) return CAVP_ERR_ALIGN;
  const long long psv = per_sample / VE, total = psv * B;
  long long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    row_scale_add_kernel<float><<<(int)nb, 256, 0, st>>>((const float*)x, (const float*)branch, sample_scale, (float*)out, psv, total);
  else
    row_scale_add_kernel<bf16_t><<<(int)nb, 256, 0, st>>>((const bf16_t*)x, (const bf16_t*)branch, sample_scale, (bf16_t*)out, psv,
                                                          total);
  CHECK_LAUNCH();
}
