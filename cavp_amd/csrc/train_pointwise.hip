// Training-side bandwidth-bound kernels for gfx950: batch-statistics BatchNorm (forward statistics, finalize,
// apply, backward reduce / apply), activation backward, column sums (bias grads), LayerNorm backward, the sigmoid
// attention-gate backward, pooling / resize backward, cross-entropy forward+backward, small-Cin conv weight
// gradient and weight (un)packing for the backward GEMMs.  NHWC rows, 16-byte channel vectors, f32 accumulation,
// per-block partial reductions in LDS followed by one f32 atomic per (block, channel).
#include <stdlib.h>

#include <algorithm>
#include <cmath>

#include "common.h"

namespace {

__device__ __forceinline__ float act_grad_from_out(float y, int act) {  // d act(x)/dx expressed through y = act(x)
  switch (act) {
    case CAVP_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case CAVP_ACT_LEAKY: return y > 0.f ? 1.f : 0.01f;
    default: return 1.f;
  }
}

inline int cdiv_h(long long a, long long b) { return (int)((a + b - 1) / b); }
inline bool dt_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH

// ------------------------------------------------------------------------------------------------------------
// Column reductions over [rows][C] (C contiguous).  256 threads = 16 column groups (VE channels each) x 16 row
// lanes; grid.y = column chunks of 16*VE channels, grid.x = row chunks.  NACC accumulators per channel.
// ------------------------------------------------------------------------------------------------------------
struct ColArgs {
  const void* a;   // primary tensor (z / dy / x)
  const void* b;   // y (activation output) or nullptr
  const void* c;   // z (pre-BN conv output) or nullptr
  const float* mean;   // MODE 1: batch mean;  MODE 0: optional per-channel shift (robust variance)
  const float* rstd;
  const float* fscale;  // MODE 1 with b == nullptr: the forward's folded scale / shift, the activation mask is re-derived
  const float* fshift;  //   from z: y = act(z * scale + shift) > 0  <=>  z * scale + shift > 0 (no residual in that case)
  float* out0;
  float* out1;
  int rows, C, lda, ldb, ldc, act, rows_per_block;
  float* part;   // deterministic mode: [2][gridDim.x][C] per-workgroup partials instead of atomics
};

// MODE 0: out0 += sum a, out1 += sum a^2       (BN forward statistics)
// MODE 1: g = a * act'(b); out0 += sum g, out1 += sum g * (c - mean) * rstd   (BN backward reduce)
// MODE 2: out0 += sum a                        (bias gradient)
// CG column groups x RL = 256 / CG row lanes: 16 x 16 for C >= 128 (bf16), 8 x 32 for the 64- / 48-channel tensors of the stem and
// layer1, where half of a 16-group workgroup idled (23 us per reduce on tensors the apply kernel walks in 7)
template <typename T, int MODE, int CG>
__global__ __launch_bounds__(256) void col_reduce_kernel(const ColArgs p) {
  constexpr int VE = VecT<T>::VE;
  constexpr int RL = 256 / CG;
  __shared__ float red[2][RL][CG * VE + 1];
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int c0 = (blockIdx.y * CG + cg) * VE;
  const bool col_ok = c0 < p.C;
  float s0[VE], s1[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) s0[e] = s1[e] = 0.f;
  float mu[VE], rs[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { mu[e] = 0.f; rs[e] = 1.f; }
  float fs[VE], fh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { fs[e] = 1.f; fh[e] = 0.f; }
  const bool from_z = MODE == 1 && p.b == nullptr;   // activation mask re-derived from z (saves reading y)
  if (MODE == 1 && col_ok) {
#pragma unroll
    for (int e = 0; e < VE; ++e) { mu[e] = p.mean[c0 + e]; rs[e] = p.rstd[c0 + e]; }
    if (from_z && p.act != CAVP_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < VE; ++e) { fs[e] = p.fscale[c0 + e]; fh[e] = p.fshift[c0 + e]; }
    }
  }
  if (MODE == 0 && col_ok && p.mean) {
#pragma unroll
    for (int e = 0; e < VE; ++e) mu[e] = p.mean[c0 + e];
  }
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.rows) r_end = p.rows;
  if (col_ok) {
    long long r = r_begin + rl;
    if (MODE == 1) {
      // UR rows per trip: all 2 UR (3 UR with y) 16-byte loads are issued before the first is consumed - with one row per trip the
      // loop carried 2 loads in flight per lane and the reduce ran at 1.7 TB/s where the apply kernels reach 3.7
      constexpr int UR = 4;   // 8: 25 % slower (registers), 1: the round-1 loop
      for (; r + RL * (UR - 1) < r_end; r += RL * UR) {
        float a4[UR][VE], z4[UR][VE], y4[UR][VE];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          VecT<T>::load((const T*)p.a + (r + RL * u) * p.lda + c0, a4[u]);
          VecT<T>::load((const T*)p.c + (r + RL * u) * p.ldc + c0, z4[u]);
          if (!from_z) VecT<T>::load((const T*)p.b + (r + RL * u) * p.ldb + c0, y4[u]);
        }
#pragma unroll
        for (int u = 0; u < UR; ++u)
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const float yv = from_z ? z4[u][e] * fs[e] + fh[e] : y4[u][e];
            const float g = a4[u][e] * act_grad_from_out(yv, p.act);
            s0[e] += g;
            s1[e] += g * (z4[u][e] - mu[e]) * rs[e];
          }
      }
    }
    for (; r < r_end; r += RL) {
      float a[VE];
      VecT<T>::load((const T*)p.a + r * p.lda + c0, a);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) { const float d = a[e] - mu[e]; s0[e] += d; s1[e] += d * d; }
      } else if (MODE == 1) {
        float y[VE], z[VE];
        VecT<T>::load((const T*)p.c + r * p.ldc + c0, z);
        if (from_z) {
#pragma unroll
          for (int e = 0; e < VE; ++e) y[e] = z[e] * fs[e] + fh[e];   // same expression (and contraction) as scale_shift_act
        } else {
          VecT<T>::load((const T*)p.b + r * p.ldb + c0, y);
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float g = a[e] * act_grad_from_out(y[e], p.act);
          s0[e] += g;
          s1[e] += g * (z[e] - mu[e]) * rs[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) s0[e] += a[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    red[0][rl][cg * VE + e] = s0[e];
    red[1][rl][cg * VE + e] = s1[e];
  }
  __syncthreads();
  // CG*VE channels x (1 or 2) stats: thread t sums the RL row lanes of one (stat, channel)
  for (int i = threadIdx.x; i < 2 * CG * VE; i += 256) {
    const int st = i / (CG * VE), ch = i - st * (CG * VE);
    if (MODE == 2 && st == 1) continue;
    const int c = blockIdx.y * CG * VE + ch;
    if (c >= p.C) continue;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RL; ++k) s += red[st][k][ch];
    if (p.part)
      p.part[((size_t)st * gridDim.x + blockIdx.x) * p.C + c] = s;
    else
      atomicAdd((st == 0 ? p.out0 : p.out1) + c, s);
  }
}

// "Flat" variant for CV = C / VE vectors per row with 256 % CV == 0 (every power-of-two channel count of the ResNet):
// 256 threads = 256 / CV consecutive rows x CV vectors, i.e. one workgroup trip reads 4 KiB of CONTIGUOUS memory per
// operand and consecutive workgroups own consecutive row ranges.  The column-chunk geometry above makes every wave fetch
// four 256-byte pieces a row pitch apart and every DRAM page is visited by C / 128 different workgroups at different
// times: 3.8 TB/s read + write where the contiguous `add_kernel` reaches 6.7 TB/s on the same tensors.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void col_reduce_flat_kernel(const ColArgs p, int CV) {
  constexpr int VE = VecT<T>::VE;
  __shared__ float red[2][256 * VE];
  const int cv = threadIdx.x % CV, rsub = threadIdx.x / CV, RPB = 256 / CV;
  const int c0 = cv * VE;
  float s0[VE], s1[VE], mu[VE], rs[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { s0[e] = s1[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; }
  if (MODE == 1) {
    VecT<float>::load(p.mean + c0, mu); VecT<float>::load(p.rstd + c0, rs);
    if constexpr (VE == 8) { VecT<float>::load(p.mean + c0 + 4, mu + 4); VecT<float>::load(p.rstd + c0 + 4, rs + 4); }
  }
  if (MODE == 0 && p.mean) {
    VecT<float>::load(p.mean + c0, mu);
    if constexpr (VE == 8) VecT<float>::load(p.mean + c0 + 4, mu + 4);
  }
  const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
  long long r_end = r_begin + p.rows_per_block;
  if (r_end > p.rows) r_end = p.rows;
  for (long long r = r_begin + rsub; r < r_end; r += RPB) {
    float a[VE];
    VecT<T>::load((const T*)p.a + r * p.lda + c0, a);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < VE; ++e) { const float d = a[e] - mu[e]; s0[e] += d; s1[e] += d * d; }
    } else if (MODE == 1) {
      float y[VE], z[VE];
      VecT<T>::load((const T*)p.b + r * p.ldb + c0, y);
      VecT<T>::load((const T*)p.c + r * p.ldc + c0, z);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float g = a[e] * act_grad_from_out(y[e], p.act);
        s0[e] += g;
        s1[e] += g * (z[e] - mu[e]) * rs[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) s0[e] += a[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    red[0][threadIdx.x * VE + e] = s0[e];   // [rsub][cv][e]
    red[1][threadIdx.x * VE + e] = s1[e];
  }
  __syncthreads();
  const int CW = CV * VE;   // channels
  for (int i = threadIdx.x; i < 2 * CW; i += 256) {
    const int st = i / CW, ch = i - st * CW;
    if (MODE == 2 && st == 1) continue;
    float s = 0.f;
    for (int k = 0; k < RPB; ++k) s += red[st][k * CW + ch];
    atomicAdd((st == 0 ? p.out0 : p.out1) + ch, s);
  }
}

inline bool flat_ok(int C, int VE, const float* v0, const float* v1) {
  const int CV = C / VE;
  return CV >= 1 && CV <= 256 && 256 % CV == 0 && (((uintptr_t)v0 | (uintptr_t)v1) & 15) == 0;
}
// rows per workgroup: a multiple of 256 / CV covering about `target_bytes` of one operand
inline int flat_rows_per_block(long long rows, int C, int es, int CV, long long target_bytes, int max_blocks) {
  const int RPB = 256 / CV;
  long long rpb = (target_bytes / ((long long)C * es) + RPB - 1) / RPB * RPB;
  if (rpb < RPB) rpb = RPB;
  while ((rows + rpb - 1) / rpb > max_blocks) rpb *= 2;
  return (int)rpb;
}

template <int MODE>
int launch_col_reduce(int dtype, ColArgs& a, hipStream_t s) {
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  // (measured: the contiguous geometry helps the read + write kernels, 1.69 -> 1.23 ms per step for scale_shift_act,
  // but not the reductions - 1.59 -> 1.85 ms for the BN backward reduce - so it stays behind CAVP_FLAT_REDUCE=1)
  static const bool flat_reduce = cavp_knob_str("CAVP_FLAT_REDUCE") != nullptr;
  a.part = nullptr;
  if (flat_reduce && !g_cavp_det.scratch && (MODE != 1 || a.b != nullptr) && flat_ok(a.C, VE, a.mean, a.rstd)) {
    const int CV = a.C / VE;
    a.rows_per_block = flat_rows_per_block(a.rows, a.C, 16 / VE, CV, 32 << 10, 4096);   // <= 4096 atomics per channel
    const int gx = (int)((a.rows + a.rows_per_block - 1) / a.rows_per_block);
    if (dtype == CAVP_F32)
      col_reduce_flat_kernel<float, MODE><<<gx, 256, 0, s>>>(a, CV);
    else
      col_reduce_flat_kernel<bf16_t, MODE><<<gx, 256, 0, s>>>(a, CV);
    CHECK_LAUNCH();
  }
  const int CG = a.C <= 8 * VE ? 8 : 16;   // column groups per workgroup
  const int gy = cdiv_h(a.C, CG * VE);
  // workgroups per launch: every one ends with 2 x 128 device-scope f32 atomics, which are served memory-side (the L2s of
  // the 8 XCDs are not coherent): at 2048 workgroups the atomics were 40 % of the BN backward reduce (1.60 ms per step,
  // 0.95 without them); 512 keeps enough loads in flight and costs 1.27 ms (sweep: 2048 / 1024 / 512 / 256 -> 1.61 / 1.31 /
  // 1.27 / 1.58 ms; end-of-round whole-step sweep 256 / 384 / 512 / 768 / 1024 -> 19.20 / 18.91 / 18.80 / 18.75 / 18.80 ms).
  // CAVP_REDUCE_BLOCKS overrides for A/B runs.
  static const int target_blocks = cavp_knob_int("CAVP_REDUCE_BLOCKS", 768);
  int gx = target_blocks / gy;
  if (gx < 1) gx = 1;
  int rpb = cdiv_h(a.rows, gx);
  if (rpb < 64) rpb = 64;
  rpb = (rpb + 31) / 32 * 32;
  gx = cdiv_h(a.rows, rpb);
  a.rows_per_block = rpb;
  bool det_err;
  a.part = cavp_det_scratch(gx, a.C, &det_err);
  if (det_err) return CAVP_ERR_WORKSPACE;
  if (dtype == CAVP_F32) {
    if (CG == 8) col_reduce_kernel<float, MODE, 8><<<dim3(gx, gy), 256, 0, s>>>(a);
    else col_reduce_kernel<float, MODE, 16><<<dim3(gx, gy), 256, 0, s>>>(a);
  } else {
    if (CG == 8) col_reduce_kernel<bf16_t, MODE, 8><<<dim3(gx, gy), 256, 0, s>>>(a);
    else col_reduce_kernel<bf16_t, MODE, 16><<<dim3(gx, gy), 256, 0, s>>>(a);
  }
  if (a.part) {
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
    return cavp_det_finish(a.part, gx, a.C, a.out0, MODE == 2 ? nullptr : a.out1, s) == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
  }
  CHECK_LAUNCH();
}

__global__ void scale_vec_kernel(const float* in, float alpha, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = alpha * in[i];
}

__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, const float* sh, float inv_count, float unbias,
                                   const float* gamma, const float* beta, float eps, float momentum, float* rmean,
                                   float* rvar, float* scale, float* shift, float* mean_out, float* rstd_out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float dm = sum[c] * inv_count;       // mean of (x - shift)
  const float m = dm + (sh ? sh[c] : 0.f);
  float var = sumsq[c] * inv_count - dm * dm;  // biased (normalisation) variance; exact when shift == mean
  if (var < 0.f) var = 0.f;
  const float rstd = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - m * sc;
  mean_out[c] = m;
  rstd_out[c] = rstd;
  if (rmean) {  // nn.BatchNorm2d running statistics: momentum update with the unbiased variance
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * unbias;
  }
}

// Row-loop geometry shared by the per-channel elementwise kernels: 256 threads = 16 column groups (VE channels each,
// 256 contiguous bytes per row) x 16 row lanes; grid.y = column chunks, grid.x = row chunks.  Each thread keeps its
// per-channel coefficients in registers for the whole row loop (the first version re-loaded 5 scalars per channel per
// element and ran at 0.9 TB/s).
struct RowLoop {
  long long rows;
  int C, rows_per_block;
};

// Combine the per-tile (mean, M2) pairs written by the conv epilogue into the batch statistics of each channel (Chan et
// al. parallel variance), then the same outputs as bn_finalize.  One wave per channel: lanes stride over the tiles.
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb, dlt = mb - mean;
  mean += dlt * (nb / nt);
  m2 += m2b + dlt * dlt * (n * nb / nt);
  n = nt;
}
__global__ __launch_bounds__(256) void bn_finalize_tiles_kernel(const float* __restrict__ ts, int tiles, int rows_per_tile,
                                                                long long M, const float* gamma, const float* beta,
                                                                float eps, float momentum, float* rmean, float* rvar,
                                                                float* scale, float* shift, float* mean_out,
                                                                float* rstd_out, int C, float* moments) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  // 8 tiles per trip with their loads issued together (same combine order as a plain loop: bit-identical): a lane's
  // <= 49 scattered 8-byte loads were otherwise serialised behind the divisions of the combine
  for (int t0 = lane; t0 < tiles; t0 += 64 * 8) {
    float2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + 64 * u;
      q[u] = t < tiles ? *(const float2*)(ts + ((size_t)t * C + c) * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + 64 * u;
      if (t < tiles) {
        long long nb = M - (long long)t * rows_per_tile;
        if (nb > rows_per_tile) nb = rows_per_tile;
        chan_combine(n, mean, m2, (float)nb, q[u].x, q[u].y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), m2b = __shfl_xor(m2, o, 64);
    // symmetric combine so every lane ends with the same value
    const float nt = n + nb;
    if (nt > 0.f) {
      const float dlt = mb - mean;
      const float nm = mean + dlt * (nb / nt);
      m2 = m2 + m2b + dlt * dlt * (n * nb / nt);
      mean = nm;
      n = nt;
    }
  }
  if (lane == 0 && moments) {   // SyncBatchNorm: this rank's (mean, M2), combined across ranks by a second call
    moments[2 * c] = mean;
    moments[2 * c + 1] = m2;
  } else if (lane == 0) {
    const float var = m2 / (float)M;
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (rmean) {
      const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * unbias;
    }
  }
}

// Same combine with one 256-thread workgroup per channel, for launches with >= 1024 tiles (the 112x112 and the 2B x 56x56
// layers: 1568 / 3136 tiles): a lane of the one-wave version walks up to 49 scattered loads in 7 dependent batches.
__global__ __launch_bounds__(256) void bn_finalize_tiles_wide_kernel(const float* __restrict__ ts, int tiles, int rows_per_tile,
                                                                     long long M, const float* gamma, const float* beta,
                                                                     float eps, float momentum, float* rmean, float* rvar,
                                                                     float* scale, float* shift, float* mean_out,
                                                                     float* rstd_out, int C, float* moments) {
  __shared__ float part[4][3];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int t0 = threadIdx.x; t0 < tiles; t0 += 256 * 8) {
    float2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + 256 * u;
      q[u] = t < tiles ? *(const float2*)(ts + ((size_t)t * C + c) * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + 256 * u;
      if (t < tiles) {
        long long nb = M - (long long)t * rows_per_tile;
        if (nb > rows_per_tile) nb = rows_per_tile;
        chan_combine(n, mean, m2, (float)nb, q[u].x, q[u].y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), m2b = __shfl_xor(m2, o, 64);
    const float nt = n + nb;
    if (nt > 0.f) {
      const float dlt = mb - mean;
      const float nm = mean + dlt * (nb / nt);
      m2 = m2 + m2b + dlt * dlt * (n * nb / nt);
      mean = nm;
      n = nt;
    }
  }
  if (lane == 0) { part[wv][0] = n; part[wv][1] = mean; part[wv][2] = m2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    n = part[0][0]; mean = part[0][1]; m2 = part[0][2];
    for (int w = 1; w < 4; ++w)
      if (part[w][0] > 0.f) chan_combine(n, mean, m2, part[w][0], part[w][1], part[w][2]);
    if (moments) {
      moments[2 * c] = mean;
      moments[2 * c + 1] = m2;
      return;
    }
    const float var = m2 / (float)M;
    const float rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (rmean) {
      const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * unbias;
    }
  }
}

// y = act(x * scale + shift + residual)
template <typename T>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              T* __restrict__ y, RowLoop g, int ldx, int ldr, int ldy,
                                                              int act) {
  constexpr int VE = VecT<T>::VE;
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (blockIdx.y * 16 + cg) * VE;
  if (c >= g.C) return;
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { sc[e] = scale ? scale[c + e] : 1.f; sh[e] = shift ? shift[c + e] : 0.f; }
  const long long r0 = (long long)blockIdx.x * g.rows_per_block;
  long long r1 = r0 + g.rows_per_block;
  if (r1 > g.rows) r1 = g.rows;
  for (long long r = r0 + rl; r < r1; r += 16) {
    float v[VE];
    VecT<T>::load(x + r * ldx + c, v);
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) {
      float rr[VE];
      VecT<T>::load(res + r * ldr + c, rr);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += rr[e];
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = apply_act(v[e], act);
    VecT<T>::store(y + r * ldy + c, v);
  }
}

// flat variant (see col_reduce_flat_kernel): CV vectors per row, 256 % CV == 0
template <typename T>
__global__ __launch_bounds__(256) void scale_shift_act_flat_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const T* __restrict__ res,
                                                                   T* __restrict__ y, long long rows, int rows_per_block,
                                                                   int CV, int ldx, int ldr, int ldy, int act) {
  constexpr int VE = VecT<T>::VE;
  const int cv = threadIdx.x % CV, rsub = threadIdx.x / CV, RPB = 256 / CV;
  const int c = cv * VE;
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
  if (scale) { VecT<float>::load(scale + c, sc); if constexpr (VE == 8) VecT<float>::load(scale + c + 4, sc + 4); }
  if (shift) { VecT<float>::load(shift + c, sh); if constexpr (VE == 8) VecT<float>::load(shift + c + 4, sh + 4); }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long long r = r0 + rsub; r < r1; r += RPB) {
    float v[VE];
    VecT<T>::load(x + r * ldx + c, v);
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) {
      float rr[VE];
      VecT<T>::load(res + r * ldr + c, rr);
#pragma unroll
      for (int e = 0; e < VE; ++e) v[e] += rr[e];
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = apply_act(v[e], act);
    VecT<T>::store(y + r * ldy + c, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_flat_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                                const T* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ sum_g,
                                                                const float* __restrict__ sum_gz, float inv_m,
                                                                T* __restrict__ dz, T* __restrict__ g_out, long long rows,
                                                                int rows_per_block, int CV, int ld_dy, int ld_y, int ld_z,
                                                                int ld_dz, int ld_g, int act, const float* __restrict__ fscale,
                                                                const float* __restrict__ fshift, float* acc_g, float* acc_gz) {
  constexpr int VE = VecT<T>::VE;
  const int cv = threadIdx.x % CV, rsub = threadIdx.x / CV, RPB = 256 / CV;
  const int c = cv * VE;
  // per-channel coefficients with 16-byte loads (C % VE == 0 and 16-byte aligned vectors: flat_ok): the first version issued
  // 7 x VE scalar loads per thread in front of a row loop of only four trips
  float ca[VE], cb[VE], cc[VE];
  {
    float rs[VE], gm[VE], sg[VE], sgz[VE], mu[VE];
#pragma unroll
    for (int q = 0; q < VE; q += 4) {
      VecT<float>::load(rstd + c + q, rs + q); VecT<float>::load(gamma + c + q, gm + q); VecT<float>::load(sum_g + c + q, sg + q);
      VecT<float>::load(sum_gz + c + q, sgz + q); VecT<float>::load(mean + c + q, mu + q);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float a = gm[e] * rs[e];
      ca[e] = a;
      cb[e] = -a * rs[e] * sgz[e] * inv_m;
      cc[e] = -a * sg[e] * inv_m - cb[e] * mu[e];
    }
    // the sums ARE the affine gradients (dbeta = sum g, dgamma = sum g * zhat): when they arrive in scratch (summed by the
    // data-gradient launch's epilogue, cavp_conv2d_nhwc_bnbwd) one thread per channel vector adds them to the parameter gradients
    if (acc_g && blockIdx.x == 0 && rsub == 0) {
#pragma unroll
      for (int e = 0; e < VE; ++e) { acc_g[c + e] += sg[e]; acc_gz[c + e] += sgz[e]; }
    }
  }
  const bool from_z = y == nullptr;   // activation mask re-derived from z (layers without a residual): y is not read
  float fs[VE], fh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { fs[e] = 1.f; fh[e] = 0.f; }
  if (from_z && act != CAVP_ACT_NONE) {
#pragma unroll
    for (int q = 0; q < VE; q += 4) { VecT<float>::load(fscale + c + q, fs + q); VecT<float>::load(fshift + c + q, fh + q); }
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long long r = r0 + rsub; r < r1; r += RPB) {
    float a[VE], yy[VE], zz[VE], o[VE], g[VE];
    VecT<T>::load(dy + r * ld_dy + c, a);
    VecT<T>::load(z + r * ld_z + c, zz);
    if (from_z) {
#pragma unroll
      for (int e = 0; e < VE; ++e) yy[e] = zz[e] * fs[e] + fh[e];
    } else {
      VecT<T>::load(y + r * ld_y + c, yy);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      g[e] = a[e] * act_grad_from_out(yy[e], act);
      o[e] = ca[e] * g[e] + cb[e] * zz[e] + cc[e];
    }
    VecT<T>::store(dz + r * ld_dz + c, o);
    if (g_out) VecT<T>::store(g_out + r * ld_g + c, g);
  }
}

// dz = gamma * rstd * (g - sum_g / M - zhat * sum_gz / M) = a_c * g + b_c * z + c_c,  g = dy * act'(y);
// optionally also writes g (skip-path gradient)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_g,
                                                           const float* __restrict__ sum_gz, float inv_m, T* __restrict__ dz,
                                                           T* __restrict__ g_out, RowLoop gm, int ld_dy, int ld_y,
                                                           int ld_z, int ld_dz, int ld_g, int act, const float* __restrict__ fscale,
                                                                const float* __restrict__ fshift, float* acc_g, float* acc_gz) {
  constexpr int VE = VecT<T>::VE;
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (blockIdx.y * 16 + cg) * VE;
  if (c >= gm.C) return;
  float ca[VE], cb[VE], cc[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    const float rs = rstd[c + e], a = gamma[c + e] * rs;
    ca[e] = a;
    cb[e] = -a * rs * sum_gz[c + e] * inv_m;
    cc[e] = -a * sum_g[c + e] * inv_m - cb[e] * mean[c + e];
    if (acc_g && blockIdx.x == 0 && rl == 0) { acc_g[c + e] += sum_g[c + e]; acc_gz[c + e] += sum_gz[c + e]; }   // (see the flat kernel)
  }
  const bool from_z = y == nullptr;   // activation mask re-derived from z (layers without a residual): y is not read
  float fs[VE], fh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    fs[e] = (from_z && act != CAVP_ACT_NONE) ? fscale[c + e] : 1.f;
    fh[e] = (from_z && act != CAVP_ACT_NONE) ? fshift[c + e] : 0.f;
  }
  const long long r0 = (long long)blockIdx.x * gm.rows_per_block;
  long long r1 = r0 + gm.rows_per_block;
  if (r1 > gm.rows) r1 = gm.rows;
  for (long long r = r0 + rl; r < r1; r += 16) {
    float a[VE], yy[VE], zz[VE], o[VE], g[VE];
    VecT<T>::load(dy + r * ld_dy + c, a);
    VecT<T>::load(z + r * ld_z + c, zz);
    if (from_z) {
#pragma unroll
      for (int e = 0; e < VE; ++e) yy[e] = zz[e] * fs[e] + fh[e];
    } else {
      VecT<T>::load(y + r * ld_y + c, yy);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      g[e] = a[e] * act_grad_from_out(yy[e], act);
      o[e] = ca[e] * g[e] + cb[e] * zz[e] + cc[e];
    }
    VecT<T>::store(dz + r * ld_dz + c, o);
    if (g_out) VecT<T>::store(g_out + r * ld_g + c, g);
  }
}

inline RowLoop row_loop_geometry(long long rows, int C, int VE, dim3& grid) {
  const int gy = cdiv_h(C, 16 * VE);
  int gx = 4096 / gy;
  if (gx < 1) gx = 1;
  long long rpb = (rows + gx - 1) / gx;
  if (rpb < 32) rpb = 32;
  rpb = (rpb + 15) / 16 * 16;
  gx = (int)((rows + rpb - 1) / rpb);
  grid = dim3(gx, gy);
  RowLoop g;
  g.rows = rows; g.C = C; g.rows_per_block = (int)rpb;
  return g;
}

// dx = dy * act'(.)   relu / leaky: from the output y;  gelu: from the pre-activation x
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ ref,
                                                      T* __restrict__ dx, long long rows, int C, int ld_dy, int ld_ref,
                                                      int ld_dx, int act) {
  constexpr int VE = VecT<T>::VE;
  const int CV = C / VE;
  const long long total = rows * CV;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / CV;
    const int c = (int)(i - r * CV) * VE;
    float a[VE], b[VE];
    VecT<T>::load(dy + r * ld_dy + c, a);
    VecT<T>::load(ref + r * ld_ref + c, b);
#pragma unroll
    for (int e = 0; e < VE; ++e) a[e] *= (act == CAVP_ACT_GELU ? gelu_grad(b[e]) : act_grad_from_out(b[e], act));
    VecT<T>::store(dx + r * ld_dx + c, a);
  }
}

// out = a + b  (gradient accumulation of two same-shaped dense tensors)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o,
                                                  long long n) {
  constexpr int VE = VecT<T>::VE;
  for (long long i = (blockIdx.x * 256ll + threadIdx.x) * VE; i < n; i += (long long)gridDim.x * 256 * VE) {
    float x[VE], y[VE];
    VecT<T>::load(a + i, x);
    VecT<T>::load(b + i, y);
#pragma unroll
    for (int e = 0; e < VE; ++e) x[e] += y[e];
    VecT<T>::store(o + i, x);
  }
}

// ------------------------------------------------------------------------------------------------------------
// max pool backward (gather form, deterministic): dx[p] = sum of dy over the windows whose recorded arg-max is p.
// The forward stores the window-relative arg-max (one byte per output element), so the backward reads dy and the
// index only - the first version re-derived every window's winner from x: 36 vector loads per input element for the
// 3x3 / stride-2 stem pool (500 us for a 232 MB problem).
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ argmax, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int k,
                                                          int stride, int pad, int Ho, int Wo) {
  constexpr int VE = VecT<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)N * H * W * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int cv, wi, hi, n;
    long long pix;
    split_index(idx, CV, W, H, cv, pix, wi, hi, n);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    // windows (ho, wo) with ho*stride - pad <= hi < ho*stride - pad + k
    int ho_lo = (hi + pad - k + stride) / stride;  // ceil((hi + pad - k + 1) / stride) for non-negative numerators
    if (hi + pad - k + 1 <= 0) ho_lo = 0;
    int ho_hi = (hi + pad) / stride;
    if (ho_hi > Ho - 1) ho_hi = Ho - 1;
    int wo_lo = (wi + pad - k + stride) / stride;
    if (wi + pad - k + 1 <= 0) wo_lo = 0;
    int wo_hi = (wi + pad) / stride;
    if (wo_hi > Wo - 1) wo_hi = Wo - 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const unsigned code = (unsigned)((hi - (ho * stride - pad)) * k + (wi - (wo * stride - pad)));
        const size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + cv * VE;
        unsigned am[VE];
        if constexpr (VE == 8) {
          const uint2 t = *(const uint2*)(argmax + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) { am[e] = (t.x >> (8 * e)) & 255u; am[4 + e] = (t.y >> (8 * e)) & 255u; }
        } else {
          const unsigned t = *(const unsigned*)(argmax + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) am[e] = (t >> (8 * e)) & 255u;
        }
        float g[VE];
        VecT<T>::load(dy + o, g);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += am[e] == code ? g[e] : 0.f;
      }
    }
    VecT<T>::store(dx + (size_t)pix * C + cv * VE, acc);
  }
}

// The same gather for stride 2 (every pool of the model), one grid row per input row: (n, hi) come from blockIdx.y and the window bounds are
// shifts.  The flat version above pays three 32-bit divisions (channel vector / W / H split) + four by the stride per 16-byte vector:
// the stem pool's backward ran 56 us on a 141 MB problem.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_s2_rows_kernel(const unsigned char* __restrict__ argmax, const T* __restrict__ dy,
                                                                  T* __restrict__ dx, int H, int W, int C, int k, int pad, int Ho, int Wo,
                                                                  int cv_shift) {
  constexpr int VE = VecT<T>::VE;
  const int CV = C / VE;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W * CV) return;
  const int wi = cv_shift >= 0 ? t >> cv_shift : t / CV;
  const int cv = t - wi * CV;
  const int row = blockIdx.y;              // n * H + hi
  const int n = row / H, hi = row - n * H;
  float acc[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) acc[e] = 0.f;
  const int ho_lo = hi + pad - k + 1 <= 0 ? 0 : (hi + pad - k + 2) >> 1;
  const int ho_hi = min((hi + pad) >> 1, Ho - 1);
  const int wo_lo = wi + pad - k + 1 <= 0 ? 0 : (wi + pad - k + 2) >> 1;
  const int wo_hi = min((wi + pad) >> 1, Wo - 1);
  for (int ho = ho_lo; ho <= ho_hi; ++ho) {
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      const unsigned code = (unsigned)((hi - (ho * 2 - pad)) * k + (wi - (wo * 2 - pad)));
      const size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * C + cv * VE;
      unsigned am[VE];
      if constexpr (VE == 8) {
        const uint2 q = *(const uint2*)(argmax + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { am[e] = (q.x >> (8 * e)) & 255u; am[4 + e] = (q.y >> (8 * e)) & 255u; }
      } else {
        const unsigned q = *(const unsigned*)(argmax + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) am[e] = (q >> (8 * e)) & 255u;
      }
      float g[VE];
      VecT<T>::load(dy + o, g);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += am[e] == code ? g[e] : 0.f;
    }
  }
  VecT<T>::store(dx + ((size_t)row * W + wi) * C + cv * VE, acc);
}

// ------------------------------------------------------------------------------------------------------------
// bilinear backward (gather form): dx[hi, wi] = sum_{ho, wo} wh(ho -> hi) * ww(wo -> wi) * dy[ho, wo]
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index_t(int dst, int in, int out, int align, int& i0, int& i1, float& lam) {
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
// range of destination indices whose two-tap source footprint can include `src_i`: src(dst) in (src_i - 1, src_i + 1)
// (the ends are exclusive, so floor / ceil of the bounds stay safe under float rounding; clamped sources at the borders
// fall inside because the range is clipped to [0, out - 1])
__device__ __forceinline__ void dst_range(int src_i, int in, int out, int align, int& lo, int& hi) {
  float xlo, xhi;
  if (align) {
    if (in > 1 && out > 1) {
      const float r = (float)(out - 1) / (float)(in - 1);
      xlo = ((float)src_i - 1.f) * r;
      xhi = ((float)src_i + 1.f) * r;
    } else {
      xlo = 0.f;
      xhi = (float)(out - 1);
    }
  } else {
    const float r = (float)out / (float)in;
    xlo = ((float)src_i - 0.5f) * r - 0.5f;
    xhi = ((float)src_i + 1.5f) * r - 0.5f;
  }
  lo = (int)floorf(xlo);
  hi = (int)ceilf(xhi);
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float tap_weight(int dst, int src_i, int in, int out, int align) {
  int i0, i1;
  float lam;
  src_index_t(dst, in, out, align, i0, i1, lam);
  float w = 0.f;
  if (i0 == src_i) w += 1.f - lam;
  if (i1 == src_i) w += lam;
  return w;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_nhwc_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N,
                                                                int Hi, int Wi, int C, int ld_dx, int Ho, int Wo,
                                                                int ld_dy, int align) {
  constexpr int VE = VecT<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)N * Hi * Wi * CV;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int cv, wi, hi, n;
    long long pix;
    split_index(idx, CV, Wi, Hi, cv, pix, wi, hi, n);
    int hlo, hhi, wlo, whi;
    dst_range(hi, Hi, Ho, align, hlo, hhi);
    dst_range(wi, Wi, Wo, align, wlo, whi);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    // Footprints of <= NWM columns (every up-sampling ratio <= 4: the decoder's 14 -> 56): the column weights are computed ONCE per
    // thread instead of once per footprint row (tap_weight is ~30 float / integer instructions; 8 x 8 footprints spent more time there
    // than on their 64 loads: 49 us for a 55 MB problem), and a row's loads are issued together.
    constexpr int NWM = 12;
    const int nw = whi - wlo + 1;
    if (nw <= NWM) {
      float wwv[NWM];
      int wof[NWM];   // column offsets, clamped into the footprint: every load below is unconditional (weight 0 beyond the footprint)
#pragma unroll
      for (int j = 0; j < NWM; ++j) {
        wwv[j] = j < nw ? tap_weight(wlo + j, wi, Wi, Wo, align) : 0.f;
        wof[j] = (j < nw ? j : nw - 1) * ld_dy;
      }
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(ho, hi, Hi, Ho, align);
        if (wh == 0.f) continue;
        const T* row = dy + ((size_t)(n * Ho + ho) * Wo + wlo) * ld_dy + cv * VE;
        uint4 graw[NWM];   // (raw 16-byte vectors: unpacked at use, 48 instead of 96 registers for the row)
#pragma unroll
        for (int j = 0; j < NWM; ++j) graw[j] = VecT<T>::load_raw(row + wof[j]);
        float r[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) r[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NWM; ++j) {
          float g[VE];
          VecT<T>::unpack(graw[j], g);
#pragma unroll
          for (int e = 0; e < VE; ++e) r[e] += wwv[j] * g[e];
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += wh * r[e];
      }
    } else {
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(ho, hi, Hi, Ho, align);
        if (wh == 0.f) continue;
        for (int wo = wlo; wo <= whi; ++wo) {
          const float ww = tap_weight(wo, wi, Wi, Wo, align);
          if (ww == 0.f) continue;
          float g[VE];
          VecT<T>::load(dy + ((size_t)(n * Ho + ho) * Wo + wo) * ld_dy + cv * VE, g);
#pragma unroll
          for (int e = 0; e < VE; ++e) acc[e] += wh * ww * g[e];
        }
      }
    }
    VecT<T>::store(dx + (size_t)pix * ld_dx + cv * VE, acc);
  }
}

// dy: NCHW f32 [N][C][Ho][Wo] (only the first n_valid images carry gradient), dx: NHWC (T) [N][Hi][Wi][C]
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_from_nchw_kernel(const float* __restrict__ dy, T* __restrict__ dx,
                                                                     int N, int n_valid, int Hi, int Wi, int C,
                                                                     int ld_dx, int Ho, int Wo, int align) {
  const long long total = (long long)N * Hi * Wi * C;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int c, wi, hi, n;
    long long pix;
    split_index(idx, C, Wi, Hi, c, pix, wi, hi, n);
    float acc = 0.f;
    if (n < n_valid) {
      int hlo, hhi, wlo, whi;
      dst_range(hi, Hi, Ho, align, hlo, hhi);
      dst_range(wi, Wi, Wo, align, wlo, whi);
      const float* base = dy + ((size_t)n * C + c) * Ho * Wo;
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(ho, hi, Hi, Ho, align);
        if (wh == 0.f) continue;
        for (int wo = wlo; wo <= whi; ++wo) {
          const float ww = tap_weight(wo, wi, Wi, Wo, align);
          if (ww != 0.f) acc += wh * ww * base[(size_t)ho * Wo + wo];
        }
      }
    }
    Elem<T>::st(dx + (size_t)pix * ld_dx + c, acc);
  }
}

// x[n, p, c] += v[n, c] * alpha   (global-average-pool backward: alpha = 1 / HW)
template <typename T>
__global__ __launch_bounds__(256) void bcast_add_kernel(T* __restrict__ x, const float* __restrict__ v, float alpha,
                                                        int N, int HW, int C, int ld) {
  constexpr int VE = VecT<T>::VE;
  const int CV = C / VE;
  const long long total = (long long)N * HW * CV;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / CV;
    const int c = (int)(i - r * CV) * VE;
    const int n = (int)(r / HW);
    float a[VE];
    VecT<T>::load(x + r * ld + c, a);
#pragma unroll
    for (int e = 0; e < VE; ++e) a[e] += alpha * v[(size_t)n * C + c + e];
    VecT<T>::store(x + r * ld + c, a);
  }
}

// ------------------------------------------------------------------------------------------------------------
// cross entropy (ignore_index) on NCHW f32 logits: pass 1 = per-pixel loss + valid count; pass 2 = gradient
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                     int n_img, int C, long long HW, int ignore, float* __restrict__ part) {
  // one (loss, count) partial per workgroup, no atomics: 32 K same-address float atomics (two per wave of a 4096-block
  // grid) serialised at ~12 ns each and made this 25 MB pass take 420 us
  __shared__ float red[8];
  const long long total = (long long)n_img * HW;
  float loss = 0.f, cnt = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long lb = label[i];
    if (lb == ignore) continue;
    if ((unsigned long long)lb >= (unsigned long long)C) { loss = NAN; continue; }   // torch device-asserts; here the loss is poisoned (no out-of-bounds read)
    const int n = (int)(i / HW);
    const float* p = logits + (size_t)n * C * HW + (i - (long long)n * HW);
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[(size_t)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[(size_t)c * HW] - m);
    loss += (m + logf(s)) - p[(size_t)lb * HW];
    cnt += 1.f;
  }
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = loss; red[4 + wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 + 2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    part[3 + 2 * blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                     int n_img, int n_total, int C, long long HW, int ignore,
                                                     const float* __restrict__ acc, float gscale,
                                                     float* __restrict__ dlogits) {
  const long long total = (long long)n_total * HW;
  const float inv = acc[1] > 0.f ? gscale / acc[1] : 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / HW);
    const long long off = (size_t)n * C * HW + (i - (long long)n * HW);
    float* d = dlogits + off;
    const long long lb = n < n_img ? label[i] : (long long)ignore;
    if (lb == ignore || (unsigned long long)lb >= (unsigned long long)C) {
      for (int c = 0; c < C; ++c) d[(size_t)c * HW] = 0.f;
      continue;
    }
    const float* p = logits + off;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, p[(size_t)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[(size_t)c * HW] - m);
    const float is = 1.f / s;
    for (int c = 0; c < C; ++c) {
      const float sm = expf(p[(size_t)c * HW] - m) * is;
      d[(size_t)c * HW] = (sm - (c == lb ? 1.f : 0.f)) * inv;
    }
  }
}
__global__ __launch_bounds__(256) void ce_finish_kernel(float* acc, int nparts, float* loss) {
  __shared__ float red[8];
  float l = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) { l += acc[2 + 2 * i]; c += acc[3 + 2 * i]; }
  l = wave_sum(l);
  c = wave_sum(c);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = l; red[4 + wv] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float lt = (red[0] + red[1]) + (red[2] + red[3]), ct = (red[4] + red[5]) + (red[6] + red[7]);
    acc[0] = lt;
    acc[1] = ct;
    loss[0] = lt / ct;   // every label ignored: 0 / 0 = NaN, as torch's CrossEntropyLoss(reduction="mean")
  }
}

// ------------------------------------------------------------------------------------------------------------
// fused segmentation head of the training step: bilinear upsample (low-res NHWC logits -> label resolution) +
// cross entropy + gradient w.r.t. the low-res logits, without the [n][C][H][W] f32 prediction or its gradient in HBM
// (224x224, B = 32: 2 x 25.7 MB written and read back; 71 classes at 512x512: 2 x 1.2 GB).
//   pass 1 (per label pixel): interpolate the C logits, online log-sum-exp -> lse[pixel], loss / count partials
//   pass 2 (per low-res logit): gather form of the upsample backward over the <= (2 * ratio)^2 label pixels whose
//           footprint holds it, recomputing that pixel's class probability exp(logit_c - lse)
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float bilerp(const T* __restrict__ img, int W, int ld, int c, int h0, int h1, int w0,
                                        int w1, float lh, float lw) {
  const float a = Elem<T>::ld(img + ((size_t)h0 * W + w0) * ld + c), b = Elem<T>::ld(img + ((size_t)h0 * W + w1) * ld + c);
  const float cc = Elem<T>::ld(img + ((size_t)h1 * W + w0) * ld + c), d = Elem<T>::ld(img + ((size_t)h1 * W + w1) * ld + c);
  return (1.f - lh) * (1.f - lw) * a + (1.f - lh) * lw * b + lh * (1.f - lw) * cc + lh * lw * d;
}
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ lo, const long long* __restrict__ label,
                                                       int n_img, int C, int Hi, int Wi, int ld, int Ho, int Wo,
                                                       int align, int ignore, float* __restrict__ lse,
                                                       float* __restrict__ part) {
  __shared__ float red[8];
  const long long HW = (long long)Ho * Wo, total = (long long)n_img * HW;
  float loss = 0.f, cnt = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long lb = label[i];
    if (lb == ignore) { lse[i] = 0.f; continue; }
    if ((unsigned long long)lb >= (unsigned long long)C) { lse[i] = 0.f; loss = NAN; continue; }   // out-of-range label: poisoned loss
    const int wo = (int)(i % Wo);
    const int ho = (int)((i / Wo) % Ho);
    const int n = (int)(i / HW);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index_t(ho, Hi, Ho, align, h0, h1, lh);
    src_index_t(wo, Wi, Wo, align, w0, w1, lw);
    const T* img = lo + (size_t)n * Hi * Wi * ld;
    float m = -INFINITY, s = 0.f, picked = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = bilerp<T>(img, Wi, ld, c, h0, h1, w0, w1, lh, lw);
      if (c == lb) picked = v;
      const float mn = fmaxf(m, v);
      s = s * expf(m - mn) + expf(v - mn);
      m = mn;
    }
    const float l = m + logf(s);
    lse[i] = l;
    loss += l - picked;
    cnt += 1.f;
  }
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = loss; red[4 + wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 + 2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    part[3 + 2 * blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}
// pass 2, tiled: one workgroup = (th x tw tile of low-res pixels, image, class).  Step 1 stages the class-c residual
// r[ho][wo] = softmax_c - onehot_c (0 where ignored) of the label-resolution region under the tile in LDS - every
// label pixel is evaluated once per class instead of once per low-res pixel it touches (4x at ratio 4), with
// independent loads; step 2 gathers each low-res logit's footprint from LDS.  (A thread-per-logit global gather was
// bound by its chain of dependent loads: 71 us for B = 32, two classes; 8 lanes per logit: 114 us.)
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ lo, const long long* __restrict__ label,
                                                       const float* __restrict__ lse, const float* __restrict__ acc,
                                                       float gscale, int n_img, int C, int Hi, int Wi, int ld, int Ho,
                                                       int Wo, int align, int ignore, int th, int tw, int RW,
                                                       int tiles_w, T* __restrict__ dlo) {
  extern __shared__ float resid[];   // [RH][RW]
  const int c = blockIdx.y % C, n = blockIdx.y / C;
  const int ty0 = (blockIdx.x / tiles_w) * th, tx0 = (blockIdx.x % tiles_w) * tw;
  const int ty1 = min(ty0 + th, Hi) - 1, tx1 = min(tx0 + tw, Wi) - 1;
  int rh0, rh1, rw0, rw1, t0, t1;
  dst_range(ty0, Hi, Ho, align, rh0, t1);
  dst_range(ty1, Hi, Ho, align, t0, rh1);
  dst_range(tx0, Wi, Wo, align, rw0, t1);
  dst_range(tx1, Wi, Wo, align, t0, rw1);
  const int rw = rw1 - rw0 + 1, cnt = (rh1 - rh0 + 1) * rw;
  const T* img = lo + (size_t)n * Hi * Wi * ld;
#pragma unroll 4
  for (int e = threadIdx.x; e < cnt; e += 256) {
    const int ho = rh0 + e / rw, wo = rw0 + e % rw;
    const size_t at = ((size_t)n * Ho + ho) * Wo + wo;
    const long long lb = label[at];
    const float l = lse[at];
    int h0, h1, w0, w1;
    float lh, lw;
    src_index_t(ho, Hi, Ho, align, h0, h1, lh);
    src_index_t(wo, Wi, Wo, align, w0, w1, lw);
    const float v = bilerp<T>(img, Wi, ld, c, h0, h1, w0, w1, lh, lw);
    resid[(ho - rh0) * RW + (wo - rw0)] =
        (lb == ignore || (unsigned long long)lb >= (unsigned long long)C) ? 0.f : expf(v - l) - (c == lb ? 1.f : 0.f);
  }
  __syncthreads();
  const int ly = threadIdx.x / tw, lx = threadIdx.x % tw;
  const int hi = ty0 + ly, wi = tx0 + lx;
  if (ly >= th || hi >= Hi || wi >= Wi) return;
  const float inv = acc[1] > 0.f ? gscale / acc[1] : 0.f;
  int hlo, hhi, wlo, whi;
  dst_range(hi, Hi, Ho, align, hlo, hhi);
  dst_range(wi, Wi, Wo, align, wlo, whi);
  float g = 0.f;
  for (int ho = hlo; ho <= hhi; ++ho) {
    const float wh = tap_weight(ho, hi, Hi, Ho, align);
    const float* row = resid + (ho - rh0) * RW - rw0;
    float a = 0.f;
    for (int wo = wlo; wo <= whi; ++wo) a += tap_weight(wo, wi, Wi, Wo, align) * row[wo];
    g += wh * a;
  }
  const size_t pix = ((size_t)n * Hi + hi) * Wi + wi;
  Elem<T>::st(dlo + pix * ld + c, g * inv);
  if (c == 0)
    for (int k = C; k < ld; ++k) Elem<T>::st(dlo + pix * ld + k, 0.f);
}
// Vector forms of the two passes for heads with many classes (22 / 71): a thread reads VE = 16 B / sizeof(T) classes of
// a corner per load instead of one, pass 2 stages the residuals of VE classes at once (workgroup = tile x image x class
// chunk, so every label pixel costs 4 vector loads per CHUNK instead of 4 scalar loads per CLASS) and four lanes share a
// low-res pixel's footprint rows.  71 classes at 512 x 512, B = 8: pass 1 470 -> ~150 us, pass 2 1000 -> ~250 us.
// Needs ld % VE == 0; the padded channels C .. ld - 1 get zeros from the chunk that holds them.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_vec_kernel(const T* __restrict__ lo, const long long* __restrict__ label,
                                                           int n_img, int C, int Hi, int Wi, int ld, int Ho, int Wo,
                                                           int align, int ignore, float* __restrict__ lse,
                                                           float* __restrict__ part) {
  constexpr int VE = VecT<T>::VE;
  __shared__ float red[8];
  const long long HW = (long long)Ho * Wo, total = (long long)n_img * HW;
  float loss = 0.f, cnt = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long lb = label[i];
    if (lb == ignore) { lse[i] = 0.f; continue; }
    if ((unsigned long long)lb >= (unsigned long long)C) { lse[i] = 0.f; loss = NAN; continue; }
    const int wo = (int)(i % Wo);
    const int ho = (int)((i / Wo) % Ho);
    const int n = (int)(i / HW);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index_t(ho, Hi, Ho, align, h0, h1, lh);
    src_index_t(wo, Wi, Wo, align, w0, w1, lw);
    const T* img = lo + (size_t)n * Hi * Wi * ld;
    const T *p00 = img + ((size_t)h0 * Wi + w0) * ld, *p01 = img + ((size_t)h0 * Wi + w1) * ld;
    const T *p10 = img + ((size_t)h1 * Wi + w0) * ld, *p11 = img + ((size_t)h1 * Wi + w1) * ld;
    const float k00 = (1.f - lh) * (1.f - lw), k01 = (1.f - lh) * lw, k10 = lh * (1.f - lw), k11 = lh * lw;
    float m = -INFINITY, s = 0.f, picked = 0.f;
    for (int q = 0; q < C; q += VE) {
      float a[VE], b[VE], cc[VE], d[VE], v[VE];
      VecT<T>::load(p00 + q, a);
      VecT<T>::load(p01 + q, b);
      VecT<T>::load(p10 + q, cc);
      VecT<T>::load(p11 + q, d);
      float cm = -INFINITY;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        v[e] = k00 * a[e] + k01 * b[e] + k10 * cc[e] + k11 * d[e];
        if (q + e < C) cm = fmaxf(cm, v[e]);
        if (q + e == lb) picked = v[e];
      }
      const float mn = fmaxf(m, cm);
      s *= expf(m - mn);
#pragma unroll
      for (int e = 0; e < VE; ++e)
        if (q + e < C) s += expf(v[e] - mn);
      m = mn;
    }
    const float l = m + logf(s);
    lse[i] = l;
    loss += l - picked;
    cnt += 1.f;
  }
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv] = loss; red[4 + wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 + 2 * blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    part[3 + 2 * blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_vec_kernel(const T* __restrict__ lo, const long long* __restrict__ label,
                                                           const float* __restrict__ lse, const float* __restrict__ acc,
                                                           float gscale, int n_img, int C, int Hi, int Wi, int ld, int Ho,
                                                           int Wo, int align, int ignore, int th, int tw, int RW,
                                                           int tiles_w, int nchunk, T* __restrict__ dlo) {
  constexpr int VE = VecT<T>::VE;
  extern __shared__ float resid[];   // [RH][RW][VE]
  const int q = (blockIdx.y % nchunk) * VE, n = blockIdx.y / nchunk;
  const int ty0 = (blockIdx.x / tiles_w) * th, tx0 = (blockIdx.x % tiles_w) * tw;
  const int ty1 = min(ty0 + th, Hi) - 1, tx1 = min(tx0 + tw, Wi) - 1;
  const int ly = (threadIdx.x >> 2) / tw, lx = (threadIdx.x >> 2) % tw, sub = threadIdx.x & 3;
  const int hi = ty0 + ly, wi = tx0 + lx;
  const bool live = ly < th && hi < Hi && wi < Wi;
  const size_t pix = ((size_t)n * Hi + hi) * Wi + wi;
  float g[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) g[e] = 0.f;
  if (q >= C) {   // a chunk of padding only
    if (live && sub == 0) VecT<T>::store(dlo + pix * ld + q, g);
    return;
  }
  int rh0, rh1, rw0, rw1, t0, t1;
  dst_range(ty0, Hi, Ho, align, rh0, t1);
  dst_range(ty1, Hi, Ho, align, t0, rh1);
  dst_range(tx0, Wi, Wo, align, rw0, t1);
  dst_range(tx1, Wi, Wo, align, t0, rw1);
  const int rw = rw1 - rw0 + 1, cnt = (rh1 - rh0 + 1) * rw;
  const T* img = lo + (size_t)n * Hi * Wi * ld + q;
#pragma unroll 2
  for (int e = threadIdx.x; e < cnt; e += 256) {
    const int ho = rh0 + e / rw, wo = rw0 + e % rw;
    const size_t at = ((size_t)n * Ho + ho) * Wo + wo;
    const long long lb = label[at];
    const float l = lse[at];
    int h0, h1, w0, w1;
    float lh, lw;
    src_index_t(ho, Hi, Ho, align, h0, h1, lh);
    src_index_t(wo, Wi, Wo, align, w0, w1, lw);
    const float k00 = (1.f - lh) * (1.f - lw), k01 = (1.f - lh) * lw, k10 = lh * (1.f - lw), k11 = lh * lw;
    float a[VE], b[VE], cc[VE], d[VE], r[VE];
    VecT<T>::load(img + ((size_t)h0 * Wi + w0) * ld, a);
    VecT<T>::load(img + ((size_t)h0 * Wi + w1) * ld, b);
    VecT<T>::load(img + ((size_t)h1 * Wi + w0) * ld, cc);
    VecT<T>::load(img + ((size_t)h1 * Wi + w1) * ld, d);
    const bool dead = lb == ignore || (unsigned long long)lb >= (unsigned long long)C;
#pragma unroll
    for (int k = 0; k < VE; ++k) {
      const float v = k00 * a[k] + k01 * b[k] + k10 * cc[k] + k11 * d[k];
      r[k] = (dead || q + k >= C) ? 0.f : expf(v - l) - (q + k == lb ? 1.f : 0.f);
    }
    float4* dst = (float4*)(resid + ((size_t)(ho - rh0) * RW + (wo - rw0)) * VE);
#pragma unroll
    for (int k = 0; k < VE / 4; ++k) dst[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
  }
  __syncthreads();
  if (live) {
    int hlo, hhi, wlo, whi;
    dst_range(hi, Hi, Ho, align, hlo, hhi);
    dst_range(wi, Wi, Wo, align, wlo, whi);
    for (int ho = hlo + sub; ho <= hhi; ho += 4) {
      const float wh = tap_weight(ho, hi, Hi, Ho, align);
      if (wh == 0.f) continue;
      const float* row = resid + ((size_t)(ho - rh0) * RW - rw0) * VE;
      for (int wo = wlo; wo <= whi; ++wo) {
        const float w = wh * tap_weight(wo, wi, Wi, Wo, align);
        const float4* src = (const float4*)(row + (size_t)wo * VE);
#pragma unroll
        for (int k = 0; k < VE / 4; ++k) {
          const float4 t = src[k];
          g[4 * k] += w * t.x; g[4 * k + 1] += w * t.y; g[4 * k + 2] += w * t.z; g[4 * k + 3] += w * t.w;
        }
      }
    }
  }
  // the four lanes of a pixel are adjacent lanes of one wave (dead pixels carry zeros)
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    g[e] += __shfl_xor(g[e], 1);
    g[e] += __shfl_xor(g[e], 2);
  }
  if (live && sub == 0) {
    const float inv = acc[1] > 0.f ? gscale / acc[1] : 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) g[e] *= inv;
    VecT<T>::store(dlo + pix * ld + q, g);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void zero_tail_kernel(T* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) Elem<T>::st(p + i, 0.f);
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient of the small-Cin 3x3 conv (NCHW f32 input): dw[co][ci][kh][kw] += sum_p dy[p][co] * x[p @ tap]
// = a TN GEMM over the pixels with K-operand "im2col(x)": the <= 27 taps of a pixel are gathered once into a
// [pixels][KP] matrix of the compute dtype (KP = 32 bf16 / 28 f32 columns, 26 MB for the 224x224 stem) and the MFMA
// weight-gradient kernel (conv_wgrad.hip) does the reduction as a 1x1 "conv" - the first version (LDS-staged
// scalar FMAs + one atomic per workgroup and weight) took 500 us for 1.4 GFLOP.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int KP>
__global__ __launch_bounds__(256) void smallcin_im2col_kernel(const float* __restrict__ x, T* __restrict__ col, int N,
                                                              int Cin, int H, int W, int stride, int Ho, int Wo) {
  constexpr int VE = VecT<T>::VE;
  const long long M = (long long)N * Ho * Wo;
  for (long long pix = blockIdx.x * 256ll + threadIdx.x; pix < M; pix += (long long)gridDim.x * 256) {
    int wo, ho, n;
    split_pixel(pix, Wo, Ho, wo, ho, n);
    float v[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      const int ci = j / 9, kh = (j % 9) / 3, kw = j % 3;
      const int hi = ho * stride - 1 + kh, wi = wo * stride - 1 + kw;
      const bool ok = ci < Cin && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      v[j] = ok ? x[(((size_t)n * Cin + ci) * H + hi) * W + wi] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < KP / VE; ++q) VecT<T>::store(col + (size_t)pix * KP + q * VE, v + q * VE);
  }
}
__global__ void smallcin_scatter_kernel(const float* __restrict__ t, float* __restrict__ dw, int Cout, int K, int KP) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // dw[co][ci][kh][kw] flat = co * K + j
  if (i < Cout * K) dw[i] += t[(i / K) * KP + (i % K)];
}

// The same weight gradient for bf16 / Cout = 64 (both stems) WITHOUT the [pixels][32] matrix in memory: one persistent launch on the matrix
// cores + a fixed-order sum of its per-workgroup partials.  D[co][k] += sum_px dy[px][co] * x[px @ tap k]: the pixels are the MFMA's
// K dimension, so both operands are read "down the columns" of LDS tiles - dy[128 px][64 co] (row pitch 136 B: the four lane groups of
// a fragment land on banks 0 / 16 / 32 / 48) with ds_read_u16, the taps from the f32 patch of the forward kernel (conv3x3_smallcin_mfma)
// rounded to bf16 exactly as the im2col matrix was.  A wave owns 32 of an item's 128 pixels = one K step of the 16x16x32 MFMA for
// 4 channel blocks x 2 tap blocks; patch and dy rows of the NEXT item travel in registers.  im2col + zero + GEMM + reduce + scatter
// (5 launches, 26 MB written and read back) took 44 .. 56 us per stem.
constexpr int kSmallcinWG = 512;   // persistent workgroups = rows of the partials buffer
template <int STRIDE>
__global__ __launch_bounds__(256) void smallcin_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                  float* __restrict__ part, int N, int Cin, int H, int W, int Ho, int Wo,
                                                                  int tiles_w, int total) {
  constexpr int TW = 128, PW = (TW - 1) * STRIDE + 3, NC = (PW + 63) / 64, DP = 68;   // DP: dy tile row pitch in bf16
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* patch = (float*)smem_raw;                                   // [Cin * 3][PW]
  bf16_t* dyt = (bf16_t*)(smem_raw + ((9 * PW * 4 + 15) & ~15));     // [TW][DP]
  const int K = Cin * 9;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;
  int poff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = kb * 16 + col;
    poff[kb] = k < K ? (k / 3) * PW + (k % 3) : 0;   // (k >= K: a finite value, the column is never stored)
  }
  f32x4_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) acc[a][kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float pv[3][NC];
  uint4 dv[4];
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
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // dy rows of the run: 128 pixels x 8 chunks of 8 channels
      const int idx = threadIdx.x + 256 * u, px = idx >> 3, c8 = idx & 7;
      const int wo = tw_ * TW + px;
      dv[u] = wo < Wo ? *(const uint4*)(dy + ((size_t)row_ * Wo + wo) * 64 + c8 * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  if ((int)blockIdx.x < total) fetch((int)blockIdx.x);
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
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
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = threadIdx.x + 256 * u, px = idx >> 3, c8 = idx & 7;
      uint2* d = (uint2*)(dyt + px * DP + c8 * 8);   // (pitch 136 B: 8-byte aligned)
      d[0] = make_uint2(dv[u].x, dv[u].y);
      d[1] = make_uint2(dv[u].z, dv[u].w);
    }
    __syncthreads();
    if (item + (int)gridDim.x < total) fetch(item + (int)gridDim.x);
    const int px0 = wave * 32 + g * 8;       // the lane's 8 K-dimension pixels
    u32x4_t bf[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = patch[poff[kb] + (px0 + j) * STRIDE];
#pragma unroll
      for (int q = 0; q < 4; ++q) bf[kb][q] = pack2bf(xv[2 * q], xv[2 * q + 1]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bf16_t* ap = dyt + px0 * DP + a * 16 + col;
      u32x4_t af;
#pragma unroll
      for (int q = 0; q < 4; ++q) af[q] = (unsigned)ap[(2 * q) * DP] | ((unsigned)ap[(2 * q + 1) * DP] << 16);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        acc[a][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf[kb]),
                                                             acc[a][kb], 0, 0, 0);
    }
  }
  // the 4 waves' accumulators -> one [64 co][32 k] partial per workgroup (row co = 16 a + 4 g + i, column k = 16 kb + col)
  __syncthreads();
  float* red = (float*)smem_raw;   // [4][64][33]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(wave * 64 + a * 16 + 4 * g + i) * 33 + kb * 16 + col] = acc[a][kb][i];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 32; i += 256) {
    const int co = i >> 5, k = i & 31;
    part[(size_t)blockIdx.x * 2048 + i] = (red[co * 33 + k] + red[(64 + co) * 33 + k]) + (red[(128 + co) * 33 + k] + red[(192 + co) * 33 + k]);
  }
}
// dw[co][k] += sum over the workgroups' partials in a fixed order: 16 threads per output (each sums every 16th row, 8 loads in flight),
// 16 consecutive outputs per workgroup.  (4 threads x 128 rows with 4 loads in flight was 32 dependent L2 round trips: the sum cost
// more than the matrix-core launch in front of it.)
__global__ __launch_bounds__(256) void smallcin_wgrad_sum_kernel(const float* __restrict__ part, int rows, float* __restrict__ dw, int K) {
  __shared__ float red[16][17];
  const int ol = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;   // index into the [64][32] partial layout
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  int r = q;
  for (; r + 16 * 7 < rows; r += 16 * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += part[(size_t)(r + 16 * u) * 2048 + o];
  }
  for (; r < rows; r += 16) s[0] += part[(size_t)r * 2048 + o];
  red[q][ol] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][ol];
    const int co = o >> 5, k = o & 31;
    if (k < K) dw[co * K + k] += t;
  }
}

// OHWI f32 gradient -> OIHW f32 parameter gradient (accumulate = add into existing .grad)
__global__ void unpack_grad_kernel(const float* __restrict__ g, float* __restrict__ o, int Cout, int Cin, int KHW,
                                   int accumulate) {
  const long long total = (long long)Cout * Cin * KHW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    // i indexes the OIHW destination: (co*Cin + ci)*KHW + t ; source = (co*KHW + t)*Cin + ci
    const int t = (int)(i % KHW);
    const long long r = i / KHW;
    const int ci = (int)(r % Cin);
    const long long co = r / Cin;
    const float v = g[(co * KHW + t) * Cin + ci];
    o[i] = accumulate ? o[i] + v : v;
  }
}

// dgrad weights: OIHW f32 -> [Cin][KH][KW][Cout] (dtype) with the taps flipped (rot180), i.e. the OHWI weight of the
// transposed convolution
template <typename T>
__global__ void pack_dgrad_kernel(const float* __restrict__ w, T* __restrict__ o, int Cout, int Cin, int KH, int KW) {
  const long long total = (long long)Cout * Cin * KH * KW;
  const int KHW = KH * KW;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    // destination i = ((ci*KH + kh')*KW + kw')*Cout + co
    const int co = (int)(i % Cout);
    const long long r = i / Cout;
    const int t = (int)(r % KHW);
    const int ci = (int)(r / KHW);
    const int kh = KH - 1 - t / KW, kw = KW - 1 - t % KW;
    Elem<T>::st(o + i, w[(((size_t)co * Cin + ci) * KH + kh) * KW + kw]);
  }
}


// ------------------------------------------------------------------------------------------------------------
// multi-tensor weight re-pack: OIHW f32 -> OHWI and/or dgrad ([Cin][rot180 taps][Cout]) in the compute dtype, every
// job of a launch described in the kernel arguments (no device-side table, hipGraph-capturable as is).
// A workgroup moves a 64 co x TCI ci x all-taps tile through LDS: the source is read as 64 contiguous runs of
// TCI*KHW floats, both destinations are written in runs of >= 32 contiguous elements (the per-tensor dgrad pack read
// its source with a Cin*KHW*4-byte stride per lane).
// ------------------------------------------------------------------------------------------------------------
constexpr int kPackMaxJobs = 48;
constexpr int kPackTCO = 64, kPackRL = 144;   // co rows per tile, max floats per row (TCI * KHW <= kPackRL)
struct PackJobDev {
  const float* src;
  void* ohwi;
  void* dgrad;
  int Cout, Cin, KHW, TCI, tiles_ci, blk0;
};
struct PackArgs {
  PackJobDev j[kPackMaxJobs];
  int njobs;
};

// Division-free index walks: the first version decomposed a flat element index with 2-4 runtime integer divisions per
// element in each of its three loops (~100 VALU per 2-byte store): 270 us per launch for 95 M weights (1.4 TB/s), VALU-bound.
// Here lanes run along the contiguous direction of the access and the (row, tap) pair of a lane group advances by a
// constant (quotient, remainder) step.
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackArgs a) {
  __shared__ float tile[kPackTCO][kPackRL + 1];
  int ji = 0;
  while (ji + 1 < a.njobs && (int)blockIdx.x >= a.j[ji + 1].blk0) ++ji;   // uniform scan over <= 48 prefix sums
  const PackJobDev& J = a.j[ji];
  const int b = blockIdx.x - J.blk0;
  const int tci = b % J.tiles_ci, tco = b / J.tiles_ci;
  const int co0 = tco * kPackTCO, ci0 = tci * J.TCI;
  const int nco = min(kPackTCO, J.Cout - co0), nci = min(J.TCI, J.Cin - ci0);
  const int KHW = J.KHW, rl = nci * KHW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // read: row r (co0 + r) = rl contiguous floats starting at (co*Cin + ci0)*KHW
  // (all loads of the thread are issued before the first LDS write: one load per loop trip left 16 dependent ~1.5 us
  // round trips per workgroup - 20 us per 32 KB tile, 400 us per launch)
  constexpr int RPW = kPackTCO / 4, CPR = (kPackRL + 63) / 64;   // rows per wave, 64-float chunks per row
  float v[RPW][CPR];
  const float* src0 = J.src + ((size_t)co0 * J.Cin + ci0) * KHW;
  const size_t rstride = (size_t)J.Cin * KHW;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < CPR; ++j) {
      const int c = lane + 64 * j;
      v[i][j] = (r < nco && c < rl) ? src0[r * rstride + c] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < CPR; ++j) {
      const int c = lane + 64 * j;
      if (r < nco && c < rl) tile[r][c] = v[i][j];
    }
  }
  __syncthreads();
  if (J.ohwi) {  // dst[(co*KHW + t)*Cin + ci]: runs of nci contiguous ci; a group of P lanes owns one (co, t) pair
    T* o = (T*)J.ohwi;
    int lg = 0;
    while ((1 << lg) < nci) ++lg;           // P = 2^lg >= nci (nci <= 64)
    const int P = 1 << lg, ci = threadIdx.x & (P - 1);
    const int step = 256 >> lg;             // pairs covered per trip
    const int sq = step / KHW, sr = step - sq * KHW;
    const int q0 = threadIdx.x >> lg;
    int r = q0 / KHW, t = q0 - r * KHW;
    for (int q = q0; q < nco * KHW; q += step) {
      if (ci < nci) Elem<T>::st(o + ((size_t)(co0 + r) * KHW + t) * J.Cin + ci0 + ci, tile[r][ci * KHW + t]);
      r += sq; t += sr;
      if (t >= KHW) { t -= KHW; ++r; }
    }
  }
  if (J.dgrad) {  // dst[(ci*KHW + (KHW-1-t))*Cout + co]: runs of nco contiguous co; a group of P lanes owns one (ci, t) pair
    T* o = (T*)J.dgrad;
    int lg = 0;
    while ((1 << lg) < nco) ++lg;
    const int P = 1 << lg, r = threadIdx.x & (P - 1);
    const int step = 256 >> lg;
    const int sq = step / KHW, sr = step - sq * KHW;
    const int q0 = threadIdx.x >> lg;
    int ci = q0 / KHW, t = q0 - ci * KHW;
    for (int q = q0; q < rl; q += step) {
      if (r < nco) Elem<T>::st(o + ((size_t)(ci0 + ci) * KHW + (KHW - 1 - t)) * J.Cout + co0 + r, tile[r][q]);
      ci += sq; t += sr;
      if (t >= KHW) { t -= KHW; ++ci; }
    }
  }
}

}  // namespace

extern "C" int cavp_pack_weights_multi(int32_t dtype, const cavp_pack_job* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += kPackMaxJobs) {
    PackArgs a{};
    a.njobs = njobs - j0 < kPackMaxJobs ? njobs - j0 : kPackMaxJobs;
    long long blk = 0;
    for (int i = 0; i < a.njobs; ++i) {
      const cavp_pack_job& jb = jobs[j0 + i];
      if (!jb.w_oihw || jb.Cout <= 0 || jb.Cin <= 0 || jb.KH <= 0 || jb.KW <= 0) return CAVP_ERR_BAD_ARG;
      PackJobDev& d = a.j[i];
      d.src = jb.w_oihw; d.ohwi = jb.ohwi; d.dgrad = jb.dgrad;
      d.Cout = jb.Cout; d.Cin = jb.Cin; d.KHW = jb.KH * jb.KW;
      if (d.KHW > kPackRL) return CAVP_ERR_UNSUPPORTED;
      int tci = kPackRL / d.KHW;
      if (tci > 64) tci = 64;
      d.TCI = tci;
      d.tiles_ci = (jb.Cin + tci - 1) / tci;
      d.blk0 = (int)blk;
      blk += (long long)d.tiles_ci * ((jb.Cout + kPackTCO - 1) / kPackTCO);
      if (blk > 0x7fffffffll) return CAVP_ERR_UNSUPPORTED;
    }
    if (dtype == CAVP_F32)
      pack_multi_kernel<float><<<(int)blk, 256, 0, s>>>(a);
    else
      pack_multi_kernel<bf16_t><<<(int)blk, 256, 0, s>>>(a);
    if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  }
  return CAVP_OK;
}

extern "C" int cavp_scale_f32(const float* in, float alpha, float* out, int32_t n, void* stream) {
  if (!in || !out || n <= 0) return CAVP_ERR_BAD_ARG;
  scale_vec_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(in, alpha, out, n);
  CHECK_LAUNCH();
}

extern "C" int cavp_colstats(int32_t dtype, const void* x, const float* shift, int64_t rows, int32_t C, int32_t ldx,
                             float* sum, float* sumsq, void* stream) {
  if (!x || !sum || !sumsq || rows <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || rows > 0x7fffffff) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x)) return CAVP_ERR_ALIGN;
  ColArgs a{};
  a.a = x; a.mean = shift; a.out0 = sum; a.out1 = sumsq; a.rows = (int)rows; a.C = C; a.lda = ldx;
  return launch_col_reduce<0>(dtype, a, (hipStream_t)stream);
}

extern "C" int cavp_bn_finalize(const float* sum, const float* sumsq, const float* stat_shift, int64_t count, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* scale, float* shift, float* mean, float* rstd, int32_t C, void* stream) {
  if (!sum || !sumsq || !gamma || !beta || !scale || !shift || !mean || !rstd || count <= 0 || C <= 0)
    return CAVP_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CAVP_ERR_BAD_ARG;
  const float unbias = count > 1 ? (float)((double)count / (double)(count - 1)) : 1.f;
  bn_finalize_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(sum, sumsq, stat_shift, (float)(1.0 / (double)count), unbias,
                                                                     gamma, beta, eps, momentum, running_mean,
                                                                     running_var, scale, shift, mean, rstd, C);
  CHECK_LAUNCH();
}

extern "C" int cavp_bn_finalize_tiles(const float* tile_stats, int32_t tiles, int32_t rows_per_tile, int64_t count,
                                      const float* gamma, const float* beta, float eps, float momentum,
                                      float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                                      float* rstd, int32_t C, void* stream) {
  if (!tile_stats || !gamma || !beta || !scale || !shift || !mean || !rstd || tiles <= 0 || rows_per_tile <= 0 || count <= 0 ||
      C <= 0 || (long long)tiles * rows_per_tile < count)
    return CAVP_ERR_BAD_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return CAVP_ERR_BAD_ARG;
  if (tiles >= 1024)
    bn_finalize_tiles_wide_kernel<<<C, 256, 0, (hipStream_t)stream>>>(tile_stats, tiles, rows_per_tile, count, gamma, beta, eps,
                                                                      momentum, running_mean, running_var, scale, shift, mean,
                                                                      rstd, C, nullptr);
  else
    bn_finalize_tiles_kernel<<<(C + 3) / 4, 256, 0, (hipStream_t)stream>>>(tile_stats, tiles, rows_per_tile, count, gamma, beta,
                                                                          eps, momentum, running_mean, running_var, scale,
                                                                          shift, mean, rstd, C, nullptr);
  CHECK_LAUNCH();
}

extern "C" int cavp_bn_tiles_to_moments(const float* tile_stats, int32_t tiles, int32_t rows_per_tile, int64_t count,
                                        float* moments, int32_t C, void* stream) {
  if (!tile_stats || !moments || tiles <= 0 || rows_per_tile <= 0 || count <= 0 || C <= 0 || (long long)tiles * rows_per_tile < count)
    return CAVP_ERR_BAD_ARG;
  if (tiles >= 1024)
    bn_finalize_tiles_wide_kernel<<<C, 256, 0, (hipStream_t)stream>>>(tile_stats, tiles, rows_per_tile, count, nullptr, nullptr, 0.f,
                                                                      0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, C,
                                                                      moments);
  else
    bn_finalize_tiles_kernel<<<(C + 3) / 4, 256, 0, (hipStream_t)stream>>>(tile_stats, tiles, rows_per_tile, count, nullptr, nullptr,
                                                                          0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                                          nullptr, C, moments);
  CHECK_LAUNCH();
}

extern "C" int cavp_scale_shift_act(int32_t dtype, const void* x, const float* scale, const float* shift,
                                    const void* residual, void* y, int64_t rows, int32_t C, int32_t ldx, int32_t ldr,
                                    int32_t ldy, int32_t act, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || ldx < C || ldy < C || (residual && ldr < C)) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE || ldy % VE || (residual && ldr % VE)) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(y) || (residual && !al16(residual))) return CAVP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (!scale && !shift && ldx == C && ldy == C && (!residual || ldr == C)) {
    // coefficient-free pass over dense tensors (the GELU of the token MLP, C = 1216): any power-of-two row length will do
    const long long vecs = rows * (long long)(C / VE);
    int cv2 = 256;
    while (cv2 > 1 && vecs % cv2) cv2 >>= 1;
    rows = vecs / cv2;
    C = cv2 * VE;
    ldx = ldy = ldr = C;
  }
  if (flat_ok(C, VE, scale, shift)) {
    const int CV = C / VE;
    static const int flat_kb = cavp_knob_int("CAVP_FLAT_SSA_KB", 16);
    const int rpb = flat_rows_per_block(rows, C, 16 / VE, CV, (long long)flat_kb << 10, 1 << 20);
    const int gx = (int)((rows + rpb - 1) / rpb);
    if (dtype == CAVP_F32)
      scale_shift_act_flat_kernel<float><<<gx, 256, 0, s>>>((const float*)x, scale, shift, (const float*)residual, (float*)y, rows, rpb, CV, ldx, ldr, ldy, act);
    else
      scale_shift_act_flat_kernel<bf16_t><<<gx, 256, 0, s>>>((const bf16_t*)x, scale, shift, (const bf16_t*)residual, (bf16_t*)y, rows, rpb, CV, ldx, ldr, ldy, act);
    CHECK_LAUNCH();
  }
  dim3 grid;
  const RowLoop g = row_loop_geometry(rows, C, VE, grid);
  if (dtype == CAVP_F32)
    scale_shift_act_kernel<float><<<grid, 256, 0, s>>>((const float*)x, scale, shift, (const float*)residual, (float*)y, g, ldx, ldr, ldy, act);
  else
    scale_shift_act_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, scale, shift, (const bf16_t*)residual, (bf16_t*)y, g, ldx, ldr, ldy, act);
  CHECK_LAUNCH();
}

// Sum the per-tile (sum g, sum g * zhat) pairs a data-gradient launch wrote (cavp_conv2d_nhwc_bnbwd) over the tiles: 256 threads =
// 16 channels x 16 tile lanes (a tile row of 16 channels is 128 contiguous bytes), lane sums in ascending tile order, then a fixed
// tree over the 16 lanes through LDS.  Adds into sum_g / sum_gz (the BatchNorm's affine-gradient buffers).
__global__ __launch_bounds__(256) void bn_bwd_sum_tiles_kernel(const float* __restrict__ part, int tiles, int C, float* __restrict__ sum_g,
                                                               float* __restrict__ sum_gz) {
  __shared__ float2 red[16][17];
  const int ch = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int t = tl;
    for (; t + 16 * 7 < tiles; t += 16 * 8) {   // 8 loads in flight per thread
      float2 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = *(const float2*)(part + ((size_t)(t + 16 * u) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 += q[u].x; a1 += q[u].y; }
    }
    for (; t < tiles; t += 16) {
      const float2 q = *(const float2*)(part + ((size_t)t * C + c) * 2);
      a0 += q.x; a1 += q.y;
    }
  }
  red[tl][ch] = make_float2(a0, a1);
  __syncthreads();
  if (tl == 0 && c < C) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s0 += red[k][ch].x; s1 += red[k][ch].y; }
    sum_g[c] += s0;
    sum_gz[c] += s1;
  }
}

extern "C" int cavp_bn_bwd_sum_tiles(const float* partials, int32_t tiles, int32_t C, float* sum_g, float* sum_gz, void* stream) {
  if (!partials || !sum_g || !sum_gz || tiles <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  bn_bwd_sum_tiles_kernel<<<(C + 15) / 16, 256, 0, (hipStream_t)stream>>>(partials, tiles, C, sum_g, sum_gz);
  CHECK_LAUNCH();
}

extern "C" int cavp_bn_act_bwd_reduce(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                                      const float* rstd, int64_t rows, int32_t C, int32_t ld_dy, int32_t ld_y,
                                      int32_t ld_z, int32_t act, float* sum_g, float* sum_gz, const float* fwd_scale,
                                      const float* fwd_shift, void* stream) {
  if (!dy || !z || !mean || !rstd || !sum_g || !sum_gz || rows <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  if (!y && act != CAVP_ACT_NONE && (!fwd_scale || !fwd_shift)) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || rows > 0x7fffffff) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld_dy % VE || (y && ld_y % VE) || ld_z % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(dy) || (y && !al16(y)) || !al16(z)) return CAVP_ERR_ALIGN;
  ColArgs a{};
  a.a = dy; a.b = y; a.c = z; a.mean = mean; a.rstd = rstd; a.out0 = sum_g; a.out1 = sum_gz;
  a.fscale = fwd_scale; a.fshift = fwd_shift;
  a.rows = (int)rows; a.C = C; a.lda = ld_dy; a.ldb = ld_y; a.ldc = ld_z; a.act = act;
  return launch_col_reduce<1>(dtype, a, (hipStream_t)stream);
}

extern "C" int cavp_bn_act_bwd_apply_acc(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                                         const float* rstd, const float* gamma, const float* sum_g, const float* sum_gz,
                                         int64_t rows, int32_t C, int32_t ld_dy, int32_t ld_y, int32_t ld_z, int32_t act,
                                         void* dz, int32_t ld_dz, void* g_out, int32_t ld_g, const float* fwd_scale,
                                         const float* fwd_shift, float* dbeta_acc, float* dgamma_acc, void* stream);
extern "C" int cavp_bn_act_bwd_apply(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                                     const float* rstd, const float* gamma, const float* sum_g, const float* sum_gz,
                                     int64_t rows, int32_t C, int32_t ld_dy, int32_t ld_y, int32_t ld_z, int32_t act,
                                     void* dz, int32_t ld_dz, void* g_out, int32_t ld_g, const float* fwd_scale,
                                     const float* fwd_shift, void* stream) {
  return cavp_bn_act_bwd_apply_acc(dtype, dy, y, z, mean, rstd, gamma, sum_g, sum_gz, rows, C, ld_dy, ld_y, ld_z, act, dz, ld_dz, g_out,
                                   ld_g, fwd_scale, fwd_shift, nullptr, nullptr, stream);
}

extern "C" int cavp_bn_act_bwd_apply_acc(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                                         const float* rstd, const float* gamma, const float* sum_g, const float* sum_gz,
                                         int64_t rows, int32_t C, int32_t ld_dy, int32_t ld_y, int32_t ld_z, int32_t act,
                                         void* dz, int32_t ld_dz, void* g_out, int32_t ld_g, const float* fwd_scale,
                                         const float* fwd_shift, float* dbeta_acc, float* dgamma_acc, void* stream) {
  if (!dy || !z || !mean || !rstd || !gamma || !sum_g || !sum_gz || !dz || rows <= 0 || C <= 0 || (dbeta_acc == nullptr) != (dgamma_acc == nullptr))
    return CAVP_ERR_BAD_ARG;
  if (!y && act != CAVP_ACT_NONE && (!fwd_scale || !fwd_shift)) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld_dy % VE || (y && ld_y % VE) || ld_z % VE || ld_dz % VE || (g_out && ld_g % VE)) return CAVP_ERR_UNSUPPORTED;
  if (!al16(dy) || (y && !al16(y)) || !al16(z) || !al16(dz) || (g_out && !al16(g_out))) return CAVP_ERR_ALIGN;
  const float inv_m = (float)(1.0 / (double)rows);
  hipStream_t s = (hipStream_t)stream;
  if (flat_ok(C, VE, mean, rstd) && al16(gamma) && al16(sum_g) && al16(sum_gz) && (!fwd_scale || (al16(fwd_scale) && al16(fwd_shift)))) {
    const int CV = C / VE;
    static const int flat_kb = cavp_knob_int("CAVP_FLAT_APPLY_KB", 16);
    const int rpb = flat_rows_per_block(rows, C, 16 / VE, CV, (long long)flat_kb << 10, 1 << 20);
    const int gx = (int)((rows + rpb - 1) / rpb);
    if (dtype == CAVP_F32)
      bn_bwd_apply_flat_kernel<float><<<gx, 256, 0, s>>>((const float*)dy, (const float*)y, (const float*)z, mean, rstd, gamma, sum_g, sum_gz, inv_m, (float*)dz, (float*)g_out, rows, rpb, CV, ld_dy, ld_y, ld_z, ld_dz, ld_g, act, fwd_scale, fwd_shift, dbeta_acc, dgamma_acc);
    else
      bn_bwd_apply_flat_kernel<bf16_t><<<gx, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)z, mean, rstd, gamma, sum_g, sum_gz, inv_m, (bf16_t*)dz, (bf16_t*)g_out, rows, rpb, CV, ld_dy, ld_y, ld_z, ld_dz, ld_g, act, fwd_scale, fwd_shift, dbeta_acc, dgamma_acc);
    CHECK_LAUNCH();
  }
  dim3 grid;
  const RowLoop g = row_loop_geometry(rows, C, VE, grid);
  if (dtype == CAVP_F32)
    bn_bwd_apply_kernel<float><<<grid, 256, 0, s>>>((const float*)dy, (const float*)y, (const float*)z, mean, rstd, gamma, sum_g, sum_gz, inv_m, (float*)dz, (float*)g_out, g, ld_dy, ld_y, ld_z, ld_dz, ld_g, act, fwd_scale, fwd_shift, dbeta_acc, dgamma_acc);
  else
    bn_bwd_apply_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)z, mean, rstd, gamma, sum_g, sum_gz, inv_m, (bf16_t*)dz, (bf16_t*)g_out, g, ld_dy, ld_y, ld_z, ld_dz, ld_g, act, fwd_scale, fwd_shift, dbeta_acc, dgamma_acc);
  CHECK_LAUNCH();
}

extern "C" int cavp_act_bwd(int32_t dtype, const void* dy, const void* ref, void* dx, int64_t rows, int32_t C,
                            int32_t ld_dy, int32_t ld_ref, int32_t ld_dx, int32_t act, void* stream) {
  if (!dy || !ref || !dx || rows <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld_dy % VE || ld_ref % VE || ld_dx % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(dy) || !al16(ref) || !al16(dx)) return CAVP_ERR_ALIGN;
  long long nb = (rows * (C / VE) + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    act_bwd_kernel<float><<<(int)nb, 256, 0, s>>>((const float*)dy, (const float*)ref, (float*)dx, rows, C, ld_dy, ld_ref, ld_dx, act);
  else
    act_bwd_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)ref, (bf16_t*)dx, rows, C, ld_dy, ld_ref, ld_dx, act);
  CHECK_LAUNCH();
}

extern "C" int cavp_add(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (n % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(a) || !al16(b) || !al16(out)) return CAVP_ERR_ALIGN;
  long long nb = (n / VE + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    add_kernel<float><<<(int)nb, 256, 0, s>>>((const float*)a, (const float*)b, (float*)out, n);
  else
    add_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
  CHECK_LAUNCH();
}

extern "C" int cavp_colsum(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, float* out,
                           void* stream) {
  if (!x || !out || rows <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || rows > 0x7fffffff) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x)) return CAVP_ERR_ALIGN;
  ColArgs a{};
  a.a = x; a.out0 = out; a.out1 = out; a.rows = (int)rows; a.C = C; a.lda = ldx;
  return launch_col_reduce<2>(dtype, a, (hipStream_t)stream);
}

// Per-tile (mean, M2) of 128-row tiles of a [rows][C] tensor in ONE pass (the tile is held in registers: sum -> mean -> centred
// squares), in the layout the conv epilogue writes (ts[tile][C][2]), for BatchNorm inputs whose producer could not fuse the
// statistics (split-K convs, the small-Cin stem, channel slices).  Replaces column sum -> mean -> centred column squares (two passes
// over the tensor, four launches) by one pass + cavp_bn_finalize_tiles; equally cancellation-free.
template <typename T>
__global__ __launch_bounds__(256) void col_tile_stats_kernel(const T* __restrict__ x, float* __restrict__ ts, long long rows, int C,
                                                             int ld) {
  constexpr int VE = VecT<T>::VE, CW = 16 * VE;
  __shared__ float red[16][CW + 1];
  __shared__ float mean_s[CW];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c0 = (blockIdx.y * 16 + cg) * VE;
  const bool ok = c0 < C;
  const long long r0 = (long long)blockIdx.x * 128;
  const int nb = (int)(rows - r0 < 128 ? rows - r0 : 128);
  float v[8][VE];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = rl + 16 * j;
    if (ok && r < nb) {
      VecT<T>::load(x + (r0 + r) * ld + c0, v[j]);
    } else {
#pragma unroll
      for (int e = 0; e < VE; ++e) v[j][e] = 0.f;
    }
  }
  float acc[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[e] += v[j][e];
    red[rl][cg * VE + e] = acc[e];
  }
  __syncthreads();
  if (threadIdx.x < CW) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
    mean_s[threadIdx.x] = t / (float)nb;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    const float m = mean_s[cg * VE + e];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (rl + 16 * j < nb) ? v[j][e] - m : 0.f;
      q += d * d;
    }
    red[rl][cg * VE + e] = q;
  }
  __syncthreads();
  if (threadIdx.x < CW) {
    const int c = blockIdx.y * CW + threadIdx.x;
    if (c < C) {
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) q += red[k][threadIdx.x];
      *(float2*)(ts + ((size_t)blockIdx.x * C + c) * 2) = make_float2(mean_s[threadIdx.x], q);
    }
  }
}

extern "C" int cavp_col_tile_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, float* tile_stats,
                                   void* stream) {
  if (!x || !tile_stats || rows <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || rows > 0x7fffffff) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || ((uintptr_t)tile_stats & 7)) return CAVP_ERR_ALIGN;
  const dim3 grid((unsigned)((rows + 127) / 128), cdiv_h(C, 16 * VE));
  if (dtype == CAVP_F32)
    col_tile_stats_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, tile_stats, rows, C, ldx);
  else
    col_tile_stats_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, tile_stats, rows, C, ldx);
  CHECK_LAUNCH();
}

// out[g][c] += sum over the `rpg` rows of group g (the per-image bias gradient of the ASPP pooled branch: one group per image).
// One workgroup per (group, 16 channel vectors): 16 row lanes x 16 vectors, LDS reduce, single owner per output -> no atomics.
template <typename T>
__global__ __launch_bounds__(256) void colsum_groups_kernel(const T* __restrict__ x, float* __restrict__ out, int rpg, int C, int ld) {
  constexpr int VE = VecT<T>::VE;
  __shared__ float red[16][16 * VE + 1];
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c0 = (blockIdx.y * 16 + cg) * VE;
  const bool ok = c0 < C;
  float acc[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) acc[e] = 0.f;
  const T* xg = x + (size_t)blockIdx.x * rpg * ld;
  if (ok)
    for (int r = rl; r < rpg; r += 16) {
      float v[VE];
      VecT<T>::load(xg + (size_t)r * ld + c0, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < VE; ++e) red[rl][cg * VE + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * VE; i += 256) {
    const int c = blockIdx.y * 16 * VE + i;
    if (c >= C) continue;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += red[k][i];
    out[(size_t)blockIdx.x * C + c] += sum;
  }
}

extern "C" int cavp_colsum_groups(int32_t dtype, const void* x, int32_t groups, int32_t rows_per_group, int32_t C, int32_t ldx,
                                  float* out, void* stream) {
  if (!x || !out || groups <= 0 || rows_per_group <= 0 || C <= 0 || ldx < C) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x)) return CAVP_ERR_ALIGN;
  const dim3 grid(groups, cdiv_h(C, 16 * VE));
  if (dtype == CAVP_F32)
    colsum_groups_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)x, out, rows_per_group, C, ldx);
  else
    colsum_groups_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, out, rows_per_group, C, ldx);
  CHECK_LAUNCH();
}

extern "C" int cavp_maxpool_bwd_nhwc(int32_t dtype, const uint8_t* argmax, const void* dy, void* dx, int32_t N, int32_t H,
                                     int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, void* stream) {
  const void* x = argmax;
  if (!x || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || k > 15 || stride <= 0 || pad < 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE) return CAVP_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 7) || !al16(dy) || !al16(dx)) return CAVP_ERR_ALIGN;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  long long nb = ((long long)N * H * W * (C / VE) + 255) / 256;
  if (nb > 32768) nb = 32768;
  hipStream_t s = (hipStream_t)stream;
  if (stride == 2 && (long long)N * H <= 65535) {   // the model's pools: one grid row per input row, no per-element divisions
    const int CV = C / VE;
    int sh = -1;
    if ((CV & (CV - 1)) == 0) { sh = 0; while ((1 << sh) < CV) ++sh; }
    const dim3 grid((W * CV + 255) / 256, N * H);
    if (dtype == CAVP_F32)
      maxpool_bwd_s2_rows_kernel<float><<<grid, 256, 0, s>>>(argmax, (const float*)dy, (float*)dx, H, W, C, k, pad, Ho, Wo, sh);
    else
      maxpool_bwd_s2_rows_kernel<bf16_t><<<grid, 256, 0, s>>>(argmax, (const bf16_t*)dy, (bf16_t*)dx, H, W, C, k, pad, Ho, Wo, sh);
    CHECK_LAUNCH();
  }
  if (dtype == CAVP_F32)
    maxpool_bwd_kernel<float><<<(int)nb, 256, 0, s>>>(argmax, (const float*)dy, (float*)dx, N, H, W, C, k, stride, pad, Ho, Wo);
  else
    maxpool_bwd_kernel<bf16_t><<<(int)nb, 256, 0, s>>>(argmax, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, k, stride, pad, Ho, Wo);
  CHECK_LAUNCH();
}

extern "C" int cavp_bilinear_bwd_nhwc(int32_t dtype, const void* dy, void* dx, int32_t N, int32_t Hi, int32_t Wi,
                                      int32_t C, int32_t ld_dx, int32_t Ho, int32_t Wo, int32_t ld_dy,
                                      int32_t align_corners, void* stream) {
  if (!dy || !dx || N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || ld_dx < C || ld_dy < C)
    return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld_dx % VE || ld_dy % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(dy) || !al16(dx)) return CAVP_ERR_ALIGN;
  long long nb = ((long long)N * Hi * Wi * (C / VE) + 255) / 256;
  if (nb > 32768) nb = 32768;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    bilinear_bwd_nhwc_kernel<float><<<(int)nb, 256, 0, s>>>((const float*)dy, (float*)dx, N, Hi, Wi, C, ld_dx, Ho, Wo, ld_dy, align_corners);
  else
    bilinear_bwd_nhwc_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((const bf16_t*)dy, (bf16_t*)dx, N, Hi, Wi, C, ld_dx, Ho, Wo, ld_dy, align_corners);
  CHECK_LAUNCH();
}

extern "C" int cavp_bilinear_bwd_nchw_to_nhwc(int32_t dtype, const float* dy_nchw, void* dx, int32_t N,
                                              int32_t n_valid, int32_t Hi, int32_t Wi, int32_t C, int32_t ld_dx,
                                              int32_t Ho, int32_t Wo, int32_t align_corners, void* stream) {
  if (!dy_nchw || !dx || N <= 0 || n_valid < 0 || n_valid > N || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 ||
      ld_dx < C)
    return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  long long nb = ((long long)N * Hi * Wi * C + 255) / 256;
  if (nb > 32768) nb = 32768;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    bilinear_bwd_from_nchw_kernel<float><<<(int)nb, 256, 0, s>>>(dy_nchw, (float*)dx, N, n_valid, Hi, Wi, C, ld_dx, Ho, Wo, align_corners);
  else
    bilinear_bwd_from_nchw_kernel<bf16_t><<<(int)nb, 256, 0, s>>>(dy_nchw, (bf16_t*)dx, N, n_valid, Hi, Wi, C, ld_dx, Ho, Wo, align_corners);
  CHECK_LAUNCH();
}

extern "C" int cavp_bcast_add_nhwc(int32_t dtype, void* x, const float* v, float alpha, int32_t N, int32_t HW,
                                   int32_t C, int32_t ld, void* stream) {
  if (!x || !v || N <= 0 || HW <= 0 || C <= 0 || ld < C) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x)) return CAVP_ERR_ALIGN;
  long long nb = ((long long)N * HW * (C / VE) + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    bcast_add_kernel<float><<<(int)nb, 256, 0, s>>>((float*)x, v, alpha, N, HW, C, ld);
  else
    bcast_add_kernel<bf16_t><<<(int)nb, 256, 0, s>>>((bf16_t*)x, v, alpha, N, HW, C, ld);
  CHECK_LAUNCH();
}

extern "C" int cavp_ce_loss_nchw(const float* logits, const int64_t* labels, int32_t n_img, int32_t n_total, int32_t C,
                                 int64_t HW, int32_t ignore_index, float grad_scale, float* loss, float* dlogits,
                                 float* scratch2, void* stream) {
  if (!logits || !labels || !loss || !scratch2 || n_img <= 0 || n_total < n_img || C <= 0 || HW <= 0)
    return CAVP_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  long long nb = ((long long)n_img * HW + 255) / 256;
  if (nb > 1024) nb = 1024;   // = (CAVP_CE_SCRATCH_FLOATS - 2) / 2 partials
  ce_fwd_kernel<<<(int)nb, 256, 0, s>>>(logits, (const long long*)labels, n_img, C, HW, ignore_index, scratch2);
  ce_finish_kernel<<<1, 256, 0, s>>>(scratch2, (int)nb, loss);
  if (dlogits) {
    long long nb2 = ((long long)n_total * HW + 255) / 256;
    if (nb2 > 8192) nb2 = 8192;
    ce_bwd_kernel<<<(int)nb2, 256, 0, s>>>(logits, (const long long*)labels, n_img, n_total, C, HW, ignore_index,
                                          scratch2, grad_scale, dlogits);
  }
  CHECK_LAUNCH();
}

namespace {
// label-resolution rows (columns) under `t` consecutive low-res rows: see dst_range
inline int head_region(int t, int in, int out, int align) {
  double r = (double)out / (double)in;
  if (align && in > 1 && out > 1) r = std::max(r, (double)(out - 1) / (double)(in - 1));
  const long long span = (long long)std::floor((t + 1) * r) + 4;
  return (int)std::min<long long>(span, out);
}
template <typename T>
int head_launch(const T* lo, const long long* lab, int n_img, int n_total, int C, int Hi, int Wi, int ld, int Ho, int Wo,
                int align, int ignore, float gscale, float* loss, T* dlo, float* lse, float* scratch2, hipStream_t s) {
  constexpr int VE = VecT<T>::VE;
  // many classes: VE classes per load (head_*_vec_kernel).  (Measured for the binary head too - 2 classes padded to one 8-channel bf16
  // vector: 77 us against 63 us with the scalar kernels, the eight exp per label pixel outweigh the wider loads.)
  const bool vec = C >= 8 && ld % VE == 0;
  long long nb = ((long long)n_img * Ho * Wo + 255) / 256;
  if (nb > 1024) nb = 1024;   // = (CAVP_CE_SCRATCH_FLOATS - 2) / 2 partials
  if (vec)
    head_fwd_vec_kernel<T><<<(int)nb, 256, 0, s>>>(lo, lab, n_img, C, Hi, Wi, ld, Ho, Wo, align, ignore, lse, scratch2);
  else
    head_fwd_kernel<T><<<(int)nb, 256, 0, s>>>(lo, lab, n_img, C, Hi, Wi, ld, Ho, Wo, align, ignore, lse, scratch2);
  ce_finish_kernel<<<1, 256, 0, s>>>(scratch2, (int)nb, loss);
  if (!dlo) return CAVP_OK;
  const int per = vec ? VE : 1;               // floats of LDS per label pixel of the region
  int th = vec ? 8 : 16, tw = th, RH = 0, RW = 0;   // vec: four lanes per low-res pixel -> th * tw <= 64
  for (;;) {
    RH = head_region(th, Hi, Ho, align);
    RW = head_region(tw, Wi, Wo, align);
    if ((long long)RH * RW * 4 * per <= 60 * 1024) break;
    if (th == 1 && tw == 1) return CAVP_ERR_UNSUPPORTED;   // one low-res pixel spans > 15 K label pixels
    if (th >= tw) th = (th + 1) / 2; else tw = (tw + 1) / 2;
  }
  const int tiles_h = (Hi + th - 1) / th, tiles_w = (Wi + tw - 1) / tw;
  const int nchunk = vec ? ld / VE : C;
  if ((long long)n_img * nchunk > 65535) return CAVP_ERR_UNSUPPORTED;
  if (vec)
    head_bwd_vec_kernel<T><<<dim3(tiles_h * tiles_w, n_img * nchunk), 256, (size_t)RH * RW * 4 * per, s>>>(
        lo, lab, lse, scratch2, gscale, n_img, C, Hi, Wi, ld, Ho, Wo, align, ignore, th, tw, RW, tiles_w, nchunk, dlo);
  else
    head_bwd_kernel<T><<<dim3(tiles_h * tiles_w, n_img * C), 256, (size_t)RH * RW * 4, s>>>(
        lo, lab, lse, scratch2, gscale, n_img, C, Hi, Wi, ld, Ho, Wo, align, ignore, th, tw, RW, tiles_w, dlo);
  const long long tail = (long long)(n_total - n_img) * Hi * Wi * ld;   // images >= n_img contribute `* 0`
  if (tail > 0) {
    long long nz = (tail + 255) / 256;
    if (nz > 4096) nz = 4096;
    zero_tail_kernel<T><<<(int)nz, 256, 0, s>>>(dlo + (size_t)n_img * Hi * Wi * ld, tail);
  }
  return CAVP_OK;
}
}  // namespace

extern "C" int cavp_upsample_ce_head(int32_t dtype, const void* lo, const int64_t* labels, int32_t n_img,
                                     int32_t n_total, int32_t C, int32_t Hi, int32_t Wi, int32_t ld, int32_t Ho,
                                     int32_t Wo, int32_t align_corners, int32_t ignore_index, float grad_scale,
                                     float* loss, void* dlo, float* lse, float* scratch2, void* stream) {
  if (!lo || !labels || !loss || !lse || !scratch2 || n_img <= 0 || n_total < n_img || C <= 0 || Hi <= 0 || Wi <= 0 ||
      Ho <= 0 || Wo <= 0 || ld < C)
    return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long long* lab = (const long long*)labels;
  int st;
  if (dtype == CAVP_F32)
    st = head_launch<float>((const float*)lo, lab, n_img, n_total, C, Hi, Wi, ld, Ho, Wo, align_corners, ignore_index,
                            grad_scale, loss, (float*)dlo, lse, scratch2, s);
  else
    st = head_launch<bf16_t>((const bf16_t*)lo, lab, n_img, n_total, C, Hi, Wi, ld, Ho, Wo, align_corners,
                             ignore_index, grad_scale, loss, (bf16_t*)dlo, lse, scratch2, s);
  if (st != CAVP_OK) return st;
  CHECK_LAUNCH();
}

namespace {
struct SmallcinPlan { cavp_conv_desc d; size_t col_bytes, tmp_bytes, ws_bytes; int KP; long long M; int Ho, Wo; };
inline bool smallcin_plan(int dtype, int N, int Cin, int H, int W, int Cout, int stride, SmallcinPlan* pl) {
  if (!dt_ok(dtype) || Cin < 1 || Cin > 3 || Cout % 8 || N <= 0 || H <= 0 || W <= 0 || stride <= 0) return false;
  pl->Ho = (H - 1) / stride + 1; pl->Wo = (W - 1) / stride + 1;
  pl->M = (long long)N * pl->Ho * pl->Wo;
  pl->KP = dtype == CAVP_F32 ? 28 : 32;
  const size_t es = dtype == CAVP_F32 ? 4 : 2;
  if (pl->M * pl->KP * (long long)es >= 0x7fffffffll) return false;
  cavp_conv_desc d{};
  d.dtype = dtype; d.N = 1; d.H = 1; d.W = (int)pl->M; d.Cin = pl->KP; d.ldx = pl->KP; d.Cout = Cout; d.ldy = Cout;
  d.KH = d.KW = 1; d.stride = 1; d.pad = 0; d.dil = 1;
  pl->d = d;
  pl->col_bytes = ((size_t)pl->M * pl->KP * es + 255) / 256 * 256;
  pl->tmp_bytes = ((size_t)Cout * pl->KP * 4 + 255) / 256 * 256;
  pl->ws_bytes = cavp_conv2d_wgrad_workspace_bytes(&pl->d);
  return true;
}
}  // namespace

extern "C" size_t cavp_conv3x3_smallcin_wgrad_workspace_bytes(int32_t dtype, int32_t N, int32_t Cin, int32_t H, int32_t W,
                                                              int32_t Cout, int32_t stride) {
  SmallcinPlan pl;
  if (!smallcin_plan(dtype, N, Cin, H, W, Cout, stride, &pl)) return 0;
  return pl.col_bytes + pl.tmp_bytes + pl.ws_bytes;
}

extern "C" int cavp_conv3x3_smallcin_wgrad(int32_t dtype, const float* x_nchw, const void* dy_nhwc, float* dw_oihw,
                                           int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t stride,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!x_nchw || !dy_nhwc || !dw_oihw || !workspace) return CAVP_ERR_BAD_ARG;
  SmallcinPlan pl;
  if (!smallcin_plan(dtype, N, Cin, H, W, Cout, stride, &pl)) return CAVP_ERR_UNSUPPORTED;
  if (workspace_bytes < pl.col_bytes + pl.tmp_bytes + pl.ws_bytes || !al16(workspace)) return CAVP_ERR_WORKSPACE;
  char* col = (char*)workspace;
  float* tmp = (float*)(col + pl.col_bytes);
  void* ws = (char*)tmp + pl.tmp_bytes;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_BF16 && Cout == 64 && (stride == 1 || stride == 2) && al16(dy_nhwc)) {   // both stems: fused matrix-core version
    const int tw = (pl.Wo + 127) / 128;
    const long long items = (long long)N * pl.Ho * tw;
    const int grid = items > kSmallcinWG ? kSmallcinWG : (int)items;
    if (items <= 0x7fffffffll && workspace_bytes >= (size_t)grid * 2048 * 4) {
      const int PW = 127 * stride + 3;
      size_t lds = ((size_t)9 * PW * 4 + 15) / 16 * 16 + (size_t)128 * 68 * 2;
      if (lds < (size_t)4 * 64 * 33 * 4) lds = (size_t)4 * 64 * 33 * 4;
      float* part = (float*)workspace;
      if (stride == 2)
        smallcin_wgrad_mfma_kernel<2><<<grid, 256, lds, s>>>(x_nchw, (const bf16_t*)dy_nhwc, part, N, Cin, H, W, pl.Ho, pl.Wo, tw, (int)items);
      else
        smallcin_wgrad_mfma_kernel<1><<<grid, 256, lds, s>>>(x_nchw, (const bf16_t*)dy_nhwc, part, N, Cin, H, W, pl.Ho, pl.Wo, tw, (int)items);
      if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
      smallcin_wgrad_sum_kernel<<<128, 256, 0, s>>>(part, grid, dw_oihw, Cin * 9);
      CHECK_LAUNCH();
    }
  }
  long long nb = (pl.M + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (dtype == CAVP_F32)
    smallcin_im2col_kernel<float, 28><<<(int)nb, 256, 0, s>>>(x_nchw, (float*)col, N, Cin, H, W, stride, pl.Ho, pl.Wo);
  else
    smallcin_im2col_kernel<bf16_t, 32><<<(int)nb, 256, 0, s>>>(x_nchw, (bf16_t*)col, N, Cin, H, W, stride, pl.Ho, pl.Wo);
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (cavp_zero_f32_async(tmp, (size_t)Cout * pl.KP * 4, s) != hipSuccess) return CAVP_ERR_LAUNCH;
  const int st = cavp_conv2d_wgrad_nhwc(&pl.d, col, dy_nhwc, tmp, nullptr, ws, pl.ws_bytes, stream);
  if (st != CAVP_OK) return st;
  const int K = Cin * 9;
  smallcin_scatter_kernel<<<(Cout * K + 255) / 256, 256, 0, s>>>(tmp, dw_oihw, Cout, K, pl.KP);
  CHECK_LAUNCH();
}

extern "C" int cavp_unpack_weight_grad(const float* g_ohwi, float* g_oihw, int32_t Cout, int32_t Cin, int32_t KH,
                                       int32_t KW, int32_t accumulate, void* stream) {
  if (!g_ohwi || !g_oihw || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return CAVP_ERR_BAD_ARG;
  long long nb = ((long long)Cout * Cin * KH * KW + 255) / 256;
  if (nb > 8192) nb = 8192;
  unpack_grad_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(g_ohwi, g_oihw, Cout, Cin, KH * KW, accumulate);
  CHECK_LAUNCH();
}

extern "C" int cavp_pack_weight_dgrad(int32_t dtype, const float* w_oihw, void* w_t, int32_t Cout, int32_t Cin,
                                      int32_t KH, int32_t KW, void* stream) {
  if (!w_oihw || !w_t || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  long long nb = ((long long)Cout * Cin * KH * KW + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    pack_dgrad_kernel<float><<<(int)nb, 256, 0, s>>>(w_oihw, (float*)w_t, Cout, Cin, KH, KW);
  else
    pack_dgrad_kernel<bf16_t><<<(int)nb, 256, 0, s>>>(w_oihw, (bf16_t*)w_t, Cout, Cin, KH, KW);
  CHECK_LAUNCH();
}
