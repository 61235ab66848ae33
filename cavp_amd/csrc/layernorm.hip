// nn.LayerNorm forward / backward over the last dim of [rows][C] for gfx950 (attn.py:130,136,229; pvt.py norm layers).
//
// HBM-bound: the forward reads x once and writes y once, the backward reads x and dy once and writes dx once.
// * G lanes cooperate on a row (G = 8 / 16 / 32 / 64, the smallest power of two that covers C with 16-byte vectors),
//   so a wave holds 64 / G rows: C = 64 (PVT stage 1) keeps all 64 lanes busy instead of 8.
// * Every lane owns PLV 16-byte vectors of the row (channel (l + G i) * VE): one wide load per operand and one wide
//   store per row instead of the 2-byte accesses of the first version (5 loads + 5 loads + 5 stores per lane per row
//   at C = 304: 210 us per call against a 90 us HBM floor).
// * Two-pass statistics (mean, then centred variance) in registers, reductions by xor-shuffles inside the G-lane group.
// * gamma / beta of the lane's channels are loaded once, before the row loop.
// * backward: dgamma / dbeta are accumulated in registers over the workgroup's rows, combined across the row groups of
//   the wave by shuffles, across the 4 waves through LDS, and leave as one f32 atomic per channel per workgroup.
#include "common.h"

namespace {

// Sum over the G lanes of a row group (G = 8 / 16 / 32 / 64 consecutive lanes), every lane ends with the total.  DPP inside a 16-lane row
// (quad_perm xor 1 / xor 2, then the half-row and row mirrors), v_permlane16_swap / v_permlane32_swap across rows: no LDS crossbar.  The
// first version walked an xor tree of __shfl_xor = ds_bpermute_b32 + address arithmetic + an lgkmcnt wait per step; the backward does four
// such sums per row, every one on the critical path between a row's loads and its store.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  if constexpr (G >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  if constexpr (G >= 32) v = xor_add<16>(v);
  if constexpr (G >= 64) v = xor_add<32>(v);
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "row group = 8 .. 64 lanes");
  return v;
}

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return __uint_as_float(pack2bf(v, 0.f) << 16); }

template <typename T, int G, int PLV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, int rows,
                                                            int C, int ldx, int ldy, float eps, const T* __restrict__ branch,
                                                            const float* __restrict__ row_scale, int rows_per_group,
                                                            T* __restrict__ sum_out) {
  constexpr int VE = VecT<T>::VE, RW = 64 / G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l = lane % G, g = lane / G;
  float ga[PLV][VE], be[PLV][VE];
  bool ok[PLV];
#pragma unroll
  for (int i = 0; i < PLV; ++i) {
    const int c = (l + G * i) * VE;
    ok[i] = c < C;
#pragma unroll
    for (int q = 0; q < VE / 4; ++q) {   // 16-byte parameter loads (gamma / beta are 16-byte aligned, C % VE == 0)
      const float4 gq = ok[i] ? *(const float4*)(gamma + c + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 bq = ok[i] ? *(const float4*)(beta + c + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      ga[i][4 * q] = gq.x; ga[i][4 * q + 1] = gq.y; ga[i][4 * q + 2] = gq.z; ga[i][4 * q + 3] = gq.w;
      be[i][4 * q] = bq.x; be[i][4 * q + 1] = bq.y; be[i][4 * q + 2] = bq.z; be[i][4 * q + 3] = bq.w;
    }
  }
  const float invC = 1.f / (float)C;
  for (long long row = ((long long)blockIdx.x * 4 + wv) * RW + g; row < rows; row += (long long)gridDim.x * 4 * RW) {
    float v[PLV][VE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PLV; ++i) {
      if (ok[i]) {
        VecT<T>::load(x + (size_t)row * ldx + (l + G * i) * VE, v[i]);
        if (branch) {   // x + factor[row group] * branch first (residual + DropPath); the sum is stored, and normalised as stored
          float b[VE];
          VecT<T>::load(branch + (size_t)row * C + (l + G * i) * VE, b);
          const float rs = row_scale[row / rows_per_group];
#pragma unroll
          for (int e = 0; e < VE; ++e) v[i][e] = round_to<T>(v[i][e] + rs * b[e]);
          VecT<T>::store(sum_out + (size_t)row * C + (l + G * i) * VE, v[i]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) v[i][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) s += v[i][e];
    }
    const float mean = group_sum<G>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PLV; ++i)
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float d = ok[i] ? v[i][e] - mean : 0.f;
        v[i][e] = d;
        q += d * d;
      }
    const float rstd = rsqrtf(group_sum<G>(q) * invC + eps);
#pragma unroll
    for (int i = 0; i < PLV; ++i) {
      if (ok[i]) {
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] = v[i][e] * rstd * ga[i][e] + be[i][e];
        VecT<T>::store(y + (size_t)row * ldy + (l + G * i) * VE, o);
      }
    }
  }
}

// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gamma;  dgamma += sum dy * xhat, dbeta += sum dy
template <typename T, int G, int PLV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const float* __restrict__ gamma, T* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int rows, int C, int ld_dy, int ld_x, int ld_dx, float eps,
                                                                int rows_per_block, float* __restrict__ det_part,
                                                                const T* add, int ld_add, T* __restrict__ dx2,
                                                                const float* __restrict__ row_scale, int rows_per_group) {
  constexpr int VE = VecT<T>::VE, RW = 64 / G, CW = G * PLV * VE;  // channels covered by a row group
  __shared__ float part[2][4][CW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l = lane % G, g = lane / G;
  float ga[PLV][VE], ag[PLV][VE], ab[PLV][VE];
  bool ok[PLV];
#pragma unroll
  for (int i = 0; i < PLV; ++i) {
    const int c = (l + G * i) * VE;
    ok[i] = c < C;
#pragma unroll
    for (int q = 0; q < VE / 4; ++q) {
      const float4 gq = ok[i] ? *(const float4*)(gamma + c + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      ga[i][4 * q] = gq.x; ga[i][4 * q + 1] = gq.y; ga[i][4 * q + 2] = gq.z; ga[i][4 * q + 3] = gq.w;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) ag[i][e] = ab[i][e] = 0.f;
  }
  const float invC = 1.f / (float)C;
  const int r_begin = blockIdx.x * rows_per_block;
  int r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  // Software pipeline over the wave's rows: the NEXT row's x / dy vectors are requested (raw, 4 registers each) before the current row
  // is reduced, so a wave always has loads in flight - the first version issued a row's loads, waited, reduced (three dependent
  // cross-lane sums), stored and only then asked for the next row: 2.5 .. 3.3 TB/s on the 2B x 3136 x 304 token tensors.
  // (two rows ahead: 48 KB of loads in flight per CU with one row ahead left the launch at 3.9 TB/s; the raw vectors cost 4 registers each
  // and the kernel sits at two waves per SIMD either way)
  uint4 nx[2][PLV], nd[2][PLV];
  int row = r_begin + wv * RW + g;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int r = row + d * 4 * RW;
    if (r < r_end) {
#pragma unroll
      for (int i = 0; i < PLV; ++i)
        if (ok[i]) {
          nx[d][i] = VecT<T>::load_raw(x + (size_t)r * ld_x + (l + G * i) * VE);
          nd[d][i] = VecT<T>::load_raw(dy + (size_t)r * ld_dy + (l + G * i) * VE);
        }
    }
  }
  for (; row < r_end; row += 4 * RW) {
    float xv[PLV][VE], dv[PLV][VE];
    float s = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < PLV; ++i) {
      if (ok[i]) {
        VecT<T>::unpack(nx[0][i], xv[i]);
        VecT<T>::unpack(nd[0][i], dv[i]);
        nx[0][i] = nx[1][i];
        nd[0][i] = nd[1][i];
      } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) xv[i][e] = dv[i][e] = 0.f;
      }
    }
    const int nrow = row + 8 * RW;
    if (nrow < r_end) {
#pragma unroll
      for (int i = 0; i < PLV; ++i)
        if (ok[i]) {
          nx[1][i] = VecT<T>::load_raw(x + (size_t)nrow * ld_x + (l + G * i) * VE);
          nd[1][i] = VecT<T>::load_raw(dy + (size_t)nrow * ld_dy + (l + G * i) * VE);
        }
    }
    // two rounds of two independent sums: (sum x, sum dy gamma), then with the mean (sum d^2, sum dy gamma d), d = x - mean
#pragma unroll
    for (int i = 0; i < PLV; ++i)
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        s += xv[i][e];
        const float dg = dv[i][e] * ga[i][e];   // (padding lanes: gamma = 0)
        s1 += dg;
      }
    const float mean = group_sum<G>(s) * invC;
    s1 = group_sum<G>(s1) * invC;
    float q = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PLV; ++i)
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float d = ok[i] ? xv[i][e] - mean : 0.f;
        xv[i][e] = d;
        q += d * d;
        s2 = fmaf(dv[i][e] * ga[i][e], d, s2);
      }
    const float rstd = rsqrtf(group_sum<G>(q) * invC + eps);
    s2 = group_sum<G>(s2) * invC * rstd;       // mean over the row of dy gamma xhat, xhat = d rstd
#pragma unroll
    for (int i = 0; i < PLV; ++i) {
      if (ok[i]) {
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float xh = xv[i][e] * rstd;
          ag[i][e] = fmaf(dv[i][e], xh, ag[i][e]);
          ab[i][e] += dv[i][e];
          o[e] = rstd * (dv[i][e] * ga[i][e] - s1 - xh * s2);
        }
        if (add) {   // the gradient the input already holds (residual branch): one pass instead of a separate add
          float r[VE];
          VecT<T>::load(add + (size_t)row * ld_add + (l + G * i) * VE, r);
#pragma unroll
          for (int e = 0; e < VE; ++e) o[e] += r[e];
        }
        VecT<T>::store(dx + (size_t)row * ld_dx + (l + G * i) * VE, o);
        if (dx2) {   // second output: the same gradient times the factor of the row's group (DropPath: mask / keep of its image)
          const float rs = row_scale[row / rows_per_group];
#pragma unroll
          for (int e = 0; e < VE; ++e) o[e] *= rs;
          VecT<T>::store(dx2 + (size_t)row * C + (l + G * i) * VE, o);
        }
      }
    }
  }
  // combine the RW row groups of the wave (lanes l, l+G, l+2G ... hold the same channels), then the 4 waves
#pragma unroll
  for (int i = 0; i < PLV; ++i)
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float a = ag[i][e], b = ab[i][e];
#pragma unroll
      for (int o = G; o < 64; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      if (g == 0) {
        part[0][wv][(l + G * i) * VE + e] = a;
        part[1][wv][(l + G * i) * VE + e] = b;
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int st = i / C, c = i - st * C;
    const float s = (part[st][0][c] + part[st][1][c]) + (part[st][2][c] + part[st][3][c]);
    if (det_part)   // deterministic mode: [2][gridDim.x][C] partials, added in workgroup order by cavp_det_finish_kernel
      det_part[((size_t)st * gridDim.x + blockIdx.x) * C + c] = s;
    else
      atomicAdd((st == 0 ? dgamma : dbeta) + c, s);
  }
}

inline bool dt_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// (G, PLV) for C channels of VE-element vectors; 0 if C is too wide
inline bool ln_geometry(int C, int VE, int* G, int* PLV) {
  const int vecs = (C + VE - 1) / VE;
  int plv = (vecs + 63) / 64;
  if (plv == 4) plv = 5;
  if (plv > 5) return false;
  int need = (vecs + plv - 1) / plv, g = 8;
  while (g < need) g <<= 1;
  if (plv > 1) g = 64;
  // 33 .. 48 vectors (C = 304: 38 bf16 vectors, every CAVP fusion norm): 16 lanes x 3 vectors leave 21 % of the lanes idle and put
  // four rows on a wave; 64 lanes x 1 vector idled 41 % with one row per wave
  if (plv == 1 && vecs > 32 && vecs <= 48) { g = 16; plv = 3; }
  *G = g;
  *PLV = plv;
  return true;
}

}  // namespace

#define LN_DISPATCH(KERNEL_CALL)                                                     \
  switch (G * 10 + PLV) {                                                            \
    case 81: KERNEL_CALL(8, 1); break;                                               \
    case 161: KERNEL_CALL(16, 1); break;                                             \
    case 163: KERNEL_CALL(16, 3); break;                                             \
    case 321: KERNEL_CALL(32, 1); break;                                             \
    case 641: KERNEL_CALL(64, 1); break;                                             \
    case 642: KERNEL_CALL(64, 2); break;                                             \
    case 643: KERNEL_CALL(64, 3); break;                                             \
    case 645: KERNEL_CALL(64, 5); break;                                             \
    default: return CAVP_ERR_UNSUPPORTED;                                            \
  }

// bf16: 5 vectors per lane would need an 80 KiB LDS partial array in the backward; C <= 1536 covers every norm layer
#define LN_DISPATCH_BF16(KERNEL_CALL)                                                \
  switch (G * 10 + PLV) {                                                            \
    case 81: KERNEL_CALL(8, 1); break;                                               \
    case 161: KERNEL_CALL(16, 1); break;                                             \
    case 163: KERNEL_CALL(16, 3); break;                                             \
    case 321: KERNEL_CALL(32, 1); break;                                             \
    case 641: KERNEL_CALL(64, 1); break;                                             \
    case 642: KERNEL_CALL(64, 2); break;                                             \
    case 643: KERNEL_CALL(64, 3); break;                                             \
    default: return CAVP_ERR_UNSUPPORTED;                                            \
  }

extern "C" int cavp_layernorm(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y,
                              int32_t rows, int32_t C, int32_t ldx, int32_t ldy, float eps, void* stream) {
  return cavp_layernorm_residual(dtype, x, nullptr, nullptr, 0, gamma, beta, nullptr, y, rows, C, ldx, ldy, eps, stream);
}

extern "C" int cavp_layernorm_residual(int32_t dtype, const void* x, const void* branch, const float* row_scale,
                                       int32_t rows_per_group, const float* gamma, const float* beta, void* y_sum, void* y,
                                       int32_t rows, int32_t C, int32_t ldx, int32_t ldy, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || ldx < C || ldy < C) return CAVP_ERR_BAD_ARG;
  if (branch && (!row_scale || !y_sum || rows_per_group <= 0 || rows % rows_per_group)) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ldx % VE || ldy % VE) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(y) || !al16(gamma) || !al16(beta) || !al16(branch) || !al16(y_sum)) return CAVP_ERR_ALIGN;
  int G, PLV;
  if (!ln_geometry(C, VE, &G, &PLV)) return CAVP_ERR_UNSUPPORTED;
  const int rpb = 4 * (64 / G);
  long long nbl = ((long long)rows + rpb - 1) / rpb;
  if (nbl > 8192) nbl = 8192;
  const int nb = (int)nbl;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32) {
#define CALL(g, p) layernorm_vec_kernel<float, g, p><<<nb, 256, 0, s>>>((const float*)x, gamma, beta, (float*)y, rows, C, ldx, ldy, eps, (const float*)branch, row_scale, rows_per_group, (float*)y_sum)
    LN_DISPATCH(CALL)
#undef CALL
  } else {
#define CALL(g, p) layernorm_vec_kernel<bf16_t, g, p><<<nb, 256, 0, s>>>((const bf16_t*)x, gamma, beta, (bf16_t*)y, rows, C, ldx, ldy, eps, (const bf16_t*)branch, row_scale, rows_per_group, (bf16_t*)y_sum)
    LN_DISPATCH_BF16(CALL)
#undef CALL
  }
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, void* dx,
                                  float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t ld_dy, int32_t ld_x,
                                  int32_t ld_dx, float eps, void* stream) {
  return cavp_layernorm_bwd_add(dtype, dy, x, gamma, nullptr, 0, dx, nullptr, nullptr, 0, dgamma, dbeta, rows, C, ld_dy, ld_x, ld_dx,
                                eps, stream);
}

extern "C" int cavp_layernorm_bwd_add(int32_t dtype, const void* dy, const void* x, const float* gamma, const void* dx_add,
                                      int32_t ld_add, void* dx, void* dx_scaled, const float* row_scale, int32_t rows_per_group,
                                      float* dgamma, float* dbeta, int32_t rows, int32_t C, int32_t ld_dy, int32_t ld_x,
                                      int32_t ld_dx, float eps, void* stream) {
  if (dx_scaled && (!row_scale || rows_per_group <= 0 || rows % rows_per_group)) return CAVP_ERR_BAD_ARG;
  if (!al16(dx_scaled)) return CAVP_ERR_ALIGN;
  if (!dy || !x || !gamma || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype)) return CAVP_ERR_UNSUPPORTED;
  const int VE = dtype == CAVP_F32 ? 4 : 8;
  if (C % VE || ld_dy % VE || ld_x % VE || ld_dx % VE || (dx_add && ld_add % VE)) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(dy) || !al16(dx) || !al16(gamma) || !al16(dx_add)) return CAVP_ERR_ALIGN;
  int G, PLV;
  if (!ln_geometry(C, VE, &G, &PLV)) return CAVP_ERR_UNSUPPORTED;
  const int rpi = 4 * (64 / G);  // rows per workgroup iteration
  // workgroups: every one ends with 2 x C f32 atomics into the SAME dgamma / dbeta (served memory-side, ~26 ns per workgroup once
  // they queue up, tools/microbench/atomic_spread.hip): 1024 workgroups only where the tensor gives each of them >= 8 K elements
  // (the PVTv2 stage-3 / stage-4 norms are 8192 x 320 / 2048 x 512: 1024 workgroups left them 8 rows each and 26 us of queueing)
  long long want = (long long)rows * C / 8192;
  int gx = want > 1024 ? 1024 : (want < 64 ? 64 : (int)want);
  int rpb = (rows + gx - 1) / gx;
  rpb = (rpb + rpi - 1) / rpi * rpi;
  gx = (rows + rpb - 1) / rpb;
  hipStream_t s = (hipStream_t)stream;
  bool det_err;
  float* det = cavp_det_scratch(gx, C, &det_err);
  if (det_err) return CAVP_ERR_WORKSPACE;
  if (dtype == CAVP_F32) {
#define CALL(g, p) layernorm_bwd_vec_kernel<float, g, p><<<gx, 256, 0, s>>>((const float*)dy, (const float*)x, gamma, (float*)dx, dgamma, dbeta, rows, C, ld_dy, ld_x, ld_dx, eps, rpb, det, (const float*)dx_add, ld_add, (float*)dx_scaled, row_scale, rows_per_group)
    LN_DISPATCH(CALL)
#undef CALL
  } else {
#define CALL(g, p) layernorm_bwd_vec_kernel<bf16_t, g, p><<<gx, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)x, gamma, (bf16_t*)dx, dgamma, dbeta, rows, C, ld_dy, ld_x, ld_dx, eps, rpb, det, (const bf16_t*)dx_add, ld_add, (bf16_t*)dx_scaled, row_scale, rows_per_group)
    LN_DISPATCH_BF16(CALL)
#undef CALL
  }
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (det && cavp_det_finish(det, gx, C, dgamma, dbeta, s) != hipSuccess) return CAVP_ERR_LAUNCH;
  return CAVP_OK;
}
