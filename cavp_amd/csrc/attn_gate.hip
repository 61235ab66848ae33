// Sigmoid-gated attention with ONE key/value token per batch item (attn.py:73-106 with N_kv == 1: the audio token),
// forward and backward, for gfx950.
//
//   s[b,h,t] = sigmoid(scale * <q[b,t,h,:], k[b,h,:]>);  o[b,t,h,:] = s[b,h,t] * v[b,h,:];  attn[b,h,t] = s
//
// HBM-bound (q in, o out; backward: q and do in, dq out).  One wave per token row; a lane owns the two channel QUADS
// l and l + 64 (4 consecutive channels = 8 bytes of bf16 / 16 bytes of f32), so head boundaries (head_dim 76, 28 ...:
// multiples of 4, not of 8) fall between lanes and every access is one wide load or store - the first version used
// 2-byte accesses, 5 per operand per row.  The per-head dot products are masked wave reductions (<= 8 heads).
#include "common.h"

namespace {

template <typename T> struct Quad;
template <> struct Quad<float> {
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Quad<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    const uint2 t = *(const uint2*)p;
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    uint2 t;
    t.x = pack2bf(v[0], v[1]);
    t.y = pack2bf(v[2], v[3]);
    *(uint2*)p = t;
  }
};

constexpr int kMaxHeads = 8;

template <typename T>
__global__ __launch_bounds__(256) void attn_gate_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, T* __restrict__ o,
                                                        float* __restrict__ attn, int B, int Tn, int heads, int hd,
                                                        float scale, long long q_rows) {
  const int lane = threadIdx.x & 63;
  const int C = heads * hd, NQ = C >> 2, qh = hd >> 2;  // quads per row / per head
  const bool ok0 = lane < NQ, ok1 = lane + 64 < NQ;
  const int h0 = ok0 ? lane / qh : -1, h1 = ok1 ? (lane + 64) / qh : -1;
  const long long rows = (long long)B * Tn;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / Tn), t = (int)(row - (long long)b * Tn);
    const T* qp = q + (size_t)(row >= q_rows ? row % q_rows : row) * C;   // q_batch < B: the query rows repeat
    const T* kp = k + (size_t)b * C;
    const T* vp = v + (size_t)b * C;
    float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f}, k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f};
    float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok0) { Quad<T>::load(qp + 4 * lane, q0); Quad<T>::load(kp + 4 * lane, k0); Quad<T>::load(vp + 4 * lane, v0); }
    if (ok1) { Quad<T>::load(qp + 4 * (lane + 64), q1); Quad<T>::load(kp + 4 * (lane + 64), k1); Quad<T>::load(vp + 4 * (lane + 64), v1); }
    const float d0 = q0[0] * k0[0] + q0[1] * k0[1] + q0[2] * k0[2] + q0[3] * k0[3];
    const float d1 = q1[0] * k1[0] + q1[1] * k1[1] + q1[2] * k1[2] + q1[3] * k1[3];
    float g0 = 0.f, g1 = 0.f;
    for (int h = 0; h < heads; ++h) {
      const float s = wave_sum((h0 == h ? d0 : 0.f) + (h1 == h ? d1 : 0.f)) * scale;
      const float g = 1.f / (1.f + expf(-s));
      if (lane == 0) attn[((size_t)b * heads + h) * Tn + t] = g;
      g0 = h0 == h ? g : g0;
      g1 = h1 == h ? g : g1;
    }
    T* op = o + (size_t)row * C;
    if (ok0) {
      const float r[4] = {g0 * v0[0], g0 * v0[1], g0 * v0[2], g0 * v0[3]};
      Quad<T>::store(op + 4 * lane, r);
    }
    if (ok1) {
      const float r[4] = {g1 * v1[0], g1 * v1[1], g1 * v1[2], g1 * v1[3]};
      Quad<T>::store(op + 4 * (lane + 64), r);
    }
  }
}

//   ds_h = <do_h, v_h> (+ dattn);  da = ds * s (1 - s);  dq = da * scale * k;  dk += da * scale * q;  dv += s * do
// grid = (token chunks, B); dk / dv: register accumulators over the workgroup's tokens -> LDS over the 4 waves -> one
// f32 atomic per channel per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void attn_gate_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                            const T* __restrict__ k, const T* __restrict__ v,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ dattn, T* __restrict__ dq,
                                                            float* __restrict__ dk, float* __restrict__ dv, int Tn,
                                                            int heads, int hd, float scale, int tok_per_block,
                                                            int q_batch, float* __restrict__ det_part) {
  __shared__ float part[2][4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.y;
  const int C = heads * hd, NQ = C >> 2, qh = hd >> 2;
  const bool ok0 = lane < NQ, ok1 = lane + 64 < NQ;
  const int h0 = ok0 ? lane / qh : -1, h1 = ok1 ? (lane + 64) / qh : -1;
  float k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f}, v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok0) { Quad<T>::load(k + (size_t)b * C + 4 * lane, k0); Quad<T>::load(v + (size_t)b * C + 4 * lane, v0); }
  if (ok1) { Quad<T>::load(k + (size_t)b * C + 4 * (lane + 64), k1); Quad<T>::load(v + (size_t)b * C + 4 * (lane + 64), v1); }
  float adk0[4] = {0.f, 0.f, 0.f, 0.f}, adk1[4] = {0.f, 0.f, 0.f, 0.f}, adv0[4] = {0.f, 0.f, 0.f, 0.f}, adv1[4] = {0.f, 0.f, 0.f, 0.f};
  const int t_begin = blockIdx.x * tok_per_block;
  int t_end = t_begin + tok_per_block;
  if (t_end > Tn) t_end = Tn;
  for (int t = t_begin + wv; t < t_end; t += 4) {
    const size_t row = ((size_t)b * Tn + t) * C;
    const size_t qrow = ((size_t)(b % q_batch) * Tn + t) * C;   // q_batch < B: the query rows repeat
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok0) { Quad<T>::load(dout + row + 4 * lane, d0); Quad<T>::load(q + qrow + 4 * lane, q0); }
    if (ok1) { Quad<T>::load(dout + row + 4 * (lane + 64), d1); Quad<T>::load(q + qrow + 4 * (lane + 64), q1); }
    const float p0 = d0[0] * v0[0] + d0[1] * v0[1] + d0[2] * v0[2] + d0[3] * v0[3];
    const float p1 = d1[0] * v1[0] + d1[1] * v1[1] + d1[2] * v1[2] + d1[3] * v1[3];
    float da0 = 0.f, da1 = 0.f, g0 = 0.f, g1 = 0.f;
    for (int h = 0; h < heads; ++h) {
      float s = wave_sum((h0 == h ? p0 : 0.f) + (h1 == h ? p1 : 0.f));
      const size_t ai = ((size_t)b * heads + h) * Tn + t;
      if (dattn) s += dattn[ai];
      const float g = attn[ai];
      const float da = s * g * (1.f - g) * scale;
      da0 = h0 == h ? da : da0; g0 = h0 == h ? g : g0;
      da1 = h1 == h ? da : da1; g1 = h1 == h ? g : g1;
    }
    if (ok0) {
      const float r[4] = {da0 * k0[0], da0 * k0[1], da0 * k0[2], da0 * k0[3]};
      Quad<T>::store(dq + row + 4 * lane, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) { adk0[e] += da0 * q0[e]; adv0[e] += g0 * d0[e]; }
    }
    if (ok1) {
      const float r[4] = {da1 * k1[0], da1 * k1[1], da1 * k1[2], da1 * k1[3]};
      Quad<T>::store(dq + row + 4 * (lane + 64), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) { adk1[e] += da1 * q1[e]; adv1[e] += g1 * d1[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    part[0][wv][4 * lane + e] = adk0[e];
    part[1][wv][4 * lane + e] = adv0[e];
    part[0][wv][4 * (lane + 64) + e] = adk1[e];
    part[1][wv][4 * (lane + 64) + e] = adv1[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int st = i / C, c = i - st * C;
    const float s = (part[st][0][c] + part[st][1][c]) + (part[st][2][c] + part[st][3][c]);
    if (det_part)   // deterministic mode: [2][gridDim.x][B * C] partials, added in token-chunk order afterwards
      det_part[((size_t)st * gridDim.x + blockIdx.x) * ((size_t)gridDim.y * C) + (size_t)b * C + c] = s;
    else
      atomicAdd((st == 0 ? dk : dv) + (size_t)b * C + c, s);
  }
}

// ---- 4-head variants (every CAVP configuration: 4 x 76 and 4 x 28 channels) ------------------------------------------------
// A 16-lane DPP row owns one head: lane (g, i) holds the head's channel quads i and i + 16, so the four per-head dot products are
// ONE row reduction (4 DPP steps) instead of `heads` masked 64-lane reductions (24 steps) per token row.
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_gate4_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, T* __restrict__ o,
                                                         float* __restrict__ attn, int B, int Tn, int hd, float scale,
                                                         long long q_rows) {
  const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
  const int C = 4 * hd, qh = hd >> 2;
  const bool ok0 = i < qh, ok1 = i + 16 < qh;
  const int c0 = g * hd + 4 * i, c1 = c0 + 64;
  const long long rows = (long long)B * Tn;
  for (long long row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / Tn), t = (int)(row - (long long)b * Tn);
    const T* qp = q + (size_t)(row >= q_rows ? row % q_rows : row) * C;   // q_batch < B: the query rows repeat
    const T* kp = k + (size_t)b * C;
    const T* vp = v + (size_t)b * C;
    float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f}, k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f};
    float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok0) { Quad<T>::load(qp + c0, q0); Quad<T>::load(kp + c0, k0); Quad<T>::load(vp + c0, v0); }
    if (ok1) { Quad<T>::load(qp + c1, q1); Quad<T>::load(kp + c1, k1); Quad<T>::load(vp + c1, v1); }
    const float d = (q0[0] * k0[0] + q0[1] * k0[1] + q0[2] * k0[2] + q0[3] * k0[3]) +
                    (q1[0] * k1[0] + q1[1] * k1[1] + q1[2] * k1[2] + q1[3] * k1[3]);
    const float gt = 1.f / (1.f + expf(-row_sum16(d) * scale));
    if (i == 0) attn[((size_t)b * 4 + g) * Tn + t] = gt;
    T* op = o + (size_t)row * C;
    if (ok0) {
      const float r[4] = {gt * v0[0], gt * v0[1], gt * v0[2], gt * v0[3]};
      Quad<T>::store(op + c0, r);
    }
    if (ok1) {
      const float r[4] = {gt * v1[0], gt * v1[1], gt * v1[2], gt * v1[3]};
      Quad<T>::store(op + c1, r);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_gate4_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                             const T* __restrict__ k, const T* __restrict__ v,
                                                             const float* __restrict__ attn,
                                                             const float* __restrict__ dattn, T* __restrict__ dq,
                                                             float* __restrict__ dk, float* __restrict__ dv, int Tn, int hd,
                                                             float scale, int tok_per_block, int q_batch,
                                                             float* __restrict__ det_part) {
  __shared__ float part[2][4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.y, g = lane >> 4, i = lane & 15;
  const int C = 4 * hd, qh = hd >> 2;
  const bool ok0 = i < qh, ok1 = i + 16 < qh;
  const int c0 = g * hd + 4 * i, c1 = c0 + 64;
  float k0[4] = {0.f, 0.f, 0.f, 0.f}, k1[4] = {0.f, 0.f, 0.f, 0.f}, v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok0) { Quad<T>::load(k + (size_t)b * C + c0, k0); Quad<T>::load(v + (size_t)b * C + c0, v0); }
  if (ok1) { Quad<T>::load(k + (size_t)b * C + c1, k1); Quad<T>::load(v + (size_t)b * C + c1, v1); }
  float adk0[4] = {0.f, 0.f, 0.f, 0.f}, adk1[4] = {0.f, 0.f, 0.f, 0.f}, adv0[4] = {0.f, 0.f, 0.f, 0.f}, adv1[4] = {0.f, 0.f, 0.f, 0.f};
  const int t_begin = blockIdx.x * tok_per_block;
  int t_end = t_begin + tok_per_block;
  if (t_end > Tn) t_end = Tn;
  for (int t = t_begin + wv; t < t_end; t += 4) {
    const size_t row = ((size_t)b * Tn + t) * C;
    const size_t qrow = ((size_t)(b % q_batch) * Tn + t) * C;   // q_batch < B: the query rows repeat
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4] = {0.f, 0.f, 0.f, 0.f}, q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok0) { Quad<T>::load(dout + row + c0, d0); Quad<T>::load(q + qrow + c0, q0); }
    if (ok1) { Quad<T>::load(dout + row + c1, d1); Quad<T>::load(q + qrow + c1, q1); }
    const size_t ai = ((size_t)b * 4 + g) * Tn + t;
    const float gt = attn[ai];
    float sdot = row_sum16((d0[0] * v0[0] + d0[1] * v0[1] + d0[2] * v0[2] + d0[3] * v0[3]) +
                           (d1[0] * v1[0] + d1[1] * v1[1] + d1[2] * v1[2] + d1[3] * v1[3]));
    if (dattn) sdot += dattn[ai];
    const float da = sdot * gt * (1.f - gt) * scale;
    if (ok0) {
      const float r[4] = {da * k0[0], da * k0[1], da * k0[2], da * k0[3]};
      Quad<T>::store(dq + row + c0, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) { adk0[e] += da * q0[e]; adv0[e] += gt * d0[e]; }
    }
    if (ok1) {
      const float r[4] = {da * k1[0], da * k1[1], da * k1[2], da * k1[3]};
      Quad<T>::store(dq + row + c1, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) { adk1[e] += da * q1[e]; adv1[e] += gt * d1[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (ok0) { part[0][wv][c0 + e] = adk0[e]; part[1][wv][c0 + e] = adv0[e]; }
    if (ok1) { part[0][wv][c1 + e] = adk1[e]; part[1][wv][c1 + e] = adv1[e]; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    const int st = j / C, c = j - st * C;
    const float s = (part[st][0][c] + part[st][1][c]) + (part[st][2][c] + part[st][3][c]);
    if (det_part)
      det_part[((size_t)st * gridDim.x + blockIdx.x) * ((size_t)gridDim.y * C) + (size_t)b * C + c] = s;
    else
      atomicAdd((st == 0 ? dk : dv) + (size_t)b * C + c, s);
  }
}

inline bool dt_ok(int dt) { return dt == CAVP_F32 || dt == CAVP_BF16; }
inline bool shape_ok(int dtype, int heads, int hd, const void* a, const void* b, const void* c, const void* d) {
  const uintptr_t al = dtype == CAVP_F32 ? 15 : 7;
  return heads <= kMaxHeads && (hd % 4) == 0 && heads * hd <= 512 &&
         !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & al);
}

}  // namespace

extern "C" int cavp_attn_gate(int32_t dtype, const void* q, const void* k, const void* v, void* o, float* attn,
                              int32_t B, int32_t T, int32_t heads, int32_t hd, float scale, int32_t q_batch, void* stream) {
  if (!q || !k || !v || !o || !attn || B <= 0 || T <= 0 || heads <= 0 || hd <= 0 || q_batch <= 0 || B % q_batch) return CAVP_ERR_BAD_ARG;
  const long long q_rows = (long long)q_batch * T;
  if (!dt_ok(dtype) || !shape_ok(dtype, heads, hd, q, k, v, o)) return CAVP_ERR_UNSUPPORTED;
  long long nbl = ((long long)B * T + 3) / 4;
  if (nbl > 16384) nbl = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (heads == 4 && hd <= 128) {   // one DPP row per head
    if (dtype == CAVP_F32)
      attn_gate4_kernel<float><<<(int)nbl, 256, 0, s>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, attn, B, T, hd, scale, q_rows);
    else
      attn_gate4_kernel<bf16_t><<<(int)nbl, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, attn, B, T, hd, scale, q_rows);
    return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
  }
  if (dtype == CAVP_F32)
    attn_gate_kernel<float><<<(int)nbl, 256, 0, s>>>((const float*)q, (const float*)k, (const float*)v, (float*)o, attn, B, T, heads, hd, scale, q_rows);
  else
    attn_gate_kernel<bf16_t><<<(int)nbl, 256, 0, s>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, attn, B, T, heads, hd, scale, q_rows);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_attn_gate_bwd(int32_t dtype, const void* dout, const void* q, const void* k, const void* v,
                                  const float* attn, const float* dattn, void* dq, float* dk, float* dv, int32_t B,
                                  int32_t T, int32_t heads, int32_t hd, float scale, int32_t q_batch, void* stream) {
  if (!dout || !q || !k || !v || !attn || !dq || !dk || !dv || B <= 0 || T <= 0 || heads <= 0 || hd <= 0 || q_batch <= 0 ||
      B % q_batch)
    return CAVP_ERR_BAD_ARG;
  if (!dt_ok(dtype) || !shape_ok(dtype, heads, hd, q, k, v, dq) || !shape_ok(dtype, heads, hd, dout, dout, dout, dout))
    return CAVP_ERR_UNSUPPORTED;
  int gx = 2048 / B;
  if (gx < 1) gx = 1;
  int tpb = (T + gx - 1) / gx;
  if (tpb < 4) tpb = 4;
  tpb = (tpb + 3) / 4 * 4;
  gx = (T + tpb - 1) / tpb;
  hipStream_t s = (hipStream_t)stream;
  bool det_err;
  float* det = cavp_det_scratch(gx, B * heads * hd, &det_err);
  if (det_err) return CAVP_ERR_WORKSPACE;
  if (heads == 4 && hd <= 128) {
    if (dtype == CAVP_F32)
      attn_gate4_bwd_kernel<float><<<dim3(gx, B), 256, 0, s>>>((const float*)dout, (const float*)q, (const float*)k, (const float*)v, attn, dattn, (float*)dq, dk, dv, T, hd, scale, tpb, q_batch, det);
    else
      attn_gate4_bwd_kernel<bf16_t><<<dim3(gx, B), 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, attn, dattn, (bf16_t*)dq, dk, dv, T, hd, scale, tpb, q_batch, det);
  } else if (dtype == CAVP_F32)
    attn_gate_bwd_kernel<float><<<dim3(gx, B), 256, 0, s>>>((const float*)dout, (const float*)q, (const float*)k, (const float*)v, attn, dattn, (float*)dq, dk, dv, T, heads, hd, scale, tpb, q_batch, det);
  else
    attn_gate_bwd_kernel<bf16_t><<<dim3(gx, B), 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, attn, dattn, (bf16_t*)dq, dk, dv, T, heads, hd, scale, tpb, q_batch, det);
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  if (det && cavp_det_finish(det, gx, B * heads * hd, dk, dv, s) != hipSuccess) return CAVP_ERR_LAUNCH;
  return CAVP_OK;
}
