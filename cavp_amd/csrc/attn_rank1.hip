// Cross-modal attention with ONE key / value token per batch item - the only way CAVP calls it (attn.py:73-106 with
// x_k = x_v = the LayerNorm'ed audio token, cavp_model.py:145-149; 4 heads, no q / k / v bias, sigmoid instead of softmax):
//
//   q = x Wq^T          s[t,h] = scale * q[t,h,:] . k[h,:]        g = sigmoid(s)
//   o[t,h,:] = g[t,h] * v[h,:]          out = x + o Wp^T + bp      (the residual is the query input itself, attn.py:153-156)
//
// Round 3 ran this as written: a 304 x 304 GEMM over all tokens for q, the gate kernel, a second 304 x 304 GEMM for the output
// projection (+ their data- and weight-gradient GEMMs): ~0.67 ms of the 15 ms training step and eight passes over
// [tokens x 304] tensors.  With one key per batch item both GEMMs collapse to rank-H operations:
//
//   u[b,h,:] = scale * Wq[h-slice,:]^T k[b,h-slice]      p[b,h,:] = Wp[:,h-slice] v[b,h-slice]         (H x C per batch item)
//   g[b,t,h] = sigmoid(x[t,:] . u[b,h,:])                out[b,t,:] = x[t,:] + bp + sum_h g[b,t,h] p[b,h,:]
//
// i.e. ONE pass over the tokens forward (read x, write out) and one backward (read dout and x, write dx) with 2 H dot products
// and 2 H axpys of length C per token, and the parameter gradients become sums over batch items of outer products of H x C
// matrices (cavp_attn1_finish).  Same math as the reference up to floating-point association (f32 accumulation throughout).
//
// Layout: a token row is handled by one wave, lane i < C / 8 owns channels 8 i .. 8 i + 7 (C <= 512, C % 8 == 0); u and p
// slices live in registers (2 x H x 8 floats per lane); dot products are reduced with DPP row sums + two butterfly steps.
#include "igemm_params.h"   // row16_sum

namespace {

template <typename T> __device__ __forceinline__ void ld8(const T* p, float* v);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* v) {
  VecT<float>::load(p, v);
  VecT<float>::load(p + 4, v + 4);
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* v) { VecT<bf16_t>::load(p, v); }
template <typename T> __device__ __forceinline__ void st8(T* p, const float* v);
template <> __device__ __forceinline__ void st8<float>(float* p, const float* v) {
  VecT<float>::store(p, v);
  VecT<float>::store(p + 4, v + 4);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* v) { VecT<bf16_t>::store(p, v); }

// 8 channels as loaded (16 bytes of bf16 / 32 bytes of f32): the prefetch rings hold rows in this form, unpacked at use
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
  __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  __device__ __forceinline__ void unpack(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw8<bf16_t> {
  uint4 a;
  __device__ __forceinline__ void zero() { a = make_uint4(0u, 0u, 0u, 0u); }
  __device__ __forceinline__ void load(const bf16_t* p) { a = *(const uint4*)p; }
  __device__ __forceinline__ void unpack(float* v) const {
    const unsigned u[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(u[i] << 16);
      v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
};

__device__ __forceinline__ void ldf8(const float* p, float* v) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void stf8(float* p, const float* v) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// sum over the 64 lanes, every lane gets the total: DPP sums inside the four 16-lane rows, then the four row totals are read
// through the scalar unit (v_readlane) - no LDS permute (two ds_bpermute round trips per reduction, eight reductions per
// token row in the backward, were a third of that kernel's time)
__device__ __forceinline__ float wave_total(float v) {
  v = row16_sum(v);
  const int i = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return (r0 + r1) + (r2 + r3);
}

// 1 / (1 + e^-s) on the hardware exp2 / rcp (|relative error| ~ 1e-6: below the f32 accumulation noise of the 304-term dot product)
__device__ __forceinline__ float sigmoidf_(float s) { return __frcp_rn(1.f + __expf(-s)); }

// u[b,h,i] = scale * sum_j Wq[h d + j, i] k[b, h d + j];  p[b,h,o] = sum_j Wp[o, h d + j] v[b, h d + j].  One workgroup per (b, h).
template <typename T>
__global__ __launch_bounds__(256) void attn1_prepare_kernel(const float* __restrict__ wq, const float* __restrict__ wp,
                                                            const T* __restrict__ k, const T* __restrict__ v, float* __restrict__ U,
                                                            float* __restrict__ P, int C, int H, float scale) {
  __shared__ float ks[128], vs[128];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, d = C / H;
  for (int j = threadIdx.x; j < d; j += 256) {
    ks[j] = scale * Elem<T>::ld(k + (size_t)b * C + h * d + j);
    vs[j] = Elem<T>::ld(v + (size_t)b * C + h * d + j);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    // (4 independent accumulators: a single fmaf chain waits for one load per step - 76 dependent L2 round trips per thread)
    const float* w = wq + (size_t)(h * d) * C + i;
    const float* wr = wp + (size_t)i * C + h * d;
    float u[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 4 <= d; j += 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[e] = fmaf(w[(size_t)(j + e) * C], ks[j + e], u[e]);
        pp[e] = fmaf(wr[j + e], vs[j + e], pp[e]);
      }
    }
    for (; j < d; ++j) {
      u[0] = fmaf(w[(size_t)j * C], ks[j], u[0]);
      pp[0] = fmaf(wr[j], vs[j], pp[0]);
    }
    U[((size_t)b * H + h) * C + i] = (u[0] + u[1]) + (u[2] + u[3]);
    P[((size_t)b * H + h) * C + i] = (pp[0] + pp[1]) + (pp[2] + pp[3]);
  }
}

// out[b,t,:] = x[b % xb, t, :] + bp + sum_h sigmoid(x . u[b,h]) p[b,h];  attn[b,h,t] = the gate.  grid (token groups, B).
template <typename T, int H>
__global__ __launch_bounds__(256, 4) void attn1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ U, const float* __restrict__ P,
                                                        const float* __restrict__ bp, T* __restrict__ out, float* __restrict__ attn, int xb,
                                                        int Tn, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const bool act = lane < (C >> 3);
  const int c0 = lane * 8;
  float u[H][8], pv[H][8], bias[8];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) { u[h][e] = 0.f; pv[h][e] = 0.f; }
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = 0.f;
  if (act) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      ldf8(U + ((size_t)b * H + h) * C + c0, u[h]);
      ldf8(P + ((size_t)b * H + h) * C + c0, pv[h]);
    }
    if (bp) ldf8(bp + c0, bias);
  }
  const T* xr = x + (size_t)(b % xb) * Tn * C;
  T* orow = out + (size_t)b * Tn * C;
  float* ar = attn + (size_t)b * H * Tn;
  // A token row per trip is a long dependent chain (load, dot, reduce, exp): a ring of NP rows keeps NP - 1 loads in flight per
  // wave (with one the kernel ran at the HBM latency per row: 1.8 us), four waves per SIMD hide the rest.
  constexpr int NP = 4;
  const int step = gridDim.x * 4, t0 = blockIdx.x * 4 + wave;
  Raw8<T> ring[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    ring[i].zero();
    if (act && t0 + i * step < Tn) ring[i].load(xr + (size_t)(t0 + i * step) * C + c0);
  }
  for (int base = t0; base < Tn; base += NP * step) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int t = base + i * step;
      if (t >= Tn) break;
      float xv[8];
      ring[i].unpack(xv);
      if (act && t + NP * step < Tn) ring[i].load(xr + (size_t)(t + NP * step) * C + c0);
      float g[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(xv[e], u[h][e], s);
        g[h] = sigmoidf_(wave_total(s));
      }
      if (act) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = xv[e] + bias[e];
#pragma unroll
          for (int h = 0; h < H; ++h) a = fmaf(g[h], pv[h][e], a);
          y[e] = a;
        }
        st8<T>(orow + (size_t)t * C + c0, y);
      }
      float gl = g[0];
#pragma unroll
      for (int h = 1; h < H; ++h) gl = lane == h ? g[h] : gl;
      if (lane < H) ar[(size_t)lane * Tn + t] = gl;
    }
  }
}

// Backward over the tokens.  grid (token groups, xb): workgroup (gx, bb) walks tokens gx*4 + wave, ... of base item bb for every
// batch item b = bb + r xb that shares its x rows (forward_train duplicates the images: r = 0, 1), so dx - the sum over r - is
// accumulated by the wave that owns the row.  Per (b, workgroup) the partial sums dU = sum_t ds x, dP = sum_t g dout go to slabs
// (combined over the 4 waves through LDS first), dbp partials to slabB; fixed-order sums in attn1_slabsum_kernel (deterministic).
template <typename T, int H>
__global__ __launch_bounds__(256, 2) void attn1_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ U,
                                                           const float* __restrict__ P, T* __restrict__ dx, float* __restrict__ slabU,
                                                           float* __restrict__ slabP, float* __restrict__ slabB, int B, int xb, int Tn, int C) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][2][H][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bb = blockIdx.y;
  const bool act = lane < (C >> 3);
  const int c0 = lane * 8, reps = B / xb, nslots = gridDim.x;
  float dbp[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) dbp[e] = 0.f;
  const T* xr = x + (size_t)bb * Tn * C;
  T* dxr = dx + (size_t)bb * Tn * C;
  for (int r = 0; r < reps; ++r) {
    const int b = bb + r * xb;
    float u[H][8], pv[H][8], dU[H][8], dP[H][8];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) { u[h][e] = 0.f; pv[h][e] = 0.f; dU[h][e] = 0.f; dP[h][e] = 0.f; }
    if (act) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        ldf8(U + ((size_t)b * H + h) * C + c0, u[h]);
        ldf8(P + ((size_t)b * H + h) * C + c0, pv[h]);
      }
    }
    const T* dyr = dout + (size_t)b * Tn * C;
    constexpr int NP = 3;   // rows in flight per wave (x and dout: 2 x NP x 8 registers)
    const int step = gridDim.x * 4, t0 = blockIdx.x * 4 + wave;
    Raw8<T> rx[NP], rd[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      rx[i].zero();
      rd[i].zero();
      if (act && t0 + i * step < Tn) {
        rx[i].load(xr + (size_t)(t0 + i * step) * C + c0);
        rd[i].load(dyr + (size_t)(t0 + i * step) * C + c0);
      }
    }
    for (int base = t0; base < Tn; base += NP * step) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int t = base + i * step;
        if (t >= Tn) break;
        float xv[8], dy[8];
        rx[i].unpack(xv);
        rd[i].unpack(dy);
        if (act && t + NP * step < Tn) {   // refill the slot: NP - 1 rows stay in flight while this one is reduced
          rx[i].load(xr + (size_t)(t + NP * step) * C + c0);
          rd[i].load(dyr + (size_t)(t + NP * step) * C + c0);
        }
        float g[H], ds[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s = fmaf(xv[e], u[h][e], s);
            q = fmaf(dy[e], pv[h][e], q);
          }
          g[h] = sigmoidf_(wave_total(s));
          ds[h] = wave_total(q) * g[h] * (1.f - g[h]);
        }
        if (act) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = dy[e];                      // the residual path (out = x + ...)
#pragma unroll
            for (int h = 0; h < H; ++h) a = fmaf(ds[h], u[h][e], a);
            o[e] = a;
          }
          if (r > 0) {                            // this wave wrote the row for the previous r: read-modify-write
            float old[8];
            ld8<T>(dxr + (size_t)t * C + c0, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += old[e];
          }
          st8<T>(dxr + (size_t)t * C + c0, o);
#pragma unroll
          for (int h = 0; h < H; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              dU[h][e] = fmaf(ds[h], xv[e], dU[h][e]);
              dP[h][e] = fmaf(g[h], dy[e], dP[h][e]);
            }
#pragma unroll
          for (int e = 0; e < 8; ++e) dbp[e] += dy[e];
        }
      }
    }
    // combine the 4 waves, then one slab row per (b, workgroup)
    if (r > 0) __syncthreads();
    if (act) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        stf8(red + ((size_t)(wave * 2 + 0) * H + h) * C + c0, dU[h]);
        stf8(red + ((size_t)(wave * 2 + 1) * H + h) * C + c0, dP[h]);
      }
    }
    __syncthreads();
    const int n = 2 * H * C;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float s = (red[i] + red[n + i]) + (red[2 * n + i] + red[3 * n + i]);
      const int which = i / (H * C), rest = i - which * (H * C);
      (which ? slabP : slabU)[((size_t)b * nslots + blockIdx.x) * (H * C) + rest] = s;
    }
  }
  __syncthreads();
  if (act) stf8(red + (size_t)wave * C + c0, dbp);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256)
    slabB[((size_t)bb * nslots + blockIdx.x) * C + i] = (red[i] + red[C + i]) + (red[2 * C + i] + red[3 * C + i]);
}

// dU[b,h,:] = sum_slots slabU, dP likewise (blocks 0 .. nb_main-1: one thread per element, the slot loads independent);
// dbp[:] += sum over (bb, slots) of slabB (the remaining blocks: one WAVE per channel, lanes stride over the rows).  Fixed order.
__global__ __launch_bounds__(256) void attn1_slabsum_kernel(const float* __restrict__ slabU, const float* __restrict__ slabP,
                                                            const float* __restrict__ slabB, float* __restrict__ dU, float* __restrict__ dP,
                                                            float* __restrict__ dbp, int B, int xb, int nslots, int HC, int C, int nb_main) {
  const int total = B * HC;
  if ((int)blockIdx.x < nb_main) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * total; i += nb_main * 256) {
      const int which = i / total, j = i - which * total, b = j / HC, rest = j - b * HC;
      const float* s = (which ? slabP : slabU) + (size_t)b * nslots * HC + rest;
      float a = 0.f;
      int z = 0;
      for (; z + 4 <= nslots; z += 4) {
        const float v0 = s[(size_t)z * HC], v1 = s[(size_t)(z + 1) * HC], v2 = s[(size_t)(z + 2) * HC], v3 = s[(size_t)(z + 3) * HC];
        a += v0; a += v1; a += v2; a += v3;
      }
      for (; z < nslots; ++z) a += s[(size_t)z * HC];
      (which ? dP : dU)[j] = a;
    }
  } else if (dbp) {
    const int lane = threadIdx.x & 63, rows = xb * nslots;
    for (int c = ((int)blockIdx.x - nb_main) * 4 + (threadIdx.x >> 6); c < C; c += ((int)gridDim.x - nb_main) * 4) {
      float a = 0.f;
      for (int z = lane; z < rows; z += 64) a += slabB[(size_t)z * C + c];
      a = wave_total(a);
      if (lane == 0) dbp[c] += a;
    }
  }
}

// Parameter / key / value gradients from the per-item H x C sums:
//   dWq[r, i] += scale * sum_b k[b, r] dU[b, r / d, i]          (blocks 0 .. C-1, one per row r)
//   dWp[o, r] += sum_b dP[b, r / d, o] v[b, r]                   (blocks C .. 2C-1, one per row o)
//   dk[b, r]   = scale * sum_i Wq[r, i] dU[b, r / d, i];  dv[b, r] = sum_o Wp[o, r] dP[b, r / d, o]     (blocks 2C .. 2C+B*H-1)
template <typename T>
__global__ __launch_bounds__(256) void attn1_finish_kernel(const float* __restrict__ wq, const float* __restrict__ wp, const T* __restrict__ k,
                                                           const T* __restrict__ v, const float* __restrict__ dU, const float* __restrict__ dP,
                                                           float* __restrict__ dwq, float* __restrict__ dwp, float* __restrict__ dk,
                                                           float* __restrict__ dv, int B, int C, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = C / H, blk = blockIdx.x;
  if (blk < C) {
    const int r = blk, h = r / d;
    for (int b = threadIdx.x; b < B; b += 256) sm[b] = scale * Elem<T>::ld(k + (size_t)b * C + r);
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      const float* du = dU + (size_t)h * C + i;
      const size_t bs = (size_t)H * C;
      float a = 0.f;
      int b = 0;
      for (; b + 8 <= B; b += 8) {   // (8 independent loads per trip; summed in batch order)
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = du[(size_t)(b + e) * bs];
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(sm[b + e], vv[e], a);
      }
      for (; b < B; ++b) a = fmaf(sm[b], du[(size_t)b * bs], a);
      dwq[(size_t)r * C + i] += a;
    }
  } else if (blk < 2 * C) {
    const int o = blk - C;
    for (int j = threadIdx.x; j < B * H; j += 256) sm[j] = dP[(size_t)j * C + o];   // [b][h]
    __syncthreads();
    for (int r = threadIdx.x; r < C; r += 256) {
      const int h = r / d;
      float a = 0.f;
      int b = 0;
      for (; b + 8 <= B; b += 8) {
        float vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = Elem<T>::ld(v + (size_t)(b + e) * C + r);
#pragma unroll
        for (int e = 0; e < 8; ++e) a = fmaf(sm[(b + e) * H + h], vv[e], a);
      }
      for (; b < B; ++b) a = fmaf(sm[b * H + h], Elem<T>::ld(v + (size_t)b * C + r), a);
      dwp[(size_t)o * C + r] += a;
    }
  } else {
    // (b, h): dk / dv of one head slice.  dk: a wave per row r (lanes stride over i: coalesced reads of Wq's row); dv: thread
    // (group, r) with three groups striding over o (76 contiguous floats of Wp per o), combined through LDS
    const int bh = blk - 2 * C, b = bh / H, h = bh - b * H;
    float* su = sm;            // dU[b,h,:]
    float* sp = sm + C;        // dP[b,h,:]
    float* part = sm + 2 * C;  // [3][d]
    for (int i = threadIdx.x; i < C; i += 256) {
      su[i] = dU[((size_t)b * H + h) * C + i];
      sp[i] = dP[((size_t)b * H + h) * C + i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < d; j += 8) {   // two rows per trip (independent load chains)
      const int j2 = j + 4;
      const float* w = wq + (size_t)(h * d + j) * C;
      const float* w2 = wq + (size_t)(h * d + (j2 < d ? j2 : j)) * C;
      float a = 0.f, a2 = 0.f;
      for (int i = lane; i < C; i += 64) {
        a = fmaf(w[i], su[i], a);
        a2 = fmaf(w2[i], su[i], a2);
      }
      a = wave_total(a);
      a2 = wave_total(a2);
      if (lane == 0) {
        dk[(size_t)b * C + h * d + j] = scale * a;
        if (j2 < d) dk[(size_t)b * C + h * d + j2] = scale * a2;
      }
    }
    const int ng = 256 / d < 3 ? 256 / d : 3;   // (d <= 128)
    const int grp = threadIdx.x / d, j = threadIdx.x - grp * d;
    if (grp < ng) {
      // (8 loads in flight per thread: a single fmaf chain over C / ng rows of Wp was ~100 dependent L2 round trips, 70 us)
      float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float* w = wp + h * d + j;
      int o = grp;
      for (; o + 7 * ng < C; o += 8 * ng) {
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = w[(size_t)(o + e * ng) * C];
#pragma unroll
        for (int e = 0; e < 8; ++e) c[e] = fmaf(wv[e], sp[o + e * ng], c[e]);
      }
      for (; o < C; o += ng) c[0] = fmaf(w[(size_t)o * C], sp[o], c[0]);
      part[grp * d + j] = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    }
    __syncthreads();
    if (threadIdx.x < d) {
      float c = part[threadIdx.x];
      for (int q = 1; q < ng; ++q) c += part[q * d + threadIdx.x];
      dv[(size_t)b * C + h * d + threadIdx.x] = c;
    }
  }
}

inline bool ok_shape(int C, int heads) { return C > 0 && C <= 512 && C % 8 == 0 && heads == 4 && C % heads == 0 && C / heads <= 128; }
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline int token_groups(int Tn, int batches, int waves = 2048) {   // workgroups (4 token rows each per trip) per batch item
  int g = (waves / 4 + batches - 1) / batches;
  if (g < 1) g = 1;
  const int need = (Tn + 3) / 4;
  return g < need ? g : need;
}

}  // namespace

extern "C" int cavp_attn1_supported(int32_t C, int32_t heads) { return ok_shape(C, heads) ? 1 : 0; }

extern "C" int cavp_attn1_prepare(int32_t dtype, const float* wq, const float* wp, const void* k, const void* v, float* U, float* P,
                                  int32_t B, int32_t C, int32_t heads, float scale, void* stream) {
  if (!wq || !wp || !k || !v || !U || !P || B <= 0) return CAVP_ERR_BAD_ARG;
  if ((dtype != CAVP_F32 && dtype != CAVP_BF16) || !ok_shape(C, heads)) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CAVP_F32)
    attn1_prepare_kernel<float><<<B * heads, 256, 0, s>>>(wq, wp, (const float*)k, (const float*)v, U, P, C, heads, scale);
  else
    attn1_prepare_kernel<bf16_t><<<B * heads, 256, 0, s>>>(wq, wp, (const bf16_t*)k, (const bf16_t*)v, U, P, C, heads, scale);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_attn1_fwd(int32_t dtype, const void* x, const float* U, const float* P, const float* bp, void* out, float* attn,
                              int32_t B, int32_t xb, int32_t T, int32_t C, int32_t heads, void* stream) {
  if (!x || !U || !P || !out || !attn || B <= 0 || xb <= 0 || B % xb || T <= 0) return CAVP_ERR_BAD_ARG;
  if ((dtype != CAVP_F32 && dtype != CAVP_BF16) || !ok_shape(C, heads)) return CAVP_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(out) || !al16(U) || !al16(P) || (bp && !al16(bp))) return CAVP_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(token_groups(T, B, 4096), B);   // four waves per SIMD
  if (dtype == CAVP_F32)
    attn1_fwd_kernel<float, 4><<<grid, 256, 0, s>>>((const float*)x, U, P, bp, (float*)out, attn, xb, T, C);
  else
    attn1_fwd_kernel<bf16_t, 4><<<grid, 256, 0, s>>>((const bf16_t*)x, U, P, bp, (bf16_t*)out, attn, xb, T, C);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" size_t cavp_attn1_bwd_workspace_bytes(int32_t B, int32_t xb, int32_t T, int32_t C, int32_t heads) {
  if (B <= 0 || xb <= 0 || B % xb || T <= 0 || !ok_shape(C, heads)) return 0;
  const size_t g = (size_t)token_groups(T, xb);
  return (2 * (size_t)B * g * heads * C + (size_t)xb * g * C) * sizeof(float);
}

extern "C" int cavp_attn1_bwd(int32_t dtype, const void* dout, const void* x, const float* U, const float* P, void* dx, float* dU,
                              float* dP, float* dbp, void* workspace, size_t workspace_bytes, int32_t B, int32_t xb, int32_t T, int32_t C,
                              int32_t heads, void* stream) {
  if (!dout || !x || !U || !P || !dx || !dU || !dP || !workspace || B <= 0 || xb <= 0 || B % xb || T <= 0) return CAVP_ERR_BAD_ARG;
  if ((dtype != CAVP_F32 && dtype != CAVP_BF16) || !ok_shape(C, heads)) return CAVP_ERR_UNSUPPORTED;
  if (!al16(dout) || !al16(x) || !al16(dx) || !al16(U) || !al16(P) || !al16(workspace)) return CAVP_ERR_ALIGN;
  if (workspace_bytes < cavp_attn1_bwd_workspace_bytes(B, xb, T, C, heads)) return CAVP_ERR_WORKSPACE;
  const int g = token_groups(T, xb);
  float* slabU = (float*)workspace;
  float* slabP = slabU + (size_t)B * g * heads * C;
  float* slabB = slabP + (size_t)B * g * heads * C;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(g, xb);
  const int lds = 4 * 2 * heads * C * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn1_bwd_kernel<float, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 4 * 512 * 4);
    (void)hipFuncSetAttribute((const void*)attn1_bwd_kernel<bf16_t, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 4 * 512 * 4);
    attr = true;
  }
  if (dtype == CAVP_F32)
    attn1_bwd_kernel<float, 4><<<grid, 256, lds, s>>>((const float*)dout, (const float*)x, U, P, (float*)dx, slabU, slabP, slabB, B, xb, T, C);
  else
    attn1_bwd_kernel<bf16_t, 4><<<grid, 256, lds, s>>>((const bf16_t*)dout, (const bf16_t*)x, U, P, (bf16_t*)dx, slabU, slabP, slabB, B, xb, T, C);
  if (hipGetLastError() != hipSuccess) return CAVP_ERR_LAUNCH;
  int nb = (2 * B * heads * C + 255) / 256;
  if (nb > 1024) nb = 1024;
  const int nb_b = (C + 3) / 4;
  attn1_slabsum_kernel<<<nb + nb_b, 256, 0, s>>>(slabU, slabP, slabB, dU, dP, dbp, B, xb, g, heads * C, C, nb);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_attn1_finish(int32_t dtype, const float* wq, const float* wp, const void* k, const void* v, const float* dU,
                                 const float* dP, float* dwq, float* dwp, float* dk, float* dv, int32_t B, int32_t C, int32_t heads,
                                 float scale, void* stream) {
  if (!wq || !wp || !k || !v || !dU || !dP || !dwq || !dwp || !dk || !dv || B <= 0) return CAVP_ERR_BAD_ARG;
  if ((dtype != CAVP_F32 && dtype != CAVP_BF16) || !ok_shape(C, heads) || B * heads > 8192) return CAVP_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  size_t lds = (size_t)2 * C + 3 * (size_t)(C / heads);
  if ((size_t)B * heads > lds) lds = (size_t)B * heads;
  lds *= sizeof(float);
  const int grid = 2 * C + B * heads;
  if (dtype == CAVP_F32)
    attn1_finish_kernel<float><<<grid, 256, lds, s>>>(wq, wp, (const float*)k, (const float*)v, dU, dP, dwq, dwp, dk, dv, B, C, heads, scale);
  else
    attn1_finish_kernel<bf16_t><<<grid, 256, lds, s>>>(wq, wp, (const bf16_t*)k, (const bf16_t*)v, dU, dP, dwq, dwp, dk, dv, B, C, heads, scale);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}
