// Shared device helpers for the CAVP gfx950 kernels.  Written for MI355X only (wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cavp_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

typedef unsigned short bf16_t;  // raw bfloat16 bits

#define CAVP_WAVE 64

// A/B and profiling knobs exist only in -DCAVP_PROFILE builds (python -m cavp_amd.build --profile; tools/bench_conv.py,
// tools/ab_bench.sh).  The product library reads no environment variable and carries no profiling branch in a kernel:
// cavp_knob_* fold to their defaults and CAVP_DBG(p, bit) to `false`.
#ifdef CAVP_PROFILE
#include <stdlib.h>
inline int cavp_knob_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline double cavp_knob_double(const char* name, double dflt) { const char* e = getenv(name); return e ? atof(e) : dflt; }
inline const char* cavp_knob_str(const char* name) { return getenv(name); }
#define CAVP_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
inline int cavp_knob_int(const char*, int dflt) { return dflt; }
inline double cavp_knob_double(const char*, double dflt) { return dflt; }
inline const char* cavp_knob_str(const char*) { return nullptr; }
#define CAVP_DBG(p, bit) false
#endif

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw_t;
// f32 -> bf16, round-to-nearest-even (same as torch .to(bfloat16)): gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32, two values per instruction); the first version spent ~6 integer VALU ops per element on it, which
// made the 16-byte-per-thread epilogues VALU-bound.
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_hw_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;  // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector load / store of VE elements as f32
template <typename T> struct VecT;
template <> struct VecT<float> {
  static constexpr int VE = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const float4 t = *(const float4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
  // the 16 raw bytes now, the VE floats later (loads of the NEXT row parked in 4 registers while the current row is reduced)
  __device__ static __forceinline__ uint4 load_raw(const float* p) { return *(const uint4*)p; }
  __device__ static __forceinline__ void unpack(const uint4& t, float* v) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
  }
};
template <> struct VecT<bf16_t> {
  static constexpr int VE = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) { unpack(*(const uint4*)p, v); }
  __device__ static __forceinline__ uint4 load_raw(const bf16_t* p) { return *(const uint4*)p; }
  __device__ static __forceinline__ void unpack(const uint4& t, float* v) {
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

// activation codes = cavp_act_t in include/cavp_hip.h
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case CAVP_ACT_RELU: return v > 0.f ? v : 0.f;
    case CAVP_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;                      // nn.LeakyReLU() default slope
    case CAVP_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // exact-erf GELU
    default: return v;
  }
}

__device__ __forceinline__ float gelu_grad(float x) {  // d/dx [0.5 x (1 + erf(x / sqrt 2))]
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// gelu(x) and gelu'(x) together, for the fused fc1 + GELU epilogue (cavp_conv_desc.aux_mode 1), where this math is NOT hidden
// behind memory traffic: two erff + one expf per element (~70 VALU) made the fused epilogue cost what the two elementwise
// passes it replaces had cost.  One exp(-x^2/2) serves the density AND the Abramowitz-Stegun 7.1.26 form of erf
// (|error| <= 1.5e-7, below bf16 and f32-accumulation noise): ~18 VALU per element.
__device__ __forceinline__ void gelu_and_grad(float x, float& g, float& dg) {
  const float ax = fabsf(x);
  const float e = __expf(-0.5f * x * x);                       // exp(-z^2), z = |x| / sqrt 2
  const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float h = 0.5f * poly * t * e;                         // 0.5 * erfc(z)
  const float cdf = x >= 0.f ? 1.f - h : h;
  g = x * cdf;
  dg = fmaf(x * e, 0.3989422804014327f, cdf);
}

// gelu(x) alone in gelu_and_grad's form (one exp, one reciprocal, |error of the CDF| <= 1.5e-7: three orders of magnitude below the bf16
// rounding of the stored result).  The bf16 inference epilogues use it: with erff the 304 -> 1216 token linear of the eval forward ran
// 252 us against 138 us for its 1216 -> 304 partner (profiles/r06_layers_eval_bf16.txt rows 90 / 91); the f32 parity path keeps erff.
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-0.5f * x * x);
  const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float h = 0.5f * poly * t * e;                         // 0.5 * erfc(|x| / sqrt 2)
  return x * (x >= 0.f ? 1.f - h : h);
}

// activation of N values with ONE (wave-uniform) dispatch on the activation code (FAST_GELU: gelu_fast instead of erff - bf16 outputs)
template <int N, bool FAST_GELU = false>
__device__ __forceinline__ void apply_act_vec(float* v, int act) {
  switch (act) {
    case CAVP_ACT_RELU:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.f);
      break;
    case CAVP_ACT_LEAKY:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
      break;
    case CAVP_ACT_GELU:
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = FAST_GELU ? gelu_fast(v[e]) : 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
      break;
    default: break;
  }
}


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

// Zero-fill of an f32 buffer as a KERNEL.  hipMemsetAsync nodes captured into a hipGraph were observed (ROCm 7.2, MI355X) to
// be mis-ordered against their neighbours from the second replay on: the four weight gradients that were cleared by a
// captured memset came out as inf / 1e25 on every replay but the first (found by the RCCL single-rank test, round 2).  A
// kernel node has ordinary stream-order dependencies.
static __global__ __launch_bounds__(256) void cavp_zero_f32_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0.f;
}
static inline hipError_t cavp_zero_f32_async(void* ptr, size_t bytes, hipStream_t s) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  cavp_zero_f32_kernel<<<dim3((unsigned)nb), dim3(256), 0, s>>>((float*)ptr, n);
  return hipGetLastError();
}

// Deterministic mode (cavp_set_deterministic, include/cavp_hip.h): the four reductions that normally end in one f32 atomic per
// (workgroup, channel) - BatchNorm / bias column sums, LayerNorm and attention-gate parameter gradients - store their
// per-workgroup partials into a caller-provided scratch buffer instead and a second kernel adds them in workgroup order.
// Process-wide state (one process per GPU); defined in pointwise.hip.
struct CavpDetState {
  float* scratch;
  size_t floats;
};
extern CavpDetState g_cavp_det;
// out0[i] += sum_p part[p][i], out1[i] += sum_p part[nparts + p][i] in a FIXED order: one wave per output element, lane l adds
// the partials l, l + 64, ... (ascending), then a fixed butterfly over the 64 lanes.  (Round 2 used one THREAD per element
// walking all nparts (up to 768) partials one dependent-latency load at a time: ~150 us per call, 9 ms per C1' step - the
// deterministic mode cost +59 %, measured in round 3.)
static __global__ __launch_bounds__(256) void cavp_det_finish_kernel(const float* __restrict__ part, int nparts, int n,
                                                                    float* __restrict__ out0, float* __restrict__ out1) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int total = out1 ? 2 * n : n;
  if (w >= total) return;
  const int st = w >= n ? 1 : 0, i = w - st * n;
  const float* src = part + (size_t)st * nparts * n + i;
  float s = 0.f;
  for (int q = lane; q < nparts; q += 64) s += src[(size_t)q * n];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) (st ? out1 : out0)[i] += s;
}
// scratch for 2 x nparts x n floats, or nullptr when the mode is off; *err is set when it is on but the buffer is too small
static inline float* cavp_det_scratch(int nparts, int n, bool* err) {
  *err = false;
  if (!g_cavp_det.scratch) return nullptr;
  if ((size_t)2 * nparts * n > g_cavp_det.floats) { *err = true; return nullptr; }
  return g_cavp_det.scratch;
}
static inline hipError_t cavp_det_finish(const float* part, int nparts, int n, float* out0, float* out1, hipStream_t s) {
  cavp_det_finish_kernel<<<dim3(((out1 ? 2 * n : n) + 3) / 4), dim3(256), 0, s>>>(part, nparts, n, out0, out1);
  return hipGetLastError();
}

// Touch every 64-byte line of the kernel-argument segment at kernel entry.  hipcc loads kernel arguments lazily, field by field
// where they are first used, and a fresh launch's argument block is in nobody's cache: the s_memtime timeline of an igemm
// workgroup (profiles/r05_igemm_tile_timeline.txt) shows 3000 .. 4000 cycles between kernel entry and the first tile - three or
// four DEPENDENT scalar-cache misses in a row.  Issued together up front they overlap into one.
template <int BYTES>
__device__ __forceinline__ void cavp_prefetch_kernargs() {
  const __attribute__((address_space(4))) unsigned* ka = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned any = 0;
#pragma unroll
  for (int o = 0; o < BYTES; o += 64) any |= ka[o / 4];
  asm volatile("" ::"s"(any));
}

// the same for BYTES at a run-time (wave-uniform) byte offset into the segment: one job of a grouped launch's table
template <int BYTES>
__device__ __forceinline__ void cavp_prefetch_kernargs_at(int byte_off) {
  const __attribute__((address_space(4))) unsigned* ka =
      (const __attribute__((address_space(4))) unsigned*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + (byte_off & ~63));
  unsigned any = 0;
#pragma unroll
  for (int o = 0; o < BYTES + 64; o += 64) any |= ka[o / 4];
  asm volatile("" ::"s"(any));
}

// s_waitcnt immediate that only waits for vmcnt <= n (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// floor(n / d) for 0 <= n < 2^31 by a host-prepared multiplier: m = floor(2^(31+L) / d) + 1, L = ceil(log2 d), shift = 31 + L
// (exact: the error term n * e / 2^(31+L), e <= 1, stays below 1/d because n < 2^31 <= 2^(31+L) / d).  The tile set-up did
// two real integer divisions per operand row (~40 VALU each, 8 per thread and tile): ~1 us of every tile at 2 waves per SIMD.
__device__ __forceinline__ int fast_div(int n, unsigned m, unsigned shift) {
  return (int)(((unsigned long long)(unsigned)n * m) >> shift);
}
inline void fast_div_prepare(int d, unsigned* m, unsigned* shift) {
  int L = 0;
  while ((1ll << L) < d) ++L;
  *m = (unsigned)(((1ull << (31 + L)) / (unsigned)d) + 1ull);
  *shift = 31u + (unsigned)L;
}

// (n, h, w) of a flat NHW pixel index / (pixel, channel-vector) of a flat element index.  The 64-bit `/` and `%` these
// replace cost ~100 VALU instructions each (five per thread in the pooling / resize / im2col kernels); indices below 2^31
// take the 32-bit route (three ~25-instruction unsigned divisions).
__device__ __forceinline__ void split_pixel(long long pix, int W, int H, int& w, int& h, int& n) {
  if (pix <= 0x7fffffffll) {
    const unsigned p = (unsigned)pix, r = p / (unsigned)W, q = r / (unsigned)H;
    w = (int)(p - r * (unsigned)W);
    h = (int)(r - q * (unsigned)H);
    n = (int)q;
  } else {
    const long long r = pix / W;
    w = (int)(pix - r * W);
    n = (int)(r / H);
    h = (int)(r - (long long)n * H);
  }
}
__device__ __forceinline__ void split_index(long long idx, int CV, int W, int H, int& cv, long long& pix, int& w, int& h,
                                            int& n) {
  if (idx <= 0x7fffffffll) {
    const unsigned u = (unsigned)idx, p = u / (unsigned)CV;
    cv = (int)(u - p * (unsigned)CV);
    pix = (long long)p;
  } else {
    pix = idx / CV;
    cv = (int)(idx - pix * CV);
  }
  split_pixel(pix, W, H, w, h, n);
}

// Bijective XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): gives each XCD a contiguous
// range of logical tile ids so neighbouring tiles (which share an operand panel) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Spatial-reduction attention (pvt_ops.hip / pvt_train.hip): 64-query blocks per workgroup.  A workgroup stages the whole K / V of
// its (image, head) - 64 KiB, two workgroups per CU - before its first query block; the smallest count (<= 8) that fits the launch
// into ONE round of 512 workgroups amortises that staging (PVTv2-B5 at 512 x 512, 16 frames: stage 1 4096 -> 512 workgroups of 8
// blocks, stage 2 2048 -> 512 of 4, stage 3 1280 -> 480 of 3, stage 4 unchanged).
inline int cavp_sra_blocks_per_wg(int Nq, int bh) {
  const int blocks = (Nq + 63) / 64;
  int qpw = 1;
  while (qpw < 8 && (long long)((blocks + qpw - 1) / qpw) * bh > 512) ++qpw;
  return qpw;
}
