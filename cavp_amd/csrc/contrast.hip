// Device side of the pixel-level audio-visual InfoNCE (reference loss/contrastive_aud.py::ContrastLoss.info_nce and
// the normalise / gather part of ::forward / ::extraction_samples).  The class-balanced SAMPLING stays on the host
// (it consumes torch.randperm from the CPU generator; reproducing the reference's indices needs the same RNG stream);
// the host hands over (image, pixel) index lists + labels of the N <= ~3000 anchors.
//   1. gather_l2norm:   A[i] = x[b_i, :, p_i] / max(||.||_2, eps)         (F.normalize(dim=1) then boolean gather)
//   2. S = A A^T / T:   the f32 MFMA igemm (cavp_conv2d_nhwc, weights = A)
//   3. infonce_rows:    one workgroup per anchor row: max, negative sum, per-positive log-prob, mean; optional dS
//   4. symm_add:        G = dS + dS^T  (anchors and contrasts are the same tensor: both roles get gradient)
//   5. dA = G A (cavp_conv2d_wgrad_nhwc), then l2norm_bwd_scatter back into the NHWC feature gradient
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one wave per anchor
__global__ __launch_bounds__(256) void gather_l2norm_kernel(const float* __restrict__ x, long long sb, long long sc,
                                                            long long sp, const int* __restrict__ ib,
                                                            const int* __restrict__ ip, int N, int C, float eps,
                                                            float* __restrict__ A, float* __restrict__ norms) {
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += gridDim.x * 4) {
    const float* src = x + (long long)ib[i] * sb + (long long)ip[i] * sp;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = src[(long long)c * sc]; q += v * v; }
    const float nrm = fmaxf(sqrtf(wave_sum(q)), eps);
    for (int c = lane; c < C; c += 64) A[(size_t)i * C + c] = src[(long long)c * sc] / nrm;
    if (lane == 0) norms[i] = nrm;
  }
}

// S: [ld][ld] (row i = anchor i, already divided by the temperature); labels: int [N]
// out_rows[i] = mean_log_prob_pos_i; dS (optional) = d(-mean_i mlpp_i)/dS * grad_scale, zero outside [N][N]
__global__ __launch_bounds__(256) void infonce_rows_kernel(const float* __restrict__ S, const int* __restrict__ lab, int N,
                                                           int ld, float eps, float* __restrict__ out_rows,
                                                           float* __restrict__ dS, float grad_scale) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  if (i >= ld) return;
  if (i >= N) {  // padding row
    if (dS) for (int j = threadIdx.x; j < ld; j += 256) dS[(size_t)i * ld + j] = 0.f;
    return;
  }
  const float* row = S + (size_t)i * ld;
  const int li = lab[i];
  float m = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) m = fmaxf(m, row[j]);
  m = block_max(m, red);
  float neg = 0.f, cnt = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const bool same = lab[j] == li;
    neg += same ? 0.f : expf(row[j] - m);
    cnt += (same && j != i) ? 1.f : 0.f;
  }
  neg = block_sum(neg, red);
  cnt = block_sum(cnt, red);
  float slp = 0.f, rsum = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    if (lab[j] == li && j != i) {
      const float l = row[j] - m, e = expf(l);
      slp += l - logf(e + neg);
      rsum += 1.f / (e + neg);
    }
  }
  slp = block_sum(slp, red);
  rsum = block_sum(rsum, red);
  if (threadIdx.x == 0) out_rows[i] = slp / (cnt + eps);
  if (dS) {
    // d(-1/N sum_i mlpp_i)/dl_ik = -c_i [ m_ik (1 - e_ik / D_ik) - e_ik n_ik R_i ],  c_i = 1 / (N (P_i + eps))
    const float ci = grad_scale / ((float)N * (cnt + eps));
    for (int j = threadIdx.x; j < ld; j += 256) {
      float g = 0.f;
      if (j < N) {
        const float e = expf(row[j] - m);
        const bool same = lab[j] == li;
        if (same && j != i) g = -ci * (1.f - e / (e + neg));
        if (!same) g = ci * e * rsum;
      }
      dS[(size_t)i * ld + j] = g;
    }
  }
}

__global__ __launch_bounds__(256) void mean_neg_kernel(const float* rows, int N, float* loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) s += rows[j];
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = -s / (float)N;
}

__global__ __launch_bounds__(256) void symm_add_kernel(const float* __restrict__ d, float* __restrict__ g, int n,
                                                       float scale, const float* __restrict__ scale_dev) {
  if (scale_dev) scale *= *scale_dev;   // the upstream gradient of the loss, read on the device (no host round trip)
  const long long total = (long long)n * n;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t / n), j = (int)(t - (long long)i * n);
    g[t] = (d[t] + d[(size_t)j * n + i]) * scale;
  }
}

// dx[b_i, :, p_i] = (dA_i - A_i <A_i, dA_i>) / ||x_i||   (anchors are distinct pixels: plain scatter)
__global__ __launch_bounds__(256) void l2norm_bwd_scatter_kernel(const float* __restrict__ dA, const float* __restrict__ A,
                                                                 const float* __restrict__ norms,
                                                                 const int* __restrict__ ib, const int* __restrict__ ip,
                                                                 int N, int C, float* __restrict__ dx, long long sb,
                                                                 long long sc, long long sp) {
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < N; i += gridDim.x * 4) {
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += A[(size_t)i * C + c] * dA[(size_t)i * C + c];
    dot = wave_sum(dot);
    const float inv = 1.f / norms[i];
    float* dst = dx + (long long)ib[i] * sb + (long long)ip[i] * sp;
    for (int c = lane; c < C; c += 64) dst[(long long)c * sc] = (dA[(size_t)i * C + c] - A[(size_t)i * C + c] * dot) * inv;
  }
}

}  // namespace
#define CHECK_LAUNCH() return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH

// F.interpolate(mode='nearest') of the label maps to the feature resolution (contrastive_aud.py:18-22): source index
// min(floor(dst * float32(in / out)), in - 1) per axis, as ATen computes it.  int64 [B][H][W] -> int32 [B][h][w].
static __global__ __launch_bounds__(256) void label_nearest_kernel(const long long* __restrict__ gt, int* __restrict__ out, int B,
                                                                   int H, int W, int h, int w, float sh, float sw) {
  const long long total = (long long)B * h * w;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long long)w * h));
    int ys = (int)floorf((float)y * sh), xs = (int)floorf((float)x * sw);
    ys = ys < H - 1 ? ys : H - 1;
    xs = xs < W - 1 ? xs : W - 1;
    out[i] = (int)gt[((size_t)b * H + ys) * W + xs];
  }
}

extern "C" int cavp_label_nearest(const int64_t* gt, int32_t* out, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w,
                                  void* stream) {
  if (!gt || !out || B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return CAVP_ERR_BAD_ARG;
  const long long total = (long long)B * h * w;
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  label_nearest_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>((const long long*)gt, out, B, H, W, h, w, (float)H / (float)h,
                                                                 (float)W / (float)w);
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}

extern "C" int cavp_gather_l2norm(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_p,
                                  const int32_t* idx_b, const int32_t* idx_p, int32_t N, int32_t C, float eps, float* A,
                                  float* norms, void* stream) {
  if (!x || !idx_b || !idx_p || !A || !norms || N <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  gather_l2norm_kernel<<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, stride_b, stride_c, stride_p, idx_b, idx_p, N, C, eps, A, norms);
  CHECK_LAUNCH();
}

extern "C" int cavp_infonce_rows(const float* S, const int32_t* labels, int32_t N, int32_t ld, float eps, float* row_mlpp,
                                 float* loss, float* dS, float grad_scale, void* stream) {
  if (!S || !labels || !row_mlpp || !loss || N <= 0 || ld < N) return CAVP_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  infonce_rows_kernel<<<ld, 256, 0, s>>>(S, labels, N, ld, eps, row_mlpp, dS, grad_scale);
  mean_neg_kernel<<<1, 256, 0, s>>>(row_mlpp, N, loss);
  CHECK_LAUNCH();
}

extern "C" int cavp_symm_add(const float* d, float* g, int32_t n, float scale, void* stream) {
  if (!d || !g || n <= 0) return CAVP_ERR_BAD_ARG;
  long long nb = ((long long)n * n + 255) / 256;
  if (nb > 8192) nb = 8192;
  symm_add_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(d, g, n, scale, nullptr);
  CHECK_LAUNCH();
}

extern "C" int cavp_symm_add_scaled(const float* d, float* g, int32_t n, float scale, const float* scale_dev, void* stream) {
  if (!d || !g || !scale_dev || n <= 0) return CAVP_ERR_BAD_ARG;
  long long nb = ((long long)n * n + 255) / 256;
  if (nb > 8192) nb = 8192;
  symm_add_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(d, g, n, scale, scale_dev);
  CHECK_LAUNCH();
}

extern "C" int cavp_l2norm_bwd_scatter(const float* dA, const float* A, const float* norms, const int32_t* idx_b,
                                       const int32_t* idx_p, int32_t N, int32_t C, float* dx, int64_t stride_b,
                                       int64_t stride_c, int64_t stride_p, void* stream) {
  if (!dA || !A || !norms || !idx_b || !idx_p || !dx || N <= 0 || C <= 0) return CAVP_ERR_BAD_ARG;
  l2norm_bwd_scatter_kernel<<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>(dA, A, norms, idx_b, idx_p, N, C, dx, stride_b, stride_c, stride_p);
  CHECK_LAUNCH();
}
