// Log-mel front-end = the trainers' preprocess_audio (trainer_cavp_vpo_mono.py:43-52,59-69; utils/sourcesep.py:23-47):
//   torchaudio MelSpectrogram(16 kHz, n_fft 512, win 400 (periodic Hann, centred in the FFT frame), hop 160, 64 mel
//   bins 125-3800 Hz, power 2, center=True / reflect padding) -> first n_frames frames -> [frame][mel] ->
//   20 log10(max(1e-5, x)) -> 2 (x - spec_min) / (spec_max - spec_min) - 1.
// One workgroup per (clip, frame): the 512 windowed samples are written to LDS in bit-reversed order, 9 radix-2
// butterfly stages (256 threads = one butterfly each, twiddles from a 256-entry LDS table) give the spectrum, the power
// of bins 0..256 goes back to LDS and 64 threads apply the (host-built, f32) mel filterbank, take the logarithm and
// normalise.  (First version: every thread evaluated one DFT bin directly, 512 bank-conflicting table gathers per
// bin: 322 us for 64 clips; this one: see profiles/r01_notes.md.)
// Output is the [N][1][frames][mels] f32 tensor the audio encoder eats.
#include "common.h"

namespace {

constexpr int kNFFT = 512, kNFREQ = 257;

__global__ __launch_bounds__(256) void mel_frontend_kernel(const float* __restrict__ wave, const float* __restrict__ window,
                                                           const float* __restrict__ fb, float* __restrict__ out, int A,
                                                           int hop, int n_frames, int n_mels, float amin, float spec_min,
                                                           float inv_range) {
  __shared__ float re[kNFFT], im[kNFFT], cs[kNFFT / 2], sn[kNFFT / 2], pw[kNFREQ + 7];
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* w = wave + (size_t)n * A;
  for (int i = tid; i < kNFFT; i += 256) {
    int j = t * hop - kNFFT / 2 + i;          // center=True
    if (j < 0) j = -j;                        // reflect (edge sample not repeated)
    if (j >= A) j = 2 * (A - 1) - j;
    j = j < 0 ? 0 : (j >= A ? A - 1 : j);     // (only for clips shorter than the padding)
    const int r = (int)(__brev((unsigned)i) >> 23);   // 9-bit reversal
    re[r] = w[j] * window[i];
    im[r] = 0.f;
  }
  {
    float s, c;
    sincospif((float)tid * (2.0f / (float)kNFFT), &s, &c);   // W^tid = exp(-2 pi i tid / 512) = (c, -s)
    cs[tid] = c;
    sn[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int st = 1; st <= 9; ++st) {
    const int half = 1 << (st - 1);
    const int j = tid & (half - 1), i0 = ((tid >> (st - 1)) << st) + j, i1 = i0 + half;
    const int tw = j << (9 - st);
    const float wr = cs[tw], wi = -sn[tw];
    const float xr = re[i1], xi = im[i1];
    const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
    const float ar = re[i0], ai = im[i0];
    re[i1] = ar - tr; im[i1] = ai - ti;
    re[i0] = ar + tr; im[i0] = ai + ti;
    __syncthreads();
  }
  for (int k = tid; k < kNFREQ; k += 256) pw[k] = re[k] * re[k] + im[k] * im[k];
  __syncthreads();
  if (tid < n_mels) {
    float m = 0.f;
    for (int k = 0; k < kNFREQ; ++k) m += fb[k * n_mels + tid] * pw[k];
    const float db = 20.f * (logf(fmaxf(amin, m)) / logf(10.f));   // sourcesep.log10 = log(x) / log(10)
    out[((size_t)n * n_frames + t) * n_mels + tid] = 2.f * (db - spec_min) * inv_range - 1.f;
  }
}

}  // namespace

extern "C" int cavp_mel_frontend(const float* wave, int32_t N, int32_t A, const float* window, const float* fb,
                                 float* out, int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mels, float amin,
                                 float spec_min, float spec_max, void* stream) {
  if (!wave || !window || !fb || !out || N <= 0 || A <= 1 || hop <= 0 || n_frames <= 0 || n_mels <= 0 || spec_max <= spec_min)
    return CAVP_ERR_BAD_ARG;
  if (n_fft != kNFFT || n_mels > 256 || N > 65535) return CAVP_ERR_UNSUPPORTED;
  if (n_frames > 1 + A / hop) return CAVP_ERR_BAD_ARG;   // torch.stft(center=True) yields 1 + A / hop frames
  mel_frontend_kernel<<<dim3(n_frames, N), 256, 0, (hipStream_t)stream>>>(wave, window, fb, out, A, hop, n_frames, n_mels, amin,
                                                                         spec_min, 1.f / (spec_max - spec_min));
  return hipGetLastError() == hipSuccess ? CAVP_OK : CAVP_ERR_LAUNCH;
}
