// STFT log-mel front-end (K1) and the Vocos iSTFT head (K16 tail) -- 1024-point FFTs in LDS, one frame per workgroup.
//
// f5e_stft_logmel: reference model/modules.py:75-101 (torchaudio MelSpectrogram(power=1, center=True, norm=None,
//   htk) -> clamp(1e-5).log()).  Window, twiddles and the [513][n_mels] filterbank are host-built constant tables.
//   Output is written token-major [B][T][n_mels] (what CFM.sample permutes to at cfm.py:372-373).
// f5e_istft_head: Vocos ISTFTHead tail (restated in-tree at runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:
//   45-59): mag = clip(exp(.), 100), S = mag (cos p + i sin p), irfft(1024) * window, overlap-add / window envelope,
//   trimmed by n_fft/2 at both ends (torch.istft(center=True)) -> [B][hop * (T - 1)].
//
// Both are HBM-bound (a frame is 4 KB in, 0.4 KB out / 4.1 KB in, 1 KB out); the radix-2 butterflies run from LDS.
#include "f5e_common.h"

namespace {

constexpr int NFFT = 1024, LOG2N = 10, NBIN = NFFT / 2 + 1;

__device__ __forceinline__ int bitrev10(int i) { return (int)(__brev((unsigned)i) >> 22); }

// in-place radix-2 DIT on bit-reversed input; tw[k] = (cos, -sin)(2 pi k / 1024), k < 512; inverse conjugates it
template <bool INV>
__device__ __forceinline__ void fft1024(float2* x, const float2* tw, int tid) {
#pragma unroll 1
  for (int s = 1; s <= LOG2N; ++s) {
    const int half = 1 << (s - 1);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int b = tid + r * 256;
      const int pos = b & (half - 1);
      const int i0 = ((b >> (s - 1)) << s) + pos;
      const int i1 = i0 + half;
      float2 w = tw[pos << (LOG2N - s)];
      if (INV) w.y = -w.y;
      const float2 u = x[i0], v = x[i1];
      const float tr = w.x * v.x - w.y * v.y;
      const float ti = w.x * v.y + w.y * v.x;
      x[i0] = make_float2(u.x + tr, u.y + ti);
      x[i1] = make_float2(u.x - tr, u.y - ti);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void stft_logmel_kernel(const float* wav, int nw, int ldw, const float* window,
                                                           const float2* tw, const float* fb, float* out, int T,
                                                           int hop, int n_mels) {
  __shared__ float2 x[NFFT];
  __shared__ float mag[NBIN + 3];
  const int tid = threadIdx.x;
  const int f = blockIdx.x, b = blockIdx.y;
  const float* w = wav + (size_t)b * ldw;
  for (int j = tid; j < NFFT; j += 256) {
    int i = f * hop + j - NFFT / 2;
    if (i < 0) i = -i;                    // reflect padding (no edge repeat)
    if (i >= nw) i = 2 * (nw - 1) - i;
    x[bitrev10(j)] = make_float2(w[i] * window[j], 0.f);
  }
  fft1024<false>(x, tw, tid);
  for (int k = tid; k < NBIN; k += 256) mag[k] = sqrtf(x[k].x * x[k].x + x[k].y * x[k].y);
  __syncthreads();
  if (tid < n_mels) {
    float acc = 0.f;
    for (int k = 0; k < NBIN; ++k) acc += mag[k] * fb[(size_t)k * n_mels + tid];
    out[((size_t)b * T + f) * n_mels + tid] = logf(fmaxf(acc, 1e-5f));
  }
}

// Same result, bit for bit, with the filterbank's structure used: a triangular mel filter is non-zero on one short run of
// FFT bins (5..60 of 513 for the 100 HTK filters), so filter m only walks its own run [lo, lo + cnt) -- in ascending bin
// order, exactly the partial sums the dense loop produces, whose other terms are +0 -- and the ~1000 non-zero weights
// (compact, <= 2048 floats) are staged in LDS once per workgroup instead of 513 dependent L2 round trips per filter
// (the dense kernel spends 70 of its 73 us at C2 in that loop).
constexpr int FB_MAX_NNZ = 2048;
__global__ __launch_bounds__(256) void stft_logmel_banded_kernel(const float* wav, int nw, int ldw, const float* window,
                                                                  const float2* tw, const float* fbc, const int* band,
                                                                  int nnz, float* out, int T, int hop, int n_mels) {
  __shared__ float2 x[NFFT];
  __shared__ float mag[NBIN + 3];
  __shared__ float wts[FB_MAX_NNZ];
  const int tid = threadIdx.x;
  const int f = blockIdx.x, b = blockIdx.y;
  const float* w = wav + (size_t)b * ldw;
  for (int j = tid; j < NFFT; j += 256) {
    int i = f * hop + j - NFFT / 2;
    if (i < 0) i = -i;                    // reflect padding (no edge repeat)
    if (i >= nw) i = 2 * (nw - 1) - i;
    x[bitrev10(j)] = make_float2(w[i] * window[j], 0.f);
  }
  for (int j = tid; j < nnz; j += 256) wts[j] = fbc[j];
  int lo = 0, cnt = 0, off = 0;
  if (tid < n_mels) { lo = band[tid * 3]; cnt = band[tid * 3 + 1]; off = band[tid * 3 + 2]; }
  fft1024<false>(x, tw, tid);
  for (int k = tid; k < NBIN; k += 256) mag[k] = sqrtf(x[k].x * x[k].x + x[k].y * x[k].y);
  __syncthreads();
  if (tid < n_mels) {
    float acc = 0.f;
    for (int k = 0; k < cnt; ++k) acc += mag[lo + k] * wts[off + k];
    out[((size_t)b * T + f) * n_mels + tid] = logf(fmaxf(acc, 1e-5f));
  }
}

// z: [B*T][2*NBIN] fp32 (log-magnitude | phase) -> frames [B*T][1024] = irfft(S) * window
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* z, int ldz, const float* window,
                                                            const float2* tw, float* frames) {
  __shared__ float2 x[NFFT];
  const int tid = threadIdx.x;
  const size_t fr = blockIdx.x;
  const float* zp = z + fr * ldz;
  for (int k = tid; k < NBIN; k += 256) {
    const float m = fminf(expf(zp[k]), 100.0f);
    float s, c;
    sincosf(zp[NBIN + k], &s, &c);
    float re = m * c, im = m * s;
    if (k == 0 || k == NFFT / 2) im = 0.f;  // irfft ignores the imaginary part of DC / Nyquist
    x[bitrev10(k)] = make_float2(re, im);
    if (k > 0 && k < NFFT / 2) x[bitrev10(NFFT - k)] = make_float2(re, -im);
  }
  fft1024<true>(x, tw, tid);
  for (int j = tid; j < NFFT; j += 256) frames[fr * NFFT + j] = x[j].x * (1.0f / NFFT) * window[j];
}

// out[b][i] = sum_f frames[b][f][i + 512 - hop f] / sum_f window^2[i + 512 - hop f]
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* frames, const float* window, float* out, int T,
                                                         int hop, int out_len) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= out_len) return;
  const int pos = i + NFFT / 2;
  int f_lo = (pos - (NFFT - 1) + hop - 1) / hop;
  if (f_lo < 0) f_lo = 0;
  int f_hi = pos / hop;
  if (f_hi > T - 1) f_hi = T - 1;
  float acc = 0.f, env = 0.f;
  for (int f = f_lo; f <= f_hi; ++f) {
    const int j = pos - f * hop;
    const float wj = window[j];
    acc += frames[((size_t)b * T + f) * NFFT + j];
    env += wj * wj;
  }
  out[(size_t)b * out_len + i] = acc / env;
}

}  // namespace

extern "C" {

int f5e_stft_logmel(hipStream_t st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                    const float* fb, float* out, int B, int n_fft, int hop, int n_mels) {
  F5E_REQUIRE(wav && window && twiddle && fb && out, "stft_logmel: null operand");
  F5E_REQUIRE(n_fft == NFFT, "stft_logmel: only n_fft = win_length = 1024 is built (got %d)", n_fft);
  F5E_REQUIRE(B > 0 && hop > 0 && n_mels > 0 && n_mels <= 256, "stft_logmel: bad shape");
  F5E_REQUIRE(nw > NFFT / 2, "stft_logmel: reflect padding needs nw > %d samples (got %d)", NFFT / 2, nw);
  const int T = 1 + nw / hop;
  hipLaunchKernelGGL(stft_logmel_kernel, dim3(T, B), dim3(256), 0, st, wav, nw, ldw, window, (const float2*)twiddle,
                     fb, out, T, hop, n_mels);
  F5E_LAUNCH_CHECK("stft_logmel");
  return F5E_OK;
}

int f5e_stft_logmel_banded(hipStream_t st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                           const float* fb_compact, const int* fb_band, int nnz, float* out, int B, int n_fft, int hop,
                           int n_mels) {
  F5E_REQUIRE(wav && window && twiddle && fb_compact && fb_band && out, "stft_logmel_banded: null operand");
  F5E_REQUIRE(n_fft == NFFT, "stft_logmel_banded: only n_fft = win_length = 1024 is built (got %d)", n_fft);
  F5E_REQUIRE(B > 0 && hop > 0 && n_mels > 0 && n_mels <= 256 && nnz > 0 && nnz <= FB_MAX_NNZ,
              "stft_logmel_banded: bad shape (n_mels <= 256, 0 < nnz=%d <= %d)", nnz, FB_MAX_NNZ);
  F5E_REQUIRE(nw > NFFT / 2, "stft_logmel_banded: reflect padding needs nw > %d samples (got %d)", NFFT / 2, nw);
  const int T = 1 + nw / hop;
  hipLaunchKernelGGL(stft_logmel_banded_kernel, dim3(T, B), dim3(256), 0, st, wav, nw, ldw, window,
                     (const float2*)twiddle, fb_compact, fb_band, nnz, out, T, hop, n_mels);
  F5E_LAUNCH_CHECK("stft_logmel_banded");
  return F5E_OK;
}

int f5e_istft_head(hipStream_t st, const float* z, int ldz, const float* window, const float* twiddle,
                   float* frames_ws, float* out, int B, int T, int n_fft, int hop) {
  F5E_REQUIRE(z && window && twiddle && frames_ws && out, "istft_head: null operand");
  F5E_REQUIRE(n_fft == NFFT, "istft_head: only n_fft = 1024 is built (got %d)", n_fft);
  F5E_REQUIRE(B > 0 && T > 1 && hop > 0 && NFFT % hop == 0, "istft_head: bad shape B=%d T=%d hop=%d", B, T, hop);
  F5E_REQUIRE(ldz >= 2 * NBIN, "istft_head: ldz=%d < %d", ldz, 2 * NBIN);
  hipLaunchKernelGGL(istft_frames_kernel, dim3(B * T), dim3(256), 0, st, z, ldz, window, (const float2*)twiddle,
                     frames_ws);
  F5E_LAUNCH_CHECK("istft_frames");
  const int out_len = hop * (T - 1);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((out_len + 255) / 256, B), dim3(256), 0, st, frames_ws, window, out, T, hop,
                     out_len);
  F5E_LAUNCH_CHECK("istft_ola");
  return F5E_OK;
}

}  // extern "C"
