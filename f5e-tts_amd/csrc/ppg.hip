// Kernels of the PPG extractor front (reference ppg/ppg_model.py:58-169): the wenet Conformer encoder runs
// its linears on f5e_gemm_f32 and its LayerNorms on f5e_layernorm; what is left is here.
//
//   f5e_kaldi_fbank    torchaudio.compliance.kaldi.fbank as called at ppg/wenet/dataset/feats.py:66-72 (25 ms / 10 ms frames
//                      at 16 kHz, 512-point FFT, povey window, pre-emphasis 0.97, DC removal, power spectrum, 80 kaldi mel
//                      bins from 20 Hz to Nyquist, log(max(., eps)); dither 0, snip_edges): one frame per workgroup.
//   f5e_glu            nn.functional.glu over the channel axis (ppg/wenet/transformer/convolution.py:119), channels-last.
//   f5e_dwconv         depthwise Conv1d(C, C, k, padding (k - 1) / 2, groups C), channels-last fp32, any odd k <= 31
//                      (ConvolutionModule.depthwise_conv, k = 15; eval-mode BatchNorm folded into weight / bias on the host).
//   f5e_softmax_rows   softmax(scale * x[row, :len]) with the keys >= len of the row's sequence set to 0
//                      (MultiHeadedAttention.forward_attention, attention.py:75-87: masked_fill(-inf), softmax, masked_fill(0)).
#include "f5e_common.h"

namespace {

constexpr int FB_NFFT = 512, FB_LOG2 = 9, FB_NBIN = FB_NFFT / 2 + 1;

__device__ __forceinline__ int bitrev9(int i) { return (int)(__brev((unsigned)i) >> 23); }

// x: bit-reversed input in LDS, 256 threads = one butterfly each per stage; tw[k] = (cos, -sin)(2 pi k / 512), k < 256
__device__ __forceinline__ void fft512(float2* x, const float2* tw, int tid) {
#pragma unroll 1
  for (int s = 1; s <= FB_LOG2; ++s) {
    const int half = 1 << (s - 1);
    __syncthreads();
    const int pos = tid & (half - 1);
    const int i0 = ((tid >> (s - 1)) << s) + pos, i1 = i0 + half;
    const float2 w = tw[pos << (FB_LOG2 - s)];
    const float2 u = x[i0], v = x[i1];
    const float tr = w.x * v.x - w.y * v.y, ti = w.x * v.y + w.y * v.x;
    x[i0] = make_float2(u.x + tr, u.y + ti);
    x[i1] = make_float2(u.x - tr, u.y - ti);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* wav, int ldw, const float* window, const float2* tw,
                                                           const float* fb, float* out, int T, int win, int shift,
                                                           int n_mels, float in_scale, float preemph, float eps) {
  __shared__ float2 x[FB_NFFT];
  __shared__ float raw[FB_NFFT];
  __shared__ float pw[FB_NBIN + 3];
  __shared__ float part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = blockIdx.x, b = blockIdx.y;
  const float* w = wav + (size_t)b * ldw + (size_t)f * shift;
  // frame (snip_edges: frame f = samples [f shift, f shift + win)), scaled to the int16 range as the reference does
  float s = 0.f;
  for (int j = tid; j < FB_NFFT; j += 256) {
    const float v = j < win ? w[j] * in_scale : 0.f;
    raw[j] = v;
    s += v;
  }
  s = wave_sum(s);
  if (lane == 0) part[wave] = s;
  __syncthreads();
  const float mean = ((part[0] + part[1]) + (part[2] + part[3])) / (float)win;   // remove_dc_offset
  for (int j = tid; j < FB_NFFT; j += 256) {
    float v = 0.f;
    if (j < win) {
      const float cur = raw[j] - mean, prev = raw[j > 0 ? j - 1 : 0] - mean;      // replicate-padded pre-emphasis
      v = (cur - preemph * prev) * window[j];
    }
    x[bitrev9(j)] = make_float2(v, 0.f);
  }
  fft512(x, tw, tid);
  for (int k = tid; k < FB_NBIN; k += 256) pw[k] = x[k].x * x[k].x + x[k].y * x[k].y;   // use_power
  __syncthreads();
  if (tid < n_mels) {
    float acc = 0.f;
    for (int k = 0; k < FB_NBIN; ++k) acc += pw[k] * fb[(size_t)k * n_mels + tid];
    out[((size_t)b * T + f) * n_mels + tid] = logf(fmaxf(acc, eps));
  }
}

__global__ void glu_kernel(const float* x, int ldx, float* y, int ldy, size_t rows, int C) {
  const int c4n = C / 4;
  const size_t total = rows * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const size_t r = i / c4n;
    const f32x4 a = *(const f32x4*)(x + r * ldx + c), g = *(const f32x4*)(x + r * ldx + C + c);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = a[u] / (1.0f + __expf(-g[u]));
    *(f32x4*)(y + r * ldy + c) = o;
  }
}

__global__ __launch_bounds__(256) void dwconv_kernel(const float* x, const float* wT, const float* bias, const float* keep,
                                                      float* y, int B, int T, int C, int K) {
  const int c4n = C / 4, pad = (K - 1) / 2;
  const size_t total = (size_t)B * T * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const size_t bt = i / c4n;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    f32x4 acc = *(const f32x4*)(bias + c);
    for (int j = 0; j < K; ++j) {
      const int tj = t + j - pad;
      // keep (optional, [B][T] 0/1): the reference zeroes padded frames BEFORE the conv (convolution.py:100-101)
      if (tj >= 0 && tj < T && (!keep || keep[b * T + tj] != 0.f))
        acc += *(const f32x4*)(wT + (size_t)j * C + c) * *(const f32x4*)(x + (b * T + tj) * C + c);
    }
    *(f32x4*)(y + (b * T + t) * C + c) = acc;
  }
}

// one wave per row; row r belongs to sequence r / rows_per_seq whose valid key count is kv_len[seq] (or L)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, int ldx, float* y, int ldy, const int* kv_len,
                                                            size_t rows, int rows_per_seq, int L, float scale) {
  const int lane = threadIdx.x & 63;
  const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int len = kv_len ? min(kv_len[r / rows_per_seq], L) : L;
  const float* xp = x + r * ldx;
  float* yp = y + r * ldy;
  float mx = -INFINITY;
  for (int k = lane; k < len; k += 64) mx = fmaxf(mx, xp[k] * scale);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int k = lane; k < len; k += 64) sm += __expf(xp[k] * scale - mx);
  sm = wave_sum(sm);
  const float inv = sm > 0.f ? 1.0f / sm : 0.f;
  for (int k = lane; k < ldy; k += 64) yp[k] = k < len ? __expf(xp[k] * scale - mx) * inv : 0.f;   // pad columns too
}

inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}

}  // namespace

extern "C" {

int f5e_kaldi_fbank(hipStream_t st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                    const float* fb, float* out, int B, int win, int shift, int n_mels, float in_scale, float preemph,
                    float eps) {
  F5E_REQUIRE(wav && window && twiddle && fb && out, "kaldi_fbank: null operand");
  F5E_REQUIRE(B > 0 && win > 0 && win <= FB_NFFT && shift > 0 && n_mels > 0 && n_mels <= 256 && nw >= win,
              "kaldi_fbank: bad shape (window %d <= %d samples, nw=%d)", win, FB_NFFT, nw);
  const int T = 1 + (nw - win) / shift;
  hipLaunchKernelGGL(kaldi_fbank_kernel, dim3(T, B), dim3(256), 0, st, wav, ldw, window, (const float2*)twiddle, fb, out,
                     T, win, shift, n_mels, in_scale, preemph, eps);
  F5E_LAUNCH_CHECK("kaldi_fbank");
  return F5E_OK;
}

int f5e_glu(hipStream_t st, const float* x, int ldx, float* y, int ldy, long long rows, int C) {
  F5E_REQUIRE(x && y && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= 2 * C && ldy >= C,
              "glu: bad arguments");
  hipLaunchKernelGGL(glu_kernel, dim3(grid_for((size_t)rows * C / 4)), dim3(256), 0, st, x, ldx, y, ldy, (size_t)rows, C);
  F5E_LAUNCH_CHECK("glu");
  return F5E_OK;
}

int f5e_dwconv(hipStream_t st, const float* x, const float* w_t, const float* bias, const float* keep, float* y, int B,
               int T, int C, int K) {
  F5E_REQUIRE(x && w_t && bias && y, "dwconv: null operand");
  F5E_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= 31 && (K & 1), "dwconv: C %% 4 == 0 and odd K <= 31");
  hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for((size_t)B * T * C / 4)), dim3(256), 0, st, x, w_t, bias, keep, y, B, T,
                     C, K);
  F5E_LAUNCH_CHECK("dwconv");
  return F5E_OK;
}

int f5e_softmax_rows(hipStream_t st, const float* x, int ldx, float* y, int ldy, const int* kv_len, long long rows,
                     int rows_per_seq, int L, float scale) {
  F5E_REQUIRE(x && y && rows > 0 && rows_per_seq > 0 && L > 0 && ldx >= L && ldy >= L, "softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ldx, y, ldy, kv_len,
                     (size_t)rows, rows_per_seq, L, scale);
  F5E_LAUNCH_CHECK("softmax_rows");
  return F5E_OK;
}

}  // extern "C"
