// Generic epilogue of the bf16 GEMM family for a wave that owns a (16 TM) x (16 TN) sub-tile with the swapped-operand
// accumulator layout   acc[i][j][r] = C[m = mbase + 16 j + fr][n = nbase + 16 i + 4 fq + r]   (fr = lane & 15,
// fq = lane >> 4): bias, then one of  bf16 store | GELU(tanh) + bf16 | fp32 store | gate * v + residual (masked rows
// skipped) | qk RMSNorm + RoPE + fragment-major q / k / v scatter.  Same arithmetic, in the same order, as the inline
// epilogue of gemm_bf16_kernel (gemm_bf16.hip), which additionally prefetches its operands and carries the fused AdaLN.
#pragma once
#include "gemm_bf16_args.h"

namespace f5e_gemm {

template <int EPI, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, const f32x4 (&acc)[TN][TM], int mbase, int nbase, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  const size_t eoff = (EPI == EPI_GATE_RES && a.eval_ptr) ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = mbase + j * 16 + fr;
    if (m >= a.M) continue;
    int seq = 0, pos = m;
    if (EPI == EPI_GATE_RES || EPI == EPI_QKV_ROPE) {
      seq = m / a.rows_per_seq;
      pos = m - seq * a.rows_per_seq;
    }
    // optional qk RMSNorm (reference modules.py:464-467 + :275-294): a whole head (64 columns) sits in one wave when
    // TN == 4: 16 values in-lane, the other 48 in the lanes fr + 16 / 32 / 48
    float qk_rn = 1.0f;
    const float* qk_w = nullptr;
    if constexpr (EPI == EPI_QKV_ROPE && TN == 4) {
      const int which_w = nbase / (a.heads * 64);
      if (a.qn_w && which_w < 2) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          f32x4 t = acc[i][j];
          if (a.bias) t += *(const f32x4*)(a.bias + nbase + i * 16 + fq * 4);
          ss += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
        }
        ss = add_xor32(add_xor16(ss));
        qk_rn = rsqrtf(ss * (1.0f / 64.0f) + a.qk_eps);
        qk_w = which_w == 0 ? a.qn_w : a.kn_w;
      }
    }
    bool live = true;
    if (EPI == EPI_GATE_RES) live = (a.seq_len == nullptr) || (pos < a.seq_len[seq]);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = nbase + i * 16 + fq * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if (a.bias) v += *(const f32x4*)(a.bias + n);
      if (EPI == EPI_BF16) {
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) = f2bf4(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_BF16_GELU) {
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) =
            f2bf4(gelu_tanh_f(v[0]), gelu_tanh_f(v[1]), gelu_tanh_f(v[2]), gelu_tanh_f(v[3]));
      } else if (EPI == EPI_F32) {
        *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
      } else if (EPI == EPI_GATE_RES) {
        if (live) {
          const f32x4 g = *(const f32x4*)(a.gate + eoff + (size_t)(seq % a.gate_rows) * a.gate_stride + n);
          float* xp = a.resid + (size_t)m * a.ldr + n;
          f32x4 x = *(const f32x4*)xp;
          x += g * v;
          *(f32x4*)xp = x;
        }
      } else if (EPI == EPI_QKV_ROPE) {
        const int inner = a.heads * 64;
        const int which = n / inner;
        const int nn = n - which * inner;
        const int head = nn >> 6, d = nn & 63;
        if (qk_w) v = v * qk_rn * *(const f32x4*)(qk_w + d);
        if (which < 2 && head < a.rope_heads) {
          const f32x4 cs = *(const f32x4*)(a.cos_sin + ((size_t)pos * 32 + (d >> 1)) * 2);
          const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
          v[0] = x0 * cs[0] - x1 * cs[1];
          v[1] = x1 * cs[0] + x0 * cs[1];
          v[2] = x2 * cs[2] - x3 * cs[3];
          v[3] = x3 * cs[2] + x2 * cs[3];
        }
        if (which == 0) v *= a.q_scale;   // softmax scale and log2(e), folded into q (attention.hip)
        const size_t sh = (size_t)seq * a.heads + head;
        // fragment-major layouts consumed by attention.hip (index maps documented there)
        const int tile = pos >> 5, pr = pos & 31;
        if (which < 2) {
          bf16* dst = (which == 0 ? a.q : a.k) + sh * a.n_pad * 64 +
                      ((size_t)(tile * 4 + (d >> 4)) * 32 + pr) * 16 + ((d >> 3) & 1) * 8 + (d & 7);
          *(bf16x4*)dst = f2bf4(v[0], v[1], v[2], v[3]);
        } else {
          const int s16 = pr >> 4, k16 = pr & 15;
          const int jj = ((k16 >> 3) << 2) | (k16 & 3), hk = (k16 >> 2) & 1;
          bf16* dst = a.vt + sh * a.n_pad * 64 +
                      ((((size_t)(tile * 2 + s16) * 2 + (d >> 5)) * 32 + (d & 31)) * 2 + hk) * 8 + jj;
          dst[0] = (bf16)v[0];
          dst[16] = (bf16)v[1];
          dst[32] = (bf16)v[2];
          dst[48] = (bf16)v[3];
        }
      }
    }
  }
}

}  // namespace f5e_gemm
