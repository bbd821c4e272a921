// Exact-fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32) for everything that runs once per sample() call or on the ODE
// state itself, where the reference computes in fp32 and bf16 rounding would cost parity for no speed:
//   time MLP + all AdaLN tables (modules.py:721-731, :311, :332), TextEmbedding / ConvNeXtV2 pointwise convs
//   (backbones/dit.py:54-87, modules.py:265-268), PPGEmbedding linears/convs (dit.py:121-138), the x-part of the
//   input projection (dit.py:173-175) and the Vocos backbone / head linears.
//
//   C[m][n] = epi( sum_k actA(A[m % a_rows][k]) * W[n][k] )
//   epi(v)  = ((act(v + bias[n])) * ch_scale[n] + addend[m % add_rows][n]) * row_scale[m]
//
// 64x64 tile, 4 waves (2x2), K-step 16, LDS rows padded to 17 floats (conflict-free ds_read_b32).
// Operands swapped like gemm_bf16.hip: a lane owns 4 consecutive n of one row m.
#include "f5e_common.h"

namespace {

struct Gemm32Args {
  const float* A; int lda; int a_rows; int a_act;
  const float* W; int ldw;
  const float* bias;
  int act;
  const float* ch_scale;
  const float* addend; int ld_add; int add_rows;
  const float* row_scale;
  float* out; int ldo;
  bf16* out_bf16; int ldo_bf16;
  int M, N, K;
  int tiles_m;
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(Gemm32Args a) {
  constexpr int BK = 16, LDS_ROW = 17;
  __shared__ float As[2][64 * LDS_ROW];
  __shared__ float Ws[2][64 * LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile_n = blockIdx.x / a.tiles_m, tile_m = blockIdx.x - tile_n * a.tiles_m;
  const int m0 = tile_m * 64, n0 = tile_n * 64;

  const int lrow = tid >> 2, lc = (tid & 3) * 4;
  const int am = min(m0 + lrow, a.M - 1) % a.a_rows;
  const int wn = min(n0 + lrow, a.N - 1);
  const float* ap = a.A + (size_t)am * a.lda + lc;
  const float* wp = a.W + (size_t)wn * a.ldw + lc;

  auto gload = [&](const float* p, int k0) -> f32x4 {
    // K % 4 == 0 and lc % 4 == 0: a float4 is either fully inside or fully outside [0, K)
    if (k0 + lc < a.K) return *(const f32x4*)(p + k0);
    return f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto sstore = [&](float* dst, f32x4 v) {
    float* d = dst + lrow * LDS_ROW + lc;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  };

  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = (a.K + BK - 1) / BK;
  f32x4 ra = gload(ap, 0), rw = gload(wp, 0);
  if (a.a_act) { ra[0] = apply_act(ra[0], a.a_act); ra[1] = apply_act(ra[1], a.a_act);
                 ra[2] = apply_act(ra[2], a.a_act); ra[3] = apply_act(ra[3], a.a_act); }
  sstore(As[0], ra);
  sstore(Ws[0], rw);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) {
      ra = gload(ap, (kt + 1) * BK);
      rw = gload(wp, (kt + 1) * BK);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float xf[2], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[j] = As[buf][(wm0 + j * 16 + fr) * LDS_ROW + kk * 4 + fq];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = Ws[buf][(wn0 + i * 16 + fr) * LDS_ROW + kk * 4 + fq];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) {
      if (a.a_act) { ra[0] = apply_act(ra[0], a.a_act); ra[1] = apply_act(ra[1], a.a_act);
                     ra[2] = apply_act(ra[2], a.a_act); ra[3] = apply_act(ra[3], a.a_act); }
      sstore(As[buf ^ 1], ra);
      sstore(Ws[buf ^ 1], rw);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm0 + j * 16 + fr;
    if (m >= a.M) continue;
    const float rs = a.row_scale ? a.row_scale[m] : 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + wn0 + i * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r;
        if (nn >= a.N) continue;
        float v = acc[i][j][r];
        if (a.bias) v += a.bias[nn];
        v = apply_act(v, a.act);
        if (a.ch_scale) v *= a.ch_scale[nn];
        if (a.addend) v += a.addend[(size_t)(m % a.add_rows) * a.ld_add + nn];
        v *= rs;
        if (a.out) a.out[(size_t)m * a.ldo + nn] = v;
        if (a.out_bf16) a.out_bf16[(size_t)m * a.ldo_bf16 + nn] = (bf16)v;
      }
    }
  }
}

}  // namespace

extern "C" int f5e_gemm_f32(hipStream_t st, const float* A, int lda, int a_rows, int a_act, const float* W, int ldw,
                            const float* bias, int act, const float* ch_scale, const float* addend, int ld_add,
                            int add_rows, const float* row_scale, float* out, int ldo, void* out_bf16, int ldo_bf16,
                            int M, int N, int K) {
  F5E_REQUIRE(A && W && (out || out_bf16), "gemm_f32: null operand");
  F5E_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32: empty problem M=%d N=%d K=%d", M, N, K);
  F5E_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm_f32: K, lda, ldw must be multiples of 4 (K=%d)", K);
  F5E_REQUIRE(a_rows > 0, "gemm_f32: a_rows must be positive");
  if (addend) F5E_REQUIRE(add_rows > 0, "gemm_f32: add_rows must be positive");
  Gemm32Args a{};
  a.A = A; a.lda = lda; a.a_rows = a_rows; a.a_act = a_act; a.W = W; a.ldw = ldw; a.bias = bias; a.act = act;
  a.ch_scale = ch_scale; a.addend = addend; a.ld_add = ld_add; a.add_rows = add_rows > 0 ? add_rows : 1;
  a.row_scale = row_scale; a.out = out; a.ldo = ldo; a.out_bf16 = (bf16*)out_bf16; a.ldo_bf16 = ldo_bf16;
  a.M = M; a.N = N; a.K = K;
  a.tiles_m = (M + 63) / 64;
  const int grid = a.tiles_m * ((N + 63) / 64);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(grid), dim3(256), 0, st, a);
  F5E_LAUNCH_CHECK("gemm_f32");
  return F5E_OK;
}
