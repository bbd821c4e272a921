// LayerNorm family (K7/K12/K13 of SURVEY 2.3) and GRN (K3). All HBM-bound: one wave per row, fp32 statistics.
//
// f5e_layernorm: y = LN(x; eps, no affine) [* gamma + beta] [* (1 + scale[seq]) + shift[seq]]
//   - AdaLayerNorm / ff_norm modulate / AdaLayerNorm_Final: reference model/modules.py:301-336, :637
//   - affine LN of ConvNeXtV2Block (modules.py:253,264) and of the Vocos backbone
// Statistics follow F.layer_norm: mean, then biased variance of the centred values (two passes in registers).
#include "f5e_common.h"

namespace {

struct LnArgs {
  const float* x; int ldx;
  void* y; int ldy; int y_bf16;
  const float* gamma; const float* beta;      // [D] or null
  const float* scale; const float* shift;     // [mod_rows][mod_stride] or null
  int mod_stride, mod_rows, rows_per_seq;
  int rows, D;
  float eps;
  const int* eval_ptr; int eval_stride;       // tables advance by eval_stride floats per ODE evaluation
};

template <int VPL>  // float4 vectors per lane: D = VPL * 256
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const float* xp = a.x + (size_t)row * a.ldx;
  f32x4 v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4*)(xp + (i * 64 + lane) * 4);
  // modulation rows are fetched while the x loads are in flight (the kernel is latency-, not bandwidth-limited)
  const int mrow = a.scale ? (row / a.rows_per_seq) % a.mod_rows : 0;
  const size_t eoff = (a.scale && a.eval_ptr) ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
  f32x4 sc[VPL], sh[VPL];
  if (a.scale) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      sc[i] = *(const f32x4*)(a.scale + eoff + (size_t)mrow * a.mod_stride + c);
      sh[i] = *(const f32x4*)(a.shift + eoff + (size_t)mrow * a.mod_stride + c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)a.D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] -= mean;
    ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)a.D + a.eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    f32x4 y = v[i] * rstd;
    if (a.gamma) y = y * *(const f32x4*)(a.gamma + c) + *(const f32x4*)(a.beta + c);
    if (a.scale) y = y * (1.0f + sc[i]) + sh[i];
    if (a.y_bf16)
      *(bf16x4*)((bf16*)a.y + (size_t)row * a.ldy + c) = f2bf4(y[0], y[1], y[2], y[3]);
    else
      *(f32x4*)((float*)a.y + (size_t)row * a.ldy + c) = y;
  }
}

// Any D % 4 == 0 up to 2048 (narrow models: the reduced PPG encoder of the tests, D = 64): same arithmetic, lanes past D / 4
// idle, up to 8 vectors per lane.  fp32 output, optional affine, no modulation.
__global__ __launch_bounds__(256) void layernorm_any_kernel(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const float* xp = a.x + (size_t)row * a.ldx;
  const int nv = a.D / 4;
  f32x4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < nv ? *(const f32x4*)(xp + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)a.D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i * 64 + lane < nv) {
      v[i] -= mean;
      ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
  const float rstd = rsqrtf(wave_sum(ss) / (float)a.D + a.eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < a.D) {
      f32x4 y = v[i] * rstd;
      if (a.gamma) y = y * *(const f32x4*)(a.gamma + c) + *(const f32x4*)(a.beta + c);
      *(f32x4*)((float*)a.y + (size_t)row * a.ldy + c) = y;
    }
  }
}

// Head of the fused-AdaLN chain (see f5e_ln_fuse in the ABI header): no normalisation here, only the row statistics, the
// row's centring offset (its exact mean) and the centred, pre-scaled bf16 copy the first consumer GEMM runs on.
template <int VPL>
__global__ __launch_bounds__(256) void adaln_pre_kernel(const float* x, int ldx, bf16* xs, int ld_xs, const float* scale,
                                                        int mod_stride, int mod_rows, int rows_per_seq,
                                                        const int* eval_ptr, int eval_stride, float* stats, int parts,
                                                        float* row_mean, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xp = x + (size_t)row * ldx;
  f32x4 v[VPL], sc[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4*)(xp + (i * 64 + lane) * 4);
  const size_t moff = (eval_ptr ? (size_t)load_uniform_i32(eval_ptr) * eval_stride : 0) + (size_t)((row / rows_per_seq) % mod_rows) * mod_stride;
#pragma unroll
  for (int i = 0; i < VPL; ++i) sc[i] = *(const f32x4*)(scale + moff + (i * 64 + lane) * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 d = v[i] - mean;
    ss += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  const float m2 = wave_sum(ss);
  // statistics are kept RELATIVE to the offset the row's xs was centred with (row_mean): here the exact mean, so 0
  if (lane < parts) {
    stats[((size_t)row * parts + lane) * 2] = 0.f;
    stats[((size_t)row * parts + lane) * 2 + 1] = m2 / (float)parts;
  }
  if (lane == 0) row_mean[row] = mean;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const f32x4 y = (v[i] - mean) * (1.0f + sc[i]);
    *(bf16x4*)(xs + (size_t)row * ld_xs + (i * 64 + lane) * 4) = f2bf4(y[0], y[1], y[2], y[3]);
  }
}

// x_transformers.RMSNorm as UNetT uses it (reference backbones/unett.py:151,161,178): F.normalize(x, dim=-1) * sqrt(D) * g
template <int VPL>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* x, int ldx, void* y, int ldy, int y_bf16,
                                                      const float* g, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xp = x + (size_t)row * ldx;
  f32x4 v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *(const f32x4*)(xp + (i * 64 + lane) * 4);
    ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  }
  const float scale = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 o = v[i] * scale * *(const f32x4*)(g + c);
    if (y_bf16) *(bf16x4*)((bf16*)y + (size_t)row * ldy + c) = f2bf4(o[0], o[1], o[2], o[3]);
    else *(f32x4*)((float*)y + (size_t)row * ldy + c) = o;
  }
}

// GRN pass 1: gx[b][c] = sqrt(sum_t x[b][t][c]^2)   x: [B][T][C] fp32
__global__ __launch_bounds__(256) void grn_norm_kernel(const float* x, float* gx, int T, int C) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int t = w; t < T; t += 4) {
      const float v = x[((size_t)b * T + t) * C + c];
      s += v * v;
    }
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < C) {
    const int l = threadIdx.x;
    gx[(size_t)b * C + c] = sqrtf((red[0][l] + red[1][l]) + (red[2][l] + red[3][l]));
  }
}

// GRN pass 1, C % 4 == 0: 16 row groups x 16 channel quads per workgroup, 8 independent 16-byte loads in flight per
// thread (the scalar kernel above walks T / 4 rows with one 4-byte load at a time: 35 us for 469 x 1024)
__global__ __launch_bounds__(256) void grn_norm4_kernel(const float* x, float* gx, int T, int C) {
  __shared__ f32x4 red[16][16];
  const int b = blockIdx.y;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + q * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* xp = x + (size_t)b * T * C + c;
    int t = rg;
    for (; t + 7 * 16 < T; t += 8 * 16) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(xp + (size_t)(t + 16 * u) * C);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u] * v[u];
    }
    for (; t < T; t += 16) {
      const f32x4 v = *(const f32x4*)(xp + (size_t)t * C);
      s += v * v;
    }
  }
  red[rg][q] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    f32x4 a = red[0][q];
#pragma unroll
    for (int r = 1; r < 16; ++r) a += red[r][q];
    *(f32x4*)(gx + (size_t)b * C + c) = f32x4{sqrtf(a[0]), sqrtf(a[1]), sqrtf(a[2]), sqrtf(a[3])};
  }
}

// GRN pass 2: y = gamma * (x * gx / (mean_c gx + 1e-6)) + beta + x        (modules.py:231-234)
__global__ __launch_bounds__(256) void grn_apply_kernel(const float* x, const float* gx, const float* gamma,
                                                        const float* beta, float* y, int T, int C, int vec) {
  __shared__ float s_mean;
  const int b = blockIdx.y;
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) s += gx[(size_t)b * C + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) s_mean = s / (float)C;
  }
  __syncthreads();
  const float inv = 1.0f / (s_mean + 1e-6f);
  const size_t total = (size_t)T * C;
  if (vec) {  // C % 4 == 0 and 16-byte aligned operands (host check): 4 channels per thread: no per-element modulo, 16-byte accesses
    const size_t quads = total / 4;
    const int cq = C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
      const int c = (int)(i % cq) * 4;
      const f32x4 v = *(const f32x4*)(x + (size_t)b * total + i * 4);
      const f32x4 g = *(const f32x4*)(gx + (size_t)b * C + c);
      *(f32x4*)(y + (size_t)b * total + i * 4) = *(const f32x4*)(gamma + c) * (v * (g * inv)) + *(const f32x4*)(beta + c) + v;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float v = x[(size_t)b * total + i];
    y[(size_t)b * total + i] = gamma[c] * (v * (gx[(size_t)b * C + c] * inv)) + beta[c] + v;
  }
}

}  // namespace

extern "C" {

int f5e_layernorm(hipStream_t st, const float* x, int ldx, void* y, int ldy, int y_bf16, const float* gamma,
                  const float* beta, const float* scale, const float* shift, int mod_stride, int mod_rows,
                  int rows_per_seq, const int* eval_ptr, int eval_stride, int rows, int D, float eps) {
  F5E_REQUIRE(x && y && rows > 0, "layernorm: null/empty");
  F5E_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm: ldx/ldy must be multiples of 4");
  F5E_REQUIRE((gamma == nullptr) == (beta == nullptr), "layernorm: gamma and beta go together");
  F5E_REQUIRE((scale == nullptr) == (shift == nullptr), "layernorm: scale and shift go together");
  if (scale) F5E_REQUIRE(mod_rows > 0 && rows_per_seq > 0 && mod_stride % 4 == 0, "layernorm: bad modulation table");
  LnArgs a{x, ldx, y, ldy, y_bf16, gamma, beta, scale, shift, mod_stride, mod_rows, rows_per_seq, rows, D, eps,
           eval_ptr, eval_stride};
  const dim3 grid((rows + 3) / 4), block(256);
  if (D % 256 != 0) {   // narrow / odd widths: plain (optionally affine) fp32 LayerNorm only
    F5E_REQUIRE(D % 4 == 0 && D >= 4 && D <= 2048 && !scale && !y_bf16,
                "layernorm: D=%d must be a multiple of 256 in [256, 2048] (any multiple of 4 for an fp32 LayerNorm without "
                "modulation)", D);
    hipLaunchKernelGGL(layernorm_any_kernel, grid, block, 0, st, a);
    F5E_LAUNCH_CHECK("layernorm");
    return F5E_OK;
  }
  F5E_REQUIRE(D % 256 == 0 && D >= 256 && D <= 2048, "layernorm: D=%d must be a multiple of 256 in [256, 2048]", D);
  switch (D / 256) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(layernorm_kernel<6>, grid, block, 0, st, a); break;
    case 7: hipLaunchKernelGGL(layernorm_kernel<7>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, st, a); break;
  }
  F5E_LAUNCH_CHECK("layernorm");
  return F5E_OK;
}

int f5e_adaln_pre(hipStream_t st, const float* x, int ldx, void* xs, int ld_xs, const float* scale, int mod_stride,
                  int mod_rows, int rows_per_seq, const int* eval_ptr, int eval_stride, float* stats, int parts,
                  float* row_mean, int rows, int D) {
  F5E_REQUIRE(x && xs && scale && stats && row_mean && rows > 0, "adaln_pre: null/empty");
  F5E_REQUIRE(D % 256 == 0 && D >= 256 && D <= 2048, "adaln_pre: D=%d must be a multiple of 256 in [256, 2048]", D);
  F5E_REQUIRE(ldx % 4 == 0 && ld_xs % 4 == 0 && mod_stride % 4 == 0, "adaln_pre: strides must be multiples of 4");
  F5E_REQUIRE(parts > 0 && parts <= 32 && mod_rows > 0 && rows_per_seq > 0, "adaln_pre: bad parts / modulation rows");
  const dim3 grid((rows + 3) / 4), block(256);
#define F5E_PRE_CASE(V)                                                                                               \
  case V: hipLaunchKernelGGL(adaln_pre_kernel<V>, grid, block, 0, st, x, ldx, (bf16*)xs, ld_xs, scale, mod_stride,    \
                             mod_rows, rows_per_seq, eval_ptr, eval_stride, stats, parts, row_mean, rows, D); break;
  switch (D / 256) {
    F5E_PRE_CASE(1) F5E_PRE_CASE(2) F5E_PRE_CASE(3) F5E_PRE_CASE(4) F5E_PRE_CASE(5) F5E_PRE_CASE(6) F5E_PRE_CASE(7)
    default: hipLaunchKernelGGL(adaln_pre_kernel<8>, grid, block, 0, st, x, ldx, (bf16*)xs, ld_xs, scale, mod_stride,
                                mod_rows, rows_per_seq, eval_ptr, eval_stride, stats, parts, row_mean, rows, D); break;
  }
#undef F5E_PRE_CASE
  F5E_LAUNCH_CHECK("adaln_pre");
  return F5E_OK;
}

int f5e_l2norm(hipStream_t st, const float* x, int ldx, void* y, int ldy, int y_bf16, const float* g, int rows, int D) {
  F5E_REQUIRE(x && y && g && rows > 0, "l2norm: null/empty");
  F5E_REQUIRE(D % 256 == 0 && D >= 256 && D <= 2048, "l2norm: D=%d must be a multiple of 256 in [256, 2048]", D);
  F5E_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "l2norm: ldx/ldy must be multiples of 4");
  const dim3 grid((rows + 3) / 4), block(256);
  switch (D / 256) {
    case 1: hipLaunchKernelGGL(l2norm_kernel<1>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 2: hipLaunchKernelGGL(l2norm_kernel<2>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 3: hipLaunchKernelGGL(l2norm_kernel<3>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 4: hipLaunchKernelGGL(l2norm_kernel<4>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 5: hipLaunchKernelGGL(l2norm_kernel<5>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 6: hipLaunchKernelGGL(l2norm_kernel<6>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    case 7: hipLaunchKernelGGL(l2norm_kernel<7>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
    default: hipLaunchKernelGGL(l2norm_kernel<8>, grid, block, 0, st, x, ldx, y, ldy, y_bf16, g, rows, D); break;
  }
  F5E_LAUNCH_CHECK("l2norm");
  return F5E_OK;
}

int f5e_grn(hipStream_t st, const float* x, float* y, float* gx_ws, const float* gamma, const float* beta, int B,
            int T, int C) {
  F5E_REQUIRE(x && y && gx_ws && gamma && beta && B > 0 && T > 0 && C > 0, "grn: null/empty");
  const bool vec = C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gx_ws | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
  if (vec) hipLaunchKernelGGL(grn_norm4_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, x, gx_ws, T, C);
  else hipLaunchKernelGGL(grn_norm_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, x, gx_ws, T, C);
  F5E_LAUNCH_CHECK("grn_norm");
  const size_t total = (size_t)T * C;
  const size_t items = vec ? total / 4 : total;
  const int gx = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
  hipLaunchKernelGGL(grn_apply_kernel, dim3(gx, B), dim3(256), 0, st, x, gx_ws, gamma, beta, y, T, C, vec ? 1 : 0);
  F5E_LAUNCH_CHECK("grn_apply");
  return F5E_OK;
}

}  // extern "C"
