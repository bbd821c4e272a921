// 1-D convolutions on the hot path.
//
// f5e_convpos: one grouped Conv1d(D, D, k=31, groups=G, pad=15) + Mish of ConvPositionEmbedding
//   (reference model/modules.py:167-190, called WITHOUT a mask at backbones/dit.py:176) as an implicit GEMM on
//   v_mfma_f32_16x16x32_bf16: per (sequence, group, 64-token tile) the K dimension is (tap, in-channel) = 31 x 64;
//   the activation tile with its 15-frame halo is staged into LDS once and every tap is just a row offset into it,
//   the per-tap weight tile [64 oc][64 ic] streams through a 3-stage LDS-DMA ring.  Groups narrower than 64 channels
//   (cpg = D/G in {16, 32, 48, 64}) use the same 64 x 64 tile with zero-padded weights: the extra input channels
//   read the neighbouring group's data against zero weights, the extra output channels are not stored.  Rows outside [0, N) are zeros
//   (the Conv1d zero padding); rows between a short item's length and N are real data, exactly as in the reference.
//   mode 0: out_bf16 = mish(conv + bias)                      (first conv, feeds the second)
//   mode 1: out_f32  = mish(conv + bias) + resid_f32          (second conv + the "+ x" of InputEmbedding, dit.py:176)
// f5e_dwconv7: depthwise Conv1d(C, C, k=7, pad=3, groups=C), channels-last fp32 (ConvNeXtV2Block.dwconv,
//   modules.py:250-252,262; Vocos ConvNeXtBlock.dwconv).
// f5e_im2col: [B][T][Cin] -> [B][T][k*Cin] fp32 patches (zero padded) so small dense convs (Vocos embed k7,
//   PPG k5: backbones/dit.py:124-132) run on f5e_gemm_f32.
#include "f5e_common.h"

namespace {

struct ConvPosArgs {
  const bf16* x; int ldx;        // [S*N][D]
  const bf16* w;                 // [G][31][64 oc][64 ic]
  const float* bias;             // [D]
  bf16* out_bf16; int ldo;
  float* out_f32; int ldo32;
  const float* resid; int ldr;
  int S, N, D, mode, tiles_t, cpg;
};

__device__ __forceinline__ void glds16c(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

__global__ __launch_bounds__(256) void convpos_kernel(ConvPosArgs a) {
  constexpr int BT = 64, HALO = 15, XR = BT + 2 * HALO;  // 94 rows
  constexpr int X_BYTES = XR * 128;                      // 12032
  constexpr int W_BYTES = 64 * 128;                      // 8192
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;
  char* Wsb = smem + ((X_BYTES + 255) & ~255);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int tt = bid % a.tiles_t; bid /= a.tiles_t;
  const int G = a.D / a.cpg;
  const int g = bid % G;
  const int c0 = g * a.cpg;  // first channel of this group
  const int seq = bid / G;
  const int t0 = tt * BT;

  // weight tile ring: tile (g, tap) is 8 KB contiguous; chunk i -> (row = i>>3, c = i&7), source chunk swizzled
  const bf16* wg = a.w + (size_t)g * 31 * 4096;
  const bf16* w_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 256 * j;
    const int row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
    w_src[j] = wg + row * 64 + c * 8;
  }
  auto stage_w = [&](int buf, int tap) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      glds16c(w_src[j] + (size_t)tap * 4096, Wsb + buf * W_BYTES + (wave * 64 + 256 * j) * 16);
  };
  stage_w(0, 0);
  stage_w(1, 1);

  // activation tile with halo (register staged: needs zero fill)
  for (int i = tid; i < XR * 8; i += 256) {
    const int row = i >> 3, c = i & 7;
    const int t = t0 - HALO + row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < a.N && c0 + c * 8 < a.D) v = *(const uint4*)(a.x + ((size_t)seq * a.N + t) * a.ldx + c0 + c * 8);
    *(uint4*)(Xs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
  }

  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 3-stage LDS-DMA ring over the 31 taps: two weight tiles in flight, counted vmcnt + raw barrier (as gemm_bf16.hip)
  int buf = 0, nbuf = 2;
  for (int tap = 0; tap < 31; ++tap) {
    if (tap < 30) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tap + 2 < 31) stage_w(nbuf, tap + 2);
    const char* Ws = Wsb + buf * W_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fq;
      bf16x8 xf[2], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm0 + j * 16 + fr + tap;
        xf[j] = *(const bf16x8*)(Xs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn0 + i * 16 + fr;
        wf[i] = *(const bf16x8*)(Ws + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    buf = (buf == 2) ? 0 : buf + 1;
    nbuf = (nbuf == 2) ? 0 : nbuf + 1;
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = t0 + wm0 + j * 16 + fr;
    if (t >= a.N) continue;
    const size_t m = (size_t)seq * a.N + t;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oc = wn0 + i * 16 + fq * 4;
      if (oc >= a.cpg) continue;
      const int n = c0 + oc;
      f32x4 v = acc[i][j] + *(const f32x4*)(a.bias + n);
      v[0] = mish_f(v[0]); v[1] = mish_f(v[1]); v[2] = mish_f(v[2]); v[3] = mish_f(v[3]);
      if (a.mode == 0) {
        *(bf16x4*)(a.out_bf16 + m * a.ldo + n) = f2bf4(v[0], v[1], v[2], v[3]);
      } else {
        v += *(const f32x4*)(a.resid + m * a.ldr + n);
        *(f32x4*)(a.out_f32 + m * a.ldo32 + n) = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void dwconv7_kernel(const float* x, const float* wT, const float* bias, float* y,
                                                       int B, int T, int C) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * T * c4n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const size_t bt = i / c4n;
    const int t = (int)(bt % T);
    const size_t b = bt / T;
    f32x4 acc = *(const f32x4*)(bias + c);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int tj = t + j - 3;
      if (tj >= 0 && tj < T)
        acc += *(const f32x4*)(wT + (size_t)j * C + c) * *(const f32x4*)(x + (b * T + tj) * C + c);
    }
    *(f32x4*)(y + (b * T + t) * C + c) = acc;
  }
}

__global__ __launch_bounds__(256) void im2col_kernel(const float* x, float* col, int B, int T, int Cin, int ksize,
                                                      int pad) {
  const size_t total = (size_t)B * T * ksize * Cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ic = (int)(i % Cin);
    size_t r = i / Cin;
    const int j = (int)(r % ksize);
    r /= ksize;
    const int t = (int)(r % T);
    const size_t b = r / T;
    const int tj = t + j - pad;
    col[i] = (tj >= 0 && tj < T) ? x[(b * T + tj) * Cin + ic] : 0.f;
  }
}

inline int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 4096 ? (g ? g : 1) : 4096);
}

}  // namespace

extern "C" {

int f5e_convpos(hipStream_t st, const void* x, int ldx, const void* w_packed, const float* bias, int mode,
                void* out_bf16, int ldo, float* out_f32, int ldo32, const float* resid, int ldr, int S, int N, int D,
                int groups) {
  F5E_REQUIRE(x && w_packed && bias, "convpos: null operand");
  F5E_REQUIRE(S > 0 && N > 0 && D > 0 && groups > 0 && D % groups == 0, "convpos: bad D=%d / groups=%d", D, groups);
  const int cpg = D / groups;
  F5E_REQUIRE(cpg % 16 == 0 && cpg <= 64, "convpos: channels per group = %d must be 16, 32, 48 or 64", cpg);
  F5E_REQUIRE(ldx % 8 == 0, "convpos: ldx must be a multiple of 8");
  if (mode == 0) F5E_REQUIRE(out_bf16 && ldo % 4 == 0, "convpos: mode 0 needs a bf16 output");
  else {
    F5E_REQUIRE(mode == 1, "convpos: mode must be 0 or 1");
    F5E_REQUIRE(out_f32 && resid && ldo32 % 4 == 0 && ldr % 4 == 0, "convpos: mode 1 needs f32 output + residual");
  }
  ConvPosArgs a{};
  a.x = (const bf16*)x; a.ldx = ldx; a.w = (const bf16*)w_packed; a.bias = bias;
  a.out_bf16 = (bf16*)out_bf16; a.ldo = ldo; a.out_f32 = out_f32; a.ldo32 = ldo32; a.resid = resid; a.ldr = ldr;
  a.S = S; a.N = N; a.D = D; a.mode = mode; a.cpg = cpg;
  a.tiles_t = (N + 63) / 64;
  const int grid = a.tiles_t * groups * S;
  const int lds = ((94 * 128 + 255) & ~255) + 3 * 8192;
  hipLaunchKernelGGL(convpos_kernel, dim3(grid), dim3(256), lds, st, a);
  F5E_LAUNCH_CHECK("convpos");
  return F5E_OK;
}

int f5e_dwconv7(hipStream_t st, const float* x, const float* w_t, const float* bias, float* y, int B, int T, int C) {
  F5E_REQUIRE(x && w_t && bias && y, "dwconv7: null operand");
  F5E_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0, "dwconv7: C=%d must be a positive multiple of 4", C);
  hipLaunchKernelGGL(dwconv7_kernel, dim3(grid_for((size_t)B * T * C / 4)), dim3(256), 0, st, x, w_t, bias, y, B, T, C);
  F5E_LAUNCH_CHECK("dwconv7");
  return F5E_OK;
}

int f5e_im2col(hipStream_t st, const float* x, float* col, int B, int T, int Cin, int ksize, int pad) {
  F5E_REQUIRE(x && col && B > 0 && T > 0 && Cin > 0 && ksize > 0 && pad >= 0, "im2col: bad arguments");
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((size_t)B * T * Cin * ksize)), dim3(256), 0, st, x, col, B, T, Cin,
                     ksize, pad);
  F5E_LAUNCH_CHECK("im2col");
  return F5E_OK;
}

}  // extern "C"
