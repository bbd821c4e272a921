// 1-D convolutions on the hot path.
//
// f5e_convpos: one grouped Conv1d(D, D, k=31, groups=G, pad=15) + Mish of ConvPositionEmbedding
//   (reference model/modules.py:167-190, called WITHOUT a mask at backbones/dit.py:176) as an implicit GEMM on
//   v_mfma_f32_16x16x32_bf16: per (sequence, group, 64-token tile) the K dimension is (tap, in-channel) = 31 x 64;
//   the activation tile with its 15-frame halo is staged into LDS once and every tap is just a row offset into it,
//   the per-tap weight tiles [64 oc][64 ic] stream through LDS-DMA rings (two kernels, see f5e_convpos).  Workgroups
//   are ordered group-fastest: id % 8 picks the XCD, so the token tiles of one group share an L2.  Groups narrower
//   than 64 channels (cpg = D/G in {16, 32, 48, 64}) use the same 64 x 64 tile with zero-padded weights: the extra
//   input channels read the neighbouring group's data against zero weights, the extra output channels are not stored.
//   Rows outside [0, N) are zeros (the Conv1d zero padding); rows between a short item's length and N are real data,
//   exactly as in the reference.
//   mode 0: out_bf16 = mish(conv + bias)                      (first conv, feeds the second)
//   mode 1: out_f32  = mish(conv + bias) + resid_f32          (second conv + the "+ x" of InputEmbedding, dit.py:176)
// f5e_dwconv7: depthwise Conv1d(C, C, k=7, pad=3, groups=C), channels-last fp32 (ConvNeXtV2Block.dwconv,
//   modules.py:250-252,262; Vocos ConvNeXtBlock.dwconv).
// f5e_im2col: [B][T][Cin] -> [B][T][k*Cin] fp32 patches (zero padded) so small dense convs (Vocos embed k7,
//   PPG k5: backbones/dit.py:124-132) run on f5e_gemm_f32.
#include <cstdlib>
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
  unsigned long long* trace;  // diagnostics: 8 stamps per workgroup (tools/convpos_time.py)
};

__device__ __forceinline__ void glds16c(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

__device__ const uint4 conv_zero16 = {0u, 0u, 0u, 0u};

// s_waitcnt with a literal count chosen by a value that is a constant after unrolling
__device__ __forceinline__ void wait_vm_lgkm0(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
}

// Output-split kernel: 2 x 2 waves own 32 x 32 outputs each and walk all 31 taps together; NST = depth of the
// weight ring (4: 44 KB of LDS, three workgroups per CU).  TRC = diagnostic build with timestamps.
template <int NST, bool TRC = false>
__global__ __launch_bounds__(256) void convpos_kernel(ConvPosArgs a) {
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (TRC) { ts[0] = __builtin_amdgcn_s_memrealtime(); ts[1] = __builtin_amdgcn_s_memtime(); }
  constexpr int BT = 64, HALO = 15, XR = BT + 2 * HALO;  // 94 rows
  constexpr int X_BYTES = 96 * 128;                      // 94 rows + 2 pad rows (3 DMA rounds of 256 chunks)
  constexpr int W_BYTES = 64 * 128;                      // 8192
  constexpr int TAPS = 31;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;
  char* Wsb = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // group fastest: workgroup id % 8 picks the XCD, so the token tiles of one group share an L2 and each XCD streams
  // 2 of the 16 groups' weights instead of all of them
  const int g = blockIdx.x, tt = blockIdx.y, seq = blockIdx.z;
  const int c0 = g * a.cpg;  // first channel of this group
  const int t0 = tt * BT;

  // epilogue operands first (oldest in the in-order vmcnt queue: the counted waits below are unaffected)
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 rv[2][2], bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc = wn0 + i * 16 + fq * 4;
    bv[i] = *(const f32x4*)(a.bias + c0 + (oc < a.cpg ? oc : 0));
  }
  if (a.mode == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = t0 + wm0 + j * 16 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oc = wn0 + i * 16 + fq * 4;
        rv[i][j] = *(const f32x4*)(a.resid + ((size_t)seq * a.N + (t < a.N ? t : a.N - 1)) * a.ldr + c0 + (oc < a.cpg ? oc : 0));
      }
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // activation tile with halo, LDS-DMA like the weights (chunk i of the tile <- source chunk swizzled); rows outside
  // [0, N) and the two pad rows read a 16-byte block of zeros instead
  {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = tid + 256 * r;
      const int row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
      const int t = t0 - HALO + row;
      const void* src = &conv_zero16;
      if (row < XR && t >= 0 && t < a.N && c0 + c * 8 < a.D) src = a.x + ((size_t)seq * a.N + t) * a.ldx + c0 + c * 8;
      glds16c(src, Xs + (wave * 64 + 256 * r) * 16);
    }
  }

  // weight tile ring: tile (g, tap) is 8 KB contiguous; chunk i -> (row = i>>3, c = i&7), source chunk swizzled
  const bf16* wg = a.w + (size_t)g * TAPS * 4096;
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
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage_w(s, s);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // NST-stage LDS-DMA ring over the 31 taps, counted vmcnt + raw barrier (as gemm_bf16.hip).  Fully unrolled: the
  // wait counts and buffer indices are literals.  At the top of tap: tiles 0 .. min(31, tap + NST - 1) - 1 are issued,
  // tile `tap` must have landed; the barrier also says every wave is done with buffer (tap - 1) % NST, which the
  // tile issued right after it overwrites.
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int issued = tap + NST - 1 < TAPS ? tap + NST - 1 : TAPS;
    wait_vm_lgkm0(2 * (issued - tap - 1));
    __builtin_amdgcn_s_barrier();
    if constexpr (TRC) { if (tap == 0) ts[2] = __builtin_amdgcn_s_memtime(); if (tap == 10) ts[3] = __builtin_amdgcn_s_memtime(); if (tap == 20) ts[4] = __builtin_amdgcn_s_memtime(); }
    const char* Ws = Wsb + (tap % NST) * W_BYTES;
    // fragment reads first, the next tap's weight DMAs behind them, then the MFMAs (K-step order: gemm_bf16.hip)
    bf16x8 xf[2][2], wf[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wm0 + j * 16 + fr + tap;
        xf[kk][j] = *(const bf16x8*)(Xs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn0 + i * 16 + fr;
        wf[kk][i] = *(const bf16x8*)(Ws + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
    }
    if (tap + NST - 1 < TAPS) stage_w((tap + NST - 1) % NST, tap + NST - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
  }

  if constexpr (TRC) ts[5] = __builtin_amdgcn_s_memtime();
  // the prefetched operands landed long ago; touching them here retires the compiler's pending-load bookkeeping in one
  // place, so it does not put a vmcnt(0) -- which would also wait for the previous STORE -- in front of every store
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    asm volatile("" : "+v"(bv[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(rv[i][j]));
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
      f32x4 v = acc[i][j] + bv[i];
      v[0] = mish_f(v[0]); v[1] = mish_f(v[1]); v[2] = mish_f(v[2]); v[3] = mish_f(v[3]);
      if (a.mode == 0) {
        *(bf16x4*)(a.out_bf16 + m * a.ldo + n) = f2bf4(v[0], v[1], v[2], v[3]);
      } else {
        v += rv[i][j];
        *(f32x4*)(a.out_f32 + m * a.ldo32 + n) = v;
      }
    }
  }
  if constexpr (TRC) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = __builtin_amdgcn_s_memtime(); ts[7] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && a.trace) {
      unsigned long long* o = a.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
  }
}

// Split-tap variant: the four waves each own the whole 64 x 64 output tile for every fourth tap and the partial sums are
// added through LDS at the end.  Against the output-split kernel above this halves the ds_read traffic per tap (16
// fragment reads feed 32 MFMAs instead of 8 feeding 8), and the tap loop has no workgroup barrier: every wave streams
// the weight tiles of its own taps through a private 4-stage LDS-DMA ring and is the only reader of it.
//   LDS: activation tile 96 x 128 B | 4 waves x 4 stages x 8 KB of weights; the partial sums 4 x [64][64 + 4] floats
//   reuse the ring space after the loop.
template <bool TRC = false>
__global__ __launch_bounds__(256) void convpos_split_kernel(ConvPosArgs a) {
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (TRC) { ts[0] = __builtin_amdgcn_s_memrealtime(); ts[1] = __builtin_amdgcn_s_memtime(); }
  constexpr int BT = 64, HALO = 15, XR = BT + 2 * HALO;
  constexpr int X_BYTES = 96 * 128, W_BYTES = 8192, NST = 4;
  constexpr int TAPS = 31, RS = 68;  // floats per row of a partial-sum tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;
  char* ring_base = smem + X_BYTES;
  float* Red = (float*)ring_base;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, tt = blockIdx.y, seq = blockIdx.z;
  const int c0 = g * a.cpg;
  const int t0 = tt * BT;
  const int fr = lane & 15, fq = lane >> 4;

  // epilogue operands first (oldest in the in-order vmcnt queue, so the counted waits below are unaffected): the
  // residual rows of mode 1, from clamped addresses
  const int oc = (lane & 15) * 4;
  const int n = c0 + (oc < a.cpg ? oc : 0);
  f32x4 rv[4];
  if (a.mode == 1) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int t = t0 + wave * 16 + p * 4 + (lane >> 4);
      rv[p] = *(const f32x4*)(a.resid + ((size_t)seq * a.N + (t < a.N ? t : a.N - 1)) * a.ldr + n);
    }
  }
  const f32x4 bias = *(const f32x4*)(a.bias + n);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 3; ++r) {  // activation tile: 3 LDS-DMAs per thread, zeros for rows outside [0, N)
    const int i = tid + 256 * r;
    const int row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
    const int t = t0 - HALO + row;
    const void* src = &conv_zero16;
    if (row < XR && t >= 0 && t < a.N && c0 + c * 8 < a.D) src = a.x + ((size_t)seq * a.N + t) * a.ldx + c0 + c * 8;
    glds16c(src, Xs + (wave * 64 + 256 * r) * 16);
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // issue order matters for the counted waits: activation DMAs before the weights

  // wave-private weight ring: tile of tap = 8 KB = 8 DMAs of 1 KB; chunk i = lane + 64 j -> (row = i >> 3, c = i & 7)
  char* ring = ring_base + wave * (NST * W_BYTES);
  const bf16* wg = a.w + (size_t)g * TAPS * 4096;
  const bf16* w_src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = lane + 64 * j;
    const int row = i >> 3, c = (i & 7) ^ ((row >> 1) & 7);
    w_src[j] = wg + row * 64 + c * 8;
  }
  auto stage_w = [&](int slot, int u) {
    int tap = wave + 4 * u;
    tap = tap < TAPS ? tap : TAPS - 1;  // wave 3 has 7 taps: its eighth tile is fetched (counts stay equal) and not used
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16c(w_src[j] + (size_t)tap * 4096, ring + slot * W_BYTES + j * 1024);
  };
#pragma unroll
  for (int u = 0; u < NST - 1; ++u) stage_w(u, u);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int u = 0; u < 8; ++u) {
    // tiles 0 .. min(8, u + 3) - 1 of this wave are issued, tile u must have landed; lgkmcnt(0): this wave's reads of
    // the slot that is re-staged below are done
    const int issued = u + NST - 1 < 8 ? u + NST - 1 : 8;
    const int pend = 8 * (issued - u - 1);
    if (pend == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (pend == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (u == 0) {
      __builtin_amdgcn_s_barrier();  // the activation tile is staged by all four waves
      if constexpr (TRC) ts[2] = __builtin_amdgcn_s_memtime();
    }
    if constexpr (TRC) { if (u == 1) ts[3] = __builtin_amdgcn_s_memtime(); if (u == 4) ts[4] = __builtin_amdgcn_s_memtime(); }
    const int tap = wave + 4 * u;
    const bool live = !(u == 7 && tap >= TAPS);      // wave-uniform: the last round has taps for waves 0-2 only
    const char* Ws = ring + (u % NST) * W_BYTES;
    bf16x8 xf[4], wf[4];
    auto reads = [&](int kk) {
      const int c = kk * 4 + fq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 16 + fr + tap;
        xf[j] = *(const bf16x8*)(Xs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 16 + fr;
        wf[i] = *(const bf16x8*)(Ws + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      }
    };
    auto mfmas = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };
    // the first half's fragment reads go out before the weight DMAs of round u + NST - 1 (K-step order: gemm_bf16.hip)
    if (live) reads(0);
    if (u + NST - 1 < 8) stage_w((u + NST - 1) % NST, u + NST - 1);
    if (!live) break;
    mfmas();
    reads(1);
    mfmas();
  }

  if constexpr (TRC) ts[5] = __builtin_amdgcn_s_memtime();
  // partial sums -> LDS (over the rings: every wave is past its last fragment read after this barrier)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* mine = Red + wave * (64 * RS);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(mine + (j * 16 + fr) * RS + i * 16 + fq * 4) = acc[i][j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(rv[p]));  // see convpos_kernel: one wait here, none between the stores

  if (oc < a.cpg) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = wave * 16 + p * 4 + (lane >> 4);
      const int t = t0 + row;
      const size_t m = (size_t)seq * a.N + t;
      const float* rp = Red + row * RS + oc;
      f32x4 v = (*(const f32x4*)rp + *(const f32x4*)(rp + 64 * RS)) + (*(const f32x4*)(rp + 2 * 64 * RS) + *(const f32x4*)(rp + 3 * 64 * RS));
      v += bias;
      v[0] = mish_f(v[0]); v[1] = mish_f(v[1]); v[2] = mish_f(v[2]); v[3] = mish_f(v[3]);
      if (t >= a.N) continue;
      if (a.mode == 0) {
        *(bf16x4*)(a.out_bf16 + m * a.ldo + n) = f2bf4(v[0], v[1], v[2], v[3]);
      } else {
        v += rv[p];
        *(f32x4*)(a.out_f32 + m * a.ldo32 + n) = v;
      }
    }
  }
  if constexpr (TRC) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = __builtin_amdgcn_s_memtime(); ts[7] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && a.trace) {
      unsigned long long* o = a.trace + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = ts[i];
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

// Diagnostics hook of the TOOLS build only (make tools-lib -> libf5e_hip_tools.so, -DF5E_TOOLS; tools/convpos_time.py):
// the shipped library has no process-wide mutable state and does not export it.
#ifdef F5E_TOOLS
static unsigned long long* g_convpos_trace = nullptr;
#define F5E_CONVPOS_TRACE g_convpos_trace
#else
#define F5E_CONVPOS_TRACE ((unsigned long long*)nullptr)
#endif

extern "C" {

#ifdef F5E_TOOLS
// 8 timestamps per workgroup of the next f5e_convpos launches: [groups * tiles * S][8] uint64 (s_memrealtime at entry,
// s_memtime at entry, first tile landed, two marks inside the tap loop, loop done, stores acknowledged, s_memrealtime at
// exit) while buf != NULL.  Process-wide, not for concurrent callers.
void f5e_debug_convpos_trace(void* buf) { g_convpos_trace = (unsigned long long*)buf; }
#endif

static int convpos_launch(hipStream_t st, const void* x, int ldx, const void* w_packed, const float* bias, int mode,
                          void* out_bf16, int ldo, float* out_f32, int ldo32, const float* resid, int ldr, int S, int N,
                          int D, int groups) {
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
  F5E_REQUIRE(a.tiles_t <= 65535 && S <= 65535, "convpos: grid too large");
  const dim3 grid(groups, a.tiles_t, S);
  const int xs = 96 * 128;
  // Two kernels, same arithmetic up to the order of the fp32 partial sums: grids that fit the chip in one round (one
  // workgroup per CU, e.g. batch 1 with CFG: 256 workgroups) are bound by weight delivery and LDS reads -> split-tap
  // kernel (140 KB LDS); larger grids take the output-split ring (44 KB LDS, 3 workgroups per CU overlap each other).
#ifdef F5E_TOOLS
  static const int forced = getenv("F5E_CONVPOS") ? atoi(getenv("F5E_CONVPOS")) : 0;  // tools build: 1 split, 2 ring
#else
  constexpr int forced = 0;
#endif
  const int n_cu = f5e_cu_count();
  const bool split = forced ? forced == 1 : (size_t)groups * a.tiles_t * S <= (size_t)n_cu;
  const int lds_split = xs + 4 * 4 * 8192, lds_ring = xs + 4 * 8192;
  a.trace = F5E_CONVPOS_TRACE;
  static F5eDeviceOnce once_split;
  if (split) F5E_OPT_IN_LDS(once_split, convpos_split_kernel<false>, lds_split);
#ifdef F5E_TOOLS
  static F5eDeviceOnce once_split_trace;
  if (split && a.trace) {
    F5E_OPT_IN_LDS(once_split_trace, convpos_split_kernel<true>, lds_split);
    hipLaunchKernelGGL(convpos_split_kernel<true>, grid, dim3(256), lds_split, st, a);
  } else if (a.trace) {
    hipLaunchKernelGGL((convpos_kernel<4, true>), grid, dim3(256), lds_ring, st, a);
  } else
#endif
  if (split) {
    hipLaunchKernelGGL(convpos_split_kernel<false>, grid, dim3(256), lds_split, st, a);
  } else {
    hipLaunchKernelGGL((convpos_kernel<4, false>), grid, dim3(256), lds_ring, st, a);
  }
  F5E_LAUNCH_CHECK("convpos");
  return F5E_OK;
}

int f5e_convpos(hipStream_t st, const void* x, int ldx, const void* w_packed, const float* bias, int mode,
                void* out_bf16, int ldo, float* out_f32, int ldo32, const float* resid, int ldr, int S, int N, int D,
                int groups) {
  return convpos_launch(st, x, ldx, w_packed, bias, mode, out_bf16, ldo, out_f32, ldo32, resid, ldr, S, N, D, groups);
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
