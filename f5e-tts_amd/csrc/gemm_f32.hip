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
  int vec;  // epilogue may move 4 columns at a time (alignment checked on the host)
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

// ---- LDS-DMA variant for every call without an A-side activation (input projection, TextEmbedding / Vocos convs, head) ----
// Same ring as gemm_bf16.hip: global_load_lds_dwordx4 into an NSTAGE-deep LDS ring, counted vmcnt + one raw
// s_barrier per K-step of 64 floats (256-byte rows, 16 chunks of 16 B; chunk ^ (row & 15) on the source address and
// on the ds_read_b128 -> each 16-lane group covers all 64 banks once).  A lane (fr, fq) reads 4 consecutive k of its
// row and feeds them to 4 MFMAs; both operands use the same k permutation, so every k is visited exactly once.
// The simple kernel above takes ~1.4 us per 16-wide K step (exposed global latency + 2 barriers): 45 us for
// 281 x 1536 x 512; this one ~8 us.
template <int BM, int BN, int WGM, int WGN, int NSTAGE>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(Gemm32Args a) {
  constexpr int BK = 64, NT = 256, ROWB = BK * 4;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int A_IT = BM * 16 / NT, W_IT = BN * 16 / NT, LPT = A_IT + W_IT;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1 && A_IT >= 1 && W_IT >= 1, "tile / wave layout");
  static_assert((NSTAGE - 2) * LPT <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_n = blockIdx.x / a.tiles_m, tile_m = blockIdx.x - tile_n * a.tiles_m;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const float* a_src[A_IT];
  const float* w_src[W_IT];
  int a_k[A_IT], w_k[W_IT];  // first k of the thread's chunk inside a K-tile (for the ragged last tile)
#pragma unroll
  for (int j = 0; j < A_IT; ++j) {
    const int i = tid + NT * j, row = i >> 4, c = (i & 15) ^ (row & 15);
    a_src[j] = a.A + (size_t)(min(m0 + row, a.M - 1) % a.a_rows) * a.lda + c * 4;
    a_k[j] = c * 4;
  }
#pragma unroll
  for (int j = 0; j < W_IT; ++j) {
    const int i = tid + NT * j, row = i >> 4, c = (i & 15) ^ (row & 15);
    w_src[j] = a.W + (size_t)min(n0 + row, a.N - 1) * a.ldw + c * 4;
    w_k[j] = c * 4;
  }
  auto dma = [](const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  };
  // K % 64 != 0 (K % 4 == 0 always): chunks of the last K-tile that start at k >= K are fetched from the row start
  // instead (any valid address: no out-of-bounds read) and overwritten with zeros by their owner once they have landed
  auto stage = [&](int buf, int kt) {
    char* base = smem32 + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_IT; ++j)
      dma(kt * BK + a_k[j] < a.K ? a_src[j] + kt * BK : a_src[j] - a_k[j], base + (wave * 64 + NT * j) * 16);
#pragma unroll
    for (int j = 0; j < W_IT; ++j)
      dma(kt * BK + w_k[j] < a.K ? w_src[j] + kt * BK : w_src[j] - w_k[j], base + A_BYTES + (wave * 64 + NT * j) * 16);
  };

  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = (a.K + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < KT) stage(s, s);
  int buf = 0, nbuf = NSTAGE - 1;
  for (int kt = 0; kt < KT; ++kt) {
    const int rem = KT - 1 - kt;
    switch (rem < NSTAGE - 2 ? rem : NSTAGE - 2) {  // tile kt landed; that many younger stages may stay in flight
      case 0: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(0) : "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPT) : "memory"); break;
    }
    if (rem == 0 && (a.K & (BK - 1))) {  // ragged tail: own chunks past K become zeros (own DMAs have landed: vmcnt(0))
      char* base = smem32 + buf * STAGE;
#pragma unroll
      for (int j = 0; j < A_IT; ++j)
        if (kt * BK + a_k[j] >= a.K) *(f32x4*)(base + (tid + NT * j) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < W_IT; ++j)
        if (kt * BK + w_k[j] >= a.K) *(f32x4*)(base + A_BYTES + (tid + NT * j) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* As = smem32 + buf * STAGE;
    const char* Ws = As + A_BYTES;
    f32x4 xf[TM], wf[TN];
    auto reads = [&](int g) {
      const int c = g * 4 + fq;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm0 + j * 16 + fr;
        xf[j] = *(const f32x4*)(As + row * ROWB + ((c ^ (row & 15)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn0 + i * 16 + fr;
        wf[i] = *(const f32x4*)(Ws + row * ROWB + ((c ^ (row & 15)) << 4));
      }
    };
    // the first fragment reads go out BEFORE the next tile's LDS-DMAs: a DMA costs the wave 100+ cycles of issue while the
    // CU's address path is busy, and reads issued behind it (the order of rounds 1-2) wait that long (gemm_bf16.hip)
    reads(0);
    if (kt + NSTAGE - 1 < KT) stage(nbuf, kt + NSTAGE - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g > 0) reads(g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][r], xf[j][r], acc[i][j], 0, 0, 0);
    }
    buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
  }

  if (a.vec) {
    // N, leading dimensions and pointers are 16-byte friendly (host check): quads of 4 columns move as one piece and
    // the addend rows are requested together, ahead of the arithmetic (one round trip instead of sixteen)
    f32x4 add[TN][TM];
    if (a.addend) {
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int m = min(m0 + wm0 + j * 16 + fr, a.M - 1);
        const float* ap = a.addend + (size_t)(m % a.add_rows) * a.ld_add;
#pragma unroll
        for (int i = 0; i < TN; ++i) add[i][j] = *(const f32x4*)(ap + min(n0 + wn0 + i * 16 + fq * 4, a.N - 4));
      }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm0 + j * 16 + fr;
      if (m >= a.M) continue;
      const float rs = a.row_scale ? a.row_scale[m] : 1.0f;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn0 + i * 16 + fq * 4;
        if (n >= a.N) continue;
        f32x4 v = acc[i][j];
        if (a.bias) v += *(const f32x4*)(a.bias + n);
        if (a.act != F5E_ACT_NONE) { v[0] = apply_act(v[0], a.act); v[1] = apply_act(v[1], a.act); v[2] = apply_act(v[2], a.act); v[3] = apply_act(v[3], a.act); }
        if (a.ch_scale) v *= *(const f32x4*)(a.ch_scale + n);
        if (a.addend) v += add[i][j];
        v *= rs;
        if (a.out) *(f32x4*)(a.out + (size_t)m * a.ldo + n) = v;
        if (a.out_bf16) *(bf16x4*)(a.out_bf16 + (size_t)m * a.ldo_bf16 + n) = f2bf4(v[0], v[1], v[2], v[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm0 + j * 16 + fr;
    if (m >= a.M) continue;
    const float rs = a.row_scale ? a.row_scale[m] : 1.0f;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
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

template <int BM, int BN, int WGM, int WGN, int NSTAGE>
int launch_dma(Gemm32Args& a, hipStream_t st) {
  a.tiles_m = (a.M + BM - 1) / BM;
  const int grid = a.tiles_m * ((a.N + BN - 1) / BN);
  constexpr int lds_max = NSTAGE * (BM + BN) * 256;
  static F5eDeviceOnce lds_once;  // > 64 KiB of dynamic LDS needs the opt-in attribute, per device (host-only call)
  if (lds_max > 65536) F5E_OPT_IN_LDS(lds_once, (gemm_f32_dma_kernel<BM, BN, WGM, WGN, NSTAGE>), lds_max);
  // a K of fewer tiles than ring slots never re-stages: slots >= KT are untouched, so they need not exist.  The per-step
  // input projection (K = 100: 2 tiles) then takes 64 instead of 96 KiB and two workgroups share a CU.
  const int kt = (a.K + 63) / 64;
  const int lds = (kt < NSTAGE ? kt : NSTAGE) * (BM + BN) * 256;
  hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, WGM, WGN, NSTAGE>), dim3(grid), dim3(256), lds, st, a);
  F5E_LAUNCH_CHECK("gemm_f32_dma");
  return F5E_OK;
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
  a.vec = N % 4 == 0 && N >= 4 && ldo % 4 == 0 && ldo_bf16 % 4 == 0 && ld_add % 4 == 0 &&
          (((uintptr_t)bias | (uintptr_t)ch_scale | (uintptr_t)addend | (uintptr_t)out) & 15) == 0 && ((uintptr_t)out_bf16 & 7) == 0;
  if (a_act == F5E_ACT_NONE && (((uintptr_t)A | (uintptr_t)W) & 15) == 0) {  // K % 4 == 0 is checked above
    // few tiles: halve BM so more CUs take part (these GEMMs have M of a few hundred rows)
    if (((M + 63) / 64) * ((N + 63) / 64) < 128) return launch_dma<32, 64, 2, 2, 4>(a, st);
    return launch_dma<64, 64, 2, 2, 3>(a, st);
  }
  a.tiles_m = (M + 63) / 64;
  const int grid = a.tiles_m * ((N + 63) / 64);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(grid), dim3(256), 0, st, a);
  F5E_LAUNCH_CHECK("gemm_f32");
  return F5E_OK;
}
