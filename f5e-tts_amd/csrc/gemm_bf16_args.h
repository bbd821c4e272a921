// Argument block, epilogue ids and the LDS-DMA helper shared by the bf16 GEMM kernels (gemm_bf16.hip: 64x64 .. 128x128
// ring kernels; gemm_bf16_pp.hip: the 256x256 ping-pong kernel for large M).  Internal to the library (not ABI).
#pragma once
#include <cstddef>

#include "f5e_common.h"

namespace f5e_gemm {

struct GemmArgs {
  // First 64 bytes = everything a workgroup needs before it can issue its first LDS-DMA: ONE scalar-cache line, fetched by
  // the first s_loads of the kernel.  (Measured with tools/gemm_trace.hip: 2500 cycles from kernel entry to the first
  // DMA when these fields were spread over three kernarg lines with two integer divisions in between; the kernarg
  // segment is cold in every launch.)
  const bf16* A; const bf16* W;
  int lda, ldw;
  int M, N, K;
  int tiles_m, tiles_n, m_major;   // m_major: 1 = row-major tile walk; <= 0: n-major in column groups of 2^-m_major n-tiles
  unsigned tile_magic;   // floor(2^32 / (m_major > 0 ? tiles_n : tiles_m << -m_major)): tile decode without a division (div_magic)
  int rows_per_seq;
  unsigned rps_magic;    // floor(2^32 / rows_per_seq)
  int n_main;            // tiles_m * tiles_n: workgroups beyond it only prefetch (F5ePrefetch), 64x64 .. 128x128 kernels
  // ---- epilogue operands ----
  const float* bias;
  void* out; int ldo;
  // gate + residual epilogue
  float* resid; int ldr;
  const float* gate; int gate_stride; int gate_rows;
  const int* eval_ptr; int eval_stride;
  const int* seq_len;
  // qkv + rope epilogue
  bf16* q; bf16* k; bf16* vt;
  int n_pad; int heads; int rope_heads;
  const float* cos_sin;  // [rows_per_seq][32][2]
  const float* qn_w; const float* kn_w; float qk_eps;  // optional RMSNorm over the head dim of q / k (qk_norm)
  float q_scale;         // factor folded into q after RoPE, before the bf16 rounding: log2(e) / sqrt(64) (attention.hip)
  unsigned long long* trace;  // diagnostics builds only (tools/gemm_trace.hip, tools/pp_timeline.py): per-workgroup timeline
  // Fused AdaLN (FUSE != 0, 64x64 tiles, small M): the LayerNorm+modulate launches between the GEMMs disappear.
  //   consumer (FUSE 1): A = xs = bf16((x - o) (1 + scale[k])) for a per-row offset o, and with c[n] = sum_k W[n][k] (1 + scale[k]),
  //     d[n] = sum_k W[n][k] shift[k] + bias[n]:   LN(x)(1+scale)+shift  @ W^T + bias = rstd (acc - (mean - o) c[n]) + d[n]
  //   producer (FUSE 2, gate+residual epilogue): next to x_new it stores xs for the NEXT consumer and, per row and
  //     64-column tile, (mean, M2) of x_new; the consumer combines the tiles with Chan's formula in a fixed order.
  const float* ln_stats; int ln_parts;
  const float* ln_c; const float* ln_d; int cd_stride, cd_rows, cd_eval_stride; float ln_eps;
  bf16* xs_out; int ld_xs; const float* next_scale; float* stats_out;
  // Centring (round 3): xs = bf16((x - row_mean[m]) (1 + scale)) and the tile means are stored RELATIVE to row_mean[m], the
  // row's mean as of the previous LayerNorm -- row means drift slowly from norm to norm, so what is rounded to bf16 is of the
  // size of the row's spread, not of its offset.  The consumer applies  rstd (acc - mean' c[n]) + d[n]  with the relative
  // mean' it finds in the statistics and (column tile 0 only) moves row_mean[m] += mean' for the producer behind it.
  float* row_mean;
  int pp_ngroup;   // gemm_bf16_pp.hip: n-tiles per column group of the tile order (0 = plain m-major)
  F5ePrefetch pf;  // weights of the next kernels, pulled into the Infinity Cache by grid-tail workgroups (small M only)
  unsigned pf_per_wg;  // 0: 32 KiB per prefetch workgroup; else the packed granule (bytes) of f5e_prefetch_run_packed
};

static_assert(offsetof(GemmArgs, n_main) + sizeof(int) == 64, "hot kernarg fields must fill exactly the first 64-byte line");

// floor(n / d) for 0 <= n < 2^31, d >= 1, magic = min(floor(2^32 / d), 2^32 - 1): one mul_hi + one correction step
// (s_mul_hi_u32 on uniform operands) instead of the ~40-instruction reciprocal sequence hipcc emits for `/`.
__host__ __device__ inline unsigned div_magic_of(int d) {
  const unsigned long long m = 0x100000000ull / (unsigned long long)(d > 0 ? d : 1);
  return m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)m;
}
__device__ __forceinline__ int div_magic(int n, int d, unsigned magic) {
  unsigned q = __umulhi((unsigned)n, magic);
  if ((unsigned)n - q * (unsigned)d >= (unsigned)d) ++q;
  return (int)q;
}

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_GATE_RES = 2, EPI_QKV_ROPE = 3, EPI_F32 = 4 };

// LDS-DMA with the `saddr` address form: 64-bit wave-uniform base in scalar registers + 32-bit byte offset per lane.  Written
// as inline assembly because the compiler folds (uniform base + lane offset) into per-lane 64-bit pointers again (one
// v_lshl_add_u64 per DMA per K-tile).  M0 = LDS byte address of the wave's first lane (the DMA adds lane * 16); one wait state
// between the write of M0 and the DMA.  Counted vmcnt waits see these like the builtin's loads (tools/check_vmcnt.py reads the ISA).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // "m0 is reserved": the clobber is what we mean -- the DMA builtins set M0 themselves
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, void* lds) {
  const unsigned l = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}


// gemm_bf16_pp.hip: 256x256 tile, 8 waves in two staggered groups (defined there; epi = EPI_* id)
int launch_pp(int epi, GemmArgs& a, hipStream_t st);
// the rule both the dispatcher and f5e_dit_forward use: launches of this many rows take the 256x256 kernel
inline bool uses_pp(int M, int K) { return K >= 128 && (M + 255) / 256 >= 44; }

}  // namespace f5e_gemm
