// bf16 MFMA GEMM with fused epilogues for the DiT block linears (K8/K11/K12/K13 of SURVEY 2.3).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]      A: activations [M][K] bf16, W: nn.Linear weight [N][K] bf16
//
// Replaces the F.linear calls at reference model/modules.py:452-454 (to_q/k/v), :495 (to_out),
// :349-350 (FeedForward) and backbones/dit.py:470 (proj_out), each fused with what follows it.
//
// gfx950 design notes
//   * 256 threads = 4 waves (2x2); block tile BM x BN, K-step 64; v_mfma_f32_16x16x32_bf16.
//   * operands are SWAPPED (W rows feed the MFMA "A" port, activation rows the "B" port) so an
//     accumulator register quad holds 4 CONSECUTIVE output columns n for one row m: bias / gate /
//     RoPE pairs are lane-local, bf16 stores are 8 B and fp32 residual updates 16 B per lane.
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4), 2 LDS stages, one barrier per K-step.
//     The LDS image is lane-linear; the bank swizzle (16-B chunk ^ ((row>>1)&7)) is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 (cdna guide 5.4 rule 21) -> the
//     16-lane ds_read_b128 groups are conflict free.
#include <cstdlib>
#include <type_traits>

#include "f5e_common.h"
#include "gemm_bf16_args.h"

namespace {

using namespace f5e_gemm;

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  // all but the N youngest vector-memory ops (LDS-DMA stages) done, and this wave's LDS reads retired
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int LPT, int EXTRA>
__device__ __forceinline__ void wait_stages(int nst) {
  // cases a kernel can reach satisfy nst * LPT + EXTRA <= 63 (static_assert at the call site); the clamp only keeps the
  // unreachable instantiations assemblable (vmcnt is a 6-bit field)
  constexpr auto C = [](int v) constexpr { return v > 63 ? 63 : v; };
  switch (nst) {
    case 0: wait_vm_lgkm0<C(EXTRA)>(); break;
    case 1: wait_vm_lgkm0<C(LPT + EXTRA)>(); break;
    case 2: wait_vm_lgkm0<C(2 * LPT + EXTRA)>(); break;
    case 3: wait_vm_lgkm0<C(3 * LPT + EXTRA)>(); break;
    case 4: wait_vm_lgkm0<C(4 * LPT + EXTRA)>(); break;
    case 5: wait_vm_lgkm0<C(5 * LPT + EXTRA)>(); break;
    default: wait_vm_lgkm0<C(6 * LPT + EXTRA)>(); break;
  }
}

// K-step 64: 128-byte LDS rows, two rows per 256-byte bank line, swizzle (row >> 1) & 7.  (Measured and dropped, DESIGN 4:
// K-step 128, 5-8 stage rings, 8 waves on the 64x64 tile, intra-workgroup split-K for the one-workgroup-per-CU launches.)
// DBG >= 3: in-kernel timestamps for tools/gemm_trace.hip (never instantiated by the library).
// NLOAD > 0: ROLE SPLIT.  The workgroup gets NLOAD extra waves that do nothing but stage (every global_load_lds of the ring +
// the counted vmcnt waits); the WGM x WGN consumer waves only ds_read and MFMA.  Why: the CU's texture-address path takes
// ~18 cycles per 1 KiB LDS-DMA instruction (tools/l2_probe.hip: 57 B/clk per CU from L2, whatever the number of waves), so a
// 64 x 64 K-step carries 288 cycles of DMA issue against 128 of MFMA -- and with every wave doing both, behind one barrier,
// the two ADD (553 cycles per step measured).  Split, the loader's issue runs beside the consumers' reads and MFMAs.
// DSTEP (role split only): K-tiles per hand-over barrier.  A barrier is a full rendezvous of loaders and consumers and costs
// ~100 cycles of skew per K-step on top of max(loader issue, consumer reads + MFMAs); handing over TWO tiles per barrier
// halves that (K % 128 == 0; ring of NSTAGE >= 3 DSTEP tiles: DSTEP being read, DSTEP landed or landing, DSTEP being issued).
// Role split: the consumers request the epilogue operands (x tile, gate, statistics: ~14 loads per lane, ~1000 cycles of the
// address path per workgroup, a cold line's latency from the Infinity Cache or HBM) behind the FIRST hand-over.  Measured
// alternatives (DESIGN 4 "Round 4"): at kernel entry they delay the first tile by ~640 cycles; behind an extra barrier that
// follows the loaders' prologue issue: 41.9 vs 40.9 ms per C2 pass; over the LAST K-tiles, where the loaders fall silent:
// fastest stand-alone, 42.3 ms in situ (the x tile is not back when the epilogue wants it).
// DBUF (role split only): 1 = the consumers double-buffer their fragments in registers (reads of tile kt under the MFMAs of
// tile kt - 1); 0 = ONE fragment set whose two 32-k halves take turns (reads of one half under the MFMAs of the other) for the
// consumers of 64 x 32 wave tiles, whose VGPR budget (128 at 16 waves, 168 at 12) does not hold two full sets.
// 128-row tiles (round 4, M in (1024, 2048]: C4's utterance lengths): the same role-split kernels on 128 x 64 (producer, 8
// consumer waves of 32 x 32), 128 x 128 and 128 x 192 (consumers, 8 / 12 waves of 64 x 32) -- again a ONE-round grid of at
// most 16 x 16 tiles, where the 64-row tiles would need two rounds of classic three-per-CU workgroups.
template <int BM, int BN, int EPI, int NSTAGE, int DBG = 0, int WGM = 2, int WGN = 2, int FUSE = 0, int NLOAD = 0, int DSTEP = 1, int DBUF = 1>
__global__ __launch_bounds__(64 * (WGM * WGN + NLOAD)) void gemm_bf16_kernel(GemmArgs a) {
  constexpr int BK = 64;
  constexpr int CPR = BK / 8;                    // 16-byte chunks per LDS row
  constexpr int ROWB = BK * 2;                   // bytes per LDS row
  auto swz = [](int row) { return (row >> 1) & 7; };
  constexpr int NCW = WGM * WGN;      // WGM x WGN consumer waves, each owning a (BM/WGM) x (BN/WGN) sub-tile
  constexpr int NC = 64 * NCW;
  constexpr int NT = NLOAD ? 64 * NLOAD : NC;   // threads that stage: the loader waves of a role-split launch, else everyone
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = (int)threadIdx.x;
  const bool is_loader = NLOAD > 0 && wave >= NCW;       // wave-uniform
  const int stid = NLOAD ? tid - NC : tid;               // index among the staging threads (loaders: 0 .. NT - 1)
  const int swave = NLOAD ? wave - NCW : wave;
  unsigned long long* trc = nullptr;
  if constexpr (DBG >= 3) {
    trc = a.trace + (size_t)blockIdx.x * 48;
    if (tid == 0) { trc[0] = __builtin_amdgcn_s_memrealtime(); trc[1] = __builtin_amdgcn_s_memtime(); }
  }

  // Kernarg lines beyond the first: request them NOW, next to the first line, so their (cold) scalar-cache misses overlap
  // instead of surfacing one by one where the compiler first needs a field (the prefetch block behind the prologue).
  if (FUSE == 1)
    asm volatile("" ::"s"(a.ln_stats), "s"(a.ln_c), "s"(a.ln_d), "s"(a.ln_parts), "s"(a.cd_stride), "s"(a.cd_rows),
                 "s"(a.cd_eval_stride), "s"(a.eval_ptr));
  if (EPI == EPI_GATE_RES)
    asm volatile("" ::"s"(a.resid), "s"(a.gate), "s"(a.seq_len), "s"(a.eval_ptr), "s"(a.ldr), "s"(a.gate_stride),
                 "s"(a.gate_rows), "s"(a.eval_stride), "s"(a.bias));
  if (FUSE == 2) asm volatile("" ::"s"(a.next_scale), "s"(a.xs_out), "s"(a.stats_out), "s"(a.ld_xs), "s"(a.row_mean));
  if (EPI == EPI_QKV_ROPE) asm volatile("" ::"s"(a.q), "s"(a.k), "s"(a.vt), "s"(a.cos_sin), "s"(a.n_pad), "s"(a.heads), "s"(a.rope_heads));

  if ((int)blockIdx.x >= a.n_main) {   // grid-tail workgroups: Infinity-Cache prefetch only (f5e_common.h), 256 threads of them
    if (NLOAD > 0 && a.pf_per_wg) f5e_prefetch_run_packed(a.pf, (int)blockIdx.x - a.n_main, tid, 64 * (NCW + NLOAD), a.pf_per_wg, smem);
    else if (tid < 256) f5e_prefetch_run(a.pf, (int)blockIdx.x - a.n_main, tid, smem);
    return;
  }
  // XCD-aware tile order: blocks that share blockIdx%8 (one XCD) walk tiles with the same n-panel.
  int bid = blockIdx.x;
  {
    const int nblk = a.n_main;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  // consecutive logical ids (= one XCD's L2) share the LARGER operand's panel: the weight panel when N*K >= M*K
  // (small batch), the activation panel when M > N (large batch: otherwise A is re-fetched once per n-panel)
  // (small batch, m_major <= 0): COLUMN GROUPS of G = 2^-m_major n-tiles -- all row tiles of n-tiles [g G, (g + 1) G), n
  // fastest, then the next group -- so that the contiguous run of logical ids one XCD owns is a (rows x G columns) block of
  // tiles instead of whole n-panels: its private L2 then pulls (rows + G) 64-row panels through the fabric instead of
  // (tiles_m + n-panels).  G is picked on the host (pick_group_shift); G = 1 is the plain n-major walk.
  int tile_m, tile_n;
  if (a.m_major > 0) {
    tile_m = div_magic(bid, a.tiles_n, a.tile_magic);
    tile_n = bid - tile_m * a.tiles_n;
  } else {
    const int sh = -a.m_major, per = a.tiles_m << sh;
    const int g = div_magic(bid, per, a.tile_magic);
    const int r = bid - g * per;
    tile_m = r >> sh;
    tile_n = (g << sh) + (r & ((1 << sh) - 1));
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging: chunk i (16 B) of a [rows][BK] bf16 tile sits at LDS byte i*16 ----
  // source chunk for LDS position (row, c) is c ^ swz(row)
  constexpr int A_IT = BM * CPR / NT;
  constexpr int W_IT = BN * CPR / NT;
  static_assert(A_IT >= 1 && W_IT >= 1 && A_IT * NT == BM * CPR && W_IT * NT == BN * CPR, "tile / thread-count mismatch");
  // Source addresses as a wave-UNIFORM 64-bit base (the tile's first row, advanced by the K-tile: scalar registers, scalar
  // adds) + a loop-invariant 32-bit byte offset per lane: the LDS-DMAs take the `saddr` form (glds16_s) and the staging of a
  // K-tile needs no vector ALU instruction -- per-lane 64-bit pointers cost one v_lshl_add_u64 per DMA per K-tile on a SIMD
  // whose vector issue the consumers' MFMAs hold half of the time.  Worth 0.5 % of a C2 pass.
  unsigned a_off[A_IT], w_off[W_IT];
#pragma unroll
  for (int j = 0; j < A_IT; ++j) {
    const int i = (stid & (NT - 1)) + NT * j;
    const int row = i / CPR, c = (i % CPR) ^ swz(row);
    a_off[j] = (unsigned)(min(m0 + row, a.M - 1) - m0) * (unsigned)(a.lda * 2) + (unsigned)c * 16u;   // m0 < M: never negative
  }
#pragma unroll
  for (int j = 0; j < W_IT; ++j) {
    const int i = (stid & (NT - 1)) + NT * j;
    const int row = i / CPR, c = (i % CPR) ^ swz(row);
    w_off[j] = (unsigned)(min(n0 + row, a.N - 1) - n0) * (unsigned)(a.ldw * 2) + (unsigned)c * 16u;
  }
  const char* const a_base = (const char*)(a.A + (size_t)m0 * a.lda);   // uniform
  const char* const w_base = (const char*)(a.W + (size_t)n0 * a.ldw);
  const int KT = a.K / BK;
  char* const ring = smem;
  auto stage = [&](int buf, int kt) {
    char* base = ring + buf * STAGE;
    const char* ak = a_base + (size_t)kt * (BK * 2);
    const char* wk = w_base + (size_t)kt * (BK * 2);
#pragma unroll
    for (int j = 0; j < A_IT; ++j) glds16_s(ak, a_off[j], base + (swave * 64 + NT * j) * 16);
#pragma unroll
    for (int j = 0; j < W_IT; ++j) glds16_s(wk, w_off[j], base + A_BYTES + (swave * 64 + NT * j) * 16);
  };

  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int fr = lane & 15, fq = lane >> 4;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // NSTAGE-deep LDS ring fed by LDS-DMA: NSTAGE-1 K-tiles are in flight while one is consumed.  Counted vmcnt +
  // raw s_barrier (a __syncthreads() would drain the DMA queue: cdna guide "Pipelining across barriers").
  constexpr int LPT = A_IT + W_IT;  // LDS-DMA instructions per thread per stage
  static_assert(NSTAGE >= 2 && NSTAGE <= 8 && (DSTEP > 1 || (NSTAGE - 2) * LPT <= 63), "vmcnt is a 6-bit counter");
  static_assert((NT & (NT - 1)) == 0, "staging thread count must be a power of two");
  static_assert(DSTEP == 1 || (NLOAD > 0 && DSTEP == 2 && NSTAGE >= 3 * DSTEP), "DSTEP: role split, ring of >= 3 steps");
  constexpr int NPRO = NSTAGE - DSTEP;      // tiles issued ahead of the first hand-over (DSTEP == 1: the classic NSTAGE - 1)
  static_assert((NSTAGE - 2 * DSTEP) * LPT <= 63, "vmcnt is a 6-bit counter");
  if (NLOAD == 0 || is_loader) {
#pragma unroll
    for (int s = 0; s < NPRO; ++s)
      if (s < KT) stage(s, s);
  }
  if constexpr (NLOAD > 0) {
    if (is_loader) {
      // ---- loader waves: wait for tiles kt .. kt + DSTEP - 1, meet the consumers at the barrier that hands them over (and
      // tells us they are done reading the DSTEP tiles before), refill those buffers with tiles kt + NPRO ...  Same RAW / WAR
      // argument as the classic ring, per step of DSTEP tiles.
      int nbuf_l = NPRO % NSTAGE;
      if constexpr (DBG >= 3) { if (stid == 0) trc[47] = __builtin_amdgcn_s_memtime(); }
      for (int kt = 0; kt < KT; kt += DSTEP) {
        // issued so far: tiles [0, min(KT, kt + NPRO)); all but the ones younger than this step's tiles must have landed:
        // min(tiles behind this step, NSTAGE - 2 DSTEP) may stay in flight
        const int rem = KT - DSTEP - kt;             // tiles behind this step's (K % (64 DSTEP) == 0: host check)
        wait_stages<LPT, 0>(rem < NSTAGE - 2 * DSTEP ? (rem < 0 ? 0 : rem) : NSTAGE - 2 * DSTEP);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int d = 0; d < DSTEP; ++d) {
          if (kt + NPRO + d < KT) stage(nbuf_l, kt + NPRO + d);
          nbuf_l = (nbuf_l + 1 == NSTAGE) ? 0 : nbuf_l + 1;
        }
      }
      if (FUSE == 1) __syncthreads();                 // the consumers' epilogue barriers (every wave of the workgroup)
      if (FUSE == 2) __builtin_amdgcn_s_barrier();
      return;
    }
  }
  // Epilogue operands are fetched right after the first stages were issued and stay in flight under the first
  // NSTAGE - 1 K-steps: vmcnt retires in order, so the waits for tiles 0 .. NSTAGE-2 (older than these loads) allow
  // NPC more outstanding operations.  NPC counts only loads that are certainly issued: unconditional (clamped
  // addresses), one 16-/8-byte vector load each, pinned between two asm memory barriers.  Over-counting would let a
  // tile be read before it landed, so loads that may be skipped (bias, seq_len) are not counted (that only makes a wait
  // conservative) and tools/check_vmcnt.py verifies on the compiled ISA that every instantiation issues >= NPC plain
  // vector loads between the prologue's LDS-DMAs and the first counted wait (run by `make check`).  `volatile` is no
  // way to pin them: hipcc turns volatile loads into flat_load sc0 sc1 + s_waitcnt vmcnt(0) each.
  asm volatile("" ::: "memory");
  if constexpr (DBG >= 3) { if (tid == 0) trc[46] = __builtin_amdgcn_s_memtime(); }
  constexpr bool PREF = (TM * TN <= 4);
  // fused AdaLN: 64-row tiles of 32 x 32 wave sub-tiles (row statistics: 4 threads per row = the first 256 threads); the
  // producer is 64 x 64, a role-split consumer may be wider (64 x 128 / 64 x 192: the A panel is staged once per K-step)
  static_assert(FUSE != 2 || (PREF && WM == 32 && WN == 32 && BN == 64 && (BM == 64 || (BM == 128 && NLOAD > 0))),
                "fused AdaLN producer: 64 (or, role split, 128) x 64 tiles of 32 x 32 wave sub-tiles");
  static_assert(FUSE != 1 || (WN == 32 && (BN == 64 || NLOAD > 0) && ((BM == 64 && WM == 32) || (BM == 128 && WM == 64 && NLOAD > 0))),
                "fused AdaLN consumer: 64-row tiles of 32 x 32 wave sub-tiles, or (role split) 128-row tiles of 64 x 32 ones");
  static_assert(DBUF == 1 || (NLOAD > 0 && DSTEP == 1), "single fragment set: role split, one tile per hand-over");
  constexpr int NSTATT = BM * 4;   // fused consumer: 4 threads per row of the tile work out its statistics
  static_assert(FUSE != 1 || NSTATT <= NC, "fused consumer: 4 statistics threads per row");
  static_assert(FUSE != 2 || EPI == EPI_GATE_RES, "the AdaLN producer is the gate+residual epilogue");
  constexpr int NPC = NLOAD ? 0 : (FUSE == 1 ? 2 * TN + 4 : (FUSE == 2 ? 3 * TM * TN : ((PREF && EPI == EPI_GATE_RES) ? 2 * TM * TN : 0)));
  static_assert((NSTAGE - 2) * LPT + NPC <= 63, "vmcnt is a 6-bit counter");
  // role split: one gate row (host check), so gate and next-norm scale depend on the column only -- TN loads, not TM x TN
  // (the epilogue operands queue on the same address path as the loaders' DMAs: every load less is ~18 cycles per wave)
  constexpr bool COLG = NLOAD > 0;
  f32x4 pf_bias[PREF ? TN : 1], pf_gate[PREF ? TN : 1][PREF ? (COLG ? 1 : TM) : 1], pf_x[PREF ? TN : 1][PREF ? TM : 1];
  f32x4 pf_c[FUSE == 1 ? TN : 1], pf_d[FUSE == 1 ? TN : 1];
  f32x4 pf_ns[FUSE == 2 ? TN : 1][FUSE == 2 ? (COLG ? 1 : TM) : 1];
  f32x2 pf_st[FUSE == 1 ? 4 : 1];
  bool pf_live[PREF ? TM : 1];
  int pf_len[PREF ? TM : 1];
  float pf_rm[FUSE == 2 ? TM : 1];   // producer: the rows' centring offsets (row means as of the previous norm)
  auto ldv4 = [](const float* ptr) -> f32x4 { return *(const f32x4*)ptr; };
  // Classic launch: right here, behind the prologue's stages (see above).  Role split: the consumers issue no DMA, and asking
  // for these at kernel entry would put their ~10 loads per lane on the address path IN FRONT of the loaders' first tiles
  // (measured: first tile landed 640 cycles later) -- they go out late in the K loop instead (see the consumer loop).
  auto fetch_epilogue_operands = [&]() {
    if (PREF) {
      if (EPI == EPI_GATE_RES) {
        const size_t eoff = a.eval_ptr ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int m = m0 + wm0 + j * 16 + fr, mc = min(m, a.M - 1);
          const int seq = div_magic(mc, a.rows_per_seq, a.rps_magic);
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            const int nc = min(n0 + wn0 + i * 16 + fq * 4, a.N - 4);
            const size_t goff = eoff + ((COLG || a.gate_rows == 1) ? 0 : (size_t)(seq % a.gate_rows) * a.gate_stride) + nc;
            if (!COLG || j == 0) pf_gate[i][COLG ? 0 : j] = ldv4(a.gate + goff);                     // counted
            pf_x[i][j] = ldv4(a.resid + (size_t)mc * a.ldr + nc);                                   // counted
            if (FUSE == 2 && (!COLG || j == 0)) pf_ns[i][COLG ? 0 : j] = ldv4(a.next_scale + goff);  // counted
          }
        }
      }
    }
    if (FUSE == 1) {
      // As few vector-memory instructions as possible: the CU's address unit takes ~16 cycles per wave-instruction whatever
      // its width, and all 12 waves of a CU run this block at once (tools/gemm_trace.hip: 16 extra loads per wave cost 2100
      // cycles before the first K-step).  One table row per evaluation (cd_rows == 1, checked on the host): c and d do not
      // depend on the row -> 2 TN loads; statistics: 4 threads per row, ln_parts / 4 <= 4 consecutive (mean, M2) pairs each
      // (the 64 lanes of a wave read one contiguous range of [M][parts][2]); entries past a thread's share repeat its last.
      const size_t eoff = a.eval_ptr ? (size_t)load_uniform_i32(a.eval_ptr) * a.cd_eval_stride : 0;
      const float* cbase = a.ln_c + eoff;
      const float* dbase = a.ln_d + eoff;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const unsigned nc = (unsigned)min(n0 + wn0 + i * 16 + fq * 4, a.N - 4);
        pf_c[i] = ldv4(cbase + nc);                                    // counted
        pf_d[i] = ldv4(dbase + nc);                                    // counted
      }
      const int pp = a.ln_parts >> 2;
      const float* sbase = a.ln_stats + (size_t)m0 * a.ln_parts * 2;          // uniform
      const unsigned rel_max = (unsigned)((a.M - m0) * a.ln_parts - 1);       // last valid pair of the tile's rows
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned rel = min((unsigned)((tid & (NSTATT - 1)) * pp + min(u, pp - 1)), rel_max);
        if (NC == NSTATT || tid < NSTATT) pf_st[u] = *(const f32x2*)(sbase + rel * 2u);   // counted (classic: every thread)
      }
    }
    if (PREF) {  // not counted: may be skipped
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + wn0 + i * 16 + fq * 4;
        pf_bias[i] = (a.bias && n < a.N) ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (EPI == EPI_GATE_RES) {  // only fetched here; compared in the epilogue (a use now would drain vmcnt)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int mc = min(m0 + wm0 + j * 16 + fr, a.M - 1);
          pf_len[j] = a.seq_len ? a.seq_len[div_magic(mc, a.rows_per_seq, a.rps_magic)] : a.rows_per_seq;
          if (FUSE == 2) pf_rm[j] = a.row_mean[mc];
        }
      }
    }
  };
  if (NLOAD == 0) fetch_epilogue_operands();

  int buf = 0, nbuf = NSTAGE - 1;
  if constexpr (DBG >= 3) { if (tid == 0) trc[2] = __builtin_amdgcn_s_memtime(); }
  if constexpr (NLOAD > 0) {
    // ---- consumer waves of a role-split launch.  Step kt: pass the barrier that hands tile kt over, issue its fragment
    // reads into one register set, run the MFMAs of tile kt - 1 from the other set under those reads, and retire the reads
    // (lgkmcnt(0)) BEFORE the next barrier -- behind it the loader refills the buffer just read.
    bf16x8 xs2[DBUF + 1][BK / 32][TM], ws2[DBUF + 1][BK / 32][TN];
    auto reads = [&](auto pc, int b) {
      constexpr int P = decltype(pc)::value;
      const char* As = ring + b * STAGE;
      const char* Ws = As + A_BYTES;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        const int c = kk * 4 + fq;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int row = wm0 + j * 16 + fr;
          xs2[P][kk][j] = *(const bf16x8*)(As + row * ROWB + ((c ^ swz(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int row = wn0 + i * 16 + fr;
          ws2[P][kk][i] = *(const bf16x8*)(Ws + row * ROWB + ((c ^ swz(row)) << 4));
        }
      }
    };
    auto mfmas = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[P][kk][i], xs2[P][kk][j], acc[i][j], 0, 0, 0);
    };
    auto body = [&](auto pc, int kt) {
      constexpr int P = decltype(pc)::value;
      // DSTEP == 2: KT is even (host check), so tile parity P == position inside the step
      if (DSTEP == 1 || P == 0) __builtin_amdgcn_s_barrier();
      if constexpr (DBG >= 3) { if (tid == 0 && kt < 36) trc[4 + kt] = __builtin_amdgcn_s_memtime(); }
      reads(pc, buf);
      if (kt > 0) mfmas(std::integral_constant<int, 1 - P>{});
      if (DSTEP == 1 || P == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the next hand-over
      buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    };
    if constexpr (DBUF == 0) {
      // ONE fragment set, pipelined by HALF K-tiles (kk = 0 / 1: 32 of the tile's 64 k): the set's two halves take turns --
      // the reads of one half are in flight under the MFMAs of the other, as with two full sets but in half the registers.
      // Every wave passes the hand-over barrier at the same time, so without this all waves of the CU read (LDS-bound),
      // then all multiply (pipe-bound), and the two phases add up.  Reads of tile kt are retired before the barrier of
      // tile kt + 1 (the last wait of the step), behind which the loader refills tile kt - 1's ... kt's buffers in turn.
      auto reads_half = [&](int kk, int b) {
        if constexpr (DBG == 5) return;
        const char* As = ring + b * STAGE;
        const char* Ws = As + A_BYTES;
        const int c = kk * 4 + fq;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int row = wm0 + j * 16 + fr;
          xs2[0][kk][j] = *(const bf16x8*)(As + row * ROWB + ((c ^ swz(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int row = wn0 + i * 16 + fr;
          ws2[0][kk][i] = *(const bf16x8*)(Ws + row * ROWB + ((c ^ swz(row)) << 4));
        }
      };
      auto mfmas_half = [&](int kk) {
        if constexpr (DBG == 4 || DBG == 5) return;   // tools/gemm_trace.hip ablations: 4 = no MFMAs, 5 = no fragment reads either
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws2[0][kk][i], xs2[0][kk][j], acc[i][j], 0, 0, 0);
      };
      static_assert(BK == 64, "half-tile pipelining: two 32-k halves per K-tile");
      for (int kt = 0; kt < KT; ++kt) {
        __builtin_amdgcn_s_barrier();
        if constexpr (DBG >= 3) { if (tid == 0 && kt < 36) trc[4 + kt] = __builtin_amdgcn_s_memtime(); }
        reads_half(0, buf);
        if (kt > 0) mfmas_half(1);                       // second half of tile kt - 1 (read and retired in the last step)
        if (kt == 0) fetch_epilogue_operands();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reads_half(1, buf);
        mfmas_half(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tile kt's reads retired before the next hand-over
        buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
      }
      mfmas_half(1);
    } else {
      for (int kt = 0; kt < KT; kt += 2) {
        body(std::integral_constant<int, 0>{}, kt);
        if (kt == 0) fetch_epilogue_operands();
        if (kt + 1 < KT) body(std::integral_constant<int, 1>{}, kt + 1);
      }
      if ((KT - 1) & 1) mfmas(std::integral_constant<int, 1>{});
      else mfmas(std::integral_constant<int, 0>{});
    }
  } else
  for (int kt = 0; kt < KT; ++kt) {
    const int rem = KT - 1 - kt;  // stages issued after tile kt
    // tile kt must have landed; the min(NSTAGE - 2, rem) younger stages may stay in flight, and so may the NPC
    // counted epilogue loads while tile kt is one of the prologue's
    const int nst = rem < NSTAGE - 2 ? rem : NSTAGE - 2;
    if (NPC > 0 && kt <= NSTAGE - 2) wait_stages<LPT, NPC>(nst);
    else wait_stages<LPT, 0>(nst);
    __builtin_amdgcn_s_barrier();  // tile kt landed for every wave; everyone is done reading tile kt-1's buffer
    if constexpr (DBG >= 3) { if (tid == 0 && kt < 36) trc[4 + kt] = __builtin_amdgcn_s_memtime(); }
    // Order inside a K-step: the tile's fragment reads FIRST, then the LDS-DMAs of tile kt + NSTAGE - 1, then the MFMAs.
    // A DMA instruction costs the wave 100-185 cycles of issue while the CU's address path is busy (all waves of the CU
    // stage right behind the same barrier); issued ahead of the ds_reads -- the order of rounds 1-2 -- that stall delayed
    // the reads and, behind them, every MFMA of the step.  Now the reads are in flight under the DMA issue and the MFMAs
    // start the moment it ends: C2 44.4 -> 43.1 ms per pass (FF2 -0.9 us, QKV / FF1 / out-proj -0.3 us each per launch;
    // variants measured: DMAs after the first half's MFMAs or after all MFMAs: +-0; under the second half's reads or between
    // the halves' MFMAs: the same gain within 0.1 ms; every DMA followed by its share of the MFMAs: +2.5 ms; DESIGN 4
    // "K-step order").
    const bool more = kt + NSTAGE - 1 < KT;
    const char* As = ring + buf * STAGE;
    const char* Ws = As + A_BYTES;
    bf16x8 xf[BK / 32][TM], wf[BK / 32][TN];
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int c = kk * 4 + fq;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm0 + j * 16 + fr;
        xf[kk][j] = *(const bf16x8*)(As + row * ROWB + ((c ^ swz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn0 + i * 16 + fr;
        wf[kk][i] = *(const bf16x8*)(Ws + row * ROWB + ((c ^ swz(row)) << 4));
      }
    }
    if (more) stage(nbuf, kt + NSTAGE - 1);
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
    buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NSTAGE) ? 0 : nbuf + 1;
  }

  if constexpr (DBG >= 3) { if (tid == 0) trc[3] = __builtin_amdgcn_s_memtime(); }
  float* fuse_lds = (float*)(smem + NSTAGE * STAGE);  // BM x 16 bytes behind the ring (FUSE != 0 launches only)
  if (FUSE == 1) {
    // Chan's parallel variance over the 64-column partials, fixed order: mean = avg(mean_p),
    // M2 = sum_p M2_p + 64 sum_p (mean_p - mean)^2, rstd = rsqrt(M2 / K + eps)
    const int sq = tid & 3, pp = a.ln_parts >> 2;
    float sm = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < pp) sm += pf_st[u][0];
    sm = add_xor2(add_xor1(sm));
    const float mean = sm / (float)a.ln_parts;
    const float cols = (float)(a.K / a.ln_parts);
    float m2 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < pp) {
        const float dm = pf_st[u][0] - mean;
        m2 += pf_st[u][1] + cols * dm * dm;
      }
    m2 = add_xor2(add_xor1(m2));
    if (sq == 0 && (NC == NSTATT || tid < NSTATT)) {
      fuse_lds[(tid >> 2) * 2] = mean;     // relative to the offset the row's xs was centred with (row_mean)
      fuse_lds[(tid >> 2) * 2 + 1] = rsqrtf(m2 / (float)a.K + a.ln_eps);
      // the producer behind this consumer centres with the row's CURRENT mean: one workgroup per row tile moves it along
      if (tile_n == 0 && m0 + (tid >> 2) < a.M) a.row_mean[m0 + (tid >> 2)] += mean;
    }
    __syncthreads();
    if constexpr (DBG >= 3) { if (tid == 0) trc[42] = __builtin_amdgcn_s_memtime(); }
  }
  f32x4 xn[FUSE == 2 ? TN : 1][FUSE == 2 ? TM : 1];
  if (PREF && EPI == EPI_GATE_RES) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm0 + j * 16 + fr;
      pf_live[j] = (m < a.M) && (m - div_magic(m, a.rows_per_seq, a.rps_magic) * a.rows_per_seq < pf_len[j]);
    }
  }

  int qkv_which = 0, qkv_head = 0, qkv_d0 = 0;
  bf16* qkv_base = nullptr;
  size_t qkv_seq_stride = 0;
  if (EPI == EPI_QKV_ROPE) {
    static_assert(EPI != EPI_QKV_ROPE || WN <= 64, "a wave's columns must stay inside one 64-wide head");
    const int inner = a.heads * 64, hb = n0 + wn0;   // hb: multiple of WN, wave-uniform
    qkv_which = hb / inner;
    qkv_head = (hb - qkv_which * inner) >> 6;
    qkv_d0 = hb & 63;
    qkv_seq_stride = (size_t)a.heads * a.n_pad * 64;
    qkv_base = (qkv_which == 0 ? a.q : (qkv_which == 1 ? a.k : a.vt)) + (size_t)qkv_head * a.n_pad * 64;
  }

  // ---- epilogue: acc[i][j][r] = C[m = m0+wm0+16j+fr][n = n0+wn0+16i+4fq+r] ----
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm0 + j * 16 + fr;
    if (FUSE == 2) {
#pragma unroll
      for (int i = 0; i < TN; ++i) xn[i][j] = pf_x[i][j];  // rows past M hold zeros (never stored)
    }
    if (m >= a.M) continue;
    float ln_mean = 0.f, ln_rstd = 1.f;
    if (FUSE == 1) {
      ln_mean = fuse_lds[(wm0 + j * 16 + fr) * 2];
      ln_rstd = fuse_lds[(wm0 + j * 16 + fr) * 2 + 1];
    }
    int seq = 0, pos = m;
    if (EPI == EPI_GATE_RES || EPI == EPI_QKV_ROPE) {
      seq = div_magic(m, a.rows_per_seq, a.rps_magic);
      pos = m - seq * a.rows_per_seq;
    }
    // optional qk RMSNorm (reference modules.py:464-467 + :275-294): needs a whole head (64 columns) in one wave,
    // i.e. the 128-wide tile; 16 values in-lane, the other 48 in the lanes fr + 16/32/48
    float qk_rn = 1.0f;
    const float* qk_w = nullptr;
    if constexpr (EPI == EPI_QKV_ROPE && WN == 64) {
      const int which_w = (n0 + wn0) / (a.heads * 64);
      if (a.qn_w && which_w < 2) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          f32x4 t = acc[i][j];
          if (a.bias) t += *(const f32x4*)(a.bias + n0 + wn0 + i * 16 + fq * 4);
          ss += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
        }
        ss = add_xor32(add_xor16(ss));
        qk_rn = rsqrtf(ss * (1.0f / 64.0f) + a.qk_eps);
        qk_w = which_w == 0 ? a.qn_w : a.kn_w;
      }
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn0 + i * 16 + fq * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if (FUSE == 1) {
        v = ln_rstd * (v - ln_mean * pf_c[i]) + pf_d[i];  // d carries the bias
      } else if (PREF) {
        v += pf_bias[i];
      } else if (a.bias) {
        const f32x4 b = *(const f32x4*)(a.bias + n);
        v += b;
      }
      if (EPI == EPI_BF16) {
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) = f2bf4(v[0], v[1], v[2], v[3]);
      } else if (EPI == EPI_BF16_GELU) {
        *(bf16x4*)((bf16*)a.out + (size_t)m * a.ldo + n) =
            f2bf4(gelu_tanh_f(v[0]), gelu_tanh_f(v[1]), gelu_tanh_f(v[2]), gelu_tanh_f(v[3]));
      } else if (EPI == EPI_F32) {
        *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
      } else if (EPI == EPI_GATE_RES && PREF) {
        if (pf_live[j]) {  // pf_x / pf_gate were fetched from clamped addresses: valid whenever m < M and n < N
          const f32x4 x_new = pf_x[i][j] + pf_gate[i][COLG ? 0 : j] * v;
          *(f32x4*)(a.resid + (size_t)m * a.ldr + n) = x_new;
          if (FUSE == 2) xn[i][j] = x_new;
        }
        if (FUSE == 2) {
          const f32x4 y = (xn[i][j] - pf_rm[j]) * (1.0f + pf_ns[i][COLG ? 0 : j]);
          *(bf16x4*)(a.xs_out + (size_t)m * a.ld_xs + n) = f2bf4(y[0], y[1], y[2], y[3]);
        }
      } else if (EPI == EPI_GATE_RES) {
        const bool live = (a.seq_len == nullptr) || (pos < a.seq_len[seq]);
        if (live) {
          const size_t eoff = a.eval_ptr ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
          const f32x4 g = *(const f32x4*)(a.gate + eoff + (size_t)(seq % a.gate_rows) * a.gate_stride + n);
          float* xp = a.resid + (size_t)m * a.ldr + n;
          f32x4 x = *(const f32x4*)xp;
          x += g * v;
          *(f32x4*)xp = x;
        }
      } else if (EPI == EPI_QKV_ROPE) {
        // fragment-major layouts consumed by attention.hip (index maps documented there).  A wave's columns lie inside
        // ONE head (WN <= 64, aligned), so q / k / v and the head are wave-uniform scalars and every offset is
        // (row part: sequence, position) + (column part: head, d) -- no division or 64-bit chain per quad.
        const int d = qkv_d0 + i * 16 + fq * 4;
        if (qk_w) v = v * qk_rn * *(const f32x4*)(qk_w + d);
        if (qkv_which < 2) {
          if (qkv_head < a.rope_heads) {
            const f32x4 cs = *(const f32x4*)(a.cos_sin + (size_t)pos * 64 + d);   // [(d >> 1)][2] floats
            const float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
            v[0] = x0 * cs[0] - x1 * cs[1];
            v[1] = x1 * cs[0] + x0 * cs[1];
            v[2] = x2 * cs[2] - x3 * cs[3];
            v[3] = x3 * cs[2] + x2 * cs[3];
          }
          if (qkv_which == 0) v *= a.q_scale;   // softmax scale and log2(e), folded in here (attention.hip)
          // [tile = pos/32][ks = d/16][pos%32][(d/8)%2][d%8]
          const int off = (pos >> 5) * 2048 + (pos & 31) * 16 + (d >> 4) * 512 + ((d >> 3) & 1) * 8 + (d & 7);
          *(bf16x4*)(qkv_base + (size_t)seq * qkv_seq_stride + off) = f2bf4(v[0], v[1], v[2], v[3]);
        } else {
          // [group = pos/16][dt = d/32][d%32][h][j], pos%16 = 8 (j>>2) + 4 h + (j&3)
          const int k16 = pos & 15;
          const int off = (pos >> 4) * 1024 + (d >> 5) * 512 + (d & 31) * 16 + ((k16 >> 2) & 1) * 8 + (((k16 >> 3) << 2) | (k16 & 3));
          bf16* dst = qkv_base + (size_t)seq * qkv_seq_stride + off;
          dst[0] = (bf16)v[0];
          dst[16] = (bf16)v[1];
          dst[32] = (bf16)v[2];
          dst[48] = (bf16)v[3];
        }
      }
    }
  }
  if constexpr (DBG >= 3) { if (tid == 0) trc[43] = __builtin_amdgcn_s_memtime(); }
  if (FUSE == 2) {
    // (mean, M2) of x_new over this wave's 32 columns, then the two waves of a row pair up through LDS
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) sm += (xn[i][j][0] + xn[i][j][1]) + (xn[i][j][2] + xn[i][j][3]);
      sm = add_xor32(add_xor16(sm));
      const float mw = sm * (1.0f / (float)WN);
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const f32x4 dv = xn[i][j] - mw;
        q2 += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
      }
      q2 = add_xor32(add_xor16(q2));
      if (fq == 0) {
        float* slot = fuse_lds + ((wm0 + j * 16 + fr) * WGN + (wave % WGN)) * 2;
        slot[0] = mw - pf_rm[j];   // relative to the row's centring offset
        slot[1] = q2;
      }
    }
    // LDS hand-off only: do not drain vmcnt here, the x / xs stores above stay in flight behind the barrier
    if constexpr (DBG >= 3) { if (tid == 0) trc[44] = __builtin_amdgcn_s_memtime(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DBG >= 3) { if (tid == 0) trc[45] = __builtin_amdgcn_s_memtime(); }
    if (tid < BM && m0 + tid < a.M) {
      const float ma = fuse_lds[tid * 4], qa = fuse_lds[tid * 4 + 1], mb = fuse_lds[tid * 4 + 2], qb = fuse_lds[tid * 4 + 3];
      const float dm = ma - mb;
      float* dst = a.stats_out + ((size_t)(m0 + tid) * a.tiles_n + tile_n) * 2;
      dst[0] = 0.5f * (ma + mb);
      dst[1] = (qa + qb) + (0.25f * (float)BN) * dm * dm;  // delta^2 na nb / (na + nb), na = nb = BN / 2
    }
  }
  if constexpr (DBG >= 3) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { trc[40] = __builtin_amdgcn_s_memtime(); trc[41] = __builtin_amdgcn_s_memrealtime(); }
  }
}

// Column-group width (as a shift) for the n-major tile walk of a tiles_m x tiles_n grid whose logical ids are dealt to the 8
// XCDs in contiguous runs (the kernel's bijection): the G = 2^s (dividing tiles_n) that minimises the number of distinct
// 64-row operand panels (A row tiles + W column tiles) summed over the XCDs' runs -- what their private L2s fetch through
// the fabric.  At M = 938: out-proj / FF2 (15 x 16 tiles) G = 4: 8 + 4 panels per XCD instead of 15 + 2; FF1 (15 x 32) G = 8.
// A few hundred integer operations; memoised per thread for the eager launch path.
inline int pick_group_shift(int tiles_m, int tiles_n, int bm, int bn) {
  struct Memo { int tm, tn, bm, bn, s; };
  static thread_local Memo memo[4] = {};
  static thread_local int next = 0;
  for (const Memo& e : memo)
    if (e.tm == tiles_m && e.tn == tiles_n && e.bm == bm && e.bn == bn) return e.s;
#ifdef F5E_TOOLS
  if (const char* f = getenv("F5E_GEMM_GROUP_SHIFT")) {   // diagnostics build only: A/B of the walk
    const int s = atoi(f);
    if (s >= 0 && s <= 4 && tiles_n % (1 << s) == 0) return s;
  }
#endif
  const int n = tiles_m * tiles_n, q8 = n >> 3, r8 = n & 7;
  long best = -1;
  int best_s = 0;
  for (int s = 0; s <= 4 && (tiles_n % (1 << s)) == 0 && (1 << s) <= tiles_n; ++s) {
    long cost = 0;
    int id = 0;
    for (int x = 0; x < 8; ++x) {
      const int cnt = q8 + (x < r8 ? 1 : 0);
      unsigned long long seen_m[4] = {}, seen_n[4] = {};   // up to 256 tiles per side
      for (int i = 0; i < cnt; ++i, ++id) {
        const int per = tiles_m << s, g = id / per, r = id - g * per;
        const int tm = r >> s, tn = (g << s) + (r & ((1 << s) - 1));
        seen_m[(tm >> 6) & 3] |= 1ull << (tm & 63);
        seen_n[(tn >> 6) & 3] |= 1ull << (tn & 63);
      }
      for (int w = 0; w < 4; ++w) cost += (long)__builtin_popcountll(seen_m[w]) * bm + (long)__builtin_popcountll(seen_n[w]) * bn;
    }
    if (best < 0 || cost < best) { best = cost; best_s = s; }
  }
  memo[next] = Memo{tiles_m, tiles_n, bm, bn, best_s};
  next = (next + 1) & 3;
  return best_s;
}

// diagnostics build only: F5E_PF_PACK=0 -> 32 KiB prefetch workgroups everywhere (the round-3 layout)
inline bool pack_prefetch_on() {
#ifdef F5E_TOOLS
  if (const char* f = getenv("F5E_PF_PACK")) return atoi(f) != 0;
#endif
  return true;
}

template <int BM, int BN, int EPI, int NSTAGE, int WGM = 2, int WGN = 2, int DBG = 0, int FUSE = 0, int NLOAD = 0, int DSTEP = 1, int DBUF = 1>
int launch(GemmArgs& a, hipStream_t st) {
  constexpr int BK = 64;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  if (a.M > a.N) {
    a.m_major = 1;
    a.tile_magic = div_magic_of(a.tiles_n);
  } else {
    const int sh = (a.tiles_m <= 256 && a.tiles_n <= 256) ? pick_group_shift(a.tiles_m, a.tiles_n, BM, BN) : 0;
    a.m_major = -sh;
    a.tile_magic = div_magic_of(a.tiles_m << sh);
  }
  a.rps_magic = div_magic_of(a.rows_per_seq);
  a.n_main = a.tiles_m * a.tiles_n;
  // prefetch workgroups ride along only where the main grid leaves room on the chip for them to start at once
  int pf_wgs = ((WGM * WGN == 4 || NLOAD > 0) && a.n_main <= 3 * 256) ? f5e_prefetch_wgs(&a.pf) : 0;
  a.pf_per_wg = 0;
  constexpr int lds = NSTAGE * (BM + BN) * BK * 2 + (FUSE ? BM * 16 : 0);
  if (NLOAD > 0 && 2 * lds > 160 * 1024 && pf_wgs > 0 && pack_prefetch_on()) {
    // one workgroup per CU (LDS): the prefetch goes to the CUs the main grid leaves idle, one packed workgroup each; a
    // grid that fills the chip hosts nothing (its prefetch workgroups could only run as a tail behind it)
    const int idle = f5e_cu_count() - a.n_main;
    if (idle <= 0) {
      pf_wgs = 0;
    } else if (pf_wgs > idle) {
      constexpr unsigned gran = 64u * (WGM * WGN + NLOAD) * 16u;   // one LDS-DMA of every thread of the workgroup
      const unsigned long long total = (unsigned long long)pf_wgs * F5E_PF_BYTES_PER_WG;
      a.pf_per_wg = (unsigned)(((total + idle - 1) / idle + gran - 1) / gran * gran);
      pf_wgs = (int)((total + a.pf_per_wg - 1) / a.pf_per_wg);
    }
  }
  const int grid = a.n_main + pf_wgs;
  if (grid == a.n_main) a.pf = F5ePrefetch{};
  static_assert(lds <= 160 * 1024, "LDS budget");
  static F5eDeviceOnce lds_once;  // > 64 KiB of dynamic LDS needs the opt-in attribute, per device (host-only call)
  if (lds > 65536) F5E_OPT_IN_LDS(lds_once, (gemm_bf16_kernel<BM, BN, EPI, NSTAGE, DBG, WGM, WGN, FUSE, NLOAD, DSTEP, DBUF>), lds);
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI, NSTAGE, DBG, WGM, WGN, FUSE, NLOAD, DSTEP, DBUF>), dim3(grid), dim3(64 * (WGM * WGN + NLOAD)), lds, st, a);
  F5E_LAUNCH_CHECK("gemm_bf16");
  return F5E_OK;
}

// tile_hint: 0 = auto; 1 = 128x128 (8 waves, 2 stages), 2 = 128x64 (3 stages), 3 = 64x64 (3 stages; 4 for a one-round grid
// with K >= 2048), 9 = the 256x256 ping-pong kernel of gemm_bf16_pp.hip (what auto picks at large M).  One code path per
// tile family: the tuning variants that were measured and lost (deeper rings, other wave grids, split-K) are in DESIGN 4.
inline bool role_split_on() {
#ifdef F5E_TOOLS
  if (const char* f = getenv("F5E_GEMM_ROLE")) return atoi(f) != 0;   // diagnostics build only: A/B against the classic ring
#endif
  return true;
}
// diagnostics build only: F5E_GEMM_VAR=0 -> one K-tile per hand-over (4- / 3-stage rings) instead of two on a ring of six
inline int role_var() {
#ifdef F5E_TOOLS
  if (const char* f = getenv("F5E_GEMM_VAR")) return atoi(f);
#endif
  return -1;
}

// diagnostics build only: F5E_GEMM_BIG=0 -> no 128-row role-split tiles (the classic 64 x 64 kernels past M = 1024)
inline bool big_tiles_on() {
#ifdef F5E_TOOLS
  if (const char* f = getenv("F5E_GEMM_BIG")) return atoi(f) != 0;
#endif
  return true;
}

template <int EPI>
int dispatch(GemmArgs& a, hipStream_t st, int tile_hint) {
  auto blocks = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  int sel = tile_hint;
  F5E_REQUIRE(sel == 0 || sel == 1 || sel == 2 || sel == 3 || sel == 9, "gemm_bf16: unknown tile_hint %d (0, 1, 2, 3 or 9)", tile_hint);
  if (a.ln_stats || a.stats_out) {  // fused AdaLN: the 64x64 tile family (f5e_ln_fuse; small row counts)
    F5E_REQUIRE(!(a.ln_stats && a.stats_out), "gemm_bf16: a launch is an AdaLN consumer or a producer, not both");
    F5E_REQUIRE(!(EPI == EPI_QKV_ROPE && a.qn_w), "gemm_bf16: fused AdaLN and qk_norm need different tiles");
    if constexpr (EPI == EPI_GATE_RES) {
      F5E_REQUIRE(a.stats_out && !a.ln_stats, "gemm_bf16: the gate+residual epilogue is the AdaLN producer");
      // one-round grids (out-projection / FF2 at batch 1: 240 workgroups on 256 CUs): role split, 4 loader + 4 consumer
      // waves, 4-stage ring -- see the NLOAD note at the kernel
      if (role_split_on() && a.gate_rows == 1 && blocks(64, 64) <= 256) {
        // two K-tiles per hand-over barrier on a ring of six (in situ at C2, ms per pass: classic ring 41.4; 4 stages, one tile
        // per barrier 40.9; this 40.7; 8-tile ring 41.2; 6 stages, one tile per barrier 41.8)
        if (a.K % 128 == 0 && role_var() != 0) return launch<64, 64, EPI, 6, 2, 2, 0, 2, 4, 2>(a, st);
        return launch<64, 64, EPI, 4, 2, 2, 0, 2, 4, 1>(a, st);   // K % 128 != 0 (or F5E_GEMM_VAR=0 in the diagnostics build)
      }
      // M in (1024, 2048] (C4's utterances): 128 x 64 tiles keep the grid to one round of role-split workgroups
      if (role_split_on() && a.gate_rows == 1 && big_tiles_on() && blocks(128, 64) <= 256) {
        if (a.K % 128 == 0) return launch<128, 64, EPI, 6, 4, 2, 0, 2, 4, 2>(a, st);
        return launch<128, 64, EPI, 4, 4, 2, 0, 2, 4, 1>(a, st);
      }
      if (a.K >= 2048 && blocks(64, 64) <= 256) return launch<64, 64, EPI, 4, 2, 2, 0, 2>(a, st);
      return launch<64, 64, EPI, 3, 2, 2, 0, 2>(a, st);
    } else {
      F5E_REQUIRE(a.ln_stats && !a.stats_out, "gemm_bf16: this epilogue can only consume AdaLN statistics");
      // one-round grids of WIDE tiles (batch 1: QKV 15 x 16 tiles of 64 x 192, FF1 15 x 16 of 64 x 128 on 256 CUs): role
      // split, 4 loader waves + 12 / 8 consumer waves of 32 x 32 sub-tiles; the A panel is staged once per K-step for 3 / 2
      // classic tiles' worth of columns, which is what the address path -- the bound resource -- is spared
      if (role_split_on()) {
        int wide = 3;
#ifdef F5E_TOOLS
        if (const char* f = getenv("F5E_GEMM_WIDE")) wide = atoi(f);   // diagnostics build only: bit 0 QKV, bit 1 the others
#endif
        if constexpr (EPI == EPI_QKV_ROPE) {
          if ((wide & 1) && a.N % 192 == 0 && blocks(64, 192) <= 256) {
            return launch<64, 192, EPI, 4, 2, 6, 0, 1, 4, 1>(a, st);
          }
          // ring of 3 x 40 KiB (a 4th stage, with the statistics scratch aliased into the ring, changed nothing in the K loop
          // and delayed the first tile: DESIGN 4)
          if (big_tiles_on() && a.N % 192 == 0 && blocks(128, 192) <= 256) return launch<128, 192, EPI, 3, 2, 6, 0, 1, 4, 1, 0>(a, st);
        } else {
          if ((wide & 2) && a.N % 128 == 0 && blocks(64, 128) <= 256) {
#ifdef F5E_TOOLS
            if (role_var() == 0) return launch<64, 128, EPI, 3, 2, 4, 0, 1, 4, 1>(a, st);   // diagnostics build: one tile per hand-over
#endif
            // two K-tiles per hand-over: a fused consumer's K is a multiple of 256 (statistics parts % 4 == 0, set_consumer)
            if (a.K % 128 == 0) return launch<64, 128, EPI, 6, 2, 4, 0, 1, 4, 2>(a, st);
          }
          if (big_tiles_on() && a.N % 128 == 0 && blocks(128, 128) <= 256) return launch<128, 128, EPI, 4, 2, 4, 0, 1, 4, 1, 0>(a, st);
        }
      }
      return launch<64, 64, EPI, 3, 2, 2, 0, 1>(a, st);
    }
  }
  // large M: the 256x256 ping-pong kernel (gemm_bf16_pp.hip).  Crossover against the 128x128 ring kernel measured on the
  // four DiT shapes: M = 7.5k ring kernel ahead by 0-40 %, M = 13k ping-pong ahead by 5-18 %, M = 30k by 8-37 %: it takes
  // over at 44 row tiles of 256, whatever N.
  if (sel == 9 || (sel == 0 && uses_pp(a.M, a.K))) return launch_pp(EPI, a, st);
  if (EPI == EPI_QKV_ROPE && a.qn_w) sel = 1;  // qk_norm reduces over a head inside one wave: 128-wide tiles only
  if (sel == 0) {
    // the largest tile that still gives ~2 blocks per CU (256 CUs): these GEMMs are latency-bound at small M (64x64 with
    // 3 stages is fastest for every DiT shape at M = 938)
    if (blocks(128, 128) >= 320) sel = 1;  // 8-wave 128x128: ahead of the smaller tiles from ~1.3 workgroups per CU on
    else if (blocks(128, 64) >= 512) sel = 2;
    else sel = 3;
  }
  // 128x128: 8 waves (4 x 2, 32 x 64 per wave), 2 stages = 64 KiB of LDS -> 2 workgroups per CU = 16 waves per CU to
  // cover the per-K-step waits.  64x64 with one workgroup per CU and a long K (FF2: K = 2048, weights streaming from
  // HBM): a 4th stage hides the HBM latency (15.5 -> 12.9 us at M = 938); at K = 1024 or 3 workgroups per CU it does not.
  if (sel == 1) return launch<128, 128, EPI, 2, 4, 2>(a, st);
  if (sel == 2) return launch<128, 64, EPI, 3>(a, st);
  if constexpr (EPI == EPI_GATE_RES) {
    if (role_split_on() && a.gate_rows == 1 && blocks(64, 64) <= 256) return launch<64, 64, EPI, 4, 2, 2, 0, 0, 4>(a, st);
  }
  if (a.K >= 2048 && blocks(64, 64) <= 256) return launch<64, 64, EPI, 4>(a, st);
  return launch<64, 64, EPI, 3>(a, st);
}

int check_common(const GemmArgs& a) {
  F5E_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_bf16: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  F5E_REQUIRE(a.K % 64 == 0, "gemm_bf16: K=%d must be a multiple of 64", a.K);
  F5E_REQUIRE(a.N % 4 == 0, "gemm_bf16: N=%d must be a multiple of 4", a.N);
  F5E_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm_bf16: lda/ldw must be multiples of 8 elements");
  F5E_REQUIRE(a.A && a.W, "gemm_bf16: null operand");
  return F5E_OK;
}

int set_consumer(GemmArgs& a, const f5e_ln_fuse* ln, const float* bias) {
  if (!ln || !ln->stats) return F5E_OK;
  F5E_REQUIRE(bias == nullptr, "gemm_bf16: with fused AdaLN the bias is part of the d table");
  F5E_REQUIRE(ln->c && ln->d && ln->parts > 0 && ln->parts <= 16 && ln->parts % 4 == 0 && ln->cd_rows == 1 && a.K % ln->parts == 0 && ln->cd_rows > 0 &&
                  ln->cd_stride % 4 == 0 && ln->cd_eval_stride % 4 == 0 && ln->rows_per_seq > 0,
              "gemm_bf16: bad fused-AdaLN consumer arguments (parts=%d: a multiple of 4 up to 16; one table row per evaluation)",
              ln->parts);
  a.ln_stats = ln->stats; a.ln_parts = ln->parts; a.ln_c = ln->c; a.ln_d = ln->d; a.cd_stride = ln->cd_stride;
  a.cd_rows = ln->cd_rows; a.cd_eval_stride = ln->cd_eval_stride; a.ln_eps = ln->eps;
  a.eval_ptr = ln->eval_ptr;
  F5E_REQUIRE(ln->row_mean, "gemm_bf16: the fused-AdaLN consumer needs f5e_ln_fuse.row_mean (the rows' centring offsets)");
  a.row_mean = ln->row_mean;
  if (a.rows_per_seq <= 0) a.rows_per_seq = ln->rows_per_seq;
  return F5E_OK;
}

}  // namespace

// Internal entry points (f5e_dit_forward): the *_ln ABI functions plus an Infinity-Cache prefetch hint
int f5e_gemm_bf16_bias_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                          int ldo, int M, int N, int K, int act, int out_f32, int tile_hint, const f5e_ln_fuse* ln,
                          const F5ePrefetch* pf) {
  GemmArgs a{};
  if (pf) a.pf = *pf;
  a.A = (const bf16*)A; a.lda = lda; a.W = (const bf16*)W; a.ldw = ldw; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.out = out; a.ldo = ldo;
  if (int e = check_common(a)) return e;
  if (int e = set_consumer(a, ln, bias)) return e;
  F5E_REQUIRE(out && ldo % 4 == 0, "gemm_bf16_bias: bad output");
  if (out_f32) {
    F5E_REQUIRE(act == F5E_ACT_NONE, "gemm_bf16_bias: f32 output supports no activation");
    return dispatch<EPI_F32>(a, st, tile_hint);
  }
  if (act == F5E_ACT_GELU_TANH) return dispatch<EPI_BF16_GELU>(a, st, tile_hint);
  F5E_REQUIRE(act == F5E_ACT_NONE, "gemm_bf16_bias: unsupported activation %d", act);
  return dispatch<EPI_BF16>(a, st, tile_hint);
}

int f5e_gemm_bf16_gate_residual_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                   const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                   int N, int K, int tile_hint, const f5e_ln_fuse* ln, const F5ePrefetch* pf) {
  GemmArgs a{};
  if (pf) a.pf = *pf;
  a.A = (const bf16*)A; a.lda = lda; a.W = (const bf16*)W; a.ldw = ldw; a.bias = bias;
  a.M = M; a.N = N; a.K = K;
  a.resid = resid; a.ldr = ldr; a.gate = gate; a.gate_stride = gate_stride; a.gate_rows = gate_rows;
  a.rows_per_seq = rows_per_seq; a.seq_len = seq_len; a.eval_ptr = eval_ptr; a.eval_stride = eval_stride;
  if (int e = check_common(a)) return e;
  F5E_REQUIRE(resid && gate && ldr % 4 == 0 && gate_stride % 4 == 0 && gate_rows > 0 && rows_per_seq > 0,
              "gemm_bf16_gate_residual: bad residual/gate arguments");
  if (ln && ln->stats_out) {
    F5E_REQUIRE(ln->xs_out && ln->next_scale && ln->row_mean && ln->ld_xs % 4 == 0 && N % 64 == 0,
                "gemm_bf16_gate_residual: AdaLN producer needs xs_out, next_scale, row_mean and N %% 64 == 0");
    a.xs_out = (bf16*)ln->xs_out; a.ld_xs = ln->ld_xs; a.next_scale = ln->next_scale; a.stats_out = ln->stats_out;
    a.row_mean = ln->row_mean;
  }
  return dispatch<EPI_GATE_RES>(a, st, tile_hint);
}

int f5e_gemm_bf16_qkv_rope_pf(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* q,
                              void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                              const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K, int tile_hint,
                              const f5e_ln_fuse* ln, const F5ePrefetch* pf) {
  GemmArgs a{};
  if (pf) a.pf = *pf;
  a.A = (const bf16*)A; a.lda = lda; a.W = (const bf16*)W; a.ldw = ldw; a.bias = bias;
  a.M = M; a.N = 3 * heads * 64; a.K = K;
  a.q = (bf16*)q; a.k = (bf16*)k; a.vt = (bf16*)vt; a.n_pad = n_pad; a.heads = heads; a.rope_heads = rope_heads;
  a.cos_sin = cos_sin; a.rows_per_seq = rows_per_seq;
  a.qn_w = q_norm_w; a.kn_w = k_norm_w; a.qk_eps = 1e-6f;
  a.q_scale = 0.125f * 1.4426950408889634f;
  F5E_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), "gemm_bf16_qkv_rope: q and k norm weights go together");
  if (int e = check_common(a)) return e;
  F5E_REQUIRE(q && k && vt && cos_sin, "gemm_bf16_qkv_rope: null output/table");
  F5E_REQUIRE(heads > 0 && rope_heads >= 0 && rope_heads <= heads, "gemm_bf16_qkv_rope: bad head counts");
  F5E_REQUIRE(rows_per_seq > 0 && n_pad >= rows_per_seq && n_pad % 64 == 0,
              "gemm_bf16_qkv_rope: n_pad=%d must be a multiple of 64 and >= rows_per_seq=%d", n_pad, rows_per_seq);
  if (int e = set_consumer(a, ln, bias)) return e;
  return dispatch<EPI_QKV_ROPE>(a, st, tile_hint);
}

extern "C" {

int f5e_gemm_bf16_bias(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                       int ldo, int M, int N, int K, int act, int out_f32, int tile_hint) {
  return f5e_gemm_bf16_bias_ln(st, A, lda, W, ldw, bias, out, ldo, M, N, K, act, out_f32, tile_hint, nullptr);
}

int f5e_gemm_bf16_bias_ln(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                          int ldo, int M, int N, int K, int act, int out_f32, int tile_hint, const f5e_ln_fuse* ln) {
  return f5e_gemm_bf16_bias_pf(st, A, lda, W, ldw, bias, out, ldo, M, N, K, act, out_f32, tile_hint, ln, nullptr);
}

int f5e_gemm_bf16_gate_residual(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                int N, int K, int tile_hint) {
  return f5e_gemm_bf16_gate_residual_ln(st, A, lda, W, ldw, bias, resid, ldr, gate, gate_stride, gate_rows, eval_ptr,
                                        eval_stride, rows_per_seq, seq_len, M, N, K, tile_hint, nullptr);
}

int f5e_gemm_bf16_gate_residual_ln(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                   const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                   int N, int K, int tile_hint, const f5e_ln_fuse* ln) {
  return f5e_gemm_bf16_gate_residual_pf(st, A, lda, W, ldw, bias, resid, ldr, gate, gate_stride, gate_rows, eval_ptr,
                                        eval_stride, rows_per_seq, seq_len, M, N, K, tile_hint, ln, nullptr);
}

int f5e_gemm_bf16_qkv_rope(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias, void* q,
                           void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                           const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K,
                           int tile_hint) {
  return f5e_gemm_bf16_qkv_rope_ln(st, A, lda, W, ldw, bias, q, k, vt, n_pad, heads, rope_heads, cos_sin, q_norm_w,
                                   k_norm_w, rows_per_seq, M, K, tile_hint, nullptr);
}

int f5e_gemm_bf16_qkv_rope_ln(hipStream_t st, const void* A, int lda, const void* W, int ldw, const float* bias,
                              void* q, void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                              const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K,
                              int tile_hint, const f5e_ln_fuse* ln) {
  return f5e_gemm_bf16_qkv_rope_pf(st, A, lda, W, ldw, bias, q, k, vt, n_pad, heads, rope_heads, cos_sin, q_norm_w,
                                   k_norm_w, rows_per_seq, M, K, tile_hint, ln, nullptr);
}

}  // extern "C"
