// Flash-style non-causal self-attention for the DiT blocks (K10 of SURVEY 2.3), head dim 64, bf16 in / bf16 out.
//
//   O[s, q, h*64 + d] = sum_key softmax_key(Q[s,h,q,:] . K[s,h,key,:] / 8 + keymask) * V[s,h,key,d]
//
// Replaces F.scaled_dot_product_attention at reference model/modules.py:482-492 (key-padding mask only, no dropout).
// Instead of a dense bool mask the kernel takes per-sequence key counts: keys >= kv_len[s] get -inf.
//
// Operand layouts ("fragment-major", written by f5e_gemm_bf16_qkv_rope; index helpers in f5e_common.h):
//   Q, K : per (s, head) tiles of 32 positions: [tile][ks = d/16][pos % 32][h = (d/8) % 2][d % 8]   (bf16)
//   V    : per (s, head) tiles of 32 keys:      [tile][s16 = (key%32)/16][dt = d/32][d % 32][h][j]  (bf16)
//          with, inside a 16-key group, key%16 = 8 (j>>2) + 4 h + (j&3)
// so that the 16 bytes one lane feeds to v_mfma_f32_32x32x16_bf16 are contiguous and the 64 lanes of a wave read
// 1 KiB contiguous: every fragment is ONE fully coalesced global_load_dwordx4 per lane, straight into registers.
// At DiT sizes K/V of a head (<= 512 KiB at N = 4096, 59 KiB at N = 469) live in L2, so there is no LDS staging, no
// barrier and no bank conflict in the main loop; waves are independent and latency is hidden by occupancy plus a
// one-tile register prefetch.
//
// Work decomposition: one wave = 32 queries (lane owns query lane&31; the two 32-lane halves own interleaved key /
// d rows as dictated by the MFMA C layout).  A workgroup = NSPLIT waves that split the key tiles of the SAME 32
// queries (flash-decoding style) and merge their (m, l, O) through LDS at the end -- this is what fills 256 CUs at
// batch 1 (N = 469: 15 q-tiles x 32 (s, head) x 4 splits = 1920 waves).
//   S^T = K . Q^T  (lane-local softmax statistics),  O^T = V^T . P^T with P taken from the S^T accumulator registers
//   (cdna guide section 3 "An accumulator tile as the next MFMA's operand").
//
// Softmax arithmetic (round 3).  The kernel is bound by VALU issue, not by the matrix pipe: per 64-key step and wave the 16
// MFMAs take 512 cycles, while round 2's softmax issued 33 v_exp (8 cycles each) + ~115 other VALU instructions (4 each) =
// ~720 cycles (instruction histogram of the ISA).  Three changes take 54 of them out of every step:
//   * Q arrives PRE-SCALED by log2(e) / sqrt(64) (f5e_gemm_bf16_qkv_rope folds it in before the bf16 rounding), so the MFMA
//     result is the score in the log2 domain;
//   * the running maximum enters through the accumulator: the first MFMA of a score chain takes C = minit, a 16-register
//     tile holding -m_run (rewritten only when m_run moves), so p = exp2(acc) with no subtraction or multiply per element;
//   * the maximum is not tracked per step at all.  Floating point is scale-free: any m_run gives the same O / l as long as
//     nothing overflows, so m_run is the maximum of the query's FIRST step, and a later step only checks its row sums
//     (already needed for l): a sum above 2^24 (some score more than ~19-24 octaves above m_run; inf / NaN included)
//     sends the whole wave through the slow path -- recompute the step's scores from the operands still at hand, take the
//     true maximum, rescale (l, O), move m_run.  p <= 2^24 keeps l and O far from the fp32 range (N <= 4096 keys) and the
//     bf16 P operand as accurate as at any other scale.  Rare by construction: every pass through it raises a query's
//     m_run by at least ~19 octaves.
// Left per step: 32 v_exp, 32 v_add (row sums, half-wave partials: the two halves meet once at the end), 16 v_cvt_pk.
#include <cstdlib>
#include "f5e_common.h"

namespace {

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v;
  bf16* o; int ldo;
  const int* kv_len;  // [S] or null (= rows_per_seq)
  int S, H, rows_per_seq, n_pad;
  int n_main;         // workgroups beyond it only prefetch (NSPLIT >= 2 launches)
  F5ePrefetch pf;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr float ROWSUM_LIMIT = 16777216.0f;   // 2^24, see the header: a step whose partial row sum exceeds it is redone

// One 64-key step of the online softmax for a wave's 32 queries (lane = query ql, key half hh).
//   qk(st, c): st[t] = c + (scores of keys 64 step + 32 t + ..., log2 domain), t = 0, 1 -- the 8 MFMAs of the step; may be
//              called twice (slow path), so the K fragments must still be at hand
//   on return st holds p = exp2(score - m_run) for the step's valid keys (0 for keys >= kv_len), l_val this lane's
//   running partial row sum, (m_run, minit, oacc) updated if the running maximum moved.
template <class QK>
__device__ __forceinline__ void softmax_step(QK&& qk, bool first, bool partial, int key_base, int kv_len, f32x16 (&st)[2],
                                             f32x16& minit, float& m_run, float& l_val, f32x16 (&oacc)[2]) {
  auto mask_tail = [&]() {
    // only the last step of a sequence: a real branch (the empty asm keeps hipcc from if-converting the block into 32
    // compares + 32 selects executed on EVERY step)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (key_base + t * 32 + (r & 3) + 8 * (r >> 2) >= kv_len) st[t][r] = -INFINITY;
  };
  if (!first) {
    qk(st, minit);                       // score - m_run
    if (partial) mask_tail();
    // ONE accumulation chain on purpose: two independent chains are SLP-packed into v_pk_add_f32, which costs more issue
    // time beside MFMAs than the two v_add_f32 it replaces (cdna guide, cycle constants: packed f32 VALU is an anti-lever)
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[t][r] = fast_exp2(st[t][r]);
        rs += st[t][r];
      }
    if (__builtin_amdgcn_ballot_w64(!(rs <= ROWSUM_LIMIT)) == 0) {   // wave-uniform; NaN / inf fail the comparison
      l_val += rs;
      return;
    }
  }
  // slow path: the first step of a query tile, or some query of the wave met scores far above its m_run
  f32x16 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  qk(st, zero);
  if (partial) mask_tail();
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
  mx = max_xor32(mx);                    // finite: every processed step holds at least one valid key
  const float m_new = fmaxf(m_run, mx);
  const float alpha = fast_exp2(m_run - m_new);   // first step: exp2(-inf) = 0
  l_val *= alpha;
  m_run = m_new;
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; minit[r] = -m_new; }
  float rs = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[t][r] = fast_exp2(st[t][r] - m_new);
      rs += st[t][r];
    }
  l_val += rs;
}

// The unchecked step (see the header): scores arrive as score - m_run (C = minit), p = exp2 of them whatever their size --
// no maximum, no comparison, no branch.  Half 0's exp2 / row-sum VALU work is written between the two halves' MFMAs and
// half 1's between the P.V MFMAs, so that one wave keeps both pipes busy (a wave issues in order: VALU instructions placed
// behind a block of MFMAs wait for all of them to ISSUE, i.e. for the matrix pipe).
template <class KF, class VF>
__device__ __forceinline__ void fast_step(KF&& kfrag, VF&& vfrag, const bf16x8 (&qf)[4], const f32x16& minit, float& l_val,
                                          f32x16 (&oacc)[2]) {
  f32x16 s0, s1;
  s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(0, 0), qf[0], minit, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(0, ks), qf[ks], s0, 0, 0, 0);
  s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(1, 0), qf[0], minit, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(1, ks), qf[ks], s1, 0, 0, 0);
  float rs = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s0[r] = fast_exp2(s0[r]); rs += s0[r]; }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = (bf16)s0[8 * s + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(0, s, dt), pf, oacc[dt], 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { s1[r] = fast_exp2(s1[r]); rs += s1[r]; }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = (bf16)s1[8 * s + j];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(1, s, dt), pf, oacc[dt], 0, 0, 0);
  }
  l_val += rs;
}

// true when this lane's row sum and output accumulators are all finite (inf and NaN fail `<=`)
__device__ __forceinline__ bool accum_finite(float l_val, const f32x16 (&oacc)[2]) {
  float mag = fabsf(l_val);   // a SUM, not a maximum: fmaxf drops NaNs, an addition keeps them (and infinities)
#pragma unroll
  for (int r = 0; r < 16; ++r) mag += fabsf(oacc[0][r]) + fabsf(oacc[1][r]);
  return mag <= 3.0e38f;
}

template <int NSPLIT>
__global__ __launch_bounds__(NSPLIT * 64, 2) void attn_fwd_kernel(AttnArgs a) {   // 2 waves per SIMD: <= 256 VGPRs
  // merge buffers: per extra wave, per lane: 32 O values + m + l
  __shared__ __attribute__((aligned(16))) float red[(NSPLIT > 1 ? NSPLIT - 1 : 1) * 64 * 34];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  if ((int)blockIdx.x >= a.n_main) {   // grid-tail workgroups: Infinity-Cache prefetch only (f5e_common.h)
    f5e_prefetch_run<NSPLIT * 64>(a.pf, (int)blockIdx.x - a.n_main, threadIdx.x, red);
    return;
  }

  const int qtiles = (a.rows_per_seq + 31) / 32;
  // XCD-aware order: blocks with equal blockIdx%8 share an XCD (and its L2); give each XCD a contiguous range of
  // logical ids so all q-tiles of one (sequence, head) -- which re-read the same K/V -- hit the same L2.
  int bid = blockIdx.x;
  {
    const int nblk = a.n_main;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int qt = bid % qtiles;
  bid /= qtiles;
  const int head = bid % a.H;
  const int seq = bid / a.H;
  const size_t sh = (size_t)seq * a.H + head;
  const int kv_len = a.kv_len ? min(a.kv_len[seq], a.rows_per_seq) : a.rows_per_seq;
  const int ntiles = (kv_len + 63) / 64;  // 64-key steps

  const bf16* Qg = a.q + sh * a.n_pad * 64;
  const bf16* Kg = a.k + sh * a.n_pad * 64;
  const bf16* Vg = a.v + sh * a.n_pad * 64;

  // Q fragments (B operand of S^T = K.Q^T): 4 x 16 B
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + ((size_t)(qt * 4 + ks) * 32 + ql) * 16 + hh * 8);

  // this wave's 64-key steps: wave, wave + NSPLIT, ...
  bf16x8 kf[2][4], vf[2][2][2];   // [t32][ks], [t32][s16][dt]
  auto load_tile = [&](int t64) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const size_t tile = (size_t)t64 * 2 + t;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[t][ks] = *(const bf16x8*)(Kg + ((tile * 4 + ks) * 32 + ql) * 16 + hh * 8);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          vf[t][s][dt] = *(const bf16x8*)(Vg + ((((tile * 2 + s) * 2 + dt) * 32 + ql) * 2 + hh) * 8);
    }
  };

  f32x16 oacc[2], minit;
  float m_run, l_val;
  bf16x8 ck[2][4], cv[2][2][2];   // the current step's fragments (kf / vf already hold the next step's)
  int t64;
  // current tile's fragments move to compute registers; the next tile's loads are issued before the MFMAs
  auto advance = [&]() -> int {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ck[t][ks] = kf[t][ks];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) cv[t][s][dt] = vf[t][s][dt];
    }
    const int cur = t64;
    t64 += NSPLIT;
    if (t64 < ntiles) load_tile(t64);
    return cur;
  };
  auto pass_begin = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; minit[r] = 0.f; }
    m_run = -INFINITY;
    l_val = 0.f;
    t64 = wave;
    if (t64 < ntiles) load_tile(t64);
  };
  auto checked_tile = [&](bool first) {   // row sums checked against ROWSUM_LIMIT, running maximum moved when needed
    const int cur = advance();
    // ---- S^T: st[t][reg] = score(key = 64 cur + 32 t + (reg&3) + 8 (reg>>2) + 4 hh, query = ql), log2 domain ----
    f32x16 st[2];
    auto qk = [&](f32x16 (&d)[2], const f32x16& c) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ck[t][0], qf[0], c, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ck[t][ks], qf[ks], d[t], 0, 0, 0);
      }
    };
    softmax_step(qk, first, cur * 64 + 64 > kv_len, cur * 64 + 4 * hh, kv_len, st, minit, m_run, l_val, oacc);
    // ---- O^T += V^T . P^T ; k-step (t, s): accumulator regs 8s..8s+7 of st[t] are exactly the B fragment ----
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[t][8 * s + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cv[t][s][dt], pf, oacc[dt], 0, 0, 0);
      }
  };

  // Fast pass (see attn_fwd_lds_kernel below for the same structure): this wave's first step establishes m_run, its further
  // FULL steps run unchecked, a partial last step (the sequence's last tile) is checked again; the accumulators are
  // inspected once at the end and the wave repeats its steps in the checked form if anything overflowed.
  pass_begin();
  if (t64 < ntiles) checked_tile(true);
  const int nfull = (ntiles * 64 > kv_len) ? ntiles - 1 : ntiles;   // steps without a key mask
  while (t64 < nfull) {
    advance();
    auto kfrag = [&](int t, int ks) { return ck[t][ks]; };
    auto vfrag = [&](int t, int s, int dt) { return cv[t][s][dt]; };
    fast_step(kfrag, vfrag, qf, minit, l_val, oacc);
  }
  if (t64 < ntiles) checked_tile(false);
  if (__builtin_amdgcn_ballot_w64(!accum_finite(l_val, oacc)) != 0) {   // wave-uniform; waves are independent until the merge
    pass_begin();
    bool first = true;
    while (t64 < ntiles) { checked_tile(first); first = false; }
  }

  float l_run = add_xor32(l_val);   // the two key halves of a query meet once, here
  // ---- merge the NSPLIT partial results (same queries, disjoint keys) ----
  // red: per extra wave [8 quads][64 lanes] f32x4 (lane-contiguous: conflict-free ds_write/read_b128) + [64 lanes] (m, l)
  if (NSPLIT > 1) {
    if (wave > 0) {
      f32x4* dq = (f32x4*)(red + (wave - 1) * 64 * 34);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        dq[g * 64 + lane] = f32x4{oacc[0][4 * g], oacc[0][4 * g + 1], oacc[0][4 * g + 2], oacc[0][4 * g + 3]};
        dq[(4 + g) * 64 + lane] = f32x4{oacc[1][4 * g], oacc[1][4 * g + 1], oacc[1][4 * g + 2], oacc[1][4 * g + 3]};
      }
      *(f32x2*)(red + (wave - 1) * 64 * 34 + 64 * 32 + lane * 2) = f32x2{m_run, l_run};
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < NSPLIT; ++w) {
      const f32x4* sq = (const f32x4*)(red + (w - 1) * 64 * 34);
      const f32x2 ml = *(const f32x2*)(red + (w - 1) * 64 * 34 + 64 * 32 + lane * 2);
      const float m_new = fmaxf(m_run, ml[0]);
      // a wave that saw no tile has m = -inf, l = 0, O = 0: its factor is exp2(-inf) = 0 (m_new is finite because
      // wave 0 always owns tile 0 when kv_len > 0)
      const float fa = fast_exp2(m_run - m_new), fb = fast_exp2(ml[0] - m_new);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 s0 = sq[g * 64 + lane], s1 = sq[(4 + g) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          oacc[0][4 * g + r] = oacc[0][4 * g + r] * fa + s0[r] * fb;
          oacc[1][4 * g + r] = oacc[1][4 * g + r] * fa + s1[r] * fb;
        }
      }
      l_run = l_run * fa + ml[1] * fb;
      m_run = m_new;
    }
  }

  // ---- normalise + store: oacc[dt][reg] = O[q = ql][d = 32 dt + (reg&3) + 8 (reg>>2) + 4 hh] ----
  // From the accumulator layout a store instruction would write 8-byte pieces of 32 different rows.  The 32 x 64 bf16 tile
  // is turned row-major in LDS (144-byte rows; `red` is free: the merge reads above are this wave's own, in order) and
  // leaves as 4 instructions of 8 full 128-byte row segments each.
  {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    char* stg = (char*)red;
    if (NSPLIT > 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(bf16x4*)(stg + ql * 144 + 64 * dt + 16 * g + 8 * hh) =
            f2bf4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + (lane >> 3), c = lane & 7;
      const int q_row = qt * 32 + r;
      const uint4 v = *(const uint4*)(stg + r * 144 + c * 16);
      if (q_row < a.rows_per_seq)
        *(uint4*)(a.o + ((size_t)seq * a.rows_per_seq + q_row) * a.ldo + head * 64 + c * 8) = v;
    }
  }
}


// ---- LDS-shared variant for large problems ------------------------------------------------------------------------
// Same math and register layout as attn_fwd_kernel, but a workgroup = 4 waves = 128 queries of one (sequence, head) and
// every 64-key K/V tile is fetched ONCE per workgroup into LDS (3-stage LDS-DMA ring, counted vmcnt + raw barrier,
// as in gemm_bf16.hip) instead of once per wave from L2: 4x less L2->CU traffic, which is what bounds the LDS-free
// kernel once S*H*N^2 is large (C3: 7.4 GB of L2 reads per call).  The fragment-major K / V tiles are contiguous in
// memory (K: 2 x 4 KiB, V: 2 x 4 KiB per 64 keys), so the DMA copies 1 KiB fragments (lanes permuted, see `stage`) and a
// fragment read is ds_read_b128 at fragment*1 KiB + lane*16 (conflict-free; round 2 read the memory order, 2-way conflicts:
// SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE, profiles/r03_d_pmc_attention_c3_b.json).
// NST = ring depth (48 / 32 KiB of LDS -> 3 / 5 workgroups per CU by LDS), OCC = waves per SIMD the register allocation must
// leave room for (__launch_bounds__ second argument: 4 -> <= 128 VGPRs, 5 -> <= 96).  Shipped: <2, 4> (C3, us per launch: 3
// stages -> 3 workgroups per CU 324; 2 stages -> 4 per CU, VGPR-limited, 292; forcing 5 per CU with 3 spills 300; row sums
// on the matrix pipe through an all-ones A operand 293 vs 287 and the GEMMs behind it 1-2 % slower under the power cap).
template <int NST, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_fwd_lds_kernel(AttnArgs a) {
  constexpr int TILE_BYTES = 16384;  // K 8 KiB + V 8 KiB per 64 keys
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;

  const int qblocks = (a.rows_per_seq + 127) / 128;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int qb = bid % qblocks;
  bid /= qblocks;
  const int head = bid % a.H;
  const int seq = bid / a.H;
  const size_t sh = (size_t)seq * a.H + head;
  const int kv_len = a.kv_len ? min(a.kv_len[seq], a.rows_per_seq) : a.rows_per_seq;
  const int ntiles = (kv_len + 63) / 64;

  const bf16* Qg = a.q + sh * a.n_pad * 64;
  const char* Kg = (const char*)(a.k + sh * a.n_pad * 64);
  const char* Vg = (const char*)(a.v + sh * a.n_pad * 64);

  const int qt = min(qb * 4 + wave, a.n_pad / 32 - 1);  // this wave's 32-query tile (clamped: pad rows are zero)
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(Qg + ((size_t)(qt * 4 + ks) * 32 + ql) * 16 + hh * 8);

  // stage t64: K bytes [t64*8192, +8192) and V bytes [t64*8192, +8192); 1024 chunks of 16 B, 4 per thread
  // LDS image of a fragment: lane l's 16 bytes at byte 16 l (conflict-free ds_read_b128: the four 16-lane groups of the
  // instruction each cover all 64 banks), i.e. LDS chunk i holds memory chunk 2 (i & 31) + (i >> 5) of the fragment -- the
  // DMA permutes on the SOURCE side (its LDS side is fixed at base + 16 lane); a wave still reads 1 KiB contiguous.
  const int src16 = ((lane & 31) * 2 + (lane >> 5)) * 16;
  auto stage = [&](int buf, int t64) {
    char* dst = smem + buf * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int off = (wave * 64 + 256 * j) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kg + (size_t)t64 * 8192 + off + src16),
                                       (__attribute__((address_space(3))) void*)(dst + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vg + (size_t)t64 * 8192 + off + src16),
                                       (__attribute__((address_space(3))) void*)(dst + 8192 + off), 16, 0, 0);
    }
  };

  f32x16 oacc[2], minit;
  float m_run, l_val;

  // Ring bookkeeping of one 64-key step: wait until tile t64 has landed (NST - 2 younger tiles of 4 DMAs each may stay in
  // flight) and every wave is done reading tile t64 - 1, stage tile t64 + NST - 1, return the slot of tile t64.
  int buf = 0, nbuf = NST - 1;
  // tile_wait: tile t64 has landed and every wave is done with tile t64 - 1; tile_stage: fetch tile t64 + NST - 1 into the
  // slot that freed.  The fast loop issues its first K-fragment reads BETWEEN the two: a DMA instruction costs the wave
  // 100+ cycles of issue while the CU's address path is busy, and reads issued behind it wait that long (gemm_bf16.hip).
  auto tile_wait = [&](int t64) -> const char* {
    const int young = min(NST - 2, ntiles - 1 - t64);   // tiles staged after t64 that may stay in flight
    if (young >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (young == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return smem + buf * TILE_BYTES;
  };
  auto tile_stage = [&](int t64) {
    if (t64 + NST - 1 < ntiles) stage(nbuf, t64 + NST - 1);
    buf = (buf == NST - 1) ? 0 : buf + 1;
    nbuf = (nbuf == NST - 1) ? 0 : nbuf + 1;
  };
  auto tile_begin = [&](int t64) -> const char* {
    const char* Ks = tile_wait(t64);
    tile_stage(t64);
    return Ks;
  };
  auto pass_begin = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; minit[r] = 0.f; }
    m_run = -INFINITY;
    l_val = 0.f;
    buf = 0;
    nbuf = NST - 1;
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
      if (t < ntiles) stage(t, t);
  };
  // K fragment (half t, ks) / V fragment (half t, s16 = s, dt) of the tile at Ks: 1 KiB each, lane-linear
  auto checked_tile = [&](int t64) {   // row sums checked against ROWSUM_LIMIT, running maximum moved when needed
    const char* Ks = tile_begin(t64);
    const char* Vs = Ks + 8192;
    f32x16 st[2];
    // the K tile stays in its ring slot until the next step's barrier: the slow path may read it again
    auto qk = [&](f32x16 (&d)[2], const f32x16& c) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 kf = *(const bf16x8*)(Ks + (t * 4 + ks) * 1024 + lane * 16);
          d[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? c : d[t], 0, 0, 0);
        }
    };
    softmax_step(qk, t64 == 0, t64 * 64 + 64 > kv_len, t64 * 64 + 4 * hh, kv_len, st, minit, m_run, l_val, oacc);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[t][8 * s + j];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 vf = *(const bf16x8*)(Vs + (((t * 2 + s) * 2 + dt) * 1024) + lane * 16);
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
      }
  };

  // Fast pass: the first step establishes m_run (checked form), every further FULL step runs unchecked (fast_step: no
  // maximum, no comparison, no branch -- ONE code path in the loop body, so nothing is copied between register sets at a
  // join), a partial last step (key mask) takes the checked form again.
  pass_begin();
  if (ntiles > 0) checked_tile(0);
  const int nfull = (ntiles > 0 && ntiles * 64 > kv_len) ? ntiles - 1 : ntiles;   // steps without a key mask
  for (int t64 = 1; t64 < nfull; ++t64) {
    const char* Ks = tile_wait(t64);
    const char* Vs = Ks + 8192;
    bf16x8 kf0[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf0[ks] = *(const bf16x8*)(Ks + ks * 1024 + lane * 16);
    tile_stage(t64);
    auto kfrag = [&](int t, int ks) { return t == 0 ? kf0[ks] : *(const bf16x8*)(Ks + (t * 4 + ks) * 1024 + lane * 16); };
    auto vfrag = [&](int t, int s, int dt) { return *(const bf16x8*)(Vs + (((t * 2 + s) * 2 + dt) * 1024) + lane * 16); };
    fast_step(kfrag, vfrag, qf, minit, l_val, oacc);
  }
  if (nfull < ntiles && ntiles > 1) checked_tile(ntiles - 1);
  // Nothing was compared on the way: if ANY query of the workgroup overflowed (inf / NaN in l or O), every wave repeats
  // the pass in the checked form (the K / V ring is filled cooperatively).
  if (__syncthreads_or(!accum_finite(l_val, oacc))) {
    pass_begin();
    for (int t64 = 0; t64 < ntiles; ++t64) checked_tile(t64);
  }

  const float l_run = add_xor32(l_val);   // the two key halves of a query meet once, here
  const int q_row = (qb * 4 + wave) * 32 + ql;
  if (q_row < a.rows_per_seq) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    bf16* op = a.o + ((size_t)seq * a.rows_per_seq + q_row) * a.ldo + head * 64 + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *(bf16x4*)(op + dt * 32 + 8 * g) = f2bf4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv,
                                                   oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
      }
  }
}

}  // namespace

int f5e_flash_attn_pf(hipStream_t st, const void* q, const void* k, const void* v, void* o, int ldo, const int* kv_len,
                      int S, int H, int rows_per_seq, int n_pad, int splits, const F5ePrefetch* pf) {
  F5E_REQUIRE(q && k && v && o, "flash_attn: null pointer");
  F5E_REQUIRE(S > 0 && H > 0 && rows_per_seq > 0, "flash_attn: empty problem");
  F5E_REQUIRE(n_pad % 64 == 0 && n_pad >= rows_per_seq, "flash_attn: n_pad=%d must be a multiple of 64 and >= %d",
              n_pad, rows_per_seq);
  F5E_REQUIRE(ldo % 8 == 0 && ldo >= H * 64 && ((uintptr_t)o & 15) == 0, "flash_attn: bad ldo=%d / output alignment", ldo);
  AttnArgs a{};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o; a.ldo = ldo;
  a.kv_len = kv_len; a.S = S; a.H = H; a.rows_per_seq = rows_per_seq; a.n_pad = n_pad;
  const int grid = ((rows_per_seq + 31) / 32) * H * S;
  a.n_main = grid;
  // Large problems: K/V shared through LDS by 128-query workgroups (splits = -1 forces it, 0 picks it when the
  // LDS-free kernel would already have >= 8 waves per CU without any KV split)
  if (splits == -1 || (splits == 0 && grid >= 8192)) {
    const int g128 = ((rows_per_seq + 127) / 128) * H * S;
#ifdef F5E_TOOLS
    // tools build only (tools/attn_time.py): ring depth / occupancy variants for A/B timing on one box
    static const int variant = getenv("F5E_ATTN_VARIANT") ? atoi(getenv("F5E_ATTN_VARIANT")) : 0;
    if (variant == 33) {
      static F5eDeviceOnce once33;
      hipLaunchKernelGGL((attn_fwd_lds_kernel<3, 3>), dim3(g128), dim3(256), 3 * 16384, st, a);
    } else if (variant == 23) {
      hipLaunchKernelGGL((attn_fwd_lds_kernel<2, 3>), dim3(g128), dim3(256), 2 * 16384, st, a);
    } else if (variant == 42) {
      hipLaunchKernelGGL((attn_fwd_lds_kernel<4, 2>), dim3(g128), dim3(256), 4 * 16384, st, a);
    } else
#endif
    hipLaunchKernelGGL((attn_fwd_lds_kernel<2, 4>), dim3(g128), dim3(256), 2 * 16384, st, a);
    F5E_LAUNCH_CHECK("flash_attn_lds");
    return F5E_OK;
  }
  if (splits <= 0) {
    // The 4-way split (256-VGPR waves, two workgroups per CU) while its grid is ONE round: up to 2 workgroups per CU; past
    // that no split at all.  Round 4, us per launch in a 22-launch graph, S x H = 32 (tools/attn_splits.py,
    // profiles/r04_ai_attn_splits.txt): 512 q-tiles: 4 splits 7.8, none 8.2; 544: 10.6 / 8.5; 960 (N = 938): 16.4 / 13.1;
    // 1408 (N = 1390): 28.3 / 27.1; two splits never ahead of both.  (Rounds 1-3 split four ways up to 1024 q-tiles and two
    // ways up to 2048: tuned at C2's 480 and never re-measured at C4's 640-1400.)
    const int ktiles = (rows_per_seq + 63) / 64;
    splits = grid > 2 * f5e_cu_count() ? 1 : 4;
    while (splits > 1 && splits > ktiles) splits >>= 1;
  }
  switch (splits) {
    case 1: {
      const int npf = f5e_prefetch_wgs(pf);
      if (npf) a.pf = *pf;
      hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(grid + npf), dim3(64), 0, st, a);
      break;
    }
    case 2: {
      const int npf = f5e_prefetch_wgs(pf);
      if (npf) a.pf = *pf;
      hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3(grid + npf), dim3(128), 0, st, a);
      break;
    }
    case 4: {
      const int npf = f5e_prefetch_wgs(pf);
      if (npf) a.pf = *pf;
      hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(grid + npf), dim3(256), 0, st, a);
      break;
    }
    default: F5E_REQUIRE(false, "flash_attn: splits must be 0 (auto), -1 (LDS-shared), 1, 2 or 4");
  }
  F5E_LAUNCH_CHECK("flash_attn");
  return F5E_OK;
}

extern "C" int f5e_flash_attn(hipStream_t st, const void* q, const void* k, const void* v, void* o, int ldo,
                              const int* kv_len, int S, int H, int rows_per_seq, int n_pad, int splits) {
  return f5e_flash_attn_pf(st, q, k, v, o, ldo, kv_len, S, H, rows_per_seq, n_pad, splits, nullptr);
}
