// Large-M bf16 GEMM for the DiT block linears: 256 x 256 tile, K-step 64, 8 waves in two groups that run half a phase
// apart ("ping-pong"): while one group issues its ds_reads and LDS-DMAs the other one is inside its MFMA cluster, so
// the matrix pipe of every SIMD (one wave of each group per SIMD) stays busy across the workgroup barriers.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]   + the epilogues of gemm_bf16_epilogue.h
//
// Why: the 128x128 ring kernel of gemm_bf16.hip tops out at 650-740 TFLOP/s at M = 60k (PMC: matrix pipe 31-35 % busy,
// 45 % of wave cycles in s_waitcnt / s_barrier): every K-step the whole workgroup stops at one barrier, and a 128x128
// tile needs 1 byte of L2->LDS fill per 64 flops.  A 256x256 tile halves the fill per flop; the phase structure below
// (after the CDNA programming guide's "256^2 8-phase" description) takes the barrier off the critical path.
//
// Geometry.  wave = wr * 4 + wc owns rows [128 wr, +128) x columns [64 wc, +64) of the tile: acc[4][8] accumulator
// quads (operands swapped as in gemm_bf16.hip: a quad = 4 consecutive n of one row).  A K-tile (X 256 x 64, W 256 x 64
// bf16 = 64 KiB) is staged as FOUR half-tiles of 128 rows, grouped by WHEN the waves need them:
//     XHa: rows {0..63} of both row halves      (the "A0" operands, needed in phase 0 of the K-tile)
//     WH0: columns {0..31} of all four wc       (the "B0" operands, phase 0)
//     WH1: columns {32..63} of all four wc      (the "B1" operands, phase 1)
//     XHb: rows {64..127} of both row halves    (the "A1" operands, phase 2)
// LDS = 2 K-tile buffers x 4 half-tiles x 16 KiB = 128 KiB, rows of 128 B with the 16-B chunk swizzle c ^ ((row>>1)&7)
// applied to the DMA source address and to the ds_read_b128 (conflict-free 16-lane groups, as in gemm_bf16.hip).
//
// Schedule.  Global phase p = 4 kt + q computes quadrant q of K-tile kt: (A0,B0), (A0,B1), (A1,B1), (A1,B0): 16 MFMAs
// (64 x 32 outputs x K = 64).  Each phase: [ds_read this phase's new operands; stage half-tile number p + 5 (two
// global_load_lds per thread); s_waitcnt vmcnt(6)]  s_barrier  [lgkmcnt(0); 16 MFMAs at raised priority]  s_barrier.
// Half-tiles are staged in the order XHa, WH0, WH1, XHb of K-tile 0, 1, ...; number h is first read in phase
// h + (h & 3 ? 0 : 1) + ... = {4 kt, 4 kt, 4 kt + 1, 4 kt + 2}, i.e. 5, 4, 4, 4 phases after it was issued.
//   RAW: vmcnt(6) in phase w leaves only the three youngest half-tiles (issued in phases w-2 .. w) in flight, so
//        everything first read in phase w + 1 has landed for the issuing wave; the reader passes a barrier that follows
//        BOTH groups' waits before its phase-(w+1) reads (group 1 runs one barrier late, so the barrier that closes
//        group 1's wait is the one that opens group 0's next load section).
//   WAR: a slot is re-staged 8 - 5 = 3 (XHa) or 4 phases after the last ds_read of its previous content; with the
//        half-phase stagger the late group's reads complete 1.5 phases before the early group's re-stage is issued.
// The tail shrinks the vmcnt allowance as staging stops (4, 2, 0).
#include <cstdlib>
#include <type_traits>

#include "gemm_bf16_epilogue.h"

namespace {

using namespace f5e_gemm;

constexpr int HALF_BYTES = 128 * 128;  // 128 rows x 64 bf16
using I0 = std::integral_constant<int, 0>;
using I1 = std::integral_constant<int, 1>;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Epilogue of a tile.  A wave owns the 128 x 64 sub-tile at rows m0 + 128 wr, columns n0 + 64 wc, accumulators acc[4][8]
// in the swapped-operand layout, and 16 KiB of the (now idle) LDS ring at smem + wave * 16 KiB as staging space.
template <int EPI>
__device__ __forceinline__ void pp_epilogue(const GemmArgs& a, char* smem, f32x4 (&acc)[4][8], int wave, int lane, int wr,
                                            int wc, int m0, int n0) {
  const int fr = lane & 15, fq = lane >> 4;
  // ---- epilogue.  With one workgroup per CU nothing hides it, and the MFMA accumulator layout stores 8 B (bf16) or 16 B
  // (fp32) pieces of 16 different rows per instruction: the memory system sees partial lines (measured: 22 us per
  // workgroup for 128 KiB of bf16 output).  So the wave's 128 x 64 sub-tile goes through its 16 KiB share of the now
  // idle LDS ring and leaves as full 128-byte (bf16) / 256-byte (fp32) row segments.  All main-loop LDS reads are
  // complete here: the balancing barrier above is the last barrier instance of both groups.
  char* reg = smem + wave * 16384;
  const int mbase = m0 + wr * 128, nbase = n0 + wc * 64;
  if constexpr (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
    if (a.N % 8 == 0 && a.ldo % 8 == 0) {
      f32x4 bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = min(nbase + i * 16 + fq * 4, a.N - 4);
        bq[i] = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = acc[i][j] + bq[i];
          if (EPI == EPI_BF16_GELU) v = f32x4{gelu_tanh_f(v[0]), gelu_tanh_f(v[1]), gelu_tanh_f(v[2]), gelu_tanh_f(v[3])};
          // row r: 8 chunks of 16 B (8 bf16), chunk index XORed with r & 7
          *(bf16x4*)(reg + r * 128 + (((i * 2 + (fq >> 1)) ^ (r & 7)) << 4) + (fq & 1) * 8) = f2bf4(v[0], v[1], v[2], v[3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own region, own writes: no workgroup barrier needed
      const int c = lane & 7;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + (lane >> 3);
        const int m = mbase + r, n = nbase + c * 8;
        const uint4 v = *(const uint4*)(reg + r * 128 + ((c ^ (r & 7)) << 4));
        if (m < a.M && n < a.N) *(uint4*)((bf16*)a.out + (size_t)m * a.ldo + n) = v;
      }
      return;
    }
  }
  if constexpr (EPI == EPI_GATE_RES) {
    // Lean read-modify-write for the common wave tile: all 128 rows inside [0, M) and inside their sequences' lengths, one
    // gate row.  Sequence, position, lengths and gate row are per-wave scalars (the wave's rows touch at most two sequences
    // when rows_per_seq >= 128), so the per-row division / modulo / loop / predicate of the general path below (~3000
    // instructions per wave; 16 us per tile at C3 = a third of the out-projection) disappears, and all 16 residual loads of
    // a 64-row half are in flight together: two memory round trips per tile instead of four.  (Both halves at once = 128
    // VGPRs of loads beside the 64 live accumulators: spills, 55 us per tile; half 1's loads issued row by row behind half
    // 0's stores: no gain, the tile's 512 KiB of read-modify-write are then at the memory system's rate.)  The LDS pass itself stays: from the
    // accumulator layout a store instruction would touch 16 rows x 64 B, which the memory system serves at half the rate of
    // whole 256-byte row segments (measured: 24 us per tile without the LDS pass).
    if (a.N % 4 == 0 && a.rows_per_seq >= 128 && nbase + 64 <= a.N && mbase + 128 <= a.M) {
      const int seq = div_magic(mbase, a.rows_per_seq, a.rps_magic);
      const int pos0 = mbase - seq * a.rows_per_seq;
      const int rb = min(128, a.rows_per_seq - pos0);  // rows [0, rb): sequence seq, [rb, 128): sequence seq + 1
      const int len_a = a.seq_len ? load_uniform_i32(a.seq_len + seq) : a.rows_per_seq;
      const int len_b = (rb < 128 && a.seq_len) ? load_uniform_i32(a.seq_len + seq + 1) : a.rows_per_seq;
      const int grow_a = seq % a.gate_rows, grow_b = (seq + 1) % a.gate_rows;
      if (len_a - pos0 >= rb && len_b >= 128 - rb && (rb == 128 || grow_b == grow_a)) {  // uniform
        const size_t eoff = a.eval_ptr ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
        const int c = lane & 15, rq = lane >> 4;
        const f32x4 ga = *(const f32x4*)(a.gate + eoff + (size_t)grow_a * a.gate_stride + nbase + c * 4);
        const f32x4 bc = a.bias ? *(const f32x4*)(a.bias + nbase + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        // uniform base + 32-bit byte offset per row (a 128-row tile of fp32 spans < 2^31 bytes).  The row stride is made
        // opaque per tile: otherwise the row offsets are hoisted out of the persistent tile loop as loop invariants, live
        // (and spilled) across the K loop
        int ldr = a.ldr;
        asm volatile("" : "+s"(ldr));
        char* const xbase = (char*)(a.resid + (size_t)mbase * ldr + nbase);
        const unsigned rstride = (unsigned)ldr * 4u, off0 = (unsigned)rq * rstride + (unsigned)c * 16u;
        // LDS: 64 rows x 256 B, the 16-byte chunk index XORed with row & 15.  Read side: row = 4 b + rq, so the address is
        // (rq * 256 + ((c ^ rq) << 4)) ^ ((b & 3) << 6) plus the constant 1024 b: four distinct registers
        const int rd0 = rq * 256 + ((c ^ rq) << 4);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int r = jj * 16 + fr;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              *(f32x4*)(reg + r * 256 + (((i * 4 + fq) ^ (r & 15)) << 4)) = acc[i][half * 4 + jj];
          }
          f32x4 xv[16];
#pragma unroll
          for (int b = 0; b < 16; ++b) xv[b] = *(const f32x4*)(xbase + (off0 + (unsigned)(half * 64 + b * 4) * rstride));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int b = 0; b < 16; ++b) {
            const f32x4 v = *(const f32x4*)(reg + ((rd0 ^ ((b & 3) << 6)) + b * 1024));
            const f32x4 xn = xv[b] + ga * (v + bc);
            *(f32x4*)(xbase + (off0 + (unsigned)(half * 64 + b * 4) * rstride)) = xn;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the region is overwritten
        }
        return;
      }
    }
  }
  if constexpr (EPI == EPI_GATE_RES) {
    if (a.N % 4 == 0) {
      const size_t eoff = a.eval_ptr ? (size_t)load_uniform_i32(a.eval_ptr) * a.eval_stride : 0;
      f32x4 bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = min(nbase + i * 16 + fq * 4, a.N - 4);
        bq[i] = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int c = lane & 15;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // 64 rows x 256 B: 16 chunks of 16 B (4 fp32) per row, chunk index XORed with r & 15
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int r = jj * 16 + fr;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *(f32x4*)(reg + r * 256 + (((i * 4 + fq) ^ (r & 15)) << 4)) = acc[i][half * 4 + jj] + bq[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int mb = mbase + half * 64;
        int seq0 = mb / a.rows_per_seq;
        int pos0 = mb - seq0 * a.rows_per_seq;
        // 8 rows per lane at a time: every load of a batch (x, gate row, sequence length) is issued from a clamped
        // address before the first use, so a batch costs one memory round trip instead of two per row
#pragma unroll
        for (int b8 = 0; b8 < 2; ++b8) {
          f32x4 xv[8], gv[8];
          int len[8], ps[8];
          bool inb[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = (b8 * 8 + u) * 4 + (lane >> 4);
            const int m = mb + r, n = nbase + c * 4;
            int seq = seq0, pos = pos0 + r;
            while (pos >= a.rows_per_seq) { pos -= a.rows_per_seq; ++seq; }
            inb[u] = m < a.M && n < a.N;
            ps[u] = pos;
            const int mc = min(m, a.M - 1), nc = min(n, a.N - 4);
            const int sc = inb[u] ? seq : mc / a.rows_per_seq;
            len[u] = a.seq_len ? a.seq_len[sc] : a.rows_per_seq;
            gv[u] = *(const f32x4*)(a.gate + eoff + (size_t)(sc % a.gate_rows) * a.gate_stride + nc);
            xv[u] = *(const f32x4*)(a.resid + (size_t)mc * a.ldr + nc);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = (b8 * 8 + u) * 4 + (lane >> 4);
            const f32x4 v = *(const f32x4*)(reg + r * 256 + ((c ^ (r & 15)) << 4));
            if (inb[u] && ps[u] < len[u])
              *(f32x4*)(a.resid + (size_t)(mb + r) * a.ldr + nbase + c * 4) = xv[u] + gv[u] * v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next half overwrites the region
      }
      return;
    }
  }
  if constexpr (EPI == EPI_QKV_ROPE) {
    // V goes out transposed inside 16-key groups ([group][d / 32][d % 32][h][j], attention.hip): from the accumulator
    // layout that is four 2-byte stores per quad (61 M scattered 2-byte stores per launch at C3).  A wave with V columns
    // (one head: 64 d) parks its 128 x 64 sub-tile in LDS and writes 16-byte pieces = 8 keys of one d instead; whole
    // groups leave as 1 KiB contiguous per instruction.  Positions restart at every sequence, so the wave's rows split
    // into at most two segments (rows_per_seq >= 128) whose first / last group may be partial: those fall back to
    // element stores of the valid keys only (the other keys of the group belong to a neighbouring wave).
    const int inner = a.heads * 64;
    if (nbase >= a.N) return;  // a wave whose columns lie past 3 * inner (heads not a multiple of 4): nothing to store
    if (nbase >= 2 * inner && a.rows_per_seq >= 128) {
      const int head = (nbase - 2 * inner) >> 6;
      f32x4 bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[i] = a.bias ? *(const f32x4*)(a.bias + nbase + i * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = acc[i][j] + bq[i];
          *(bf16x4*)(reg + r * 128 + (((i * 2 + (fq >> 1)) ^ (r & 7)) << 4) + (fq & 1) * 8) = f2bf4(v[0], v[1], v[2], v[3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int rows_valid = min(128, a.M - mbase);
      const int seq0 = mbase / a.rows_per_seq, pos0 = mbase - seq0 * a.rows_per_seq;
      const int rb = min(128, a.rows_per_seq - pos0);  // rows [0, rb) are sequence seq0, [rb, 128) sequence seq0 + 1
      // A 16-byte piece = 8 keys of one d = two hardware-transposed LDS reads (ds_read_b64_tr_b16: the 16 lanes of a group
      // supply the addresses of 4 rows x 16 columns and lane i gets column i of the 4 rows) instead of eight 2-byte reads:
      // group g4 = lane >> 4 takes d block g4 & 1 and key half hk = g4 >> 1, lane i of the group the d column i.  The 64
      // lanes of a store instruction still cover one contiguous 1 KiB ([32 d][2 h][8 keys]).  Every lane supplies a valid
      // address (rows clamped into the segment): the instruction needs EXEC all ones, the branches around it are uniform.
      using s16x4 = short __attribute__((ext_vector_type(4)));
      const int g4 = lane >> 4, i16 = lane & 15, d16 = g4 & 1, hk = g4 >> 1, tq = (lane >> 2) & 3, tp = lane & 3;
      for (int seg = 0; seg < 2; ++seg) {
        const int rstart = seg == 0 ? 0 : rb;
        const int rcount = min(seg == 0 ? rb : 128 - rb, rows_valid - rstart);
        if (rcount <= 0) break;
        const int pstart = seg == 0 ? pos0 : 0;
        bf16* vb = a.vt + ((size_t)(seq0 + seg) * a.heads + head) * a.n_pad * 64;
#pragma unroll 3
        for (int G = pstart >> 4; G <= (pstart + rcount - 1) >> 4; ++G) {
          const bool whole = G * 16 >= pstart && G * 16 + 15 < pstart + rcount;
          const int rel = rstart + G * 16 - pstart + 4 * hk + tq;   // LDS row of key 16 G + 4 hk + tq
          const int ra = min(max(rel, rstart), rstart + rcount - 1), rb8 = min(max(rel + 8, rstart), rstart + rcount - 1);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int d0 = dt * 32 + d16 * 16 + 4 * tp;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(reg + ra * 128 + (((d0 >> 3) ^ (ra & 7)) << 4) + (d0 & 7) * 2));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(reg + rb8 * 128 + (((d0 >> 3) ^ (rb8 & 7)) << 4) + (d0 & 7) * 2));
            bf16* dst = vb + ((size_t)(G * 2 + dt) * 32 + d16 * 16 + i16) * 16 + hk * 8;
            if (whole) {
              *(uint4*)dst = make_uint4(((const unsigned*)&lo)[0], ((const unsigned*)&lo)[1], ((const unsigned*)&hi)[0], ((const unsigned*)&hi)[1]);
            } else {
              // first / last group of a segment: the other keys belong to a neighbouring wave.  Each half of the piece is 4
              // consecutive keys: whole halves leave as 8 bytes, aligned pairs as 4 (sequence starts are even whenever
              // rows_per_seq is), single keys as 2
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                const s16x4 v4 = hf ? hi : lo;
                const int k0 = G * 16 + 8 * hf + 4 * hk;
                short* d4 = (short*)dst + 4 * hf;
                const bool o0 = k0 >= pstart && k0 < pstart + rcount, o1 = k0 + 1 >= pstart && k0 + 1 < pstart + rcount;
                const bool o2 = k0 + 2 >= pstart && k0 + 2 < pstart + rcount, o3 = k0 + 3 >= pstart && k0 + 3 < pstart + rcount;
                if (o0 && o3) {
                  *(s16x4*)d4 = v4;
                } else {
                  if (o0 && o1) *(unsigned*)d4 = ((const unsigned*)&v4)[0];
                  else { if (o0) d4[0] = v4[0]; if (o1) d4[1] = v4[1]; }
                  if (o2 && o3) *(unsigned*)(d4 + 2) = ((const unsigned*)&v4)[1];
                  else { if (o2) d4[2] = v4[2]; if (o3) d4[3] = v4[3]; }
                }
              }
            }
          }
        }
      }
      return;
    }
  }
  if constexpr (EPI == EPI_QKV_ROPE) {
    // q / k waves (one head each): bias, optional qk RMSNorm and RoPE in registers, then out through LDS (below).
    const int inner = a.heads * 64;
    if (nbase < 2 * inner) {
      const int which = nbase >= inner ? 1 : 0;
      const int head = (nbase - which * inner) >> 6;
      bf16* base = (which == 0 ? a.q : a.k) + (size_t)head * a.n_pad * 64;
      const bool rope_on = head < a.rope_heads;
      const float* qk_w = a.qn_w ? (which == 0 ? a.qn_w : a.kn_w) : nullptr;
      const float qsc = which == 0 ? a.q_scale : 1.0f;
      f32x4 bq[4], wq[4];
      int csoff[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = i * 16 + fq * 4;
        bq[i] = a.bias ? *(const f32x4*)(a.bias + nbase + d) : f32x4{0.f, 0.f, 0.f, 0.f};
        wq[i] = qk_w ? *(const f32x4*)(qk_w + d) : f32x4{1.f, 1.f, 1.f, 1.f};
        csoff[i] = d;  // (d >> 1) * 2 floats
      }
      const int seq0 = mbase / a.rows_per_seq, pos0 = mbase - seq0 * a.rows_per_seq;
      const size_t seq_stride = (size_t)a.heads * a.n_pad * 64;
      // The fragment-major image of one (position, 16-d block) is 32 contiguous bytes and consecutive positions of a
      // 32-position tile follow each other, but the accumulator layout hands a lane 8 bytes of 16 different rows: 61 M
      // 8-byte stores per launch at C3, each instruction touching 16 partial lines.  So the wave's RoPE'd 128 x 64
      // sub-tile is parked in its 16 KiB share of the idle ring as [ks = d / 16][row][32 B] and leaves as 16-byte pieces,
      // 32 consecutive rows (1 KiB, contiguous inside a position tile) per store instruction.  The four 8-byte slots of a
      // row chunk are XORed with (row >> 2) & 3 so the 16 lanes of a ds_write_b64 group hit 16 different bank pairs.
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = j * 16 + fr;
        int pos = pos0 + r;
        while (pos >= a.rows_per_seq) pos -= a.rows_per_seq;
        const float* csrow = a.cos_sin + (size_t)min(pos, a.rows_per_seq - 1) * 64;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[i][j] + bq[i];
        if (qk_w) {  // RMSNorm over the head's 64 values: 16 in this lane, 48 in the lanes fr + 16 / 32 / 48
          float ss = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
          ss = add_xor32(add_xor16(ss));
          const float rn = rsqrtf(ss * (1.0f / 64.0f) + a.qk_eps);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] * rn * wq[i];
        }
        const int xr = (r >> 2) & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 o = v[i];
          if (rope_on) {
            const f32x4 cs = *(const f32x4*)(csrow + csoff[i]);
            o[0] = v[i][0] * cs[0] - v[i][1] * cs[1];
            o[1] = v[i][1] * cs[0] + v[i][0] * cs[1];
            o[2] = v[i][2] * cs[2] - v[i][3] * cs[3];
            o[3] = v[i][3] * cs[2] + v[i][2] * cs[3];
          }
          o *= qsc;   // q only: softmax scale and log2(e) (attention.hip); 1 for k
          *(bf16x4*)(reg + i * 4096 + r * 32 + ((fq ^ xr) << 3)) = f2bf4(o[0], o[1], o[2], o[3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own region, own writes: no workgroup barrier needed
      const int hh = lane & 1, rl = lane >> 1;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int ks = it >> 2, r = (it & 3) * 32 + rl;
        const int xr = (r >> 2) & 3;
        uint4 piece = *(const uint4*)(reg + ks * 4096 + r * 32 + ((hh ^ (xr >> 1)) << 4));
        if (xr & 1) piece = make_uint4(piece.z, piece.w, piece.x, piece.y);  // the two 8-byte slots of the piece were swapped
        int seq = seq0, pos = pos0 + r;
        while (pos >= a.rows_per_seq) { pos -= a.rows_per_seq; ++seq; }
        if (mbase + r < a.M)
          *(uint4*)(base + seq * seq_stride + (size_t)(pos >> 5) * 2048 + ks * 512 + (pos & 31) * 16 + hh * 8) = piece;
      }
      return;
    }
  }
  gemm_epilogue<EPI, 8, 4>(a, acc, mbase, nbase, lane);
}

// TRACE (tools build only, tools/pp_timeline.py): wall-clock stamps (100 MHz) per workgroup and tile: entry, first K-tile
// landed (includes the previous tile's store acknowledgements: vmcnt is in order), K loop done, epilogue issued
template <int EPI, int TRACE>
__device__ __forceinline__ void pp_tile(const GemmArgs& a, char* smem, int phys_id, int n_tiles, int iter = 0) {
  const int tid = threadIdx.x;
  auto stamp = [&](int k) {
    if (TRACE && tid == 0 && a.trace && iter < 16)
      a.trace[((size_t)blockIdx.x * 16 + iter) * 4 + k] = __builtin_amdgcn_s_memrealtime();
    // shader-clock copies of stamps 1 and 2 behind the table: effective core clock inside the K loop
    if (TRACE && tid == 0 && a.trace && iter < 16 && (k == 1 || k == 2))
      a.trace[(size_t)gridDim.x * 64 + ((size_t)blockIdx.x * 16 + iter) * 2 + (k - 1)] = __builtin_readcyclecounter();
  };
  stamp(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware tile order (same bijection as gemm_bf16.hip): blocks sharing blockIdx % 8 walk neighbouring tiles
  int bid = phys_id;
  {
    const int nblk = n_tiles;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  int tile_m, tile_n;
  if (a.m_major && a.pp_ngroup > 0) {
    // Column groups: all row panels for n-tiles [g G, (g + 1) G), then the next group.  With plain m-major order the 32 CUs
    // of an XCD work on ~2.7 row panels x ALL n-tiles at once, i.e. on the whole weight matrix (QKV: 6 MB against 4 MB of
    // L2): the weights stream from the Infinity Cache once per round.  Grouped, an XCD's 32 concurrent tiles are 8 row
    // panels x G n-tiles: G x 512 KiB of weights stay L2-resident and an activation panel is read once per group.
    const int G = a.pp_ngroup, per_group = a.tiles_m * G;
    const int g = bid / per_group, r = bid - g * per_group;
    const int gw = min(G, a.tiles_n - g * G);   // the last group may be narrower
    tile_m = r / gw;
    tile_n = g * G + (r - tile_m * gw);
  } else if (a.m_major) {
    tile_m = bid / a.tiles_n;
    tile_n = bid - tile_m * a.tiles_n;
  } else {
    tile_n = bid / a.tiles_m;
    tile_m = bid - tile_n * a.tiles_m;
  }
  const int m0 = tile_m * 256, n0 = tile_n * 256;

  // DMA sources: thread tid moves chunks i = tid and tid + 512 of every half-tile; slot row r = i >> 3 holds, at
  // physical chunk i & 7, the logical chunk (i & 7) ^ ((r >> 1) & 7) of the source row
  const bf16* src[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = tid + 512 * j, r = i >> 3, c = (i & 7) ^ ((r >> 1) & 7);
    const int xa = m0 + (r >> 6) * 128 + (r & 63);
    const int wa = n0 + (r >> 5) * 64 + (r & 31);
    src[0][j] = a.A + (size_t)min(xa, a.M - 1) * a.lda + c * 8;       // XHa
    src[1][j] = a.W + (size_t)min(wa, a.N - 1) * a.ldw + c * 8;       // WH0
    src[2][j] = a.W + (size_t)min(wa + 32, a.N - 1) * a.ldw + c * 8;  // WH1
    src[3][j] = a.A + (size_t)min(xa + 64, a.M - 1) * a.lda + c * 8;  // XHb
  }
  const int KT = a.K / 64, TI = 4 * KT;
  auto stage = [&](int idx) {
    const int kt = idx >> 2, type = idx & 3;
    char* dst = smem + (((kt & 1) << 2) + type) * HALF_BYTES + wave * 1024;
    const bf16* s0 = type == 0 ? src[0][0] : (type == 1 ? src[1][0] : (type == 2 ? src[2][0] : src[3][0]));
    const bf16* s1 = type == 0 ? src[0][1] : (type == 1 ? src[1][1] : (type == 2 ? src[2][1] : src[3][1]));
    glds16(s0 + kt * 64, dst);
    glds16(s1 + kt * 64, dst + 512 * 16);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a slot: row * 128 + ((c ^ sw) << 4), c = kk * 4 + fq, sw = (fr >> 1) & 7
  const int sw = (fr >> 1) & 7;
  const int xoff = (wr * 64 + fr) * 128, woff = (wc * 32 + fr) * 128;
  bf16x8 xf[2][4], wf0[2][2], wf1[2][2];
  auto read_x = [&](const char* slot) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xf[kk][j] = *(const bf16x8*)(slot + xoff + j * 16 * 128 + (((kk * 4 + fq) ^ sw) << 4));
  };
  auto read_w = [&](const char* slot, bf16x8 (&wf)[2][2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        wf[kk][i] = *(const bf16x8*)(slot + woff + i * 16 * 128 + (((kk * 4 + fq) ^ sw) << 4));
  };

  // prologue: half-tiles 0..4 (K-tile 0 complete + XHa of K-tile 1); the first two must have landed before phase 0
#pragma unroll
  for (int h = 0; h < 5; ++h) stage(h);   // dispatch guarantees KT >= 2
  wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  stamp(1);
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier (half a phase) behind group 0

  auto phase_tail = [&](int p) {  // stage half-tile p + 5, then leave only the 3 youngest half-tiles in flight
    const int idx = p + 5;
    if (idx < TI) stage(idx);
    const int rem = TI - 1 - idx;  // half-tiles still to be staged after this phase
    if (rem >= 0) wait_vm<6>();
    else if (rem == -1) wait_vm<4>();
    else if (rem == -2) wait_vm<2>();
    else wait_vm<0>();
  };
  auto mfma_quadrant = [&](auto qa_c, auto qb_c, bf16x8 (&wf)[2][2]) {
    constexpr int qa = decltype(qa_c)::value, qb = decltype(qb_c)::value;  // compile-time accumulator indices
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[qb * 2 + i][qa * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], xf[kk][j], acc[qb * 2 + i][qa * 4 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
  };

  for (int kt = 0; kt < KT; ++kt) {
    const char* buf = smem + ((kt & 1) << 2) * HALF_BYTES;
    const int p = kt * 4;
    // phase 0: (A0, B0)
    read_x(buf);
    read_w(buf + HALF_BYTES, wf0);
    phase_tail(p);
    mfma_quadrant(I0{}, I0{}, wf0);
    // phase 1: (A0, B1)
    read_w(buf + 2 * HALF_BYTES, wf1);
    phase_tail(p + 1);
    mfma_quadrant(I0{}, I1{}, wf1);
    // phase 2: (A1, B1)
    read_x(buf + 3 * HALF_BYTES);
    phase_tail(p + 2);
    mfma_quadrant(I1{}, I1{}, wf1);
    // phase 3: (A1, B0), operands already in registers
    phase_tail(p + 3);
    mfma_quadrant(I1{}, I0{}, wf0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier
  stamp(2);

  pp_epilogue<EPI>(a, smem, acc, wave, lane, wr, wc, m0, n0);
  stamp(3);
}

// Persistent launch: one workgroup per CU walks tiles phys_id = blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of
// 8, so a workgroup keeps its XCD class and the XCD-aware order of pp_tile is unchanged).  Two reasons.  (1) With 128 KiB
// of LDS only one workgroup fits a CU, so nothing overlaps a workgroup's exit (store acknowledgement) and its
// successor's launch + cold prologue; in a loop the next tile's prologue DMAs are issued right behind the stores.
// (2) All CUs run the same tile schedule in lock step: 256 CUs finish their K loops together and store 32 MiB in one
// burst, each waiting for the whole burst to drain (the stores cost 63 of 310 us at C3's FF1, the HBM being idle the
// rest of the time).  Workgroups that own one tile fewer than the busiest ones (n_tiles % gridDim.x != 0) have a tile time of
// slack: they start late by an even share of it, which takes the CUs out of phase at no cost in makespan.
template <int EPI, int TRACE = 0>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n_tiles = a.tiles_m * a.tiles_n, grid = gridDim.x;
  const int extra = n_tiles % grid;  // workgroups [0, extra) own one more tile
  if (extra > 0 && (int)blockIdx.x >= extra) {
    // ~5000 cycles per K-tile measured; s_sleep 127 = 8128 cycles
    const long long slack = (long long)(a.K / 64) * 5000;
    const int naps = (int)(slack * ((int)blockIdx.x - extra) / (grid - extra) / 8128);
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
  }
  int iter = 0;
  for (int p = blockIdx.x; p < n_tiles; p += grid, ++iter) {
    pp_tile<EPI, TRACE>(a, smem, p, n_tiles, iter);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // epilogue LDS reads done before the next prologue's DMAs land
    __builtin_amdgcn_s_barrier();
  }
}

template <int EPI, int TRACE = 0>
int launch_pp_t(GemmArgs& a, hipStream_t st) {
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + 255) / 256;
  a.m_major = a.M > a.N;
  a.rps_magic = div_magic_of(a.rows_per_seq);
  constexpr int lds = 8 * HALF_BYTES;
  static F5eDeviceOnce lds_once;  // 128 KiB of dynamic LDS needs the opt-in attribute, per device (host-only call)
  F5E_OPT_IN_LDS(lds_once, (gemm_bf16_pp_kernel<EPI, TRACE>), lds);
  int n_cu = f5e_cu_count() / 8 * 8;
  if (n_cu == 0) n_cu = 8;
  // column groups of 4 n-tiles (for QKV exactly the q, k and v blocks): see pp_tile; 2 / 3 / 6 measured no better
  a.pp_ngroup = a.tiles_n > 4 ? 4 : 0;
  const int n_tiles = a.tiles_m * a.tiles_n;
  // persistent: one workgroup per CU (a multiple of 8, so a workgroup keeps its XCD class across its tiles)
  const int grid = n_tiles < n_cu ? (n_tiles + 7) / 8 * 8 : n_cu;
  hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI, TRACE>), dim3(grid), dim3(512), lds, st, a);
  F5E_LAUNCH_CHECK("gemm_bf16_pp");
  return F5E_OK;
}

}  // namespace

namespace f5e_gemm {

int launch_pp(int epi, GemmArgs& a, hipStream_t st) {
  F5E_REQUIRE(a.K % 64 == 0 && a.K >= 128, "gemm_bf16_pp: K=%d must be a multiple of 64 and >= 128", a.K);
  F5E_REQUIRE(!a.ln_stats && !a.stats_out, "gemm_bf16_pp: the fused AdaLN runs on the 64x64 tile family");
#ifdef F5E_TOOLS
  // tools build only (tools/pp_timeline.py): per-tile timestamps into the buffer named by F5E_PP_TRACE
  if (const char* tr = getenv("F5E_PP_TRACE")) {
    a.trace = (unsigned long long*)strtoull(tr, nullptr, 0);
    switch (epi) {
      case EPI_GATE_RES: return launch_pp_t<EPI_GATE_RES, 1>(a, st);
      case EPI_QKV_ROPE: return launch_pp_t<EPI_QKV_ROPE, 1>(a, st);
      default: return launch_pp_t<EPI_BF16_GELU, 1>(a, st);
    }
  }
#endif
  switch (epi) {
    case EPI_BF16: return launch_pp_t<EPI_BF16>(a, st);
    case EPI_BF16_GELU: return launch_pp_t<EPI_BF16_GELU>(a, st);
    case EPI_GATE_RES: return launch_pp_t<EPI_GATE_RES>(a, st);
    case EPI_QKV_ROPE: return launch_pp_t<EPI_QKV_ROPE>(a, st);
    case EPI_F32: return launch_pp_t<EPI_F32>(a, st);
  }
  f5e_set_error("gemm_bf16_pp: unknown epilogue %d", epi);
  return F5E_ERR_BAD_SHAPE;
}

}  // namespace f5e_gemm
