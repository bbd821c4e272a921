// Flash-style non-causal self-attention for the DiT blocks (K10 of SURVEY 2.3), head dim 64, bf16 in / bf16 out.
//
//   O[s, q, h*64 + d] = sum_key softmax_key(Q[s,h,q,:] . K[s,h,key,:] / 8 + keymask) * V[s,h,key,d]
//
// Replaces F.scaled_dot_product_attention at reference model/modules.py:482-492 (key-padding mask only, no dropout).
// Instead of a dense bool mask the kernel takes per-sequence key counts: keys >= kv_len[s] get -inf.
//
// Layouts (written by f5e_gemm_bf16_qkv_rope):
//   Q, K : [S][H][n_pad][64] bf16   (RoPE already applied)
//   Vt   : [S][H][64][n_pad] bf16   (V transposed: key-contiguous, so the PV "A" fragments are plain 8-B LDS reads)
//   O    : [S*rows_per_seq][H*64] bf16 (token-major, ready to be the A operand of the out-projection GEMM)
//
// gfx950 design
//   * one workgroup = NW waves x 32 query rows; every wave owns 32 queries for the whole key sweep.
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): a lane owns ONE query (lane&31) and 32 of the
//     64 keys of a tile, so max/sum are 31 in-lane ops + one half-wave exchange; O^T = V^T . P^T keeps that
//     query-per-lane ownership, so the online-softmax rescale is lane-local too.
//   * P never leaves registers: the S^T accumulator is converted to bf16 and used directly as the MFMA B operand
//     (cdna guide section 3, "An accumulator tile as the next MFMA's operand"); the permuted k order
//     (key = 16s + 8(j>>2) + 4h + (j&3)) is matched by the V^T fragment addresses.
//   * K tile [64 keys][64 d] lives in LDS with the 16-B chunk XOR swizzle (chunk ^ ((row>>1)&7)): conflict-free
//     ds_read_b128; V^T tile [64 d][64 keys] uses 136-B rows: conflict-free ds_read_b64.  Register-staged,
//     double-buffered (loads for tile t+1 are issued before tile t's MFMAs, written after them): 1 barrier per tile.
#include "f5e_common.h"

namespace {

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* vt;
  bf16* o; int ldo;
  const int* kv_len;  // [S] or null (= rows_per_seq)
  int S, H, rows_per_seq, n_pad;
  float scale_log2e;  // (1/sqrt(64)) * log2(e)
};

constexpr int KT = 64;         // keys per tile
constexpr int VT_ROW = 136;    // bytes per V^T LDS row (64 keys * 2 B + 8 B pad)
constexpr int K_BYTES = KT * 128;
constexpr int V_BYTES = 64 * VT_ROW;
constexpr int STAGE_BYTES = K_BYTES + V_BYTES;

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(AttnArgs a) {
  constexpr int NT = NW * 64;
  constexpr int CH = 512 / NT;  // 16-B chunks per thread per operand tile
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;

  const int qtiles = (a.rows_per_seq + NW * 32 - 1) / (NW * 32);
  int bid = blockIdx.x;
  const int qt = bid % qtiles;
  bid /= qtiles;
  const int head = bid % a.H;
  const int seq = bid / a.H;
  const size_t sh = (size_t)seq * a.H + head;
  const int kv_len = a.kv_len ? min(a.kv_len[seq], a.rows_per_seq) : a.rows_per_seq;
  const int ntiles = (kv_len + KT - 1) / KT;

  const bf16* Kg = a.k + sh * a.n_pad * 64;
  const bf16* Vg = a.vt + sh * 64 * a.n_pad;

  // ---- Q fragments: B operand of S^T = K.Q^T : lane supplies Q[q][16ks + 8hh .. +8] ----
  const int q_row = qt * NW * 32 + wave * 32 + ql;
  const int q_row_c = min(q_row, a.n_pad - 1);
  bf16x8 qf[4];
  {
    const bf16* qp = a.q + (sh * a.n_pad + q_row_c) * 64 + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- staging plan ----
  // K tile: chunk i -> (row = i>>3 key, c = i&7); V^T tile: chunk i -> (row = i>>3 d, c = i&7: keys 8c..8c+7)
  uint4 kreg[CH], vreg[CH];
  auto load_tile = [&](int t) {
    const int key0 = t * KT;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = tid + NT * j;
      const int row = i >> 3, c = i & 7;
      kreg[j] = *(const uint4*)(Kg + (size_t)(key0 + row) * 64 + c * 8);
      vreg[j] = *(const uint4*)(Vg + (size_t)row * a.n_pad + key0 + c * 8);
    }
  };
  auto store_tile = [&](int buf) {
    char* Ks = smem + buf * STAGE_BYTES;
    char* Vs = Ks + K_BYTES;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int i = tid + NT * j;
      const int row = i >> 3, c = i & 7;
      *(uint4*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kreg[j];
      uint2* vp = (uint2*)(Vs + row * VT_ROW + c * 16);
      vp[0] = make_uint2(vreg[j].x, vreg[j].y);
      vp[1] = make_uint2(vreg[j].z, vreg[j].w);
    }
  };

  f32x16 oacc[2];
  oacc[0] = f32x16{0.f};
  oacc[1] = f32x16{0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  if (ntiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(t + 1);
    const char* Ks = smem + buf * STAGE_BYTES;
    const char* Vs = Ks + K_BYTES;

    // ---- S^T tiles: st[kt2][reg] = score(key = 32 kt2 + (reg&3) + 8(reg>>2) + 4hh, query = ql) ----
    f32x16 st[2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt2][r] = 0.f;
      const int row = kt2 * 32 + ql;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks * 2 + hh;
        const bf16x8 kf = *(const bf16x8*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
        st[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kt2], 0, 0, 0);
      }
    }

    // ---- online softmax (log2 domain) ----
    const int key_base = t * KT + 4 * hh;
    const bool partial = (t * KT + KT > kv_len);
    float mx = -INFINITY;
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = st[kt2][r] * a.scale_log2e;
        if (partial) {
          const int key = key_base + kt2 * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= kv_len) s = -INFINITY;
        }
        st[kt2][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);  // m_run = -inf on the first tile -> 0
    float rs = 0.f;
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(st[kt2][r] - m_new);
        st[kt2][r] = p;
        rs += p;
      }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }

    // ---- O^T += V^T . P^T ; k-step (kt2, s): key(j) = 32kt2 + 16s + 8(j>>2) + 4hh + (j&3) ----
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)st[kt2][8 * s + j];
        const int koff = (kt2 * 32 + s * 16 + 4 * hh) * 2;  // bytes
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* vrow = Vs + (dt * 32 + ql) * VT_ROW + koff;
          const uint2 lo = *(const uint2*)(vrow);
          const uint2 hi = *(const uint2*)(vrow + 16);
          union { uint4 u; bf16x8 v; } vf;
          vf.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf, oacc[dt], 0, 0, 0);
        }
      }

    if (t + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise + store: oacc[dt][reg] = O[q = ql][d = 32dt + (reg&3) + 8(reg>>2) + 4hh] ----
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

extern "C" int f5e_flash_attn(hipStream_t st, const void* q, const void* k, const void* vt, void* o, int ldo,
                              const int* kv_len, int S, int H, int rows_per_seq, int n_pad, int waves) {
  F5E_REQUIRE(q && k && vt && o, "flash_attn: null pointer");
  F5E_REQUIRE(S > 0 && H > 0 && rows_per_seq > 0, "flash_attn: empty problem");
  F5E_REQUIRE(n_pad % 64 == 0 && n_pad >= rows_per_seq, "flash_attn: n_pad=%d must be a multiple of 64 and >= %d",
              n_pad, rows_per_seq);
  F5E_REQUIRE(ldo % 4 == 0 && ldo >= H * 64, "flash_attn: bad ldo=%d", ldo);
  AttnArgs a{};
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.vt = (const bf16*)vt; a.o = (bf16*)o; a.ldo = ldo;
  a.kv_len = kv_len; a.S = S; a.H = H; a.rows_per_seq = rows_per_seq; a.n_pad = n_pad;
  a.scale_log2e = 0.125f * 1.4426950408889634f;
  if (waves <= 0) {
    // fill the 256 CUs: prefer 128-query workgroups only when that still yields >= 256 of them
    const int g4 = ((rows_per_seq + 127) / 128) * H * S;
    waves = g4 >= 256 ? 4 : 2;
  }
  if (waves == 4) {
    const int grid = ((rows_per_seq + 127) / 128) * H * S;
    hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, a);
  } else if (waves == 2) {
    const int grid = ((rows_per_seq + 63) / 64) * H * S;
    hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3(grid), dim3(128), 0, st, a);
  } else {
    F5E_REQUIRE(false, "flash_attn: waves must be 0 (auto), 2 or 4");
  }
  F5E_LAUNCH_CHECK("flash_attn");
  return F5E_OK;
}
