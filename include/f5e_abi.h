/* f5e_abi.h -- C ABI of libf5e_hip.so: the MI355X (gfx950) kernels of the F5E-TTS flow-matching inference hot path.
 *
 * The reference (kaleo996/F5E-TTS) has no FFI for this path: it is a chain of PyTorch nn.Module calls.  Each entry
 * point below therefore replaces a *PyTorch op sequence*; the comment on each names the reference lines
 * (paths relative to src/f5_tts/).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions (lower side)
 *   - plain pointers and ints only; every pointer is a DEVICE pointer unless the name says "host".
 *   - the caller owns every buffer (workspace included); the library never allocates, frees or synchronises inside
 *     an op, so every op is capturable into a hipGraph (f5e_graph_*).
 *   - hipStream_t is passed as void* (the value of torch.cuda.current_stream().cuda_stream).
 *   - row-major; "ld*" = leading dimension in ELEMENTS; bf16 = 16-bit brain float, f32 = IEEE binary32.
 *   - return value: 0 = F5E_OK, negative = error; message via f5e_last_error() (thread local).  No C++ exception
 *     crosses the ABI.  Re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef F5E_ABI_H_
#define F5E_ABI_H_

#ifdef __cplusplus
extern "C" {
#endif

#define F5E_ABI_VERSION 2   /* 2: f5e_ln_fuse.row_stats / f5e_ln_finalize / F5E_WS_LN_ROWSTATS removed, f5e_sample_loop added */

/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#if defined(__GNUC__)
#define F5E_API __attribute__((visibility("default")))
#else
#define F5E_API
#endif

enum { F5E_OK = 0, F5E_ERR_BAD_SHAPE = -1, F5E_ERR_UNSUPPORTED = -2, F5E_ERR_HIP = -3 };

/* activation selectors */
enum { F5E_ACT_NONE = 0, F5E_ACT_SILU = 1, F5E_ACT_GELU_ERF = 2, F5E_ACT_GELU_TANH = 3, F5E_ACT_RELU = 4, F5E_ACT_MISH = 5 };

#ifdef F5E_STREAM_T
typedef F5E_STREAM_T f5e_stream; /* library build: the real hipStream_t (same ABI: one pointer) */
#else
typedef void* f5e_stream; /* hipStream_t */
#endif

F5E_API int f5e_abi_version(void);
F5E_API const char* f5e_last_error(void);
/* 0 when the current HIP device is gfx950; F5E_ERR_UNSUPPORTED otherwise (host call, no kernel launch). */
F5E_API int f5e_check_device(void);

/* ---------------------------------------------------------------- bf16 MFMA GEMMs (hot loop) ----------------- */

/* out[M][N] = act(A[M][K] . W[N][K]^T + bias).  A, W bf16; bias f32[N] or NULL; out bf16 (out_f32 = 0) or f32.
 * act: F5E_ACT_NONE or F5E_ACT_GELU_TANH.  K % 64 == 0, N % 4 == 0.  tile_hint: 0 = auto, or force a tile family: 1 = 128x128,
 * 2 = 128x64, 3 = 64x64, 9 = the 256x256 ping-pong kernel (what auto picks from 44 row tiles of 256 on).
 * Replaces: FeedForward project_in + GELU(tanh) (model/modules.py:348-349,625), proj_out (backbones/dit.py:470). */
F5E_API int f5e_gemm_bf16_bias(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                       int ldo, int M, int N, int K, int act, int out_f32, int tile_hint);

/* resid[m][n] += gate[(m / rows_per_seq) % gate_rows][n] * (A . W^T + bias)[m][n], skipped for rows whose position
 * (m % rows_per_seq) >= seq_len[m / rows_per_seq] when seq_len != NULL.  gate is read at
 * gate + (*eval_ptr) * eval_stride when eval_ptr != NULL.
 * Replaces: attn.to_out + masked_fill + "x + gate_msa * attn" (modules.py:494-501,635) and
 *           ff.ff[2] + "x + gate_mlp * ff" (modules.py:350,639). */
F5E_API int f5e_gemm_bf16_gate_residual(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                int N, int K, int tile_hint);

/* Fused to_q/to_k/to_v (+bias) + rotary embedding on the first rope_heads heads of q and k.  W = [3*heads*64][K]
 * (rows: q | k | v).  Outputs: q, k [S][heads][n_pad][64] bf16, vt [S][heads][64][n_pad] bf16 (V transposed).
 * cos_sin: [rows_per_seq][32][2] f32 from f5e_rope_table.  Pad rows/columns of q/k/vt are never written: the
 * caller zero-fills them once.
 * q is stored PRE-SCALED by log2(e) / sqrt(64) (folded in after RoPE, before the bf16 rounding): f5e_flash_attn's scores
 * are then in the log2 domain with no multiply per element (csrc/attention.hip).
 * q_norm_w / k_norm_w: optional f32[64] RMSNorm weights applied per head before RoPE (qk_norm = "rms_norm", eps 1e-6).
 * Replaces: modules.py:452-461 (projections + head split), :464-467 (qk norm) and :470-480 (apply_rotary_pos_emb). */
F5E_API int f5e_gemm_bf16_qkv_rope(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias, void* q,
                           void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                           const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K,
                           int tile_hint);

/* Fused AdaLayerNorm for small row counts (batch-1 sampling): the LayerNorm + modulate launch in front of a linear
 * (modules.py:308-314 + :452-454; :637 + :349; :329-335 + dit.py:470) is folded into the GEMMs either side of it.
 *   producer = the gate+residual GEMM before it: next to x_new it writes xs = bf16((x_new - o) (1 + next_scale[n])) and,
 *     per row and 64-column tile, (mean - o, M2) of x_new into stats_out [M][N / 64][2], o = row_mean[m];
 *   consumer = the linear after it, run on A = xs:  out = rstd (acc - mean' c[n]) + d[n]  with the per-evaluation
 *     tables c[n] = sum_k W[n][k] (1 + scale[k]),  d[n] = sum_k W[n][k] shift[k] + bias[n]  (row r =
 *     (m / rows_per_seq) % cd_rows, advanced by (*eval_ptr) * cd_eval_stride), mean' (relative to o) / rstd combined from
 *     the `parts` tile statistics (Chan's formula, fixed order); pass bias = NULL to the GEMM.  Its column tile 0 then moves
 *     row_mean[m] += mean', so the next producer centres with the row's current mean.
 *   Why centred: bf16(x (1 + scale)) would round at the size of the row's OFFSET, and the error of the normalised result
 *     would grow like |mean| / std (off-centre rows of trained networks).  Row means drift slowly from norm to norm, so
 *     with o = the mean as of the previous norm the rounding is at the size of the row's spread, as in the unfused form.
 * Only one side is used per launch; leave the other side's pointers NULL.  Runs on the 64x64 tile family whatever M (the
 * fusion pays at small row counts, where the LayerNorm launches are pure latency; at large M keep f5e_layernorm).
 * Consumer limits: parts a multiple of 4 up to 16 (D <= 1024), cd_rows == 1 (one table row per evaluation). */
typedef struct f5e_ln_fuse {
  const float* stats; int parts;                       /* consumer */
  const float* c; const float* d; int cd_stride; int cd_rows; int cd_eval_stride;
  const int* eval_ptr; int rows_per_seq; float eps;
  void* xs_out; int ld_xs; const float* next_scale;    /* producer (next_scale uses the gate's strides / rows) */
  float* stats_out;
  float* row_mean;                                     /* both sides: [M] f32 centring offsets, see below */
} f5e_ln_fuse;

F5E_API int f5e_gemm_bf16_bias_ln(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias, void* out,
                          int ldo, int M, int N, int K, int act, int out_f32, int tile_hint, const f5e_ln_fuse* ln);
F5E_API int f5e_gemm_bf16_gate_residual_ln(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias,
                                   float* resid, int ldr, const float* gate, int gate_stride, int gate_rows,
                                   const int* eval_ptr, int eval_stride, int rows_per_seq, const int* seq_len, int M,
                                   int N, int K, int tile_hint, const f5e_ln_fuse* ln);
F5E_API int f5e_gemm_bf16_qkv_rope_ln(f5e_stream st, const void* A, int lda, const void* W, int ldw, const float* bias,
                              void* q, void* k, void* vt, int n_pad, int heads, int rope_heads, const float* cos_sin,
                              const float* q_norm_w, const float* k_norm_w, int rows_per_seq, int M, int K,
                              int tile_hint, const f5e_ln_fuse* ln);

/* ---------------------------------------------------------------- attention ---------------------------------- */

/* o[S*rows_per_seq][ldo] (bf16, column = head*64 + d) = softmax(q k^T / 8 + keymask) v, keys >= kv_len[s] masked.
 * q as f5e_gemm_bf16_qkv_rope writes it: already multiplied by log2(e) / 8, so the kernel computes 2^(q' k^T - max).
 * q, k, v use the fragment-major layouts documented in csrc/attention.hip; splits: KV splits per 32-query tile
 * (0 = auto, 1, 2 or 4; -1 = the LDS-shared 128-query kernel that auto picks for large problems).
 * Replaces: F.scaled_dot_product_attention + transpose/reshape (modules.py:482-492). */
F5E_API int f5e_flash_attn(f5e_stream st, const void* q, const void* k, const void* vt, void* o, int ldo, const int* kv_len,
                   int S, int H, int rows_per_seq, int n_pad, int splits);

/* ---------------------------------------------------------------- normalisation ------------------------------ */

/* y = LN(x; eps)  [* gamma + beta]  [* (1 + scale[r]) + shift[r]],  r = (row / rows_per_seq) % mod_rows.
 * x f32 [rows][D]; y bf16 or f32.  D % 256 == 0, D <= 2048.  scale/shift advance by (*eval_ptr) * eval_stride.
 * Replaces: AdaLayerNorm / ff_norm modulation / AdaLayerNorm_Final (modules.py:308-314,329-335,637) and the affine
 * LayerNorms of ConvNeXtV2Block (modules.py:253,264) and Vocos. */
F5E_API int f5e_layernorm(f5e_stream st, const float* x, int ldx, void* y, int ldy, int y_bf16, const float* gamma,
                  const float* beta, const float* scale, const float* shift, int mod_stride, int mod_rows,
                  int rows_per_seq, const int* eval_ptr, int eval_stride, int rows, int D, float eps);

/* First producer of the fused-AdaLN chain (block 0 has no GEMM in front of its norm): row_mean[row] = the row's exact
 * mean, xs = bf16((x - mean) (1 + scale[r])) and stats [rows][parts][2] holding `parts` equal shares (0, M2 / parts) of
 * each row's statistics (tile means relative to row_mean). */
F5E_API int f5e_adaln_pre(f5e_stream st, const float* x, int ldx, void* xs, int ld_xs, const float* scale, int mod_stride,
                  int mod_rows, int rows_per_seq, const int* eval_ptr, int eval_stride, float* stats, int parts,
                  float* row_mean, int rows, int D);

/* x_transformers.RMSNorm used by UNetT (backbones/unett.py:151,161,178): y = x / max(||x||_2, 1e-12) * sqrt(D) * g. */
F5E_API int f5e_l2norm(f5e_stream st, const float* x, int ldx, void* y, int ldy, int y_bf16, const float* g, int rows, int D);

/* GRN over the sequence axis (modules.py:225-234): x, y f32 [B][T][C]; gx_ws f32 [B][C] scratch. */
F5E_API int f5e_grn(f5e_stream st, const float* x, float* y, float* gx_ws, const float* gamma, const float* beta, int B, int T,
            int C);

/* ---------------------------------------------------------------- exact-fp32 GEMM (once per call) ------------ */

/* C[m][n] = ((act(sum_k actA(A[m % a_rows][k]) W[n][k] + bias[n])) * ch_scale[n] + addend[m % add_rows][n]) * row_scale[m]
 * written to out (f32) and/or out_bf16.  K, lda, ldw multiples of 4.  NULL skips a term.
 * Replaces the fp32 F.linear calls of TimestepEmbedding, AdaLN emb, TextEmbedding, PPGEmbedding, InputEmbedding.proj
 * (x columns per step; cond/text/ppg columns once per call) and Vocos. */
F5E_API int f5e_gemm_f32(f5e_stream st, const float* A, int lda, int a_rows, int a_act, const float* W, int ldw,
                 const float* bias, int act, const float* ch_scale, const float* addend, int ld_add, int add_rows,
                 const float* row_scale, float* out, int ldo, void* out_bf16, int ldo_bf16, int M, int N, int K);

/* ---------------------------------------------------------------- convolutions ------------------------------- */

/* One grouped Conv1d(D, D, 31, groups, padding = 15) + Mish of ConvPositionEmbedding (modules.py:167-190, called with
 * mask=None at backbones/dit.py:176).  x bf16 [S*N][ldx]; w_packed bf16 [groups][31][64 oc][64 ic], zero padded when
 * D/groups (16, 32, 48 or 64) is below 64.  mode 0: out_bf16 = mish(conv + bias); mode 1: out_f32 = mish(..) + resid. */
F5E_API int f5e_convpos(f5e_stream st, const void* x, int ldx, const void* w_packed, const float* bias, int mode,
                void* out_bf16, int ldo, float* out_f32, int ldo32, const float* resid, int ldr, int S, int N, int D,
                int groups);

/* Depthwise Conv1d(C, C, 7, padding 3, groups C), channels-last f32 [B][T][C]; w_t = weight transposed to [7][C]. */
F5E_API int f5e_dwconv7(f5e_stream st, const float* x, const float* w_t, const float* bias, float* y, int B, int T, int C);

/* col[b][t][j*Cin + ic] = x[b][t + j - pad][ic] (0 outside [0, T)). */
F5E_API int f5e_im2col(f5e_stream st, const float* x, float* col, int B, int T, int Cin, int ksize, int pad);

/* ---------------------------------------------------------------- sampler elementwise ------------------------ */

/* out[e] = cat(sin(a), cos(a)), a = (scale * t[e]) * freqs[k]   (modules.py:154-161; freqs = host constant [dim/2]) */
F5E_API int f5e_sinus_embed(f5e_stream st, const float* t, const float* freqs, float* out, int E, int dim, float scale);
/* out[n][i] = (cos, sin)(n * inv_freq[i])   (x_transformers RotaryEmbedding.forward_from_seq_len) */
F5E_API int f5e_rope_table(f5e_stream st, const float* inv_freq, float* out, int N, int half);
/* out[b][n] = (table[ids[b][n]] + pos[min(n, max_pos-1)]) * keep[b][n]   (backbones/dit.py:68-80).  table f32
 * [table_rows][TD]; an id outside [0, table_rows) is clamped (nn.Embedding raises IndexError there: the host side checks
 * ids that start on the host, the clamp only keeps device-resident garbage from reading outside the table). */
F5E_API int f5e_text_gather(f5e_stream st, const int* ids, const float* table, const float* pos, const float* keep, float* out,
                    int B, int N, int TD, int max_pos, int table_rows);
/* v = p0 | p0 + (p0 - p1) w0 | w0 (p2 - p1) + w1 (p1 - p0) + p0   (mode 0 | 1 | 2; p_k = pred + k * branch_stride);
 * dst = base + coef[*eval_ptr] * v; traj (optional) gets a copy.   (model/cfm.py:447, :187, :310 + Euler/midpoint)
 * done_ctr (optional, one zero-initialised u32): the last workgroup to finish does ++*eval_ptr and re-zeroes it. */
F5E_API int f5e_ode_update(f5e_stream st, const float* pred, long long branch_stride, int mode, float w0, float w1,
                   const float* base, float* dst, float* traj, const float* coef, int* eval_ptr, unsigned* done_ctr,
                   long long n);
/* Same, with the trajectory row picked on the device: traj row (*eval_ptr + 1) / traj_div of [rows][traj_stride] floats
 * (euler: traj_div 1; midpoint's second stage: 2), so a captured step needs no per-step copy.  traj_stride 0 = plain. */
F5E_API int f5e_ode_update_traj(f5e_stream st, const float* pred, long long branch_stride, int mode, float w0, float w1,
                        const float* base, float* dst, float* traj, long long traj_stride, int traj_div,
                        const float* coef, int* eval_ptr, unsigned* done_ctr, long long n);
F5E_API int f5e_advance_eval(f5e_stream st, int* eval_ptr);
/* out = mask ? cond : y   (cfm.py:476); mask u8 [rows], tensors f32 [rows][C] */
F5E_API int f5e_stitch(f5e_stream st, const float* cond, const float* y, const unsigned char* mask, float* out, long long rows,
               int C);
F5E_API int f5e_cast_bf16(f5e_stream st, const float* x, void* y, long long n);
/* y f32 = x bf16 (exact).  Used to rebuild the fused-AdaLN tables from the bf16 weights the MFMA kernels read. */
F5E_API int f5e_cast_f32(f5e_stream st, const void* x, float* y, long long n);
/* out = a x + b y + c  (y may be NULL).  (1 - t_inter) y0 + t_inter cond of duplicate_test (model/cfm.py:460-465). */
F5E_API int f5e_axpby(f5e_stream st, const float* x, const float* y, float* out, float a, float b, float c, long long n);
/* GumbelVectorQuantizer eval forward (model/modules.py:881-950): logits f32 [rows][ld] (groups * num_vars used) ->
 * targets i32 [rows][groups] (first maximal index), out f32 [rows][groups * var_dim] gathered from vars f32
 * [(combine_groups ? 1 : groups) * num_vars][var_dim]; stats (optional) f32[2] = (code_perplexity, prob_perplexity). */
F5E_API int f5e_vq_eval(f5e_stream st, const float* logits, int ld, const float* vars, int combine_groups, float* out,
                int* targets, float* stats, int rows, int groups, int num_vars, int var_dim);

/* ---------------------------------------------------------------- mel / vocoder ------------------------------ */

/* out[B][T][n_mels] = log(clamp(|STFT(wav)| . fb, 1e-5)), T = 1 + nw / hop, reflect-padded, centred (modules.py:75-101).
 * window f32 [1024]; twiddle f32 [512][2] = (cos, -sin)(2 pi k / 1024); fb f32 [513][n_mels]. */
F5E_API int f5e_stft_logmel(f5e_stream st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                    const float* fb, float* out, int B, int n_fft, int hop, int n_mels);
/* The same, bit for bit, with the filterbank handed over banded: filter m is non-zero on FFT bins [lo, lo + cnt) only
 * (fb_band i32 [n_mels][3] = lo, cnt, offset into fb_compact; fb_compact f32 [nnz <= 2048] = those weights, filter after
 * filter).  What MelSpec uses; the dense form above stays for arbitrary filterbanks. */
F5E_API int f5e_stft_logmel_banded(f5e_stream st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                           const float* fb_compact, const int* fb_band, int nnz, float* out, int B, int n_fft, int hop,
                           int n_mels);
/* Vocos ISTFTHead tail: z f32 [B*T][ldz] (513 log-magnitudes | 513 phases) -> out f32 [B][hop * (T - 1)];
 * frames_ws f32 [B*T][1024] scratch. */
F5E_API int f5e_istft_head(f5e_stream st, const float* z, int ldz, const float* window, const float* twiddle, float* frames_ws,
                   float* out, int B, int T, int n_fft, int hop);

/* ---------------------------------------------------------------- PPG extractor front ------------ */

/* torchaudio.compliance.kaldi.fbank as the reference calls it (ppg/wenet/dataset/feats.py:66-72): frame f = samples
 * [f shift, f shift + win) of wav * in_scale (snip_edges), minus its mean, pre-emphasised (x[j] - preemph x[j-1], x[-1] = x[0]),
 * times window[win] (povey), zero-padded to 512, |rfft|^2 . fb[257][n_mels], log(max(., eps)).
 * out f32 [B][T][n_mels], T = 1 + (nw - win) / shift.  twiddle f32 [256][2] = (cos, -sin)(2 pi k / 512). */
F5E_API int f5e_kaldi_fbank(f5e_stream st, const float* wav, int nw, int ldw, const float* window, const float* twiddle,
                    const float* fb, float* out, int B, int win, int shift, int n_mels, float in_scale, float preemph,
                    float eps);
/* y[r][c] = x[r][c] * sigmoid(x[r][C + c])   (F.glu over channels, ppg/wenet/transformer/convolution.py:119) */
F5E_API int f5e_glu(f5e_stream st, const float* x, int ldx, float* y, int ldy, long long rows, int C);
/* Depthwise Conv1d(C, C, K, padding (K-1)/2, groups C), channels-last f32 [B][T][C]; w_t = weight transposed to [K][C];
 * keep (optional f32 [B][T], 0/1): frames with keep == 0 count as zeros (convolution.py:100-101).  Odd K <= 31. */
F5E_API int f5e_dwconv(f5e_stream st, const float* x, const float* w_t, const float* bias, const float* keep, float* y, int B,
               int T, int C, int K);
/* y[r][k] = softmax_k(scale * x[r][k], k < len) for k < len, 0 for len <= k < ldy; len = kv_len[r / rows_per_seq] or L
 * (ppg/wenet/transformer/attention.py:75-87). */
F5E_API int f5e_softmax_rows(f5e_stream st, const float* x, int ldx, float* y, int ldy, const int* kv_len, long long rows,
                     int rows_per_seq, int L, float scale);

/* ---------------------------------------------------------------- fused DiT evaluation ----------------------- */

typedef struct f5e_dit_block_weights {
  const void* w_qkv;  const float* b_qkv;  /* bf16 [3*H*64][D], f32 [3*H*64] */
  const void* w_out;  const float* b_out;  /* bf16 [D][H*64] */
  const void* w_ff1;  const float* b_ff1;  /* bf16 [FF][D] */
  const void* w_ff2;  const float* b_ff2;  /* bf16 [D][FF] */
  const float* q_norm_w; const float* k_norm_w; /* f32 [64] each, or NULL (qk_norm off) */
} f5e_dit_block_weights;

typedef struct f5e_dit_plan {
  int S, B, N, n_pad, D, H, rope_heads, FF, L, mel;
  int mod_rows;               /* rows of the modulation table per evaluation (1, or B for per-item times) */
  /* inputs */
  const float* y;             /* [B*N][mel] f32: network input (ODE state) */
  const float* w_x; int ldw_x;/* input_embed.proj.weight[:, :mel] f32 */
  const float* in_const;      /* [S*N][D] f32: cond/text/ppg part of the input projection + bias, per branch */
  const void* convpos_w1; const float* convpos_b1;
  const void* convpos_w2; const float* convpos_b2;
  int convpos_groups;         /* ConvPositionEmbedding groups (16 in the reference) */
  const float* rope_cs;       /* [N][32][2] */
  const int* seq_len;         /* [S] valid frames per sequence, or NULL (= N, the reference's mask=None case) */
  const float* mod;           /* [E][mod_rows][L*6*D + 2*D] f32 */
  const int* eval_ptr;        /* device int: current evaluation index into mod */
  const f5e_dit_block_weights* blocks; /* HOST array [L] */
  const void* w_proj; const float* b_proj; /* bf16 [mel][D], f32 [mel] */
  const float* w_skip;        /* long_skip_connection.weight f32 [D][2D] or NULL (backbones/dit.py:264,466-467) */
  float* skip_res; float* skip_tmp; /* [S*N][D] f32 each, only with w_skip */
  /* workspace (caller-owned, sizes in elements) */
  float* h0; void* h0_bf16; void* c1;   /* [S*N][D] f32 / bf16 / bf16 */
  float* x;                              /* [S*N][D] f32 residual stream */
  void* hn;                              /* [S*N][D] bf16 */
  void* q; void* k; void* vt;            /* [S][H][n_pad][64] x2, [S][H][64][n_pad] bf16, zero-filled once */
  void* ao;                              /* [S*N][H*64] bf16 */
  void* ff;                              /* [S*N][FF] bf16 */
  float* pred;                           /* [S*N][mel] f32 */
  /* optional instrumentation (eager launches only, never inside graph capture) */
  void* timer;                           /* from f5e_timer_create, or NULL */
  int timer_op;                          /* F5E_OP_* op class to bracket with HIP events; a NEGATIVE value -(mask) selects
                                            every class whose bit (1 << F5E_OP_x) is set in mask */
  /* fused AdaLN (f5e_ln_fuse): 2 L + 1 LayerNorm launches become one f5e_adaln_pre.  Needs no long skip / qk_norm. */
  int fuse_ln;                           /* 0 = separate LayerNorm launches */
  float* ln_stats;                       /* [S*N][D / 64][2] f32 workspace */
  float* ln_rowmean;                     /* [S*N] f32 workspace: the rows' centring offsets (f5e_ln_fuse.row_mean) */
  const float* cd;                       /* [E][mod_rows][cd_stride] f32: per block c_qkv | d_qkv | c_ff1 | d_ff1 */
  int cd_stride;                         /*   (3 H 64, 3 H 64, FF, FF), then c_proj | d_proj (mel, mel)          */
  /* batch-1 chains (fused AdaLN path): a few grid-tail workgroups of each block launch pull the weights of the launch after
   * next into the 256 MB Infinity Cache (the 646 MB of block weights cycle through it, so every GEMM otherwise streams
   * from HBM).  Pure performance hint: results never depend on it. */
  int mall_prefetch;
} f5e_dit_plan;

/* Workspace planner (SURVEY 8b lower side): byte size and 256-byte-aligned offset of every caller-owned buffer of
 * f5e_dit_plan inside ONE arena of `total` bytes.  Only the shape fields of `shape` are read (S, N, D, H, FF, mel, fuse_ln
 * and whether w_skip is set); n_pad = N rounded up to 64 is what the plan must carry.  Buffers that the shape does not
 * need (ln_stats without fuse_ln, skip_* without w_skip) get size 0.  q, k and vt must be zero-filled once by the caller
 * (their pad rows are never written).  Host call, no kernel launch, no allocation. */
enum { F5E_WS_H0 = 0, F5E_WS_H0_BF16, F5E_WS_C1, F5E_WS_X, F5E_WS_HN, F5E_WS_Q, F5E_WS_K, F5E_WS_VT, F5E_WS_AO, F5E_WS_FF,
       F5E_WS_PRED, F5E_WS_LN_STATS, F5E_WS_SKIP_RES, F5E_WS_SKIP_TMP, F5E_WS_LN_ROWMEAN, F5E_WS_COUNT };
typedef struct f5e_dit_workspace {
  int n_pad;
  unsigned long long bytes[F5E_WS_COUNT];
  unsigned long long offset[F5E_WS_COUNT];
  unsigned long long total;
} f5e_dit_workspace;
F5E_API int f5e_workspace_bytes(const f5e_dit_plan* shape, f5e_dit_workspace* out_host);

enum { F5E_OP_NONE = 0, F5E_OP_INPROJ = 1, F5E_OP_CONVPOS = 2, F5E_OP_LN = 3, F5E_OP_QKV = 4, F5E_OP_ATTN = 5,
       F5E_OP_OUT = 6, F5E_OP_FF1 = 7, F5E_OP_FF2 = 8, F5E_OP_FINAL = 9 };

/* HIP-event timer: host-side helpers (these DO allocate / synchronise; they are not ops).  f5e_dit_forward records
 * one start/stop pair around every launch of plan->timer_op, on the stream the kernels are launched on. */
F5E_API int f5e_timer_create(int capacity, void** timer_out);
F5E_API int f5e_timer_destroy(void* timer);
F5E_API int f5e_timer_reset(void* timer);
F5E_API int f5e_timer_read(void* timer, float* ms_out_host, int max_out, int* count_out_host);
F5E_API int f5e_timer_read_ops(void* timer, int* ops_out_host, int max_out, int* count_out_host); /* op class of each pair */

#ifdef F5E_TOOLS
/* Diagnostics, TOOLS build only (make -C f5e-tts_amd/csrc tools-lib -> libf5e_hip_tools.so; tools/convpos_time.py): while
 * buf != NULL, f5e_convpos launches a build of its kernels that writes 8 timestamps per workgroup.  This is
 * process-wide mutable state, which is why the shipped libf5e_hip.so neither contains nor exports it. */
F5E_API void f5e_debug_convpos_trace(void* buf);
#endif

/* One DiT.sample evaluation for S = branches * B sequences (backbones/dit.py:452-470 after the cached embeddings). */
F5E_API int f5e_dit_forward(f5e_stream st, const f5e_dit_plan* plan);

/* The fixed-grid ODE loop of the samplers (model/cfm.py:430-471 with torchdiffeq's euler / midpoint): enqueues
 * `steps` steps of   [f5e_dit_forward(eval_a); f5e_ode_update]   (euler) or
 *                    [f5e_dit_forward(eval_a); update y_mid = y + coef . v; f5e_dit_forward(eval_b); update y = y + coef . v]
 * (midpoint; eval_b reads y_mid) on `st`.  Nothing is synchronised or allocated: wrap the call in f5e_graph_begin / _end to
 * get the whole loop as ONE executable graph (what the Python host does), or call it with steps = 1 per step.
 * Per-step scalars come from device tables indexed by *eval_ptr (the plans' modulation / fused-AdaLN tables, `coef`), which
 * every update advances by itself through done_ctr -- so one enqueued step serves the whole grid. */
typedef struct f5e_loop_plan {
  const f5e_dit_plan* eval_a;  /* the step's (first) evaluation: y = the ODE state, pred = all branches' outputs */
  const f5e_dit_plan* eval_b;  /* midpoint: second evaluation (y = y_mid); NULL = euler */
  int steps;                   /* steps to enqueue */
  int mode; float w0, w1;      /* guidance combine of f5e_ode_update: 0 plain, 1 CFG, 2 three-branch */
  long long n;                 /* elements of the ODE state: B * N * mel */
  float* y;                    /* [n] state: y(t_k) in, y(t_k + steps) out */
  float* y_mid;                /* [n] midpoint scratch (eval_b->y), NULL for euler */
  const float* pred;           /* [branches][n]: where the evaluations leave their outputs, branch 0 first */
  const float* coef;           /* [E] step coefficients per evaluation: dt (euler) | dt / 2, dt (midpoint) */
  int* eval_ptr;               /* device int: evaluation counter (also eval_a / eval_b ->eval_ptr) */
  unsigned* done_ctr;          /* device u32, zero: arrival counter of the self-advancing update */
  float* traj;                 /* optional [rows][n]: row (evaluation + 1) / evals_per_step receives y after every step */
} f5e_loop_plan;
F5E_API int f5e_sample_loop(f5e_stream st, const f5e_loop_plan* loop);

/* ---------------------------------------------------------------- hipGraph capture --------------------------- */
F5E_API int f5e_graph_begin(f5e_stream st);
F5E_API int f5e_graph_end(f5e_stream st, void** graph_exec_out);
F5E_API int f5e_graph_launch(void* graph_exec, f5e_stream st);
F5E_API int f5e_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* F5E_ABI_H_ */
