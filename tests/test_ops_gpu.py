"""Per-kernel parity: every C-ABI op (through f5e_tts_amd.ops -> libf5e_hip.so) against a plain fp32 PyTorch/oracle
restatement of the same op on the same seeded inputs.  Tolerances are written next to each comparison:
bf16-output ops are compared at bf16 resolution (2^-8 relative), fp32 ops at 1e-4..1e-5."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import f5e_oracle as O
from tools import synth as SY

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    import f5e_tts_amd.ops as ops_mod
    ops_mod.require_device()
    return ops_mod


def g(seed):
    return torch.Generator().manual_seed(seed)


def dev(t):
    return t.cuda().contiguous()


QSCALE = 0.125 * 1.4426950408889634     # log2(e) / sqrt(64): folded into q by the QKV epilogue (f5e_abi.h)


def close(a, b, rtol, atol, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} " \
                          f"(ref max {float(b.abs().max()):.3e})"


@pytest.mark.parametrize("M,N,K,hint", [(938, 2048, 1024, 0), (938, 1024, 2048, 1), (130, 192, 128, 2), (77, 100, 64, 3),
                                        (1, 64, 64, 0), (256, 256, 256, 1),
                                        # hint 9 = the 256x256 ping-pong kernel (auto-selected at large M)
                                        (938, 2048, 1024, 9), (700, 768, 512, 9), (300, 100, 128, 9), (256, 256, 128, 9),
                                        (1300, 512, 2048, 9), (5000, 640, 192, 9)])
def test_gemm_bf16_bias(ops, M, N, K, hint):
    a = torch.randn(M, K, generator=g(1)).to(BF)
    w = (torch.randn(N, K, generator=g(2)) / math.sqrt(K)).to(BF)
    b = torch.randn(N, generator=g(3))
    ref = a.float() @ w.float().T + b
    out32 = torch.empty(M, N, device="cuda")
    ops.gemm_bf16_bias(dev(a), dev(w), dev(b), out32, tile_hint=hint)
    close(out32, ref, 1e-4, 1e-4, "f32 out")          # fp32 accumulate, only summation order differs
    out16 = torch.empty(M, N, device="cuda", dtype=BF)
    ops.gemm_bf16_bias(dev(a), dev(w), dev(b), out16, tile_hint=hint)
    close(out16, ref, 2 ** -7, 1e-3, "bf16 out")      # one bf16 rounding of the result
    ops.gemm_bf16_bias(dev(a), dev(w), dev(b), out16, act=ops.ACT_GELU_TANH, tile_hint=hint)
    close(out16, F.gelu(ref, approximate="tanh"), 2 ** -7, 2e-3, "gelu")


def test_gemm_bf16_identity_asymmetric(ops):
    """A = I against an asymmetric W catches transposed / permuted fragment maps (cdna guide section 3)."""
    for n, hints in ((128, (1, 2, 3)), (512, (1, 9))):
        a = torch.eye(n).to(BF)
        w = (torch.arange(n * n).reshape(n, n) % 251).float().to(BF)  # exactly representable, asymmetric
        out = torch.empty(n, n, device="cuda")
        for hint in hints:
            ops.gemm_bf16_bias(dev(a), dev(w), None, out, tile_hint=hint)
            assert torch.equal(out.cpu(), w.float().T), f"n {n} tile {hint}"


@pytest.mark.parametrize("M,N,K,rps,hint", [(938, 1024, 1024, 469, 0), (200, 256, 512, 50, 0), (938, 1024, 2048, 469, 9),
                                            (600, 768, 256, 100, 9),
                                            (786, 512, 256, 131, 9), (1028, 256, 128, 257, 9),   # odd rows per sequence
                                            (900, 192, 128, 300, 9), (900, 320, 128, 150, 9), (700, 64, 128, 350, 9),  # N % 256 != 0
                                            ])
def test_gemm_bf16_gate_residual(ops, M, N, K, rps, hint):
    S = M // rps
    a = torch.randn(M, K, generator=g(4)).to(BF)
    w = (torch.randn(N, K, generator=g(5)) / math.sqrt(K)).to(BF)
    b = torch.randn(N, generator=g(6))
    x = torch.randn(M, N, generator=g(7))
    table = torch.randn(3, 2, 5 * N, generator=g(8))            # [eval][rows][stuff]
    gate_view = table[0, :, N:2 * N]
    lens = torch.tensor([rps - 3 * (i + 1) for i in range(S)], dtype=torch.int32)
    lin = a.float() @ w.float().T + b
    e = 2
    gsel = table[e, :, N:2 * N]
    ref = x.clone()
    for m in range(M):
        s, pos = divmod(m, rps)
        if pos < lens[s]:
            ref[m] += gsel[s % 2] * lin[m]
    xd = dev(x)
    td = dev(table)
    ev = torch.tensor([e], dtype=torch.int32, device="cuda")
    ops.gemm_bf16_gate_residual(dev(a), dev(w), dev(b), xd, td[0, :, N:2 * N], rps, seq_len=dev(lens), eval_ptr=ev,
                                eval_stride=table.stride(0), tile_hint=hint)
    close(xd, ref, 1e-4, 2e-4, "gate residual")
    assert gate_view.shape == (2, N)


@pytest.mark.parametrize("hint", [9])
def test_gate_residual_pingpong_lean_and_general_wave_tiles(ops, hint):
    """256x256 kernel: a wave tile (128 rows) that is fully live and sees ONE gate row takes the lean read-modify-write (also
    across a sequence boundary); one with masked rows, rows past M or two gate rows takes the general path.  Both in one
    launch, against the fp32 reference; masked rows must stay bit-identical."""
    rps, N, K = 300, 512, 256
    for gate_rows, lens_l in [(1, [300, 300, 170, 300, 300]), (2, [300, 300, 300, 300, 300]), (1, [300] * 5)]:
        S = len(lens_l)
        M = S * rps
        a = torch.randn(M, K, generator=g(70)).to(BF)
        w = (torch.randn(N, K, generator=g(71)) / math.sqrt(K)).to(BF)
        b = torch.randn(N, generator=g(72))
        x = torch.randn(M, N, generator=g(73))
        gate = torch.randn(gate_rows, N, generator=g(74))
        lens = torch.tensor(lens_l, dtype=torch.int32)
        lin = a.float() @ w.float().T + b
        seq = torch.arange(M) // rps
        live = (torch.arange(M) % rps) < lens[seq]
        ref = torch.where(live[:, None], x + gate[seq % gate_rows] * lin, x)
        xd = dev(x)
        ops.gemm_bf16_gate_residual(dev(a), dev(w), dev(b), xd, dev(gate), rps, seq_len=dev(lens), tile_hint=hint)
        close(xd, ref, 1e-4, 2e-4, f"gate residual {gate_rows} {lens_l}")
        assert torch.equal(xd.cpu()[~live], x[~live])


@pytest.mark.parametrize("S,N,H,rope_heads,K,hint", [(2, 469, 16, 16, 1024, 0), (3, 70, 2, 1, 128, 0), (2, 469, 16, 16, 1024, 9),
                                                     (3, 150, 12, 1, 768, 9),
                                                     # odd rows per sequence: V^T key groups start at odd offsets (2-byte pieces),
                                                     # a sequence boundary in every other wave tile
                                                     (3, 131, 4, 1, 256, 9), (2, 257, 2, 2, 128, 9), (5, 199, 4, 4, 128, 9),
                                                     # heads not a multiple of 4: the last 256-column tile has waves past 3 * inner
                                                     (2, 300, 6, 6, 128, 9), (2, 260, 10, 1, 128, 9), (2, 300, 6, 2, 128, 9)])
def test_qkv_rope(ops, S, N, H, rope_heads, K, hint):
    inner = H * 64
    n_pad = (N + 63) // 64 * 64
    a = torch.randn(S * N, K, generator=g(9)).to(BF)
    w = (torch.randn(3 * inner, K, generator=g(10)) / math.sqrt(K)).to(BF)
    b = torch.randn(3 * inner, generator=g(11))
    lin = (a.float() @ w.float().T + b).view(S, N, 3, H, 64).permute(2, 0, 3, 1, 4)  # [3,S,H,N,64]
    freqs = O.rope_freqs(N, 64)
    q_ref, k_ref, v_ref = lin[0].clone(), lin[1].clone(), lin[2]
    q_ref[:, :rope_heads] = O.apply_rope(q_ref[:, :rope_heads], freqs)
    k_ref[:, :rope_heads] = O.apply_rope(k_ref[:, :rope_heads], freqs)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    cs = torch.empty(N, 32, 2, device="cuda")
    ops.rope_table(dev(inv), cs)
    ang = torch.outer(torch.arange(N).float(), inv)
    close(cs[..., 0], ang.cos(), 0, 2e-6, "cos table")
    close(cs[..., 1], ang.sin(), 0, 2e-6, "sin table")
    q = torch.zeros(S, H, n_pad, 64, device="cuda", dtype=BF)
    k = torch.zeros_like(q)
    vt = torch.zeros(S, H, 64, n_pad, device="cuda", dtype=BF)
    ops.gemm_bf16_qkv_rope(dev(a), dev(w), dev(b), q, k, vt, H, rope_heads, cs, N, tile_hint=hint)
    qi, vi = ops.qk_frag_index(n_pad), ops.v_frag_index(n_pad)   # fragment-major layouts -> [pos, d]
    unq = lambda t, idx: t.cpu().view(S, H, -1)[:, :, idx]
    close(unq(q, qi)[:, :, :N], q_ref * QSCALE, 2 ** -7, 4e-3 * QSCALE, "q (pre-scaled by log2(e) / 8)")
    close(unq(k, qi)[:, :, :N], k_ref, 2 ** -7, 4e-3, "k")
    close(unq(vt, vi)[:, :, :N], v_ref, 2 ** -7, 4e-3, "v")
    if n_pad > N:   # pad positions are never written
        assert float(unq(q, qi)[:, :, N:].abs().max()) == 0 and float(unq(vt, vi)[:, :, N:].abs().max()) == 0


def pack_qkv(ops, q, k, v, n_pad):
    """[S, H, N, 64] tensors -> zero-padded fragment-major device buffers (what the QKV GEMM epilogue writes)."""
    S, H, N, _ = q.shape
    qi, vi = ops.qk_frag_index(n_pad)[:N].reshape(-1), ops.v_frag_index(n_pad)[:N].reshape(-1)
    outs = []
    for t, idx in ((q, qi), (k, qi), (v, vi)):
        buf = torch.zeros(S, H, n_pad * 64, dtype=BF)
        buf[:, :, idx] = t.reshape(S, H, N * 64)
        outs.append(buf.view(S, H, n_pad, 64).cuda())
    return outs


@pytest.mark.parametrize("S,H,N,waves,masked", [(2, 16, 469, 0, False), (2, 4, 469, 4, True), (3, 2, 64, 2, True),
                                                (1, 2, 130, 4, False), (2, 2, 33, 1, True), (1, 16, 1875, 0, False),
                                                (2, 4, 469, -1, True), (1, 2, 130, -1, False), (3, 2, 64, -1, True),
                                                (2, 2, 33, -1, True), (8, 16, 938, 0, True),
                                                # round 4 split rule: 512 query tiles = the last 4-way split, 544 and C4's
                                                # 640 / 960 / 1408 unsplit (auto), ragged lengths on both sides
                                                (2, 16, 512, 0, True), (2, 16, 530, 0, True), (2, 16, 640, 0, False),
                                                (2, 16, 938, 0, True), (2, 16, 1390, 0, False)])
def test_flash_attn(ops, S, H, N, waves, masked):
    n_pad = (N + 63) // 64 * 64
    # q as f5e_gemm_bf16_qkv_rope hands it over: pre-multiplied by log2(e) / 8 before the bf16 rounding (f5e_abi.h)
    q = (torch.randn(S, H, N, 64, generator=g(12)) * QSCALE).to(BF)
    k = torch.randn(S, H, N, 64, generator=g(13)).to(BF)
    v = torch.randn(S, H, N, 64, generator=g(14)).to(BF)
    lens = torch.tensor([N - 5 * i for i in range(S)], dtype=torch.int32) if masked else None
    s = (q.float() @ k.float().transpose(-1, -2)) * math.log(2.0)           # 2^(q' k) = e^(q k / 8)
    if masked:
        km = torch.arange(N)[None, :] < lens[:, None]
        s = s.masked_fill(~km[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(S * N, H * 64)
    qd, kd, vtd = pack_qkv(ops, q, k, v, n_pad)
    out = torch.empty(S * N, H * 64, device="cuda", dtype=BF)
    ops.flash_attn(qd, kd, vtd, out, N, kv_len=dev(lens) if masked else None, waves=waves)
    # P is rounded to bf16 before P.V and the output to bf16: 2^-7 relative + small absolute
    close(out, ref, 2 ** -6, 6e-3, "attention")


@pytest.mark.parametrize("factor", [1.5, 4.0, 12.0])
def test_flash_attn_spike_forces_rescale(ops, factor):
    """A late key whose score towers over everything before it (cdna guide rule 26: a rare data-dependent branch needs an
    input that forces it).  The kernel keeps the maximum of a query's FIRST 64-key step and only checks later row sums:
    factor 1.5 (~17 octaves above it) stays on the fast path with p ~ 2^17; factor 4 (~46 octaves) trips the row-sum limit
    and takes the slow path (recompute, true maximum, rescale of l and O); factor 12 (~138 octaves) overflows exp2 to inf
    first -- the inf must be caught the same way.  Spikes in the first step, in a middle step and in the masked last step."""
    S, H, N = 1, 1, 300
    q0 = torch.randn(S, H, N, 64, generator=g(15))
    k = torch.randn(S, H, N, 64, generator=g(16)).to(BF)
    v = torch.randn(S, H, N, 64, generator=g(17)).to(BF)
    q = (q0 * QSCALE).to(BF)
    for key, qi in ((200, 17), (40, 99), (290, 250)):
        k[0, 0, key] = (q0[0, 0, qi] * factor).to(BF)   # score ~ factor |q|^2 / 8 >> the others
    lens = torch.tensor([295], dtype=torch.int32)
    s = (q.float() @ k.float().transpose(-1, -2)) * math.log(2.0)
    s = s.masked_fill(~(torch.arange(N) < 295)[None, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(S * N, H * 64)
    out = torch.empty(S * N, 64, device="cuda", dtype=BF)
    for splits in (1, 2, 4, -1):
        ops.flash_attn(*pack_qkv(ops, q, k, v, 320), out, N, kv_len=dev(lens), waves=splits)
        assert torch.isfinite(out.float()).all()
        close(out, ref, 2 ** -6, 6e-3, f"attention spike x{factor}, {splits} KV splits")


@pytest.mark.parametrize("D", [256, 512, 768, 1024])
def test_layernorm_variants(ops, D):
    rows, rps = 150, 50
    x = torch.randn(rows, D, generator=g(18)) * 3 + 1
    tab = torch.randn(2, 2, 6 * D, generator=g(19)) * 0.5
    e = 1
    sc, sh = tab[e, :, D:2 * D], tab[e, :, 0:D]
    ln = F.layer_norm(x, (D,), eps=1e-6)
    seq = torch.arange(rows) // rps % 2
    ref = ln * (1 + sc[seq]) + sh[seq]
    td = dev(tab)
    ev = torch.tensor([e], dtype=torch.int32, device="cuda")
    out = torch.empty(rows, D, device="cuda", dtype=BF)
    ops.layernorm(dev(x), out, scale=td[0, :, D:2 * D], shift=td[0, :, 0:D], rows_per_seq=rps, eval_ptr=ev,
                  eval_stride=tab.stride(0))
    close(out, ref, 2 ** -7, 2e-3, "ln modulate bf16")
    gam, bet = torch.randn(D, generator=g(20)), torch.randn(D, generator=g(21))
    out32 = torch.empty(rows, D, device="cuda")
    ops.layernorm(dev(x), out32, gamma=dev(gam), beta=dev(bet))
    close(out32, F.layer_norm(x, (D,), gam, bet, eps=1e-6), 1e-5, 1e-5, "ln affine f32")


@pytest.mark.parametrize("B,T,C", [(2, 77, 192), (1, 469, 1024), (2, 300, 1024), (1, 33, 66), (3, 5, 64)])
def test_grn(ops, B, T, C):
    x = torch.randn(B, T, C, generator=g(22))
    gam, bet = torch.randn(1, 1, C, generator=g(23)), torch.randn(1, 1, C, generator=g(24))
    out = torch.empty(B, T, C, device="cuda")
    ops.grn(dev(x), out, dev(gam.view(-1)), dev(bet.view(-1)), torch.empty(B, C, device="cuda"))
    close(out, O.grn(x, gam, bet), 1e-5, 1e-5, "grn")


@pytest.mark.parametrize("M,N,K", [(100, 1024, 100), (469, 512, 1024), (33, 1026, 512), (5, 64, 4), (281, 1536, 512),
                                   (938, 1024, 1536), (70, 100, 64)])
def test_gemm_f32(ops, M, N, K):
    a = torch.randn(M, K + 12, generator=g(25))[:, :K]          # strided views on purpose (lda != K)
    w = torch.randn(N, K + 8, generator=g(26))[:, :K] / math.sqrt(K)
    b = torch.randn(N, generator=g(27))
    ref = a @ w.T + b
    out = torch.empty(M, N, device="cuda")
    ad = torch.empty(M, K + 12, device="cuda").copy_(torch.cat([a, torch.zeros(M, 12)], 1))[:, :K]
    wd = torch.empty(N, K + 8, device="cuda").copy_(torch.cat([w, torch.zeros(N, 8)], 1))[:, :K]
    ops.gemm_f32(ad, wd, dev(b), out=out)
    close(out, ref, 1e-5, 2e-5, "plain")
    # full epilogue: silu on A, gelu(erf), channel scale, addend with row wrap, row scale, bf16 copy, A row wrap
    cs, rs = torch.randn(N, generator=g(28)), (torch.rand(2 * M, generator=g(29)) > 0.3).float()
    add = torch.randn(M, N, generator=g(30))
    ref2 = ((F.gelu(F.silu(a) @ w.T + b) * cs)[torch.arange(2 * M) % M] + add[torch.arange(2 * M) % M]) * rs[:, None]
    out2 = torch.empty(2 * M, N, device="cuda")
    out2b = torch.empty(2 * M, N, device="cuda", dtype=BF)
    ops.gemm_f32(ad, wd, dev(b), out=out2, out_bf16=out2b, M=2 * M, a_act=ops.ACT_SILU, act=ops.ACT_GELU_ERF,
                 ch_scale=dev(cs), addend=dev(add), row_scale=dev(rs))
    close(out2, ref2, 1e-5, 3e-5, "full epilogue")
    close(out2b, ref2, 2 ** -7, 1e-3, "bf16 copy")
    # the same epilogue without the A-side activation: K % 64 == 0 shapes take the LDS-DMA kernel here
    ref3 = ((F.gelu(a @ w.T + b) * cs)[torch.arange(2 * M) % M] + add[torch.arange(2 * M) % M]) * rs[:, None]
    ops.gemm_f32(ad, wd, dev(b), out=out2, out_bf16=out2b, M=2 * M, act=ops.ACT_GELU_ERF, ch_scale=dev(cs),
                 addend=dev(add), row_scale=dev(rs))
    close(out2, ref3, 1e-5, 3e-5, "full epilogue, plain A")
    close(out2b, ref3, 2 ** -7, 1e-3, "bf16 copy, plain A")
    for act, fn in ((ops.ACT_RELU, F.relu), (ops.ACT_MISH, F.mish), (ops.ACT_GELU_TANH, lambda t: F.gelu(t, approximate="tanh"))):
        ops.gemm_f32(ad, wd, dev(b), out=out, act=act)
        close(out, fn(ref), 1e-5, 3e-5, f"act {act}")


@pytest.mark.parametrize("S,N,D,G", [(2, 469, 1024, 16), (3, 70, 128, 2), (1, 5, 64, 1), (2, 100, 768, 16),
                                     (1, 130, 256, 16), (2, 65, 512, 16),
                                     # more workgroups than CUs: the output-split ring kernel (the others: split-tap)
                                     (8, 300, 1024, 16), (40, 130, 256, 16), (30, 64, 384, 8)])
def test_convpos(ops, S, N, D, G):
    cpg = D // G
    x = torch.randn(S, N, D, generator=g(31)).to(BF)
    w = (torch.randn(D, cpg, 31, generator=g(32)) / math.sqrt(cpg * 31)).to(BF)
    b = torch.randn(D, generator=g(33)) * 0.1
    res = torch.randn(S * N, D, generator=g(34))
    conv = F.conv1d(x.float().permute(0, 2, 1), w.float(), b, padding=15, groups=G).permute(0, 2, 1).reshape(S * N, D)
    wp = ops.pack_convpos_weight(w.float(), G)  # [G][tap][64 oc][64 ic], zero padded
    o16 = torch.empty(S * N, D, device="cuda", dtype=BF)
    ops.convpos(dev(x.view(S * N, D)), dev(wp), dev(b), S, N, out_bf16=o16)
    close(o16, F.mish(conv), 2 ** -7, 2e-3, "mode 0")
    o32 = torch.empty(S * N, D, device="cuda")
    ops.convpos(dev(x.view(S * N, D)), dev(wp), dev(b), S, N, out_f32=o32, resid=dev(res))
    close(o32, F.mish(conv) + res, 1e-4, 2e-4, "mode 1")


def test_dwconv7_im2col(ops):
    B, T, C = 2, 50, 512
    x = torch.randn(B, T, C, generator=g(35))
    w = torch.randn(C, 1, 7, generator=g(36))
    b = torch.randn(C, generator=g(37))
    ref = F.conv1d(x.transpose(1, 2), w, b, padding=3, groups=C).transpose(1, 2)
    out = torch.empty(B, T, C, device="cuda")
    ops.dwconv7(dev(x), dev(w[:, 0, :].T), dev(b), out)
    close(out, ref, 1e-5, 1e-5, "dwconv7")
    Cin, ks = 100, 7
    xi = torch.randn(B, T, Cin, generator=g(38))
    wi = torch.randn(64, Cin, ks, generator=g(39)) / 26
    col = torch.empty(B, T, ks * Cin, device="cuda")
    ops.im2col(dev(xi), col, ks, 3)
    o = torch.empty(B * T, 64, device="cuda")
    ops.gemm_f32(col.view(B * T, ks * Cin), dev(wi.permute(0, 2, 1).reshape(64, ks * Cin)), None, out=o)
    ref = F.conv1d(xi.transpose(1, 2), wi, None, padding=3).transpose(1, 2).reshape(B * T, 64)
    close(o, ref, 1e-5, 2e-5, "im2col conv")


def test_sampler_elementwise(ops):
    t = O.sway_time_grid(32, -1.0)
    half = 128
    freqs = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    out = torch.empty(33, 256, device="cuda")
    ops.sinus_embed(dev(t), dev(freqs), out)
    close(out, O.sinus_embedding(t), 0, 2e-4, "sinus")  # |arg| up to 1000: 1 ulp of arg is 6e-5
    B, N, TD = 2, 40, 64
    ids = torch.randint(0, 30, (B, N), generator=g(40), dtype=torch.int32)
    table, pos = torch.randn(30, TD, generator=g(41)), O.text_pos_table(TD, 64)
    keep = (ids != 0).float()
    te = torch.empty(B, N, TD, device="cuda")
    ops.text_gather(dev(ids), dev(table), dev(pos), dev(keep), te)
    close(te, (table[ids.long()] + pos[:N][None]) * keep[..., None], 0, 0, "text gather")
    n = B * N * 20
    pred = torch.randn(3, n, generator=g(42))
    y = torch.randn(n, generator=g(43))
    coef = torch.tensor([0.1, 0.25, 0.5])
    ev = torch.tensor([1], dtype=torch.int32, device="cuda")
    for mode, ref in ((0, y + 0.25 * pred[0]), (1, y + 0.25 * (pred[0] + (pred[0] - pred[1]) * 2.0)),
                      (2, y + 0.25 * (2.0 * (pred[2] - pred[1]) + 3.0 * (pred[1] - pred[0]) + pred[0]))):
        dst, traj = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        ops.ode_update(dev(pred), n, mode, 2.0, 3.0, dev(y), dst, dev(coef), ev, traj)
        close(dst, ref, 1e-6, 1e-6, f"ode mode {mode}")
        assert torch.equal(dst, traj)
    done = torch.zeros(1, dtype=torch.int32, device="cuda")
    for expect in (2, 3):   # auto-advance: the kernel bumps the evaluation counter itself and re-arms the ticket
        ops.ode_update(dev(pred), n, 0, 0.0, 0.0, dev(y), dst, torch.ones(8, device="cuda"), ev, None, done)
        assert int(ev.item()) == expect and int(done.item()) == 0
    ev.fill_(1)
    ops.advance_eval(ev)
    assert int(ev.item()) == 2
    mask = (torch.rand(B * N, generator=g(44)) > 0.5)
    c, yy = torch.randn(B * N, 20, generator=g(45)), torch.randn(B * N, 20, generator=g(46))
    so = torch.empty(B * N, 20, device="cuda")
    ops.stitch(dev(c), dev(yy), dev(mask.to(torch.uint8)), so)
    assert torch.equal(so.cpu(), torch.where(mask[:, None], c, yy))
    xb = torch.empty(n, device="cuda", dtype=BF)
    ops.cast_bf16(dev(y), xb)
    assert torch.equal(xb.cpu(), y.to(BF))


def fft_tables():
    k = torch.arange(512, dtype=torch.float64)
    tw = torch.stack((torch.cos(2 * math.pi * k / 1024), -torch.sin(2 * math.pi * k / 1024)), -1).float()
    return torch.hann_window(1024), tw


@pytest.mark.parametrize("B,frames", [(1, 188), (3, 21)])
def test_stft_logmel(ops, B, frames):
    wav = SY.synthetic_ref_wave(frames, batch=B)
    win, tw = fft_tables()
    out = torch.empty(B, frames, 100, device="cuda")
    ops.stft_logmel(dev(wav), dev(win), dev(tw), dev(O.mel_filterbank_htk()), out, 1024, 256)
    ref = O.log_mel_spectrogram(wav).permute(0, 2, 1)
    close(out, ref, 1e-4, 2e-4, "log-mel")  # log of a sum of |X|: fp32 FFT ordering differences only


def test_stft_logmel_banded_is_bit_identical_to_dense(ops):
    """The banded log-mel kernel (filter runs from LDS) must reproduce the dense kernel bit for bit: same products, same
    ascending-bin summation order, the skipped terms are exact zeros."""
    from f5e_tts_amd.engine import mel_filterbank
    wav = SY.synthetic_ref_wave(375, batch=3)
    win, tw = fft_tables()
    fb = dev(mel_filterbank(513, 100, 24000))
    band = ops.band_filterbank(fb)
    assert band is not None and band[0].numel() <= 2048 and band[1].shape == (100, 3)
    assert float(band[0].sum()) == pytest.approx(float(fb.sum()), rel=1e-6)
    dense, banded = torch.empty(3, 375, 100, device="cuda"), torch.empty(3, 375, 100, device="cuda")
    ops.stft_logmel(dev(wav), dev(win), dev(tw), fb, dense, 1024, 256)
    ops.stft_logmel_banded(dev(wav), dev(win), dev(tw), band[0], band[1], banded, 1024, 256)
    assert torch.equal(dense, banded)


@pytest.mark.parametrize("B,T", [(1, 281), (2, 9)])
def test_istft_head(ops, B, T):
    z = torch.randn(B * T, 1026, generator=g(47))
    z[:, :513] = z[:, :513] * 1.5 + 1.0   # some log-magnitudes beyond the exp clip at log(100) = 4.6
    z[0, 5] = 9.0
    win, tw = fft_tables()
    mag, ph = z.view(B, T, 1026).transpose(1, 2).chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    ref = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), 1024, 256, 1024, win, center=True)
    out = torch.empty(B, 256 * (T - 1), device="cuda")
    ops.istft_head(dev(z), dev(win), dev(tw), torch.empty(B * T, 1024, device="cuda"), out, B, T, 1024, 256)
    close(out, ref, 1e-4, 1e-4 * float(ref.abs().max()), "istft")


@pytest.mark.parametrize("ratio", [0.5, 2.0, 8.0, 40.0])
def test_fused_adaln_is_centred_error_does_not_grow_with_row_mean_over_std(ops, ratio):
    """the fused AdaLN used to round xs = bf16(x (1 + scale)) BEFORE the mean was removed, so its error
    grew like |row mean| / std (the separate LayerNorm rounds the normalised value itself) -- unsafe on trained checkpoints
    with off-centre rows.  Now xs = bf16((x - o)(1 + scale)) with o = the row's mean as of the previous norm
    (f5e_ln_fuse.row_mean): rows with |mean| = ratio * std, a massive-activation channel on top, head of the chain (o = the
    exact mean) and a producer -> consumer hop whose update MOVES every row's mean by ~0.3 std (o = the mean before the update).
    Bound: relative RMS error <= 2^-8 * 2 of the output spread whatever the ratio (was allowed 2^-8 (1 + ratio))."""
    M, D, NO, N = 300, 1024, 1024, 300
    x = torch.randn(M, D, generator=g(90)) + ratio
    x[:, 7] += 40.0                                                         # a massive-activation channel on top
    mod = torch.randn(1, 3 * D, generator=g(91)) * 0.3
    scale, shift, gate = mod[:, :D], mod[:, D:2 * D], mod[:, 2 * D:]
    w = (torch.randn(NO, D, generator=g(92)) / math.sqrt(D)).to(BF)
    b = torch.randn(NO, generator=g(93)) * 0.1
    wf = w.float()
    cd = (dev(((1 + scale) @ wf.T).contiguous()), dev((shift @ wf.T + b).contiguous()))
    xd, modd = dev(x), dev(mod)
    xs, stats = torch.empty(M, D, device="cuda", dtype=BF), torch.empty(M, D // 64, 2, device="cuda")
    rm = torch.empty(M, device="cuda")

    def errors(x_cpu, x_dev, out):
        ref = (F.layer_norm(x_cpu, (D,), eps=1e-6) * (1 + scale) + shift) @ wf.T + b
        hn, unf = torch.empty(M, D, device="cuda", dtype=BF), torch.empty(M, NO, device="cuda")
        ops.layernorm(x_dev, hn, scale=modd[:, :D], shift=modd[:, D:2 * D], rows_per_seq=N)
        ops.gemm_bf16_bias(hn, dev(w), dev(b), unf)
        spread = float(ref.std())
        return (float((out.cpu() - ref).pow(2).mean().sqrt()) / spread, float((unf.cpu() - ref).pow(2).mean().sqrt()) / spread)

    # head of the chain: exact row means
    ops.adaln_pre(xd, xs, modd[:, :D], stats, rm, N)
    close(rm, x.mean(1), 1e-5, 1e-5 * (1 + ratio), "row_mean of the head")
    out = torch.empty(M, NO, device="cuda")
    ops.gemm_bf16_bias(xs, dev(w), None, out, ln=ops.ln_consumer(stats, cd[0], cd[1], N, rm))
    e_f, e_u = errors(x, xd, out)
    print("head, mean/std %.1f: fused rms %.2e, separate rms %.2e" % (ratio, e_f, e_u))
    assert e_u < 2 ** -8 and e_f < 2 ** -8 * 2
    close(rm, x.mean(1), 1e-5, 1e-5 * (1 + ratio), "row_mean after a consumer with zero drift")
    # one producer hop: x2 = x + gate * (a2 @ w2^T + b2), an update that shifts every row's mean by about 0.3 std
    K2 = 2 * D
    a2 = torch.randn(M, K2, generator=g(94)).to(BF)
    w2 = (torch.randn(D, K2, generator=g(95)) / math.sqrt(K2)).to(BF)
    b2 = torch.randn(D, generator=g(96)) * 0.1 + 0.5 * torch.sign(gate.view(-1)) / gate.abs().mean()   # gate * b2 has mean ~ 0.5
    xs2, st2 = torch.zeros(M, D, device="cuda", dtype=BF), torch.zeros(M, D // 64, 2, device="cuda")
    x2d = xd.clone()
    ops.gemm_bf16_gate_residual(dev(a2), dev(w2), dev(b2.view(-1)), x2d, modd[:, 2 * D:], N, ln=ops.ln_producer(xs2, modd[:, :D], st2, rm))
    x2 = x2d.cpu()
    drift = float((x2.mean(1) - x.mean(1)).abs().mean() / x2.std(1).mean())
    out2 = torch.empty(M, NO, device="cuda")
    ops.gemm_bf16_bias(xs2, dev(w), None, out2, ln=ops.ln_consumer(st2, cd[0], cd[1], N, rm))
    e_f2, e_u2 = errors(x2, x2d, out2)
    print("hop, mean/std %.1f, drift %.2f std: fused rms %.2e, separate rms %.2e" % (ratio, drift, e_f2, e_u2))
    assert drift > 0.15
    assert e_u2 < 2 ** -8 and e_f2 < 2 ** -8 * 2
    close(rm, x2.mean(1), 1e-4, 1e-5 * (1 + ratio), "row_mean moved along by the consumer")


def _stft_fixture():
    import os

    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stft_head.npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_stft_logmel_against_reference_conv_stft_fixture(ops, tag):
    """HIP log-mel against the magnitude spectrogram produced by the REFERENCE's conv_stft.STFT.transform (fixture
    tests/golden/stft_head.npz): log(clamp(|S| . fb, 1e-5)) with the HTK filterbank of the product code."""
    from f5e_tts_amd.engine import mel_filterbank
    gfx = _stft_fixture()
    wav, mag = gfx[f"fwd_{tag}/wav"], gfx[f"fwd_{tag}/mag"]
    B, T = wav.shape[0], mag.shape[2]
    fb = mel_filterbank(513, 100, 24000)
    win, tw = fft_tables()
    out = torch.empty(B, T, 100, device="cuda")
    ops.stft_logmel(dev(wav), dev(win), dev(tw), dev(fb), out, 1024, 256)
    want = torch.matmul(mag.transpose(1, 2), fb).clamp(min=1e-5).log()
    close(out, want, 1e-3, 1e-3, "log-mel vs reference STFT magnitude")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_istft_head_against_reference_istft_head_fixture(ops, tag):
    """HIP iSTFT head against audio produced by the REFERENCE's ISTFTHead (export_vocoder_to_onnx.py:45-59) on the same
    pre-activations; the common hop * (T - 1) samples of batch item 0 (the reference's conv inverse leaves later batch
    items un-normalised, see tests/test_oracle_golden.py), and item 1 after multiplying by the window envelope."""
    gfx = _stft_fixture()
    z, ref = gfx[f"inv_{tag}/z"], gfx[f"inv_{tag}/audio"]
    B, T = z.shape[0], z.shape[1]
    win, tw = fft_tables()
    out = torch.empty(B, 256 * (T - 1), device="cuda")
    ops.istft_head(dev(z.reshape(B * T, 1026).contiguous()), dev(win), dev(tw), torch.empty(B * T, 1024, device="cuda"),
                   out, B, T, 1024, 256)
    common = ref[:, : 256 * (T - 1)]
    assert float((out[0].cpu() - common[0]).abs().max()) < 2e-4 * float(common[0].abs().max())
    if B > 1:
        env = torch.zeros(256 * T + 1024)
        for f in range(T):
            env[f * 256: f * 256 + 1024] += win ** 2
        env = env[512: 512 + 256 * (T - 1)]
        assert float((out[1].cpu() * env - common[1]).abs().max()) < 2e-4 * float(common[1].abs().max())


@pytest.mark.parametrize("hint", [0, 9])
def test_qkv_rope_with_qk_rmsnorm(ops, hint):
    """qk_norm = 'rms_norm' (reference modules.py:464-467): RMSNorm over the 64-d head before RoPE, q and k only."""
    S, N, H, K = 2, 150, 12, 768
    inner, n_pad = H * 64, 192
    a = torch.randn(S * N, K, generator=g(50)).to(BF)
    w = (torch.randn(3 * inner, K, generator=g(51)) / math.sqrt(K)).to(BF)
    b = torch.randn(3 * inner, generator=g(52))
    qw, kw = 1 + 0.2 * torch.randn(64, generator=g(53)), 1 + 0.2 * torch.randn(64, generator=g(54))
    lin = (a.float() @ w.float().T + b).view(S, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    freqs = O.rope_freqs(N, 64)
    q_ref = O.apply_rope(O.rms_norm(lin[0], qw), freqs)
    k_ref = O.apply_rope(O.rms_norm(lin[1], kw), freqs)
    cs = torch.empty(N, 32, 2, device="cuda")
    ops.rope_table(dev(1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))), cs)
    q = torch.zeros(S, H, n_pad, 64, device="cuda", dtype=BF)
    k, vt = torch.zeros_like(q), torch.zeros_like(q)
    ops.gemm_bf16_qkv_rope(dev(a), dev(w), dev(b), q, k, vt, H, H, cs, N, q_norm_w=dev(qw), k_norm_w=dev(kw),
                           tile_hint=hint)
    qi, vi = ops.qk_frag_index(n_pad), ops.v_frag_index(n_pad)
    unq = lambda t, idx: t.cpu().view(S, H, -1)[:, :, idx]
    close(unq(q, qi)[:, :, :N], q_ref * QSCALE, 2 ** -7, 4e-3 * QSCALE, "q normed (pre-scaled by log2(e) / 8)")
    close(unq(k, qi)[:, :, :N], k_ref, 2 ** -7, 4e-3, "k normed")
    close(unq(vt, vi)[:, :, :N], lin[2], 2 ** -7, 4e-3, "v untouched")


@pytest.mark.parametrize("S,N,D,NO,mean_shift,masked,hint", [
    (2, 469, 1024, 3072, 0.0, False, 0), (2, 100, 768, 1536, 0.7, True, 0), (1, 33, 1024, 100, 0.3, False, 0),
    # fused launches always run on the 64 x 64 tile family, whatever the hint and the row count: hint 9 (ignored by them;
    # the unfused reference launches below do take the 256 x 256 kernel) and a row count past the ping-pong crossover
    # (44 row tiles of 256), where a zero-initialised f5e_ln_fuse used to be refused
    (2, 469, 1024, 3072, 0.0, False, 9), (3, 150, 768, 1536, 0.7, True, 9), (24, 470, 1024, 256, 0.3, False, 0),
    # M in (1024, 2048]: the 128-row role-split tiles (producer 128 x 64, consumer 128 x 128), a partial last row tile and
    # a grid that fills all 256 CUs (M = 2048)
    (2, 900, 1024, 3072, 0.3, True, 0), (2, 1024, 1024, 2048, 0.0, False, 0), (3, 500, 768, 1536, 0.7, True, 0),
    ])
def test_fused_adaln_chain(ops, S, N, D, NO, mean_shift, masked, hint):
    """LayerNorm+modulate folded into the GEMMs either side of it (f5e_ln_fuse): adaln_pre / the gate+residual producer
    -> consumer linear, against LN(x)(1+scale)+shift -> linear in fp32 (reference modules.py:308-314 + :452-454,
    :637 + :349) and against the unfused HIP ops."""
    M, P = S * N, D // 64
    x = torch.randn(M, D, generator=g(70)) * 1.5 + mean_shift
    mod = torch.randn(1, 3 * D, generator=g(71)) * 0.3                      # scale | shift | gate
    scale, shift, gate = mod[:, :D], mod[:, D:2 * D], mod[:, 2 * D:]
    w = (torch.randn(NO, D, generator=g(72)) / math.sqrt(D)).to(BF)
    b = torch.randn(NO, generator=g(73)) * 0.1
    wf = w.float()
    ref = (F.layer_norm(x, (D,), eps=1e-6) * (1 + scale) + shift) @ wf.T + b
    c = ((1 + scale) @ wf.T).contiguous()                                      # tables, one row (mod_rows = 1)
    dd = (shift @ wf.T + b).contiguous()
    xd, modd = dev(x), dev(mod)
    xs = torch.empty(M, D, device="cuda", dtype=BF)
    stats = torch.empty(M, P, 2, device="cuda")
    rm = torch.empty(M, device="cuda")
    ops.adaln_pre(xd, xs, modd[:, :D], stats, rm, N)
    close(rm, x.mean(1), 1e-5, 1e-6, "pre row mean")
    close(xs, ((x - rm.cpu()[:, None]) * (1 + scale)).to(BF), 2 ** -7, 1e-6, "pre xs (centred)")   # a bf16 ulp at rounding ties
    assert float(stats[:, :, 0].abs().max()) == 0                        # tile means relative to the exact mean
    close(stats[:, :, 1].sum(1), ((x - x.mean(1, keepdim=True)) ** 2).sum(1), 1e-5, 1e-4, "pre M2")
    out = torch.empty(M, NO, device="cuda")
    ops.gemm_bf16_bias(xs, dev(w), None, out, ln=ops.ln_consumer(stats, dev(c), dev(dd), N, rm), tile_hint=hint)
    rm0 = rm.cpu().clone()
    close(rm0, x.mean(1), 1e-5, 1e-6, "row mean unchanged by a consumer without drift")
    hn = torch.empty(M, D, device="cuda", dtype=BF)
    ops.layernorm(xd, hn, scale=modd[:, :D], shift=modd[:, D:2 * D], rows_per_seq=N)
    unf = torch.empty(M, NO, device="cuda")
    ops.gemm_bf16_bias(hn, dev(w), dev(b), unf)
    spread = float(ref.std())
    e_f, e_u = float((out.cpu() - ref).abs().max()) / spread, float((unf.cpu() - ref).abs().max()) / spread
    assert e_f < 3e-2 and e_f < 2.5 * e_u + 1e-3, (e_f, e_u)              # bf16-level, on par with the unfused path
    assert float((out.cpu() - ref).pow(2).mean().sqrt()) / spread < 4e-3

    # producer: gate + residual GEMM that also emits the next consumer's xs and tile statistics
    K2 = 2 * D
    a2 = (torch.randn(M, K2, generator=g(74))).to(BF)
    w2 = (torch.randn(D, K2, generator=g(75)) / math.sqrt(K2)).to(BF)
    b2 = torch.randn(D, generator=g(76)) * 0.1
    seq_len = torch.tensor([N - 7, N, N - 20][:S], dtype=torch.int32) if masked else None
    x_plain, x_fused = xd.clone(), xd.clone()
    ops.gemm_bf16_gate_residual(dev(a2), dev(w2), dev(b2), x_plain, modd[:, 2 * D:], N,
                                seq_len=dev(seq_len) if masked else None, tile_hint=hint)
    xs2 = torch.zeros(M, D, device="cuda", dtype=BF)
    st2 = torch.zeros(M, P, 2, device="cuda")
    ops.gemm_bf16_gate_residual(dev(a2), dev(w2), dev(b2), x_fused, modd[:, 2 * D:], N,
                                seq_len=dev(seq_len) if masked else None, ln=ops.ln_producer(xs2, modd[:, :D], st2, rm),
                                tile_hint=hint)
    assert torch.equal(x_plain, x_fused)                                    # the residual update itself is unchanged
    xn = x_fused.cpu()
    if masked:
        assert torch.equal(xn[N - 7:N], x[N - 7:N])                         # masked rows keep x ...
    close(xs2, ((xn - rm0[:, None]) * (1 + scale)).to(BF), 2 ** -7, 1e-6, "producer xs (centred with the previous mean)")
    tiles = xn.view(M, P, 64)
    close(st2[:, :, 0], tiles.mean(2) - rm0[:, None], 1e-5, 2e-6, "producer tile mean (relative)")
    close(st2[:, :, 1], ((tiles - tiles.mean(2, keepdim=True)) ** 2).sum(2), 1e-4, 1e-4, "producer tile M2")
    out2 = torch.empty(M, NO, device="cuda")
    ops.gemm_bf16_bias(xs2, dev(w), None, out2, ln=ops.ln_consumer(st2, dev(c), dev(dd), N, rm), tile_hint=hint)
    close(rm, xn.mean(1), 1e-4, 1e-5, "row mean moved along by the consumer")
    ref2 = (F.layer_norm(xn, (D,), eps=1e-6) * (1 + scale) + shift) @ wf.T + b
    assert float((out2.cpu() - ref2).pow(2).mean().sqrt()) / float(ref2.std()) < 4e-3


# ------------------------------------------------------------------ role-split GEMMs (round 4): every instantiation by shape

@pytest.mark.parametrize("S,N,H,K", [
    (2, 469, 16, 1024),    # C2: 15 x 16 tiles of 64 x 192 on 256 CUs -> role split (12 consumer + 4 loader waves)
    (3, 150, 12, 768),     # 8 x 12 wide tiles, D = 768 (the Small PPG model's widths)
    (1, 1, 4, 256),        # a single row: one row tile, K = 4 tiles (the ring's depth)
    (2, 131, 5, 512),      # odd rows per sequence, heads not a multiple of 4
    (2, 1300, 16, 1024),   # 41 x 16 wide tiles, 21 x 16 of 128 x 192: past one round either way -> the classic 64 x 64 consumer
    (2, 640, 16, 1024),    # M = 1280: 10 x 16 tiles of 128 x 192 (12 consumer waves of 64 x 32, one fragment set)
    (2, 1024, 16, 1024),   # M = 2048: 16 x 16 = 256 tiles, the whole chip
    (4, 333, 12, 768),     # M = 1332, D = 768: 11 x 12 tiles, odd rows per sequence
])
def test_fused_qkv_rope_consumer_role_split_and_classic(ops, S, N, H, K):
    """f5e_gemm_bf16_qkv_rope_ln (fused AdaLN consumer + RoPE + fragment-major q / k / v^T): against the fp32 reference
    LN(x)(1+scale)+shift -> linear -> RoPE (reference modules.py:308-314 + :452-480) and against the UNFUSED HIP ops
    (f5e_layernorm + f5e_gemm_bf16_qkv_rope), at shapes that select the wide role-split kernel and the classic one."""
    D, M, inner, P = K, S * N, H * 64, K // 64
    n_pad = (N + 63) // 64 * 64
    x = torch.randn(M, D, generator=g(80)) * 1.3 + 0.2
    mod = torch.randn(1, 2 * D, generator=g(81)) * 0.3
    scale, shift = mod[:, :D], mod[:, D:]
    w = (torch.randn(3 * inner, D, generator=g(82)) / math.sqrt(D)).to(BF)
    b = torch.randn(3 * inner, generator=g(83)) * 0.1
    wf = w.float()
    lin = ((F.layer_norm(x, (D,), eps=1e-6) * (1 + scale) + shift) @ wf.T + b).view(S, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    freqs = O.rope_freqs(N, 64)
    q_ref, k_ref, v_ref = O.apply_rope(lin[0], freqs), O.apply_rope(lin[1], freqs), lin[2]
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    cs = torch.empty(N, 32, 2, device="cuda")
    ops.rope_table(dev(inv), cs)
    c = ((1 + scale) @ wf.T).contiguous()
    dd = (shift @ wf.T + b).contiguous()
    xd, modd = dev(x), dev(mod)
    xs = torch.empty(M, D, device="cuda", dtype=BF)
    stats = torch.empty(M, P, 2, device="cuda")
    rm = torch.empty(M, device="cuda")
    ops.adaln_pre(xd, xs, modd[:, :D], stats, rm, N)

    def run(a, bias, ln):
        q = torch.zeros(S, H, n_pad, 64, device="cuda", dtype=BF)
        k = torch.zeros_like(q)
        vt = torch.zeros(S, H, 64, n_pad, device="cuda", dtype=BF)
        ops.gemm_bf16_qkv_rope(a, dev(w), bias, q, k, vt, H, H, cs, N, ln=ln)
        return q, k, vt

    qf, kf, vf = run(xs, None, ops.ln_consumer(stats, dev(c), dev(dd), N, rm))
    hn = torch.empty(M, D, device="cuda", dtype=BF)
    ops.layernorm(xd, hn, scale=modd[:, :D], shift=modd[:, D:], rows_per_seq=N)
    qu, ku, vu = run(hn, dev(b), None)
    qi, vi = ops.qk_frag_index(n_pad), ops.v_frag_index(n_pad)
    unq = lambda t, idx: t.cpu().float().view(S, H, -1)[:, :, idx]   # noqa: E731
    for name, fused, unf, ref, idx, sc in (("q", qf, qu, q_ref, qi, QSCALE), ("k", kf, ku, k_ref, qi, 1.0), ("v", vf, vu, v_ref, vi, 1.0)):
        got, base, want = unq(fused, idx)[:, :, :N], unq(unf, idx)[:, :, :N], ref * sc
        spread = float(want.std())
        e_f, e_u = float((got - want).abs().max()) / spread, float((base - want).abs().max()) / spread
        assert e_f < 4e-2 and e_f < 2.5 * e_u + 2e-3, (name, e_f, e_u)     # bf16-level, on par with the unfused path
        assert float((got - want).pow(2).mean().sqrt()) / spread < 5e-3, name
        if n_pad > N:
            assert float(unq(fused, idx)[:, :, N:].abs().max()) == 0         # pad positions are never written


@pytest.mark.parametrize("S,N,D,NO,K_unused", [
    (2, 469, 1024, 2048, 0),   # C2's FF1: 15 x 16 tiles of 64 x 128, two K-tiles per hand-over (K % 128 == 0)
    (1, 77, 256, 256, 0),      # K = 256: four K-tiles = less than the ring of six, two row tiles
    (3, 150, 768, 1536, 0),    # the Small PPG model's FF1
    (2, 469, 1024, 2176, 0),   # 17 column tiles x 15 = 255: still one round
    (2, 469, 1024, 2304, 0),   # 18 x 15 = 270 > 256: the classic consumer
    (2, 900, 1024, 2048, 0),   # M = 1800: 15 x 16 tiles of 128 x 128 (8 consumer waves of 64 x 32)
    (3, 600, 768, 1536, 0),    # M = 1800, D = 768: 15 x 12 tiles
    (2, 1100, 1024, 2048, 0),  # M = 2200: 18 x 16 > 256 -> the classic consumer
])
def test_fused_gelu_consumer_role_split_and_classic(ops, S, N, D, NO, K_unused):
    """f5e_gemm_bf16_bias_ln with GELU(tanh) (FF1 behind the fused AdaLN, reference modules.py:637 + :348-349) on the wide
    role-split tiles and on the classic ones: against fp32 and against the unfused HIP ops."""
    M, P = S * N, D // 64
    x = torch.randn(M, D, generator=g(84)) * 1.2 - 0.4
    mod = torch.randn(1, 2 * D, generator=g(85)) * 0.3
    scale, shift = mod[:, :D], mod[:, D:]
    w = (torch.randn(NO, D, generator=g(86)) / math.sqrt(D)).to(BF)
    b = torch.randn(NO, generator=g(87)) * 0.1
    wf = w.float()
    ref = F.gelu((F.layer_norm(x, (D,), eps=1e-6) * (1 + scale) + shift) @ wf.T + b, approximate="tanh")
    c, dd = ((1 + scale) @ wf.T).contiguous(), (shift @ wf.T + b).contiguous()
    xd, modd = dev(x), dev(mod)
    xs = torch.empty(M, D, device="cuda", dtype=BF)
    stats = torch.empty(M, P, 2, device="cuda")
    rm = torch.empty(M, device="cuda")
    ops.adaln_pre(xd, xs, modd[:, :D], stats, rm, N)
    out = torch.empty(M, NO, device="cuda", dtype=BF)
    ops.gemm_bf16_bias(xs, dev(w), None, out, act=ops.ACT_GELU_TANH, ln=ops.ln_consumer(stats, dev(c), dev(dd), N, rm))
    hn = torch.empty(M, D, device="cuda", dtype=BF)
    ops.layernorm(xd, hn, scale=modd[:, :D], shift=modd[:, D:], rows_per_seq=N)
    unf = torch.empty(M, NO, device="cuda", dtype=BF)
    ops.gemm_bf16_bias(hn, dev(w), dev(b), unf, act=ops.ACT_GELU_TANH)
    spread = float(ref.std())
    e_f, e_u = float((out.cpu().float() - ref).abs().max()) / spread, float((unf.cpu().float() - ref).abs().max()) / spread
    assert e_f < 4e-2 and e_f < 2.5 * e_u + 2e-3, (e_f, e_u)
    assert float((out.cpu().float() - ref).pow(2).mean().sqrt()) / spread < 6e-3


@pytest.mark.parametrize("M,N,K,rps", [
    (938, 1024, 1024, 469),   # out-projection at C2: ring of 6, two K-tiles per hand-over
    (938, 1024, 2048, 469),   # FF2 at C2
    (200, 256, 192, 50),      # K = 192: one tile per hand-over, 4-stage ring, 4 sequences
    (64, 64, 64, 64),         # a single tile, a single K-tile
    (1, 128, 128, 1),         # one row
    (130, 1024, 320, 65),     # K = 320 (5 tiles: odd count), a partial last row tile
])
def test_gate_residual_role_split_matches_reference(ops, M, N, K, rps):
    """The one-round gate+residual GEMM (role split) without the AdaLN producer: x += gate * (a @ w.T + b) in fp32, masked
    rows untouched (reference modules.py:494-501, 635, 639)."""
    S = M // rps
    a = torch.randn(M, K, generator=g(88)).to(BF)
    w = (torch.randn(N, K, generator=g(89)) / math.sqrt(K)).to(BF)
    b = torch.randn(N, generator=g(90)) * 0.1
    gate = torch.randn(1, N, generator=g(91)) * 0.5
    x = torch.randn(M, N, generator=g(92))
    seq_len = torch.tensor([max(1, rps - 3 * i) for i in range(S)], dtype=torch.int32)
    live = (torch.arange(M) % rps)[:, None] < seq_len[torch.arange(M) // rps][:, None]
    ref = torch.where(live, x + gate * (a.float() @ w.float().T + b), x)
    xd = dev(x)
    ops.gemm_bf16_gate_residual(dev(a), dev(w), dev(b), xd, dev(gate), rps, seq_len=dev(seq_len))
    close(xd, ref, 1e-3, 2e-3, "gated residual update")
    assert torch.equal(xd.cpu()[~live.expand_as(x)], x[~live.expand_as(x)])   # masked rows: bit-identical
