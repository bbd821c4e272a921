"""Vocos mel vocoder on libf5e_hip.so.

The reference calls the third-party ``vocos`` package (``Vocos.from_hparams(...).decode(mel)``, reference
infer/utils_infer.py:113-124,489); the package and its weights are not in the tree, so this is a restatement of the
published ``charactr/vocos-mel-24khz`` architecture with the package's ``state_dict`` key names:
embed Conv1d(100,512,7) -> LN -> 8 x ConvNeXt(512,1536, layer-scale gamma) -> LN -> Linear(512,1026) -> iSTFT head.
"parity unpinned" for the third-party arithmetic (no golden vectors exist in the reference); the HIP path is checked
against the oracle restatement and torch.istft."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import nn

from . import _C, ops
from .engine import fft_tables

F32 = torch.float32


class _ConvNeXtBlock(nn.Module):
    def __init__(self, dim, intermediate_dim, layer_scale_init_value):
        super().__init__()
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(intermediate_dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim))


class _Backbone(nn.Module):
    def __init__(self, input_channels, dim, intermediate_dim, num_layers):
        super().__init__()
        self.embed = nn.Conv1d(input_channels, dim, kernel_size=7, padding=3)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.convnext = nn.ModuleList([_ConvNeXtBlock(dim, intermediate_dim, 1.0 / num_layers)
                                       for _ in range(num_layers)])
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-6)


class _ISTFT(nn.Module):
    def __init__(self, n_fft):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))


class _Head(nn.Module):
    def __init__(self, dim, n_fft, hop_length):
        super().__init__()
        self.out = nn.Linear(dim, n_fft + 2)
        self.istft = _ISTFT(n_fft)
        self.n_fft, self.hop_length = n_fft, hop_length


class Vocos(nn.Module):
    def __init__(self, input_channels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024, hop_length=256):
        super().__init__()
        if n_fft != 1024 or dim % 256 or input_channels % 4:
            raise _C.F5EError("Vocos HIP path: n_fft = 1024, dim % 256 == 0, input_channels % 4 == 0")
        self.backbone = _Backbone(input_channels, dim, intermediate_dim, num_layers)
        self.head = _Head(dim, n_fft, hop_length)
        self._packed = None

    @classmethod
    def from_hparams(cls, config_path: str) -> "Vocos":
        """Reads the vocos config.yaml (backbone / head init_args)."""
        import yaml
        with open(config_path, "r") as f:
            cfg = yaml.safe_load(f)
        b = cfg.get("backbone", {}).get("init_args", {})
        h = cfg.get("head", {}).get("init_args", {})
        return cls(input_channels=b.get("input_channels", 100), dim=b.get("dim", 512),
                   intermediate_dim=b.get("intermediate_dim", 1536), num_layers=b.get("num_layers", 8),
                   n_fft=h.get("n_fft", 1024), hop_length=h.get("hop_length", 256))

    def _pack(self, dv):
        tensors = list(self.parameters())
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + (str(dv),)
        if self._packed is None or self._packed[0] != sig:
            f = lambda t: t.detach().to(dv, F32).contiguous()  # noqa: E731
            bb = self.backbone
            ew = f(bb.embed.weight)  # [dim][cin][7] -> [dim][7*cin] (tap-major, matches f5e_im2col)
            blocks = []
            for blk in bb.convnext:
                blocks.append(dict(dw_w=f(blk.dwconv.weight)[:, 0, :].t().contiguous(), dw_b=f(blk.dwconv.bias),
                                   ln_g=f(blk.norm.weight), ln_b=f(blk.norm.bias), w1=f(blk.pwconv1.weight),
                                   b1=f(blk.pwconv1.bias), w2=f(blk.pwconv2.weight), b2=f(blk.pwconv2.bias),
                                   gamma=f(blk.gamma)))
            win, tw = fft_tables(dv)
            self._packed = (sig, dict(
                embed_w=ew.permute(0, 2, 1).reshape(ew.shape[0], -1).contiguous(), embed_b=f(bb.embed.bias),
                norm_g=f(bb.norm.weight), norm_b=f(bb.norm.bias), blocks=blocks,
                fin_g=f(bb.final_layer_norm.weight), fin_b=f(bb.final_layer_norm.bias),
                out_w=f(self.head.out.weight), out_b=f(self.head.out.bias), window=win, twiddle=tw))
        return self._packed[1]

    @torch.no_grad()
    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [b, 100, t] on the GPU -> wav [b, hop * (t - 1)] (ISTFTHead padding='center')."""
        ops.require_device()
        if mel.ndim != 3:
            raise _C.F5EError("Vocos.decode expects [b, n_mels, t]")
        dv = mel.device
        w = self._pack(dv)
        B, Cin, T = mel.shape
        if T < 2:
            raise _C.F5EError("Vocos.decode needs at least 2 frames")
        x = mel.to(F32).transpose(1, 2).contiguous()  # token-major [b, t, 100]
        dim = w["embed_b"].shape[0]
        col = torch.empty(B, T, 7 * Cin, device=dv)
        ops.im2col(x, col, 7, 3)
        h = torch.empty(B * T, dim, device=dv)
        ops.gemm_f32(col.view(B * T, 7 * Cin), w["embed_w"], w["embed_b"], out=h)
        hn = torch.empty_like(h)
        ops.layernorm(h, hn, gamma=w["norm_g"], beta=w["norm_b"])
        h = hn
        c = torch.empty_like(h)
        n = torch.empty_like(h)
        inter = w["blocks"][0]["w1"].shape[0] if w["blocks"] else dim
        p1 = torch.empty(B * T, inter, device=dv)
        for blk in w["blocks"]:
            ops.dwconv7(h.view(B, T, dim), blk["dw_w"], blk["dw_b"], c.view(B, T, dim))
            ops.layernorm(c, n, gamma=blk["ln_g"], beta=blk["ln_b"])
            ops.gemm_f32(n, blk["w1"], blk["b1"], out=p1, act=ops.ACT_GELU_ERF)
            h_new = torch.empty_like(h)
            ops.gemm_f32(p1, blk["w2"], blk["b2"], out=h_new, ch_scale=blk["gamma"], addend=h)
            h = h_new
        ops.layernorm(h, n, gamma=w["fin_g"], beta=w["fin_b"])
        n_fft, hop = self.head.n_fft, self.head.hop_length
        z = torch.empty(B * T, n_fft + 2, device=dv)
        ops.gemm_f32(n, w["out_w"], w["out_b"], out=z)
        frames = torch.empty(B * T, n_fft, device=dv)
        wav = torch.empty(B, hop * (T - 1), device=dv)
        ops.istft_head(z, w["window"], w["twiddle"], frames, wav, B, T, n_fft, hop)
        return wav

    def forward(self, mel):
        return self.decode(mel)


def load_vocos(local_path: Optional[str], device) -> Vocos:
    """``config.yaml`` + ``pytorch_model.bin`` from a local vocos-mel-24khz directory (the reference's is_local branch,
    infer/utils_infer.py:113-124).  There is no network here: a missing directory is an error, not a download."""
    if not local_path or not os.path.isdir(local_path):
        raise FileNotFoundError(f"local Vocos checkpoint directory not found: {local_path!r} (no network access)")
    voc = Vocos.from_hparams(os.path.join(local_path, "config.yaml"))
    state = torch.load(os.path.join(local_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    state = {k: v for k, v in state.items() if not k.startswith("feature_extractor.")}
    voc.load_state_dict(state)
    return voc.eval().to(device)
