"""DiT backbone mirror (reference model/backbones/dit.py:37-283,417-472): same constructor signature, attributes
(``.dim``, ``.depth``), ``state_dict`` keys and ``sample`` / ``clear_cache`` methods.  The members are parameter
containers; the computation is ``engine.DiTEngine`` on libf5e_hip.so.  Training (``forward``) is out of scope."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ... import _C
from ...engine import DiTConfig, DiTEngine
from ..modules import (AdaLayerNorm_Final, ConvNeXtV2Block, ConvPositionEmbedding, DiTBlock, GumbelVectorQuantizer,
                       TimestepEmbedding)

F32, I32 = torch.float32, torch.int32


class TextEmbedding(nn.Module):
    def __init__(self, text_num_embeds, text_dim, mask_padding=True, conv_layers=0, conv_mult=2):
        super().__init__()
        self.text_embed = nn.Embedding(text_num_embeds + 1, text_dim)  # index 0 = filler token
        self.mask_padding = mask_padding
        self.extra_modeling = conv_layers > 0
        if self.extra_modeling:
            self.precompute_max_pos = 4096
            self.text_blocks = nn.Sequential(
                *[ConvNeXtV2Block(text_dim, text_dim * conv_mult) for _ in range(conv_layers)])


class PPGInputTranspose(nn.Identity):
    pass


class PPGEmbedding(nn.Module):
    """reference backbones/dit.py:93-138: the conv variant (keys ppg_proj.{0,2,3,6,7,10,11,15}) or, with
    ``use_transformer``, nn.TransformerEncoder + Linear (keys ppg_proj.0.layers.{i}.*, ppg_proj.1.*).  Parameter
    containers only: the computation is engine.DiTEngine.ppg_embed on libf5e_hip.so."""

    def __init__(self, ppg_dim, text_dim, use_transformer=False, transformer_config=dict()):
        super().__init__()
        self.ppg_dim, self.text_dim, self.use_transformer = ppg_dim, text_dim, use_transformer
        if use_transformer:
            import torch.nn.functional as F
            self.nhead = transformer_config["nhead"]
            self.ppg_proj = nn.Sequential(
                nn.TransformerEncoder(
                    nn.TransformerEncoderLayer(d_model=ppg_dim, nhead=transformer_config["nhead"],
                                               dim_feedforward=transformer_config["dim_feedforward"],
                                               dropout=transformer_config["dropout"], activation=F.gelu, batch_first=True),
                    num_layers=transformer_config["num_layers"], enable_nested_tensor=False),
                nn.Linear(ppg_dim, text_dim))
            return
        layers = [nn.Linear(ppg_dim, ppg_dim), PPGInputTranspose()]
        for _ in range(3):
            layers += [nn.Conv1d(ppg_dim, ppg_dim, kernel_size=5, padding="same"), nn.BatchNorm1d(ppg_dim), nn.ReLU(),
                       nn.Dropout(0.5)]
        layers += [PPGInputTranspose(), nn.Linear(ppg_dim, text_dim)]
        self.ppg_proj = nn.Sequential(*layers)


class InputEmbedding(nn.Module):
    def __init__(self, mel_dim, text_dim, out_dim, use_ppg):
        super().__init__()
        self.use_ppg = use_ppg
        self.proj = nn.Linear(mel_dim * 2 + text_dim * (2 if use_ppg else 1), out_dim)
        self.conv_pos_embed = ConvPositionEmbedding(dim=out_dim)


class RotaryEmbedding(nn.Module):
    """Container for x_transformers' persistent ``inv_freq`` buffer (a checkpoint key)."""

    def __init__(self, dim, theta=10000.0):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim)))


class DiT(nn.Module):
    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100,
                 text_num_embeds=256, text_dim=None, text_mask_padding=True, qk_norm=None, conv_layers=0,
                 pe_attn_head=None, long_skip_connection=False, checkpoint_activations=False,
                 ppg_config=dict(use_ppg=False), cb_config=dict(use_codebook=False)):
        super().__init__()
        self.time_embed = TimestepEmbedding(dim)
        if text_dim is None:
            text_dim = mel_dim
        self.text_embed = TextEmbedding(text_num_embeds, text_dim, mask_padding=text_mask_padding,
                                        conv_layers=conv_layers)
        self.text_cond, self.text_uncond = None, None
        self.use_ppg = bool(ppg_config["use_ppg"])
        if self.use_ppg:
            self.ppg_embed = PPGEmbedding(ppg_config["ppg_dim"], text_dim,
                                          use_transformer=ppg_config.get("use_transformer", False),
                                          transformer_config=ppg_config.get("transformer_config", dict()))
        self.use_codebook = bool(cb_config["use_codebook"])
        if self.use_codebook:
            # Owned for state_dict compatibility (strict checkpoint loads) and as the parity-only eval op; the reference
            # applies it in the training forward only, DiT.sample never touches it (dit.py:296-308)
            self.quantizer = GumbelVectorQuantizer(
                dim=text_dim, num_vars=cb_config["num_vars"],
                temp=(cb_config["temp_start"], cb_config["temp_stop"], cb_config["temp_decay"]),
                groups=cb_config["groups"], combine_groups=cb_config["combine_groups"], vq_dim=text_dim,
                time_first=True, weight_proj_depth=cb_config["weight_proj_depth"],
                weight_proj_factor=cb_config["weight_proj_factor"])
        self.input_embed = InputEmbedding(mel_dim, text_dim, dim, self.use_ppg)
        self.rotary_embed = RotaryEmbedding(dim_head)
        self.dim, self.depth = dim, depth
        self.transformer_blocks = nn.ModuleList(
            [DiTBlock(dim=dim, heads=heads, dim_head=dim_head, ff_mult=ff_mult, dropout=dropout, qk_norm=qk_norm,
                      pe_attn_head=pe_attn_head) for _ in range(depth)])
        self.long_skip_connection = nn.Linear(dim * 2, dim, bias=False) if long_skip_connection else None
        self.norm_out = AdaLayerNorm_Final(dim)
        self.proj_out = nn.Linear(dim, mel_dim)
        self.checkpoint_activations = checkpoint_activations
        self.cfg = DiTConfig(dim=dim, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, mel_dim=mel_dim,
                             text_num_embeds=text_num_embeds, text_dim=text_dim, text_mask_padding=text_mask_padding,
                             qk_norm=qk_norm, conv_layers=conv_layers, pe_attn_head=pe_attn_head,
                             long_skip_connection=long_skip_connection, use_ppg=self.use_ppg,
                             ppg_dim=ppg_config.get("ppg_dim", 256) if self.use_ppg else 256,
                             ppg_transformer=bool(self.use_ppg and ppg_config.get("use_transformer", False)),
                             ppg_heads=(ppg_config.get("transformer_config") or {}).get("nhead", 4))
        self._engine = None
        self._tensor_list = None
        self.initialize_weights()

    def initialize_weights(self):
        """AdaLN-zero init (reference backbones/dit.py:273-283)."""
        for block in self.transformer_blocks:
            nn.init.constant_(block.attn_norm.linear.weight, 0)
            nn.init.constant_(block.attn_norm.linear.bias, 0)
        nn.init.constant_(self.norm_out.linear.weight, 0)
        nn.init.constant_(self.norm_out.linear.bias, 0)
        nn.init.constant_(self.proj_out.weight, 0)
        nn.init.constant_(self.proj_out.bias, 0)

    def clear_cache(self):
        self.text_cond, self.text_uncond = None, None

    def invalidate_engine(self):
        """Call after REPLACING a Parameter object of a sub-module (in-place writes, .to(), load_state_dict are seen)."""
        self._tensor_list = None

    def _apply(self, fn, *a, **kw):
        self._tensor_list = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._tensor_list = None
        return super().load_state_dict(*a, **kw)

    # ------------------------------------------------------------------ HIP engine

    def engine(self) -> DiTEngine:
        """Repacked device weights; rebuilt when any parameter was replaced or written in place."""
        tensors = self._tensor_list
        if tensors is None:  # the module walk costs ~1 ms: done once, redone after _apply / load_state_dict
            tensors = self._tensor_list = list(self.parameters()) + list(self.buffers())
        dev = tensors[0].device
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + (str(dev),)
        if self._engine is None or self._engine[0] != sig:
            if dev.type != "cuda":
                raise _C.F5EError(f"DiT lives on {dev}: move it to the GPU (there is no CPU path)")
            self._engine = (sig, DiTEngine(self.state_dict(), self.cfg, dev))
        return self._engine[1]

    @torch.no_grad()
    def sample(self, x, cond, text, ppg, time, drop_audio_cond, drop_text, drop_ppg, mask: Optional[torch.Tensor] = None):
        """One velocity evaluation (reference backbones/dit.py:417-472).  x, cond [b, n, mel]; text int [b, nt] or
        None; time 0-dim or [b]; mask bool [b, n] (lens_to_mask form) or None -> pred [b, n, mel] f32."""
        eng = self.engine()
        B, N = x.shape[0], x.shape[1]
        dv = eng.device
        if time.ndim == 0:
            time = time.repeat(B)
        mod = eng.time_tables(time.to(dv, F32).view(1, B))  # [1, B, row_stride]
        if drop_text:
            if self.text_uncond is None:
                self.text_uncond = eng.text_embed(text, B, N, True)
            text_embed = self.text_uncond
        else:
            if self.text_cond is None:
                self.text_cond = eng.text_embed(text, B, N, False)
            text_embed = self.text_cond
        ppg_embed = eng.ppg_embed(ppg, B, N, drop_ppg) if self.use_ppg else None
        in_const = torch.empty(B * N, self.dim, device=dv)
        eng.input_const(cond.to(dv, F32).contiguous(), text_embed, ppg_embed, drop_audio_cond, in_const)
        seq_len = None
        if mask is not None:
            seq_len = mask.sum(-1).to(I32).contiguous()
            if not torch.equal(mask, torch.arange(N, device=mask.device)[None] < seq_len[:, None]):
                raise _C.F5EError("DiT.sample takes key-padding masks of the lens_to_mask form only")
        y = x.to(dv, F32).contiguous()
        plan = eng.make_plan(B, B, N, y, in_const, mod, None, eng.rope_table(N), seq_len)
        return eng.forward(plan).view(B, N, -1).clone()

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward (reference backbones/dit.py:474-549) is out of scope of the "
                                  "MI355X inference path (SURVEY section 8)")
