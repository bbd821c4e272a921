"""CFM sampler mirror (reference model/cfm.py:34-482): ``sample`` / ``sample_tts`` / ``sample_vc`` with the reference's
signatures and return values ``(out, trajectory)``.  Prep (masks, padding, duration clamp, seeded noise, sway grid)
is index/bookkeeping work done with torch on the host or device; the mel front-end, every network evaluation, the CFG
combination and the ODE update run in libf5e_hip.so (engine.run_ode).  Training ``forward`` is out of scope.

Documented deviations from the reference-on-GPU (both are identical to the reference-on-CPU):
  * seeded noise is drawn with the CPU generator in fp32 and uploaded;
  * the time grid / RoPE angles are always fp32.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import _C
from ..engine import SamplerInputs, h2d, ode_setup, run_ode
from .modules import MelSpec
from .utils import default, exists, intersperse, lens_to_mask, list_str_to_idx, list_str_to_tensor

F32, I32 = torch.float32, torch.int32


class CFM(nn.Module):
    def __init__(self, transformer: nn.Module, sigma=0.0, odeint_kwargs: dict = dict(method="euler"),
                 audio_drop_prob=0.3, cond_drop_prob=0.2, num_channels=None, mel_spec_module: nn.Module | None = None,
                 mel_spec_kwargs: dict = dict(), frac_lengths_mask: tuple[float, float] = (0.7, 1.0),
                 vocab_char_map: dict[str, int] | None = None, ppg_config=dict(use_ppg=False),
                 cb_config=dict(use_codebook=False)):
        super().__init__()
        self.frac_lengths_mask = frac_lengths_mask
        self.mel_spec = default(mel_spec_module, MelSpec(**mel_spec_kwargs))
        self.num_channels = default(num_channels, self.mel_spec.n_mel_channels)
        self.audio_drop_prob, self.cond_drop_prob = audio_drop_prob, cond_drop_prob
        self.transformer = transformer
        self.dim = transformer.dim
        self.sigma = sigma
        self.odeint_kwargs = odeint_kwargs
        self.vocab_char_map = vocab_char_map
        self.use_ppg = ppg_config["use_ppg"]
        self.use_cross_mask = ppg_config.get("use_cross_mask", False)
        if self.use_ppg:
            self.combined_cond_drop_prob = ppg_config.get("combined_cond_drop_prob")
        self.use_codebook = cb_config["use_codebook"]
        self.use_align_loss = cb_config.get("use_align_loss", False)
        self.use_graph = True  # hipGraph replay of the ODE step; set False to launch eagerly (debugging)
        self.chains = None  # None = auto (parallel CFG-branch chains at small batch); 1 = always one batched forward
        self.kernel_timer = None  # engine.KernelTimer: per-launch HIP-event timing of one op class (eager mode only)
        import threading
        self._tls = threading.local()  # per-thread capture stream: callers may sample concurrently

    @property
    def device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ shared prep (cfm.py:368-428, 449-469)

    def _prepare(self, cond, text, duration, lens, steps, sway_sampling_coef, seed, max_duration, no_ref_audio,
                 duplicate_test, t_inter, edit_mask, use_text):
        if self.training:  # Module.eval() walks ~2500 sub-modules (2 ms): only when it changes something
            self.eval()
        dv = self.device
        if dv.type != "cuda":
            raise _C.F5EError(f"CFM lives on {dv}: move it to the GPU (there is no CPU path)")
        if cond.ndim == 2:  # raw wave -> log-mel through the HIP STFT kernel
            cond = self.mel_spec(cond.to(dv)).permute(0, 2, 1)
            assert cond.shape[-1] == self.num_channels
        cond = cond.to(dv, F32)
        batch, cond_seq_len = cond.shape[:2]
        # Lengths, durations and masks are worked out on the HOST: nothing here waits for the GPU unless the caller
        # hands lens / duration / token ids as device tensors, so the prep of one call overlaps the ODE loop of the
        # previous one (the reference syncs at cfm.py:412 `duration.amax()`).
        lens_h = lens.detach().to("cpu", torch.long) if exists(lens) else torch.full((batch,), cond_seq_len, dtype=torch.long)
        if use_text and isinstance(text, list):
            if exists(self.vocab_char_map):
                if self.use_align_loss or self.use_cross_mask:
                    text = intersperse(text)
                text = list_str_to_idx(text, self.vocab_char_map)
            else:
                text = list_str_to_tensor(text)
            assert text.shape[0] == batch
        if not use_text:
            text = None
        cond_mask = lens_to_mask(lens_h)
        if isinstance(duration, int):
            dur_h = torch.full((batch,), duration, dtype=torch.long)
        else:
            dur_h = duration.detach().to("cpu", torch.long)
        if text is not None:
            text_h = text.detach().to("cpu")
            rows = getattr(getattr(self.transformer, "text_embed", None), "text_embed", None)
            if rows is not None and text_h.numel() and int(text_h.max()) + 1 >= rows.num_embeddings:
                # nn.Embedding raises here in the reference (vocab file / checkpoint mismatch); the HIP gather would clamp
                raise IndexError(f"token id {int(text_h.max())} out of range for a text embedding of "
                                 f"{rows.num_embeddings} rows (ids are shifted by +1, backbones/dit.py:55)")
            dur_h = torch.maximum(torch.maximum((text_h != -1).sum(dim=-1), lens_h) + 1, dur_h)
            text = h2d(text, dv)
        else:
            dur_h = torch.maximum(lens_h + 1, dur_h)
        dur_h = dur_h.clamp(max=max_duration)
        n = int(dur_h.amax())
        test_cond = None
        if duplicate_test:
            test_cond = F.pad(cond, (0, 0, cond_seq_len, n - 2 * cond_seq_len), value=0.0)
        cond = F.pad(cond, (0, 0, 0, n - cond_seq_len), value=0.0)
        if no_ref_audio:
            cond = torch.zeros_like(cond)
        if edit_mask is not None:
            cond_mask = h2d(cond_mask, dv) & edit_mask.to(dv)
        cond_mask = h2d(F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False), dv).unsqueeze(-1)
        from .. import ops
        cond = cond.contiguous()
        step_cond = ops.stitch(cond, torch.zeros_like(cond), cond_mask.reshape(-1).to(torch.uint8).contiguous(),
                               torch.empty_like(cond))    # where(cond_mask, cond, 0), cfm.py:423
        seq_len = dur_h.to(I32) if batch > 1 else None  # reference: mask = None for single inference (cfm.py:425-428)
        # seeded noise, one item at a time, CPU generator (cfm.py:452-457)
        y0 = torch.zeros(batch, n, self.num_channels, dtype=F32)
        for i, dur in enumerate(dur_h.tolist()):
            if exists(seed):
                # same stream as the reference's `torch.manual_seed(seed); torch.randn(...)` on the default CPU generator,
                # but private to this call: callers sample from several threads on one model (utils_infer.py:511) and
                # the seed + draw pair on the global generator is not atomic
                gen = torch.Generator().manual_seed(seed)
                y0[i, :dur] = torch.randn(dur, self.num_channels, dtype=F32, generator=gen)
            else:
                y0[i, :dur] = torch.randn(dur, self.num_channels, dtype=F32)
        y0 = h2d(y0, dv)
        t_start = 0
        if duplicate_test:
            t_start = t_inter
            y0 = ops.axpby(y0, test_cond.contiguous(), torch.empty_like(y0), 1 - t_start, t_start)   # cfm.py:463
            steps = int(steps * (1 - t_start))
        t = torch.linspace(t_start, 1, steps + 1, dtype=F32)
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
        return dict(cond=cond, cond_mask=cond_mask, step_cond=step_cond, text=text, seq_len=seq_len, y0=y0, t=t)

    def _integrate(self, prep, ppg, branches, mode, w0, w1, vocoder):
        eng = self.transformer.engine()
        inp = SamplerInputs(step_cond=prep["step_cond"], text=prep["text"], ppg=ppg, y0=prep["y0"], t=prep["t"],
                            seq_len=prep["seq_len"], branches=branches, mode=mode, w0=w0, w1=w1,
                            method=self.odeint_kwargs.get("method", "euler"))
        if self.use_graph:
            # graph CAPTURE needs a non-default stream (one per thread); what runs -- eager steps, graph launches, the
            # once-per-call kernels -- stays on the caller's current stream, as every other op of this package does
            side = getattr(self._tls, "stream", None)
            if side is None:
                side = self._tls.stream = torch.cuda.Stream(device=eng.device)
            setup = ode_setup(eng, inp)   # pinned uploads + shared tables on the caller's stream, see ode_setup
            trajectory = run_ode(eng, inp, use_graph=True, chains=self.chains, setup=setup, capture_stream=side)
        else:
            trajectory = run_ode(eng, inp, use_graph=False, timer=self.kernel_timer, chains=self.chains)
        self.transformer.clear_cache()
        out = torch.empty_like(trajectory[-1])
        from .. import ops
        ops.stitch(prep["cond"].contiguous(), trajectory[-1], prep["cond_mask"].reshape(-1).to(torch.uint8).contiguous(),
                   out)
        if exists(vocoder):
            out = vocoder(out.permute(0, 2, 1))
        return out, trajectory

    # ------------------------------------------------------------------ public samplers

    @torch.no_grad()
    def sample(self, cond, text, ppg=None, duration=None, *, lens=None, steps=32, cfg_strength=1.0,
               sway_sampling_coef=None, seed: Optional[int] = None, max_duration=4096,
               vocoder: Optional[Callable] = None, no_ref_audio=False, duplicate_test=False, t_inter=0.1,
               edit_mask=None):
        """reference model/cfm.py:349-482: pred + (pred - null) * cfg_strength, Euler/midpoint on the sway grid.

        Streams: everything this call launches runs on the CALLER's current stream (a private side stream only captures
        graphs).  Calls from several threads on the default stream are correct but serialise; to overlap independent
        utterances give each thread its own stream (`with torch.cuda.stream(s): cfm.sample(...)`, or
        eval.eval_infer_batch.RankWorkers) -- loop states and memory pools are keyed by (thread, stream)."""
        prep = self._prepare(cond, text, duration, lens, steps, sway_sampling_coef, seed, max_duration, no_ref_audio,
                             duplicate_test, t_inter, edit_mask, use_text=True)
        if cfg_strength < 1e-5:
            return self._integrate(prep, ppg, [(False, False, False)], 0, 0.0, 0.0, vocoder)
        return self._integrate(prep, ppg, [(False, False, False), (True, True, True)], 1, float(cfg_strength), 0.0,
                               vocoder)

    @torch.no_grad()
    def sample_tts(self, cond, text, duration=None, *, lens=None, steps=32, alpha_spk=1.0, alpha_txt=1.0,
                   sway_sampling_coef=None, seed: Optional[int] = None, max_duration=4096,
                   vocoder: Optional[Callable] = None, no_ref_audio=False, duplicate_test=False, t_inter=0.1,
                   edit_mask=None):
        """reference model/cfm.py:94-223: alpha_spk (spk_txt - txt) + alpha_txt (txt - null) + null."""
        prep = self._prepare(cond, text, duration, lens, steps, sway_sampling_coef, seed, max_duration, no_ref_audio,
                             duplicate_test, t_inter, edit_mask, use_text=True)
        branches = [(True, True, True), (True, False, True), (False, False, True)]  # null, txt, spk_txt
        return self._integrate(prep, None, branches, 2, float(alpha_spk), float(alpha_txt), vocoder)

    @torch.no_grad()
    def sample_vc(self, cond, ppg=None, duration=None, *, lens=None, steps=32, alpha_spk=1.0, alpha_ppg=1.0,
                  sway_sampling_coef=None, seed: Optional[int] = None, max_duration=4096,
                  vocoder: Optional[Callable] = None, no_ref_audio=False, duplicate_test=False, t_inter=0.1,
                  edit_mask=None):
        """reference model/cfm.py:226-346: alpha_spk (spk_ppg - ppg) + alpha_ppg (ppg - null) + null, text=None."""
        prep = self._prepare(cond, None, duration, lens, steps, sway_sampling_coef, seed, max_duration, no_ref_audio,
                             duplicate_test, t_inter, edit_mask, use_text=False)
        branches = [(True, True, True), (True, True, False), (False, True, False)]  # null, ppg, spk_ppg
        return self._integrate(prep, ppg, branches, 2, float(alpha_spk), float(alpha_ppg), vocoder)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("flow-matching training loss (reference model/cfm.py:484-590) is out of scope of "
                                  "the MI355X inference path (SURVEY section 8)")
