"""Mirror of reference train/parse_cfg.py: splits the yaml ``ppg_config`` / ``codebook_config`` blocks into the dicts
handed to the backbone (``DiT(ppg_config=, cb_config=)``), to ``CFM`` and to the PPG front-end.  Works on plain dicts
(yaml.safe_load) as well as attribute-style configs."""
from __future__ import annotations

from typing import Any, Dict, Tuple


def _get(cfg: Any, key: str, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _need(cfg: Any, key: str):
    v = _get(cfg, key, KeyError)
    if v is KeyError:
        raise KeyError(f"config block lacks required key {key!r}")
    return v


def parse_ppg_config(ppg_cfg) -> Tuple[Dict, Dict, Dict]:
    """-> (transformer_ppg_config, cfm_ppg_config, trainer_ppg_config)   (reference parse_cfg.py:1-47)"""
    use_cross_mask = bool(_get(ppg_cfg, "use_cross_mask", False))
    cross_mask_config = {}
    if use_cross_mask:
        cross_mask_config = dict(cross_mask_prob=_need(_need(ppg_cfg, "cross_mask_config"), "cross_mask_prob"))
    use_transformer = bool(_get(ppg_cfg, "use_transformer", False))
    transformer_config = {}
    if use_transformer:
        tc = _need(ppg_cfg, "transformer_config")
        transformer_config = {k: _need(tc, k) for k in ("num_layers", "nhead", "dim_feedforward", "dropout")}
    transformer_ppg_config = dict(use_ppg=True, ppg_dim=_need(ppg_cfg, "dim"), use_cross_mask=use_cross_mask,
                                  cross_mask_config=cross_mask_config, use_transformer=use_transformer,
                                  transformer_config=transformer_config)
    cfm_ppg_config = dict(use_ppg=True, combined_cond_drop_prob=_need(ppg_cfg, "combined_cond_drop_prob"),
                          use_cross_mask=use_cross_mask)
    m = _need(ppg_cfg, "map")
    trainer_ppg_config = dict(use_ppg=True, model_path=_need(ppg_cfg, "model_path"), config=_need(ppg_cfg, "config"),
                              frame_length=_need(ppg_cfg, "frame_length"),
                              mel_frame_shift=_need(ppg_cfg, "mel_frame_shift"), dim=_need(ppg_cfg, "dim"),
                              output_type=_need(ppg_cfg, "output_type"), map_mix_ratio=_need(m, "map_mix_ratio"),
                              global_phn_center_path=_need(m, "global_phn_center_path"),
                              para_softmax_path=_need(m, "para_softmax_path"))
    return transformer_ppg_config, cfm_ppg_config, trainer_ppg_config


def parse_codebook_config(codebook_cfg) -> Tuple[Dict, Dict]:
    """-> (transformer_codebook_config, cfm_codebook_config)   (reference parse_cfg.py:49-90)"""
    use_perplex_loss = bool(_get(codebook_cfg, "use_perplex_loss", False))
    perplex = {}
    if use_perplex_loss:
        pc = _need(codebook_cfg, "perplex_loss_config")
        perplex = dict(perplex_loss_prob=_need(pc, "perplex_loss_prob"), perplex_loss_weight=_need(pc, "perplex_loss_weight"))
    use_align_loss = bool(_get(codebook_cfg, "use_align_loss", False))
    align = {}
    if use_align_loss:
        align = dict(align_loss_weight=_need(_need(codebook_cfg, "align_loss_config"), "align_loss_weight"))
    transformer = dict(use_codebook=True,
                       **{k: _need(codebook_cfg, k) for k in ("num_vars", "temp_start", "temp_stop", "temp_decay", "groups",
                                                             "combine_groups", "weight_proj_depth", "weight_proj_factor")},
                       use_perplex_loss=use_perplex_loss, perplex_loss_config=perplex, use_align_loss=use_align_loss,
                       align_loss_config=align)
    return transformer, dict(use_codebook=True, use_align_loss=use_align_loss)


def parse_model_yaml(cfg: Dict) -> Dict:
    """The pieces the inference drivers need from a model yaml (reference eval/eval_infer_batch_tts.py:86-104):
    arch kwargs, and the backbone / CFM / front-end PPG and codebook dicts."""
    model = cfg["model"]
    arch = dict(model["arch"])
    arch.pop("checkpoint_activations", None)
    off = dict(use_ppg=False)
    t_ppg, c_ppg, f_ppg = parse_ppg_config(model["ppg_config"]) if model.get("use_ppg", False) else (off, off, off)
    if model.get("use_codebook", False):
        if not model.get("use_ppg", False):
            raise ValueError("use_codebook needs use_ppg (reference eval_infer_batch_tts.py:98-102)")
        t_cb, c_cb = parse_codebook_config(model["codebook_config"])
    else:
        t_cb = c_cb = dict(use_codebook=False)
    return dict(arch=arch, transformer_ppg_config=t_ppg, cfm_ppg_config=c_ppg, frontend_ppg_config=f_ppg,
                transformer_codebook_config=t_cb, cfm_codebook_config=c_cb, mel_spec=dict(model.get("mel_spec", {})))
