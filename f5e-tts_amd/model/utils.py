"""Host-side tensor helpers of the sampler (mirror of reference model/utils.py:31-100 and durpred/utils.py:10-17).
Integer / bool index preparation only -- no floating-point compute happens here."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch


def exists(v) -> bool:
    return v is not None


def default(v, d):
    return v if v is not None else d


def lens_to_mask(t: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
    """bool [b, n]: position < length (reference model/utils.py:41-46)."""
    if length is None:
        length = int(t.amax())
    return torch.arange(length, device=t.device)[None, :] < t[:, None]


def _pad_rows(rows: List[List[int]], padding_value: int) -> torch.Tensor:
    width = max((len(r) for r in rows), default=0)
    out = torch.full((len(rows), width), padding_value, dtype=torch.long)
    for i, r in enumerate(rows):
        if r:
            out[i, : len(r)] = torch.tensor(r, dtype=torch.long)
    return out


def list_str_to_tensor(text: Sequence[str], padding_value: int = -1) -> torch.Tensor:
    """UTF-8 byte tokeniser (reference model/utils.py:80-83)."""
    return _pad_rows([list(bytes(t, "UTF-8")) for t in text], padding_value)


def list_str_to_idx(text: Sequence, vocab_char_map: Dict[str, int], padding_value: int = -1) -> torch.Tensor:
    """Vocabulary lookup, unknown -> 0, right-padded with -1 (reference model/utils.py:87-100)."""
    return _pad_rows([[vocab_char_map.get(c, 0) for c in t] for t in text], padding_value)


def intersperse(text: Sequence, sep: str = "_") -> List[List[str]]:
    """sep between (and around) every token (reference durpred/utils.py:10-17); only used with align-loss models."""
    out = []
    for sentence in text:
        row = [sep] * (len(sentence) * 2 + 1)
        row[1::2] = list(sentence)
        out.append(row)
    return out


def get_tokenizer(vocab_file: str):
    """'custom' tokenizer branch of reference model/utils.py:136-175: one token per line, index = line number."""
    vocab_char_map = {}
    with open(vocab_file, "r", encoding="utf-8") as f:
        for i, char in enumerate(f):
            vocab_char_map[char[:-1]] = i
    return vocab_char_map, len(vocab_char_map)
