from .backbones.dit import DiT
from .backbones.unett import UNetT
from .cfm import CFM
from .modules import DiTBlock, MelSpec

__all__ = ["CFM", "DiT", "UNetT", "DiTBlock", "MelSpec"]
