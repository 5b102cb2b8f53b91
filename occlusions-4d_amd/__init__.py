"""occlusions-4d hot path (point-transformer encode + cross-attention implicit
decode of 4D query points) as hand-written HIP for MI355X / gfx950.

Host side mirrors the reference's ``model/`` + ``eval/inference.py`` interface
(same class names, constructor kwargs, parameter names and forward signatures,
SURVEY.md §8(b)); every forward runs on the C-ABI library ``libocc4d.so``
(include/occ4d.h) and raises if it is missing -- there is no CPU fallback.
"""
from . import configs  # noqa: F401  (pure host code; no native dependency)

__all__ = ['configs']
