"""occlusions-4d hot path (point-transformer encode + cross-attention implicit
decode of 4D query points) as hand-written HIP for MI355X / gfx950.

Host side mirrors the reference's ``model/`` + ``eval/inference.py`` interface
(same module / class names, constructor kwargs, parameter names and forward
signatures); every forward runs on the C-ABI library
``libocc4d.so`` (include/occ4d.h) and raises if it is missing -- there is no CPU
fallback.  ``configs`` is pure host code (named configurations, synthetic inputs).
"""
from . import configs  # noqa: F401
from . import _lib  # noqa: F401
from . import kernels  # noqa: F401   (`with occlusions4d_amd.kernels(precision='bf16x6'):` -- per-thread kernel selection)
from . import ops  # noqa: F401
from . import autograd  # noqa: F401
from . import point_transformer_layer, modules, model, geometry, implicit, inference, distributed, training  # noqa: F401
from . import evaluation  # noqa: F401
from . import cpu_twin  # noqa: F401   (explicit opt-in only: pk.cpu_twin.enable(); never a fallback)

__all__ = ['configs', 'kernels', 'ops', 'point_transformer_layer', 'modules', 'model', 'geometry', 'implicit',
           'inference', 'distributed', 'autograd', 'training', 'evaluation']
