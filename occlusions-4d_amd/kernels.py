"""Which kernel variants and which arithmetic a forward / backward pass runs on: selected per MODULE and per CALL,
never by assigning module globals.

The reference's modules are called concurrently -- ``nn.DataParallel`` runs ``forward`` from one Python thread per GPU
(train.py:305) -- so the choice between the default fp32 kernels, their A/B partners and the opt-in split-precision
kernels cannot live in process-global mutable state (SURVEY.md 8(b) "no global mutable state").  Three layers, the most
specific one wins:

1. process defaults: read ONCE from the ``OCC4D_*`` environment variables at import (``defaults()``);
2. a thread-local scope, ``with kernels.use(logit_precision='bf16x6'): ...`` (``occlusions4d_amd.kernels(...)`` is the
   same function): everything the CURRENT THREAD runs inside the block.  Backward passes run on autograd's own threads:
   every autograd Function of this package records the selection of its forward call and restores it around its
   backward (``carries_selection``), so a training step is consistent without any global either;
3. a module attribute: ``LocalPclResnetFC(...).kernel_selection = dict(precision='f16x3')`` (or the ``precision``
   property of PointTransformerLayer / PointTransformerBlock / LocalPclResnetFC) pins ONE module, whatever scope calls it.

``Selection.flags()`` maps a selection to the ``OCC4D_PATH_*`` bits of include/occ4d.h.
"""
import contextlib
import dataclasses
import functools
import os
import sys
import threading
import types

PRECISIONS = ('f32', 'bf16x6', 'f16x3')
# 'f32'     every GEMM on v_mfma_f32_16x16x4_f32 (the measured `value` of bench.py)
# 'bf16x6'  bf16 x 3 pieces per operand, 6 partial products, fp32 accumulate: fp32-class, no range restriction
# 'f16x3'   fp16 x 2 pieces per operand, 3 partial products, fp32 accumulate (round 6): half the matrix instructions of
#           bf16x6; inference forwards only; |weight| < 255, |activation| < 65504 (csrc/bf16x6.hpp)


def _env_flag(name, default):
    return os.environ.get(name, default) != '0'


@dataclasses.dataclass(frozen=True)
class Selection:
    fused_attention: bool = True      # False: the unfused kernel chain (OCC4D_PATH_UNFUSED)
    attn16: bool = True               # d = 416: csrc/crossattn16p.hip; False: csrc/crossattn.hip (OCC4D_PATH_FIRST_GEN)
    trunk_kernels: bool = True        # row-resident trunk kernels; False: the generic Linear kernel
    trunk4: bool = False              # half-CU trunk kernels (csrc/trunk4.hip): measured slower at the decode chunk
    fused_interp: bool = False        # A/B only: lin_z table term of block i + 1 in block i's epilogue
    logit_precision: str = 'f32'      # the d = 416 attention layers' GEMMs
    trunk_precision: str = 'f32'      # the decoder's 416-input Linear layers (residual blocks, query projection, layer3)
    train_precision: str = 'f32'      # training path: forward Linears, data gradients, pair-tensor recompute ('f32' | 'bf16x6')
    checkpoint_attention: bool = True     # training: cross-attention recomputes its pair tensors in backward
    stored_attention_form: str = 'merged'  # with checkpoint_attention off: 'merged' | 'as_written'
    checkpoint_chunk: int = 32768     # queries per recompute chunk in backward
    store_pairs: str = 'all'          # training, checkpointed attention: what the fused forward kernel leaves in HBM for backward --
                                      # 'none' (backward recomputes a, logits, pe), 'logits' (backward skips GEMM2; + 1.6 GB per
                                      # layer at config 5), 'all' (nothing recomputed; + 6.4 GB per layer)
    decode_streams: int = 2           # inference: HIP streams the decode mini-batches alternate between (1 = the reference's serial loop)

    def __post_init__(self):
        assert self.logit_precision in PRECISIONS, self.logit_precision
        assert self.trunk_precision in PRECISIONS, self.trunk_precision
        assert self.train_precision in ('f32', 'bf16x6'), self.train_precision
        assert self.stored_attention_form in ('merged', 'as_written'), self.stored_attention_form
        assert self.store_pairs in ('none', 'logits', 'all'), self.store_pairs
        split = {p for p in (self.logit_precision, self.trunk_precision) if p != 'f32'}
        assert len(split) <= 1, 'one split scheme per module: logit %s, trunk %s' % (self.logit_precision, self.trunk_precision)

    def flags(self):
        """OCC4D_PATH_* bits for the library's path-level entry points (inference forwards)."""
        from . import _lib as L
        f = L.PATH_DEFAULT
        if not self.fused_attention:
            f |= L.PATH_UNFUSED
        if not self.attn16:
            f |= L.PATH_FIRST_GEN
        if self.logit_precision != 'f32':
            f |= L.PATH_BF16X6
        if self.trunk_precision != 'f32':
            f |= L.PATH_BF16X6_TRUNK
        if 'f16x3' in (self.logit_precision, self.trunk_precision):
            f |= L.PATH_SPLIT_F16
        if not self.trunk_kernels:
            f |= L.PATH_GENERIC_LINEAR
        if self.trunk4:
            f |= L.PATH_TRUNK4
        if self.fused_interp:
            f |= L.PATH_FUSED_INTERP
        return f

    def replace(self, **kw):
        return dataclasses.replace(self, **_expand(kw))


def _expand(kw):
    """`precision=p` is shorthand for logit_precision = trunk_precision = p."""
    kw = dict(kw)
    if 'precision' in kw:
        p = kw.pop('precision')
        kw.setdefault('logit_precision', p)
        kw.setdefault('trunk_precision', p)
    unknown = set(kw) - {f.name for f in dataclasses.fields(Selection)}
    if unknown:
        raise TypeError('unknown kernel selection field(s): %s' % ', '.join(sorted(unknown)))
    return kw


def _from_environment():
    return Selection(
        trunk4=_env_flag('OCC4D_TRUNK4', '0'),
        fused_interp=_env_flag('OCC4D_FUSED_INTERP', '0'),
        logit_precision=os.environ.get('OCC4D_LOGIT_PRECISION', 'f32'),
        trunk_precision=os.environ.get('OCC4D_TRUNK_PRECISION', 'f32'),
        train_precision=os.environ.get('OCC4D_TRAIN_PRECISION', 'f32'),
        stored_attention_form=os.environ.get('OCC4D_STORED_ATTENTION_FORM', 'merged'),
        checkpoint_chunk=int(os.environ.get('OCC4D_CHECKPOINT_CHUNK', '32768')),
        store_pairs=os.environ.get('OCC4D_STORE_PAIRS', 'all'),
        decode_streams=int(os.environ.get('OCC4D_DECODE_STREAMS', '2')))


_DEFAULTS = _from_environment()         # immutable; the environment is read once
_tls = threading.local()


def defaults():
    return _DEFAULTS


def scope():
    """The selection of the current thread's innermost `use` block, or the process defaults."""
    stack = getattr(_tls, 'stack', None)
    return stack[-1] if stack else _DEFAULTS


def current(module=None):
    """The selection a call into `module` runs on: the thread's scope with the module's own `kernel_selection` (a dict of
    Selection fields, `precision` shorthand allowed) applied on top."""
    sel = scope()
    own = getattr(module, 'kernel_selection', None) if module is not None else None
    return sel.replace(**own) if own else sel


@contextlib.contextmanager
def use(selection=None, **fields):
    """Thread-local scope: `with use(precision='bf16x6'):` or `with use(some_selection):`."""
    base = scope() if selection is None else selection
    sel = base.replace(**fields) if fields else base
    stack = getattr(_tls, 'stack', None)
    if stack is None:
        stack = _tls.stack = []
    stack.append(sel)
    try:
        yield sel
    finally:
        stack.pop()


def carries_selection(fn_class):
    """Class decorator for torch.autograd.Function subclasses: the selection in force when `forward` ran is restored
    around `backward`, which autograd calls on its own worker thread (where no `use` block of the caller is visible)."""
    fwd, bwd = fn_class.forward, fn_class.backward

    @staticmethod
    @functools.wraps(fwd)
    def forward(ctx, *args, **kw):
        ctx._occ4d_selection = scope()
        return fwd(ctx, *args, **kw)

    @staticmethod
    @functools.wraps(bwd)
    def backward(ctx, *grads):
        with use(ctx._occ4d_selection):
            return bwd(ctx, *grads)
    fn_class.forward, fn_class.backward = forward, backward
    return fn_class


class HasKernelSelection:
    """Mixin for the modules that own a kernel choice: `module.precision = 'bf16x6'` / `module.kernel_selection =
    dict(...)`; None / {} = inherit the caller's scope."""
    kernel_selection = None

    @property
    def precision(self):
        own = self.kernel_selection or {}
        return own.get('precision', own.get('logit_precision'))

    @precision.setter
    def precision(self, value):
        own = dict(self.kernel_selection or {})
        for k in ('precision', 'logit_precision', 'trunk_precision'):
            own.pop(k, None)
        if value is not None:
            assert value in PRECISIONS, value
            own['precision'] = value
        self.kernel_selection = own or None


class _CallableModule(types.ModuleType):
    """`with occlusions4d_amd.kernels(precision='bf16x6'):` == `with occlusions4d_amd.kernels.use(...)`."""

    def __call__(self, *args, **kw):
        return use(*args, **kw)


sys.modules[__name__].__class__ = _CallableModule
