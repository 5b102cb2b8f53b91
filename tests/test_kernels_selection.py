"""Kernel / precision selection is per thread and per module (occlusions4d_amd.kernels), never a module global:
the reference's nn.DataParallel calls forward from one Python thread per GPU (train.py:305; SURVEY.md 8(b) "no global
mutable state").  Host logic only -- runs without a GPU."""
import threading

import pytest
import torch

import occlusions4d_amd as pk

K = pk.kernels
L = pk._lib


def test_flags_map_to_the_header_bits():
    d = K.defaults()
    assert d.flags() == L.PATH_DEFAULT or d != K.Selection()          # (an OCC4D_* environment may seed the defaults)
    s = K.Selection()
    assert s.flags() == 0
    assert s.replace(fused_attention=False).flags() == L.PATH_UNFUSED
    assert s.replace(attn16=False).flags() == L.PATH_FIRST_GEN
    assert s.replace(trunk_kernels=False).flags() == L.PATH_GENERIC_LINEAR
    assert s.replace(trunk4=True).flags() == L.PATH_TRUNK4
    assert s.replace(fused_interp=True).flags() == L.PATH_FUSED_INTERP
    assert s.replace(logit_precision='bf16x6').flags() == L.PATH_BF16X6
    assert s.replace(precision='bf16x6').flags() == L.PATH_BF16X6 | L.PATH_BF16X6_TRUNK
    assert s.replace(precision='f16x3').flags() == L.PATH_BF16X6 | L.PATH_BF16X6_TRUNK | L.PATH_SPLIT_F16
    assert s.replace(trunk_precision='f16x3').flags() == L.PATH_BF16X6_TRUNK | L.PATH_SPLIT_F16
    with pytest.raises(AssertionError):
        s.replace(logit_precision='bf16x6', trunk_precision='f16x3')   # one split scheme per module
    with pytest.raises(AssertionError):
        s.replace(precision='fp8')
    with pytest.raises(TypeError):
        s.replace(no_such_switch=True)
    with pytest.raises(Exception):
        s.trunk4 = True                                                # frozen


def test_store_pairs_modes_are_validated():
    """Selection.store_pairs (training: what the fused forward attention kernel keeps for backward) takes three values."""
    from occlusions4d_amd import kernels
    assert kernels.defaults().store_pairs in ('none', 'logits', 'all')
    for mode in ('none', 'logits', 'all'):
        with kernels.use(store_pairs=mode):
            assert kernels.scope().store_pairs == mode
    with pytest.raises(AssertionError):
        kernels.defaults().replace(store_pairs='everything')


def test_scopes_nest_and_unwind():
    base = K.scope()
    with pk.kernels(precision='bf16x6') as a:
        assert K.scope() is a and a.logit_precision == a.trunk_precision == 'bf16x6'
        with K.use(trunk_precision='f32', decode_streams=1) as b:
            assert K.scope() is b and b.logit_precision == 'bf16x6' and b.trunk_precision == 'f32' and b.decode_streams == 1
        assert K.scope() is a
        with pytest.raises(RuntimeError):
            with K.use(fused_attention=False):
                raise RuntimeError('unwinds on exceptions')
        assert K.scope() is a
    assert K.scope() is base
    assert pk.point_transformer_layer.path_flags() == base.flags()


def test_a_scope_is_invisible_to_other_threads():
    seen, go, done = {}, threading.Barrier(3), threading.Barrier(3)

    def worker(name, precision):
        with pk.kernels(logit_precision=precision):
            go.wait()                       # all three threads are inside their own scope now
            seen[name] = (K.scope().logit_precision, pk.point_transformer_layer.path_flags())
            done.wait()
    ts = [threading.Thread(target=worker, args=('a', 'bf16x6')), threading.Thread(target=worker, args=('b', 'f16x3'))]
    for t in ts:
        t.start()
    go.wait()
    seen['main'] = (K.scope().logit_precision, pk.point_transformer_layer.path_flags())
    done.wait()
    for t in ts:
        t.join()
    assert seen['a'] == ('bf16x6', L.PATH_BF16X6)
    assert seen['b'] == ('f16x3', L.PATH_BF16X6 | L.PATH_SPLIT_F16)
    assert seen['main'] == (K.defaults().logit_precision, K.defaults().flags())


def test_a_module_attribute_pins_one_module():
    pa, ia, _ = pk.configs.model_args('greater', 768)
    dec = pk.implicit.LocalPclResnetFC(**ia)
    other = pk.implicit.LocalPclResnetFC(**ia)
    flags = pk.point_transformer_layer.path_flags
    assert dec.precision is None and flags(dec) == K.defaults().flags()
    dec.precision = 'f16x3'
    assert dec.kernel_selection == dict(precision='f16x3')
    assert flags(dec) & L.PATH_SPLIT_F16 and not flags(other) & L.PATH_SPLIT_F16
    with pk.kernels(precision='bf16x6', trunk4=False):
        assert flags(dec) & L.PATH_SPLIT_F16                     # the module's own choice wins over the caller's scope
        assert flags(other) == K.defaults().replace(precision='bf16x6').flags()
    dec.precision = None
    assert dec.kernel_selection is None and flags(dec) == K.defaults().flags()
    layer = dec.pt_blocks[0].layer2
    layer.kernel_selection = dict(fused_attention=False)
    assert flags(layer) & L.PATH_UNFUSED and not flags(dec) & L.PATH_UNFUSED
    assert 'kernel_selection' not in dec.state_dict() and not any('kernel' in k for k in dec.state_dict())


def test_backward_runs_under_the_selection_of_its_forward():
    seen = []

    @K.carries_selection
    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            seen.append(('fwd', K.scope().train_precision, K.scope().checkpoint_chunk))
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(('bwd', K.scope().train_precision, K.scope().checkpoint_chunk))
            return g * 2
    x = torch.ones(3, requires_grad=True)
    with pk.kernels(train_precision='bf16x6', checkpoint_chunk=128):
        y = Probe.apply(x)
    out = {}

    def run_backward():                       # a different thread, no scope of its own: autograd's situation on a GPU
        y.sum().backward()
        out['scope_after'] = K.scope()
    t = threading.Thread(target=run_backward)
    t.start()
    t.join()
    assert seen == [('fwd', 'bf16x6', 128), ('bwd', 'bf16x6', 128)]
    assert out['scope_after'] is K.defaults()
    assert torch.equal(x.grad, torch.full((3,), 2.0))


def test_no_module_level_switches_are_left():
    ptl = pk.point_transformer_layer
    for name in ('USE_FUSED_ATTENTION', 'USE_ATTN16', 'USE_TRUNK_KERNELS', 'USE_TRUNK4', 'LOGIT_PRECISION', 'TRUNK_PRECISION',
                 'FUSED_INTERP', 'CHECKPOINT_ATTENTION', 'STORED_ATTENTION_FORM', '_CHECKPOINT_CHUNK'):
        assert not hasattr(ptl, name), name
    assert not hasattr(pk.autograd, 'TRAIN_PRECISION') and not hasattr(pk.inference, 'DECODE_STREAMS')
