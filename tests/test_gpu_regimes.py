"""GPU parity outside the init-scale / tie-free regime (VERDICT r3 "What's weak" 1): the HIP path against fixtures the
REAL reference produced (oracle/gen_golden.py:regimes) for

* clouds zero-padded as the reference's data path pads them (utils/geometry.py:315-325): hundreds of coincident points
  in the encoder's self-kNN-16, the FPS chain and the max-pool (G5p, G10p);
* cross- / self-attention layers with weights x4 / x8 and features x4 -- logits far beyond the softmax's exp range, one
  neighbour dominating -- with all-equal logits and with a single dominant neighbour (G2r), through every attention
  kernel generation, the unfused chain and the opt-in split-bf16 logit mode;
* the decoder with scaled cross-attention weights and with queries 3x outside the cuboid (G8r), default kernels and
  every kernel variant.

Bound for the scaled cases: max(1e-4, 2 max|ref32 - ref64|) against the reference's fp64 run (golden_cases.regime_bound):
at these magnitudes (outputs up to 2e2) no fp32 op order can promise 1e-4 absolute.  Measured errors are printed (-s) and
recorded in DESIGN.md section 2."""
import contextlib

import numpy as np
import pytest
import torch

import golden_cases as gc
from conftest import load_golden

pytestmark = pytest.mark.gpu
T = gc.as_tensor


@pytest.fixture(scope='module')
def pk():
    import occlusions4d_amd
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    occlusions4d_amd._lib.lib()
    return occlusions4d_amd


def dev(a):
    return T(a).cuda()


def err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == np.asarray(b).shape, (a.shape, np.asarray(b).shape)
    return float(np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max()) if a.size else 0.0


ATTN_PATHS = ['attn16p', 'first', 'chain', 'bf16x6', 'f16x3']
SPLIT = ('bf16x6', 'f16x3')


@contextlib.contextmanager
def attention_path(pk, which):
    """Selects the kernel generation the inference layer takes: 'attn16p' (default, csrc/crossattn16p.hip), 'first'
    (crossattn.hip), 'chain' (unfused kernels), 'bf16x6' (round 5: every attention GEMM on three-way split bf16 MFMAs, six
    partial products) / 'f16x3' (round 6: two fp16 pieces, three partial products), both csrc/crossattn_bf16x6.hip and
    both held to the fp32 paths' own bound, no relaxation: that is their claim.  A thread-local scope (kernels.py)."""
    with pk.kernels(attn16=which in ('attn16p',) + SPLIT, fused_attention=which != 'chain',
                    logit_precision=which if which in SPLIT else 'f32'):
        yield


# ------------------------------------------------------------------ G2r: one attention layer
@pytest.mark.parametrize('path', ATTN_PATHS)
@pytest.mark.parametrize('case', gc.PTL_REGIME_CASES, ids=lambda c: c['name'])
def test_pt_layer_regimes(pk, case, path):
    if case['dim'] not in pk.ops.FUSED_ATTN_DIMS and path != 'chain':
        pytest.skip('encoder widths run the unfused chain only')
    if case['dim'] != 416 and path in ('attn16p',) + SPLIT:
        pytest.skip('crossattn16p.hip / crossattn_bf16x6.hip are built for d = 416')
    x, pos, x2, pos2, sd = gc.ptl_inputs(case)
    layer = pk.point_transformer_layer.PointTransformerLayer(case['dim'], num_neighbors=case['k'],
                                                             dim2=case.get('dim2')).cuda()
    layer.load_state_dict(sd)
    args = (dev(x)[None], dev(pos)[None]) + ((dev(x2)[None], dev(pos2)[None]) if x2 is not None else ())
    with torch.no_grad(), attention_path(pk, path):
        agg = layer(*args)[0]
    g = load_golden('g2r_ptl_' + case['name'])
    assert torch.isfinite(agg).all()
    bound = gc.regime_bound(g, 'agg')
    e64, e32 = err(agg, g['agg64']), err(agg, g['agg'])
    print('\n[g2r %s / %s] max|x| %.3g  |hip - ref64| %.3g  |hip - ref32| %.3g  |ref32 - ref64| %.3g  bound %.3g'
          % (case['name'], path, float(np.abs(g['agg64']).max()), e64, e32, err(g['agg'], g['agg64']), bound))
    assert e64 <= bound, '%s/%s: %.3g > %.3g' % (case['name'], path, e64, bound)


# ------------------------------------------------------------------ G5p: encoder on zero-padded clouds
@pytest.mark.parametrize('case', gc.ENC_PAD_CASES, ids=lambda c: c['name'])
def test_encoder_on_zero_padded_clouds(pk, case):
    pcl, pa, sd = gc.enc_inputs(case)
    net = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out, xg, _ = net(pcl.cuda(), False)
    g = load_golden('g5_enc_' + case['name'])
    assert np.array_equal(out[0, :, :3].cpu().numpy(), g['pcl_out'][:, :3])     # FPS subset, bit exact
    e = err(out[0], g['pcl_out'])
    print('\n[g5p %s] |hip - ref| %.3g (abstract), %.3g (global)' % (case['name'], e, err(xg[0], g['x_global'])))
    assert e <= 1e-4 and err(xg[0], g['x_global']) <= 1e-4


@pytest.mark.parametrize('case', gc.ENC_PAD_CASES, ids=lambda c: c['name'])
def test_geometry_of_zero_padded_clouds_is_the_restated_torch_cluster(pk, case):
    """Every level's FPS subset and max-pool neighbour SETS on the padded cloud against oracle/cluster.py (first
    argmax; exact kNN).  Neighbour lists are compared as multisets of COORDINATES: which of several coincident points is
    listed is implementation-defined in the reference as well."""
    from oracle import cluster
    pcl, pa, sd = gc.enc_inputs(case)
    p = pcl[0, :, :3].contiguous()
    for level in range(pa['down_blocks']):
        n_new = int(np.ceil(p.shape[0] / 3))
        inds = pk.ops.fps_auto(p.cuda(), n_new).cpu().long()
        ref = torch.sort(cluster.fps(p, None, ratio=1.0 / 3, random_start=False))[0]
        assert torch.equal(p[inds], p[ref]), 'level %d' % level
        nn = pk.ops.knn(p[inds].cuda(), p.cuda(), pa['down_neighbors'], metric=0).cpu().long()
        nn_ref = cluster.knn(p, p[ref], pa['down_neighbors'])[1].view(n_new, -1)
        a = torch.sort(((p[inds][:, None] - p[nn]) ** 2).sum(-1), dim=1)[0]
        b = torch.sort(((p[ref][:, None] - p[nn_ref]) ** 2).sum(-1), dim=1)[0]
        assert torch.equal(a, b), 'level %d' % level
        p = p[inds]


# ------------------------------------------------------------------ G8r: decoder
DEC_VARIANTS = ['default', 'trunk4', 'generic_trunk', 'first', 'chain', 'bf16x6', 'bf16x6_trunk', 'bf16x6_all',
                'f16x3', 'f16x3_trunk', 'f16x3_all']


@contextlib.contextmanager
def decoder_variant(pk, variant):
    # the trunk's Linear layers (<scheme>_trunk) / the whole decoder (<scheme>_all) on the split-precision kernels, held to
    # the fp32 paths' own bound
    scheme = variant.split('_')[0] if variant.split('_')[0] in SPLIT else None
    trunk = scheme if scheme and variant.endswith(('_trunk', '_all')) else 'f32'
    path = variant if variant in ('first', 'chain') + SPLIT else (scheme if scheme and variant.endswith('_all') else 'attn16p')
    with attention_path(pk, path), pk.kernels(trunk4=variant == 'trunk4', trunk_kernels=variant != 'generic_trunk',
                                              trunk_precision=trunk):
        yield


@pytest.mark.parametrize('variant', DEC_VARIANTS)
@pytest.mark.parametrize('case', gc.DEC_REGIME_CASES, ids=lambda c: c['name'])
def test_decoder_regimes(pk, case, variant):
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    with torch.no_grad(), decoder_variant(pk, variant):
        out, pen = net(dev(q), dev(abstract), dev(fglob), None)
    g = load_golden('g8r_dec_' + case['name'])
    assert torch.isfinite(out).all() and torch.isfinite(pen).all()
    bo, bp = gc.regime_bound(g, 'output'), gc.regime_bound(g, 'penult')
    eo, ep = err(out, g['output64']), err(pen[:, ::8], g['penult64'])
    print('\n[g8r %s / %s] output: max|x| %.3g |hip - ref64| %.3g (ref32: %.3g, bound %.3g)   penult: max|x| %.3g '
          '|hip - ref64| %.3g (ref32: %.3g, bound %.3g)'
          % (case['name'], variant, float(np.abs(g['output64']).max()), eo, err(g['output'], g['output64']), bo,
             float(np.abs(g['penult64']).max()), ep, err(g['penult'], g['penult64']), bp))
    assert eo <= bo and ep <= bp


# ------------------------------------------------------------------ G10p: perform_inference on zero-padded clouds
@pytest.mark.parametrize('case', gc.INFER_PAD_CASES, ids=lambda c: c['name'])
def test_perform_inference_on_zero_padded_clouds(pk, case):
    from oracle import path as op
    pcl, pa, ia, inf, esd, dsd = gc.infer_inputs(case)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    dec.load_state_dict(dsd)
    res = pk.inference.perform_inference(
        pcl.clone(), None, None, [enc, dec], torch.device('cuda:0'), 'if', inf['min_z'], inf['cube_bounds'],
        inf['color_mode'], case['time_idx'], None, sample_implicit=True, num_sample=case['num_sample'],
        point_sample_mode='grid', batch_size=case['batch_size'],
        predict_segmentation=inf['predict_segmentation'], track_mode='none', semantic_classes=13,
        density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4, compress_air=True)
    g = load_golden('g10_infer_' + case['name'])
    assert err(res['pcl_abstract'], g['pcl_abstract']) <= 1e-4
    assert err(res['features_global'], g['features_global']) <= 1e-4
    amb = op.tie_ambiguous(T(res['points_query']), T(g['pcl_abstract']), ia['num_local_features'],
                           ia['cross_attn_neighbors']).numpy()
    if inf['data_kind'] == 'greater':
        assert not amb.any()
    assert (~amb).sum() > 0.5 * amb.size
    e = err(res['implicit_output'][~amb], g['implicit_output'][~amb])
    print('\n[g10p %s] |hip - ref| %.3g on %d of %d queries' % (case['name'], e, int((~amb).sum()), amb.size))
    assert e <= 1e-4
