"""GPU, BASELINE.json full sizes (configs[1] GREATER and configs[2] CARLA, n_points = 14336,
~0.5 M grid queries): direct parity against the CPU oracle where the oracle finishes in seconds
(the encode; a query sample of the decode) and size-independent properties on the full grid
(neighbour lists sorted and self-first, FPS unique/ascending/maximin, batch-split invariance,
determinism, post-op ranges)."""
import numpy as np
import pytest
import torch

import occlusions4d_amd as pk
from oracle import path as op

pytestmark = pytest.mark.gpu
TOL = 1e-4
N_POINTS, VIDEO_LEN, NUM_SAMPLE, BATCH, SEED = 14336, 12, 524288, 32768, 1830


@pytest.fixture(scope='module', params=['greater', 'carla'])
def scene(request):
    kind = request.param
    pa, ia, inf = pk.configs.model_args(kind, N_POINTS)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
    pcl = pk.configs.synthetic_pcl(kind, N_POINTS, VIDEO_LEN, SEED)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    q = pk.geometry.sample_implicit_points_blind_numpy(NUM_SAMPLE, inf['min_z'], inf['cube_bounds'], 3, kind, 4, 'grid')
    with torch.no_grad():
        ab, fg, _ = enc(pcl.cuda(), False)
    return dict(kind=kind, pa=pa, ia=ia, inf=inf, esd=esd, dsd=dsd, pcl=pcl, enc=enc, dec=dec, q=q,
                ab=ab[0], fg=fg[0])


def test_grid_size(scene):
    assert scene['q'].shape[0] == (534528 if scene['kind'] == 'greater' else 541314)


def test_encode_matches_oracle_at_full_size(scene):
    ref_ab, ref_fg = op.encoder_forward(scene['esd'], scene['pa'], scene['pcl'])
    ab, fg = scene['ab'].cpu(), scene['fg'].cpu()
    assert ab.shape == ref_ab[0].shape == ((531, 291) if scene['kind'] == 'greater' else (2124, 291))
    assert torch.equal(ab[:, :3], ref_ab[0][:, :3])          # three FPS levels: identical subsets, bit exact
    assert (ab - ref_ab[0]).abs().max() <= TOL
    assert (fg - ref_fg[0]).abs().max() <= TOL


def test_decode_sample_matches_oracle_at_full_abstract_size(scene):
    rng = np.random.default_rng(3)
    sel = np.sort(rng.choice(scene['q'].shape[0], size=1536, replace=False))
    qs = torch.from_numpy(scene['q'][sel])
    ab_cpu, fg_cpu = scene['ab'].cpu(), scene['fg'].cpu()
    with op.stable_ties():                                    # the product's documented tie rule
        ref, ref_pen = op.decoder_forward(scene['dsd'], scene['ia'], qs, ab_cpu, fg_cpu)
    with torch.no_grad():
        out, pen = scene['dec'](qs.cuda(), scene['ab'], scene['fg'], None)
    assert (out.cpu() - ref).abs().max() <= TOL
    assert (pen.cpu() - ref_pen).abs().max() <= TOL
    # and the reference itself (unstable ties) wherever it is well defined
    amb = op.tie_ambiguous(qs, ab_cpu, scene['ia']['num_local_features'], scene['ia']['cross_attn_neighbors'])
    ref2, _ = op.decoder_forward(scene['dsd'], scene['ia'], qs, ab_cpu, fg_cpu)
    assert (out.cpu()[~amb] - ref2[~amb]).abs().max() <= TOL
    if scene['kind'] == 'greater':
        assert not amb.any()


def test_full_grid_decode_properties(scene):
    q = torch.from_numpy(scene['q']).cuda()
    inf, dec = scene['inf'], scene['dec']
    with torch.no_grad():
        res = pk.inference.infer_device(scene['pcl'].cuda(), q, scene['enc'], dec, BATCH, inf['color_mode'],
                                        inf['predict_segmentation'], 'none', 13)
        out = res['implicit_output']
        assert out.shape == (q.shape[0], scene['ia']['d_out']) and torch.isfinite(out).all()
        # determinism: a second full pass is bit-identical
        res2 = pk.inference.infer_device(scene['pcl'].cuda(), q, scene['enc'], dec, BATCH, inf['color_mode'],
                                         inf['predict_segmentation'], 'none', 13)
        assert torch.equal(out, res2['implicit_output'])
        # batch-split invariance: a different mini-batching gives the same rows up to fp32 rounding
        # (a query's place inside its 9-query workgroup decides whether its softmax is reduced in
        # one piece or merged from four partials, so the last bits may differ)
        lo = 200000
        part, _ = dec(q[lo:lo + 5000], scene['ab'], scene['fg'], None)
        pk.ops.squash(part, pk.inference.squash_codes(dec.d_out, inf['color_mode'], inf['predict_segmentation'],
                                                      'none', 13))
        assert (part - out[lo:lo + 5000]).abs().max() <= 1e-5
    # post-op ranges (eval/inference.py:218-243)
    assert (out[:, 0] >= 0).all() and (out[:, 0] <= 1).all()
    assert (out[:, 1:4] >= 0).all() and (out[:, 1:4] <= 1).all()
    if inf['predict_segmentation']:
        assert (out[:, -13:] >= 0).all() and (out[:, -13:] <= 1).all()


def test_knn_properties_at_full_size(scene):
    pos = scene['pcl'][0, :, :3].cuda()
    idx = pk.ops.knn(pos, pos, 16, metric=0)
    assert torch.equal(idx[:, 0].cpu(), torch.arange(N_POINTS, dtype=torch.int32))   # self first (d = 0)
    d = ((pos[:, None, :] - pos[idx.long()]) ** 2).sum(-1)
    assert (d[:, 1:] >= d[:, :-1]).all()                                               # sorted nearest first
    assert (torch.sort(idx, dim=1)[0][:, 1:] != torch.sort(idx, dim=1)[0][:, :-1]).all()   # no duplicates
    # every excluded point is at least as far as the k-th neighbour (sample of rows, exhaustive columns)
    rows = torch.arange(0, N_POINTS, 97, device='cuda')
    full = ((pos[rows, None, :] - pos[None, :, :]) ** 2).sum(-1)
    kth = d[rows, -1]
    assert ((full < kth[:, None]).sum(1) <= 16).all()
    # query -> abstract, Euclidean metric with distances
    q = torch.from_numpy(scene['q'][::53]).cuda()
    i8, d8 = pk.ops.knn(q, scene['ab'], 8, metric=1, return_dist=True)
    assert (d8[:, 1:] >= d8[:, :-1]).all()
    ref = torch.linalg.norm(q[:, None, :3] - scene['ab'][i8.long()][:, :, :3], dim=-1)
    assert torch.allclose(d8, ref, atol=1e-5)


def test_fps_properties_at_full_size(scene):
    pos = scene['pcl'][0, :, :3].cuda()
    m = -(-N_POINTS // 3)
    sel, order = pk.ops.fps(pos, m, return_order=True)
    s = sel.cpu().numpy()
    assert s.shape == (m,) and (np.diff(s) > 0).all() and s[0] == 0 and s[-1] < N_POINTS      # ascending, unique
    assert np.array_equal(np.sort(order.cpu().numpy()), s) and int(order[0]) == 0
    # farthest-point property: the running min distance at selection time never increases
    p = pos.cpu().double()
    o = order.cpu().long()
    step = [float(((p[o[:t]] - p[o[t]]) ** 2).sum(1).min()) for t in (1, 2, 5, 50, 500, 2000, m - 1)]
    assert all(a >= b for a, b in zip(step, step[1:]))
    # maximin: no unselected point is farther from the sample than the last selected one was
    mask = torch.ones(N_POINTS, dtype=torch.bool)
    mask[o] = False
    rest = p[mask][::7]
    dmin = torch.cdist(rest, p[o]).min(1)[0] ** 2
    assert float(dmin.max()) <= step[-1] * (1 + 1e-6) + 1e-9
