"""CPU: host-side logic of the product package that needs no GPU (grid generation, post-op
codes, configuration tables, state-dict naming, derived/merged weights)."""
import numpy as np
import pytest
import torch

import golden_cases as gc
import occlusions4d_amd as pk
from conftest import load_golden
from oracle import path as op


def test_grid_matches_reference_golden():
    g = load_golden('g9_grid')
    for case in gc.GRID_CASES:
        pts = pk.geometry.sample_implicit_points_blind_numpy(case['num_sample'], case['min_z'], case['cube_bounds'],
                                                             case['time_idx'], case['kind'], 4, 'grid')
        n = case['name']
        assert pts.dtype == np.float32 and pts.shape == (int(g[n + '_n'][0]), 4)
        assert np.array_equal(pts[:130], g[n + '_head']) and np.array_equal(pts[-130:], g[n + '_tail'])
        assert np.array_equal(pts.astype(np.float64).sum(axis=0), g[n + '_sum'])


def test_grid_random_mode_and_errors():
    np.random.seed(3)
    a = pk.geometry.sample_implicit_points_blind_numpy(100, -1.0, 5.0, 2, 'greater', 4, 'random')
    np.random.seed(3)
    b = op.sample_query_points(100, -1.0, 5.0, 2, 'greater', 4, 'random')
    assert np.array_equal(a, b) and a.shape == (100, 4) and (a[:, 3] == 2).all()
    with pytest.raises(ValueError):
        pk.geometry.sample_implicit_points_blind_numpy(10, -1.0, 5.0, 0, 'kitti', 4, 'grid')
    with pytest.raises(ValueError):
        pk.geometry.sample_implicit_points_blind_numpy(10, -1.0, 5.0, 0, 'greater', 4, 'sobol')


@pytest.mark.parametrize('color_mode,g,seg,track', [('rgb_nosigmoid', 5, False, 'none'), ('rgb', 18, True, 'none'),
                                                    ('rgb', 5, False, 'one'), ('hsv', 16, False, 'none'),
                                                    ('bins', 11, False, 'one')])
def test_squash_codes_equal_reference_sequence(color_mode, g, seg, track):
    codes = pk.inference.squash_codes(g, color_mode, seg, track, 13)
    x = torch.randn(64, g) * 3
    ref = op.squash_outputs(x.clone(), color_mode, seg, track, 13)
    got = x.clone()
    for c, code in enumerate(codes):
        if code == 1:
            got[:, c] = torch.sigmoid(got[:, c])
        elif code == 2:
            got[:, c] = got[:, c].clamp(0, 1)
    assert torch.equal(got, ref)


def test_state_dict_names_follow_reference_layout():
    for kind in ('greater', 'carla'):
        pa, ia, _ = pk.configs.model_args(kind, 2048)
        enc = pk.model.PointCompletionNetV3(**pa)
        dec = pk.implicit.LocalPclResnetFC(**ia)
        assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == pk.configs.encoder_param_shapes(pa)
        assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == pk.configs.decoder_param_shapes(ia)
    assert sum(p.numel() for p in enc.parameters()) > 1_100_000
    assert sum(v.numel() for v in pk.model.PointCompletionNetV3(**pk.configs.model_args('greater')[0]).parameters()) == 1122568
    assert sum(v.numel() for v in pk.implicit.LocalPclResnetFC(**pk.configs.model_args('greater')[1]).parameters()) == 6087173


def test_cross_attention_placement_and_errors():
    _, ia, _ = pk.configs.model_args('greater')
    dec = pk.implicit.LocalPclResnetFC(**ia)
    assert dec.use_pt_inds == {2: 0, 4: 1}
    with pytest.raises(NotImplementedError):
        pk.implicit.LocalPclResnetFC(**dict(ia, cr_attn_type='cs'))
    with pytest.raises(ValueError):
        pk.implicit.LocalPclResnetFC(**dict(ia, cr_attn_type='cx'))
    with pytest.raises(ValueError):
        pk.implicit.ResnetBlockFC(activation='gelu')
    with pytest.raises(ValueError):
        pk.modules.DownTransition(8, 16, norm_type='group')


def test_inference_forward_has_no_cpu_fallback():
    """A module left on the CPU cannot run: building the library's weight view rejects CPU parameters."""
    blk = pk.modules.PointTransformerBlock(32, 32, 32, num_neighbors=4, d_hidden_abstract=16)
    with pytest.raises(RuntimeError, match='no CPU fallback'), torch.no_grad():
        blk(torch.zeros(1, 8, 32), torch.zeros(1, 8, 3), torch.zeros(1, 9, 16), torch.zeros(1, 9, 3))


def test_training_forward_has_no_cpu_fallback_either():
    blk = pk.implicit.ResnetBlockFC(8, 8, 8)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        blk(torch.zeros(2, 8, requires_grad=True))


def test_synthetic_inputs_are_deterministic_and_tie_free():
    a = pk.configs.synthetic_pcl('carla', 2048, 12, 5)
    b = pk.configs.synthetic_pcl('carla', 2048, 12, 5)
    assert torch.equal(a, b) and a.shape == (1, 2048, 8)
    assert np.unique(a[0, :, :3].numpy(), axis=0).shape[0] == 2048
    w1 = pk.configs.fill_state_dict({'l.weight': (4, 8), 'l.bias': (4,)}, 3)
    w2 = pk.configs.fill_state_dict({'l.weight': (4, 8), 'l.bias': (4,)}, 3)
    assert all(torch.equal(w1[k], w2[k]) for k in w1) and w1['l.weight'].abs().max() <= 1 / np.sqrt(8)


def test_tracking_loss_reads_the_channel_of_the_colour_mode():
    """training.implicit_loss reads the tracking logit where utils.get_track_idx puts it (utils/utils.py:204-224): channel 4
    for the three-channel colour modes, 15 for 'hsv', 10 for 'bins' (never silently a colour logit)."""
    import occlusions4d_amd as pk
    tgt = torch.zeros((1, 8, 6))
    tgt[..., 0] = 1.0                  # (every point solid: the tracking term has supervised points)
    tgt[..., 4] = 1.0
    for mode, width, idx in (('rgb', 5, 4), ('hsv', 16, 15), ('bins', 11, 10)):
        out = torch.zeros((1, 8, width), requires_grad=True)
        pk.training.implicit_loss(out, tgt, density_lw=0.0, color_lw=0.0, tracking_lw=0.5, color_mode=mode).backward()
        touched = out.grad.abs().sum(dim=(0, 1)).nonzero()[:, 0].tolist()
        assert touched == [idx], (mode, touched)
    with pytest.raises(ValueError):
        pk.training.implicit_loss(torch.zeros((1, 8, 5)), tgt, density_lw=1.0, tracking_lw=0.5, color_mode='nonsense')


def test_evaluate_clip_shares_the_encode_only_when_it_is_deterministic(monkeypatch):
    """evaluation.evaluate_clip: one encode per clip for a deterministic encoder; an encoder built with
    fps_random_start=True is re-encoded for every output frame, as eval/test.py:67-86 does."""
    import types
    import numpy as np
    import occlusions4d_amd as pk
    seen = []

    def fake(pcl_input, sem, target, networks, device, mode, *a, encoded=None, return_encoded=False, **kw):
        seen.append(encoded)
        res = dict(pcl_abstract=np.zeros((2, 4), np.float32), output_solid=np.zeros((1, 9), np.float32),
                   output_air=np.zeros((1, 5), np.float32), points_query=np.zeros((2, 4), np.float32))
        if return_encoded:
            res['_encoded'] = ('abstract', 'global')
        return res
    monkeypatch.setattr(pk.inference, 'perform_inference', fake)
    args = types.SimpleNamespace(track_mode='none', min_z=-1.0, cr_cube_bounds=5.0, color_mode='rgb', sample_implicit=True,
                                 num_sample=8, point_sample_mode='grid', implicit_batch_size=8, segmentation_lw=0.0,
                                 point_occupancy_radius=0.2, semantic_classes=13, density_threshold=0.5, cube_mode=4)
    batch = dict(pcl_input=torch.zeros((1, 4, 8)), pcl_input_sem=torch.zeros((1, 4, 1)),
                 pcl_target=[torch.zeros((1, 3, 9)) for _ in range(3)], meta_data=dict(pcl_target_size=[3, 3, 3]))

    class Enc(torch.nn.Module):
        def __init__(self, random_start):
            super().__init__()
            self.down = torch.nn.Module()
            self.down.fps_random_start = random_start
    for random_start, want in ((False, [None, ('abstract', 'global'), ('abstract', 'global')]), (True, [None, None, None])):
        del seen[:]
        out = pk.evaluation.evaluate_clip(batch, [Enc(random_start), None], 'cpu', args, 'greater')
        assert len(out) == 3 and seen == want
