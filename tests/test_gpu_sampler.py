"""GPU parity for the training-time point sampler: with the reference's seeds the HIP-backed
sampler returns the reference sampler's supervision points bit for bit (golden G13, produced by the reference class
on CPU); at the training configuration's size the sampler's defining invariants are checked on the device."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from conftest import load_golden

pytestmark = pytest.mark.gpu
KEYS = ['solid_input', 'air_input', 'solid_target', 'air_target', 'solid_sbs', 'air_sbs']


@pytest.fixture(scope='module')
def pk():
    import occlusions4d_amd
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    occlusions4d_amd._lib.lib()
    return occlusions4d_amd


class _Log:
    def __init__(self):
        self.warnings = 0

    def warning(self, *a, **k):
        self.warnings += 1


@pytest.mark.parametrize('case', gc.SAMPLER_CASES, ids=lambda c: c['name'])
def test_sampler_replays_reference_draws(pk, case):
    g = load_golden('g13_sampler_' + case['name'])
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **gc.sampler_config(case))
    np.random.seed(case['seed'])
    torch.manual_seed(case['seed'])
    res = sampler([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z).cuda() for z in sizes],
                  torch.from_numpy(valo).cuda(), torch.from_numpy(num_valo).cuda(), case['time_idx'])
    for key, val in zip(KEYS, res):
        assert tuple(val.shape) == g[key].shape, key
        assert np.array_equal(val.cpu().numpy(), g[key]), key
    assert all(v.is_cuda for v in res[:4])


@pytest.mark.parametrize('case', gc.SAMPLER_CASES[:2], ids=lambda c: c['name'])
def test_sampler_takes_host_resident_sizes_as_the_reference_pipeline_passes_them(pk, case):
    """The reference's pipeline moves pcl_target to the GPU and leaves meta_data['pcl_target_size'] (and the valo id
    tensors) on the host (pipeline.py:75-91); the few-reads path reads sizes and device counts in one transfer, which
    must not mix devices."""
    g = load_golden('g13_sampler_' + case['name'])
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **gc.sampler_config(case))
    np.random.seed(case['seed'])
    torch.manual_seed(case['seed'])
    res = sampler([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z) for z in sizes],
                  torch.from_numpy(valo), torch.from_numpy(num_valo), case['time_idx'])
    for key, val in zip(KEYS, res):
        assert np.array_equal(val.cpu().numpy(), g[key]), key


def test_filter_air_solid_gap_matches_oracle(pk):
    from oracle import sampler as osamp
    rng = np.random.default_rng(4)
    rows = rng.uniform(-5, 5, size=(3000, 7)).astype(np.float32)
    tgt = rng.uniform(-5, 5, size=(2000, 3)).astype(np.float32)
    kept, dist, ratio = pk.geometry.filter_air_solid_gap(torch.from_numpy(rows).cuda(), torch.from_numpy(tgt).cuda(), 512,
                                                         0.35)
    r_kept, r_dist, r_ratio = osamp.air_solid_gap(torch.from_numpy(rows), torch.from_numpy(tgt), 0.35)
    assert np.array_equal(kept.cpu().numpy(), r_kept.numpy())
    assert np.array_equal(dist.cpu().numpy(), r_dist.numpy())
    assert abs(float(ratio) - float(r_ratio)) < 1e-6
    # nothing survives / everything survives
    k0, d0, _ = pk.geometry.filter_air_solid_gap(torch.from_numpy(rows).cuda(), torch.from_numpy(tgt).cuda(), 512, 1e3)
    assert k0.shape == (0, 7) and d0.shape == (0,)
    k1, _, _ = pk.geometry.filter_air_solid_gap(torch.from_numpy(rows).cuda(), torch.from_numpy(tgt).cuda(), 512, -1.0)
    assert np.array_equal(k1.cpu().numpy(), rows)


@pytest.mark.parametrize('kind,n,nq,r', [('uniform', 5000, 4000, 0.35), ('uniform', 57344, 20000, 0.2), ('lattice', 4096, 3000, 0.5),
                                          ('single', 1, 500, 0.3), ('wide', 3000, 2000, 0.05), ('dups', 2000, 1500, 0.25),
                                          ('flat', 4000, 3000, 0.3)])
def test_radius_grid_decisions_equal_the_1nn_search(pk, kind, n, nq, r):
    """ops.RadiusGrid.far (uniform grid, 27 cells per query) == (1-NN distance > r) of the streaming kNN kernel, bit for
    bit: lattice targets put many queries EXACTLY at the radius, queries reach far outside the targets' box, a cloud
    wider than 64 cells of the radius gets coarser cells, a flat cloud has one cell along an axis."""
    rng = np.random.default_rng(n + nq)
    if kind == 'lattice':
        g = np.arange(16, dtype=np.float32) * 0.5
        tgt = np.stack(np.meshgrid(g, g, g, indexing='ij'), -1).reshape(-1, 3)
        q = tgt[rng.integers(0, len(tgt), nq)] + rng.choice([0.0, 0.5, -0.5, 0.25], size=(nq, 3)).astype(np.float32)
    elif kind == 'single':
        tgt = np.array([[0.1, -0.2, 0.3]], dtype=np.float32)
        q = rng.normal(scale=0.3, size=(nq, 3)).astype(np.float32) + tgt
    elif kind == 'wide':
        tgt = rng.uniform(-40, 40, size=(n, 3)).astype(np.float32)
        q = tgt[rng.integers(0, n, nq)] + rng.normal(scale=0.04, size=(nq, 3)).astype(np.float32)
    elif kind == 'dups':
        tgt = np.repeat(rng.uniform(-2, 2, size=(n // 4, 3)).astype(np.float32), 4, axis=0)
        q = rng.uniform(-3, 3, size=(nq, 3)).astype(np.float32)
    elif kind == 'flat':
        tgt = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
        tgt[:, 2] = 1.25
        q = rng.uniform(-5, 5, size=(nq, 3)).astype(np.float32)
        q[:, 2] = rng.choice([1.25, 1.0, 1.6, 9.0], size=nq)
    else:
        tgt = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
        q = rng.uniform(-8, 8, size=(nq, 3)).astype(np.float32)
    tg, qg = torch.from_numpy(tgt).cuda(), torch.from_numpy(q.astype(np.float32)).cuda()
    _, dist = pk.ops.knn(qg, tg, 1, metric=1, return_dist=True)
    want = (dist[:, 0] > r)
    grid = pk.ops.RadiusGrid(tg, r)
    far = grid.far(qg, r)
    assert torch.equal(far > 0.5, want), int(((far > 0.5) != want).sum())
    half = grid.far(qg, 0.5 * r)                       # any radius up to the one the grid was built for
    assert torch.equal(half > 0.5, dist[:, 0] > 0.5 * r)
    if kind == 'lattice':
        assert int((dist[:, 0] == r).sum()) > 100       # the boundary case is exercised: exactly at the radius = not far
    rows = torch.cat([qg, torch.arange(nq, device='cuda', dtype=torch.float32)[:, None]], dim=1)
    kept = pk.geometry._gap_rows(rows, tg, r)
    assert torch.equal(kept, pk.geometry.filter_air_solid_gap(rows, tg, 0, r)[0])


def test_sampler_grid_filter_switch_changes_nothing(pk, monkeypatch):
    """The sampler's outputs with the grid radius test equal those with the 1-NN search (same RNG stream)."""
    case = dict(name='sw', kind='carla', bias='low_moving_vehped_sembal', frames=3, m=20000, num_solid=2048,
                num_air=3000, time_idx=1, segm=True, seed=78)
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    cfg = gc.sampler_config(case)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(pk.geometry, 'GRID_GAP_FILTER', on)
        sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **cfg)
        np.random.seed(5)
        torch.manual_seed(5)
        outs.append(sampler([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z).cuda() for z in sizes],
                            torch.from_numpy(valo).cuda(), torch.from_numpy(num_valo).cuda(), case['time_idx']))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('kind,bias,batch,segm', [
    ('carla', 'low_moving_vehped_sembal', 1, True), ('carla', 'low_moving_vehped_sembal', 2, False),
    ('carla', 'sembal', 1, True), ('carla', 'vehped', 2, True), ('carla', 'low', 1, True), ('carla', 'moving', 1, True),
    ('carla', 'none', 1, True), ('greater', 'none', 2, False), ('greater', 'low_moving', 1, False)])
def test_sampler_few_reads_path_equals_the_step_by_step_path(pk, monkeypatch, kind, bias, batch, segm):
    """geometry.SAMPLER_FAST (three device->host reads per element, every draw made up front) returns the points of the
    step-by-step path (G13: the reference's) bit for bit and leaves both host generators in the same state."""
    case = dict(name='fast', kind=kind, bias=bias, frames=3, m=9000, num_solid=1536, num_air=2200, time_idx=1, segm=segm,
                seed=81, batch=batch)
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    cfg = gc.sampler_config(case)
    dev = ([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z).cuda() for z in sizes],
           torch.from_numpy(valo).cuda(), torch.from_numpy(num_valo).cuda())
    outs, tails = [], []
    for fast in (True, False):
        monkeypatch.setattr(pk.geometry, 'SAMPLER_FAST', fast)
        sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **cfg)
        if fast and pk.geometry.GRID_GAP_FILTER:            # the few-reads path must not fall back for these inputs
            monkeypatch.setattr(sampler, '_element', lambda *a, **k: (_ for _ in ()).throw(AssertionError('fell back')))
        np.random.seed(6)
        torch.manual_seed(6)
        outs.append([sampler(*dev, t) for t in range(3)])
        tails.append((np.random.rand(), float(torch.rand(1))))
    for ra, rb in zip(*outs):
        for a, b in zip(ra, rb):
            assert a.shape == b.shape and a.device == b.device and torch.equal(a, b)
    assert tails[0] == tails[1]


def test_sampler_few_reads_path_falls_back_and_reports_like_the_step_by_step_path(pk, monkeypatch):
    """Semantic ids the grouped selection does not cover (negative / non-integral) send the element to the step-by-step
    path before any draw; a cloud that is too small raises the reference's error; a gap filter that keeps too few
    candidates warns once per doubling."""
    case = dict(name='fb', kind='carla', bias='low_moving_vehped_sembal', frames=3, m=6000, num_solid=512, num_air=800,
                time_idx=1, segm=True, seed=82)
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    cfg = gc.sampler_config(case)
    for f in frames:
        f[0, :40, 5] = -2.0
        f[0, 40:60, 5] = 3.5
    dev = ([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z).cuda() for z in sizes],
           torch.from_numpy(valo).cuda(), torch.from_numpy(num_valo).cuda())
    outs = []
    for fast in (True, False):
        monkeypatch.setattr(pk.geometry, 'SAMPLER_FAST', fast)
        sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **cfg)
        np.random.seed(7)
        torch.manual_seed(7)
        outs.append(sampler(*dev, 1))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    monkeypatch.setattr(pk.geometry, 'SAMPLER_FAST', True)
    small = [f[:, :200].contiguous() for f in dev[0]]
    with pytest.raises(RuntimeError, match='cur_tgt_pcl_count'):
        pk.geometry.GuidedImplicitPointSampler(_Log(), **cfg)(small, [z.clamp(max=200) for z in dev[1]], dev[2], dev[3], 1)
    # dense target cloud, large radius: few air candidates survive -> doublings are reported, counts still exact
    dense = dict(cfg, point_occupancy_radius=1.2)
    warns = []
    for fast in (True, False):
        monkeypatch.setattr(pk.geometry, 'SAMPLER_FAST', fast)
        log = _Log()
        np.random.seed(8)
        torch.manual_seed(8)
        try:
            res = pk.geometry.GuidedImplicitPointSampler(log, **dense)(*dev, 1)
            warns.append((log.warnings, [r.clone() for r in res]))
        except RuntimeError as e:
            warns.append((str(e), None))
    assert warns[0][0] == warns[1][0]
    if warns[0][1] is not None:
        assert warns[0][0] > 0
        for a, b in zip(warns[0][1], warns[1][1]):
            assert torch.equal(a, b)


def test_side_stream_sampler_draws_the_same_points(pk):
    """training.SideStreamSampler (the next step's points drawn on a side stream) consumes the random stream exactly as
    direct calls do: two consecutive draws equal two consecutive rounds of direct sampler calls."""
    case = dict(name='ss', kind='carla', bias='low_moving_vehped_sembal', frames=3, m=20000, num_solid=1024,
                num_air=1500, time_idx=1, segm=True, seed=79)
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    cfg = gc.sampler_config(case)
    dev = ([torch.from_numpy(f).cuda() for f in frames], [torch.from_numpy(z).cuda() for z in sizes],
           torch.from_numpy(valo).cuda(), torch.from_numpy(num_valo).cuda())
    sampler = pk.geometry.GuidedImplicitPointSampler(_Log(), **cfg)
    np.random.seed(9)
    torch.manual_seed(9)
    direct = []
    for _ in range(2):
        qs, ts = [], []
        for t in range(3):
            (si, ai, st, at, _, _) = sampler(*dev, t)
            qs.append(torch.cat([si, ai], dim=1)[0])
            ts.append(torch.cat([st, at], dim=1)[0])
        direct.append((torch.stack(qs), torch.stack(ts)))
    np.random.seed(9)
    torch.manual_seed(9)
    side = pk.training.SideStreamSampler(sampler, 3)
    busy = torch.zeros((4096, 4096), device='cuda')
    for i in range(2):
        # a real training loop uploads new target frames on the MAIN stream every step: fresh device copies
        # behind queued main-stream work, an event right after the upload, and the side stream waits for that event only
        for _ in range(3):
            busy = busy * 1.0001 + 1.0           # main-stream work queued in front of the upload
        fresh = ([f.clone() for f in dev[0]], [z.clone() for z in dev[1]], dev[2].clone(), dev[3].clone())
        ready = torch.cuda.Event()
        ready.record()
        for _ in range(3):
            busy = busy * 1.0001 + 1.0           # main-stream work the draw must not wait for
        side.draw(*fresh, ready=ready)
        del fresh                                # (record_stream keeps the allocator from recycling them under the draw)
        q, tgt = side.take()
        assert torch.equal(q, direct[i][0]) and torch.equal(tgt, direct[i][1])
    assert q.shape == (3, 2524, 4) and tgt.shape == (3, 2524, 6)
    # without an event the draw orders itself behind everything queued on the current stream so far: still the same points
    np.random.seed(9)
    torch.manual_seed(9)
    side.draw(*dev)
    q0, t0 = side.take()
    assert torch.equal(q0, direct[0][0]) and torch.equal(t0, direct[0][1])


def test_sampler_invariants_at_training_size(pk):
    """BASELINE config 5 sizes (num_solid 7168, num_air 10035, ~57 K target points, CARLA cuboid): every solid
    query lies within radius/2 of a target point and carries that point's colour / tag; every air query is farther
    than the radius from ALL target points and inside the output cuboid; counts and constant columns are exact."""
    case = dict(name='big', kind='carla', bias='low_moving_vehped_sembal', frames=3, m=57344, num_solid=7168,
                num_air=10035, time_idx=1, segm=True, seed=77)
    frames, sizes, valo, num_valo = gc.sampler_inputs(case)
    cfg = gc.sampler_config(case)
    log = _Log()
    sampler = pk.geometry.GuidedImplicitPointSampler(log, **cfg)
    np.random.seed(1)
    torch.manual_seed(1)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    (sq, aq, st, at, ss, as_) = sampler(dev, [torch.from_numpy(z).cuda() for z in sizes], torch.from_numpy(valo).cuda(),
                                        torch.from_numpy(num_valo).cuda(), case['time_idx'])
    assert sq.shape == (1, 7168, 4) and aq.shape == (1, 10035, 4) and st.shape == (1, 7168, 6) and at.shape == (1, 10035, 6)
    r = cfg['point_occupancy_radius']
    tgt = pk.geometry.filter_pcl_bounds_carla_output_torch(dev[1][0, :int(sizes[1][0])], min_z=-1.0, other_bounds=16.0)
    _, d_solid = pk.ops.knn(sq[0], tgt, 1, metric=1, return_dist=True)
    assert float(d_solid.max()) <= r / 2.0 + 1e-6
    _, d_air = pk.ops.knn(aq[0], tgt, 1, metric=1, return_dist=True)
    assert float(d_air.min()) > r
    assert torch.all(sq[0, :, 3] == 1.0) and torch.all(aq[0, :, 3] == 1.0)             # t = time_idx
    assert torch.all(st[0, :, 0] == 1.0) and torch.all(at[0, :, 0] == 0.0) and torch.all(at[0, :, 1:] == -1.0)
    assert torch.all((st[0, :, 5] >= 0) & (st[0, :, 5] < 13))
    # regular air points (the last block) are inside the output cuboid
    n_reg = 10035 - sum(int(as_[0, j] * 10035) for j in (1, 2, 3))
    reg = aq[0, -n_reg:]
    assert torch.all((reg[:, 0] >= 0) & (reg[:, 0] <= 40.0) & (reg[:, 1].abs() <= 16.0) & (reg[:, 2] >= -1.0) &
                     (reg[:, 2] <= 6.4))
    assert abs(float(ss.sum()) - 1.0) < 1e-5 and abs(float(as_.sum()) - 1.0) < 1e-5
