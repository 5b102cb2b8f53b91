"""GPU parity for the multi-workgroup farthest-point sampler: bit-identical to the
single-workgroup kernel and to the oracle's restated torch_cluster.fps for every workgroup count, arbitrary start
index, and -- at the dataloader's size (172 032 -> 14 336) -- the greedy invariant checked step by step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pk():
    import occlusions4d_amd
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    occlusions4d_amd._lib.lib()
    return occlusions4d_amd


def _cloud(n, seed, dup=False):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    if dup and n > 40:
        p[20:30] = p[5:15]            # exact duplicates: ties must go to the lowest index
        p[-3:] = 0.0                  # zero padding rows of the data pipeline
    return p


def _oracle_fps(p, m, start):
    """oracle/cluster.py's restatement with an explicit start index (numpy, fp32, no FMA)."""
    p = p.astype(np.float32)

    def sq(i):
        d = p - p[i]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    order = [start]
    mind = sq(start)
    for _ in range(1, m):
        nxt = int(np.argmax(mind))
        order.append(nxt)
        mind = np.minimum(mind, sq(nxt))
    return np.array(order)


@pytest.mark.parametrize('n,m,start,wgs', [
    (1, 1, 0, 0), (2, 2, 1, 0), (100, 34, 0, 1), (100, 34, 99, 3), (777, 259, 13, 0), (2048, 683, 0, 5),
    (5000, 1667, 4999, 16), (14336, 4779, 0, 0), (14336, 4779, 7000, 7), (28672, 9558, 0, 0)])
def test_fps_coop_matches_oracle_and_single_workgroup(pk, n, m, start, wgs):
    p = _cloud(n, n + m, dup=True)
    dev = torch.from_numpy(p).cuda()
    idx, order = pk.ops.fps_coop(dev, m, start=start, n_workgroups=wgs, return_order=True)
    order = order.cpu().numpy()
    idx = idx.cpu().numpy()
    if n <= 5000:
        assert np.array_equal(order, _oracle_fps(p, m, start))
    if start == 0:
        ref_idx, ref_order = pk.ops.fps(dev, m, return_order=True)
        assert np.array_equal(order, ref_order.cpu().numpy())
        assert np.array_equal(idx, ref_idx.cpu().numpy())
    assert np.array_equal(idx, np.sort(order)) and len(np.unique(idx)) == m or n > 40   # duplicates may repeat late
    assert np.array_equal(idx, np.sort(order))


def test_fps_coop_strided_rows(pk):
    """xyz as the leading columns of a wider point cloud (row stride 8), as the dataloader holds it."""
    rng = np.random.default_rng(3)
    pcl = torch.from_numpy(rng.uniform(-5, 5, size=(6000, 8)).astype(np.float32)).cuda()
    a = pk.ops.fps_coop(pcl[:, :3], 2000, start=17, return_order=True)[1]
    b = pk.ops.fps_coop(pcl[:, :3].contiguous(), 2000, start=17, return_order=True)[1]
    assert torch.equal(a, b)


def test_fps_coop_dataloader_size_greedy_invariant(pk):
    """172 032 points -> 14 336 (12 frames x 14 336, the clip the reference's loader subsamples): every sample is
    the lowest-index maximiser of the running min squared distance (checked with torch on the device, same
    fp32 operation order)."""
    n, m, start = 172032, 14336, 123457
    g = torch.Generator(device='cuda').manual_seed(7)
    p = (torch.rand((n, 3), device='cuda', generator=g) * 10 - 5)
    idx, order = pk.ops.fps_coop(p, m, start=start, return_order=True)
    order = order.long()
    assert int(order[0]) == start
    assert torch.equal(idx.long(), torch.sort(order)[0]) and torch.unique(order).numel() == m

    def sq(i):
        d = p - p[i]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    mind = sq(order[0])
    bad = 0
    for s in range(1, m):
        o = order[s]
        mx = mind.max()
        if s % 7 == 0 or s < 64:      # full lowest-index check on a subset of steps, max-value check on all
            first = torch.nonzero(mind == mx)[0, 0]
            bad += int(first != o)
        bad += int(mind[o] != mx)
        mind = torch.minimum(mind, sq(o))
    assert bad == 0


def test_fps_coop_argument_errors(pk):
    p = torch.zeros((10, 3), device='cuda')
    with pytest.raises(AssertionError):
        pk.ops.fps_coop(p, 11)
    with pytest.raises(AssertionError):
        pk.ops.fps_coop(p, 5, start=10)
    with pytest.raises(AssertionError):
        pk.ops.fps_coop(p, 5, n_workgroups=17)
    with pytest.raises(RuntimeError):
        pk.ops.fps_coop(p.cpu(), 5)


# ------------------------------------------------------------------ subsample_pad_pcl_torch (utils/geometry.py:294-376)
import golden_cases as gc  # noqa: E402
from conftest import load_golden  # noqa: E402


@pytest.mark.parametrize('case', gc.SUBSAMPLE_CASES, ids=lambda c: c['name'])
def test_subsample_pad_pcl_torch_matches_reference(pk, case):
    """Same seeds -> same random draws -> the reference's rows, bit for bit (the FPS start index is drawn with the
    same torch.randint call the oracle's torch_cluster stand-in makes)."""
    g = load_golden('g12_subsample')
    pcl = torch.from_numpy(gc.subsample_inputs(case)).cuda()
    for batched in (False, True):
        if batched and case['n'] > case['n_desired']:
            continue                       # the reference's subsample branch asserts shape[0] == n_desired: (N, D) only
        np.random.seed(case['seed'])
        torch.manual_seed(case['seed'])
        res = pk.geometry.subsample_pad_pcl_torch(pcl[None] if batched else pcl, case['n_desired'],
                                                  sample_mode=case['mode'], retain_vehped=bool(case.get('retain')),
                                                  segm_idx=case.get('segm_idx'))
        assert res.is_cuda
        res = res[0] if batched else res
        assert np.array_equal(res.cpu().numpy(), g[case['name']])


def test_subsample_pad_pcl_torch_errors_and_clip_size(pk):
    pcl = torch.zeros((10, 8), device='cuda')
    with pytest.raises(RuntimeError):
        pk.geometry.subsample_pad_pcl_torch(pcl, 11, subsample_only=True)
    with pytest.raises(AssertionError):
        pk.geometry.subsample_pad_pcl_torch(pcl, 5, sample_mode='grid')
    # the loader's real size: 12 frames x 14336 points -> n_points 14336; spacing invariant of an FPS subset:
    # its minimum pairwise distance is at least that of a random subset of the same size
    g = torch.Generator(device='cuda').manual_seed(11)
    clip = torch.rand((172032, 8), device='cuda', generator=g) * 10 - 5
    torch.manual_seed(5)
    np.random.seed(5)
    sub = pk.geometry.subsample_pad_pcl_torch(clip, 14336, sample_mode='farthest_point')
    rnd = pk.geometry.subsample_pad_pcl_torch(clip, 14336, sample_mode='random')
    assert sub.shape == rnd.shape == (14336, 8)

    def min_spacing(x):
        _, d = pk.ops.knn(x[:, :3].contiguous(), x[:, :3].contiguous(), 2, metric=1, return_dist=True)
        return float(d[:, 1].min())
    assert min_spacing(sub) > 3 * min_spacing(rnd)
    pk.ops.check_pending()


def test_down_transition_random_start_and_large_clouds(pk):
    """fps_random_start=True (the reference's training default) draws the first sample from torch's generator;
    clouds above the single-workgroup limit take the cooperative kernel and match the oracle's geometry."""
    from oracle import cluster
    rng = np.random.default_rng(9)
    p = torch.from_numpy(rng.uniform(-5, 5, size=(3001, 3)).astype(np.float32))
    dt = pk.modules.DownTransition(8, 16, factor=3, knn_k=12, fps_random_start=True).cuda()
    torch.manual_seed(42)
    inds, p_sub, nn_idx = dt.geometry(p.cuda())
    torch.manual_seed(42)
    ref = torch.sort(cluster.fps(p, None, 1.0 / 3.0, True))[0]
    assert np.array_equal(inds.cpu().numpy(), ref.numpy())
    big = torch.from_numpy(rng.uniform(-5, 5, size=(28672, 3)).astype(np.float32)).cuda()
    dt0 = pk.modules.DownTransition(8, 16, factor=3, knn_k=12, fps_random_start=False).cuda()
    a = dt0.geometry(big)[0]
    b = pk.ops.fps(big, 9558)
    assert torch.equal(a, b)
    pk.ops.check_pending()


@pytest.mark.parametrize('kind,n,m', [('lattice', 20000, 5000), ('half_lattice', 28672, 7168), ('planar', 18000, 4500),
                                      ('clusters', 28672, 7168), ('all_equal', 17000, 300), ('line', 16500, 4000)])
@pytest.mark.parametrize('wgs', [16, 5])
def test_fps_coop_several_samples_per_exchange_on_tie_storms(pk, kind, n, m, wgs):
    """Round 4: an exchange of the cooperative kernel yields every further sample it can PROVE (candidate + floor per
    workgroup).  The proof rests on strict comparisons; clouds where almost every step is an exact tie (integer / half-
    integer lattices with repeats, planar and collinear clouds, tight far-apart clusters with duplicates and padding rows,
    a constant cloud) must give the single-workgroup kernel's selection order, index for index."""
    from test_gpu_fps_pruned import _cloud as cloud
    rng = np.random.default_rng(n + m + wgs)
    p = torch.from_numpy(cloud(kind, n, rng).astype(np.float32)).cuda()
    ref_idx, ref_order = pk.ops.fps(p, m, return_order=True)
    idx, order = pk.ops.fps_coop(p, m, start=0, n_workgroups=wgs, return_order=True)
    assert torch.equal(order, ref_order)
    if int(torch.unique(order).numel()) == m:          # (once every distinct point is taken the rule re-picks: the sorted
        assert torch.equal(idx, ref_idx)               # list then has fewer than m entries and an unspecified tail)


# ------------------------------------------------------------------ time-out handling (round 5: no poisoned step)
@pytest.mark.parametrize('n,m,start,fail_round', [(28672, 7168, 0, 5), (28672, 9558, 123, 0), (16385, 5462, 0, 700),
                                                  (2048, 683, 7, 2)])
def test_fps_coop_forced_timeout_is_repaired_in_stream_order(pk, n, m, start, fail_round):
    """A bounded inter-workgroup spin that times out (declared here in a chosen round through occ4d_fps_coop_debug)
    must never hand undefined indices to a consumer: the gated single-workgroup launch behind the cooperative kernel
    recomputes the selection -- same indices, status word 2 -- and the host only warns.  The consumer here is queued on
    the stream right behind the launch, without any host look at the status (what a training step does)."""
    p = torch.from_numpy(_cloud(n, 5 * n + m, dup=True)).cuda()
    want_idx, want_order = pk.ops.fps(p, m, start=start, return_order=True)
    pk.ops.fps_coop_debug(fail_round=fail_round)
    try:
        pk.ops.check_pending()
        idx, order = pk.ops.fps_coop(p, m, start=start, n_workgroups=16, return_order=True, check=False)
        gathered = p[idx.long()]                                   # a consumer in stream order
        with pytest.warns(UserWarning, match='recomputed by the single-workgroup kernel'):
            pk.ops.check_pending()                                 # ...and the deferred look only warns
        assert torch.equal(idx, want_idx) and torch.equal(order, want_order)
        assert torch.equal(gathered, p[want_idx.long()])
        with pytest.warns(UserWarning, match='recomputed by the single-workgroup kernel'):
            idx2 = pk.ops.fps_coop(p, m, start=start, n_workgroups=16, check=True)
        assert torch.equal(idx2, want_idx)
        if n >= pk.ops.FPS_COOP_MIN_POINTS:                        # the route a training step takes
            idx3 = pk.ops.fps_auto(p, m, start=start)
            assert torch.equal(idx3, want_idx)
            with pytest.warns(UserWarning):
                pk.ops.check_pending()
    finally:
        pk.ops.fps_coop_debug()
        pk.ops.check_pending()
    # with the default settings the status is 0 again: no warning
    import warnings as _w
    with _w.catch_warnings():
        _w.simplefilter('error')
        idx4 = pk.ops.fps_coop(p, m, start=start, n_workgroups=16, check=True)
    assert torch.equal(idx4, want_idx)


def test_fps_coop_timeout_above_the_repair_range_retries_then_raises(pk):
    """Clouds above the single-workgroup kernel's 32768 points (the dataloader's whole clips, host-checked): a timed-out
    launch is relaunched; a time-out that persists (forced here) raises instead of returning undefined indices."""
    p = torch.from_numpy(_cloud(40000, 3)).cuda()
    pk.ops.fps_coop_debug(fail_round=1)
    try:
        with pytest.warns(UserWarning, match='relaunching'):
            with pytest.raises(RuntimeError, match='timed out'):
                pk.ops.fps_coop(p, 2000, check=True)
    finally:
        pk.ops.fps_coop_debug()
    assert pk.ops.fps_coop(p, 2000, check=True).shape == (2000,)


def test_captured_fps_coop_replays_the_repair(pk):
    """Inside a hipGraph the host never looks at the status word; the gated repair launch is captured with the
    cooperative kernel, so a replay that times out still leaves valid indices."""
    p = torch.from_numpy(_cloud(28672, 77)).cuda()
    want = pk.ops.fps(p, 7168)
    out = torch.zeros(7168, dtype=torch.int32, device='cuda')
    pk.ops.fps_coop_debug(fail_round=3)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pk.ops.fps_coop(p, 7168, n_workgroups=16, check=False)           # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side)
        with pytest.warns(UserWarning):
            pk.ops.check_pending()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out.copy_(pk.ops.fps_coop(p, 7168, n_workgroups=16, check=False))
        for _ in range(2):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want)
    finally:
        pk.ops.fps_coop_debug()
