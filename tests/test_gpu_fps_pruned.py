"""The spatially pruned single-workgroup FPS (csrc/fps_bucket.hip, 1536 .. 16384 points) selects exactly what the
reference's greedy scan selects: same order, lowest index on ties (oracle/cluster.py restates torch_cluster.fps,
which the reference's DownTransition calls: model/point_transformer/modules.py:67-80).  The pruning reorders the points
along a Morton curve and skips whole buckets, so the cases here stress what that could break: exact ties (lattice
points, duplicates), degenerate boxes (planar / collinear / all-equal clouds), padding buckets (n not a multiple of
2048), strided rows, and the cooperative multi-workgroup kernel (an independent exhaustive implementation) as a second
reference at sizes where the numpy oracle is slow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pk():
    import importlib
    return importlib.import_module('occlusions-4d_amd')


def _oracle_order(p, m):
    p = p.astype(np.float32)

    def sq(i):
        d = p - p[i]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    order = [0]
    mind = sq(0)
    for _ in range(1, m):
        nxt = int(np.argmax(mind))            # first (lowest-index) maximum
        order.append(nxt)
        mind = np.minimum(mind, sq(nxt))
    return np.array(order, dtype=np.int32)


def _cloud(kind, n, rng):
    if kind == 'uniform':
        return rng.uniform(-5, 5, size=(n, 3))
    if kind == 'lattice':                      # integer lattice with repeats: almost every step is a tie
        return rng.integers(0, 12, size=(n, 3)).astype(np.float64)
    if kind == 'half_lattice':                 # ties between distinct points at fractional coordinates
        return rng.integers(-40, 40, size=(n, 3)) * 0.125
    if kind == 'planar':                       # zero-extent axis: Morton scale 0 on z
        p = rng.uniform(-3, 3, size=(n, 3))
        p[:, 2] = 1.5
        return p
    if kind == 'line':
        p = np.zeros((n, 3))
        p[:, 0] = rng.uniform(-100, 100, size=n)
        return p
    if kind == 'clusters':                     # far-apart tight clusters + duplicates + zero padding rows
        c = rng.uniform(-50, 50, size=(7, 3))
        p = c[rng.integers(0, 7, size=n)] + rng.normal(scale=0.01, size=(n, 3))
        p[100:400] = p[1000:1300]
        p[-50:] = 0.0
        return p
    if kind == 'all_equal':
        return np.full((n, 3), 0.25)
    raise ValueError(kind)


@pytest.mark.parametrize('kind,n,m', [
    ('uniform', 9600, 3200), ('uniform', 10241, 10241), ('uniform', 12289, 900), ('uniform', 14336, 4779),
    ('uniform', 14337, 1200), ('uniform', 16384, 2000), ('uniform', 15361, 900),
    ('lattice', 9700, 2500), ('lattice', 14336, 3000), ('half_lattice', 10000, 2400), ('planar', 11000, 1700),
    ('line', 9601, 1400), ('clusters', 12000, 2000), ('all_equal', 9800, 50),
    # the smaller encoder levels: one to five buckets per wave, part of them padding
    ('uniform', 2049, 683), ('uniform', 3000, 3000), ('uniform', 7168, 1000), ('uniform', 7169, 1000),
    ('uniform', 9558, 3186), ('lattice', 2500, 2500), ('half_lattice', 7000, 2400), ('all_equal', 2100, 50),
    ('uniform', 1536, 512), ('uniform', 1593, 531), ('lattice', 1600, 1600), ('all_equal', 1700, 40),
    ('clusters', 4779, 1593), ('planar', 2048, 2048), ('line', 4097, 700),
    # below the pruned kernel's range: the exhaustive kernel
    ('uniform', 1535, 500), ('lattice', 1200, 1200), ('all_equal', 900, 30)])
def test_pruned_fps_matches_the_greedy_scan(pk, kind, n, m):
    rng = np.random.default_rng(n * 7 + m)
    p = _cloud(kind, n, rng).astype(np.float32)
    sel, order = pk.ops.fps(torch.from_numpy(p).cuda(), m, return_order=True)
    ref = _oracle_order(p, m)
    got = order.cpu().numpy()
    assert np.array_equal(got, ref), (kind, n, m, int(np.argmax(got != ref)))
    expect_sorted = np.unique(ref)             # (duplicates in `ref` only once all distances are 0)
    assert np.array_equal(sel.cpu().numpy()[:expect_sorted.size], expect_sorted.astype(np.int32))


def test_pruned_fps_tie_storms(pk):
    """Clouds on a coarse integer grid (few distinct coordinates: running mins tie in large groups, floors tie with
    candidates, several waves publish the same distance): the multi-sample rounds must still make the sequential scan's
    picks, lowest index first."""
    rng = np.random.default_rng(20260928)
    for trial in range(10):
        n = int(rng.integers(1536, 5000))
        k = int(rng.integers(2, 7))
        p = rng.integers(0, k, size=(n, 3)).astype(np.float32) * np.float32(0.5)
        if trial % 2:
            p += rng.integers(0, 2, size=(n, 3)).astype(np.float32) * np.float32(1e-3)   # near-duplicates
        m = int(rng.integers(n // 8, n // 2))
        sel, order = pk.ops.fps(torch.from_numpy(p).cuda(), m, return_order=True)
        ref = _oracle_order(p, m)
        got = order.cpu().numpy()
        assert np.array_equal(got, ref), (trial, n, m, k, int(np.argmax(got != ref)))


def test_pruned_fps_against_the_cooperative_kernel_random(pk):
    rng = np.random.default_rng(2718)
    kinds = ['uniform', 'lattice', 'half_lattice', 'planar', 'clusters']
    for trial in range(12):
        n = int(rng.integers(2049, 16385)) if trial % 3 == 0 else int(rng.integers(9600, 16385))
        m = int(rng.integers(n // 8, n // 2))
        p = torch.from_numpy(_cloud(kinds[trial % len(kinds)], n, rng).astype(np.float32)).cuda()
        a, ao = pk.ops.fps(p, m, return_order=True)
        b, bo = pk.ops.fps_coop(p, m, start=0, n_workgroups=int(rng.choice([2, 4, 7])), return_order=True)
        assert torch.equal(ao, bo), (trial, n, m, int((ao != bo).nonzero()[0]))
        distinct = torch.unique(ao).numel()        # (repeats only once every remaining distance is 0: lattice clouds)
        assert torch.equal(a[:distinct], b[:distinct])


def test_pruned_fps_strided_rows_and_model_clouds(pk):
    """xyz columns of the (N, 8) point-cloud rows (row stride 8), the encoder's three levels."""
    pcl = pk.configs.synthetic_pcl('greater', 14336, 12)[0].cuda()
    level = pcl[:, :3]
    for m in (4779, 1593, 531):
        a, ao = pk.ops.fps(level, m, return_order=True)
        b, bo = pk.ops.fps(level.contiguous(), m, return_order=True)
        assert torch.equal(ao, bo) and torch.equal(a, b)
        if level.shape[0] > 2048:
            c, co = pk.ops.fps_coop(level.contiguous(), m, start=0, n_workgroups=2, return_order=True)
            assert torch.equal(ao, co)
        assert torch.equal(a, torch.sort(ao)[0].to(a.dtype))
        level = level[a.long()]


@pytest.mark.parametrize('n,m,start', [(100, 34, 99), (2048, 683, 17), (4779, 1593, 4778), (9558, 1000, 5000),
                                        (14336, 2000, 7001), (16384, 900, 16383), (28672, 600, 12345)])
def test_single_workgroup_fps_with_a_start_index(pk, n, m, start):
    """occ4d_fps_start_f32 (random_start=True passes its draw): every single-workgroup kernel (256 / 512 threads,
    pruned) against the greedy scan begun at `start`."""
    rng = np.random.default_rng(n + start)
    p = _cloud('uniform' if n % 2 else 'half_lattice', n, rng).astype(np.float32)
    sel, order = pk.ops.fps(torch.from_numpy(p).cuda(), m, return_order=True, start=start)

    def sq(i):
        d = p - p[i]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    ref = [start]
    mind = sq(start)
    for _ in range(1, m):
        nxt = int(np.argmax(mind))
        ref.append(nxt)
        mind = np.minimum(mind, sq(nxt))
    assert np.array_equal(order.cpu().numpy(), np.array(ref, dtype=np.int32))
    assert np.array_equal(sel.cpu().numpy(), np.sort(np.array(ref, dtype=np.int32)))
    with pytest.raises(AssertionError):
        pk.ops.fps(torch.from_numpy(p).cuda(), m, start=n)


@pytest.mark.parametrize('kind,n', [('uniform', 14336), ('uniform', 5000), ('lattice', 4096), ('padded', 2048),
                                    ('duplicates', 3000), ('planar', 2500)])
def test_nested_levels_are_prefixes_of_the_first_selection_order(kind, n):
    """Exact-in-R refactoring (iv): with a deterministic start the farthest-point subsets of consecutive DownTransitions
    are the first m_l picks of level 0's selection order.  Three levels computed that way (modules.NestedFps: ONE FPS
    launch) against three explicit FPS launches on the successive sub-clouds: the same indices, bit for bit -- on
    tie-free clouds, integer lattices (exact distance ties at every step), clouds with coincident points and the
    zero-padded clouds of the reference's data path."""
    import occlusions4d_amd as pk
    rng = np.random.default_rng(1000 + n)
    if kind == 'uniform':
        p = rng.uniform(-5, 5, size=(n, 3))
    elif kind == 'lattice':
        p = rng.integers(0, 12, size=(n, 3)).astype(np.float64)          # many exact ties AND duplicates
    elif kind == 'padded':
        p = np.concatenate([rng.uniform(-5, 5, size=(n - n // 4, 3)), np.zeros((n // 4, 3))])
    elif kind == 'duplicates':
        base = rng.uniform(-5, 5, size=(n // 3, 3))
        p = np.concatenate([base, base, base])[rng.permutation(3 * (n // 3))]
    else:
        p = np.concatenate([rng.uniform(-5, 5, size=(n, 2)), np.zeros((n, 1))], axis=1)
    p = torch.from_numpy(p.astype(np.float32)).cuda()
    down = [pk.modules.DownTransition(8, 16, factor=3, knn_k=4, fps_random_start=False) for _ in range(3)]
    old = pk.modules.NESTED_FPS
    try:
        results = {}
        for nested_on in (False, True):
            pk.modules.NESTED_FPS = nested_on
            chain, cur, got = pk.modules.NestedFps(), p, []
            for d in down:
                inds, cur = d.sample(cur, nested=chain)
                got.append((inds.cpu(), cur.cpu()))
            results[nested_on] = got
    finally:
        pk.modules.NESTED_FPS = old
    for (ia, pa), (ib, pb) in zip(results[False], results[True]):
        assert ia.dtype == ib.dtype == torch.int32 and torch.equal(ia, ib) and torch.equal(pa, pb)
    # and a random start (the training default) never takes the prefix path
    rnd = pk.modules.DownTransition(8, 16, factor=3, knn_k=4, fps_random_start=True)
    chain = pk.modules.NestedFps()
    torch.manual_seed(3)
    rnd.sample(p, nested=chain)
    assert chain.order is None


@pytest.mark.parametrize('kind,n', [('uniform', 14336), ('uniform', 3584), ('lattice', 4096), ('padded', 2048),
                                    ('duplicates', 3000), ('uniform', 50)])
def test_pooling_neighbours_are_a_prefix_of_the_self_knn_lists(kind, n):
    """Exact-in-R refactoring (v): the knn_k nearest full-cloud points of a SAMPLED point (DownTransition.neighbours, one
    kNN launch) = the first knn_k entries of that point's row of the preceding block's self-kNN lists (K = 16), bit for
    bit -- also where distances tie (lattices, coincident points, the zero padding rows: lowest index first in both)."""
    import occlusions4d_amd as pk
    rng = np.random.default_rng(2000 + n)
    if kind == 'uniform':
        p = rng.uniform(-5, 5, size=(n, 3))
    elif kind == 'lattice':
        p = rng.integers(0, 12, size=(n, 3)).astype(np.float64)
    elif kind == 'padded':
        p = np.concatenate([rng.uniform(-5, 5, size=(n - n // 4, 3)), np.zeros((n // 4, 3))])
    else:
        base = rng.uniform(-5, 5, size=(n // 3, 3))
        p = np.concatenate([base, base, base])[rng.permutation(3 * (n // 3))]
    p = torch.from_numpy(p.astype(np.float32)).cuda()
    for k_self, k_pool, factor in ((16, 12, 4), (16, 16, 3), (8, 3, 2)):
        down = pk.modules.DownTransition(8, 16, factor=factor, knn_k=k_pool, fps_random_start=False)
        inds, p_sub = down.sample(p)
        launched = down.neighbours(p_sub, p)
        self_idx = pk.ops.knn(p, p, k_self, metric=0)
        derived = pk.modules.pool_neighbours_from_self_knn(self_idx, inds, k_pool)
        assert derived.dtype == launched.dtype == torch.int32 and derived.shape == launched.shape
        assert torch.equal(derived, launched)


def test_encoder_with_and_without_the_derived_geometry_is_bit_identical():
    """PointCompletionNetV3.forward with refactorings (iv) and (v) on (default) and off: the same floats."""
    import occlusions4d_amd as pk
    torch.manual_seed(5)
    net = pk.model.PointCompletionNetV3(n_input=2048, d_in=8, d_feat=16, down_blocks=3, transition_factor=3,
                                        pt_num_neighbors=16, down_neighbors=12, abstract_levels=2,
                                        fps_random_start=False).cuda().eval()
    pcl = torch.randn(2, 2048, 8, device='cuda')
    pcl[1, 1500:] = 0.0
    old = (pk.modules.NESTED_FPS, pk.modules.POOL_FROM_SELF_KNN)
    try:
        outs = []
        for on in (True, False):
            pk.modules.NESTED_FPS = pk.modules.POOL_FROM_SELF_KNN = on
            with torch.no_grad():
                (a, g, coords) = net(pcl, True)
            outs.append((a.clone(), g.clone(), [c.clone() for c in coords]))
    finally:
        (pk.modules.NESTED_FPS, pk.modules.POOL_FROM_SELF_KNN) = old
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(x, y) for x, y in zip(outs[0][2], outs[1][2]))


# ------------------------------------------------------------------ round 5: the nested levels as ONE library kernel
@pytest.mark.parametrize('n,m_levels', [(14336, (4779, 1593, 531)), (28672, (9558, 3186, 1062)), (2048, (683, 228, 76)),
                                        (100, (34, 12, 4)), (3, (1, 1, 1))])
def test_nested_level_kernel_equals_searchsorted_sort(pk, n, m_levels):
    """occ4d_nested_fps_level_i32 (binary search of every pick in the sorted original indices + bit-set compaction)
    against the torch expression it replaces (searchsorted + sort + index), level by level down a chain."""
    rng = np.random.default_rng(n)
    p = torch.from_numpy(rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)).cuda()
    inds, order = pk.ops.fps_auto(p, m_levels[0], start=0, return_order=True)
    orig_t = inds.long()
    orig_k = inds
    for m in m_levels[1:]:
        pos_t = torch.sort(torch.searchsorted(orig_t, order[:m].long()))[0]
        pos_k, nxt_k = pk.ops.nested_fps_level(order, orig_k, m)
        assert pos_k.dtype == torch.int32 and torch.equal(pos_k.long(), pos_t)
        orig_t = orig_t[pos_t]
        assert torch.equal(nxt_k.long(), orig_t)
        orig_k = nxt_k


def test_nested_level_kernel_with_repeated_picks(pk):
    """A constant cloud (fewer distinct points than samples): the selection order repeats index 0 and so does the sorted
    subset; the kernel returns what the torch expression returned (repeated positions), nothing uninitialised."""
    p = torch.zeros((64, 3), device='cuda')
    inds, order = pk.ops.fps(p, 22, return_order=True)
    assert set(order.tolist()) == {0} and set(inds.tolist()) == {0}
    pos = torch.full((8,), -7, dtype=torch.int32, device='cuda')
    pos, nxt = pk.ops.nested_fps_level(order, inds, 8)
    want = torch.sort(torch.searchsorted(inds.long(), order[:8].long()))[0]
    assert torch.equal(pos.long(), want) and torch.equal(nxt, inds[pos.long()])


@pytest.mark.parametrize('kind', ['greater', 'carla'])
def test_inference_step_launches_library_kernels_only(pk, kind):
    """Round 5: one encode + decode (model.forward, the decoder's mini-batches, the squash) issues no
    ATen kernel -- the nested-FPS prefix, the pooling-list slices, the xyz packing and the pos | features | level-id
    layout of the abstract cloud are library kernels now."""
    from torch.profiler import ProfilerActivity, profile
    pa, ia, inf = pk.configs.model_args(kind, 2048)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, 5)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    pcl = pk.configs.synthetic_pcl(kind, 2048, 4, 5).cuda()
    q = pk.geometry.sample_implicit_points_blind_device(4096, inf['min_z'], inf['cube_bounds'], 1, inf['data_kind'],
                                                        inf['cube_mode'], 'grid', pcl.device)

    def step():
        with torch.no_grad():
            return pk.inference.infer_device(pcl, q, enc, dec, 2048, inf['color_mode'], inf['predict_segmentation'],
                                             'none', 13)
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA or 'kernel' in e.key.lower()]
    foreign = [k for k in names if 'at::native' in k or 'rocclr' in k or 'emcpy' in k or 'emset' in k]
    assert names and not foreign, foreign


def test_nested_level_with_repeated_picks_is_ascending_and_repeats_the_last():
    """a selection order that names a point more than once (a cloud with fewer distinct points than samples):
    occ4d_nested_fps_level_i32 returns the distinct positions first, ascending, and repeats the last of them behind --
    never the raw, unsorted search results.  The CPU twin's statement of the same rule is the reference here."""
    import occlusions4d_amd as pk
    orig = torch.tensor([2, 5, 7, 11, 13, 20, 21, 40], dtype=torch.int32)
    order = torch.tensor([13, 2, 13, 40, 2, 2, 7, 13], dtype=torch.int32)      # 4 distinct picks among the first 6
    pos, nxt = pk.ops.nested_fps_level(order.cuda(), orig.cuda(), 6)
    assert pos.tolist() == [0, 4, 7, 7, 7, 7] and nxt.tolist() == [2, 13, 40, 40, 40, 40]
    pos, nxt = pk.ops.nested_fps_level(order.cuda(), orig.cuda(), 8)           # + pick 7
    assert pos.tolist() == [0, 2, 4, 7, 7, 7, 7, 7] and nxt.tolist() == [2, 7, 13, 40, 40, 40, 40, 40]
    pos, _ = pk.ops.nested_fps_level(torch.tensor([5, 21, 2], dtype=torch.int32).cuda(), orig.cuda(), 3)   # no repeats
    assert pos.tolist() == [0, 1, 6]
