"""The exact grid kNN (csrc/gridrad.hip, occ4d_knn_grid_f32) returns the brute-force kernel's lists bit for bit:
indices AND distances, both metrics, every k, on clouds that stress what a grid could get wrong -- exact ties (lattices,
duplicates, the zero padding rows of the reference's data path), degenerate boxes (flat / collinear / all-equal clouds),
strongly non-uniform density (tight clusters far apart: many empty rings), queries far outside the data's box, fewer
data points than one cell row, strided rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(kind, n, rng):
    if kind == 'uniform':
        return rng.uniform(-5, 5, size=(n, 3))
    if kind == 'scene':                        # flat-ish extent like the CARLA cuboid
        return rng.uniform([0, -16, -1], [40, 16, 6.4], size=(n, 3))
    if kind == 'lattice':
        return rng.integers(0, 10, size=(n, 3)).astype(np.float64)
    if kind == 'padded':
        p = rng.uniform(-5, 5, size=(n, 3))
        p[n - n // 4:] = 0.0
        return p
    if kind == 'flat':
        p = rng.uniform(-3, 3, size=(n, 3))
        p[:, 2] = 1.5
        return p
    if kind == 'line':
        p = np.zeros((n, 3))
        p[:, 0] = rng.uniform(-100, 100, size=n)
        return p
    if kind == 'clusters':
        c = rng.uniform(-50, 50, size=(5, 3))
        return c[rng.integers(0, 5, size=n)] + rng.normal(scale=0.01, size=(n, 3))
    if kind == 'all_equal':
        return np.full((n, 3), 0.25)
    raise ValueError(kind)


def _both(pk, q, d, k, metric):
    lib, ops = pk._lib.lib(), pk.ops
    qg, dg = torch.from_numpy(q.astype(np.float32)).cuda(), torch.from_numpy(d.astype(np.float32)).cuda()
    out = []
    for grid in (False, True):
        old = (ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS)
        ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS = grid, 1, 1
        try:
            out.append(ops.knn(qg, dg, k, metric=metric, return_dist=True))
        finally:
            ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS = old
    return out


@pytest.mark.parametrize('kind,n', [('uniform', 5000), ('scene', 28672), ('lattice', 4096), ('padded', 2048), ('flat', 3000),
                                    ('line', 1500), ('clusters', 6000), ('all_equal', 700), ('uniform', 40)])
@pytest.mark.parametrize('k,metric', [(16, 0), (14, 0), (8, 1), (1, 1), (12, 0)])
def test_grid_knn_equals_brute_force_self(kind, n, k, metric):
    import occlusions4d_amd as pk
    rng = np.random.default_rng(n + 7 * k)
    d = _cloud(kind, n, rng)
    (bi, bd), (gi, gd) = _both(pk, d, d, k, metric)
    assert gi.dtype == torch.int32 and torch.equal(gi, bi), int((gi != bi).any(dim=1).sum())
    assert torch.equal(gd, bd)


@pytest.mark.parametrize('offset', [1e3, 1e5, 3e6])
@pytest.mark.parametrize('k,metric', [(16, 0), (8, 1)])
def test_grid_knn_on_unnormalised_world_coordinates(offset, k, metric):
    """clouds whose coordinates are large against their extent (|origin| / cell edge up to ~1e7: one ulp of the
    origin is then a sizeable fraction of a cell).  The ring-termination margin scales with the coordinates' ulp, so the
    grid lists stay those of the brute-force kernel bit for bit (fp32 coordinates are what they are: both see the same)."""
    import occlusions4d_amd as pk
    rng = np.random.default_rng(int(offset) + k)
    d = (rng.uniform(-5, 5, size=(6000, 3)) + np.array([offset, -0.7 * offset, 0.3 * offset])).astype(np.float32)
    (bi, bd), (gi, gd) = _both(pk, d, d, k, metric)
    assert torch.equal(gi, bi) and torch.equal(gd, bd)


@pytest.mark.parametrize('kind,nd,nq', [('scene', 4248, 20000), ('uniform', 531, 3000), ('clusters', 3000, 4000),
                                        ('lattice', 1000, 2000), ('flat', 2000, 1500)])
@pytest.mark.parametrize('k,metric', [(14, 0), (8, 1)])
def test_grid_knn_equals_brute_force_cross_with_far_queries(kind, nd, nq, k, metric):
    """Queries from inside the data's box, from just outside and from 3 x its extent away (the decoder's grid queries
    against the abstract cloud); query rows strided (x, y, z, t)."""
    import occlusions4d_amd as pk
    rng = np.random.default_rng(nd + nq + k)
    d = _cloud(kind, nd, rng)
    lo, hi = d.min(0), d.max(0)
    ext = np.maximum(hi - lo, 1.0)
    q = np.concatenate([rng.uniform(lo, hi, size=(nq // 2, 3)), rng.uniform(lo - 0.2 * ext, hi + 0.2 * ext, size=(nq // 4, 3)),
                        rng.uniform(lo - 3 * ext, hi + 3 * ext, size=(nq - nq // 2 - nq // 4, 3))])
    q4 = np.concatenate([q, np.zeros((nq, 1))], axis=1)
    lib, ops = pk._lib.lib(), pk.ops
    qg = torch.from_numpy(q4.astype(np.float32)).cuda()[:, :3]             # strided view
    (bi, bd), (gi, gd) = _both(pk, q4[:, :3], d, k, metric)
    assert torch.equal(gi, bi) and torch.equal(gd, bd)
    old = (ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS)
    ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS = True, 1, 1
    try:
        si = ops.knn(qg, torch.from_numpy(d.astype(np.float32)).cuda(), k, metric=metric)
    finally:
        ops.KNN_GRID, ops.KNN_GRID_MIN_DATA, ops.KNN_GRID_MIN_PAIRS = old
    assert torch.equal(si, bi)


def test_grid_knn_is_what_large_searches_use_and_rejects_bad_arguments():
    import ctypes as C
    import occlusions4d_amd as pk
    ops, lib = pk.ops, pk._lib.lib()
    assert ops.KNN_GRID and ops.KNN_GRID_MIN_PAIRS == 1 << 26
    p = torch.rand(6000, 3, device='cuda')
    ws = torch.empty((int(lib.occ4d_radius_grid_workspace_bytes(6000)) + 3) // 4, device='cuda')
    idx = torch.empty((6000, 16), dtype=torch.int32, device='cuda')
    P = ops._ptr
    assert lib.occ4d_knn_grid_f32(P(p), 3, 6000, P(p), 3, 6000, 17, 0, P(idx), None, P(ws), None) == pk._lib.EINVAL
    assert lib.occ4d_knn_grid_f32(P(p), 3, 6000, P(p), 3, 10, 16, 0, P(idx), None, P(ws), None) == pk._lib.EINVAL
    assert lib.occ4d_knn_grid_f32(P(p), 3, 6000, P(p), 3, 6000, 16, 2, P(idx), None, P(ws), None) == pk._lib.EINVAL
    assert lib.occ4d_knn_grid_f32(P(p), 3, 6000, P(p), 3, 6000, 16, 0, P(idx), None, None, None) == pk._lib.EINVAL
    assert lib.occ4d_knn_grid_f32(P(p), 3, 0, P(p), 3, 6000, 16, 0, P(idx), None, P(ws), None) == 0


def test_library_layer_uses_the_grid_for_a_large_self_knn_and_gets_the_same_block():
    """The path-level entry point computes the neighbour lists itself when none are passed (occ4d_pt_layer_fwd_f32 with
    knn_idx = NULL: the C-ABI-only binder's case); from 2^26 pairs it takes the grid search.  Same output as with the
    brute-force lists handed in."""
    import occlusions4d_amd as pk
    n, dim = 9000, 36
    rng = np.random.default_rng(12)
    x = torch.from_numpy(rng.normal(size=(1, n, dim)).astype(np.float32)).cuda()
    pos = torch.from_numpy(rng.uniform([0, -16, -1], [40, 16, 6.4], size=(1, n, 3)).astype(np.float32)).cuda()
    blk = pk.modules.PointTransformerBlock(dim, dim, dim, num_neighbors=16).cuda().eval()
    blk.load_state_dict(pk.configs.fill_state_dict(blk, 99))
    old = pk.ops.KNN_GRID
    try:
        pk.ops.KNN_GRID = False
        idx = pk.ops.knn(pos[0], pos[0], 16, metric=0)
    finally:
        pk.ops.KNN_GRID = old
    with torch.no_grad():
        given = blk(x, pos, knn_idx=idx[None])[0]
        own = blk(x, pos)[0]
    assert torch.equal(given, own)
