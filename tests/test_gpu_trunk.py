"""GPU: the row-resident fused trunk kernels (csrc/trunk.hip) against fp64 torch on the same inputs and against the
generic Linear chain they replace: ResnetBlockFC (model/implicit.py:92-101) with and without the fused
interpolation term of the next block, and the 416-input Linear layers around the cross-attention."""
import numpy as np
import pytest
import torch

import occlusions4d_amd as pk

pytestmark = pytest.mark.gpu
H = 416


def _weights(rng, n_out, scale=0.05):
    w = torch.from_numpy((scale * rng.normal(size=(n_out, H))).astype(np.float32)).cuda()
    b = torch.from_numpy((0.1 * rng.normal(size=(n_out,))).astype(np.float32)).cuda()
    return w, b


def _interp(rng, n, m=531, k=8):
    ztab = torch.from_numpy(rng.normal(size=(m, 6 * H)).astype(np.float32)).cuda()
    zconst = torch.from_numpy(rng.normal(size=(6 * H,)).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.integers(0, m, size=(n, k)).astype(np.int32)).cuda()
    w = torch.from_numpy(rng.uniform(size=(n, k)).astype(np.float32)).cuda()
    w = (w / w.sum(1, keepdim=True)).contiguous()
    return ztab, zconst, idx, w


def _interp_ref(ztab, zconst, idx, w, i):
    z = ztab[:, i * H:(i + 1) * H].double()
    return zconst[i * H:(i + 1) * H].double() + (w.double()[:, :, None] * z[idx.long()]).sum(1)


def test_packing_layouts():
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.normal(size=(832, H)).astype(np.float32)).cuda()
    p = pk.ops.pack_trunk_rows(w).cpu().numpy().reshape(27, 52, 64, 4)
    wn = w.cpu().numpy()
    for (s, nt, t, g, r, e) in [(0, 0, 0, 0, 0, 0), (3, 1, 25, 3, 15, 3), (25, 0, 7, 2, 9, 1), (11, 1, 13, 1, 4, 2)]:
        assert p[s, nt * 26 + t, g * 16 + r, e] == wn[32 * s + 16 * nt + r, 16 * t + 4 * g + e]
    assert np.array_equal(p[26], p[0])
    w1 = torch.from_numpy(rng.normal(size=(H, H)).astype(np.float32)).cuda()
    q = pk.ops.pack_trunk_cols(w1).cpu().numpy().reshape(14, 52, 64, 4)
    w1n = w1.cpu().numpy()
    for (j, nt, tt, g, r, e) in [(0, 0, 0, 0, 0, 0), (12, 25, 1, 3, 15, 3), (5, 13, 0, 2, 7, 1)]:
        assert q[j, nt * 2 + tt, g * 16 + r, e] == w1n[16 * nt + r, 32 * j + 16 * tt + 4 * g + e]
    assert pk._lib.lib().occ4d_trunk_packed_floats(416) == p[:14].size and pk._lib.lib().occ4d_trunk_width() == H


def test_packing_layouts_half_cu():
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.normal(size=(832, H)).astype(np.float32)).cuda()
    p = pk.ops.pack_trunk4_rows(w).cpu().numpy().reshape(53, 26, 64, 4)
    wn = w.cpu().numpy()
    for (s, t, g, r, e) in [(0, 0, 0, 0, 0), (7, 25, 3, 15, 3), (51, 7, 2, 9, 1), (23, 13, 1, 4, 2)]:
        assert p[s, t, g * 16 + r, e] == wn[16 * s + r, 16 * t + 4 * g + e]
    assert np.array_equal(p[52], p[0])
    w1 = torch.from_numpy(rng.normal(size=(H, H)).astype(np.float32)).cuda()
    q = pk.ops.pack_trunk4_cols(w1).cpu().numpy().reshape(27, 26, 64, 4)
    w1n = w1.cpu().numpy()
    for (j, nt, g, r, e) in [(0, 0, 0, 0, 0), (25, 25, 3, 15, 3), (11, 13, 2, 7, 1)]:
        assert q[j, nt, g * 16 + r, e] == w1n[16 * nt + r, 16 * j + 4 * g + e]
    assert pk._lib.lib().occ4d_trunk4_packed_floats(416) == q.size


GENERATIONS = {'half_cu': ('pack_trunk4_rows', 'pack_trunk4_cols'), 'full_cu': ('pack_trunk_rows', 'pack_trunk_cols')}


@pytest.fixture(params=sorted(GENERATIONS))
def packers(request):
    """(rows packer, cols packer) of csrc/trunk4.hip (4-wave workgroups, two per CU) and of csrc/trunk.hip"""
    rows, cols = GENERATIONS[request.param]
    return getattr(pk.ops, rows), getattr(pk.ops, cols)


@pytest.mark.parametrize('n', [1, 15, 16, 17, 63, 64, 65, 129, 1000, 4099])
@pytest.mark.parametrize('with_interp', [False, True])
def test_resblock_matches_fp64(n, with_interp, packers):
    pack_rows, pack_cols = packers
    rng = np.random.default_rng(n + 7 * with_interp)
    x = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    w0, b0 = _weights(rng, H)
    w1, b1 = _weights(rng, H)
    interp, extra = None, 0.0
    if with_interp:
        ztab, zconst, idx, w = _interp(rng, n)
        interp = (zconst[3 * H:4 * H], ztab[:, 3 * H:4 * H], idx, w)
        extra = _interp_ref(ztab, zconst, idx, w, 3)
    xd = x.double()
    hd = torch.relu(xd) @ w0.double().T + b0.double()
    want = xd + torch.relu(hd) @ w1.double().T + b1.double() + extra
    got = pk.ops.resblock(x, pack_rows(w0), b0, pack_cols(w1), b1, interp=interp)
    assert float((got.double() - want).abs().max()) < 2e-5
    # in place, and identical to the out-of-place result
    x2 = x.clone()
    pk.ops.resblock(x2, pack_rows(w0), b0, pack_cols(w1), b1, out=x2, interp=interp)
    assert torch.equal(x2, got)
    # the generic chain it replaces agrees to fp32 rounding
    h = pk.ops.linear(x, w0, b0, relu_in=True)
    ref = pk.ops.linear(h, w1, b1, relu_in=True, residual=x)
    if with_interp:
        pk.ops.interp_add(ref, interp[0], interp[1], idx, w)
    assert float((got - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('n', [1, 16, 250, 3000])
@pytest.mark.parametrize('cfg', [dict(n_out=832), dict(n_out=416, relu_in=True), dict(n_out=416, residual=True),
                                 dict(n_out=416, residual=True, interp=True), dict(n_out=32)])
def test_rowlin_matches_fp64(n, cfg, packers):
    pack_rows = packers[0]
    rng = np.random.default_rng(n + cfg['n_out'])
    x = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    w, b = _weights(rng, cfg['n_out'])
    xd = torch.relu(x.double()) if cfg.get('relu_in') else x.double()
    want = xd @ w.double().T + b.double()
    res, interp = None, None
    if cfg.get('residual'):
        res = torch.from_numpy(rng.normal(size=(n, cfg['n_out'])).astype(np.float32)).cuda()
        want = want + res.double()
    if cfg.get('interp'):
        ztab, zconst, idx, wi = _interp(rng, n)
        interp = (zconst[H:2 * H], ztab[:, H:2 * H], idx, wi)
        want = want + _interp_ref(ztab, zconst, idx, wi, 1)
    out = res.clone() if res is not None else None           # res aliases out (x += layer3(agg) in place)
    got = pk.ops.rowlin(x, pack_rows(w), b, cfg['n_out'], relu_in=bool(cfg.get('relu_in')),
                        residual=out, out=out, interp=interp)
    assert float((got.double() - want).abs().max()) < 2e-5


def test_trunk_kernels_reject_bad_arguments():
    x = torch.zeros((8, 300), device='cuda')
    b = torch.zeros((H,), device='cuda')
    with pytest.raises(AssertionError):
        pk.ops.resblock(x, None, b, None, b)                      # wrong width
    with pytest.raises(AssertionError):
        pk.ops.resblock(torch.zeros((8, H), device='cuda'), None, b, None, b)   # null weights -> OCC4D_EINVAL
    with pytest.raises(AssertionError):
        pk.ops.pack_trunk_rows(torch.zeros((40, H), device='cuda'))
    with pytest.raises(RuntimeError):
        pk.ops.pack_trunk_rows(torch.zeros((32, H)))           # CPU tensor: no fallback
    rc = pk._lib.lib().occ4d_rowlin_f32(None, 416, None, 416, None, None, 416, 0, None, 0, None, None, 0, None, None, 0, 8, None)   # nulls
    assert rc == pk._lib.EINVAL


def test_empty_batch():
    rng = np.random.default_rng(1)
    w0, b0 = _weights(rng, H)
    x = torch.zeros((0, H), device='cuda')
    assert pk.ops.resblock(x, pk.ops.pack_trunk_rows(w0), b0, pk.ops.pack_trunk_cols(w0), b0).shape == (0, H)
    assert pk.ops.rowlin(x, pk.ops.pack_trunk_rows(w0), b0, H).shape == (0, H)
    assert pk.ops.resblock(x, pk.ops.pack_trunk4_rows(w0), b0, pk.ops.pack_trunk4_cols(w0), b0).shape == (0, H)
    assert pk.ops.rowlin(x, pk.ops.pack_trunk4_rows(w0), b0, H).shape == (0, H)
    rc = pk._lib.lib().occ4d_rowlin4_f32(None, 416, None, 416, None, None, 416, 0, None, 0, None, None, 0, None, None, 0, 8, None)
    assert rc == pk._lib.EINVAL


@pytest.mark.parametrize('half_cu', [False, True])
@pytest.mark.parametrize('n,n_out', [(1, 416), (130, 832), (3000, 416), (777, 32)])
def test_rowlin_output_mask(n, n_out, half_cu):
    """occ4d_rowlin_masked_f32 / occ4d_rowlin4_masked_f32: the data gradient of a relu_in Linear, dx = (x > 0) . (g W),
    with the mask in the epilogue -- equal to the unmasked kernel followed by the masking pass, bit for bit."""
    rng = np.random.default_rng(n + n_out)
    g = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    w, b = _weights(rng, n_out)
    mask = torch.from_numpy(rng.normal(size=(n, n_out)).astype(np.float32)).cuda()
    mask[0, :5] = 0.0                                           # (zero counts as "not positive")
    p = pk.ops.pack_trunk4_rows(w) if half_cu else pk.ops.pack_trunk_rows(w)
    plain = pk.ops.rowlin(g, p, b, n_out)
    got = pk.ops.rowlin(g, p, b, n_out, mask=mask)
    assert torch.equal(got, torch.where(mask > 0, plain, torch.zeros_like(plain)))
    assert torch.equal(got, pk.ops.relu_mask(plain, mask))


@pytest.mark.parametrize('n,n_out,masked', [(512 * 64 + 37 * 64 + 5, 416, False), (2 * 512 * 64 + 64, 832, True),
                                            (512 * 64 + 128 * 64, 416, True), (512 * 64 + 129 * 64, 32, False),
                                            (512 * 64 + 3, 112, False)])
def test_half_cu_rowlin_tail_split(n, n_out, masked):
    """occ4d_rowlin4_f32 / _masked with row tiles behind the last full dispatch round (512 workgroups on a 256-CU part):
    up to a quarter round of them is split over four workgroups by output stage range -- same results as the 8-wave
    kernel row for row (both sum k in the same order), also at the split threshold, one tile past it, and with fewer
    than eight stages (no split)."""
    rng = np.random.default_rng(n_out)
    x = torch.from_numpy(rng.normal(size=(n, H)).astype(np.float32)).cuda()
    w, b = _weights(rng, n_out)
    res = torch.from_numpy(rng.normal(size=(n, n_out)).astype(np.float32)).cuda()
    mask = torch.from_numpy(rng.normal(size=(n, n_out)).astype(np.float32)).cuda() if masked else None
    got = pk.ops.rowlin(x, pk.ops.pack_trunk4_rows(w), b, n_out, relu_in=True, residual=res, mask=mask)
    if n_out % 32 == 0:
        ref = pk.ops.rowlin(x, pk.ops.pack_trunk_rows(w), b, n_out, relu_in=True, residual=res, mask=mask)
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    rows = torch.from_numpy(rng.integers(0, n, size=4000)).cuda()
    rows[:64] = torch.arange(n - 64, n)                    # the very last rows
    want = torch.relu(x[rows]).double() @ w.double().t() + b.double() + res[rows].double()
    if masked:
        want = torch.where(mask[rows] > 0, want, torch.zeros_like(want))
    assert float((got[rows].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())

