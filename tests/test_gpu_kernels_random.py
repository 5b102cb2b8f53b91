"""Seeded random sweeps over the kernel argument space (shapes that are not multiples of any tile, strided views,
every epilogue flag, all k, both metrics, every threads-per-query path): each kernel of the C ABI against a plain
numpy / PyTorch-CPU statement of the same op.  Bit-exact for index / selection kernels, 2e-5 relative to the
magnitude of the result for fp32 GEMM-like kernels (different summation order only)."""
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


def C(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_linear_random_shapes_and_flags(pk):
    rng = np.random.default_rng(2024)
    for trial in range(48):
        M = int(rng.choice([1, 2, 31, 127, 128, 129, 257, 1000, 4097]))
        K = int(rng.choice([3, 8, 32, 36, 68, 72, 100, 144, 288, 416, 832]))
        N = int(rng.choice([1, 5, 18, 32, 36, 72, 128, 144, 288, 416, 832]))
        x = rng.normal(size=(M, K)).astype(np.float32)
        w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
        b = rng.normal(size=(N,)).astype(np.float32) if rng.random() < 0.7 else None
        relu_in, relu_out = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        res = rng.normal(size=(M, N)).astype(np.float32) if (rng.random() < 0.4 and not relu_out) else None
        add_div = int(rng.choice([1, 3, 14]))
        add = rng.normal(size=(-(-M // add_div), N)).astype(np.float32) if rng.random() < 0.3 else None
        sub = sub_idx = None
        if rng.random() < 0.3:
            sub = rng.normal(size=(17, N)).astype(np.float32)
            sub_idx = rng.integers(0, 17, size=M).astype(np.int32)
        xin = np.maximum(x, 0) if relu_in else x
        ref = xin.astype(np.float64) @ w.T.astype(np.float64)
        if b is not None:
            ref = ref + b
        if add is not None:
            ref = ref + add[np.arange(M) // add_div]
        if sub is not None:
            ref = ref - sub[sub_idx]
        if relu_out:
            ref = np.maximum(ref, 0)
        if res is not None:
            ref = ref + res
        xd = C(x)
        if trial % 3 == 0 and K % 4 == 0:          # strided input view (row stride > K)
            wide = torch.zeros((M, K + 8), device='cuda')
            wide[:, :K] = xd
            xd = wide[:, :K]
        got = pk.ops.linear(xd, C(w), C(b) if b is not None else None, relu_in=relu_in, relu_out=relu_out,
                            residual=C(res) if res is not None else None,
                            add_rows=C(add) if add is not None else None, add_div=add_div,
                            sub_rows=C(sub) if sub is not None else None,
                            sub_idx=C(sub_idx) if sub_idx is not None else None)
        err = np.abs(got.cpu().numpy() - ref).max()
        assert err <= 2e-5 * max(1.0, np.abs(ref).max()), (trial, M, K, N, err)


def _brute_knn(q, d, k, metric):
    dx = q[:, None, 0] - d[None, :, 0]
    dy = q[:, None, 1] - d[None, :, 1]
    dz = q[:, None, 2] - d[None, :, 2]
    if metric == 0:
        dist = (dx * dx + dy * dy) + dz * dz
    else:
        dist = None                                # fused-multiply-add chain: compared through indices of metric 0
    order = np.argsort(dist, axis=1, kind='stable')[:, :k]
    return order, np.take_along_axis(dist, order, axis=1)


def test_knn_random_shapes_all_paths(pk):
    """metric 0 against a stable numpy argsort of the same fp32 arithmetic (bit-exact indices and distances), for
    every k, query counts on both sides of the threads-per-query thresholds, strided inputs, duplicate points."""
    rng = np.random.default_rng(7)
    for trial in range(30):
        nq = int(rng.choice([1, 5, 64, 255, 1000, 3000, 25000]))
        nd = int(rng.choice([16, 100, 255, 256, 531, 1025, 4000]))
        k = int(rng.integers(1, min(16, nd) + 1))
        q = rng.uniform(-5, 5, size=(nq, 4)).astype(np.float32)
        d = rng.uniform(-5, 5, size=(nd, 5)).astype(np.float32)
        if trial % 4 == 0 and nd > 40:
            d[20:30, :3] = d[5:15, :3]            # exact duplicates: lowest index first
        idx, dist = pk.ops.knn(C(q)[:, :3], C(d)[:, :3], k, metric=0, return_dist=True)
        ref_i, ref_d = _brute_knn(q, d, k, 0)
        assert np.array_equal(idx.cpu().numpy(), ref_i), (trial, nq, nd, k)
        assert np.array_equal(dist.cpu().numpy(), ref_d)
        # metric 1 orders like metric 0 except where the square roots collide: distances must be non-decreasing,
        # equal to sqrt of the fma chain of the returned index, and the index set equal wherever no two sqrt tie
        idx1, dist1 = pk.ops.knn(C(q)[:, :3], C(d)[:, :3], k, metric=1, return_dist=True)
        d1 = dist1.cpu().numpy()
        assert np.all(np.diff(d1, axis=1) >= 0)
        sel = d[idx1.cpu().numpy().astype(np.int64), :3]
        diff = (q[:, None, :3] - sel).astype(np.float64)
        assert np.abs(np.sqrt((diff ** 2).sum(-1)) - d1).max() <= 1e-5
        assert np.allclose(d1, np.sqrt(ref_d), rtol=1e-6, atol=0)


def test_fps_single_vs_cooperative_random(pk):
    rng = np.random.default_rng(11)
    for trial in range(10):
        n = int(rng.choice([3, 64, 777, 2049, 9000, 20000, 32768]))
        m = int(rng.integers(1, n + 1)) if n < 3000 else int(rng.integers(n // 8, n // 3))
        p = C(rng.uniform(-5, 5, size=(n, 3)).astype(np.float32))
        a, ao = pk.ops.fps(p, m, return_order=True)
        wgs = int(rng.choice([0, 1, 2, 5, 16]))
        if -(-n // max(1, wgs or 1)) > 16 * 512:
            wgs = 0
        b, bo = pk.ops.fps_coop(p, m, start=0, n_workgroups=wgs, return_order=True)
        assert torch.equal(a, b) and torch.equal(ao, bo), (trial, n, m, wgs)


def test_compaction_and_gather_random(pk):
    rng = np.random.default_rng(13)
    for trial in range(20):
        n = int(rng.choice([1, 2, 255, 256, 257, 1000, 70000, 300000]))
        d = int(rng.choice([1, 3, 4, 7, 9, 11]))
        rows = rng.normal(size=(n, d)).astype(np.float32)
        key = rng.uniform(0, 1, size=n).astype(np.float32)
        thr = float(rng.choice([-1.0, 0.0, 0.3, 0.5, 0.99, 2.0]))
        if n > 10:
            key[3] = thr
        strict = bool(rng.random() < 0.5)
        kept, kk = pk.ops.compact_rows(C(rows), C(key), thr, strict=strict)
        mask = key > np.float32(thr) if strict else key >= np.float32(thr)
        assert np.array_equal(kept.cpu().numpy(), rows[mask]) and np.array_equal(kk.cpu().numpy(), key[mask])
        idx = rng.integers(0, n, size=int(rng.integers(1, 2000))).astype(np.int32)
        assert np.array_equal(pk.ops.gather_rows(C(rows), C(idx)).cpu().numpy(), rows[idx])


def test_pool_norm_interp_random(pk):
    rng = np.random.default_rng(17)
    for trial in range(12):
        n = int(rng.choice([1, 33, 500, 4779]))
        m = int(rng.choice([1, 40, 531]))
        d = int(rng.choice([4, 36, 72, 144, 288, 416]))
        k = int(rng.integers(1, 13))
        y = rng.normal(size=(n, d)).astype(np.float32)
        idx = rng.integers(0, n, size=(m, k)).astype(np.int32)
        got = pk.ops.maxpool_gather(C(y), C(idx)).cpu().numpy()
        assert np.array_equal(got, y[idx].max(axis=1))
        g, b = rng.normal(size=d).astype(np.float32), rng.normal(size=d).astype(np.float32)
        ln = pk.ops.layernorm(C(y), C(g), C(b), eps=1e-5, relu=bool(trial % 2)).cpu().numpy()
        ref = torch.nn.functional.layer_norm(torch.from_numpy(y), (d,), torch.from_numpy(g), torch.from_numpy(b), 1e-5)
        ref = torch.relu(ref) if trial % 2 else ref
        assert np.abs(ln - ref.numpy()).max() <= 2e-5
        assert np.abs(pk.ops.mean_rows(C(y)).cpu().numpy() - y.astype(np.float64).mean(axis=0)).max() <= 1e-5


# ------------------------------------------------------------------ backward kernels vs torch autograd (CPU, fp64)
def _rel(got, ref):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else got
    ref = ref.detach().double().numpy() if isinstance(ref, torch.Tensor) else ref
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())


def test_weight_gradient_random(pk):
    rng = np.random.default_rng(31)
    for trial in range(24):
        M = int(rng.choice([1, 15, 16, 17, 255, 1000, 5003, 40000]))
        N = int(rng.choice([4, 5, 36, 72, 128, 130, 416, 832]))
        K = int(rng.choice([3, 4, 32, 36, 68, 100, 288, 416, 832]))
        g = rng.normal(size=(M, N)).astype(np.float32)
        x = rng.normal(size=(M, K)).astype(np.float32)
        relu_x, bias = bool(rng.random() < 0.5), bool(rng.random() < 0.6)
        res = pk.ops.linear_wgrad(C(g), C(x), bias=bias, relu_x=relu_x)
        dw, db = res if bias else (res, None)
        xr = np.maximum(x, 0) if relu_x else x
        ref = g.astype(np.float64).T @ xr.astype(np.float64)
        assert _rel(dw, ref) <= 3e-6, (trial, M, N, K)
        if bias:
            assert _rel(db, g.astype(np.float64).sum(axis=0)) <= 3e-6
        assert _rel(pk.ops.colsum(C(g)), g.astype(np.float64).sum(axis=0)) <= 3e-6
        ref_mask = np.where(x[:, :min(N, K)] > 0, g[:, :min(N, K)], 0)
        assert np.array_equal(pk.ops.relu_mask(C(g[:, :min(N, K)]), C(x[:, :min(N, K)])).cpu().numpy(), ref_mask)


@pytest.mark.parametrize('M,N,K', [(4096, 416, 64), (5003, 416, 832), (40001, 832, 128), (70000, 416, 832), (4097, 832, 64)])
def test_weight_gradient_training_shapes(pk, M, N, K):
    """The weight-gradient GEMM at the decoder's widths with many rows (the shapes that carry a training step's FLOPs),
    ragged M, bias gradient, relu on the x operand and accumulation into an existing gradient, against fp64."""
    rng = np.random.default_rng(M + N + K)
    g = rng.normal(size=(M, N)).astype(np.float32)
    x = rng.normal(size=(M, K)).astype(np.float32)
    for relu_x, bias in ((False, True), (True, False)):
        res = pk.ops.linear_wgrad(C(g), C(x), bias=bias, relu_x=relu_x)
        dw, db = res if bias else (res, None)
        xr = np.maximum(x, 0) if relu_x else x
        assert _rel(dw, g.astype(np.float64).T @ xr.astype(np.float64)) <= 3e-6
        if bias:
            assert _rel(db, g.astype(np.float64).sum(axis=0)) <= 3e-6
    acc = torch.ones((N, K), device='cuda')
    pk.ops.linear_wgrad(C(g), C(x), out=acc, accumulate=True)
    assert _rel(acc, 1.0 + g.astype(np.float64).T @ x.astype(np.float64)) <= 3e-6


def test_scatter_and_pool_gradients_random(pk):
    rng = np.random.default_rng(37)
    for trial in range(12):
        n, m, d, k = int(rng.choice([1, 40, 900])), int(rng.choice([1, 17, 531])), int(rng.choice([4, 36, 288, 416])), \
            int(rng.integers(1, 15))
        src = rng.normal(size=(n * k, d)).astype(np.float32)
        idx = rng.integers(0, m, size=(n, k)).astype(np.int32)
        got = pk.ops.scatter_add_rows(C(src), C(idx), m, scale=-1.0)
        ref = np.zeros((m, d))
        np.add.at(ref, idx.reshape(-1), -src.astype(np.float64))
        assert _rel(got, ref) <= 1e-5
        assert _rel(pk.ops.segment_sum(C(src), k), src.astype(np.float64).reshape(n, k, d).sum(axis=1)) <= 3e-6
        # max-pool backward: gradient goes to the FIRST maximal neighbour
        y = rng.normal(size=(m, d)).astype(np.float32)
        dz = rng.normal(size=(n, d)).astype(np.float32)
        yt = torch.from_numpy(y).double().requires_grad_(True)
        gathered = yt[torch.from_numpy(idx).long()]                       # (n,k,d)
        first = gathered.detach().numpy().argmax(axis=1)                   # first maximum
        ref = np.zeros((m, d))
        np.add.at(ref, (idx[np.arange(n)[:, None], first], np.arange(d)[None, :]), dz.astype(np.float64))
        assert _rel(pk.ops.maxpool_gather_bwd(C(y), C(idx), C(dz)), ref) <= 1e-5
        # interpolation backward
        w = rng.uniform(size=(n, k)).astype(np.float32)
        dy = rng.normal(size=(n, d)).astype(np.float32)
        ref = np.zeros((m, d))
        np.add.at(ref, idx.reshape(-1), (w[:, :, None] * dy[:, None, :]).reshape(n * k, d).astype(np.float64))
        assert _rel(pk.ops.interp_bwd(C(dy), C(idx), C(w), m), ref) <= 1e-5


def test_attention_chain_gradients_random(pk):
    """softmax-aggregate, position-hidden and LayerNorm backward against torch autograd of the same formulas."""
    rng = np.random.default_rng(41)
    for trial in range(8):
        n, m, d, k = int(rng.choice([1, 33, 300])), int(rng.choice([14, 76, 531])), int(rng.choice([36, 72, 288, 416])), \
            int(rng.integers(1, 15))
        idx = rng.integers(0, m, size=(n, k)).astype(np.int32)
        logits = rng.normal(size=(n * k, d)).astype(np.float32)
        v = rng.normal(size=(m, d)).astype(np.float32)
        pe = rng.normal(size=(n * k, d)).astype(np.float32)
        dagg = rng.normal(size=(n, d)).astype(np.float32)
        lt, vt, pt = (torch.from_numpy(a).double().requires_grad_(True) for a in (logits, v, pe))
        att = torch.softmax(lt.view(n, k, d) / float(np.float32(np.sqrt(d))), dim=1)
        agg = (att * (vt[torch.from_numpy(idx).long()] + pt.view(n, k, d))).sum(dim=1)
        agg.backward(torch.from_numpy(dagg).double())
        fwd = pk.ops.pt_softmax_agg(C(logits), C(v), C(pe), C(idx))
        assert _rel(fwd, agg) <= 3e-6
        dl, dpe, dv = pk.ops.pt_softmax_agg_bwd(C(logits), C(v), C(pe), C(idx), C(dagg))
        assert _rel(dl, lt.grad) <= 1e-5 and _rel(dpe, pt.grad) <= 1e-5 and _rel(dv, vt.grad) <= 1e-5
        # position hidden
        h = 32
        pos, pos2 = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32), rng.uniform(-5, 5, size=(m, 3)).astype(np.float32)
        P1, c1 = rng.normal(size=(h, 3)).astype(np.float32), rng.normal(size=(h,)).astype(np.float32)
        gr = rng.normal(size=(n * k, h)).astype(np.float32)
        Pt, ct = torch.from_numpy(P1).double().requires_grad_(True), torch.from_numpy(c1).double().requires_grad_(True)
        rel = (torch.from_numpy(pos).double()[:, None, :] - torch.from_numpy(pos2).double()[torch.from_numpy(idx).long()])
        r = torch.relu(rel.view(n * k, 3) @ Pt.t() + ct)
        r.backward(torch.from_numpy(gr).double())
        r_dev = pk.ops.pt_pos_hidden(C(pos), C(pos2), C(idx), C(P1), C(c1))
        assert _rel(r_dev, r) <= 3e-6
        dP1, dc1 = pk.ops.pt_pos_hidden_bwd(C(pos), C(pos2), C(idx), r_dev, C(gr))
        assert _rel(dP1, Pt.grad) <= 2e-5 and _rel(dc1, ct.grad) <= 2e-5
        # LayerNorm backward
        x = rng.normal(size=(n * k, d)).astype(np.float32)
        gam = rng.normal(size=d).astype(np.float32)
        go = rng.normal(size=(n * k, d)).astype(np.float32)
        xt, gt = torch.from_numpy(x).double().requires_grad_(True), torch.from_numpy(gam).double().requires_grad_(True)
        bt = torch.zeros(d, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.layer_norm(xt, (d,), gt, bt, 1e-5).backward(torch.from_numpy(go).double())
        dx, dg, dbeta = pk.ops.layernorm_bwd(C(x), C(gam), C(go), 1e-5)
        assert _rel(dx, xt.grad) <= 2e-5 and _rel(dg, gt.grad) <= 2e-5 and _rel(dbeta, bt.grad) <= 2e-5


@pytest.mark.parametrize('k', [8, 12, 14, 16, 13])
@pytest.mark.parametrize('d,with_pe,scale', [(416, True, 1.0), (36, True, 1.0), (288, False, 1.0), (416, True, 300.0), (38, True, 1.0)])
def test_softmax_aggregate_backward_wide_and_scalar_kernels(pk, k, d, with_pe, scale):
    """The 16-byte-lane kernel (compile-time k in {8, 12, 14, 16}, d % 4 == 0: pair tensors as dwordx4, value gradients
    reduced by a separate segment sum) and the scalar kernel (any k, d) against torch autograd in fp64: with and without
    positional encodings, logits far outside the exp range (one neighbour takes all the weight, the others underflow), a
    width that is not a multiple of 4, query counts that do not fill a workgroup."""
    rng = np.random.default_rng(1000 * k + d)
    for n, m in ((1, 14), (257, 76), (3000, 531)):
        idx = rng.integers(0, m, size=(n, k)).astype(np.int32)
        logits = (scale * rng.normal(size=(n * k, d))).astype(np.float32)
        v = rng.normal(size=(m, d)).astype(np.float32)
        pe = rng.normal(size=(n * k, d)).astype(np.float32)
        dagg = rng.normal(size=(n, d)).astype(np.float32)
        lt, vt, pt = (torch.from_numpy(a).double().requires_grad_(True) for a in (logits, v, pe))
        att = torch.softmax(lt.view(n, k, d) / float(np.float32(np.sqrt(d))), dim=1)
        val = vt[torch.from_numpy(idx).long()] + (pt.view(n, k, d) if with_pe else 0.0)
        (att * val).sum(dim=1).backward(torch.from_numpy(dagg).double())
        dl, dpe, dv = pk.ops.pt_softmax_agg_bwd(C(logits), C(v), C(pe) if with_pe else None, C(idx), C(dagg))
        assert _rel(dl, lt.grad) <= 1e-5 and _rel(dv, vt.grad) <= 1e-5
        if with_pe:
            assert _rel(dpe, pt.grad) <= 1e-5
        else:
            assert dpe is None


def test_short_vector_reduce_of_the_bias_gradient(pk):
    """The bias gradient of a wide layer = sum of up to 113 per-slice partials, by the 4-wave short-vector reduce:
    against fp64 column sums at the training shapes (N = 416 / 832, many slices) and for a ragged N."""
    rng = np.random.default_rng(5)
    for (M, N, K) in ((68812, 416, 416), (40000, 832, 416), (4100, 420, 64), (300, 36, 36)):
        g = rng.normal(size=(M, N)).astype(np.float32)
        x = rng.normal(size=(M, K)).astype(np.float32)
        dw, db = pk.ops.linear_wgrad(C(g), C(x), bias=True)
        assert _rel(db, g.astype(np.float64).sum(axis=0)) <= 3e-6
        assert _rel(dw, g.astype(np.float64).T @ x.astype(np.float64)) <= 3e-6
        assert _rel(pk.ops.colsum(C(g)), g.astype(np.float64).sum(axis=0)) <= 3e-6


@pytest.mark.parametrize('n,k,m,h', [(5, 3, 9, 32), (3000, 14, 700, 32), (9000, 16, 300, 24), (700, 7, 50, 64), (2000, 14, 90, 48),
                                     (40000, 14, 1062, 32)])
def test_pos_hidden_backward_sizes_and_widths(pk, n, k, m, h):
    """occ4d_pt_pos_hidden_bwd_f32 (64 / h pairs per wave slice, eight pairs in flight per lane, block-level LDS reduction):
    hidden widths that do and do not divide 64, pair counts below one block, beyond the grid and with ragged tails."""
    rng = np.random.default_rng(n + h)
    pos, pos2 = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32), rng.uniform(-5, 5, size=(m, 3)).astype(np.float32)
    idx = rng.integers(0, m, size=(n, k)).astype(np.int32)
    P1, c1 = rng.normal(size=(h, 3)).astype(np.float32), rng.normal(size=(h,)).astype(np.float32)
    gr = rng.normal(size=(n * k, h)).astype(np.float32)
    r_dev = pk.ops.pt_pos_hidden(C(pos), C(pos2), C(idx), C(P1), C(c1))
    dP1, dc1 = pk.ops.pt_pos_hidden_bwd(C(pos), C(pos2), C(idx), r_dev, C(gr))
    rel = (torch.from_numpy(pos).double()[:, None, :] - torch.from_numpy(pos2).double()[torch.from_numpy(idx).long()]).view(n * k, 3)
    gm = torch.from_numpy(gr).double() * (r_dev.cpu().double() > 0)
    assert _rel(dP1, gm.t() @ rel) <= 2e-5 and _rel(dc1, gm.sum(0)) <= 2e-5


@pytest.mark.parametrize('n,d', [(3, 36), (20000, 416), (9000, 72), (300, 600), (70000, 144)])
def test_layernorm_backward_sizes(pk, n, d):
    """occ4d_layernorm_bwd_f32: rows beyond the grid (grid-stride loop with dgamma / dbeta in registers) and a width above
    the register path (per-row atomics)."""
    rng = np.random.default_rng(n + d)
    x = rng.normal(size=(n, d)).astype(np.float32)
    gam = rng.normal(size=d).astype(np.float32)
    go = rng.normal(size=(n, d)).astype(np.float32)
    xt, gt = torch.from_numpy(x).double().requires_grad_(True), torch.from_numpy(gam).double().requires_grad_(True)
    bt = torch.zeros(d, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.layer_norm(xt, (d,), gt, bt, 1e-5).backward(torch.from_numpy(go).double())
    dx, dg, dbeta = pk.ops.layernorm_bwd(C(x), C(gam), C(go), 1e-5)
    assert _rel(dx, xt.grad) <= 2e-5 and _rel(dg, gt.grad) <= 3e-5 and _rel(dbeta, bt.grad) <= 3e-5


def test_matmul_f64_and_its_gradients(pk):
    """occ4d_matmul_f64 (the merged-weight products, fp64): contiguous, transposed-view and vector operands against
    torch on the CPU, and the autograd Function's gradients."""
    rng = np.random.default_rng(5)
    for (m, k, n) in [(832, 416, 416), (33, 17, 5), (416, 32, 1), (1, 1, 1)]:
        a = torch.from_numpy(rng.normal(size=(m, k))).cuda()
        b = torch.from_numpy(rng.normal(size=(k, n))).cuda()
        want = a.cpu() @ b.cpu()
        assert float((pk.ops.matmul_f64(a, b).cpu() - want).abs().max()) < 1e-12 * k
        at = torch.from_numpy(rng.normal(size=(k, m))).cuda()
        assert float((pk.ops.matmul_f64(at.t(), b).cpu() - at.cpu().t() @ b.cpu()).abs().max()) < 1e-12 * k
    a = torch.from_numpy(rng.normal(size=(40, 24))).cuda().requires_grad_(True)
    v = torch.from_numpy(rng.normal(size=(24,))).cuda().requires_grad_(True)
    b = torch.from_numpy(rng.normal(size=(24, 9))).cuda().requires_grad_(True)
    go, gv = torch.from_numpy(rng.normal(size=(40, 9))).cuda(), torch.from_numpy(rng.normal(size=(40,))).cuda()
    ((pk.autograd.matmul64(a, b) * go).sum() + (pk.autograd.matmul64(a, v) * gv).sum()).backward()
    ar, vr, br = (t.detach().cpu().requires_grad_(True) for t in (a, v, b))
    (((ar @ br) * go.cpu()).sum() + ((ar @ vr) * gv.cpu()).sum()).backward()
    for got, ref in ((a.grad, ar.grad), (v.grad, vr.grad), (b.grad, br.grad)):
        assert float((got.cpu() - ref).abs().max()) < 1e-11
    with pytest.raises(AssertionError):
        pk.ops.matmul_f64(torch.zeros((2, 2), dtype=torch.float64), torch.zeros((2, 2), dtype=torch.float64))   # CPU tensors
