"""GPU parity: the HIP path (through the C ABI) against (a) the golden vectors the real
reference produced and (b) the CPU oracle on the same seeded inputs.  Bit-exact for
indices and kNN distances; 1e-4 absolute for floating-point outputs (BASELINE.json
north_star tolerance) -- most checks are far tighter."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-4
T = gc.as_tensor


@pytest.fixture(scope='module')
def pk():
    import occlusions4d_amd
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    occlusions4d_amd._lib.lib()   # fail loudly if the native library is missing
    return occlusions4d_amd


def dev(a):
    return T(a).cuda()


def close(a, b, tol=TOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol, 'max abs err %.3g > %.3g' % (err, tol)


# ------------------------------------------------------------------ kNN (E4, K1/K6/K8)
@pytest.mark.parametrize('case', gc.KNN_CASES, ids=lambda c: c['name'])
def test_knn_matches_reference_indices(pk, case):
    q, d = gc.knn_inputs(case)
    idx = pk.point_transformer_layer.kNN_torch(dev(q)[None], dev(d)[None], case['k'])[0]
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.cpu().numpy(), load_golden('g1_knn_' + case['name'])['idx'])


def test_knn_strided_xyz_and_large(pk):
    from oracle import path as op
    rng = np.random.default_rng(5)
    pcl = rng.uniform(-5, 5, size=(3000, 8)).astype(np.float32)
    idx = pk.point_transformer_layer.kNN_torch(dev(pcl)[None, :, :3], dev(pcl)[None, :, :3], 16)[0].cpu()
    ref = op.knn_indices(T(pcl)[None, :, :3], T(pcl)[None, :, :3], 16)[0]
    assert torch.equal(idx, ref)
    assert torch.equal(idx[:, 0], torch.arange(3000))      # a point is its own nearest neighbour


def test_knn_rejects_bad_k(pk):
    x = torch.zeros(1, 4, 3).cuda()
    with pytest.raises(AssertionError):
        pk.point_transformer_layer.kNN_torch(x, x, 17)
    with pytest.raises(AssertionError):
        pk.point_transformer_layer.kNN_torch(x, x, 5)       # n_data < k


@pytest.mark.parametrize('case', gc.MYKNN_CASES, ids=lambda c: c['name'])
def test_my_knn_torch_bit_exact(pk, case):
    q, key = gc.myknn_inputs(case)
    inds, knn, dists = pk.geometry.my_knn_torch(dev(q), dev(key), case['k'], return_inds=True, return_knn=True,
                                                return_dists=True)
    g = load_golden('g6_myknn_' + case['name'])
    assert np.array_equal(inds.cpu().numpy(), g['inds'])
    assert np.array_equal(dists.cpu().numpy(), g['dists'])          # sqrt(fma chain), bit for bit
    assert np.array_equal(knn.cpu().numpy(), key[g['inds']])


# ------------------------------------------------------------------ FPS (E7, K5)
@pytest.mark.parametrize('n', [1, 2, 63, 76, 531, 1024, 1593, 2048, 4779, 14336])
def test_fps_matches_restated_torch_cluster(pk, n):
    from oracle import cluster
    rng = np.random.default_rng(100 + n)
    p = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    m = int(np.ceil(n / 3))
    got, order = pk.ops.fps(dev(p), m, return_order=True)
    ref_order = cluster.fps(T(p), None, ratio=1.0 / 3, random_start=False)
    assert ref_order.numel() == m
    assert np.array_equal(order.cpu().numpy(), ref_order.numpy().astype(np.int32))
    assert np.array_equal(got.cpu().numpy(), np.sort(ref_order.numpy()).astype(np.int32))


# ------------------------------------------------------------------ Linear (K3/K11)
@pytest.mark.parametrize('shape', [(1, 8, 36), (37, 36, 36), (300, 68, 416), (129, 416, 416), (515, 832, 416),
                                   (64, 32, 832), (200, 416, 5), (200, 416, 18), (100, 288, 128), (77, 144, 288),
                                   (50, 72, 144)])
def test_linear_against_torch_fp32(pk, shape):
    M, K, N = shape
    rng = np.random.default_rng(M * 1000 + K + N)
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=(N,)).astype(np.float32)
    r = rng.normal(size=(M, N)).astype(np.float32)
    ref = torch.nn.functional.linear(torch.relu(T(x)).double(), T(w).double(), T(b).double())
    ref = torch.relu(ref) + T(r).double()
    got = pk.ops.linear(dev(x), dev(w), dev(b), relu_in=True, relu_out=True, residual=dev(r))
    close(got, ref.float(), 2e-5)
    # asymmetric check without epilogue (catches row/col transposition)
    close(pk.ops.linear(dev(x), dev(w)), (T(x).double() @ T(w).double().T).float(), 2e-5)


def test_linear_gather_epilogue(pk):
    rng = np.random.default_rng(7)
    M, K, N, k = 280, 32, 832, 14
    x = rng.normal(size=(M, K)).astype(np.float32)
    w = rng.normal(size=(N, K)).astype(np.float32)
    add = rng.normal(size=(M // k, N)).astype(np.float32)
    sub = rng.normal(size=(50, N)).astype(np.float32)
    si = rng.integers(0, 50, size=M).astype(np.int32)
    ref = T(x).double() @ T(w).double().T + T(add).double().repeat_interleave(k, 0) - T(sub).double()[si.astype(np.int64)]
    got = pk.ops.linear(dev(x), dev(w), add_rows=dev(add), add_div=k, sub_rows=dev(sub), sub_idx=dev(si))
    close(got, ref.float(), 2e-5)


# ------------------------------------------------------------------ PT layer / block (E2/E3)
@pytest.mark.parametrize('case', gc.PTL_CASES, ids=lambda c: c['name'])
def test_pt_layer(pk, case):
    x, pos, x2, pos2, sd = gc.ptl_inputs(case)
    layer = pk.point_transformer_layer.PointTransformerLayer(case['dim'], num_neighbors=case['k'],
                                                             dim2=case.get('dim2')).cuda()
    layer.load_state_dict(sd)
    args = (dev(x)[None], dev(pos)[None]) + ((dev(x2)[None], dev(pos2)[None]) if x2 is not None else ())
    with torch.no_grad():
        close(layer(*args)[0], load_golden('g2_ptl_' + case['name'])['agg'])


@pytest.mark.parametrize('case', gc.PTB_CASES, ids=lambda c: c['name'])
def test_pt_block(pk, case):
    x, pos, x2, pos2, sd = gc.ptb_inputs(case)
    blk = pk.modules.PointTransformerBlock(case['dim'], case['dim'], case['dim'], num_neighbors=case['k'],
                                           d_hidden_abstract=case.get('dim2')).cuda()
    blk.load_state_dict(sd)
    args = (dev(x)[None], dev(pos)[None]) + ((dev(x2)[None], dev(pos2)[None]) if x2 is not None else ())
    with torch.no_grad():
        z, p = blk(*args)
    close(z[0], load_golden('g3_ptb_' + case['name'])['z'])


@pytest.mark.parametrize('case', gc.DOWN_CASES + gc.DOWN_BATCHNORM_CASES, ids=lambda c: c['name'])
def test_down_transition(pk, case):
    x, pos, sd = gc.down_inputs(case)
    dt = pk.modules.DownTransition(case['d_in'], case['d_out'], factor=3, knn_k=case['k'], norm_type=case['norm'],
                                   fps_random_start=False).cuda().eval()
    dt.load_state_dict(sd)
    with torch.no_grad():
        z, p_sub = dt(dev(x)[None], dev(pos)[None])
    g = load_golden('g4_down_' + case['name'])
    assert np.array_equal(p_sub[0].cpu().numpy(), g['p_sub'])
    close(z[0], g['z'])


# ------------------------------------------------------------------ encoder (E1)
@pytest.mark.parametrize('case', gc.ENC_CASES, ids=lambda c: c['name'])
def test_encoder(pk, case):
    pcl, pa, sd = gc.enc_inputs(case)
    net = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out, xg, lc = net(pcl.cuda(), False)
    g = load_golden('g5_enc_' + case['name'])
    assert lc is None
    assert np.array_equal(out[0, :, :3].cpu().numpy(), g['pcl_out'][:, :3])     # FPS subset, bit exact
    close(out[0], g['pcl_out'])
    close(xg[0], g['x_global'])


# ------------------------------------------------------------------ decoder (D1-D7)
def test_posenc(pk):
    g = load_golden('g7_posenc')
    enc = pk.implicit.positional_encode(dev(g['points']), 0.1, 8)
    assert enc.shape == (512, 68)
    close(enc, g['enc'], 2e-6)
    assert np.array_equal(enc[:, :4].cpu().numpy(), g['points'])


@pytest.mark.parametrize('case', gc.DEC_CASES + gc.DEC_SWISH_CASES, ids=lambda c: c['name'])
def test_decoder(pk, case):
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out, pen = net(dev(q), dev(abstract), dev(fglob), None)
        out_b, pen_b = net(dev(q)[None], dev(abstract)[None], dev(fglob)[None], None)   # batched form
    g = load_golden('g8_dec_' + case['name'])
    close(out, g['output'])
    close(pen[:, ::8], g['penult'])
    assert out_b.shape == (1,) + tuple(out.shape) and torch.equal(out_b[0], out)


@pytest.mark.parametrize('variant', ['trunk4', 'generic_trunk', 'first_gen', 'unfused', 'fused_interp'])
@pytest.mark.parametrize('case', gc.DEC_CASES, ids=lambda c: c['name'])
def test_decoder_kernel_variants(pk, case, variant):
    """The opt-in / fallback kernel selections of the decoder (OCC4D_PATH_* flags of the library's path-level entry
    points) against the same golden vectors (G8): half-CU trunk kernels (csrc/trunk4.hip), the generic Linear kernels in
    place of the row-resident ones, the first-generation attention kernel (csrc/crossattn.hip) and the unfused attention
    chain.  The library's launch-event hook tells which kernels really ran."""
    selection = dict(trunk4=variant == 'trunk4', trunk_kernels=variant != 'generic_trunk', attn16=variant != 'first_gen',
                     fused_attention=variant != 'unfused', fused_interp=variant == 'fused_interp')
    try:
        q, abstract, fglob, ia, sd = gc.dec_inputs(case)
        net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
        net.load_state_dict(sd)
        counts = {}
        for fam in ('resblock', 'cross_attn'):
            timer = pk.ops.KernelTimer(lambda name, fam=fam, **shape: name == fam)
            pk.ops.set_kernel_timer(timer)
            with torch.no_grad(), pk.kernels(**selection):
                out, pen = net(dev(q), dev(abstract), dev(fglob), None)
            pk.ops.set_kernel_timer(None)
            counts[fam] = timer.summary().get(fam, dict(launches=0))['launches']
    finally:
        pk.ops.set_kernel_timer(None)
    g = load_golden('g8_dec_' + case['name'])
    close(out, g['output'])
    close(pen[:, ::8], g['penult'])
    assert counts['resblock'] == (0 if variant == 'generic_trunk' else ia['n_blocks'])      # fused residual blocks
    assert counts['cross_attn'] == (0 if variant == 'unfused' else ia['cross_attn_layers'])   # fused attention launches


def test_decoder_batch_split_invariance(pk):
    """A query's result must not depend on which mini-batch it travels in (beyond fp32 rounding)."""
    case = gc.DEC_CASES[1]
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    a, g = dev(abstract), dev(fglob)
    with torch.no_grad():
        full, _ = net(dev(q), a, g, None)
        parts = torch.cat([net(dev(q[lo:lo + 100]), a, g, None)[0] for lo in range(0, q.shape[0], 100)])
    assert (full - parts).abs().max() <= 1e-5     # equal up to the rounding of the in-workgroup reduction order


# ------------------------------------------------------------------ perform_inference (D8)
@pytest.mark.parametrize('case', gc.INFER_CASES, ids=lambda c: c['name'])
def test_perform_inference(pk, case):
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
    close(res['pcl_abstract'], g['pcl_abstract'])
    close(res['features_global'], g['features_global'])
    # Queries whose k-th / (k+1)-th abstract neighbours are equidistant (systematic for CARLA's
    # two-level abstract cloud) are implementation-defined in the reference (unstable sort);
    # everywhere else the reference's output is matched, and on ALL rows the product's
    # documented rule (lowest index first) is matched against the oracle run with that rule.
    from oracle import path as op
    amb = op.tie_ambiguous(T(res['points_query']), T(g['pcl_abstract']), ia['num_local_features'],
                           ia['cross_attn_neighbors']).numpy()
    if inf['data_kind'] == 'greater':
        assert not amb.any()
    assert (~amb).sum() > 0.5 * amb.size
    close(res['implicit_output'][~amb], g['implicit_output'][~amb])
    with op.stable_ties():
        ref = op.perform_inference(
            pcl.clone(), esd, pa, dsd, ia, inf['min_z'], inf['cube_bounds'], inf['color_mode'],
            case['time_idx'], num_sample=case['num_sample'], point_sample_mode='grid',
            batch_size=case['batch_size'], predict_segmentation=inf['predict_segmentation'],
            track_mode='none', semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'],
            cube_mode=4, compress_air=True)
    close(res['implicit_output'], ref['implicit_output'])
    dens = ref['implicit_output'][:, 0]
    slack = int((np.abs(dens - 0.5) < TOL).sum())
    assert abs(res['output_solid'].shape[0] - ref['output_solid'].shape[0]) <= slack
    assert res['output_solid'].shape[0] + res['output_air'].shape[0] == dens.shape[0]
    assert res['output_air'].shape[1] == g['air_head'].shape[1]
    assert all(v.dtype == np.float32 for k, v in res.items() if k not in ('output_air',))


CARLA_INFER = [c for c in gc.INFER_CASES + gc.INFER_PAD_CASES if c['kind'] == 'carla']


@pytest.mark.parametrize('case', CARLA_INFER, ids=lambda c: c['name'])
def test_perform_inference_carla_every_row_with_the_reference_lists(pk, case):
    """Round 5.  CARLA's two-level abstract cloud holds every coarse point twice; the reference's
    unstable sort decides which of two equidistant points sits at rank k, so 42 % of these queries used to be masked
    out of the comparison with the reference.  The fixture now carries the neighbour lists the reference run took;
    fed to the product (perform_inference(neighbour_lists=...) -> LocalPclResnetFC.forward(knn_local, knn_cross) ->
    occ4d_decoder_query_fwd_f32), EVERY row is compared with the reference's output at 1e-4: no mask."""
    pcl, pa, ia, inf, esd, dsd = gc.infer_inputs(case)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    dec.load_state_dict(dsd)
    g = load_golden('g10_infer_' + case['name'])
    res = pk.inference.perform_inference(
        pcl.clone(), None, None, [enc, dec], torch.device('cuda:0'), 'if', inf['min_z'], inf['cube_bounds'],
        inf['color_mode'], case['time_idx'], None, sample_implicit=True, num_sample=case['num_sample'],
        point_sample_mode='grid', batch_size=case['batch_size'],
        predict_segmentation=inf['predict_segmentation'], track_mode='none', semantic_classes=13,
        density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4, compress_air=True,
        neighbour_lists=(g['knn_local'], g['knn_cross']))
    close(res['pcl_abstract'], g['pcl_abstract'])
    close(res['implicit_output'], g['implicit_output'])               # 100 % of the rows
    dens = g['implicit_output'][:, 0]
    slack = int((np.abs(dens - 0.5) < TOL).sum())
    assert abs(res['output_solid'].shape[0] - int(g['n_solid'][0])) <= slack
    from oracle import path as op
    amb = op.tie_ambiguous(T(res['points_query']), T(g['pcl_abstract']), ia['num_local_features'],
                           ia['cross_attn_neighbors']).numpy()
    assert amb.mean() > 0.2                                            # the rows the mask used to hide are in there
    print('\n[g10 %s] %d rows, %.1f %% tie-ambiguous, max |hip - ref| %.3g' % (
        case['name'], amb.size, 100 * amb.mean(), np.abs(res['implicit_output'] - g['implicit_output']).max()))


@pytest.mark.parametrize('case', gc.DEC_TWOLEVEL_CASES, ids=lambda c: c['name'])
def test_decoder_two_level_cloud(pk, case):
    """G8t: decoder on an abstract cloud with the CARLA encoder's two-level structure (coincident coordinates), (a) with
    the reference's own lists: every row at 1e-4; (b) searched by the library: the documented lowest-index rule, equal
    to the oracle under that rule on every row and to the reference wherever the reference is defined; (c) the training
    (autograd) forward takes the same lists."""
    from oracle import path as op
    q, abstract, fglob, ia, sd = gc.dec_twolevel_inputs(case)
    g = load_golden('g8_dec_' + case['name'])
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    kl, kc = dev(g['knn_local']), dev(g['knn_cross'])
    with torch.no_grad():
        out, pen = net(dev(q), dev(abstract), dev(fglob), None, knn_local=kl, knn_cross=kc)
        out64, _ = net(dev(q), dev(abstract), dev(fglob), None, knn_local=kl.long(), knn_cross=kc.long()[None])
        free, _ = net(dev(q), dev(abstract), dev(fglob), None)
    close(out, g['output'])
    close(pen[:, ::8], g['penult'])
    assert torch.equal(out, out64)
    with op.stable_ties():
        ref, _ = op.decoder_forward(sd, ia, T(q), T(abstract), T(fglob))
    close(free, ref)
    amb = op.tie_ambiguous(T(q), T(abstract), ia['num_local_features'], ia['cross_attn_neighbors']).numpy()
    close(free[~amb], g['output'][~amb])
    d = pk.ops.knn_dists(dev(q), dev(abstract), kl, metric=1)
    assert np.array_equal(d.cpu().numpy(), g['knn_local_dists'])
    idx, dist = pk.ops.knn(dev(q), dev(abstract), 8, metric=1, return_dist=True)
    assert torch.equal(pk.ops.knn_dists(dev(q), dev(abstract), idx, metric=1), dist)
    fa = dev(abstract).clone().requires_grad_(True)
    out_t, _ = net(dev(q), fa, dev(fglob), None, knn_local=kl, knn_cross=kc)
    assert out_t.requires_grad
    close(out_t, g['output'])


# ------------------------------------------------------------------ fused vs unfused attention
@pytest.mark.parametrize('k,n,dim,dim2', [(14, 1003, 416, 288), (13, 100, 416, 288), (12, 9, 416, 288),
                                          (8, 37, 416, 288), (7, 500, 416, 288), (3, 64, 416, 288),
                                          (1, 10, 416, 288), (14, 531, 288, 288), (5, 1, 416, 288),
                                          (14, 8, 416, 288), (14, 10, 416, 288), (14, 17, 416, 288), (14, 18, 416, 288),
                                          (14, 19, 416, 288), (11, 2, 416, 288), (2, 4000, 416, 288)])
@pytest.mark.parametrize('generation', ['attn16p', 'first', 'bf16x6', 'f16x3'])
def test_fused_attention_matches_unfused_chain(pk, k, n, dim, dim2, generation):
    """The fused kernels (9 queries x 14 rows packed per workgroup, masked slots for k < 14, ragged
    tail for n % 9 != 0) against the unfused kernel chain on the same inputs: the paired-workgroup 16 x 16
    MFMA kernel (crossattn16p.hip, d = 416) and the first-generation 32 x 32 kernel (crossattn.hip, d = 288 / 416)."""
    if generation != 'first' and dim != 416:
        pytest.skip('crossattn16p.hip / crossattn_bf16x6.hip are built for d = 416')
    rng = np.random.default_rng(1000 * k + n)
    m = 76
    x = rng.normal(size=(n, dim)).astype(np.float32)
    pos = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    x2 = rng.normal(size=(m, dim2)).astype(np.float32)
    pos2 = rng.uniform(-5, 5, size=(m, 3)).astype(np.float32)
    ptl = pk.point_transformer_layer
    layer = ptl.PointTransformerLayer(dim, num_neighbors=k, dim2=dim2).cuda()
    sd = pk.configs.fill_state_dict(layer, 4242 + k)
    layer.load_state_dict(sd)
    args = (dev(x)[None], dev(pos)[None], dev(x2)[None], dev(pos2)[None])
    with torch.no_grad():
        with pk.kernels(attn16=generation != 'first',
                        logit_precision=generation if generation in ('bf16x6', 'f16x3') else 'f32'):
            fused = layer(*args)[0]
        with pk.kernels(fused_attention=False, logit_precision='f32'):
            chain = layer(*args)[0]
    assert torch.isfinite(fused).all()
    close(fused, chain, 2e-5)


@pytest.mark.parametrize('k,n,m', [(14, 1003, 76), (14, 9, 531), (14, 10, 20), (13, 100, 76), (8, 37, 76), (1, 10, 76),
                                   (14, 4000, 531), (5, 1, 14)])
def test_f16w_attention_kernel_matches_the_default_fp16_kernel(pk, k, n, m):
    """csrc/crossattn_f16w.hip (32 x 32 x 16 instructions, 9 queries per workgroup, the 9th spread over the four waves) is
    an A/B switch (OCC4D_F16W=1), so the path-level tests only reach it with that variable set: here its C entry point
    is called directly, on the contract of occ4d_pt_cross_attn_f16x3_prescaled_f32 (aq, kt * hidden scale), against
    that kernel: same scheme, same products, different accumulation order."""
    from occlusions4d_amd import ops
    L, d = ops._lib.lib(), 416
    rng = np.random.default_rng(31 * k + n)
    hs = float(L.occ4d_pt_cross_attn_f16x3_hidden_scale())
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()   # noqa: E731
    aq, kt, vt = dev(hs * rng.normal(size=(n, 2 * d))), dev(hs * rng.normal(size=(m, 2 * d))), dev(rng.normal(size=(m, d)))
    qpos, apos = dev(rng.uniform(-5, 5, size=(n, 3))), dev(rng.uniform(-5, 5, size=(m, 3)))
    idx = ops.knn(qpos, apos, k, metric=0)
    P1, c1 = dev(rng.normal(size=(32, 3))), dev(rng.normal(size=(32,)))
    wp, w2, p2 = dev(0.1 * rng.normal(size=(2 * d, 32))), dev(0.03 * rng.normal(size=(d, 2 * d))), dev(0.1 * rng.normal(size=(d, 32)))
    outs = []
    for size, pack, run in ((L.occ4d_pt_cross_attn_f16x3_stream_floats, L.occ4d_pack_attn_f16x3_stream_f32,
                             L.occ4d_pt_cross_attn_f16x3_prescaled_f32),
                            (L.occ4d_pt_cross_attn_f16w_stream_floats, L.occ4d_pack_attn_f16w_stream_f32,
                             L.occ4d_pt_cross_attn_f16w_f32)):
        ws = torch.empty((int(size()),), dtype=torch.float32, device='cuda')
        ops._lib.check(pack(ops._ptr(w2), ops._ptr(wp), ops._ptr(p2), ops._ptr(ws), ops._stream()))
        out = torch.full((n, d), float('nan'), device='cuda')
        ops._lib.check(run(ops._ptr(aq), 2 * d, ops._ptr(qpos), 3, ops._ptr(apos), 3, ops._ptr(idx), ops._ptr(kt), 2 * d,
                           ops._ptr(vt), d, ops._ptr(P1), ops._ptr(c1), ops._ptr(ws), ops._ptr(out), d, n, m, k, d,
                           float(np.sqrt(np.float32(d))), ops._stream()))
        outs.append(out)
    assert torch.isfinite(outs[1]).all()
    close(outs[1], outs[0], 2e-5)


@pytest.mark.parametrize('n,dim', [(14336, 36), (3584, 72), (896, 144), (224, 288), (16, 36), (21, 72), (1001, 20), (333, 100),
                                   (62, 260), (4099, 4)])
def test_fused_self_attention_matches_unfused_chain(pk, n, dim):
    """The encoder's self-attention as one kernel (csrc/selfattn16.hip: a wave per query, its 16 neighbours = the 16 MFMA
    columns, weights staged through LDS and padded to the tile grid in the kernel) against the unfused five-launch
    chain: every encoder width of the two published configurations, widths that are not multiples of 16 (padding in
    channels AND hidden units), row counts that are not multiples of the 4 queries of a workgroup, exactly as many points
    as neighbours."""
    rng = np.random.default_rng(n + dim)
    x = rng.normal(size=(n, dim)).astype(np.float32)
    pos = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    ptl = pk.point_transformer_layer
    layer = ptl.PointTransformerLayer(dim, num_neighbors=16).cuda()
    layer.load_state_dict(pk.configs.fill_state_dict(layer, 77 + dim))
    args = (dev(x)[None], dev(pos)[None])
    with torch.no_grad():
        fused = layer(*args)[0]
        with pk.kernels(fused_attention=False):
            chain = layer(*args)[0]
    assert torch.isfinite(fused).all() and fused.shape == (n, dim)
    close(fused, chain, 2e-5)


@pytest.mark.parametrize('case', gc.TRACK_CASES, ids=lambda c: c['name'])
def test_perform_inference_tracks_and_gt_labels(pk, case):
    """D8 branches: track_mode 'all' and ground-truth 1-NN labelling, against the reference's vectors."""
    pcl, sem, target, pa, ia, inf, esd, dsd = gc.track_inputs(case)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    dec.load_state_dict(dsd)
    res = pk.inference.perform_inference(
        pcl.clone(), sem.copy(), target.copy(), [enc, dec], torch.device('cuda:0'), 'if', inf['min_z'],
        inf['cube_bounds'], inf['color_mode'], case['time_idx'], None, sample_implicit=True,
        num_sample=case['num_sample'], point_sample_mode='grid', batch_size=case['batch_size'],
        predict_segmentation=False, track_mode='all', semantic_classes=13, density_threshold=0.5,
        data_kind='greater', cube_mode=4, compress_air=True, point_occupancy_radius=0.8)
    g = load_golden('g11_tracks_' + case['name'])
    close(res['pcl_abstract'], g['pcl_abstract'])
    close(res['features_global'], g['features_global'])
    out, ref = res['implicit_output'], g['implicit_output']
    # the merged mark_track channel is an argmax over reruns of scores compared with 0.5: equal wherever no
    # score sits within tolerance of a decision boundary
    same_id = out[:, 4] == ref[:, 4]
    assert same_id.mean() > 0.995
    cols = [0, 1, 2, 3]
    close(out[:, cols], ref[:, cols])
    assert abs(res['output_solid'].shape[0] - int(g['n_solid'][0])) <= int((np.abs(ref[:, 0] - 0.5) < TOL).sum())
    if res['output_solid'].shape[0] == int(g['n_solid'][0]):
        # nearest target point + label (sklearn KDTree in the reference, streaming k=1 kernel here)
        assert (res['gt_solid'] == g['gt_solid']).all(axis=1).mean() > 0.995
        assert (res['gt_air'] == g['gt_air']).all(axis=1).mean() > 0.995
        assert res['gt_air'].shape[1] == 2


# ------------------------------------------------------------------ opt-in split-precision kernels: bf16 x 3 pieces
# (6 products, round 5) and fp16 x 2 pieces (3 products, round 6), selected per call (`with pk.kernels(...)`)
SPLIT_SCHEMES = ['bf16x6', 'f16x3']


@pytest.mark.parametrize('n,n_out,relu_in,with_res', [(1000, 416, True, True), (257, 416, True, False), (32, 832, False, False),
                                                      (5, 208, False, True), (3000, 832, False, False), (256, 1664, True, True)])
@pytest.mark.parametrize('scheme', SPLIT_SCHEMES)
def test_split_precision_rowlin_against_fp64(pk, n, n_out, relu_in, with_res, scheme):
    """csrc/trunk_bf16x6.hip alone: y = [res +] W [relu](x) + b, K = 416, split operands (bf16 x 3 pieces, six partial
    products / fp16 x 2 pieces, three), against fp64 -- at the accuracy of an fp32 GEMM (a few 2^-24 of sum |w||x|), ragged
    row counts included.  The SAME bound for both schemes."""
    rng = np.random.default_rng(n + n_out)
    x = (3.0 * rng.normal(size=(n, 416))).astype(np.float32)
    w = (rng.normal(size=(n_out, 416)) / np.sqrt(416)).astype(np.float32)
    b = rng.normal(size=(n_out,)).astype(np.float32)
    r = rng.normal(size=(n, n_out)).astype(np.float32) if with_res else None
    xin = np.maximum(x, 0) if relu_in else x
    ref = xin.astype(np.float64) @ w.astype(np.float64).T + b + (r if with_res else 0.0)
    got = pk.ops.rowlin_bf16x6(dev(x), dev(w), dev(b), relu_in=relu_in, res=None if r is None else dev(r), scheme=scheme)
    f32 = (T(xin) @ T(w).T + T(b) + (T(r) if with_res else 0.0)).numpy()
    e6, e32 = np.abs(got.cpu().numpy() - ref).max(), np.abs(f32 - ref).max()
    scale = (np.abs(xin).astype(np.float64) @ np.abs(w).astype(np.float64).T).max()
    print('\n[rowlin %s %d x %d] |split - f64| %.3g, |torch f32 - f64| %.3g, sum|w||x| %.3g' % (scheme, n, n_out, e6, e32, scale))
    assert e6 <= 8 * 2.0 ** -24 * scale
    # in place over the residual rows (how the decoder calls it)
    if with_res:
        rr = dev(r).clone()
        pk.ops.rowlin_bf16x6(dev(x), dev(w), dev(b), relu_in=relu_in, res=rr, out=rr, scheme=scheme)
        assert torch.equal(rr, got)


@pytest.mark.parametrize('n', [32256, 1500, 129, 128, 127, 17, 1])
def test_fused_fp16_residual_block_against_fp64(pk, n):
    """csrc/resblock_f16x3.hip: y = x + W1 relu(W0 relu(x) + b0) + b1 in ONE launch (fp16 x 2 pieces, three products, the
    hidden activation in registers) against fp64 at the accuracy of two chained fp32 GEMMs, against the two-launch form of
    csrc/trunk_bf16x6.hip (same products: equal to rounding), ragged row counts, and in place (how the decoder calls it)."""
    rng = np.random.default_rng(n)
    x = (2.0 * rng.normal(size=(n, 416))).astype(np.float32)
    w0, w1 = ((rng.normal(size=(416, 416)) / np.sqrt(416)).astype(np.float32) for _ in range(2))
    b0, b1 = (rng.normal(size=(416,)).astype(np.float32) for _ in range(2))
    x64, w064, w164 = x.astype(np.float64), w0.astype(np.float64), w1.astype(np.float64)
    h64 = np.maximum(x64, 0) @ w064.T + b0
    ref = x64 + np.maximum(h64, 0) @ w164.T + b1
    got = pk.ops.resblock_f16x3(dev(x), dev(w0), dev(b0), dev(w1), dev(b1))
    scale1 = (np.abs(np.maximum(x64, 0)) @ np.abs(w064).T).max()
    scale2 = (np.abs(np.maximum(h64, 0)) @ np.abs(w164).T).max()
    # layer 2 sees layer 1's error through |W1| (row sums ~ 416 / sqrt(416) * E|w|): bound both terms generously
    bound = 8 * 2.0 ** -24 * (scale2 + scale1 * np.abs(w164).sum(axis=1).max())
    err = np.abs(got.cpu().numpy() - ref).max()
    print('\n[resblock f16x3 %d rows] |fused - f64| %.3g (bound %.3g)' % (n, err, bound))
    assert torch.isfinite(got).all() and err <= bound
    h = pk.ops.rowlin_bf16x6(dev(x), dev(w0), dev(b0), relu_in=True, scheme='f16x3')
    two = pk.ops.rowlin_bf16x6(h, dev(w1), dev(b1), relu_in=True, res=dev(x), scheme='f16x3')
    assert np.abs((got - two).cpu().numpy()).max() <= bound       # (h is rounded independently in the two forms)
    xin = dev(x).clone()
    pk.ops.resblock_f16x3(xin, dev(w0), dev(b0), dev(w1), dev(b1), out=xin)
    assert torch.equal(xin, got)


@pytest.mark.parametrize('n,n_out,after', [(1500, 416, False), (1500, 416, True), (700, 832, False), (257, 208, True)])
def test_split_precision_rowlin_masked_epilogue(pk, n, n_out, after):
    """occ4d_rowlin_bf16x6_masked_f32 (training data gradients): y = [mask > 0] (x W^T [+ res]) [+ res] -- the residual
    before or after the mask -- against fp64 at the accuracy of an fp32 GEMM, and the masked zeros exactly zero."""
    rng = np.random.default_rng(3 * n + n_out)
    x = rng.normal(size=(n, 416)).astype(np.float32)
    w = (rng.normal(size=(n_out, 416)) / np.sqrt(416)).astype(np.float32)
    r = rng.normal(size=(n, n_out)).astype(np.float32)
    m = rng.normal(size=(n, n_out)).astype(np.float32)
    m[::7, ::5] = 0.0                                               # (mask == 0 counts as masked)
    lin = x.astype(np.float64) @ w.astype(np.float64).T
    ref = np.where(m > 0, lin, 0.0) + r if after else np.where(m > 0, lin + r, 0.0)
    got = pk.ops.rowlin_bf16x6(dev(x), dev(w), None, res=dev(r), mask=dev(m), res_after_mask=after).cpu().numpy()
    scale = (np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T).max()
    assert np.abs(got - ref).max() <= 8 * 2.0 ** -24 * scale
    if not after:
        assert (got[m <= 0] == 0).all()
    else:
        assert (got[m <= 0] == r[m <= 0]).all()


@pytest.mark.parametrize('scheme', SPLIT_SCHEMES)
@pytest.mark.parametrize('case', gc.DEC_CASES + gc.DEC_TWOLEVEL_CASES, ids=lambda c: c['name'])
def test_decoder_entirely_on_split_precision(pk, scheme, case):
    """Attention AND trunk on the split-precision kernels (every GEMM of the decoder except lin_in / lin_out and the
    per-scene tables): the golden vectors at the fp32 path's own bar, in both split schemes."""
    two = case in gc.DEC_TWOLEVEL_CASES
    q, abstract, fglob, ia, sd = (gc.dec_twolevel_inputs if two else gc.dec_inputs)(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    g = load_golden('g8_dec_' + case['name'])
    kw = dict(knn_local=dev(g['knn_local']), knn_cross=dev(g['knn_cross'])) if two else {}
    with torch.no_grad(), pk.kernels(precision=scheme):
        out, pen = net(dev(q), dev(abstract), dev(fglob), None, **kw)
    print('\n[%s all %s] |out - ref| %.3g' % (scheme, case['name'], float(np.abs(out.cpu().numpy() - g['output']).max())))
    close(out, g['output'], 2e-5)
    close(pen[:, ::8], g['penult'], 4e-5 if two else 2e-5)


@pytest.mark.parametrize('scheme', SPLIT_SCHEMES)
@pytest.mark.parametrize('case', gc.DEC_CASES + gc.DEC_TWOLEVEL_CASES, ids=lambda c: c['name'])
def test_decoder_with_split_precision_attention(pk, scheme, case):
    """Opt-in modes of csrc/crossattn_bf16x6.hip: every GEMM of the two cross-attention layers on 16x16x32 MFMAs with
    both operands split (bf16 x 3 pieces, six partial products / fp16 x 2 pieces, three): the golden vectors at the fp32
    path's own bar, and the library must really have taken that kernel (the results differ from the fp32 kernel's)."""
    two = case in gc.DEC_TWOLEVEL_CASES
    q, abstract, fglob, ia, sd = (gc.dec_twolevel_inputs if two else gc.dec_inputs)(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    g = load_golden('g8_dec_' + case['name'])
    kw = dict(knn_local=dev(g['knn_local']), knn_cross=dev(g['knn_cross'])) if two else {}
    with torch.no_grad():
        with pk.kernels(logit_precision=scheme):
            out, pen = net(dev(q), dev(abstract), dev(fglob), None, **kw)
        out32, pen32 = net(dev(q), dev(abstract), dev(fglob), None, **kw)
    close(out, g['output'], 2e-5)
    close(pen[:, ::8], g['penult'], 4e-5 if two else 2e-5)
    d = float((out - out32).abs().max())
    print('\n[%s %s] |split - f32 path| %.3g, |split - ref| %.3g, |f32 path - ref| %.3g' % (
        scheme, case['name'], d, float(np.abs(out.cpu().numpy() - g['output']).max()),
        float(np.abs(out32.cpu().numpy() - g['output']).max())))
    assert 0.0 < d < 2e-5                 # a different kernel ran, and it agrees


# ------------------------------------------------------------------ edge cases
def test_empty_and_single_query_batches(pk):
    case = gc.DEC_CASES[0]
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    a, g = dev(abstract), dev(fglob)
    with torch.no_grad():
        out0, pen0 = net(dev(q[:0]), a, g, None)
        assert out0.shape == (0, ia['d_out']) and pen0.shape == (0, ia['d_hidden'])
        full, _ = net(dev(q[:10]), a, g, None)
        one, _ = net(dev(q[3:4]), a, g, None)
    assert (one[0] - full[3]).abs().max() <= 1e-5
    # the device-resident driver with fewer queries than one mini-batch, and with none at all
    enc_pa, _, inf = pk.configs.model_args('greater', 512)
    enc = pk.model.PointCompletionNetV3(**enc_pa).cuda().eval()
    enc.load_state_dict(pk.configs.fill_state_dict(pk.configs.encoder_param_shapes(enc_pa), 3))
    pcl = pk.configs.synthetic_pcl('greater', 512, 4, 3).cuda()
    with torch.no_grad():
        res = pk.inference.infer_device(pcl, dev(q[:7]), enc, net, 4, 'rgb_nosigmoid')
        assert res['implicit_output'].shape == (7, 5) and torch.isfinite(res['implicit_output']).all()
        res0 = pk.inference.infer_device(pcl, dev(q[:0]), enc, net, 4, 'rgb_nosigmoid')
        assert res0['implicit_output'].shape == (0, 5)


def test_minimum_cloud_sizes(pk):
    """Smallest clouds the architecture admits: every level needs >= its neighbour count."""
    pa, _, _ = pk.configs.model_args('greater', 432)       # 432 -> 144 -> 48 -> 16 points (K = 16 at the last level)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval()
    sd = pk.configs.fill_state_dict(pk.configs.encoder_param_shapes(pa), 9)
    enc.load_state_dict(sd)
    pcl = pk.configs.synthetic_pcl('greater', 432, 4, 9)
    from oracle import path as op
    with torch.no_grad():
        out, xg, _ = enc(pcl.cuda(), False)
    ref_out, ref_xg = op.encoder_forward(sd, pa, pcl)
    assert out.shape == (1, 16, 291)
    close(out[0], ref_out[0])
    close(xg[0], ref_xg[0])
    with pytest.raises(AssertionError):                    # one point fewer at the last level: kNN cannot be served
        pa2, _, _ = pk.configs.model_args('greater', 405)  # 405 -> 135 -> 45 -> 15
        enc2 = pk.model.PointCompletionNetV3(**pa2).cuda().eval()
        with torch.no_grad():
            enc2(pk.configs.synthetic_pcl('greater', 405, 4, 9).cuda(), False)
