"""The C ABI is self-sufficient (SURVEY.md 8(b) "minimum set"): this file binds libocc4d.so with
ctypes ALONE -- its own struct definitions and argument types restated from include/occ4d.h, exactly the stub a
maintainer of the reference would write -- and reproduces the reference's golden vectors for

    G2 / G3  PointTransformerLayer / PointTransformerBlock   occ4d_pt_layer_prepare_f32 + occ4d_pt_layer_fwd_f32
    G4       DownTransition                                  occ4d_fps_f32 + occ4d_knn_f32 + occ4d_down_pool_fwd_f32
    G8       LocalPclResnetFC                                occ4d_decoder_prepare_f32 + _prepare_scene_f32 + _query_fwd_f32

from the reference's own parameter tensors (state_dict layout, torch Linear (out, in)).  torch is used for device memory
only; nothing of occlusions-4d_amd/ops.py, _lib.py or the nn.Module mirrors takes part (golden_cases supplies the seeded
INPUTS, as for every other parity test).  The stage packers are checked against their index formulas restated with
torch reshapes."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
T = gc.as_tensor
F, I, S = C.c_void_p, C.c_void_p, C.c_void_p


class LayerW(C.Structure):                     # occ4d_pt_layer_weights
    _fields_ = [(n, C.c_int32) for n in ('dim', 'dim2', 'pos_hidden', 'cross', 'd_in', 'd_out', 'r0', 'r1')] + \
               [(n, C.c_void_p) for n in ('to_q', 'to_k', 'to_v', 'pos0_w', 'pos0_b', 'pos2_w', 'pos2_b', 'attn0_w', 'attn0_b',
                                          'attn2_w', 'attn2_b', 'pre_w', 'pre_b', 'post_w', 'post_b')]


class DecoderW(C.Structure):                   # occ4d_decoder_weights
    _fields_ = [(n, C.c_int32) for n in ('d_in', 'n_freq', 'd_hidden', 'd_out', 'd_latent', 'd_latent_local', 'n_blocks',
                                         'n_cross', 'k_local', 'k_cross', 'activation', 'lin_in_ld')] + \
               [('base_frequency', C.c_float), ('reserved', C.c_float)] + \
               [(n, C.c_void_p) for n in ('lin_in_w', 'lin_in_b', 'lin_out_w', 'lin_out_b')] + \
               [(n, C.c_void_p * 16) for n in ('lin_z_w', 'lin_z_b', 'fc0_w', 'fc0_b', 'fc1_w', 'fc1_b')] + \
               [('cross_after', C.c_int32 * 4), ('cross', LayerW * 4)]


@pytest.fixture(scope='module')
def lib():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    h = C.CDLL(os.path.join(ROOT, 'occlusions-4d_amd', 'libocc4d.so'))
    LW, DW = C.POINTER(LayerW), C.POINTER(DecoderW)
    sig = {
        'occ4d_last_error': (C.c_char_p, []),
        'occ4d_knn_f32': (C.c_int, [F, C.c_int64, C.c_int, F, C.c_int64, C.c_int, C.c_int, C.c_int, I, C.c_int, F, S]),
        'occ4d_fps_f32': (C.c_int, [F, C.c_int64, C.c_int, C.c_int, I, I, S]),
        'occ4d_gather_rows_f32': (C.c_int, [F, C.c_int64, I, C.c_int, C.c_int, F, C.c_int64, S]),
        'occ4d_pt_layer_prepared_floats': (C.c_int64, [LW, C.c_int]),
        'occ4d_pt_layer_prepare_f32': (C.c_int, [LW, F, C.c_int, S]),
        'occ4d_pt_layer_workspace_floats': (C.c_int64, [LW, C.c_int, C.c_int, C.c_int, C.c_int]),
        'occ4d_pt_layer_fwd_f32': (C.c_int, [LW, F, F, C.c_int64, F, C.c_int64, C.c_int, F, C.c_int64, F, C.c_int64, C.c_int,
                                             C.c_int, I, F, F, C.c_int64, F, C.c_int, C.c_void_p, S]),
        'occ4d_down_pool_fwd_f32': (C.c_int, [F, C.c_int64, C.c_int, C.c_int, F, F, C.c_int, C.c_int, F, F, F, F, C.c_float, I,
                                              C.c_int, C.c_int, F, C.c_int64, F, S]),
        'occ4d_decoder_prepared_floats': (C.c_int64, [DW, C.c_int]),
        'occ4d_decoder_prepare_f32': (C.c_int, [DW, F, C.c_int, S]),
        'occ4d_decoder_scene_floats': (C.c_int64, [DW, C.c_int]),
        'occ4d_decoder_prepare_scene_f32': (C.c_int, [DW, F, F, C.c_int64, F, C.c_int64, F, C.c_int, F, C.c_int, S]),
        'occ4d_decoder_query_workspace_floats': (C.c_int64, [DW, C.c_int, C.c_int, C.c_int]),
        'occ4d_decoder_query_fwd_f32': (C.c_int, [DW, F, F, C.c_int, F, C.c_int64, C.c_int, I, I, F, C.c_int64, F, C.c_int64,
                                                  F, C.c_int, C.c_void_p, S]),
        'occ4d_knn_dists_f32': (C.c_int, [F, C.c_int64, C.c_int, F, C.c_int64, C.c_int, I, C.c_int, C.c_int, F, S]),
        'occ4d_pack_trunk_rows_f32': (C.c_int, [F, C.c_int64, C.c_int, F, S]),
        'occ4d_pack_trunk_cols_f32': (C.c_int, [F, C.c_int64, F, S]),
        'occ4d_pack_trunk4_rows_f32': (C.c_int, [F, C.c_int64, C.c_int, F, S]),
        'occ4d_pack_trunk4_cols_f32': (C.c_int, [F, C.c_int64, F, S]),
        'occ4d_pack_attn16p_stream_f32': (C.c_int, [F, F, F, F, S]),
        'occ4d_trunk_packed_floats': (C.c_int64, [C.c_int]),
        'occ4d_trunk4_packed_floats': (C.c_int64, [C.c_int]),
        'occ4d_pt_cross_attn16p_stream_floats': (C.c_int64, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def ok(lib, rc):
    assert rc == 0, lib.occ4d_last_error().decode()


def dev(a):
    return T(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def buf(n_floats):
    assert n_floats >= 0
    return torch.empty((max(1, n_floats),), dtype=torch.float32, device='cuda')


def err(a, b):
    a = a.detach().cpu().numpy()
    assert a.shape == np.asarray(b).shape, (a.shape, np.asarray(b).shape)
    return float(np.abs(a - b).max())


def layer_struct(sd, dim, dim2, cross, prefix='', block=False, keep=None):
    """occ4d_pt_layer_weights from the reference's state_dict entries (device copies are appended to `keep`)."""
    g = {k[len(prefix):]: dev(v.numpy()) for k, v in sd.items() if k.startswith(prefix)}
    keep.append(g)
    lp = 'layer2.' if block else ''
    w = LayerW(dim=dim, dim2=dim2, pos_hidden=g[lp + 'pos_mlp.0.weight'].shape[0], cross=int(cross), d_in=dim, d_out=dim)
    for field, name in (('to_q', 'to_q.weight'), ('to_k', 'to_k.weight'), ('to_v', 'to_v.weight'),
                        ('pos0_w', 'pos_mlp.0.weight'), ('pos0_b', 'pos_mlp.0.bias'), ('pos2_w', 'pos_mlp.2.weight'),
                        ('pos2_b', 'pos_mlp.2.bias'), ('attn0_w', 'attn_mlp.0.weight'), ('attn0_b', 'attn_mlp.0.bias'),
                        ('attn2_w', 'attn_mlp.2.weight'), ('attn2_b', 'attn_mlp.2.bias')):
        setattr(w, field, g[lp + name].data_ptr())
    if block:
        w.pre_w, w.pre_b = g['layer1.weight'].data_ptr(), g['layer1.bias'].data_ptr()
        w.post_w, w.post_b = g['layer3.weight'].data_ptr(), g['layer3.bias'].data_ptr()
    return w


def run_layer(lib, w, x, pos, x2, pos2, k, flags=0):
    n, m = x.shape[0], (x2.shape[0] if x2 is not None else 0)
    prep = buf(lib.occ4d_pt_layer_prepared_floats(C.byref(w), flags))
    ok(lib, lib.occ4d_pt_layer_prepare_f32(C.byref(w), ptr(prep), flags, stream()))
    ws = buf(lib.occ4d_pt_layer_workspace_floats(C.byref(w), n, m, k, flags))
    out = torch.empty((n, w.d_out if w.post_w else w.dim), dtype=torch.float32, device='cuda')
    ok(lib, lib.occ4d_pt_layer_fwd_f32(C.byref(w), ptr(prep), ptr(x), x.stride(0), ptr(pos), pos.stride(0), n, ptr(x2),
                                       x2.stride(0) if x2 is not None else 0, ptr(pos2),
                                       pos2.stride(0) if pos2 is not None else 0, m, k, None, None, ptr(out), out.stride(0),
                                       ptr(ws), flags, None, stream()))
    torch.cuda.synchronize()
    return out


# ------------------------------------------------------------------ G2 / G3
@pytest.mark.parametrize('flags', [0, 1, 2, 8, 16], ids=['default', 'unfused', 'first_gen', 'generic_linear', 'trunk4'])
@pytest.mark.parametrize('case', gc.PTL_CASES, ids=lambda c: c['name'])
def test_g2_pt_layer_through_the_c_abi(lib, case, flags):
    x, pos, x2, pos2, sd = gc.ptl_inputs(case)
    keep = []
    w = layer_struct(sd, case['dim'], case.get('dim2', case['dim']), x2 is not None, keep=keep)
    out = run_layer(lib, w, dev(x), dev(pos), None if x2 is None else dev(x2), None if pos2 is None else dev(pos2),
                    case['k'], flags)
    assert err(out, load_golden('g2_ptl_' + case['name'])['agg']) <= 1e-4


@pytest.mark.parametrize('case', gc.PTL_REGIME_CASES, ids=lambda c: c['name'])
def test_g2r_regimes_through_the_c_abi(lib, case):
    x, pos, x2, pos2, sd = gc.ptl_inputs(case)
    keep = []
    w = layer_struct(sd, case['dim'], case.get('dim2', case['dim']), x2 is not None, keep=keep)
    out = run_layer(lib, w, dev(x), dev(pos), None if x2 is None else dev(x2), None if pos2 is None else dev(pos2), case['k'])
    g = load_golden('g2r_ptl_' + case['name'])
    assert float(np.abs(out.cpu().numpy().astype(np.float64) - g['agg64']).max()) <= gc.regime_bound(g, 'agg')


@pytest.mark.parametrize('case', gc.PTB_CASES, ids=lambda c: c['name'])
def test_g3_pt_block_through_the_c_abi(lib, case):
    x, pos, x2, pos2, sd = gc.ptb_inputs(case)
    keep = []
    w = layer_struct(sd, case['dim'], case.get('dim2', case['dim']), x2 is not None, block=True, keep=keep)
    out = run_layer(lib, w, dev(x), dev(pos), None if x2 is None else dev(x2), None if pos2 is None else dev(pos2), case['k'])
    assert err(out, load_golden('g3_ptb_' + case['name'])['z']) <= 1e-4


# ------------------------------------------------------------------ G4
@pytest.mark.parametrize('case', gc.DOWN_CASES, ids=lambda c: c['name'])
def test_g4_down_transition_through_the_c_abi(lib, case):
    x, pos, sd = gc.down_inputs(case)
    xd, pd = dev(x), dev(pos)
    g = {k: dev(v.numpy()) for k, v in sd.items()}
    n, k = case['n'], case['k']
    n_new = -(-n // 3)
    inds = torch.empty((n_new,), dtype=torch.int32, device='cuda')
    ok(lib, lib.occ4d_fps_f32(ptr(pd), 3, n, n_new, ptr(inds), None, stream()))
    p_sub = torch.empty((n_new, 3), dtype=torch.float32, device='cuda')
    ok(lib, lib.occ4d_gather_rows_f32(ptr(pd), 3, ptr(inds), n_new, 3, ptr(p_sub), 3, stream()))
    nn = torch.empty((n_new, k), dtype=torch.int32, device='cuda')
    ok(lib, lib.occ4d_knn_f32(ptr(p_sub), 3, n_new, ptr(pd), 3, n, k, 0, ptr(nn), 0, None, stream()))
    z = torch.empty((n_new, case['d_out']), dtype=torch.float32, device='cuda')
    ws = buf(n * case['d_out'])
    layer = case['norm'] == 'layer'
    ok(lib, lib.occ4d_down_pool_fwd_f32(ptr(xd), case['d_in'], n, case['d_in'], ptr(g['mlp.0.weight']), ptr(g['mlp.0.bias']),
                                        case['d_out'], 1 if layer else 0, ptr(g.get('mlp.1.weight')), ptr(g.get('mlp.1.bias')),
                                        None, None, 1e-5, ptr(nn), n_new, k, ptr(z), case['d_out'], ptr(ws), stream()))
    torch.cuda.synchronize()
    gold = load_golden('g4_down_' + case['name'])
    assert np.array_equal(p_sub.cpu().numpy(), gold['p_sub'])
    assert err(z, gold['z']) <= 1e-4


# ------------------------------------------------------------------ G8
def decoder_struct(sd, ia, keep):
    g = {k: dev(v.numpy()) for k, v in sd.items()}
    keep.append(g)
    P = ia['d_in'] * (2 * ia['pos_encoding_freqs'] + 1)
    assert P % 4 == 0
    nb, nc = ia['n_blocks'], ia['cross_attn_layers']
    w = DecoderW(d_in=ia['d_in'], n_freq=ia['pos_encoding_freqs'], d_hidden=ia['d_hidden'], d_out=ia['d_out'],
                 d_latent=ia['d_latent'], d_latent_local=ia['d_latent_local'], n_blocks=nb, n_cross=nc,
                 k_local=ia['num_local_features'], k_cross=ia['cross_attn_neighbors'], activation=0, lin_in_ld=P,
                 base_frequency=0.1)
    w.lin_in_w, w.lin_in_b = g['lin_in.weight'].data_ptr(), g['lin_in.bias'].data_ptr()
    w.lin_out_w, w.lin_out_b = g['lin_out.weight'].data_ptr(), g['lin_out.bias'].data_ptr()
    for i in range(nb):
        w.lin_z_w[i], w.lin_z_b[i] = g['lin_z.%d.weight' % i].data_ptr(), g['lin_z.%d.bias' % i].data_ptr()
        w.fc0_w[i], w.fc0_b[i] = g['blocks.%d.fc_0.weight' % i].data_ptr(), g['blocks.%d.fc_0.bias' % i].data_ptr()
        w.fc1_w[i], w.fc1_b[i] = g['blocks.%d.fc_1.weight' % i].data_ptr(), g['blocks.%d.fc_1.bias' % i].data_ptr()
    for j in range(nc):
        w.cross_after[j] = int((j + 1) * nb / (nc + 1))          # model/implicit.py:265
        w.cross[j] = layer_struct(sd, ia['d_latent'], ia['d_latent_local'], True, prefix='pt_blocks.%d.' % j, block=True,
                                  keep=keep)
    return w


def run_decoder(lib, w, q, abstract, fglob, flags=0, batch=None, knn_local=None, knn_cross=None):
    m, n = abstract.shape[0], q.shape[0]
    prep = buf(lib.occ4d_decoder_prepared_floats(C.byref(w), flags))
    ok(lib, lib.occ4d_decoder_prepare_f32(C.byref(w), ptr(prep), flags, stream()))
    scene = buf(lib.occ4d_decoder_scene_floats(C.byref(w), m))
    feats = abstract[:, 3:].contiguous()
    ok(lib, lib.occ4d_decoder_prepare_scene_f32(C.byref(w), ptr(prep), ptr(abstract), abstract.stride(0), ptr(feats),
                                                feats.stride(0), ptr(fglob), m, ptr(scene), flags, stream()))
    out = torch.empty((n, w.d_out), dtype=torch.float32, device='cuda')
    pen = torch.empty((n, w.d_hidden), dtype=torch.float32, device='cuda')
    batch = batch or max(1, n)
    ws = buf(lib.occ4d_decoder_query_workspace_floats(C.byref(w), min(n, batch), m, flags))
    for lo in range(0, n, batch):
        c = min(batch, n - lo)
        ok(lib, lib.occ4d_decoder_query_fwd_f32(C.byref(w), ptr(prep), ptr(scene), m, ptr(q[lo:]), q.stride(0), c,
                                                ptr(None if knn_local is None else knn_local[lo:]),
                                                ptr(None if knn_cross is None else knn_cross[lo:]),
                                                ptr(out[lo:]), out.stride(0), ptr(pen[lo:]), pen.stride(0), ptr(ws), flags,
                                                None, stream()))
    torch.cuda.synchronize()
    return out, pen


@pytest.mark.parametrize('flags', [0, 1, 2, 8, 16], ids=['default', 'unfused', 'first_gen', 'generic_linear', 'trunk4'])
@pytest.mark.parametrize('case', gc.DEC_CASES, ids=lambda c: c['name'])
def test_g8_decoder_through_the_c_abi(lib, case, flags):
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    keep = []
    w = decoder_struct(sd, ia, keep)
    out, pen = run_decoder(lib, w, dev(q), dev(abstract), dev(fglob), flags, batch=100 if flags == 0 else None)
    g = load_golden('g8_dec_' + case['name'])
    assert err(out, g['output']) <= 1e-4
    assert err(pen[:, ::8], g['penult']) <= 1e-4


@pytest.mark.parametrize('flags', [0, 1], ids=['default', 'unfused'])
@pytest.mark.parametrize('case', gc.DEC_TWOLEVEL_CASES, ids=lambda c: c['name'])
def test_g8t_two_level_cloud_with_the_reference_lists_through_the_c_abi(lib, case, flags):
    """CARLA-structured abstract cloud (every coarse point twice): the caller hands the library the neighbour lists the
    reference run took (knn_local / knn_cross of occ4d_decoder_query_fwd_f32) and EVERY row matches the reference."""
    q, abstract, fglob, ia, sd = gc.dec_twolevel_inputs(case)
    g = load_golden('g8_dec_' + case['name'])
    keep = []
    w = decoder_struct(sd, ia, keep)
    kl, kc = dev(g['knn_local']), dev(g['knn_cross'])
    assert kl.dtype == torch.int32 and kc.dtype == torch.int32
    out, pen = run_decoder(lib, w, dev(q), dev(abstract), dev(fglob), flags, batch=200, knn_local=kl, knn_cross=kc)
    assert err(out, g['output']) <= 1e-4
    assert err(pen[:, ::8], g['penult']) <= 1e-4
    # the distances the library derives from the given indices are the reference's, bit for bit
    d = torch.empty(kl.shape, dtype=torch.float32, device='cuda')
    a = dev(abstract)
    ok(lib, lib.occ4d_knn_dists_f32(ptr(dev(q)), 4, q.shape[0], ptr(a), a.stride(0), a.shape[0], ptr(kl), kl.shape[1], 1,
                                    ptr(d), stream()))
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), g['knn_local_dists'])
    # out-of-range entries of a caller's list are clamped, never dereferenced
    bad = kc.clone()
    bad[::7, 3] = 1 << 30
    bad[::5, 0] = -5
    out2, _ = run_decoder(lib, w, dev(q), dev(abstract), dev(fglob), flags, knn_local=kl, knn_cross=bad)
    assert torch.isfinite(out2).all()


@pytest.mark.parametrize('case', gc.DEC_REGIME_CASES, ids=lambda c: c['name'])
def test_g8r_regimes_through_the_c_abi(lib, case):
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    keep = []
    w = decoder_struct(sd, ia, keep)
    out, pen = run_decoder(lib, w, dev(q), dev(abstract), dev(fglob))
    g = load_golden('g8r_dec_' + case['name'])
    assert float(np.abs(out.cpu().numpy().astype(np.float64) - g['output64']).max()) <= gc.regime_bound(g, 'output')
    assert float(np.abs(pen[:, ::8].cpu().numpy().astype(np.float64) - g['penult64']).max()) <= gc.regime_bound(g, 'penult')


def test_argument_errors_are_status_codes_not_crashes(lib):
    case = gc.DEC_CASES[0]
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    keep = []
    w = decoder_struct(sd, ia, keep)
    w.activation = 7
    assert lib.occ4d_decoder_prepared_floats(C.byref(w), 0) == -1
    assert b'Unknown activation' in lib.occ4d_last_error()
    w.activation = 0
    prep = buf(lib.occ4d_decoder_prepared_floats(C.byref(w), 0))
    ok(lib, lib.occ4d_decoder_prepare_f32(C.byref(w), ptr(prep), 0, stream()))
    a = dev(abstract[:5])                     # 5 abstract points cannot serve 8 / 14 neighbours
    scene = buf(lib.occ4d_decoder_scene_floats(C.byref(w), 5))
    rc = lib.occ4d_decoder_prepare_scene_f32(C.byref(w), ptr(prep), ptr(a), a.stride(0), ptr(a[:, 3:].contiguous()), 288,
                                             ptr(dev(fglob)), 5, ptr(scene), 0, stream())
    assert rc == -1 and b'abstract points' in lib.occ4d_last_error()


# ------------------------------------------------------------------ packers against their index formulas
def test_packers_match_the_layouts_of_the_header(lib):
    rng = np.random.default_rng(3)
    w = dev(rng.normal(size=(832, 416)).astype(np.float32))
    sq = dev(rng.normal(size=(416, 416)).astype(np.float32))

    def call(fn, n_floats, *args):
        out = torch.full((n_floats,), float('nan'), dtype=torch.float32, device='cuda')
        ok(lib, fn(*args, ptr(out), stream()))
        torch.cuda.synchronize()
        return out
    # "rows" packing: P[s][(nt*26 + t)*256 + (g*16 + r)*4 + e] = W[32 s + 16 nt + r][16 t + 4 g + e]; + a copy of stage 0
    got = call(lib.occ4d_pack_trunk_rows_f32, lib.occ4d_trunk_packed_floats(832), ptr(w), 416, 832)
    p = w.reshape(26, 2, 16, 26, 4, 4).permute(0, 1, 3, 4, 2, 5).reshape(26, -1)
    assert torch.equal(got, torch.cat([p, p[:1]]).reshape(-1))
    # "cols" packing: P[j][(nt*2 + tt)*256 + (g*16 + r)*4 + e] = W[16 nt + r][32 j + 16 tt + 4 g + e]
    got = call(lib.occ4d_pack_trunk_cols_f32, lib.occ4d_trunk_packed_floats(416), ptr(sq), 416)
    p = sq.reshape(26, 16, 13, 2, 4, 4).permute(2, 0, 3, 4, 1, 5).reshape(13, -1)
    assert torch.equal(got, torch.cat([p, p[:1]]).reshape(-1))
    # half-CU packings
    got = call(lib.occ4d_pack_trunk4_rows_f32, lib.occ4d_trunk4_packed_floats(832), ptr(w), 416, 832)
    p = w.reshape(52, 16, 26, 4, 4).permute(0, 2, 3, 1, 4).reshape(52, -1)
    assert torch.equal(got, torch.cat([p, p[:1]]).reshape(-1))
    got = call(lib.occ4d_pack_trunk4_cols_f32, lib.occ4d_trunk4_packed_floats(416), ptr(sq), 416)
    p = sq.reshape(26, 16, 26, 4, 4).permute(2, 0, 3, 1, 4).reshape(26, -1)
    assert torch.equal(got, torch.cat([p, p[:1]]).reshape(-1))
    # strided source (the lin_z local columns are a view of the parameter)
    wide = dev(rng.normal(size=(416, 544)).astype(np.float32))
    got = call(lib.occ4d_pack_trunk_rows_f32, lib.occ4d_trunk_packed_floats(416), ptr(wide[:, 128:]), 544, 416)
    v = wide[:, 128:].contiguous()
    p = v.reshape(13, 2, 16, 26, 4, 4).permute(0, 1, 3, 4, 2, 5).reshape(13, -1)
    assert torch.equal(got, torch.cat([p, p[:1]]).reshape(-1))
    # attention stream: 52 hidden stages (26 W2 fragments + 2 Wp fragments), then P2 in two stages, 4 zero fragments last
    w2, wp, p2 = (dev(rng.normal(size=s).astype(np.float32)) for s in ((416, 832), (832, 32), (416, 32)))
    got = call(lib.occ4d_pack_attn16p_stream_f32, lib.occ4d_pt_cross_attn16p_stream_floats(), ptr(w2), ptr(wp), ptr(p2))
    a = w2.reshape(26, 16, 52, 4, 4).permute(2, 0, 3, 1, 4).reshape(52, 26 * 256)
    b = wp.reshape(52, 16, 2, 4, 4).permute(0, 2, 4, 1, 3).reshape(52, 2 * 256)
    c = p2.reshape(26, 16, 2, 4, 4).permute(0, 2, 4, 1, 3).reshape(26, 2 * 256)
    want = torch.cat([torch.cat([a, b], 1), c[:14].reshape(1, -1),
                      torch.cat([c[14:].reshape(1, -1), torch.zeros((1, 1024), device='cuda')], 1)], 0)
    assert torch.equal(got, want.reshape(-1))


def test_the_binding_printed_in_integration_md_runs_as_printed():
    """INTEGRATION.md B shows the ctypes patch a maintainer of the reference would apply to model/implicit.py.  The code
    block is executed verbatim (only the library path is substituted) on a parameter container with the reference's
    attribute names, and must reproduce the reference's golden vector G8."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## B. Library-level binding'):]
    code = sec[sec.index('```python') + len('```python'):]
    code = code[:code.index('```')]
    code = code.replace('/path/to/occlusions-4d_amd/libocc4d.so', os.path.join(ROOT, 'occlusions-4d_amd', 'libocc4d.so'))
    ns = {}
    exec(compile(code, 'INTEGRATION.md#B', 'exec'), ns)
    import occlusions4d_amd as pk           # (parameter container with the reference's attribute names; no compute through it)
    case = gc.DEC_CASES[1]
    q, abstract, fglob, ia, sd = gc.dec_inputs(case)
    net = pk.implicit.LocalPclResnetFC(**ia).cuda().eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out, pen = ns['forward'](net, dev(q), dev(abstract), dev(fglob), None)
    torch.cuda.synchronize()
    g = load_golden('g8_dec_' + case['name'])
    assert err(out, g['output']) <= 1e-4 and err(pen[:, ::8], g['penult']) <= 1e-4
