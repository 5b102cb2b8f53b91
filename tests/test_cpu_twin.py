"""The g++ twin of the C ABI (occlusions-4d_amd/csrc_cpu/occ4d_twin.cpp -> libocc4d_cpu.so; SURVEY.md 8(b): "each with a
CPU twin compiled by g++ for config 1"; BASELINE configs[0]: "runs without a GPU").  Explicit opt-in only
(occlusions4d_amd.cpu_twin.enable()); the product stays loud without the HIP library (test_abi.py).

(i)  BASELINE configs[0] through the PRODUCT modules (PointCompletionNetV3, LocalPclResnetFC, perform_inference) on the
     twin = the reference's golden run G10 (and the checkpoint fixtures G17).
(ii) Container only: the REFERENCE's own eval/inference.py -- load_models and perform_inference -- driving the product
     modules bound as INTEGRATION.md section A binds them: the drop-in boundary executed under the reference's caller."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import occlusions4d_amd as pk
from conftest import GOLDEN, load_golden
from oracle import ref_import


@pytest.fixture
def twin():
    pk.cpu_twin.enable()
    try:
        yield pk
    finally:
        pk.cpu_twin.disable()


def _kw(case, inf):
    return dict(sample_implicit=True, num_sample=case['num_sample'], point_sample_mode='grid', batch_size=case['batch_size'],
                predict_segmentation=inf['predict_segmentation'], track_mode='none', semantic_classes=13,
                density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4, compress_air=True)


def test_the_twin_is_never_loaded_unasked():
    assert not pk.cpu_twin.enabled() and not pk._lib.is_twin()
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        pk.ops.knn(torch.zeros(4, 3), torch.zeros(4, 3), 2)            # CPU tensors: rejected, no fallback
    with pk.cpu_twin.loaded():
        assert pk._lib.lib().occ4d_is_cpu_twin() == 1
        idx = pk.ops.knn(torch.tensor([[0., 0, 0]]), torch.tensor([[1., 0, 0], [0.5, 0, 0], [3, 0, 0]]), 2)
        assert idx.tolist() == [[1, 0]]
        with pytest.raises(NotImplementedError, match='not part of the CPU twin'):
            pk._lib.lib().occ4d_pack_trunk_rows_f32(None, 0, 0, None, None)
    assert not pk.cpu_twin.enabled()
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        pk.ops.knn(torch.zeros(4, 3), torch.zeros(4, 3), 2)


def test_twin_exports_the_header_prototypes_of_the_minimum_set(twin):
    lib = pk._lib.lib()
    for name in ('occ4d_knn_f32', 'occ4d_fps_f32', 'occ4d_fps_start_f32', 'occ4d_linear_f32', 'occ4d_pt_layer_prepare_f32',
                 'occ4d_pt_layer_fwd_f32', 'occ4d_down_pool_fwd_f32', 'occ4d_decoder_prepare_f32',
                 'occ4d_decoder_prepare_scene_f32', 'occ4d_decoder_query_fwd_f32', 'occ4d_squash_f32',
                 'occ4d_grid_points_f32', 'occ4d_split_count_f32', 'occ4d_split_write_f32', 'occ4d_posenc_f32'):
        fn = getattr(lib, name)
        assert fn.argtypes == pk._lib.SIGNATURES[name][1] and fn.restype == pk._lib.SIGNATURES[name][0], name
    assert lib.occ4d_abi_version() == pk._lib.ABI_VERSION


@pytest.mark.parametrize('case', gc.INFER_CASES, ids=lambda c: c['name'])
def test_config1_through_the_product_modules_on_the_twin(twin, case):
    """BASELINE configs[0] (n_points 2048, video_len 4, 8640 grid queries) and the small CARLA case: the product's
    perform_inference on CPU tensors = the reference's run (G10) at 1e-4; FPS subsets bit-exact."""
    g = load_golden('g10_infer_' + case['name'])
    pcl, pa, ia, inf, esd, dsd = gc.infer_inputs(case)
    enc = pk.model.PointCompletionNetV3(**pa).eval()
    enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).eval()
    dec.load_state_dict(dsd)
    lists = (g['knn_local'], g['knn_cross']) if 'knn_local' in g else None      # (CARLA: the reference run's own tie order)
    res = pk.inference.perform_inference(pcl.clone(), None, None, [enc, dec], torch.device('cpu'), 'if', inf['min_z'],
                                         inf['cube_bounds'], inf['color_mode'], case['time_idx'], None,
                                         neighbour_lists=lists, **_kw(case, inf))
    assert np.array_equal(res['pcl_abstract'][:, :3], g['pcl_abstract'][:, :3])
    assert np.abs(res['pcl_abstract'] - g['pcl_abstract']).max() < 1e-4
    assert np.abs(res['features_global'] - g['features_global']).max() < 1e-4
    assert res['implicit_output'].shape == g['implicit_output'].shape
    assert np.abs(res['implicit_output'] - g['implicit_output']).max() < 1e-4
    near = int((np.abs(g['implicit_output'][:, 0] - 0.5) < 1e-4).sum())
    assert abs(res['output_solid'].shape[0] - int(g['n_solid'][0])) <= near
    assert res['output_air'].dtype == np.float64 and res['output_air'].shape[1] == 5


@pytest.mark.skipif(not ref_import.available(), reason='the reference is only mounted in the build container')
@pytest.mark.parametrize('case', gc.CKPT_CASES, ids=lambda c: c['name'])
def test_the_references_own_caller_drives_the_product_modules(twin, case):
    """INTEGRATION.md section A, executed: the reference's eval/inference.py with `model` / `implicit` bound to this
    package.  ITS load_models (:23-80) builds the PRODUCT's networks from a reference-format checkpoint (constructor
    kwargs, state_dict keys, legacy rename, fps_random_start override) and ITS perform_inference (:83-325) -- grid
    sampling, batch loop, post-ops, threshold split -- calls the product's forwards.  The result must equal what the
    same code returned with the reference's own modules (fixture G17)."""
    ref = ref_import.load()
    inf_mod = ref.inference
    saved = (inf_mod.model, inf_mod.implicit)
    inf_mod.model, inf_mod.implicit = pk.model, pk.implicit        # sys.modules['model'] / ['implicit'] of section A
    real_load = torch.load
    torch.load = lambda *a, **k: real_load(*a, **dict(k, weights_only=False))     # (torch >= 2.6 default; the reference predates it)
    try:
        (nets, _, _, pcl_args, _, epoch) = inf_mod.load_models(os.path.join(GOLDEN, gc.CKPT_DIR), torch.device('cpu'),
                                                               epoch=case['epoch_arg'])
    finally:
        torch.load = real_load
        inf_mod.model, inf_mod.implicit = saved
    assert isinstance(nets[0], pk.model.PointCompletionNetV3) and isinstance(nets[1], pk.implicit.LocalPclResnetFC)
    assert epoch == case['epoch'] and pcl_args['fps_random_start'] is False
    for net in nets:
        net.eval()
    _, _, inf = gc.ckpt_model_args(case)
    pcl = pk.configs.synthetic_pcl(gc.CKPT_INFER['kind'], gc.CKPT_INFER['n'], gc.CKPT_INFER['video_len'], gc.CKPT_INFER['seed'])
    with torch.no_grad():                                            # (eval/test.py:32-34)
        res = inf_mod.perform_inference(
            pcl.clone(), None, None, nets, torch.device('cpu'), 'if', inf['min_z'], inf['cube_bounds'], inf['color_mode'],
            gc.CKPT_INFER['time_idx'], None, sample_implicit=True, num_sample=gc.CKPT_INFER['num_sample'],
            point_sample_mode='grid', batch_size=gc.CKPT_INFER['batch_size'], predict_segmentation=False,
            track_mode='none', semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4,
            compress_air=True)
    g = load_golden('g17_ckpt_' + case['name'])
    assert np.array_equal(res['pcl_abstract'][:, :3], g['pcl_abstract'][:, :3])
    assert np.abs(res['pcl_abstract'] - g['pcl_abstract']).max() < 1e-4
    assert np.abs(res['implicit_output'] - g['implicit_output']).max() < 1e-4
    assert res['output_solid'].shape[0] == int(g['n_solid'][0])


@pytest.mark.skipif(not ref_import.available(), reason='the reference is only mounted in the build container')
def test_the_references_perform_inference_on_config1(twin):
    """The same at BASELINE configs[0]: the reference's perform_inference over the product's networks = G10."""
    case = gc.INFER_CASES[0]
    ref = ref_import.load()
    g = load_golden('g10_infer_' + case['name'])
    pcl, pa, ia, inf, esd, dsd = gc.infer_inputs(case)
    enc = pk.model.PointCompletionNetV3(**pa).eval()
    enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).eval()
    dec.load_state_dict(dsd)
    with torch.no_grad():
        res = ref.inference.perform_inference(pcl.clone(), None, None, [enc, dec], torch.device('cpu'), 'if', inf['min_z'],
                                              inf['cube_bounds'], inf['color_mode'], case['time_idx'], None, **_kw(case, inf))
    assert np.array_equal(res['pcl_abstract'][:, :3], g['pcl_abstract'][:, :3])
    assert np.abs(res['implicit_output'] - g['implicit_output']).max() < 1e-4
    assert res['output_solid'].shape[0] == int(g['n_solid'][0]) and res['output_air'].shape == (8640 - res['output_solid'].shape[0], 5)


def test_twin_nested_level_rule_for_repeated_picks(twin):
    """The rule the HIP kernel is tested against (tests/test_gpu_fps_pruned.py): distinct positions first, ascending, the
    last one repeated behind them."""
    orig = torch.tensor([2, 5, 7, 11, 13, 20, 21, 40], dtype=torch.int32)
    order = torch.tensor([13, 2, 13, 40, 2, 2, 7, 13], dtype=torch.int32)
    pos, nxt = pk.ops.nested_fps_level(order, orig, 6)
    assert pos.tolist() == [0, 4, 7, 7, 7, 7] and nxt.tolist() == [2, 13, 40, 40, 40, 40]
    pos, nxt = pk.ops.nested_fps_level(order, orig, 8)
    assert pos.tolist() == [0, 2, 4, 7, 7, 7, 7, 7] and nxt.tolist() == [2, 7, 13, 40, 40, 40, 40, 40]
