"""The checkpoint contract (the state_dict keys and the checkpoint dict are part of the drop-in
boundary).  Fixtures: tests/golden/g17_ckpt_greater_small/{checkpoint.pth, model_3.pth}, written by oracle/gen_golden.py
from the REFERENCE's modules in the reference's layout (train.py:339-350), and g17_ckpt_*.npz = what the reference's own
load_models -> perform_inference (eval/inference.py:23-80, 83-325) returned for them."""
import argparse
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
import occlusions4d_amd as pk
from conftest import GOLDEN, load_golden
from oracle import path as op

CKPT = os.path.join(GOLDEN, gc.CKPT_DIR)


@pytest.mark.parametrize('case', gc.CKPT_CASES, ids=lambda c: c['name'])
def test_load_models_contract_on_the_host(case, capsys):
    """Return arity / types / values of load_models as eval/inference.py:23-80 documents them; directory + epoch file
    naming (:39-41); fps_random_start forced off (:59); the legacy `pt_block.` prefix renamed (:62-63,
    utils/utils.py:127-135); every parameter bit-identical to the checkpoint's tensor.  (Modules are built and loaded on
    the CPU: no kernel runs.)"""
    g = load_golden('g17_ckpt_' + case['name'])
    ret = pk.inference.load_models(CKPT, torch.device('cpu'), epoch=case['epoch_arg'])
    assert isinstance(ret, tuple) and len(ret) == 6
    (networks, train_args, dset_args, pcl_args, implicit_args, epoch) = ret
    assert isinstance(networks, list) and len(networks) == 2
    assert isinstance(networks[0], pk.model.PointCompletionNetV3) and isinstance(networks[1], pk.implicit.LocalPclResnetFC)
    assert isinstance(train_args, argparse.Namespace) and train_args.name == 'g17_' + case['name']
    assert isinstance(dset_args, dict) and dset_args['n_points'] == gc.CKPT_INFER['n']
    pa, ia, _ = gc.ckpt_model_args(case)
    assert pa['fps_random_start'] is True and pcl_args == dict(pa, fps_random_start=False)      # (:59)
    assert implicit_args == ia and epoch == case['epoch']
    said = capsys.readouterr().out
    assert case['file'] in said and '=> Loaded epoch (1-based): %d' % (case['epoch'] + 1) in said
    # the same module tree as the reference's: its state_dict keys, name for name
    assert sorted(networks[0].state_dict().keys()) == list(g['encoder_keys'])
    assert sorted(networks[1].state_dict().keys()) == list(g['decoder_keys'])
    raw = torch.load(os.path.join(CKPT, case['file']), map_location='cpu', weights_only=False)
    assert set(raw) == {'optimizer', 'lr_scheduler', 'scaler', 'epoch', 'args', 'pcl_args', 'dset_args', 'implicit_args',
                        'pcl_net', 'implicit_net'}                                                 # train.py:339-350
    legacy_keys = [k for k in raw['implicit_net'] if k.startswith('pt_block.')]
    assert bool(legacy_keys) == case['legacy']
    dsd = networks[1].state_dict()
    for k, v in raw['implicit_net'].items():
        k2 = 'pt_blocks.0.' + k[len('pt_block.'):] if k.startswith('pt_block.') else k
        assert torch.equal(dsd[k2], v), k
    for k, v in raw['pcl_net'].items():
        assert torch.equal(networks[0].state_dict()[k], v), k
    # an explicit file path works as the directory form does
    again = pk.inference.load_models(os.path.join(CKPT, case['file']), torch.device('cpu'))
    assert again[5] == epoch and again[3] == pcl_args


def test_load_models_rejects_a_missing_path():
    with pytest.raises(AssertionError):
        pk.inference.load_models(os.path.join(CKPT, 'no_such_dir'), torch.device('cpu'))


@pytest.mark.parametrize('case', gc.CKPT_CASES, ids=lambda c: c['name'])
def test_oracle_reproduces_the_reference_run_of_the_checkpoint(case):
    """The CPU oracle on the checkpoint's tensors (small widths: 4 / 8 / 16 / 32 encoder, 48-wide decoder, one or two cross
    layers) against the reference's outputs for that checkpoint: pins the oracle away from the published widths too."""
    g = load_golden('g17_ckpt_' + case['name'])
    raw = torch.load(os.path.join(CKPT, case['file']), map_location='cpu', weights_only=False)
    dsd = {('pt_blocks.0.' + k[len('pt_block.'):] if k.startswith('pt_block.') else k): v
           for k, v in raw['implicit_net'].items()}
    pa, ia, inf = gc.ckpt_model_args(case)
    pcl = pk.configs.synthetic_pcl(gc.CKPT_INFER['kind'], gc.CKPT_INFER['n'], gc.CKPT_INFER['video_len'], gc.CKPT_INFER['seed'])
    res = op.perform_inference(pcl.clone(), raw['pcl_net'], dict(pa, fps_random_start=False), dsd, ia, inf['min_z'],
                               inf['cube_bounds'], inf['color_mode'], gc.CKPT_INFER['time_idx'],
                               num_sample=gc.CKPT_INFER['num_sample'], point_sample_mode='grid',
                               batch_size=gc.CKPT_INFER['batch_size'], predict_segmentation=False, track_mode='none',
                               semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4,
                               compress_air=True)
    assert np.array_equal(res['pcl_abstract'][:, :3], g['pcl_abstract'][:, :3])
    assert np.abs(res['pcl_abstract'] - g['pcl_abstract']).max() < 1e-5
    assert np.abs(res['implicit_output'] - g['implicit_output']).max() < 1e-5
    assert res['output_solid'].shape[0] == int(g['n_solid'][0])


@pytest.mark.gpu
@pytest.mark.parametrize('case', gc.CKPT_CASES, ids=lambda c: c['name'])
def test_checkpoint_to_inference_on_the_gpu(case):
    """load_models(path, cuda) -> networks -> perform_inference: the reference's own outputs for this checkpoint at 1e-4
    (eval/test.py's sequence: :44 load_models, :75 perform_inference)."""
    assert torch.cuda.is_available()
    g = load_golden('g17_ckpt_' + case['name'])
    device = torch.device('cuda')
    (networks, _, _, pcl_args, implicit_args, epoch) = pk.inference.load_models(CKPT, device, epoch=case['epoch_arg'])
    assert epoch == case['epoch'] and pcl_args['fps_random_start'] is False
    assert all(p.is_cuda for net in networks for p in net.parameters())
    for net in networks:
        net.eval()
    _, _, inf = gc.ckpt_model_args(case)
    pcl = pk.configs.synthetic_pcl(gc.CKPT_INFER['kind'], gc.CKPT_INFER['n'], gc.CKPT_INFER['video_len'], gc.CKPT_INFER['seed'])
    res = pk.inference.perform_inference(
        pcl.clone(), None, None, networks, device, 'if', inf['min_z'], inf['cube_bounds'], inf['color_mode'],
        gc.CKPT_INFER['time_idx'], None, sample_implicit=True, num_sample=gc.CKPT_INFER['num_sample'],
        point_sample_mode='grid', batch_size=gc.CKPT_INFER['batch_size'], predict_segmentation=False, track_mode='none',
        semantic_classes=13, density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=4, compress_air=True)
    assert np.array_equal(res['pcl_abstract'][:, :3], g['pcl_abstract'][:, :3])
    assert np.abs(res['pcl_abstract'] - g['pcl_abstract']).max() < 1e-4
    assert np.abs(res['features_global'] - g['features_global']).max() < 1e-4
    assert res['implicit_output'].shape == g['implicit_output'].shape
    assert np.abs(res['implicit_output'] - g['implicit_output']).max() < 1e-4
    near = int((np.abs(g['implicit_output'][:, 0] - 0.5) < 1e-4).sum())
    assert abs(res['output_solid'].shape[0] - int(g['n_solid'][0])) <= near
