"""Inference driver on the HIP library.

Interface mirror of the reference's eval/inference.py: ``load_models`` (:23-80) and
``perform_inference`` (:83-325, same positional/keyword arguments and result dict).
Differences that do not change results: the query grid is uploaded once and every
mini-batch, the post-ops (sigmoid / clamp, :218-243) and the concatenation stay on the
device; one device-to-host copy happens at the end instead of one per batch (:206,245).
The optional ground-truth 1-NN labelling branch (:270-276, sklearn KDTree on targets)
needs dataset targets and is out of the timed path; it is not provided here.
"""
import os

import numpy as np
import torch

from . import geometry
from . import implicit
from . import model
from . import ops


def get_track_idx(color_mode):
    """Channel of mark_track in the implicit output (utils/utils.py:204-224)."""
    table = {'rgb': 4, 'rgb_nosigmoid': 4, 'hsv': 15, 'bins': 10}
    if color_mode not in table:
        raise ValueError()
    return table[color_mode]


def load_models(checkpoint_path, device, epoch=-1, logger=None):
    """Builds [pcl_net, implicit_net] from a reference checkpoint (keys args, dset_args,
    pcl_args, implicit_args, pcl_net, implicit_net, epoch -- train.py:339-350)."""
    print_fn = logger.info if logger is not None else print
    assert os.path.exists(checkpoint_path)
    if os.path.isdir(checkpoint_path):
        checkpoint_path = os.path.join(checkpoint_path, f'model_{epoch}.pth' if epoch >= 0 else 'checkpoint.pth')
    print_fn('Loading weights from: ' + checkpoint_path)
    ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    train_args, dset_args = ckpt['args'], ckpt['dset_args']
    pcl_args, implicit_args = dict(ckpt['pcl_args']), dict(ckpt['implicit_args'])
    pcl_args['fps_random_start'] = False          # deterministic at test time (:59)
    dec_sd = {(('pt_blocks.0.' + k[len('pt_block.'):]) if k.startswith('pt_block.') else k): v
              for k, v in ckpt['implicit_net'].items()}   # legacy key rename (:62-63)
    pcl_net = model.PointCompletionNetV3(**pcl_args).to(device)
    pcl_net.load_state_dict(ckpt['pcl_net'])
    implicit_net = implicit.LocalPclResnetFC(**implicit_args).to(device)
    implicit_net.load_state_dict(dec_sd)
    epoch = ckpt['epoch']
    print_fn('=> Loaded epoch (1-based): ' + str(epoch + 1))
    return ([pcl_net, implicit_net], train_args, dset_args, pcl_args, implicit_args, epoch)


def squash_codes(d_out, color_mode, predict_segmentation, track_mode, semantic_classes):
    """Per-channel post-op codes for occ4d_squash_f32 (0 identity, 1 sigmoid, 2 clamp[0,1]),
    equivalent to the in-place sequence of eval/inference.py:218-243 (a later sigmoid on a
    channel composes with an earlier op exactly as the reference's sequential writes do
    only when ranges do not overlap; overlapping ranges are rejected)."""
    codes = [0] * d_out
    applied = [0] * d_out

    def put(lo, hi, code):
        for c in range(lo, hi):
            c = c % d_out
            codes[c] = code
            applied[c] += 1
    put(0, 1, 1)
    if color_mode == 'rgb':
        put(1, 4, 1)
    elif color_mode == 'rgb_nosigmoid':
        put(1, 4, 2)
    elif color_mode == 'hsv':
        put(1, 13, 1)
        put(13, 15, 2)
    elif color_mode == 'bins':
        put(1, 10, 1)
    if predict_segmentation:
        put(d_out - semantic_classes, d_out, 1)
    if track_mode != 'none':
        ti = get_track_idx(color_mode)
        put(ti, ti + 1, 1)
    assert max(applied) <= 1, 'overlapping post-op channel ranges'
    return codes


def perform_inference(pcl_input, pcl_input_sem, pcl_target_frame, networks, device, task, min_z,
                      cube_bounds, color_mode, time_idx, logger,
                      sample_implicit=True, num_sample=16384, point_sample_mode='random',
                      batch_size=1024, predict_segmentation=False, track_mode='none',
                      point_occupancy_radius=0.2, semantic_classes=13,
                      density_threshold=0.5, data_kind='', cube_mode=4, compress_air=False):
    """One encode of the input point-cloud video + decode of all query points of one output
    frame.  Returns dict(output_solid, output_air, pcl_abstract, features_global,
    implicit_output, points_query) of float32 numpy arrays."""
    assert task == 'if'
    assert sample_implicit
    if track_mode == 'all':
        raise NotImplementedError("track_mode 'all' (one rerun per instance id, eval/inference.py:144-161) "
                                  "is not part of the benchmarked path")
    if pcl_target_frame is not None:
        raise NotImplementedError('ground-truth 1-NN labelling (eval/inference.py:270-276) is out of scope')
    pcl_net, implicit_net = networks
    if isinstance(pcl_input, np.ndarray):
        pcl_input = torch.from_numpy(pcl_input).unsqueeze(0)
    pcl_input = pcl_input.to(device)
    points_query = geometry.sample_implicit_points_blind_numpy(
        num_sample, min_z, cube_bounds, time_idx, data_kind, cube_mode, point_sample_mode)
    with torch.no_grad():
        res = infer_device(pcl_input, torch.from_numpy(points_query).to(device), pcl_net, implicit_net,
                           batch_size, color_mode, predict_segmentation, track_mode, semantic_classes)
        implicit_output = res['implicit_output'].cpu().numpy()
        pcl_abstract = res['pcl_abstract'].cpu().numpy() if res['pcl_abstract'] is not None else None
        features_global = res['features_global'].cpu().numpy()
    points_io = np.concatenate([points_query, implicit_output], axis=-1)
    solid = points_io[points_io[..., 4] >= density_threshold]
    air = points_io[points_io[..., 4] < density_threshold]
    if compress_air:
        air_segm = air[..., -semantic_classes:].argmax(axis=-1)
        air = np.concatenate([air[..., :3], air[..., 4:5], air_segm[..., None]], axis=-1)
    return dict(output_solid=solid, output_air=air, pcl_abstract=pcl_abstract,
                features_global=features_global, implicit_output=implicit_output, points_query=points_query)


def infer_device(pcl_input, points_query, pcl_net, implicit_net, batch_size, color_mode,
                 predict_segmentation=False, track_mode='none', semantic_classes=13):
    """Device-resident core of perform_inference: encode once, decode every mini-batch, squash.
    All tensors are CUDA; returns CUDA tensors (implicit_output (N,G), pcl_abstract (M,3+E),
    features_global (D))."""
    (pcl_abstract, features_global, _) = pcl_net(pcl_input, False)
    if pcl_abstract is not None:
        pcl_abstract = pcl_abstract.squeeze(0)
    features_global = features_global.squeeze(0)
    n = points_query.shape[0]
    out = torch.empty((n, implicit_net.d_out), dtype=torch.float32, device=points_query.device)
    decode_batches(implicit_net, points_query, 0, n, batch_size, pcl_abstract, features_global, out)
    ops.squash(out, squash_codes(implicit_net.d_out, color_mode, predict_segmentation, track_mode,
                                 semantic_classes))
    return dict(implicit_output=out, pcl_abstract=pcl_abstract, features_global=features_global)


DECODE_STREAMS = int(os.environ.get('OCC4D_DECODE_STREAMS', '2'))   # 1 = the reference's strictly serial loop


def decode_batches(implicit_net, points_query, lo, hi, batch_size, pcl_abstract, features_global, out, out_offset=0):
    """Runs implicit_net on points_query[lo:hi] in mini-batches of `batch_size` (the reference's
    loop, eval/inference.py:204-246) and writes rows into out[out_offset:].  Mini-batches are
    independent, so consecutive ones alternate between DECODE_STREAMS HIP streams: a 32768-query
    batch is exactly one wave of 256 workgroups for the row-tiled kernels, and the next batch's
    kernels fill the tail of the previous one's instead of waiting behind it.  The first batch runs
    on the caller's stream so the per-scene tables are built (and cached) before the side streams
    start."""
    main = torch.cuda.current_stream()
    starts = list(range(lo, hi, batch_size))
    side = [torch.cuda.Stream() for _ in range(DECODE_STREAMS)] if DECODE_STREAMS > 1 and len(starts) > 2 else []
    for bi, b in enumerate(starts):
        e = min(hi, b + batch_size)
        if bi == 0 or not side:
            (o, _) = implicit_net(points_query[b:e], pcl_abstract, features_global, None)
            out[out_offset + b - lo:out_offset + e - lo] = o
            if bi == 0:
                for st in side:
                    st.wait_stream(main)
            continue
        st = side[bi % len(side)]
        with torch.cuda.stream(st):
            (o, _) = implicit_net(points_query[b:e], pcl_abstract, features_global, None)
            out[out_offset + b - lo:out_offset + e - lo] = o
    for st in side:
        main.wait_stream(st)
    return out
