"""Per-clip evaluation loop and its on-disk contract.

Interface mirror of the body of the reference's eval/test.py:test (:31-135): for one data-loader batch (one clip),
every output frame is decoded by ``inference.perform_inference`` with the reference's arguments and the results are
collected as the list the downstream visualisation reads (utils/utils.py:400-479):

    pcl_all[time_idx] = (pcl_input (N,8), pcl_abstract (M,3+E), output_solid (S,4+G), pcl_target_frame (T,9-11),
                         output_air (A,5))                                  [+ (pcl_input_sem, points_query) with save_gt]

written as ``<log_dir>/test_<tag>/pcl_io_s<step>.p`` with ``pickle.dump`` (utils/logvis.py:222-234), next to
``metadata_s<step>.p`` = (meta_data, cam_RT, cam_K).  The published eval/test.py reads ``args.save_gt``, which no
parser defines; here it is an explicit argument, default False.

One difference that does not change results: the reference re-encodes the same input cloud for every output frame
(:67-86); the encode is deterministic, so it is done once per clip and shared (``reuse_encode=False`` restores the
per-frame encode; it is also what happens, automatically, for an encoder built with fps_random_start=True).
"""
import os
import pickle

import numpy as np
import torch

from . import inference


def evaluate_clip(batch, networks, device, args, data_kind, logger=None, save_gt=False, reuse_encode=True):
    """batch: dict with 'pcl_input' (1,N,8), 'pcl_input_sem' (1,N,1-3), 'pcl_target' list of (1,T,9-11) tensors and
    batch['meta_data']['pcl_target_size'] (list of (1,) tensors), as the reference's test data loader yields them
    (data/data_greater.py:593-606, data/data_carla.py:651-661).  args: namespace with the test_args fields used
    below (args.py:311-410).  Returns pcl_all (list over output frames of tuples of numpy arrays)."""
    # One encode per clip is only equivalent to the reference's encode per output frame when the encode is
    # deterministic: a network built with fps_random_start=True (the constructor default; the reference's test path
    # builds its networks with False, eval/inference.py:59) draws a new FPS start per call, so it is re-encoded per frame.
    if reuse_encode and any(getattr(m, 'fps_random_start', False) for m in networks[0].modules()):
        reuse_encode = False
    pcl_input = batch['pcl_input']
    pcl_input_numpy = pcl_input[0].detach().cpu().numpy()
    pcl_input_sem_numpy = batch['pcl_input_sem'][0].detach().cpu().numpy()
    sem_inference = pcl_input_sem_numpy if args.track_mode != 'none' else None
    pcl_target = batch['pcl_target']
    sizes = batch['meta_data']['pcl_target_size']
    pcl_all = []
    encoded = None
    for time_idx in range(len(pcl_target)):
        frame = pcl_target[time_idx][0].detach().cpu().numpy()
        frame = frame[:int(sizes[time_idx].item() if torch.is_tensor(sizes[time_idx]) else sizes[time_idx])]
        res = inference.perform_inference(
            pcl_input.clone(), sem_inference, frame if save_gt else None, networks, device, 'if', args.min_z,
            args.cr_cube_bounds, args.color_mode, time_idx, logger, sample_implicit=args.sample_implicit,
            num_sample=args.num_sample, point_sample_mode=args.point_sample_mode, batch_size=args.implicit_batch_size,
            predict_segmentation=args.segmentation_lw > 0.0, track_mode=args.track_mode,
            point_occupancy_radius=args.point_occupancy_radius, semantic_classes=args.semantic_classes,
            density_threshold=args.density_threshold, data_kind=data_kind, cube_mode=args.cube_mode, compress_air=True,
            encoded=encoded if reuse_encode else None, return_encoded=reuse_encode)
        if reuse_encode and args.track_mode in ('none', 'one'):
            encoded = res.pop('_encoded')
        else:
            res.pop('_encoded', None)
        item = (pcl_input_numpy, res['pcl_abstract'], res['output_solid'], frame, res['output_air'])
        if save_gt:
            item = item + (pcl_input_sem_numpy, res['points_query'])
        pcl_all.append(item)
    return pcl_all


def store_clip(pcl_all, log_dir, test_tag, cur_step, meta=None):
    """Writes pcl_io_s{step}.p (and metadata_s{step}.p when `meta` = (meta_data, cam_RT, cam_K) is given) under
    <log_dir>/test_<tag>/ exactly as eval/test.py:120-135 does through logvis.save_pickle.  Returns the path."""
    folder = os.path.join(log_dir, 'test_' + test_tag)
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, 'pcl_io_s%d.p' % cur_step)
    with open(path, 'wb') as f:
        pickle.dump(pcl_all, f)
    if meta is not None:
        with open(os.path.join(folder, 'metadata_s%d.p' % cur_step), 'wb') as f:
            pickle.dump(meta, f)
    return path


def load_clip(path):
    """Reads a pcl_io_s{step}.p back; checks the tuple contract."""
    with open(path, 'rb') as f:
        pcl_all = pickle.load(f)
    assert isinstance(pcl_all, list)
    for item in pcl_all:
        assert isinstance(item, tuple) and len(item) in (5, 7)
        assert all(isinstance(a, np.ndarray) for a in item)
    return pcl_all
