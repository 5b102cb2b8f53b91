"""Inference driver on the HIP library.

Interface mirror of the reference's eval/inference.py: ``load_models`` (:23-80) and
``perform_inference`` (:83-325, same positional/keyword arguments and result dict).
Differences that do not change results: the query grid is uploaded once and every
mini-batch, the post-ops (sigmoid / clamp, :218-243) and the concatenation stay on the
device; one device-to-host copy happens at the end instead of one per batch (:206,245).
The optional ground-truth 1-NN labelling branch (:270-276, sklearn KDTree on targets)
is provided on the streaming k = 1 kernel; track_mode 'all' reruns the path once per instance id.
"""
import os

import numpy as np
import torch

from . import geometry
from . import implicit
from . import kernels
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


PINNED_HOST_IO = os.environ.get('OCC4D_PINNED_HOST_IO', '1') == '1'

# Side streams live as long as the process, one set per device.  torch's caching allocator keeps a block pool PER STREAM:
# with fresh `torch.cuda.Stream()` objects in every call the decode workspaces (465 MB at the BASELINE grid) and the
# result buffers were allocated again by every perform_inference call until torch's 32-stream pool had gone round
# (profiles/host_boundary_probe.py: +8 ms per call).
_STREAMS = {}


def side_streams(device, role, count=1):
    """`count` persistent HIP streams of `device` for `role` ('decode' / 'copy')."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, role)
    have = _STREAMS.setdefault(key, [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device=torch.device('cuda', index)))
    return have[:count]


class _HostCopies:
    """Device -> host result copies of perform_inference: every array goes into a page-locked buffer from torch's caching
    host allocator with a non-blocking copy on a side stream that waits for the producing stream, so that the copies
    overlap each other and the remaining device work, and the host blocks ONCE, at the end.  (The reference moves each
    array with a blocking `.cpu()` through pageable memory: eval/inference.py:218-246; at 0.53 M queries that was 24 ms
    of a 144 ms call.)  The numpy arrays handed out own their buffers (views of the pinned tensors, kept alive by numpy's
    base reference); the allocator reuses a block only after the caller has dropped the array."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.on = PINNED_HOST_IO and self.device.type == 'cuda'
        self.stream = side_streams(self.device, 'copy')[0] if self.on else None
        self.pending = []

    def fetch(self, t, dtype=None):
        """Schedules the copy of device tensor `t` (optionally converted to `dtype` on the device) and returns a handle;
        `result(handle)` after `wait()` gives the numpy array."""
        if t is None:
            return None
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if not self.on:
            return t.cpu().numpy()
        t = t.contiguous()
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.stream.wait_stream(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(self.stream):
            host.copy_(t, non_blocking=True)
        t.record_stream(self.stream)
        self.pending.append(host)
        return host

    def wait(self):
        if self.on:
            self.stream.synchronize()

    @staticmethod
    def result(h):
        return h.numpy() if torch.is_tensor(h) else h


def perform_inference(pcl_input, pcl_input_sem, pcl_target_frame, networks, device, task, min_z,
                      cube_bounds, color_mode, time_idx, logger,
                      sample_implicit=True, num_sample=16384, point_sample_mode='random',
                      batch_size=1024, predict_segmentation=False, track_mode='none',
                      point_occupancy_radius=0.2, semantic_classes=13,
                      density_threshold=0.5, data_kind='', cube_mode=4, compress_air=False,
                      encoded=None, return_encoded=False, neighbour_lists=None):
    """One encode of the input point-cloud video + decode of all query points of one output
    frame.  Returns dict(output_solid, output_air, pcl_abstract, features_global,
    implicit_output, points_query) of float32 numpy arrays.
    Extensions (keyword only, default = the reference's behaviour): `encoded` = the (pcl_abstract, features_global)
    device tensors of an earlier call on the same input cloud (the reference's eval loop re-encodes the clip for every
    output frame, eval/test.py:67-86); `return_encoded` adds them to the result as '_encoded'; `neighbour_lists` =
    (knn_local (N_q, 8), knn_cross (N_q, 14)) integer arrays: the decoder's neighbour lists of a particular run of the
    reference for these queries (LocalPclResnetFC.forward's extension; either entry may be None)."""
    assert task == 'if'
    assert sample_implicit
    output_track_idx = get_track_idx(color_mode)
    input_inst_idx = 0 if data_kind == 'greater' else 1
    pcl_net, implicit_net = networks
    if isinstance(pcl_input, np.ndarray):
        pcl_input = torch.from_numpy(pcl_input).unsqueeze(0)
    pcl_input = pcl_input.to(device)

    # One rerun per tracked instance (track_mode 'all', :144-161), otherwise a single run.
    if track_mode in ('none', 'one'):
        track_instance_ids = [-1]
    else:
        assert data_kind == 'greater'
        assert pcl_input_sem.shape[-1] == 1
        if isinstance(pcl_input_sem, np.ndarray):
            sem_numpy = pcl_input_sem
            pcl_input_sem = torch.from_numpy(pcl_input_sem).unsqueeze(0).to(device)
        else:
            sem_numpy = pcl_input_sem[0].detach().cpu().numpy()
            pcl_input_sem = pcl_input_sem.to(device)
        ids, counts = np.unique(sem_numpy, return_counts=True)
        track_instance_ids = [int(i) for i, c in zip(ids, counts) if i >= 0 and c >= 16]

    queries_dev = geometry.sample_implicit_points_blind_device(
        num_sample, min_z, cube_bounds, time_idx, data_kind, cube_mode, point_sample_mode, device)
    copies = _HostCopies(device)
    single_run = len(track_instance_ids) == 1 and track_instance_ids[0] == -1
    points_query = copies.fetch(queries_dev)              # (under the encode / decode that follows)
    all_abstract, all_global, all_output = [], [], []
    with torch.no_grad():
        for inst_id in track_instance_ids:
            if inst_id >= 0:                  # mark the instance to follow in the input cloud (:190-193)
                pcl_input[..., -1] = (pcl_input_sem[..., input_inst_idx] == inst_id)
            res = infer_device(pcl_input, queries_dev, pcl_net, implicit_net, batch_size, color_mode,
                               predict_segmentation, track_mode, semantic_classes,
                               encoded=encoded if inst_id < 0 else None, neighbour_lists=neighbour_lists)
            output_dev = res['implicit_output']
            all_output.append(copies.fetch(output_dev))
            all_abstract.append(copies.fetch(res['pcl_abstract']))
            all_global.append(copies.fetch(res['features_global']))
        if not single_run:                    # the merge of the per-instance reruns is host arithmetic
            copies.wait()
            (pcl_abstract, features_global, implicit_output) = multi_track_merge(
                track_instance_ids, [copies.result(h) for h in all_abstract], [copies.result(h) for h in all_global],
                [copies.result(h) for h in all_output], output_track_idx)

        gt_available = pcl_target_frame is not None
        if gt_available:                      # nearest ground-truth point of every query (:270-276)
            target_labels, nn_indices = get_1nn_label(queries_dev[:, :3], pcl_target_frame, point_occupancy_radius,
                                                      device)
            points_nngt = np.concatenate([target_labels[:, None], pcl_target_frame[nn_indices]], axis=-1)

        # density-threshold split + compress_air on the device (:279-305): order-preserving compaction
        if not single_run:                    # merged on the host; one upload
            output_dev = torch.from_numpy(implicit_output).to(device)
        solid, air = ops.split_solid_air(queries_dev, output_dev, density_threshold, compress_air, semantic_classes)
        # (the reference's concatenate with the int64 argmax promotes the compressed air rows to float64: converted on
        # the device, not by a host pass over the array)
        solid_h = copies.fetch(solid)
        air_h = copies.fetch(air, torch.float64 if compress_air else None)
        copies.wait()
        solid, air = copies.result(solid_h), copies.result(air_h)
        points_query = copies.result(points_query)
        if single_run:
            (pcl_abstract, features_global, implicit_output) = (copies.result(all_abstract[0]),
                                                                 copies.result(all_global[0]), copies.result(all_output[0]))
    ops.check_pending()                      # cooperative-FPS status words (everything above has completed)
    result = dict(output_solid=solid, output_air=air, pcl_abstract=pcl_abstract,
                  features_global=features_global, implicit_output=implicit_output, points_query=points_query)
    if return_encoded:
        result['_encoded'] = (res['pcl_abstract'], res['features_global'])
    if gt_available:
        solid_mask = implicit_output[..., 0] >= density_threshold
        gt_solid, gt_air = points_nngt[solid_mask], points_nngt[~solid_mask]
        if compress_air:
            gt_air = np.concatenate([gt_air[..., :1], gt_air[..., 4:5]], axis=-1)
        result['gt_solid'], result['gt_air'] = gt_solid, gt_air
    return result


def get_1nn_label(points_query_xyz, pcl_target_frame, thresh, device):
    """Pseudo label of every query from its nearest target point (utils/geometry.py:444-455, an sklearn
    KDTree there): label = (distance < thresh), plus the neighbour's index.  Streaming k = 1 kernel."""
    target_xyz = torch.from_numpy(np.ascontiguousarray(pcl_target_frame[..., :3], dtype=np.float32)).to(device)
    idx, dist = ops.knn(points_query_xyz, target_xyz, 1, metric=1, return_dist=True)
    labels = (dist[:, 0] < thresh).cpu().numpy() * 1
    return labels, idx[:, 0].cpu().numpy().astype(np.int64)


def multi_track_merge(track_instance_ids, pcl_abstract, features_global, implicit_output, output_track_idx):
    """Merge the per-instance reruns (utils/utils.py:343-397): features and outputs are averaged, the
    mark_track channel becomes the id of the most confident run (>= 0.5), else -1."""
    assert len(pcl_abstract) == len(features_global) == len(implicit_output)
    if len(pcl_abstract) == 1 and track_instance_ids[0] == -1:
        return (pcl_abstract[0], features_global[0], implicit_output[0])
    merged_abstract = np.mean(pcl_abstract, axis=0) if pcl_abstract[0] is not None else None
    merged_global = np.mean(features_global, axis=0)
    merged_output = np.mean(implicit_output, axis=0)
    winner = -np.ones_like(merged_output[..., 0])
    best = np.zeros_like(merged_output[..., 0])
    for inst_id, out in zip(track_instance_ids, implicit_output):
        score = out[..., output_track_idx]
        winner[np.logical_and(score >= 0.5, score >= best)] = inst_id
        best = np.maximum(score, best)
    merged_output[..., output_track_idx] = winner
    return (merged_abstract, merged_global, merged_output)


def infer_device(pcl_input, points_query, pcl_net, implicit_net, batch_size, color_mode,
                 predict_segmentation=False, track_mode='none', semantic_classes=13, encoded=None,
                 neighbour_lists=None):
    """Device-resident core of perform_inference: encode once, decode every mini-batch, squash.
    All tensors are CUDA; returns CUDA tensors (implicit_output (N,G), pcl_abstract (M,3+E),
    features_global (D))."""
    if encoded is not None:
        (pcl_abstract, features_global) = encoded
    else:
        (pcl_abstract, features_global, _) = pcl_net(pcl_input, False)
        if pcl_abstract is not None:
            pcl_abstract = pcl_abstract.squeeze(0)
        features_global = features_global.squeeze(0)
    n = points_query.shape[0]
    out = torch.empty((n, implicit_net.d_out), dtype=torch.float32, device=points_query.device)
    lists = None
    if neighbour_lists is not None:
        lists = tuple(None if a is None else torch.as_tensor(a).to(points_query.device) for a in neighbour_lists)
        assert len(lists) == 2 and all(a is None or a.shape[0] == n for a in lists)
    decode_batches(implicit_net, points_query, 0, n, batch_size, pcl_abstract, features_global, out, lists=lists)
    ops.squash(out, squash_codes(implicit_net.d_out, color_mode, predict_segmentation, track_mode,
                                 semantic_classes))
    return dict(implicit_output=out, pcl_abstract=pcl_abstract, features_global=features_global)


# decode streams: kernels.Selection.decode_streams (default 2, OCC4D_DECODE_STREAMS; 1 = the reference's strictly serial loop)
# The fused attention kernel packs 9 queries per workgroup and one workgroup occupies a CU (129 KB LDS): a mini-batch
# of 9 * 256 * r queries is exactly r full rounds of the 256 CUs.  The caller's batch_size (a memory knob in the
# reference) is rounded DOWN to such a multiple (32768 -> 32256: 14 full rounds instead of 14.2 -> 15); every query
# is still decoded, results do not depend on the split (tests).  0 disables.
DECODE_ALIGN = int(os.environ.get('OCC4D_DECODE_ALIGN', str(9 * 256)))


def decode_chunk(batch_size):
    if DECODE_ALIGN > 0 and batch_size >= DECODE_ALIGN:
        return batch_size - batch_size % DECODE_ALIGN
    return batch_size


def _decode_into(implicit_net, q, pcl_abstract, features_global, out_rows, lists=None):
    """One mini-batch: the network's raw outputs written into `out_rows`.  The library-backed decoder writes them in
    place and skips the penultimate activation perform_inference discards (eval/inference.py:211); any other module
    with the reference's forward signature is called as the reference calls it."""
    direct = getattr(implicit_net, 'forward_output_only', None)
    kw = {} if lists is None else dict(knn_local=lists[0], knn_cross=lists[1])
    if direct is not None and getattr(implicit_net, 'num_local_features', 0) > 0 and q.dim() == 2 and q.shape[0] > 0:
        direct(q, pcl_abstract, features_global, None, out_rows, **kw)
        return
    (o, _) = implicit_net(q, pcl_abstract, features_global, None, **kw)
    out_rows.copy_(o)


def decode_batches(implicit_net, points_query, lo, hi, batch_size, pcl_abstract, features_global, out, out_offset=0,
                   lists=None):
    """Runs implicit_net on points_query[lo:hi] in mini-batches of `batch_size` (the reference's
    loop, eval/inference.py:204-246) and writes rows into out[out_offset:].  Mini-batches are
    independent, so consecutive ones alternate between `decode_streams` (kernels.Selection) HIP streams: a 32768-query
    batch is exactly one wave of 256 workgroups for the row-tiled kernels, and the next batch's
    kernels fill the tail of the previous one's instead of waiting behind it.  The first batch runs
    on the caller's stream so the per-scene tables are built (and cached) before the side streams
    start."""
    batch_size = decode_chunk(batch_size)
    starts = list(range(lo, hi, batch_size))
    n_streams = kernels.scope().decode_streams if points_query.is_cuda else 1      # (host tensors: the explicit CPU twin)
    main = torch.cuda.current_stream() if points_query.is_cuda else None
    side = side_streams(points_query.device, 'decode', n_streams) if n_streams > 1 and len(starts) > 2 else []
    def rows(b, e):           # the caller's neighbour lists of these queries (rows are indexed like points_query)
        return None if lists is None else tuple(None if a is None else a[b:e] for a in lists)

    for bi, b in enumerate(starts):
        e = min(hi, b + batch_size)
        if bi == 0 or not side:
            _decode_into(implicit_net, points_query[b:e], pcl_abstract, features_global, out[out_offset + b - lo:out_offset + e - lo],
                         rows(b, e))
            if bi == 0:
                for st in side:
                    st.wait_stream(main)
            continue
        st = side[bi % len(side)]
        with torch.cuda.stream(st):
            _decode_into(implicit_net, points_query[b:e], pcl_abstract, features_global, out[out_offset + b - lo:out_offset + e - lo],
                         rows(b, e))
    for st in side:
        main.wait_stream(st)
    return out
