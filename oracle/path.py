"""CPU restatement of the reference hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py): plain PyTorch-CPU, fp32, written
from scratch as free functions over a ``state_dict`` (parameter names are the
reference's).  Every function cites the reference lines it
follows (paths relative to /root/reference).  It is checked against golden
vectors produced by the real reference (tests/golden, oracle/gen_golden.py).
"""

import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import cluster

# Tie handling.  The reference orders equal distances with an UNSTABLE sort
# (argsort(), model/point_transformer_layer.py:97) / topk (utils/geometry.py:484): which of
# two equidistant points it keeps at the k-th rank is implementation-defined (and differs
# between its CPU and CUDA runs).  CARLA's two-level abstract cloud contains every coarse
# point twice (same xyz, different features, model/model.py:202-228), so such ties are
# systematic there.  False (default): restate the reference literally (what the golden
# vectors pin).  True: the product's documented rule -- lowest index first.
STABLE_TIES = False


class stable_ties:
    """with stable_ties(): ... -> neighbour lists use the lowest-index-first rule."""

    def __enter__(self):
        global STABLE_TIES
        self._old, STABLE_TIES = STABLE_TIES, True

    def __exit__(self, *exc):
        global STABLE_TIES
        STABLE_TIES = self._old


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------

# ---- ReLU audit (tests only).  Gradients of two correct fp32 implementations can differ by O(1 / rows) when a ReLU
# input sits within rounding of zero and lands on different sides (its mask then differs, which moves a gradient by one
# summand).  relu_kinks() numbers every ReLU call of a forward pass, records the units with |input| < tau, and can
# replay the pass with the decision of chosen units forced either way, so that a test can show: "for SOME assignment
# of the few ambiguous units the gradients agree to the strict tolerance" (tests/test_gpu_training.py).
_RELU_HOOK = None


def _relu(x):
    return F.relu(x) if _RELU_HOOK is None else _RELU_HOOK(x)


@contextlib.contextmanager
def relu_kinks(tau, forced=None):
    """with relu_kinks(tau) as log: ...forward...  ->  log = [(call, flat index, input value), ...] of the ReLU units
    with |input| < tau.  forced = {(call, flat index): bool}: those units pass (True) / block (False) regardless of
    their sign; every other unit behaves as F.relu (same value, same gradient)."""
    global _RELU_HOOK
    log, state = [], dict(n=0)

    def hook(x):
        call = state['n']
        state['n'] += 1
        xd = x.detach()
        near = (xd.abs() < tau).reshape(-1).nonzero().reshape(-1)
        for i in near.tolist():
            log.append((call, i, float(xd.reshape(-1)[i])))
        mask = xd > 0
        if forced:
            mask = mask.clone()
            for (c, i), v in forced.items():
                if c == call:
                    mask.view(-1)[i] = bool(v)
        return x * mask.to(x.dtype)
    old, _RELU_HOOK = _RELU_HOOK, hook
    try:
        yield log
    finally:
        _RELU_HOOK = old


def _lin(sd, name, x):
    return F.linear(x, sd[name + '.weight'], sd.get(name + '.bias'))


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------
# E4 / E5: kNN_torch, index_points   (model/point_transformer_layer.py:16-113)
# ----------------------------------------------------------------------------
def knn_indices(query, dataset, k, chunk=1024):
    """(B,N0,3),(B,N1,3) -> (B,N0,k) int64.  Follows square_distance (:16-30) and
    kNN_torch (:76-99): d = sum_c (q_c - p_c)^2 over the broadcast difference,
    then argsort and keep the first k.  Rows are processed in chunks (same
    arithmetic per row; the reference materialises the full N0 x N1 matrix)."""
    assert query.dim() == 3 and dataset.dim() == 3
    assert query.shape[0] == dataset.shape[0] and query.shape[2] == dataset.shape[2]
    out = []
    for lo in range(0, query.shape[1], chunk):
        q = query[:, lo:lo + chunk]
        d = torch.sum((q[:, :, None] - dataset[:, None]) ** 2, dim=-1)
        out.append(d.argsort(stable=True)[:, :, :k] if STABLE_TIES else d.argsort()[:, :, :k])
    return torch.cat(out, dim=1)


def gather_rows(points, idx):
    """index_points (:102-113): (B,N,C),(B,S,K) -> (B,S,K,C)."""
    B = idx.shape[0]
    flat = idx.reshape(B, -1)
    res = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return res.reshape(*idx.shape, -1)


# ----------------------------------------------------------------------------
# E3: PointTransformerLayer.forward   (model/point_transformer_layer.py:148-183)
# ----------------------------------------------------------------------------
def pt_layer(sd, x, pos, x2=None, pos2=None, num_neighbors=16, idx=None):
    """idx (B,N,K) int64: the neighbour lists of a particular reference run (its unstable argsort decides the order of
    equidistant points); None = searched here as the reference does."""
    if x2 is None:
        x2, pos2 = x, pos
    if idx is None:
        idx = knn_indices(pos, pos2, num_neighbors)             # :167
    nb_xyz = gather_rows(pos2, idx)                             # :168
    q = _lin(sd, 'to_q', x)                                     # :170
    k = gather_rows(_lin(sd, 'to_k', x2), idx)                  # :171
    v = gather_rows(_lin(sd, 'to_v', x2), idx)                  # :172
    rel = pos[:, :, None] - nb_xyz
    pe = _lin(sd, 'pos_mlp.2', _relu(_lin(sd, 'pos_mlp.0', rel)))           # :174
    a = _lin(sd, 'attn_mlp.2', _relu(_lin(sd, 'attn_mlp.0', q[:, :, None] - k + pe)))  # :176
    a = F.softmax(a / np.sqrt(k.shape[-1]), dim=-2)             # :177 (per channel over K)
    return torch.einsum('bijd,bijd->bid', a, v + pe)            # :179


# ----------------------------------------------------------------------------
# E2: PointTransformerBlock.forward   (model/modules.py:45-67)
# ----------------------------------------------------------------------------
def pt_block(sd, x, p, x2=None, p2=None, num_neighbors=16, idx=None):
    assert x.shape[:2] == p.shape[:2]
    if x2 is not None:
        assert x2.shape[:2] == p2.shape[:2]
    y = _lin(sd, 'layer1', x)
    y = pt_layer(_sub(sd, 'layer2.'), y, p, x2, p2, num_neighbors, idx)
    y = _lin(sd, 'layer3', y)
    return x + y, p


# ----------------------------------------------------------------------------
# E6 (+E7): DownTransition.forward   (model/modules.py:113-163)
# ----------------------------------------------------------------------------
def down_transition(sd, x, p, factor, knn_k, norm_type='none', return_inds=False, training=False):
    assert x.shape[:2] == p.shape[:2]
    B, N, d_in = x.shape
    n_new = int(np.ceil(N / factor))                            # :126
    p_flat = p.reshape(B * N, 3)
    batch = torch.arange(B).repeat_interleave(N)
    inds = cluster.fps(p_flat, batch, ratio=1.0 / factor, random_start=False)  # :133
    inds = torch.sort(inds)[0]                                  # :135
    p_sub = p_flat[inds]
    batch_sub = torch.arange(B).repeat_interleave(n_new)
    nn = cluster.knn(p_flat, p_sub, knn_k, batch, batch_sub)[1].view(B * n_new, knn_k)  # :142-146
    y = _lin(sd, 'mlp.0', x.reshape(B * N, d_in))               # :152
    if norm_type == 'layer':
        y = F.layer_norm(y, (y.shape[-1],), sd['mlp.1.weight'], sd['mlp.1.bias'], 1e-5)
    elif norm_type == 'batch':                                  # :98-102: BatchNorm1d(eps 1e-3, momentum 0.1)
        # eval mode: running statistics; training mode: the statistics of the B N rows, and the running statistics in
        # `sd` move towards them in place (unbiased variance), as the module does
        y = F.batch_norm(y, sd['mlp.1.running_mean'], sd['mlp.1.running_var'], sd['mlp.1.weight'], sd['mlp.1.bias'],
                         training, 0.1 if training else 0.0, 1e-3)
    elif norm_type != 'none':
        raise ValueError(norm_type)
    y = _relu(y)
    z = y[nn[:, 0]]                                             # :156-158
    for i in range(1, knn_k):
        z = torch.maximum(z, y[nn[:, i]])
    z = z.view(B, n_new, -1)
    p_sub = p_sub.view(B, n_new, 3)
    if return_inds:
        return z, p_sub, inds, nn
    return z, p_sub


# ----------------------------------------------------------------------------
# E1: PointCompletionNetV3.forward   (model/model.py:148-233)
# ----------------------------------------------------------------------------
def encoder_forward(sd, cfg, pcl):
    """pcl (B,N,d_in) -> (pcl_out (B,M,3+D), x_global (B,global_dim)).
    cfg keys: down_blocks, transition_factor, pt_num_neighbors, pt_norm_type,
    down_neighbors, abstract_levels (model/model.py:18-22)."""
    nb = cfg['down_blocks']
    x = _lin(sd, 'pre_mlp.2', _relu(_lin(sd, 'pre_mlp.0', pcl)))   # :167
    pos = pcl[..., :3]                                               # :168
    skips = []
    x_global = None
    for i in range(2 * nb + 1):
        bsd = _sub(sd, 'blocks.%d.' % i)
        if i % 2 == 0:
            x, pos = pt_block(bsd, x, pos, num_neighbors=cfg['pt_num_neighbors'])
        else:
            x, pos = down_transition(bsd, x, pos, cfg['transition_factor'],
                                     cfg['down_neighbors'], cfg['pt_norm_type'])
        if i == 2 * nb:                                              # :188-190
            x_global = _lin(sd, 'global_mlp.2', _relu(_lin(sd, 'global_mlp.0', x.mean(dim=1))))
        if cfg['abstract_levels'] > 1 and i % 2 == 1:                # :202-207
            j = 0
            while ('abstract_skip_mlps.%d.weight' % j) in sd:
                w = sd['abstract_skip_mlps.%d.weight' % j]
                if w.shape[1] == x.shape[-1]:
                    y = _lin(sd, 'abstract_skip_mlps.%d' % j, x)
                    y[..., -1] = j + 1.0
                    skips.append(torch.cat([pos, y], dim=-1))
                j += 1
    out = torch.cat([pos, x], dim=-1)                                # :220
    if cfg['abstract_levels'] > 1:                                   # :224-228
        out[..., -1] = cfg['abstract_levels']
        assert len(skips) == cfg['abstract_levels'] - 1
        out = torch.cat([torch.cat(skips, dim=1), out], dim=1)
    return out, x_global


# ----------------------------------------------------------------------------
# D5: positional_encode   (model/implicit.py:20-43)
# ----------------------------------------------------------------------------
def positional_encode(points, base_frequency, num_powers):
    parts = [points]
    for p in range(num_powers):
        omega = base_frequency * (2 ** p) * np.pi * 2.0   # python double -> fp32 multiply
        parts.append(torch.sin(points * omega))
        parts.append(torch.cos(points * omega))
    return torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------
# D2: geometry.my_knn_torch   (utils/geometry.py:458-503)
# ----------------------------------------------------------------------------
def knn_with_dists(pcl_query, pcl_key, k, chunk=8192):
    """(N,>=3),(M,>=3) -> inds (N,k) int64, dists (N,k) fp32 (Euclidean, via
    linalg.norm of the broadcast difference, then topk(largest=False) along the
    key axis).  Chunked over queries; per-query arithmetic is unchanged."""
    inds, dists = [], []
    for lo in range(0, pcl_query.shape[0], chunk):
        q = pcl_query[lo:lo + chunk]
        diffs = q[None, :, :3] - pcl_key[:, None, :3]            # :479  (M,n,3)
        d = torch.linalg.norm(diffs, axis=-1, ord=2)             # :481
        if STABLE_TIES:
            dk, ik = torch.sort(d, dim=0, stable=True)
            dk, ik = dk[:k], ik[:k]
        else:
            dk, ik = d.topk(k, dim=0, largest=False)             # :484
        inds.append(ik.permute(1, 0))
        dists.append(dk.permute(1, 0))
    return torch.cat(inds), torch.cat(dists)


def tie_ambiguous(points_query, points_abstract, k_interp, k_attn):
    """(N,) bool: queries whose k-th and (k+1)-th nearest abstract points are equidistant for
    the interpolation kNN (Euclidean norm, k_interp) or the attention kNN (squared sum,
    k_attn) -- for these the reference's own result is implementation-defined."""
    q, a = points_query[:, :3], points_abstract[:, :3]
    amb = torch.zeros(q.shape[0], dtype=torch.bool)
    for lo in range(0, q.shape[0], 4096):
        qq = q[lo:lo + 4096]
        dn = torch.linalg.norm(qq[None] - a[:, None], axis=-1, ord=2).T.sort(dim=1)[0]
        ds = torch.sum((qq[:, None] - a[None]) ** 2, dim=-1).sort(dim=1)[0]
        t = torch.zeros(qq.shape[0], dtype=torch.bool)
        if a.shape[0] > k_interp:
            t |= dn[:, k_interp - 1] == dn[:, k_interp]
        if a.shape[0] > k_attn:
            t |= ds[:, k_attn - 1] == ds[:, k_attn]
        amb[lo:lo + 4096] = t
    return amb


def _act(name, x):
    if name == 'relu':
        return _relu(x)
    if name == 'swish':                                          # model/implicit.py:46-64
        return x * torch.sigmoid(x)
    raise ValueError('Unknown activation: ' + str(name))


# ----------------------------------------------------------------------------
# D1, D3, D4, D6, D7: LocalPclResnetFC.forward / do_forward_attention
#                      (model/implicit.py:271-445)
# ----------------------------------------------------------------------------
def decoder_forward(sd, cfg, points_query, points_abstract, features_global,
                    features_abstract=None, knn_local=None, knn_cross=None):
    """points_query (N,4); points_abstract (M,3+E) (or (M,3) with features_abstract
    (M,E)); features_global (D,) -> (output (N,G), penult (N,H)).
    cfg keys: n_blocks, pos_encoding_freqs, activation, num_local_features,
    cross_attn_neighbors, cross_attn_layers, d_latent, d_latent_local.
    knn_local (N,8) / knn_cross (N,14) int64: the lists my_knn_torch (:328) / kNN_torch
    (model/point_transformer_layer.py:167) returned in a particular reference run; the distances of knn_local
    are recomputed with the same linalg.norm on the gathered differences."""
    if features_abstract is None:                                # :286-290
        features_abstract = points_abstract[..., 3:]
        points_abstract = points_abstract[..., :3]
    assert points_query.dim() == 2, 'oracle takes the un-batched (B == 1) form'
    act = cfg.get('activation', 'relu')
    nblk = cfg['n_blocks']
    L = cfg['cross_attn_layers']
    use_at = {int((i + 1) * nblk / (L + 1)): i for i in range(L)}   # :264-268

    # D2 + D3: local feature interpolation (:328-342)
    abstract = torch.cat([points_abstract, features_abstract], dim=-1)
    if knn_local is None:
        inds, dists = knn_with_dists(points_query, abstract, cfg['num_local_features'])
    else:
        inds = torch.as_tensor(knn_local).long()
        dists = torch.linalg.norm(points_query[:, None, :3] - points_abstract[inds], axis=-1, ord=2)   # :479-481
    w = 1.0 / (dists + 1e-4)
    w = F.normalize(w, p=1, dim=-1)
    f_local = torch.einsum('ik,ikf->if', w, features_abstract[inds])
    f_query = torch.cat([features_global[None, :].expand(points_query.shape[0], -1), f_local], dim=-1)
    assert f_query.shape[-1] == cfg['d_latent']

    # D4 (:380-445)
    pe = positional_encode(points_query, 0.1, cfg['pos_encoding_freqs']) \
        if cfg['pos_encoding_freqs'] > 0 else points_query
    x = _lin(sd, 'lin_in', pe)
    p = pe[..., :3][None]                                        # :424 (raw xyz come first)
    for i in range(nblk):
        x = x + _lin(sd, 'lin_z.%d' % i, f_query)                # :416-417
        h = _lin(sd, 'blocks.%d.fc_0' % i, _act(act, x))         # D6 (:92-101)
        x = x + _lin(sd, 'blocks.%d.fc_1' % i, _act(act, h))
        if i in use_at:                                          # :421-439
            bsd = _sub(sd, 'pt_blocks.%d.' % use_at[i])
            x = pt_block(bsd, x[None], p, features_abstract[None], points_abstract[None],
                         cfg['cross_attn_neighbors'],
                         None if knn_cross is None else torch.as_tensor(knn_cross).long()[None])[0][0]
    penult = x
    out = _lin(sd, 'lin_out', _act(act, x))                      # :441-443
    return out, penult


# ----------------------------------------------------------------------------
# D9: sample_implicit_points_blind_numpy, grid / random modes
#     (utils/geometry.py:1199-1283)
# ----------------------------------------------------------------------------
def query_cuboid(min_z, cube_bounds, data_kind, cube_mode):
    cb = cube_bounds
    if data_kind == 'greater':                                   # :1215-1218
        return (-cb, cb), (-cb, cb), (min_z, cb)
    if data_kind == 'carla':                                     # :1220-1241
        table = {1: (2.0, 1.0, 0.5), 2: (2.4, 0.8, 0.4), 3: (2.2, 1.0, 0.4), 4: (2.5, 1.0, 0.4)}
        fx, fy, fz = table[cube_mode]
        return (0.0, cb * fx), (-cb * fy, cb * fy), (min_z, cb * fz)
    raise ValueError(data_kind)


def sample_query_points(num_sample, min_z, cube_bounds, time_idx, data_kind, cube_mode,
                        point_sample_mode):
    (x0, x1), (y0, y1), (z0, z1) = query_cuboid(min_z, cube_bounds, data_kind, cube_mode)
    if point_sample_mode == 'random':                            # :1247-1255
        n = num_sample
        px = np.random.rand(n).astype(np.float32) * (x1 - x0) + x0
        py = np.random.rand(n).astype(np.float32) * (y1 - y0) + y0
        pz = np.random.rand(n).astype(np.float32) * (z1 - z0) + z0
        xyz = np.stack([px, py, pz], axis=-1)
    elif point_sample_mode == 'grid':                            # :1257-1275
        per_unit = np.cbrt(num_sample / ((x1 - x0) * (y1 - y0) * (z1 - z0)))
        nx = int(np.ceil(per_unit * (x1 - x0)))
        ny = int(np.ceil(per_unit * (y1 - y0)))
        nz = int(np.ceil(per_unit * (z1 - z0)))
        gx = (np.arange(nx, dtype=np.float32) + 0.5) * ((x1 - x0) / nx) + x0
        gy = (np.arange(ny, dtype=np.float32) + 0.5) * ((y1 - y0) / ny) + y0
        gz = (np.arange(nz, dtype=np.float32) + 0.5) * ((z1 - z0) / nz) + z0
        n = nx * ny * nz
        xyz = np.stack([np.repeat(gx, ny * nz), np.tile(np.repeat(gy, nz), nx),
                        np.tile(gz, nx * ny)], axis=-1)          # x slowest, z fastest
    else:
        raise ValueError(point_sample_mode)
    t = np.ones((n, 1), dtype=np.float32) * time_idx             # :1281
    return np.concatenate([xyz, t], axis=-1)


# ----------------------------------------------------------------------------
# D8: perform_inference   (eval/inference.py:83-325), track_mode none/one
# ----------------------------------------------------------------------------
def subsample_pad_pcl(pcl, n_desired, sample_mode='random', subsample_only=False, retain_vehped=False,
                      segm_idx=None):
    """utils/geometry.py:294-376: zero-pad a too-small cloud; subsample a too-large one uniformly at random
    (numpy global stream, sorted indices) or by torch_cluster.fps with ratio n_remain/N - 1e-7 and a random
    start (oracle/cluster.py: torch global stream), indices sorted; retain_vehped keeps tags 4 and 10 and
    samples the rest from everything except tag 10 (the reference's masks)."""
    assert sample_mode in ['random', 'farthest_point']
    no_batch = pcl.dim() == 2
    if no_batch:
        pcl = pcl[None]
    (B, N, D) = pcl.shape
    if N < n_desired:
        if subsample_only:
            raise RuntimeError('Too few input points: %d vs %d.' % (N, n_desired))
        result = torch.cat([pcl, torch.zeros((B, n_desired - N, D), dtype=pcl.dtype)], dim=1)
        return result[0] if no_batch else result
    if N == n_desired:
        return pcl[0] if no_batch else pcl
    assert B == 1
    n_remain = n_desired
    if retain_vehped:
        tags = pcl[0, :, segm_idx].numpy()
        retain_inds = np.where(np.logical_or(tags == 4, tags == 10))[0]
        remain_inds = np.where(tags != 10)[0]
        n_remain -= retain_inds.shape[0]
    else:
        remain_inds = np.arange(N)
    if sample_mode == 'random':
        inds = np.random.choice(remain_inds, n_remain, replace=False)
        inds.sort()
        result = pcl[:, inds]
    else:
        assert not retain_vehped
        inds = torch.sort(cluster.fps(pcl[0, :, :3], None, n_remain / N - 1e-7, True))[0]
        result = pcl[0][inds].view(B, n_remain, D)
    if no_batch:
        result = result[0]
    if retain_vehped:
        result = torch.cat([pcl[0][retain_inds], result], dim=0)
    assert result.shape[0] == n_desired
    return result


def track_channel(color_mode):                                   # utils/utils.py:204-224
    return {'rgb': 4, 'rgb_nosigmoid': 4, 'hsv': 15, 'bins': 10}[color_mode]


def squash_outputs(o, color_mode, predict_segmentation, track_mode, semantic_classes):
    """In-place post-ops of eval/inference.py:218-243 on an (n,G) tensor."""
    o[..., 0] = torch.sigmoid(o[..., 0])
    if color_mode == 'rgb':
        o[..., 1:4] = torch.sigmoid(o[..., 1:4])
    elif color_mode == 'rgb_nosigmoid':
        o[..., 1:4] = torch.clamp(o[..., 1:4].clone(), min=0.0, max=1.0)
    elif color_mode == 'hsv':
        o[..., 1:13] = torch.sigmoid(o[..., 1:13])
        o[..., 13:15] = torch.clamp(o[..., 13:15].clone(), min=0.0, max=1.0)
    elif color_mode == 'bins':
        o[..., 1:10] = torch.sigmoid(o[..., 1:10])
    if predict_segmentation:
        o[..., -semantic_classes:] = torch.sigmoid(o[..., -semantic_classes:])
    if track_mode != 'none':
        ti = track_channel(color_mode)
        o[..., ti] = torch.sigmoid(o[..., ti])
    return o


def merge_tracks(track_instance_ids, pcl_abstract, features_global, implicit_output, output_track_idx):
    """utils/utils.py:343-397 (multi_track_merge)."""
    if len(pcl_abstract) == 1 and track_instance_ids[0] == -1:                # :369-370
        return pcl_abstract[0], features_global[0], implicit_output[0]
    m_abs = np.mean(pcl_abstract, axis=0)                                     # :373-378
    m_glob = np.mean(features_global, axis=0)
    m_out = np.mean(implicit_output, axis=0)
    mark = -np.ones_like(m_out[..., 0])                                       # :383-390
    conf = np.zeros_like(m_out[..., 0])
    for t, inst in enumerate(track_instance_ids):
        score = implicit_output[t][..., output_track_idx]
        mark[np.logical_and(score >= 0.5, score >= conf)] = inst
        conf = np.maximum(score, conf)
    m_out[..., output_track_idx] = mark
    return m_abs, m_glob, m_out


def nn1_label(points_query, pcl_target_xyz, thresh):
    """utils/geometry.py:444-455 (get_1nn_label): sklearn KDTree, Euclidean."""
    import sklearn.neighbors
    kdt = sklearn.neighbors.KDTree(pcl_target_xyz, leaf_size=30, metric='euclidean')
    dist, ind = kdt.query(points_query, k=1, return_distance=True)
    return (dist[:, 0] < thresh) * 1, ind


def perform_inference(pcl_input, enc_sd, enc_cfg, dec_sd, dec_cfg, min_z, cube_bounds, color_mode,
                      time_idx, num_sample=16384, point_sample_mode='random', batch_size=1024,
                      predict_segmentation=False, track_mode='none', semantic_classes=13,
                      density_threshold=0.5, data_kind='', cube_mode=4, compress_air=False,
                      pcl_input_sem=None, pcl_target_frame=None, point_occupancy_radius=0.2, neighbour_lists=None):
    """neighbour_lists = (knn_local (N_q,8), knn_cross (N_q,14)): a reference run's own decoder lists (or None)."""
    if isinstance(pcl_input, np.ndarray):
        pcl_input = torch.from_numpy(pcl_input).unsqueeze(0)
    ti = track_channel(color_mode)
    inst_col = 0 if data_kind == 'greater' else 1
    if track_mode in ('none', 'one'):                                         # :140-142
        track_ids = [-1]
    else:                                                                     # :144-161
        assert data_kind == 'greater' and pcl_input_sem.shape[-1] == 1
        sem_np = pcl_input_sem if isinstance(pcl_input_sem, np.ndarray) else pcl_input_sem[0].numpy()
        sem = torch.from_numpy(sem_np).unsqueeze(0) if isinstance(pcl_input_sem, np.ndarray) else pcl_input_sem
        ids, counts = np.unique(sem_np, return_counts=True)
        track_ids = [int(i) for i, c in zip(ids, counts) if i >= 0 and c >= 16]
    points_query = sample_query_points(num_sample, min_z, cube_bounds, time_idx, data_kind,
                                       cube_mode, point_sample_mode)          # :175
    runs_abs, runs_glob, runs_out = [], [], []
    for inst in track_ids:                                                    # :187
        if inst >= 0:
            pcl_input[..., -1] = (sem[..., inst_col] == inst)                 # :190-193
        pcl_abstract, f_global = encoder_forward(enc_sd, enc_cfg, pcl_input)  # :195
        pcl_abstract, f_global = pcl_abstract[0], f_global[0]
        outs = []
        for lo in range(0, points_query.shape[0], batch_size):                # :204
            q = torch.from_numpy(points_query[lo:lo + batch_size])
            kw = {} if neighbour_lists is None else dict(knn_local=neighbour_lists[0][lo:lo + batch_size],
                                                         knn_cross=neighbour_lists[1][lo:lo + batch_size])
            o, _ = decoder_forward(dec_sd, dec_cfg, q, pcl_abstract, f_global, **kw)    # :211
            outs.append(squash_outputs(o, color_mode, predict_segmentation, track_mode,
                                       semantic_classes).numpy())
        runs_out.append(np.concatenate(outs, axis=0))
        runs_abs.append(pcl_abstract.numpy())
        runs_glob.append(f_global.numpy())
    pcl_abstract, f_global, implicit_output = merge_tracks(track_ids, runs_abs, runs_glob, runs_out, ti)   # :265
    io = np.concatenate([points_query, implicit_output], axis=-1)             # :279
    keep = io[..., 4] >= density_threshold                                    # :283-284
    solid, air = io[keep], io[~keep]
    result = dict(pcl_abstract=pcl_abstract, features_global=f_global, implicit_output=implicit_output,
                  points_query=points_query)
    if pcl_target_frame is not None:                                          # :270-276, :285-287
        labels, nn_idx = nn1_label(points_query[:, :3], pcl_target_frame[..., :3], point_occupancy_radius)
        nngt = np.concatenate([labels[:, None], pcl_target_frame[nn_idx][:, 0, :]], axis=-1)
        result['gt_solid'], result['gt_air'] = nngt[keep], nngt[~keep]
    if compress_air:                                                          # :299-311
        seg = air[..., -semantic_classes:].argmax(axis=-1)
        air = np.concatenate([air[..., :3], air[..., 4:5], seg[..., None]], axis=-1)
        if pcl_target_frame is not None:
            result['gt_air'] = np.concatenate([result['gt_air'][..., :1], result['gt_air'][..., 4:5]], axis=-1)
    result['output_solid'], result['output_air'] = solid, air
    return result
