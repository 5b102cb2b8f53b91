"""Named hot-path configurations, synthetic inputs and deterministic weights.

The reference builds its two networks from argparse values in train.py:194-265
and stores the constructor kwargs (``pcl_args`` / ``implicit_args``) in the
checkpoint (train.py:339-350).  Datasets and checkpoints are not available, so
the BASELINE.json configs are reproduced here as those kwargs dicts, together
with a seeded synthetic point-cloud video of the right layout and a
platform-independent weight filler (numpy PCG64; the U(-1/sqrt(fan_in), ..) law
of torch's Linear default) so the CPU oracle and the GPU path see identical
weights without shipping a 29 MB checkpoint.
"""
import numpy as np
import torch


def model_args(kind, n_points=14336):
    """kwargs for PointCompletionNetV3 / LocalPclResnetFC plus the inference
    settings of the README commands (README.md:36,41,52) pushed through
    train.py:194-265 and the args.py defaults."""
    kind = kind.lower()
    assert kind in ('greater', 'carla')
    carla = kind == 'carla'
    d_feat, down_blocks, global_size = 36, 3, 128
    pcl_args = dict(
        mixed_precision=False, n_input=n_points, n_output=n_points, d_in=8, d_out=1,
        d_feat=d_feat, down_blocks=down_blocks, up_blocks=down_blocks, transition_factor=3,
        pt_num_neighbors=16, pt_norm_type='layer' if carla else 'none', down_neighbors=12,
        abstract_levels=2 if carla else 1, skip_connections=False, enable_decoder=False,
        output_featurized=True, output_global_emb=True, global_dim=global_size,
        fps_random_start=False)
    d_local = d_feat * 2 ** down_blocks
    g = 1 + 3 + 1 + (13 if carla else 0)
    implicit_args = dict(
        mixed_precision=False, d_in=4, d_hidden=global_size + d_local, d_out=g,
        d_latent=global_size + d_local, n_blocks=6, pos_encoding_freqs=8, activation='relu',
        num_local_features=8, local_mode='attention', d_latent_local=d_local,
        cross_attn_neighbors=14, cross_attn_layers=2, cr_attn_type='cc')
    infer_args = dict(
        min_z=-1.0, cube_bounds=16.0 if carla else 5.0,
        color_mode='rgb' if carla else 'rgb_nosigmoid',
        predict_segmentation=carla, track_mode='none', semantic_classes=13,
        density_threshold=0.5, data_kind=kind, cube_mode=4, point_sample_mode='grid')
    return pcl_args, implicit_args, infer_args


def input_cuboid(kind):
    """Input point-cloud bounds: GREATER cube +-5 with min_z -1; CARLA cube_mode 4
    with pt_cube_bounds 20 (utils/geometry.py:216-219)."""
    if kind == 'greater':
        return (-5.0, 5.0), (-5.0, 5.0), (-1.0, 5.0)
    return (-14.0, 50.0), (-20.0, 20.0), (-1.0, 10.0)


def synthetic_pcl(kind, n_points, video_len, seed=1830):
    """(1, n_points, 8) fp32 rows (x,y,z,R,G,B,t,mark_track): half uniform in the
    input cuboid, half jittered around 32 planar patches; exact duplicate xyz are
    re-drawn so that neighbour order is tie-free."""
    rng = np.random.default_rng(seed)
    (x0, x1), (y0, y1), (z0, z1) = input_cuboid(kind)
    lo = np.array([x0, y0, z0], dtype=np.float64)
    hi = np.array([x1, y1, z1], dtype=np.float64)
    n_u = n_points // 2
    pts_u = rng.uniform(lo, hi, size=(n_u, 3))
    n_s = n_points - n_u
    centres = rng.uniform(lo, hi, size=(32, 3))
    axes = rng.normal(size=(32, 2, 3))
    which = rng.integers(0, 32, size=n_s)
    uv = rng.uniform(-1.0, 1.0, size=(n_s, 2, 1)) * (0.12 * (hi - lo).min())
    pts_s = centres[which] + (axes[which] * uv).sum(axis=1) + rng.normal(scale=0.05, size=(n_s, 3))
    pts_s = np.clip(pts_s, lo, hi)
    xyz = np.concatenate([pts_u, pts_s]).astype(np.float32)
    xyz = xyz[rng.permutation(n_points)]
    for _ in range(8):
        _, first = np.unique(xyz, axis=0, return_index=True)
        if first.size == n_points:
            break
        dup = np.setdiff1d(np.arange(n_points), first)
        xyz[dup] = rng.uniform(lo, hi, size=(dup.size, 3)).astype(np.float32)
    rgb = rng.uniform(0.0, 1.0, size=(n_points, 3)).astype(np.float32)
    t = rng.integers(0, video_len, size=(n_points, 1)).astype(np.float32)
    mark = np.zeros((n_points, 1), dtype=np.float32)
    return torch.from_numpy(np.concatenate([xyz, rgb, t, mark], axis=1)[None])


def fill_state_dict(module_or_shapes, seed):
    """Deterministic weights for every entry of a state_dict (or {name: shape}):
    a rank-2 '*.weight' and its '*.bias' ~ U(-b, b) with b = 1/sqrt(fan_in); a
    rank-1 '*.weight' (LayerNorm gain) = 1 + 0.1 U(-1,1), its bias 0.1 U(-1,1)."""
    if hasattr(module_or_shapes, 'state_dict'):
        shapes = {k: tuple(v.shape) for k, v in module_or_shapes.state_dict().items()}
    else:
        shapes = dict(module_or_shapes)
    rng = np.random.default_rng(seed)
    fan_in = {n[:-7]: s[1] for n, s in shapes.items() if n.endswith('.weight') and len(s) == 2}
    out = {}
    for name, shp in shapes.items():
        base = name.rsplit('.', 1)[0]
        u = rng.uniform(-1.0, 1.0, size=shp)
        if base in fan_in:
            val = u / np.sqrt(fan_in[base])
        elif name.endswith('.weight'):
            val = 1.0 + 0.1 * u
        else:
            val = 0.1 * u
        out[name] = torch.from_numpy(val.astype(np.float32))
    return out


def _lin_shapes(s, name, o, i, bias=True):
    s[name + '.weight'] = (o, i)
    if bias:
        s[name + '.bias'] = (o,)


def _ptb_shapes(s, prefix, dim, dim2=None):
    dim2 = dim if dim2 is None else dim2
    _lin_shapes(s, prefix + 'layer1', dim, dim)
    _lin_shapes(s, prefix + 'layer2.to_q', dim, dim, False)
    _lin_shapes(s, prefix + 'layer2.to_k', dim, dim2, False)
    _lin_shapes(s, prefix + 'layer2.to_v', dim, dim2, False)
    _lin_shapes(s, prefix + 'layer2.pos_mlp.0', 32, 3)
    _lin_shapes(s, prefix + 'layer2.pos_mlp.2', dim, 32)
    _lin_shapes(s, prefix + 'layer2.attn_mlp.0', 2 * dim, dim)
    _lin_shapes(s, prefix + 'layer2.attn_mlp.2', dim, 2 * dim)
    _lin_shapes(s, prefix + 'layer3', dim, dim)


def encoder_param_shapes(pcl_args):
    """state_dict layout of PointCompletionNetV3 (SURVEY.md §8(b) parameter names)."""
    d, nb, gd = pcl_args['d_feat'], pcl_args['down_blocks'], pcl_args['global_dim']
    s = {}
    _lin_shapes(s, 'pre_mlp.0', d, pcl_args['d_in'])
    _lin_shapes(s, 'pre_mlp.2', d, d)
    dim = d
    for b in range(nb):
        _ptb_shapes(s, 'blocks.%d.' % (2 * b), dim)
        _lin_shapes(s, 'blocks.%d.mlp.0' % (2 * b + 1), 2 * dim, dim)
        if pcl_args['pt_norm_type'] == 'layer':
            s['blocks.%d.mlp.1.weight' % (2 * b + 1)] = (2 * dim,)
            s['blocks.%d.mlp.1.bias' % (2 * b + 1)] = (2 * dim,)
        dim *= 2
    _ptb_shapes(s, 'blocks.%d.' % (2 * nb), dim)
    _lin_shapes(s, 'global_mlp.0', gd, dim)
    _lin_shapes(s, 'global_mlp.2', gd, gd)
    al = pcl_args['abstract_levels']
    for j in range(al - 1):
        _lin_shapes(s, 'abstract_skip_mlps.%d' % j, dim, dim // 2 ** (al - 1 - j))
    return s


def decoder_param_shapes(implicit_args):
    """state_dict layout of LocalPclResnetFC."""
    h, g, dl = implicit_args['d_hidden'], implicit_args['d_out'], implicit_args['d_latent']
    e = implicit_args['d_latent_local']
    f = implicit_args['pos_encoding_freqs']
    d_in = implicit_args['d_in'] * (2 * f + 1) if f > 0 else implicit_args['d_in']
    s = {}
    _lin_shapes(s, 'lin_in', h, d_in)
    _lin_shapes(s, 'lin_out', g, h)
    for b in range(implicit_args['n_blocks']):
        _lin_shapes(s, 'blocks.%d.fc_0' % b, h, h)
        _lin_shapes(s, 'blocks.%d.fc_1' % b, h, h)
    for b in range(implicit_args['n_blocks']):
        _lin_shapes(s, 'lin_z.%d' % b, h, dl)
    for i in range(implicit_args['cross_attn_layers']):
        _ptb_shapes(s, 'pt_blocks.%d.' % i, dl, e)
    return s


def synthetic_weights(pcl_args, implicit_args, seed=1830):
    """(encoder_state_dict, decoder_state_dict) filled deterministically."""
    return (fill_state_dict(encoder_param_shapes(pcl_args), seed),
            fill_state_dict(decoder_param_shapes(implicit_args), seed + 1))


def synthetic_target_frames(kind, m, frames, seed, batch=1):
    """Seeded stand-in for the dataloader's supervision point clouds (datasets are unavailable): list-T of
    (B, M, E) float32 frames + sizes + vehicle / pedestrian ids, in the layout GuidedImplicitPointSampler expects
    (GREATER E=9: x,y,z,instance,view,R,G,B,mark; CARLA E=11: x,y,z,cos,instance,semantic,view,R,G,B,mark).
    A static background shared by all frames plus a blob that moves from frame to frame."""
    rng = np.random.default_rng(seed)
    carla = kind == 'carla'
    lo, hi = (np.array([-5.0, -5.0, -1.0]), np.array([5.0, 5.0, 5.0])) if not carla else \
        (np.array([0.0, -16.0, -1.0]), np.array([40.0, 16.0, 6.4]))
    n_blob = m // 5
    base = rng.uniform(lo, hi, size=(batch, m - n_blob, 3))
    out, sizes = [], []
    for t in range(frames):
        centre = lo + (hi - lo) * (0.25 + 0.5 * t / max(1, frames - 1))
        blob = centre + rng.normal(scale=0.6, size=(batch, n_blob, 3))
        xyz = np.concatenate([base + rng.normal(scale=0.005, size=base.shape), blob], axis=1)
        cols = [xyz]
        if carla:
            cols.append(rng.uniform(-1, 1, size=(batch, m, 1)))
        cols.append(rng.integers(-1, 6, size=(batch, m, 1)).astype(np.float64))
        if carla:
            cols.append(rng.integers(0, 13, size=(batch, m, 1)).astype(np.float64))
        cols += [rng.integers(0, 3, size=(batch, m, 1)).astype(np.float64), rng.uniform(size=(batch, m, 3)),
                 rng.integers(0, 2, size=(batch, m, 1)).astype(np.float64)]
        rows = np.concatenate(cols, axis=-1).astype(np.float32)
        for b in range(batch):
            rows[b] = rows[b][rng.permutation(m)]
        out.append(torch.from_numpy(rows))
        sizes.append(torch.full((batch,), m, dtype=torch.int64))
    valo = torch.tensor([[0, 2, 5, 0]] * batch, dtype=torch.int64)
    return out, sizes, valo, torch.full((batch,), 3, dtype=torch.int64)
