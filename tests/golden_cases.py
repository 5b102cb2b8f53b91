"""Case definitions + seeded input builders shared by oracle/gen_golden.py (which
ran the real reference on them in the build container) and the parity tests
(which rebuild the identical inputs anywhere and compare with tests/golden/*.npz).

Inputs come from numpy's PCG64 ``default_rng(seed)`` (stream stable across numpy
versions) so only outputs need to be stored.
"""
import numpy as np
import torch

import occlusions4d_amd as pk

cfg = pk.configs


def _rng(seed):
    return np.random.default_rng(seed)


def _cloud(rng, n, scale=5.0):
    return rng.uniform(-scale, scale, size=(n, 3)).astype(np.float32)


# ---------------------------------------------------------------- G1
KNN_CASES = [
    dict(name='self_n700_k16', n0=700, n1=700, k=16, self_=True, seed=11),
    dict(name='cross_n300_m531_k14', n0=300, n1=531, k=14, self_=False, seed=12),
    dict(name='cross_n257_m76_k12', n0=257, n1=76, k=12, self_=False, seed=13),
    dict(name='tiny_n5_m9_k8', n0=5, n1=9, k=8, self_=False, seed=14),
]


def knn_inputs(case):
    rng = _rng(case['seed'])
    d = _cloud(rng, case['n1'])
    q = d if case['self_'] else _cloud(rng, case['n0'])
    return q, d


# ---------------------------------------------------------------- G2 / G3
PTL_CASES = [
    dict(name='self_d36_k16', dim=36, k=16, n=200, seed=21),
    dict(name='self_d72_k16', dim=72, k=16, n=90, seed=22),
    dict(name='cross_d416_e288_k14', dim=416, dim2=288, k=14, n=64, m=76, seed=23),
]
PTB_CASES = [
    dict(name='self_d36_k16', dim=36, k=16, n=150, seed=31),
    dict(name='cross_d416_e288_k14', dim=416, dim2=288, k=14, n=48, m=100, seed=32),
]


# Round 4: the same layer outside the init-scale regime (VERDICT r3 "What's weak" 1).  `wscale` multiplies every
# rank-2 weight of the layer (biases stay), `fscale` the input features; `zero` / `scale` edit single tensors.  With
# wscale s the attention logits grow ~ s^3: at 4 / 8 they are far beyond the softmax's exp range (one neighbour
# dominates most channels); zero attn_mlp.2.weight makes every logit equal (uniform 1/K weights).  The reference is
# run in fp32 AND fp64 on these; parity bound = max(1e-4, 2 max|ref32 - ref64|).
PTL_REGIME_CASES = [
    dict(name='cross_d416_w4', dim=416, dim2=288, k=14, n=100, m=76, seed=25, wscale=4.0, fscale=4.0),
    dict(name='cross_d416_w8', dim=416, dim2=288, k=14, n=100, m=76, seed=26, wscale=8.0, fscale=4.0),
    dict(name='cross_d416_equal_logits', dim=416, dim2=288, k=14, n=100, m=76, seed=27, zero=['attn_mlp.2.weight']),
    dict(name='cross_d416_dominant', dim=416, dim2=288, k=14, n=100, m=76, seed=28,
         scale={'attn_mlp.2.weight': 64.0}),
    dict(name='cross_d288_w8', dim=288, dim2=288, k=14, n=60, m=50, seed=30, wscale=8.0, fscale=4.0),
    dict(name='self_d36_w8', dim=36, k=16, n=200, seed=29, wscale=8.0, fscale=4.0),
]


def regime_edit(sd, case, prefix=''):
    """Applies a case's wscale / zero / scale edits to the entries of `sd` whose name starts with `prefix`."""
    out = dict(sd)
    for name, v in sd.items():
        if not name.startswith(prefix):
            continue
        short = name[len(prefix):]
        if case.get('wscale') is not None and v.dim() == 2:
            out[name] = v * np.float32(case['wscale'])
        if short in case.get('zero', ()):
            out[name] = torch.zeros_like(v)
        if short in case.get('scale', {}):
            out[name] = v * np.float32(case['scale'][short])
    return out


def _ptl_shapes(dim, dim2):
    s = {}
    cfg._ptb_shapes(s, '', dim, dim2)
    return {k[len('layer2.'):]: v for k, v in s.items() if k.startswith('layer2.')}


def ptl_inputs(case):
    rng = _rng(case['seed'])
    x = rng.normal(size=(case['n'], case['dim'])).astype(np.float32)
    pos = _cloud(rng, case['n'])
    x2 = pos2 = None
    if 'dim2' in case:
        x2 = rng.normal(size=(case['m'], case['dim2'])).astype(np.float32)
        pos2 = _cloud(rng, case['m'])
    sd = cfg.fill_state_dict(_ptl_shapes(case['dim'], case.get('dim2')), case['seed'] + 1000)
    sd = regime_edit(sd, case)
    fs = np.float32(case.get('fscale', 1.0))
    x = x * fs
    x2 = None if x2 is None else x2 * fs
    return x, pos, x2, pos2, sd


def ptb_inputs(case):
    rng = _rng(case['seed'])
    x = rng.normal(size=(case['n'], case['dim'])).astype(np.float32)
    pos = _cloud(rng, case['n'])
    x2 = pos2 = None
    if 'dim2' in case:
        x2 = rng.normal(size=(case['m'], case['dim2'])).astype(np.float32)
        pos2 = _cloud(rng, case['m'])
    s = {}
    cfg._ptb_shapes(s, '', case['dim'], case.get('dim2'))
    return x, pos, x2, pos2, cfg.fill_state_dict(s, case['seed'] + 1000)


# ---------------------------------------------------------------- G4
DOWN_CASES = [
    dict(name='none_n301_36to72_k12', n=301, d_in=36, d_out=72, k=12, norm='none', seed=41),
    dict(name='layer_n200_72to144_k12', n=200, d_in=72, d_out=144, k=12, norm='layer', seed=42),
]


# Round 4: the BatchNorm variant of DownTransition (model/modules.py:98-102; a constructor option no published
# configuration uses), in eval mode: running statistics as a trained checkpoint would carry them.
DOWN_BATCHNORM_CASES = [
    dict(name='batch_n200_72to144_k12', n=200, d_in=72, d_out=144, k=12, norm='batch', seed=43),
]


def down_inputs(case):
    rng = _rng(case['seed'])
    x = rng.normal(size=(case['n'], case['d_in'])).astype(np.float32)
    pos = _cloud(rng, case['n'])
    s = {'mlp.0.weight': (case['d_out'], case['d_in']), 'mlp.0.bias': (case['d_out'],)}
    if case['norm'] in ('layer', 'batch'):
        s['mlp.1.weight'] = (case['d_out'],)
        s['mlp.1.bias'] = (case['d_out'],)
    sd = cfg.fill_state_dict(s, case['seed'] + 1000)
    if case['norm'] == 'batch':
        sd['mlp.1.running_mean'] = torch.from_numpy((0.3 * rng.normal(size=(case['d_out'],))).astype(np.float32))
        sd['mlp.1.running_var'] = torch.from_numpy(rng.uniform(0.05, 2.0, size=(case['d_out'],)).astype(np.float32))
        sd['mlp.1.num_batches_tracked'] = torch.tensor(1234, dtype=torch.int64)
    return x, pos, sd


# ---------------------------------------------------------------- G5
ENC_CASES = [
    dict(name='greater_n512', kind='greater', n=512, video_len=4, seed=51),
    dict(name='carla_n512', kind='carla', n=512, video_len=4, seed=52),
    dict(name='greater_n2048', kind='greater', n=2048, video_len=4, seed=1830),
    dict(name='carla_n2048', kind='carla', n=2048, video_len=4, seed=1831),
]


# Round 4: clouds as the reference's data path hands them over when a clip has fewer than n_points points: the real
# rows followed by ALL-ZERO rows (utils/geometry.py:315-325) -- hundreds of coincident points in the self-kNN-16, the
# FPS and the max-pool.  Outputs stay well defined (identical rows: any tie order gives the same result).
ENC_PAD_CASES = [
    dict(name='greater_n2048_pad25', kind='greater', n=2048, n_real=1536, video_len=4, seed=53),
    dict(name='carla_n2048_pad25', kind='carla', n=2048, n_real=1536, video_len=4, seed=54),
]


def padded_pcl(case):
    """(1, n, 8): n_real synthetic rows then n - n_real all-zero rows (what subsample_pad_pcl_torch returns for a
    short cloud; oracle/gen_golden.py checks this against the reference function)."""
    pcl = cfg.synthetic_pcl(case['kind'], case.get('n_real', case['n']), case['video_len'], case['seed'])
    pad = case['n'] - pcl.shape[1]
    if pad > 0:
        pcl = torch.cat([pcl, torch.zeros((1, pad, pcl.shape[2]), dtype=pcl.dtype)], dim=1)
    return pcl


def enc_inputs(case):
    pa, ia, _ = cfg.model_args(case['kind'], case['n'])
    pcl = padded_pcl(case)
    sd = cfg.fill_state_dict(cfg.encoder_param_shapes(pa), case['seed'] + 1000)
    return pcl, pa, sd


# ---------------------------------------------------------------- G6
MYKNN_CASES = [
    dict(name='n400_m531_k8', n=400, m=531, k=8, e=5, seed=61),
    dict(name='n300_m2124_k8', n=300, m=2124, k=8, e=0, seed=62),
    dict(name='n64_m76_k1', n=64, m=76, k=1, e=2, seed=63),
]


def myknn_inputs(case):
    rng = _rng(case['seed'])
    q = np.concatenate([_cloud(rng, case['n']), rng.normal(size=(case['n'], 1)).astype(np.float32)], axis=1)
    key = np.concatenate([_cloud(rng, case['m']),
                          rng.normal(size=(case['m'], case['e'])).astype(np.float32)], axis=1)
    return q, key


# ---------------------------------------------------------------- G7
def posenc_inputs():
    rng = _rng(71)
    pts = rng.uniform(-40.0, 40.0, size=(512, 4)).astype(np.float32)
    pts[:, 3] = rng.integers(0, 12, size=512).astype(np.float32)
    pts[0] = [40.0, -40.0, 10.0, 11.0]
    pts[1] = [-40.0, 40.0, -1.0, 0.0]
    pts[2] = [0.0, 0.0, 0.0, 0.0]
    pts[3] = [39.869873, -19.83606, 6.103448, 3.0]
    return pts


# ---------------------------------------------------------------- G8
DEC_CASES = [
    dict(name='greater_m76_q256', kind='greater', m=76, nq=256, seed=81),
    dict(name='greater_m531_q256', kind='greater', m=531, nq=256, seed=82),
    dict(name='carla_m2124_q256', kind='carla', m=2124, nq=256, seed=83),
    dict(name='greater_m531_q1', kind='greater', m=531, nq=1, seed=84),
]


# Round 4 regimes: `attn_wscale` multiplies every rank-2 weight of the cross-attention layers (pt_blocks.*.layer2.*),
# `fscale` the abstract features and the global embedding, `qscale` pushes the queries that many times outside the
# query cuboid (inverse-distance weights 1 / (d + 1e-4) with large d, far neighbours for the attention).
# Round 4: the reference's other activation option (model/implicit.py:46-64: swish = x * sigmoid(x))
DEC_SWISH_CASES = [
    dict(name='greater_m531_q256_swish', kind='greater', m=531, nq=256, seed=90, activation='swish'),
    dict(name='carla_m2124_q256_swish', kind='carla', m=2124, nq=256, seed=91, activation='swish'),
]

# Round 5: gradients of the training path for the reference's options no published configuration trains with -- the
# swish activation (model/implicit.py:46-64) -- taken from the reference's own autograd on CPU (G15)
TRAIN_OPTION_CASES = [
    dict(name='greater_m76_swish', base=dict(name='greater_m76_q256_swish_t', kind='greater', m=76, nq=256, seed=93,
                                             activation='swish'), nq=96, seed=94),
    dict(name='carla_m300_swish', base=dict(name='carla_m300_q256_swish_t', kind='carla', m=300, nq=256, seed=95,
                                            activation='swish'), nq=64, seed=96),
]
TRAIN_OPTION_PARAMS = ['lin_out.weight', 'lin_in.bias', 'blocks.0.fc_0.weight', 'blocks.5.fc_1.bias', 'lin_z.3.weight',
                       'pt_blocks.1.layer2.to_q.weight', 'pt_blocks.0.layer2.attn_mlp.0.weight', 'pt_blocks.0.layer3.bias']


def grad_sample(a):
    """What the G15 fixtures keep of a parameter gradient: every 5th row and 3rd column of a matrix, a vector whole."""
    return a[::5, ::3] if a.ndim == 2 else a


def train_option_inputs(case):
    """(q, abstract, fglob, ia, sd, go, gp): decoder inputs + the cotangents of (output, penult) that define the scalar
    loss = sum(output * go) + sum(penult * gp)."""
    q, abstract, fglob, ia, sd = dec_inputs(case['base'])
    q = q[:case['nq']]
    rng = _rng(case['seed'])
    go = rng.normal(size=(q.shape[0], ia['d_out'])).astype(np.float32)
    gp = (0.1 * rng.normal(size=(q.shape[0], ia['d_hidden']))).astype(np.float32)
    return q, abstract, fglob, ia, sd, go, gp


# Round 5: DownTransition(norm_type='batch') in TRAINING mode (model/modules.py:98-102): batch statistics, running-stat
# update, gradients -- from the reference's own module and autograd on CPU (G16); B = 2 clouds share the statistics
DOWN_BN_TRAIN_CASES = [
    dict(name='batch_train_n200_72to144_k12', n=200, d_in=72, d_out=144, k=12, norm='batch', seed=44, batch=2),
    dict(name='batch_train_n301_36to72_k12', n=301, d_in=36, d_out=72, k=12, norm='batch', seed=45, batch=1),
]


def down_bn_train_inputs(case):
    """(x (B,n,d_in), pos (B,n,3), state_dict, cotangent of z)."""
    xs, ps = [], []
    for b in range(case['batch']):
        x, pos, sd = down_inputs(dict(case, seed=case['seed'] + 100 * b))
        xs.append(x)
        ps.append(pos)
    _, _, sd = down_inputs(case)
    rng = _rng(case['seed'] + 7)
    n_new = int(np.ceil(case['n'] / 3))
    gz = rng.normal(size=(case['batch'], n_new, case['d_out'])).astype(np.float32)
    return np.stack(xs), np.stack(ps), sd, gz


DEC_REGIME_CASES = [
    dict(name='greater_m531_q256_w4', kind='greater', m=531, nq=256, seed=85, attn_wscale=4.0, fscale=4.0),
    dict(name='greater_m531_q256_w8', kind='greater', m=531, nq=256, seed=86, attn_wscale=8.0, fscale=4.0),
    dict(name='carla_m2124_q256_w4', kind='carla', m=2124, nq=256, seed=87, attn_wscale=4.0, fscale=4.0),
    dict(name='greater_m531_q256_far', kind='greater', m=531, nq=256, seed=88, qscale=3.0),
    dict(name='carla_m2124_q256_far', kind='carla', m=2124, nq=256, seed=89, qscale=3.0),
]


def dec_inputs(case):
    """Queries inside the query cuboid, abstract cloud inside the input cuboid with
    N(0, 0.5) features (the scale the encoder emits), N(0, 0.3) global embedding."""
    rng = _rng(case['seed'])
    _, ia, inf = cfg.model_args(case['kind'])
    ia = dict(ia, activation=case.get('activation', 'relu'))
    (x0, x1), (y0, y1), (z0, z1) = cfg.input_cuboid(case['kind'])
    lo, hi = np.array([x0, y0, z0]), np.array([x1, y1, z1])
    xyz = rng.uniform(lo, hi, size=(case['m'], 3)).astype(np.float32)
    feats = (0.5 * rng.normal(size=(case['m'], ia['d_latent_local']))).astype(np.float32)
    abstract = np.concatenate([xyz, feats], axis=1)
    fglob = (0.3 * rng.normal(size=(ia['d_latent'] - ia['d_latent_local'],))).astype(np.float32)
    q = np.concatenate([rng.uniform(lo, hi, size=(case['nq'], 3)),
                        rng.integers(0, 12, size=(case['nq'], 1))], axis=1).astype(np.float32)
    sd = cfg.fill_state_dict(cfg.decoder_param_shapes(ia), case['seed'] + 1000)
    if case.get('attn_wscale') is not None:
        for name in list(sd):
            if '.layer2.' in name and sd[name].dim() == 2:
                sd[name] = sd[name] * np.float32(case['attn_wscale'])
    fs = np.float32(case.get('fscale', 1.0))
    abstract[:, 3:] *= fs
    fglob = fglob * fs
    if case.get('qscale') is not None:
        centre = (0.5 * (lo + hi)).astype(np.float32)
        q[:, :3] = centre + np.float32(case['qscale']) * (q[:, :3] - centre)
    return q, abstract, fglob, ia, sd


# Round 5: a decoder case whose abstract cloud has the STRUCTURE of the CARLA encoder's two-level output
# (model/model.py:202-228: the finer level's points first, level id 1 in the last feature channel, then the coarse level,
# level id 2 -- and every coarse point is one of the finer points, so its coordinates appear twice).  Equidistant
# neighbours at the k = 8 / k = 14 rank are systematic here; the fixture stores the lists the reference took.
DEC_TWOLEVEL_CASES = [
    dict(name='carla_twolevel_q512', kind='carla', m_fine=1593, m_coarse=531, nq=512, seed=92),
]


def dec_twolevel_inputs(case):
    q, abstract, fglob, ia, sd = dec_inputs(dict(case, m=case['m_fine'] + case['m_coarse']))
    rng = _rng(case['seed'] + 5)
    mf = case['m_fine']
    chosen = np.sort(rng.choice(mf, size=case['m_coarse'], replace=False))
    abstract[mf:, :3] = abstract[chosen, :3]
    abstract[:mf, -1] = 1.0
    abstract[mf:, -1] = 2.0
    return q, abstract, fglob, ia, sd


# ---------------------------------------------------------------- G9
GRID_CASES = [
    dict(name='greater_8192', kind='greater', num_sample=8192, min_z=-1.0, cube_bounds=5.0, time_idx=3),
    dict(name='greater_524288', kind='greater', num_sample=524288, min_z=-1.0, cube_bounds=5.0, time_idx=3),
    dict(name='carla_524288', kind='carla', num_sample=524288, min_z=-1.0, cube_bounds=16.0, time_idx=3),
    dict(name='greater_2097152', kind='greater', num_sample=2097152, min_z=-1.0, cube_bounds=5.0, time_idx=7),
    dict(name='carla_8192', kind='carla', num_sample=8192, min_z=-1.0, cube_bounds=16.0, time_idx=0),
]

# ---------------------------------------------------------------- G10
INFER_CASES = [
    dict(name='config1_greater', kind='greater', n=2048, video_len=4, num_sample=8192, batch_size=4096,
         time_idx=3, seed=1830),
    dict(name='small_carla', kind='carla', n=768, video_len=4, num_sample=2048, batch_size=1024,
         time_idx=2, seed=1832),
]


INFER_PAD_CASES = [
    dict(name='greater_pad25', kind='greater', n=2048, n_real=1536, video_len=4, num_sample=2048, batch_size=1024,
         time_idx=2, seed=1834),
    dict(name='carla_pad25', kind='carla', n=1024, n_real=768, video_len=4, num_sample=2048, batch_size=1024,
         time_idx=1, seed=1835),
]


def infer_inputs(case):
    pa, ia, inf = cfg.model_args(case['kind'], case['n'])
    pcl = padded_pcl(case)
    esd, dsd = cfg.synthetic_weights(pa, ia, case['seed'])
    return pcl, pa, ia, inf, esd, dsd


TRACK_CASES = [
    dict(name='greater_tracks_gt', kind='greater', n=768, video_len=4, num_sample=1500, batch_size=512,
         time_idx=1, seed=1840, n_instances=3, n_target=400),
]


def track_inputs(case):
    """GREATER-layout clip with per-point instance ids (some ids below the 16-point minimum, some -1) and
    a target frame (x, y, z, instance_id, view_idx, R, G, B, mark_track) for the 1-NN labelling branch."""
    pa, ia, inf = cfg.model_args(case['kind'], case['n'])
    pcl = cfg.synthetic_pcl(case['kind'], case['n'], case['video_len'], case['seed'])
    esd, dsd = cfg.synthetic_weights(pa, ia, case['seed'])
    rng = _rng(case['seed'] + 7)
    sem = rng.integers(-1, case['n_instances'], size=(case['n'], 1)).astype(np.float32)
    sem[:5] = 7.0                      # an instance with fewer than 16 points: must be ignored
    (x0, x1), (y0, y1), (z0, z1) = cfg.input_cuboid(case['kind'])
    txyz = rng.uniform([x0, y0, z0], [x1, y1, z1], size=(case['n_target'], 3))
    rest = np.concatenate([rng.integers(-1, case['n_instances'], size=(case['n_target'], 1)),
                           rng.integers(0, 3, size=(case['n_target'], 1)), rng.uniform(size=(case['n_target'], 3)),
                           rng.integers(0, 2, size=(case['n_target'], 1))], axis=1)
    target = np.concatenate([txyz, rest], axis=1).astype(np.float32)
    return pcl, sem, target, pa, ia, inf, esd, dsd


# ---------------------------------------------------------------- G12 (dataloader subsample / pad, 8(f) rank 4)
SUBSAMPLE_CASES = [
    dict(name='pad_n700_to1024', n=700, d=8, n_desired=1024, mode='random', seed=121),
    dict(name='equal_n512', n=512, d=8, n_desired=512, mode='farthest_point', seed=122),
    dict(name='random_n3000_to1024', n=3000, d=8, n_desired=1024, mode='random', seed=123),
    dict(name='fps_n3000_to1024', n=3000, d=8, n_desired=1024, mode='farthest_point', seed=124),
    dict(name='fps_n20000_to2048', n=20000, d=9, n_desired=2048, mode='farthest_point', seed=125),
    dict(name='retain_n4000_to1500', n=4000, d=9, n_desired=1500, mode='random', seed=126, retain=True, segm_idx=8),
]


def subsample_inputs(case):
    """(N, D) cloud: xyz in the GREATER cuboid, remaining columns uniform; an integer semantic-tag column when the
    case retains vehicles / pedestrians."""
    rng = _rng(case['seed'])
    pcl = rng.uniform(-5.0, 5.0, size=(case['n'], case['d'])).astype(np.float32)
    if case.get('retain'):
        pcl[:, case['segm_idx']] = rng.integers(0, 13, size=case['n']).astype(np.float32)
    return pcl


# ---------------------------------------------------------------- G13 (training-time point sampler, 8(f) rank 2)
SAMPLER_CASES = [
    dict(name='greater_none', kind='greater', bias='none', frames=3, m=2500, num_solid=512, num_air=768,
         time_idx=1, segm=False, seed=131),
    dict(name='greater_moving', kind='greater', bias='moving', frames=3, m=2500, num_solid=512, num_air=768,
         time_idx=0, segm=False, seed=132),
    dict(name='carla_all', kind='carla', bias='low_moving_vehped_ivalo_sembal', frames=3, m=5000, num_solid=640,
         num_air=960, time_idx=2, segm=True, seed=133),
    dict(name='carla_vehped_batch2', kind='carla', bias='vehped', frames=2, m=4000, num_solid=256, num_air=300,
         time_idx=1, segm=True, seed=134, batch=2),
]


def sampler_config(case):
    return dict(min_z=-1.0, cube_bounds=5.0 if case['kind'] == 'greater' else 16.0, point_occupancy_radius=0.2,
                num_solid=case['num_solid'], num_air=case['num_air'], predict_segmentation=case['segm'],
                semantic_classes=13, predict_tracking=False, data_kind=case['kind'], point_sample_bias=case['bias'],
                cube_mode=4)


def sampler_inputs(case):
    """List-T of (B, M, E) target frames + sizes + vehicle/pedestrian ids.  A static background (shared by all
    frames, jittered per frame) plus a blob that moves between frames, so that 'moving' finds dynamic regions.
    GREATER rows: x,y,z,instance,view,R,G,B,mark (E=9); CARLA rows: x,y,z,cos,instance,semantic,view,R,G,B,mark."""
    rng = _rng(case['seed'])
    B, M, T = case.get('batch', 1), case['m'], case['frames']
    carla = case['kind'] == 'carla'
    lo, hi = (np.array([-5.0, -5.0, -1.0]), np.array([5.0, 5.0, 5.0])) if not carla else \
        (np.array([-6.0, -16.0, -1.0]), np.array([42.0, 16.0, 6.4]))         # a little beyond the CARLA output cuboid
    frames, sizes = [], []
    n_blob = M // 5
    base = rng.uniform(lo, hi, size=(B, M - n_blob, 3))
    for t in range(T):
        centre = lo + (hi - lo) * (0.3 + 0.2 * t)
        blob = centre + rng.normal(scale=0.6, size=(B, n_blob, 3))
        xyz = np.concatenate([base + rng.normal(scale=0.005, size=base.shape), blob], axis=1)
        inst = rng.integers(-1, 6, size=(B, M, 1)).astype(np.float64)
        view = rng.integers(0, 3, size=(B, M, 1)).astype(np.float64)
        rgb = rng.uniform(size=(B, M, 3))
        mark = rng.integers(0, 2, size=(B, M, 1)).astype(np.float64)
        if carla:
            cos = rng.uniform(-1, 1, size=(B, M, 1))
            sem = rng.integers(0, 15, size=(B, M, 1)).astype(np.float64)          # 13, 14 map to "other"
            rows = np.concatenate([xyz, cos, inst, sem, view, rgb, mark], axis=-1)
        else:
            rows = np.concatenate([xyz, inst, view, rgb, mark], axis=-1)
        rows = rows.astype(np.float32)
        for b in range(B):                                                     # shuffled clouds, as the loader makes them
            rows[b] = rows[b][rng.permutation(M)]
        frames.append(rows)
        sizes.append(np.full((B,), M - 7 * t, dtype=np.int64))               # a few padded rows at the end
    valo = np.tile(np.array([[0, 2, 5, 0, 0, 0]], dtype=np.int64), (B, 1))
    num_valo = np.full((B,), 3, dtype=np.int64)
    return frames, sizes, valo, num_valo


# ---------------------------------------------------------------- G14 (training losses, 8(f) rank 1)
LOSS_CASES = [
    # the published GREATER training command (README.md:36): rgb_nosigmoid, density + colour + tracking
    dict(name='greater_nosigmoid', color_mode='rgb_nosigmoid', d_out=5, frames=3, batch=2, n=257, seed=141,
         density_lw=1.0, color_lw=1.0, segmentation_lw=0.0, tracking_lw=1.0),
    dict(name='greater_sigmoid', color_mode='rgb', d_out=5, frames=2, batch=1, n=300, seed=142,
         density_lw=1.0, color_lw=0.5, segmentation_lw=0.0, tracking_lw=0.25),
    # the published CARLA training command (README.md:41): density + segmentation
    dict(name='carla_segm', color_mode='rgb', d_out=18, frames=4, batch=1, n=311, seed=143,
         density_lw=1.0, color_lw=0.0, segmentation_lw=0.6, tracking_lw=0.0),
    dict(name='carla_all_terms', color_mode='rgb_nosigmoid', d_out=18, frames=2, batch=2, n=200, seed=144,
         density_lw=0.7, color_lw=0.9, segmentation_lw=0.6, tracking_lw=0.3),
]


# Round 4: the colour losses of the reference's other two colour modes (loss.py:85-149; utils.get_track_idx: the tracking
# logit sits at channel 15 / 10).  'hsv': 12 hue bins (CE / 2, only where saturation and value >= 0.2 and only when at
# least 16 such points exist) + L1 on saturation and value; 'bins': 6 saturated colours + black / gray / white (CE / 3).
LOSS_COLOR_CASES = [
    dict(name='greater_hsv', color_mode='hsv', d_out=16, frames=3, batch=2, n=257, seed=145,
         density_lw=1.0, color_lw=1.0, segmentation_lw=0.0, tracking_lw=1.0),
    dict(name='carla_hsv_segm', color_mode='hsv', d_out=29, frames=2, batch=1, n=300, seed=146,
         density_lw=0.7, color_lw=0.9, segmentation_lw=0.6, tracking_lw=0.3),
    dict(name='hsv_too_few_hues', color_mode='hsv', d_out=16, frames=2, batch=1, n=24, seed=147,
         density_lw=1.0, color_lw=1.0, segmentation_lw=0.0, tracking_lw=0.0),
    dict(name='greater_bins', color_mode='bins', d_out=11, frames=3, batch=2, n=257, seed=148,
         density_lw=1.0, color_lw=1.0, segmentation_lw=0.0, tracking_lw=1.0),
    dict(name='carla_bins_segm', color_mode='bins', d_out=24, frames=2, batch=1, n=300, seed=149,
         density_lw=0.7, color_lw=0.9, segmentation_lw=0.6, tracking_lw=0.3),
]


def loss_inputs(case):
    """Raw decoder outputs (T, B, N, G) (logits, colour channels spread beyond [0, 1] so that the clamp of
    rgb_nosigmoid matters) and targets (T, B, N, 6) = (density, R, G, B, mark_track, segm) with the -1 "not
    available" markers the data loaders produce: missing colours, untracked points, unlabelled points."""
    rng = _rng(case['seed'])
    T, B, N, G = case['frames'], case['batch'], case['n'], case['d_out']
    raw = rng.normal(scale=1.5, size=(T, B, N, G)).astype(np.float32)
    dens = (rng.uniform(size=(T, B, N, 1)) < 0.45).astype(np.float32)
    rgb = rng.uniform(size=(T, B, N, 3)).astype(np.float32)
    rgb[rng.uniform(size=(T, B, N)) < 0.2] = -1.0                 # colour not available
    rgb = np.where(dens > 0.5, rgb, 0.0).astype(np.float32)       # air rows carry zeros
    mark = rng.integers(-1, 2, size=(T, B, N, 1)).astype(np.float32)
    segm = rng.integers(-1, 13, size=(T, B, N, 1)).astype(np.float32)
    target = np.concatenate([dens, rgb, mark, segm], axis=-1).astype(np.float32)
    return raw, target


def loss_kwargs(case):
    return {k: case[k] for k in ('density_lw', 'color_lw', 'segmentation_lw', 'tracking_lw', 'color_mode')}


def regime_bound(golden, key, floor=1e-4):
    """Parity bound for the scaled-weight regime fixtures: max(floor, 2 max|ref32 - ref64|), ref32 / ref64 = the
    reference's own fp32 and fp64 results (`key`, `key + '64'`).  At these scales no fp32 op order can promise 1e-4
    absolute (outputs reach 1e2); an implementation has to be as close to the fp64 value of the reference's formula as
    the reference's own fp32 run is, within a factor of two."""
    return max(floor, 2.0 * float(np.abs(golden[key].astype(np.float64) - golden[key + '64']).max()))


def as_tensor(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------- G17: reference-format checkpoints (eval/inference.py:23-80)
# Small-width networks saved by oracle/gen_golden.py with the REFERENCE's own modules and checkpoint layout
# (train.py:339-350: optimizer / lr_scheduler / scaler / epoch / args / pcl_args / dset_args / implicit_args / pcl_net /
# implicit_net) under tests/golden/<dir>/: `checkpoint.pth` with today's parameter names and `model_<epoch>.pth` of a
# one-cross-layer decoder with the legacy `pt_block.` prefix (utils/utils.py:127-135), plus the reference's
# load_models -> perform_inference outputs for both.
CKPT_DIR = 'g17_ckpt_greater_small'
CKPT_CASES = [
    dict(name='current', file='checkpoint.pth', epoch_arg=-1, epoch=7, cross_attn_layers=2, legacy=False, seed=1850),
    dict(name='legacy', file='model_3.pth', epoch_arg=3, epoch=3, cross_attn_layers=1, legacy=True, seed=1851),
]
CKPT_INFER = dict(kind='greater', n=768, video_len=4, num_sample=1024, batch_size=512, time_idx=2, seed=1852)


def ckpt_model_args(case):
    """Small-width constructor kwargs with the structure train.py:194-265 gives them (d_latent_local = d_feat * 2^blocks,
    d_hidden = d_latent = global + local); fps_random_start is True in a training checkpoint (args.py default), which
    load_models must override."""
    pa, ia, inf = cfg.model_args('greater', CKPT_INFER['n'])
    d_feat, g = 4, 16
    pa.update(d_feat=d_feat, global_dim=g, fps_random_start=True)
    d_local = d_feat * 2 ** pa['down_blocks']
    ia.update(d_hidden=g + d_local, d_latent=g + d_local, d_latent_local=d_local,
              cross_attn_layers=case['cross_attn_layers'], cr_attn_type='c' * case['cross_attn_layers'])
    return pa, ia, inf
