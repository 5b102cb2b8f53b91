"""CPU restatement of the training-time point sampler.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows utils/geometry.py of the reference:
``GuidedImplicitPointSampler`` (:578-1105), ``sample_random_uniform_3ball`` (:562-575),
``sample_implicit_points_blind_torch`` (:1108-1161), ``filter_air_solid_gap`` (:1164-1196),
``filter_pcl_bounds_carla_output_torch`` (:224-260), ``get_vehped_points`` (:1323-1332).

The sampler is random; it is pinned by replaying the reference's draws: every random number
comes from torch's or numpy's GLOBAL CPU generator, in the reference's call order, so that the
same seeds give the same supervision points bit for bit (tests/golden g13, produced by the
reference class itself on CPU).  Plain PyTorch-CPU, written from scratch as free functions.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import path as op

LOW_Z = (0.0, 2.0)            # 'low' bias prefers this height band (:614-615)


def ball_offsets(count, max_radius, min_radius=0.0):
    """Uniform points in a 3-ball shell (:562-575): direction = normalised torch.randn draw,
    radius = cbrt of a numpy uniform draw scaled into [min_radius, max_radius]."""
    direction = F.normalize(torch.randn(count, 3, dtype=torch.float32), p=2, dim=-1)
    radius = torch.tensor(np.cbrt(np.random.rand(count).astype(np.float32)))
    radius = radius * (max_radius - min_radius) + min_radius
    return direction * radius[:, None]


def carla_output_mask(pcl, min_z, bounds, cube_mode):
    """Rows inside the CARLA output cuboid (:224-260, :175-188)."""
    sx, sy, sz = {1: (2.0, 1.0, 0.5), 2: (2.4, 0.8, 0.4), 3: (2.2, 1.0, 0.4), 4: (2.5, 1.0, 0.4)}[cube_mode]
    x, y, z = pcl[..., 0], pcl[..., 1], pcl[..., 2]
    mx = torch.logical_and(0.0 <= x, x <= bounds * sx)
    my = torch.logical_and(-bounds * sy <= y, y <= bounds * sy)
    mz = torch.logical_and(min_z <= z, z <= bounds * sz)
    return torch.logical_and(torch.logical_and(mx, my), mz)


def vehped_rows(pcl, segm_idx):
    """Pedestrian (tag 4) rows followed by vehicle (tag 10) rows (:1323-1332)."""
    return torch.cat([pcl[pcl[..., segm_idx] == 4], pcl[pcl[..., segm_idx] == 10]], dim=0)


def blind_points(data_kind, count, cube_mode, bounds, min_z):
    """Uniform xyz in the output cuboid (:1108-1161); draw order: GREATER (n,2) then (n,1); CARLA x, y, z."""
    if data_kind == 'greater':
        xy = torch.rand((count, 2)) * bounds * 2.0 - bounds
        z = torch.rand((count, 1)) * (bounds - min_z) + min_z
        return torch.cat([xy, z], dim=-1)
    if data_kind == 'carla':
        sx, ys, yo, sz = {1: (2.0, 2.0, 1.0, 0.5), 2: (2.4, 1.6, 0.8, 0.4), 3: (2.2, 2.0, 1.0, 0.4),
                          4: (2.5, 2.0, 1.0, 0.4)}[cube_mode]
        x = torch.rand((count, 1)) * bounds * sx
        y = torch.rand((count, 1)) * bounds * ys - bounds * yo
        z = torch.rand((count, 1)) * (bounds * sz - min_z) + min_z
        return torch.cat([x, y, z], dim=-1)
    raise ValueError()


def air_solid_gap(rows, target_xyz, radius):
    """Keep the rows whose nearest target point is farther than `radius` (:1164-1196).  The reference
    takes the minimum over target slices of the 1-NN distance; the minimum over slices is the global
    1-NN distance, computed here in one pass.  Returns (rows kept, their distances, kept fraction)."""
    _, dist = op.knn_with_dists(rows, target_xyz, 1)
    dist = dist[:, 0]
    good = dist > radius
    return rows[good], dist[good], good.sum() / dist.shape[0]


def first_rows(t, count):
    """First `count` rows, doubling the tensor while it is too short (:1094-1105)."""
    while t.shape[0] < count:
        t = torch.cat([t, t], dim=0)
    return t[:count].clone()


class SamplerConfig:
    def __init__(self, min_z=-1.0, cube_bounds=10.0, point_occupancy_radius=0.25, num_solid=1024, num_air=1024,
                 predict_segmentation=False, semantic_classes=13, predict_tracking=False, data_kind='',
                 point_sample_bias='none', cube_mode=4):
        self.__dict__.update(locals())
        del self.__dict__['self']


def solid_pairs(cfg, tgt, tgt_unique, valo_ids, time_idx):
    """Solid query points + targets for one frame (:765-930)."""
    carla = cfg.data_kind == 'carla'
    inst_idx, segm_idx, view_idx = (4, 5, 6) if carla else (3, 3, 4)
    bias = cfg.point_sample_bias
    shares = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, 0.0])     # regular, low, moving, vehped, ivalo, sembal
    if 'low' in bias:
        low = tgt[torch.logical_and(LOW_Z[0] <= tgt[..., 2], tgt[..., 2] <= LOW_Z[1])]
        if low.shape[0] >= 256:
            shares[1] += 1.0
    if 'moving' in bias:
        if tgt_unique.shape[0] >= 256:
            shares[2] += 0.4
        elif tgt_unique.shape[0] >= 16:
            shares[2] += tgt_unique.shape[0] * 0.4 / 256.0
    if 'vehped' in bias:
        assert carla
        vehped = vehped_rows(tgt, segm_idx)
        if vehped.shape[0] >= 256:
            shares[3] += 0.2
        elif vehped.shape[0] >= 16:
            shares[3] += vehped.shape[0] * 0.2 / 256.0
    if 'ivalo' in bias:
        assert carla
        if len(valo_ids) > 0:
            visible = vehped_rows(tgt[tgt[..., view_idx] == 0], segm_idx)
            vis_ids = sorted(list(visible[..., inst_idx].type(torch.int32).unique().numpy()))
            hidden = vehped_rows(tgt[tgt[..., view_idx] != 0], segm_idx)
            parts = []
            for vid in valo_ids:
                rows = hidden[hidden[..., inst_idx] == vid]
                parts.append(rows)
                if vid not in vis_ids:          # fully occluded instance: counted twice
                    parts.append(rows)
            ivalo = torch.cat(parts, dim=0)
            if ivalo.shape[0] >= 256:
                shares[4] += 0.2
            elif ivalo.shape[0] >= 16:
                shares[4] += min(ivalo.shape[0] * 0.2 / 256.0, 0.2)
    if 'sembal' in bias:
        assert carla
        shares[5] += 0.4
    shares /= shares.sum()

    pool = []
    counts = [int(shares[i] * cfg.num_solid) for i in range(1, 6)]
    n_low, n_moving, n_vehped, n_ivalo, n_sembal = counts
    if n_low > 0:
        pool.append(low[torch.randint(0, low.shape[0], (n_low, ))])
    if n_moving > 0:
        pool.append(tgt_unique[torch.randint(0, tgt_unique.shape[0], (n_moving, ))])
    if n_vehped > 0:
        pool.append(vehped[torch.randint(0, vehped.shape[0], (n_vehped, ))])
    if n_ivalo > 0:
        pool.append(ivalo[torch.randint(0, ivalo.shape[0], (n_ivalo, ))])
    if n_sembal > 0:
        ids = list(tgt[..., segm_idx].type(torch.int32).unique().numpy())
        used = 0
        for sid in ids:
            rows = tgt[tgt[..., segm_idx] == sid]
            if rows.shape[0] >= 16:
                per = n_sembal // len(ids)
                pool.append(rows[torch.randint(0, rows.shape[0], (per, ))])
                used += per
        n_sembal = used
    n_regular = cfg.num_solid - n_low - n_moving - n_vehped - n_ivalo - n_sembal
    if n_regular > 0:
        pool.append(tgt[torch.randint(0, tgt.shape[0], (n_regular, ))])
    chosen = torch.cat(pool, dim=0)
    assert chosen.shape[0] == cfg.num_solid
    xyz = chosen[..., :3] + ball_offsets(cfg.num_solid, cfg.point_occupancy_radius / 2.0)
    query = torch.cat([xyz, torch.ones_like(xyz[..., 0:1]) * time_idx], dim=-1)
    target = torch.cat([torch.ones_like(xyz[..., 0:1]), chosen[..., -4:]], dim=-1)     # density, R, G, B, mark
    if cfg.predict_segmentation:
        segm = chosen[..., segm_idx:segm_idx + 1].clone()
        segm[segm >= cfg.semantic_classes] = 3
        target = torch.cat([target, segm], dim=-1)
    else:
        target = torch.cat([target, -torch.ones_like(target[..., 0:1])], dim=-1)
    return query, target, shares


def air_pairs(cfg, tgt, other_unique, solid_query, time_idx):
    """Air query points + targets for one frame (:932-1092)."""
    r = cfg.point_occupancy_radius
    tgt_xyz = tgt[..., :3]
    shares = torch.tensor([0.5, 0.0, 0.3, 0.2])               # regular, moving, near solid queries, near target
    if 'moving' in cfg.point_sample_bias:
        if other_unique.shape[0] >= 256:
            shares[1] += 0.4
        elif other_unique.shape[0] >= 16:
            shares[1] += other_unique.shape[0] * 0.4 / 256.0
    shares /= shares.sum()
    points, dists = [], []

    def keep(cand, count):
        kept, d, _ = air_solid_gap(cand, tgt_xyz, r)
        points.append(first_rows(kept, count))
        dists.append(first_rows(d, count))

    n_moving = int(shares[1] * cfg.num_air)
    if n_moving > 0:
        draw = int(n_moving * 1.6)
        cand = other_unique[torch.randint(0, other_unique.shape[0], (draw, ))][..., :3]
        keep(cand + ball_offsets(draw, r * 2.0), n_moving)
    n_hsq = int(shares[2] * cfg.num_air)
    if n_hsq > 0:
        draw = int(n_hsq * 2.0)
        cand = solid_query[torch.randint(0, solid_query.shape[0], (draw, ))][..., :3]
        keep(cand + ball_offsets(draw, max_radius=r * 3.0, min_radius=r), n_hsq)
    n_ht = int(shares[3] * cfg.num_air)
    if n_ht > 0:
        draw = int(n_ht * 2.0)
        cand = tgt[torch.randint(0, tgt.shape[0], (draw, ))][..., :3]
        keep(cand + ball_offsets(draw, max_radius=r * 3.0, min_radius=r), n_ht)
    n_regular = cfg.num_air - n_moving - n_hsq - n_ht
    if n_regular > 0:
        draw = int(n_regular * (1.3 if cfg.data_kind == 'greater' else 1.1))
        keep(blind_points(cfg.data_kind, draw, cfg.cube_mode, cfg.cube_bounds, cfg.min_z), n_regular)
    xyz = torch.cat(points, dim=0)
    assert xyz.shape[0] == cfg.num_air
    query = torch.cat([xyz, torch.ones_like(xyz[..., 0:1]) * time_idx], dim=-1)
    target = -torch.ones((cfg.num_air, 6), dtype=tgt.dtype)
    target[..., 0] = 0.0
    return query, target, shares, torch.cat(dists, dim=0)


def sample_frame(cfg, pcl_target, pcl_target_size, valo_ids, num_valo_ids, time_idx):
    """GuidedImplicitPointSampler.forward (:617-763): list-T of (B,M,E) target frames -> (solid_input (B,S,4),
    air_input (B,A,4), solid_target (B,S,6), air_target (B,A,6), solid_sbs (B,6), air_sbs (B,4))."""
    frame, sizes = pcl_target[time_idx], pcl_target_size[time_idx]
    (B, M, E) = frame.shape
    assert torch.all(sizes <= M)
    assert E == (9 if cfg.data_kind == 'greater' else 11)
    other = other_sizes = None
    if len(pcl_target) > 1:
        other_time = np.random.randint(len(pcl_target) - 1)
        if other_time == time_idx:
            other_time += 1
        other, other_sizes = pcl_target[other_time], pcl_target_size[other_time]
    outs = [[] for _ in range(6)]
    for i in range(B):
        tgt = frame[i, :int(sizes[i])]
        ids = sorted(list(valo_ids[i, :int(num_valo_ids[i])].numpy()))
        if cfg.data_kind == 'carla':
            tgt = tgt[carla_output_mask(tgt, cfg.min_z, cfg.cube_bounds, cfg.cube_mode)]
        if tgt.shape[0] < 256:
            raise RuntimeError('Invalid due to cur_tgt_pcl_count: %d' % tgt.shape[0])
        tgt_unique = other_unique = None
        if 'moving' in cfg.point_sample_bias:
            oth = other[i, :int(other_sizes[i])]
            if cfg.data_kind == 'carla':
                oth = oth[carla_output_mask(oth, cfg.min_z, cfg.cube_bounds, cfg.cube_mode)]
            # the reference checks the TARGET's count here for CARLA and the other frame's for GREATER (:704-706)
            check = tgt.shape[0] if cfg.data_kind == 'carla' else int(other_sizes[i])
            if check < 256:
                raise RuntimeError('Invalid due to cur_other_pcl_count: %d' % check)
            max_slice = int((2 ** 27) // cfg.num_air)
            n_slices = int(np.ceil(tgt.shape[0] / max_slice))
            used = tgt.shape[0] // n_slices + 1
            tgt_sub, oth_sub = tgt[:used], oth[:used]       # dynamic regions from (shuffled) sub-clouds
            tgt_unique = air_solid_gap(tgt_sub, oth_sub[..., :3], cfg.point_occupancy_radius * 2.0)[0]
            other_unique = air_solid_gap(oth_sub, tgt_sub[..., :3], cfg.point_occupancy_radius * 2.0)[0]
        sq, st, ss = solid_pairs(cfg, tgt, tgt_unique, ids, time_idx)
        aq, at, as_, _ = air_pairs(cfg, tgt, other_unique, sq, time_idx)
        for lst, v in zip(outs, (sq, aq, st, at, ss, as_)):
            lst.append(v)
    return tuple(torch.stack(v) for v in outs)
