"""Hot-path geometry helpers on the HIP library.

Interface mirror of the two hot functions of the reference's utils/geometry.py:
``my_knn_torch`` (:458-503) and ``sample_implicit_points_blind_numpy`` (:1199-1283).
Everything else in that file (camera / lidar transforms, guided samplers, cuboid
filters) is data preparation or training-only and out of scope.
"""
import numpy as np
import torch

from . import ops


def my_knn_torch(pcl_query, pcl_key, num_neighbors, bidirectional=False,
                 return_inds=False, return_knn=True, return_dists=False):
    """For each query row (x,y,z,*) the K nearest key rows by 3-D Euclidean distance.
    Returns a tuple with, in order and as requested: inds (N,K) int64, the gathered key rows
    (N,K,E), dists (N,K).  Streaming top-k kernel; distance arithmetic bit-identical to the
    reference's torch.linalg.norm on CPU (sqrt of an x,y,z fused-multiply-add chain)."""
    assert return_inds or return_knn or return_dists
    assert pcl_query.dim() == 2 and pcl_key.dim() == 2
    if bidirectional:
        raise NotImplementedError()
    idx, dist = ops.knn(pcl_query, pcl_key, num_neighbors, metric=1, return_dist=True)
    result = tuple()
    if return_inds:
        result += (idx.to(torch.int64), )
    if return_knn:
        rows = ops.gather_rows(pcl_key, idx.view(-1))
        result += (rows.view(idx.shape[0], num_neighbors, pcl_key.shape[1]), )
    if return_dists:
        result += (dist, )
    return result


def _query_bounds(min_z, cube_bounds, data_kind, cube_mode):
    cb = cube_bounds
    if data_kind == 'greater':
        return (-cb, cb), (-cb, cb), (min_z, cb)
    if data_kind == 'carla':
        scale = {1: (2.0, 1.0, 0.5), 2: (2.4, 0.8, 0.4), 3: (2.2, 1.0, 0.4), 4: (2.5, 1.0, 0.4)}[cube_mode]
        return (0.0, cb * scale[0]), (-cb * scale[1], cb * scale[1]), (min_z, cb * scale[2])
    raise ValueError(data_kind)


def sample_implicit_points_blind_numpy(num_sample, min_z, cube_bounds, time_idx, data_kind,
                                       cube_mode, point_sample_mode):
    """(N,4) float32 query points (x,y,z,t) inside the output cuboid: uniformly random, or a
    cell-centred grid whose per-axis counts are ceil(cbrt(num_sample / volume) * extent)
    (x slowest, z fastest); t is the constant time_idx."""
    bounds = _query_bounds(min_z, cube_bounds, data_kind, cube_mode)
    ext = [hi - lo for lo, hi in bounds]
    if point_sample_mode == 'random':
        cols = [np.random.rand(num_sample).astype(np.float32) * e + lo for (lo, _), e in zip(bounds, ext)]
        xyz = np.stack(cols, axis=-1)
        n = num_sample
    elif point_sample_mode == 'grid':
        density = np.cbrt(num_sample / (ext[0] * ext[1] * ext[2]))
        counts = [int(np.ceil(density * e)) for e in ext]
        axes = [(np.arange(c, dtype=np.float32) + 0.5) * (e / c) + lo
                for c, e, (lo, _) in zip(counts, ext, bounds)]
        nx, ny, nz = counts
        n = nx * ny * nz
        xyz = np.empty((nx, ny, nz, 3), dtype=np.float32)
        xyz[..., 0] = axes[0][:, None, None]
        xyz[..., 1] = axes[1][None, :, None]
        xyz[..., 2] = axes[2][None, None, :]
        xyz = xyz.reshape(n, 3)
    else:
        raise ValueError(point_sample_mode)
    t = np.full((n, 1), time_idx, dtype=np.float32)
    return np.concatenate([xyz, t], axis=-1)
