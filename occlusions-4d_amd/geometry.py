"""Hot-path geometry helpers on the HIP library.

Interface mirror of the hot functions of the reference's utils/geometry.py:
``my_knn_torch`` (:458-503), ``sample_implicit_points_blind_numpy`` (:1199-1283) and the
dataloader's ``subsample_pad_pcl_torch`` (:294-376).
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


def subsample_pad_pcl_torch(pcl, n_desired, sample_mode='random', subsample_only=False,
                            retain_vehped=False, segm_idx=None):
    """Zero-pads a too-small point cloud (B,N,D) / (N,D) to n_desired rows, or subsamples a too-large one:
    uniformly at random (numpy's global stream, ascending indices) or by farthest point sampling over xyz with a
    random first sample (torch's global CPU generator) -- the cooperative multi-workgroup FPS kernel, N <= 262144
    (a 12-frame clip of 172 K points -> 14 336 in ~50 ms where the reference spends seconds per clip in CPU
    dataloader workers).  retain_vehped keeps semantic tags 4 and 10 and samples the rest (reference masks).
    CUDA tensors only."""
    assert sample_mode in ['random', 'farthest_point']
    no_batch = (len(pcl.shape) == 2)
    if no_batch:
        pcl = pcl.unsqueeze(0)
    (B, N, D) = pcl.shape
    if N < n_desired:
        if subsample_only:
            raise RuntimeError('Too few input points: ' + str(N) + ' vs ' + str(n_desired) + '.')
        zeros = torch.zeros((B, n_desired - N, D), dtype=pcl.dtype, device=pcl.device)
        result = torch.cat((pcl, zeros), axis=1)
        return result.squeeze(0) if no_batch else result
    if N == n_desired:
        return pcl.squeeze(0) if no_batch else pcl
    assert B == 1
    n_remain = n_desired
    if retain_vehped:
        tags = pcl[0, :, segm_idx].cpu().numpy()
        retain_inds = np.where(np.logical_or(tags == 4, tags == 10))[0]
        remain_inds = np.where(tags != 10)[0]
        n_remain -= retain_inds.shape[0]
    else:
        remain_inds = np.arange(N)
    flat = pcl[0]
    if sample_mode == 'random':
        inds = np.random.choice(remain_inds, n_remain, replace=False)
        inds.sort()
        inds = torch.from_numpy(inds.astype(np.int32)).to(pcl.device)
    else:
        assert not retain_vehped
        # torch_cluster's sample count: ceil(float32(N) * float32(ratio)), ratio = n_remain / N - 1e-7
        m = int(np.ceil(np.float32(N) * np.float32(n_remain / N - 1e-7)))
        assert m == n_remain, 'fps ratio does not reproduce n_desired (the reference would fail in .view)'
        start = int(torch.randint(N, (1,)).item())
        inds = ops.fps_coop(flat[:, :3], m, start=start)             # ascending int32
    result = ops.gather_rows(flat, inds).view(B, n_remain, D)
    if no_batch:
        result = result.squeeze(0)
    if retain_vehped:
        keep = torch.from_numpy(retain_inds.astype(np.int32)).to(pcl.device)
        result = torch.cat([ops.gather_rows(flat, keep), result], dim=0)
    assert result.shape[0] == n_desired
    return result


def _query_bounds(min_z, cube_bounds, data_kind, cube_mode):
    cb = cube_bounds
    if data_kind == 'greater':
        return (-cb, cb), (-cb, cb), (min_z, cb)
    if data_kind == 'carla':
        scale = {1: (2.0, 1.0, 0.5), 2: (2.4, 0.8, 0.4), 3: (2.2, 1.0, 0.4), 4: (2.5, 1.0, 0.4)}[cube_mode]
        return (0.0, cb * scale[0]), (-cb * scale[1], cb * scale[1]), (min_z, cb * scale[2])
    raise ValueError(data_kind)


def _grid_counts(num_sample, ext):
    density = np.cbrt(num_sample / (ext[0] * ext[1] * ext[2]))
    return [int(np.ceil(density * e)) for e in ext]


def sample_implicit_points_blind_device(num_sample, min_z, cube_bounds, time_idx, data_kind, cube_mode,
                                        point_sample_mode, device):
    """sample_implicit_points_blind_numpy with the result resident on `device`: the grid is generated by
    a kernel (bit-identical to the host grid: same fp32 operation order), so a 0.5 M-query frame costs
    no host work and no 8.5 MB upload.  'random' draws on the host (numpy's global stream, as the
    reference does) and uploads."""
    if point_sample_mode != 'grid':
        pts = sample_implicit_points_blind_numpy(num_sample, min_z, cube_bounds, time_idx, data_kind, cube_mode,
                                                 point_sample_mode)
        return torch.from_numpy(pts).to(device)
    bounds = _query_bounds(min_z, cube_bounds, data_kind, cube_mode)
    ext = [hi - lo for lo, hi in bounds]
    counts = _grid_counts(num_sample, ext)
    return ops.grid_points(counts, [lo for lo, _ in bounds], [e / c for e, c in zip(ext, counts)], time_idx, device)


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
        counts = _grid_counts(num_sample, ext)
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
