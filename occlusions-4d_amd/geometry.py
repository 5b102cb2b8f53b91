"""Hot-path geometry helpers on the HIP library.

Interface mirror of the hot functions of the reference's utils/geometry.py:
``my_knn_torch`` (:458-503), ``sample_implicit_points_blind_numpy`` (:1199-1283) and the
dataloader's ``subsample_pad_pcl_torch`` (:294-376).
Everything else in that file (camera / lidar transforms, guided samplers, cuboid
filters) is data preparation or training-only and out of scope.
"""
import os
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


# --------------------------------------------------------------------------------------------------------------
# Training-time point sampler: utils/geometry.py:562-1196 of the reference.
# The heavy part -- 1-NN distances of up to ~20 K candidates against ~57 K target points and the selection that
# follows -- runs on the kNN and compaction kernels; index draws, bias bookkeeping and tensor glue stay on the
# host / in torch.  Every random number comes from torch's or numpy's GLOBAL CPU generator in the reference's
# call order (the reference draws the uniform air points from the device generator when it runs on a GPU; on
# CPU it uses the same CPU generator as here), so equal seeds reproduce the reference's CPU results bit for bit.
# --------------------------------------------------------------------------------------------------------------
def sample_random_uniform_3ball(num_points, max_radius, min_radius=0.0):
    """(N,3) CPU tensor of points uniform in the ball shell min_radius <= |v| <= max_radius."""
    uvw = torch.nn.functional.normalize(torch.randn(num_points, 3, dtype=torch.float32), p=2, dim=-1)
    radius = torch.tensor(np.cbrt(np.random.rand(num_points).astype(np.float32)))
    radius = radius * (max_radius - min_radius) + min_radius
    return uvw * radius[:, None]


def filter_pcl_bounds_torch(pcl, x_min=-10.0, x_max=10.0, y_min=-10.0, y_max=10.0, z_min=-10.0, z_max=10.0):
    mask_x = torch.logical_and(x_min <= pcl[..., 0], pcl[..., 0] <= x_max)
    mask_y = torch.logical_and(y_min <= pcl[..., 1], pcl[..., 1] <= y_max)
    mask_z = torch.logical_and(z_min <= pcl[..., 2], pcl[..., 2] <= z_max)
    return pcl[torch.logical_and(torch.logical_and(mask_x, mask_y), mask_z)]


_CARLA_OUTPUT_SCALE = {1: (2.0, 1.0, 0.5), 2: (2.4, 0.8, 0.4), 3: (2.2, 1.0, 0.4), 4: (2.5, 1.0, 0.4)}


def filter_pcl_bounds_carla_output_torch(pcl, min_z=-0.5, other_bounds=16.0, padding=0.0, cube_mode=4):
    sx, sy, sz = _CARLA_OUTPUT_SCALE[cube_mode]
    return filter_pcl_bounds_torch(pcl, x_min=0.0 - padding, x_max=other_bounds * sx + padding,
                                   y_min=-other_bounds * sy - padding, y_max=other_bounds * sy + padding,
                                   z_min=min_z, z_max=other_bounds * sz)


def get_vehped_points(pcl, segm_idx):
    return torch.cat([pcl[pcl[..., segm_idx] == 4], pcl[pcl[..., segm_idx] == 10]], dim=0)


def sample_implicit_points_blind_torch(data_kind, num_sample, cube_mode, cube_bounds, min_z, device):
    """(N,3) uniform xyz inside the output cuboid, drawn on the host generator and uploaded."""
    if data_kind == 'greater':
        xy = torch.rand((num_sample, 2)) * cube_bounds * 2.0 - cube_bounds
        z = torch.rand((num_sample, 1)) * (cube_bounds - min_z) + min_z
        pts = torch.cat([xy, z], dim=-1)
    elif data_kind == 'carla':
        if cube_mode not in _CARLA_OUTPUT_SCALE:
            raise ValueError()
        sx, sy, sz = _CARLA_OUTPUT_SCALE[cube_mode]
        x = torch.rand((num_sample, 1)) * cube_bounds * sx
        y = torch.rand((num_sample, 1)) * cube_bounds * (2.0 * sy) - cube_bounds * sy
        z = torch.rand((num_sample, 1)) * (cube_bounds * sz - min_z) + min_z
        pts = torch.cat([x, y, z], dim=-1)
    else:
        raise ValueError()
    return pts.to(device)


def filter_air_solid_gap(to_filter, target_coords, target_slice_size, point_occupancy_radius):
    """Rows of to_filter (N,D) whose nearest target point is farther than the radius -> (rows kept, their 1-NN
    distances, kept fraction).  The reference bounds memory by taking the minimum over target slices; the
    streaming kNN kernel never materialises N x M, so target_slice_size is accepted and unused (the minimum over
    slices IS the global 1-NN distance)."""
    _, dist = ops.knn(to_filter, target_coords, 1, metric=1, return_dist=True)
    dist = dist[:, 0]
    kept, kept_dist = ops.compact_rows(to_filter, dist, point_occupancy_radius, strict=True)
    ratio = torch.tensor(kept.shape[0] / max(1, to_filter.shape[0]), device=to_filter.device)
    return (kept, kept_dist, ratio)


# The sampler itself only uses the ROWS the gap filter keeps (utils/geometry.py:692, :956-1010 discard the distances), i.e.
# the decision "no target point within the radius": a uniform-grid radius test (ops.RadiusGrid, csrc/gridrad.hip) makes
# the same decisions as the 1-NN search bit for bit and visits a few hundred targets per candidate instead of all 57 K.
# '0': the 1-NN search (filter_air_solid_gap above, which also returns the distances).
GRID_GAP_FILTER = os.environ.get('OCC4D_GRID_GAP_FILTER', '1') == '1'


def _gap_rows(to_filter, target_coords, radius, grid=None):
    """filter_air_solid_gap(...)[0]: the rows of to_filter with no target point within `radius`.  `grid`: a RadiusGrid
    of target_coords built for a radius >= `radius` (reused across the calls of one frame)."""
    if not GRID_GAP_FILTER:
        return filter_air_solid_gap(to_filter, target_coords, 0, radius)[0]
    if grid is None:
        grid = ops.RadiusGrid(target_coords, radius)
    far = grid.far(to_filter, radius)
    return ops.compact_rows(to_filter, far, 0.5, strict=True)[0]


# '1' (default): GuidedImplicitPointSampler reads device-side counts three times per batch element instead of once per
# selection (same draws, same points: tests/test_gpu_sampler.py); '0': the step-by-step path everywhere.
SAMPLER_FAST = os.environ.get('OCC4D_SAMPLER_FAST', '1') == '1'
_SEGM_BINS = 64            # semantic ids the fast path groups rows by (CARLA has 23); others take the step-by-step path


def _upload(cpu_tensor, device):
    """Host->device copy of a freshly drawn CPU tensor (pageable: staging it through pinned memory costs more than the
    copy of these ~100 KB)."""
    return cpu_tensor.to(device)


def _take(rows, cpu_inds):
    """rows[cpu_inds] for a CPU LongTensor of indices, on the gather kernel."""
    return ops.gather_rows(rows, cpu_inds.to(torch.int32).to(rows.device))


class GuidedImplicitPointSampler(torch.nn.Module):
    """Training-time sampler of solid / air supervision points (no learnable parameters).  Same constructor and
    forward() as the reference class; target point clouds are CUDA tensors."""

    def __init__(self, logger, min_z=-1.0, cube_bounds=10.0, point_occupancy_radius=0.25, num_solid=1024,
                 num_air=1024, predict_segmentation=False, semantic_classes=13, predict_tracking=False,
                 data_kind='', point_sample_bias='none', cube_mode=4):
        super().__init__()
        self.logger = logger
        self.min_z, self.cube_bounds = min_z, cube_bounds
        self.point_occupancy_radius = point_occupancy_radius
        self.num_solid, self.num_air = num_solid, num_air
        self.predict_segmentation, self.semantic_classes = predict_segmentation, semantic_classes
        self.predict_tracking = predict_tracking
        self.data_kind, self.point_sample_bias, self.cube_mode = data_kind, point_sample_bias, cube_mode
        self.low_prefer_min_z, self.low_prefer_max_z = 0.0, 2.0

    def forward(self, pcl_target, pcl_target_size, valo_ids, num_valo_ids, time_idx):
        """pcl_target: list-T of (B,M,E); returns (solid_input (B,S,4), air_input (B,A,4), solid_target (B,S,6),
        air_target (B,A,6), solid_sbs (B,6), air_sbs (B,4))."""
        frame, sizes = pcl_target[time_idx], pcl_target_size[time_idx]
        (B, M, E) = frame.shape
        if self.data_kind == 'greater':
            assert E == 9
        elif self.data_kind == 'carla':
            assert E == 11
        other = other_sizes = None
        if len(pcl_target) > 1:
            other_time = np.random.randint(len(pcl_target) - 1)
            if other_time == time_idx:
                other_time += 1
            other, other_sizes = pcl_target[other_time], pcl_target_size[other_time]
        # The reference's pipeline moves the clouds to the GPU and leaves meta_data['pcl_target_size'] on the host
        # (pipeline.py:75-91): the sizes follow their cloud here (a few bytes, no stall), so that the one-transfer
        # reads below never mix devices.
        sizes = torch.as_tensor(sizes).to(frame.device)
        if other_sizes is not None:
            other_sizes = torch.as_tensor(other_sizes).to(other.device)
        fast = (SAMPLER_FAST and GRID_GAP_FILTER and frame.is_cuda and 'ivalo' not in self.point_sample_bias
                and self.num_solid > 0 and self.num_air > 0)
        if not fast:
            assert torch.all(sizes <= M)
        outs = [[] for _ in range(6)]
        for i in range(B):
            res = self._element_fast(frame, sizes, other, other_sizes, i, time_idx) if fast else None
            if res is None:
                res = self._element(frame, sizes, other, other_sizes, valo_ids, num_valo_ids, i, time_idx)
            for lst, v in zip(outs, res):
                lst.append(v)
        return tuple(torch.stack(v) for v in outs)

    def _element(self, frame, sizes, other, other_sizes, valo_ids, num_valo_ids, i, time_idx):
        """One batch element, step by step as the reference does it (utils/geometry.py:650-760): every selection sizes
        its result through a device->host read."""
        carla = self.data_kind == 'carla'
        tgt = frame[i, :int(sizes[i].item())]
        ids = sorted(list(valo_ids[i, :int(num_valo_ids[i].item())].detach().cpu().numpy()))
        if carla:
            tgt = filter_pcl_bounds_carla_output_torch(tgt, min_z=self.min_z, other_bounds=self.cube_bounds,
                                                       cube_mode=self.cube_mode)
        if tgt.shape[0] < 256:
            raise RuntimeError(f'Invalid due to cur_tgt_pcl_count: {tgt.shape[0]}')
        max_slice = int((2 ** 27) // self.num_air)
        used = tgt.shape[0] // int(np.ceil(tgt.shape[0] / max_slice)) + 1
        tgt_unique = other_unique = None
        if 'moving' in self.point_sample_bias:
            oth_count = int(other_sizes[i].item())
            oth = other[i, :oth_count]
            if carla:
                oth = filter_pcl_bounds_carla_output_torch(oth, min_z=self.min_z, other_bounds=self.cube_bounds,
                                                           cube_mode=self.cube_mode)
                oth_count = tgt.shape[0]            # sic (utils/geometry.py:704)
            if oth_count < 256:
                raise RuntimeError(f'Invalid due to cur_other_pcl_count: {oth_count}')
            tgt_sub, oth_sub = tgt[:used], oth[:used]
            r2 = self.point_occupancy_radius * 2.0
            tgt_unique = _gap_rows(tgt_sub, oth_sub[..., :3], r2)
            other_unique = _gap_rows(oth_sub, tgt_sub[..., :3], r2)
        (sq, st, ss) = self.construct_solid_input_target(tgt, tgt_unique, ids, time_idx)
        (aq, at, as_) = self.construct_air_input_target(tgt, other_unique, sq, ids, time_idx)
        return (sq, aq, st, at, ss, as_)

    def _bounds_key(self, rows, size):
        """1.0 for the rows of one padded cloud (M, E) that the reference keeps: index < size and, for CARLA, inside the
        output cuboid (filter_pcl_bounds_carla_output_torch's comparisons) -- as a key for the compaction kernel."""
        keep = torch.arange(rows.shape[0], device=rows.device) < size
        if self.data_kind == 'carla':
            sx, sy, sz = _CARLA_OUTPUT_SCALE[self.cube_mode]
            ob = self.cube_bounds
            for col, lo, hi in ((0, 0.0, ob * sx), (1, -ob * sy, ob * sy), (2, self.min_z, ob * sz)):
                keep = keep & (lo <= rows[:, col]) & (rows[:, col] <= hi)
        return keep.to(torch.float32)

    def _element_fast(self, frame, sizes, other, other_sizes, i, time_idx):
        """The same element with THREE device->host reads instead of ~25: every selection writes into a buffer sized for
        the worst case and leaves its count on the device (ops.compact_rows_nosync); the counts of a stage are read in
        one transfer, all of the element's host-generator draws are then made -- in the reference's call order, with the
        reference's arguments (golden G13 pins both) -- and uploaded in two copies.  Returns None (before any draw) when
        the element needs the step-by-step path (semantic ids outside 0..63 or not integral)."""
        dev = frame.device
        carla = self.data_kind == 'carla'
        inst_idx, segm_idx, view_idx = (4, 5, 6) if carla else (3, 3, 4)
        bias = self.point_sample_bias
        moving = 'moving' in bias
        r = self.point_occupancy_radius
        M = frame.shape[1]

        # ---- stage A: the two clouds of the element, compacted; their sizes
        tgt_buf, tgt_cnt = ops.compact_rows_nosync(frame[i], self._bounds_key(frame[i], sizes[i]), 0.5)
        heads = [tgt_cnt.to(torch.int64), sizes.reshape(-1).to(torch.int64)]
        if moving:
            oth_buf, oth_cnt = ops.compact_rows_nosync(other[i], self._bounds_key(other[i], other_sizes[i]), 0.5)
            heads += [oth_cnt.to(torch.int64), other_sizes[i].reshape(1).to(torch.int64)]
        host = torch.cat(heads).cpu().tolist()                                         # (sync 1)
        tgt_n = host[0]
        assert all(z <= M for z in host[1:1 + sizes.numel()])
        if tgt_n < 256:
            raise RuntimeError(f'Invalid due to cur_tgt_pcl_count: {tgt_n}')
        tgt = tgt_buf[:tgt_n]
        max_slice = int((2 ** 27) // self.num_air)
        used = tgt_n // int(np.ceil(tgt_n / max_slice)) + 1

        # ---- stage B: every selection the biases need, counts left on the device
        counts = []                                           # device tensors, read together
        if moving:
            oth_n = host[-2]
            oth_count = tgt_n if carla else host[-1]          # sic (utils/geometry.py:704)
            if oth_count < 256:
                raise RuntimeError(f'Invalid due to cur_other_pcl_count: {oth_count}')
            tgt_sub, oth_sub = tgt[:used], oth_buf[:oth_n][:used]
            r2 = r * 2.0
            uq_buf, uq_cnt = ops.compact_rows_nosync(tgt_sub, ops.RadiusGrid(oth_sub[..., :3], r2).far(tgt_sub, r2), 0.5)
            ou_buf, ou_cnt = ops.compact_rows_nosync(oth_sub, ops.RadiusGrid(tgt_sub[..., :3], r2).far(oth_sub, r2), 0.5)
            counts += [uq_cnt, ou_cnt]
        if 'low' in bias:
            z = tgt[:, 2]
            low_buf, low_cnt = ops.compact_rows_nosync(
                tgt, torch.logical_and(self.low_prefer_min_z <= z, z <= self.low_prefer_max_z).to(torch.float32), 0.5)
            counts.append(low_cnt)
        by_class = 'vehped' in bias or 'sembal' in bias
        if by_class:
            assert carla
            segm = tgt[:, segm_idx]
            segm_i = segm.to(torch.int32)
            # rows grouped by semantic id, original order inside a group (= tgt[segm == id] for every id at once)
            sorted_ids, order = torch.sort(segm_i, stable=True)
            order = order.to(torch.int32)
            # first position of every id 0 .. _SEGM_BINS in the sorted list (a histogram by atomics on 64 addresses costs
            # milliseconds): [0] = rows with negative ids, tgt_n - [-1] = rows with ids too large
            first = torch.searchsorted(sorted_ids, torch.arange(_SEGM_BINS + 1, device=dev, dtype=torch.int32))
            odd = (segm != segm_i.to(torch.float32)).any().reshape(1)
        parts = [c.to(torch.float32) for c in counts]
        if by_class:
            parts += [first.to(torch.float32), odd.to(torch.float32)]
        stage = torch.cat(parts).cpu().tolist() if parts else []                         # (sync 2)
        it = iter(stage)
        uq_n = ou_n = low_n = 0
        if moving:
            uq_n, ou_n = int(next(it)), int(next(it))
        if 'low' in bias:
            low_n = int(next(it))
        if by_class:
            seg_off = np.array([int(next(it)) for _ in range(_SEGM_BINS + 1)], dtype=np.int64)
            if seg_off[0] or seg_off[-1] != tgt_n or next(it):
                return None
            seg_count = [int(c) for c in np.diff(seg_off)]

        # ---- host: shares, counts and EVERY draw of the element, in the reference's order
        shares = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, 0.0])     # regular, low, moving, vehped, ivalo, sembal
        if 'low' in bias and low_n >= 256:
            shares[1] += 1.0
        if moving:
            if uq_n >= 256:
                shares[2] += 0.4
            elif uq_n >= 16:
                shares[2] += uq_n * 0.4 / 256.0
        if 'vehped' in bias:
            vehped_n = seg_count[4] + seg_count[10]
            if vehped_n >= 256:
                shares[3] += 0.2
            elif vehped_n >= 16:
                shares[3] += vehped_n * 0.2 / 256.0
        if 'sembal' in bias:
            shares[5] += 0.4
        shares /= shares.sum()
        n_low, n_moving, n_vehped, n_ivalo, n_sembal = [int(shares[j] * self.num_solid) for j in range(1, 6)]
        assert n_ivalo == 0
        ints = []                                             # (what the indices address, CPU LongTensor)
        if n_low > 0:
            ints.append(('low', torch.randint(0, low_n, (n_low, ))))
        if n_moving > 0:
            ints.append(('uq', torch.randint(0, uq_n, (n_moving, ))))
        if n_vehped > 0:
            pick = torch.randint(0, vehped_n, (n_vehped, ))
            c4 = seg_count[4]                                 # (get_vehped_points: the rows of class 4, then of class 10)
            ints.append(('order', torch.where(pick < c4, pick + int(seg_off[4]), pick - c4 + int(seg_off[10]))))
        if n_sembal > 0:
            seg_ids = [sid for sid in range(_SEGM_BINS) if seg_count[sid] > 0]
            taken = 0
            for sid in seg_ids:
                if seg_count[sid] >= 16:
                    per = n_sembal // len(seg_ids)
                    ints.append(('order', torch.randint(0, seg_count[sid], (per, )) + int(seg_off[sid])))
                    taken += per
            n_sembal = taken
        n_regular = self.num_solid - n_low - n_moving - n_vehped - n_ivalo - n_sembal
        if n_regular > 0:
            ints.append(('tgt', torch.randint(0, tgt_n, (n_regular, ))))
        floats = [sample_random_uniform_3ball(self.num_solid, r / 2.0)]
        air_shares = torch.tensor([0.5, 0.0, 0.3, 0.2])       # regular, moving, hard_solid_query, hard_target
        if moving:
            if ou_n >= 256:
                air_shares[1] += 0.4
            elif ou_n >= 16:
                air_shares[1] += ou_n * 0.4 / 256.0
        air_shares /= air_shares.sum()
        a_moving = int(air_shares[1] * self.num_air)
        a_hsq = int(air_shares[2] * self.num_air)
        a_ht = int(air_shares[3] * self.num_air)
        a_regular = self.num_air - a_moving - a_hsq - a_ht
        air = []                                              # (source, rows wanted, warn)
        if a_moving > 0:
            draw = int(a_moving * 1.6)
            ints.append(('ou', torch.randint(0, ou_n, (draw, ))))
            floats.append(sample_random_uniform_3ball(draw, r * 2.0))
            air.append(('ou', a_moving, False))
        if a_hsq > 0:
            draw = int(a_hsq * 2.0)
            ints.append(('solid', torch.randint(0, self.num_solid, (draw, ))))
            floats.append(sample_random_uniform_3ball(draw, max_radius=r * 3.0, min_radius=r))
            air.append(('solid', a_hsq, True))
        if a_ht > 0:
            draw = int(a_ht * 2.0)
            ints.append(('tgt', torch.randint(0, tgt_n, (draw, ))))
            floats.append(sample_random_uniform_3ball(draw, max_radius=r * 3.0, min_radius=r))
            air.append(('tgt', a_ht, True))
        if a_regular > 0:
            draw = int(a_regular * (1.3 if self.data_kind == 'greater' else 1.1))
            floats.append(sample_implicit_points_blind_torch(self.data_kind, draw, self.cube_mode, self.cube_bounds,
                                                             self.min_z, 'cpu'))
            air.append(('blind', a_regular, True))
        n_solid_ints = len(ints) - sum(1 for a in air if a[0] != 'blind')
        idx_dev = _upload(torch.cat([t for _, t in ints]).to(torch.int32), dev) if ints else None
        flt_dev = _upload(torch.cat(floats), dev)
        idx_parts, at = [], 0
        for _, t in ints:
            idx_parts.append(idx_dev[at:at + t.numel()])
            at += t.numel()
        flt_parts, at = [], 0
        for t in floats:
            flt_parts.append(flt_dev[at:at + t.shape[0]])
            at += t.shape[0]

        # ---- solid points: gathers in pool order, no host read
        sources = {'tgt': tgt}
        if moving:
            sources.update(uq=uq_buf, ou=ou_buf)
        if 'low' in bias:
            sources['low'] = low_buf

        def rows_of(kind, ix):
            if kind == 'order':
                return ops.gather_rows(tgt, order[ix.long()])
            return ops.gather_rows(sources[kind], ix)
        chosen = torch.cat([rows_of(kind, ix) for (kind, _), ix in zip(ints[:n_solid_ints], idx_parts)], dim=0)
        assert chosen.shape[0] == self.num_solid
        xyz = ops.add_rows(chosen[..., :3], flt_parts[0])
        sq = torch.cat([xyz, torch.ones_like(xyz[..., 0:1]) * time_idx], dim=-1)
        st = torch.cat([torch.ones_like(xyz[..., 0:1]), chosen[..., -4:]], dim=-1)
        if self.predict_segmentation:
            sg = chosen[..., segm_idx:segm_idx + 1].clone()
            sg[sg >= self.semantic_classes] = 3            # = Other
            st = torch.cat([st, sg], dim=-1)
        else:
            st = torch.cat([st, -torch.ones_like(st[..., 0:1])], dim=-1)

        # ---- air points: ONE radius test of all candidates against the target frame, then per-source compaction
        sources['solid'] = sq
        cands = []
        for j, (kind, _, _) in enumerate(air):
            if kind == 'blind':
                cands.append(flt_parts[1 + j])
            else:
                cands.append(ops.add_rows(ops.gather_rows(sources[kind], idx_parts[n_solid_ints + j])[..., :3],
                                          flt_parts[1 + j]))
        allc = torch.cat(cands, dim=0)
        far = ops.RadiusGrid(tgt[..., :3], r).far(allc, r)
        points, kept_counts, at = [], [], 0
        for cand, (_, want, _) in zip(cands, air):
            kept, cnt = ops.compact_rows_nosync(allc[at:at + cand.shape[0]], far[at:at + cand.shape[0]], 0.5)
            at += cand.shape[0]
            # select_safely without the read: a too-short tensor doubled until it suffices = its rows repeated cyclically
            cyc = torch.remainder(torch.arange(want, device=dev, dtype=torch.int32), cnt.clamp(min=1))
            points.append(ops.gather_rows(kept, cyc))
            kept_counts.append(cnt)
        axyz = torch.cat(points, dim=0)
        assert axyz.shape[0] == self.num_air
        aq = torch.cat([axyz, torch.ones_like(axyz[..., 0:1]) * time_idx], dim=-1)
        atg = -torch.ones((self.num_air, 6), device=dev, dtype=tgt.dtype)
        atg[..., 0] = 0.0
        for c, (_, want, warn) in zip(torch.cat(kept_counts).cpu().tolist(), air):          # (sync 3)
            if c == 0:
                raise RuntimeError('select_safely: no candidate survived the air / solid gap filter')
            while c < want:
                if warn and self.logger is not None:
                    self.logger.warning(f'Size {c} is insufficient for {want}!')
                c *= 2
        return (sq, aq, st, atg, shares, air_shares)

    def construct_solid_input_target(self, cur_tgt_pcl, cur_tgt_unique, cur_valo_ids, time_idx):
        tgt = cur_tgt_pcl
        carla = self.data_kind == 'carla'
        inst_idx, segm_idx, view_idx = (4, 5, 6) if carla else (3, 3, 4)
        bias = self.point_sample_bias
        shares = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, 0.0])     # regular, low, moving, vehped, ivalo, sembal
        if 'low' in bias:
            low = tgt[torch.logical_and(self.low_prefer_min_z <= tgt[..., 2], tgt[..., 2] <= self.low_prefer_max_z)]
            if low.shape[0] >= 256:
                shares[1] += 1.0
        if 'moving' in bias:
            if cur_tgt_unique.shape[0] >= 256:
                shares[2] += 0.4
            elif cur_tgt_unique.shape[0] >= 16:
                shares[2] += cur_tgt_unique.shape[0] * 0.4 / 256.0
        if 'vehped' in bias:
            assert carla
            vehped = get_vehped_points(tgt, segm_idx)
            if vehped.shape[0] >= 256:
                shares[3] += 0.2
            elif vehped.shape[0] >= 16:
                shares[3] += vehped.shape[0] * 0.2 / 256.0
        if 'ivalo' in bias:
            assert carla
            if len(cur_valo_ids) > 0:
                visible = get_vehped_points(tgt[tgt[..., view_idx] == 0], segm_idx)
                vis_ids = sorted(list(visible[..., inst_idx].type(torch.int32).unique().detach().cpu().numpy()))
                hidden = get_vehped_points(tgt[tgt[..., view_idx] != 0], segm_idx)
                parts = []
                for vid in cur_valo_ids:
                    rows = hidden[hidden[..., inst_idx] == vid]
                    parts.append(rows)
                    if vid not in vis_ids:                # total occlusion: oversample by adding twice
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
        n_low, n_moving, n_vehped, n_ivalo, n_sembal = [int(shares[j] * self.num_solid) for j in range(1, 6)]
        if n_low > 0:
            pool.append(_take(low, torch.randint(0, low.shape[0], (n_low, ))))
        if n_moving > 0:
            pool.append(_take(cur_tgt_unique, torch.randint(0, cur_tgt_unique.shape[0], (n_moving, ))))
        if n_vehped > 0:
            pool.append(_take(vehped, torch.randint(0, vehped.shape[0], (n_vehped, ))))
        if n_ivalo > 0:
            pool.append(_take(ivalo, torch.randint(0, ivalo.shape[0], (n_ivalo, ))))
        if n_sembal > 0:
            seg_ids = list(tgt[..., segm_idx].type(torch.int32).unique().detach().cpu().numpy())
            used = 0
            for sid in seg_ids:
                rows = tgt[tgt[..., segm_idx] == sid]
                if rows.shape[0] >= 16:
                    per = n_sembal // len(seg_ids)
                    pool.append(_take(rows, torch.randint(0, rows.shape[0], (per, ))))
                    used += per
            n_sembal = used
        n_regular = self.num_solid - n_low - n_moving - n_vehped - n_ivalo - n_sembal
        if n_regular > 0:
            pool.append(_take(tgt, torch.randint(0, tgt.shape[0], (n_regular, ))))
        chosen = torch.cat(pool, dim=0)
        assert chosen.shape[0] == self.num_solid
        offset = sample_random_uniform_3ball(self.num_solid, self.point_occupancy_radius / 2.0).to(chosen.device)
        xyz = ops.add_rows(chosen[..., :3], offset)
        query = torch.cat([xyz, torch.ones_like(xyz[..., 0:1]) * time_idx], dim=-1)
        target = torch.cat([torch.ones_like(xyz[..., 0:1]), chosen[..., -4:]], dim=-1)
        if self.predict_segmentation:
            segm = chosen[..., segm_idx:segm_idx + 1].clone()
            segm[segm >= self.semantic_classes] = 3            # = Other
            target = torch.cat([target, segm], dim=-1)
        else:
            target = torch.cat([target, -torch.ones_like(target[..., 0:1])], dim=-1)
        return (query, target, shares)

    def construct_air_input_target(self, cur_tgt_pcl, cur_other_unique, cur_solid_input, cur_valo_ids, time_idx):
        tgt = cur_tgt_pcl
        r = self.point_occupancy_radius
        tgt_xyz = tgt[..., :3]
        shares = torch.tensor([0.5, 0.0, 0.3, 0.2])           # regular, moving, hard_solid_query, hard_target
        if 'moving' in self.point_sample_bias:
            if cur_other_unique.shape[0] >= 256:
                shares[1] += 0.4
            elif cur_other_unique.shape[0] >= 16:
                shares[1] += cur_other_unique.shape[0] * 0.4 / 256.0
        shares /= shares.sum()
        points = []
        grid = ops.RadiusGrid(tgt_xyz, r) if GRID_GAP_FILTER else None      # one grid of the target frame, four filters

        def keep(cand, count, warn=True):
            # (the reference also carries the kept candidates' 1-NN distances along and never uses them: :1003-1010)
            kept = _gap_rows(cand, tgt_xyz, r, grid)
            points.append(self.select_safely(kept, count, warn_insufficient=warn))

        n_moving = int(shares[1] * self.num_air)
        if n_moving > 0:
            draw = int(n_moving * 1.6)
            cand = _take(cur_other_unique, torch.randint(0, cur_other_unique.shape[0], (draw, )))[..., :3]
            keep(ops.add_rows(cand, sample_random_uniform_3ball(draw, r * 2.0).to(tgt.device)), n_moving, warn=False)
        n_hsq = int(shares[2] * self.num_air)
        if n_hsq > 0:
            draw = int(n_hsq * 2.0)
            cand = _take(cur_solid_input, torch.randint(0, cur_solid_input.shape[0], (draw, )))[..., :3]
            keep(ops.add_rows(cand, sample_random_uniform_3ball(draw, max_radius=r * 3.0, min_radius=r).to(tgt.device)),
                 n_hsq)
        n_ht = int(shares[3] * self.num_air)
        if n_ht > 0:
            draw = int(n_ht * 2.0)
            cand = _take(tgt, torch.randint(0, tgt.shape[0], (draw, )))[..., :3]
            keep(ops.add_rows(cand, sample_random_uniform_3ball(draw, max_radius=r * 3.0, min_radius=r).to(tgt.device)),
                 n_ht)
        n_regular = self.num_air - n_moving - n_hsq - n_ht
        if n_regular > 0:
            draw = int(n_regular * (1.3 if self.data_kind == 'greater' else 1.1))
            keep(sample_implicit_points_blind_torch(self.data_kind, draw, self.cube_mode, self.cube_bounds, self.min_z,
                                                    tgt.device), n_regular)
        xyz = torch.cat(points, dim=0)
        assert xyz.shape[0] == self.num_air
        query = torch.cat([xyz, torch.ones_like(xyz[..., 0:1]) * time_idx], dim=-1)
        target = -torch.ones((self.num_air, 6), device=tgt.device, dtype=tgt.dtype)
        target[..., 0] = 0.0
        return (query, target, shares)

    def select_safely(self, pcl, num_select, warn_insufficient=True):
        """First num_select rows; a too-short tensor is doubled until it suffices."""
        if pcl.shape[0] == 0:
            raise RuntimeError('select_safely: no candidate survived the air / solid gap filter')
        while pcl.shape[0] < num_select:
            if warn_insufficient and self.logger is not None:
                self.logger.warning(f'Size {pcl.shape[0]} is insufficient for {num_select}!')
            pcl = torch.cat([pcl, pcl], dim=0)
        return pcl[:num_select].clone()
