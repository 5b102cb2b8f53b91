"""Training step on the HIP library (BASELINE config 5).

The reference's train loop (train.py:38-159) wraps a pipeline whose call arity does not match the
published models, so the step is restated against model/ signatures: encode the
point-cloud video, decode supervision query points of every target frame, sum the implicit
losses of loss.py (density BCE :50-64, colour L1 :66-154, segmentation CE :156-173, tracking
BCE :175-194), backward through occlusions4d_amd.autograd, all-reduce gradients across ranks
(one process per GPU, RCCL) instead of nn.DataParallel, clip, AdamW step.

torch supplies the tape, the loss reductions, the optimiser and torch.distributed; every network
forward and backward kernel is libocc4d.so.  The training-time point sampler
(utils/geometry.py:578-1105, rank 2 of 8(f)) is not built: query points and their targets are
inputs of the step.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops


def _masked_mean(values, mask):
    """mean of values[mask] without data-dependent shapes (capturable in a hipGraph); same value up to fp32
    summation order."""
    m = mask.to(values.dtype)
    while m.dim() < values.dim():
        m = m[..., None]
    return (values * m).sum() / (m.sum() * (values.numel() // mask.numel()))


def implicit_loss(implicit_output, implicit_target, density_lw=1.0, color_lw=0.0, segmentation_lw=0.0,
                  tracking_lw=0.0, color_mode='rgb', semantic_classes=13, static_shapes=False):
    """implicit_output (T,N,G) logits (density, R, G, B, mark_track, segm?); implicit_target (T,N,6) with
    (density, R, G, B, mark_track, segm).  Per-frame means averaged over frames, weighted sum.
    static_shapes=True computes the masked means by weighting instead of boolean indexing (no host sync, no
    data-dependent shapes): the form GraphedTrainStep captures."""
    total = implicit_output.new_zeros(())
    nf = implicit_output.shape[0]
    for t in range(nf):
        o, y = implicit_output[t], implicit_target[t]
        if static_shapes:
            if density_lw > 0.0:
                total = total + density_lw * F.binary_cross_entropy_with_logits(o[:, 0], y[:, 0]) / nf
            if color_lw > 0.0:
                pred = torch.sigmoid(o[:, 1:4]) if color_mode == 'rgb' else o[:, 1:4]
                total = total + color_lw * _masked_mean((pred - y[:, 1:4]).abs(), y[:, 0] >= 0.1) / nf
            if segmentation_lw > 0.0:
                lab = y[:, -1].to(torch.int64)
                ce = F.cross_entropy(o[:, -semantic_classes:], lab.clamp(min=0), reduction='none')
                total = total + segmentation_lw * _masked_mean(ce, lab >= 0) / nf
            if tracking_lw > 0.0:
                bce = F.binary_cross_entropy_with_logits(o[:, 4], y[:, 4].clamp(min=0.0), reduction='none')
                total = total + tracking_lw * _masked_mean(bce, (y[:, 0] >= 0.1) & (y[:, 4] >= 0.0)) / nf
            continue
        if density_lw > 0.0:
            total = total + density_lw * F.binary_cross_entropy_with_logits(o[:, 0], y[:, 0]) / nf
        if color_lw > 0.0:
            solid = y[:, 0] >= 0.1
            pred = torch.sigmoid(o[solid, 1:4]) if color_mode == 'rgb' else o[solid, 1:4]
            total = total + color_lw * F.l1_loss(pred, y[solid, 1:4]) / nf
        if segmentation_lw > 0.0:
            lab = y[:, -1].to(torch.int64)
            keep = lab >= 0
            total = total + segmentation_lw * F.cross_entropy(o[keep][:, -semantic_classes:], lab[keep]) / nf
        if tracking_lw > 0.0:
            keep = (y[:, 0] >= 0.1) & (y[:, 4] >= 0.0)
            total = total + tracking_lw * F.binary_cross_entropy_with_logits(o[keep, 4], y[keep, 4]) / nf
    return total


def allreduce_gradients(params, world=None):
    """Average gradients over ranks with ONE flat all-reduce (28.8 MB for the 7.21 M parameters:
    latency/bandwidth of a single bucket; xGMI is point to point, so few large messages)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = world or dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class TrainStep:
    """One optimisation step: forward (encoder + decoder per target frame), losses, backward,
    gradient all-reduce, clip (train.py:107-109, max norm 0.2), optimiser step."""

    def __init__(self, pcl_net, implicit_net, lr=1e-3, weight_decay=1e-2, grad_clip=0.2, loss_kwargs=None):
        self.pcl_net, self.implicit_net = pcl_net, implicit_net
        self.params = list(pcl_net.parameters()) + list(implicit_net.parameters())
        self.optimizer = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.grad_clip = grad_clip
        self.loss_kwargs = loss_kwargs or {}

    def forward_loss(self, pcl_input, points_query, implicit_target):
        """pcl_input (1,N,8); points_query (T,Nq,4); implicit_target (T,Nq,6) -> scalar loss."""
        (pcl_abstract, features_global, _) = self.pcl_net(pcl_input, False)
        outs = [self.implicit_net(points_query[t], pcl_abstract[0], features_global[0], None)[0]
                for t in range(points_query.shape[0])]
        return implicit_loss(torch.stack(outs), implicit_target, **self.loss_kwargs)

    def __call__(self, pcl_input, points_query, implicit_target):
        ops.check_pending(wait=False)          # status of earlier steps' cooperative FPS launches (no stall)
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.forward_loss(pcl_input, points_query, implicit_target)
        loss.backward()
        allreduce_gradients(self.params)
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
        self.optimizer.step()
        return loss.detach()


class GraphedTrainStep(TrainStep):
    """TrainStep replayed as ONE captured hipGraph: forward, losses, backward, gradient all-reduce, clip and the
    AdamW update of a step are ~2200 kernel launches issued from Python; captured once (static shapes, static input
    buffers, capturable optimiser, masked-mean losses), a step is a single graph launch.  Restrictions: fixed
    shapes, fps_random_start=False (the start index would be frozen into the graph), no guided sampler inside."""

    def __init__(self, pcl_net, implicit_net, lr=1e-3, weight_decay=1e-2, grad_clip=0.2, loss_kwargs=None):
        super().__init__(pcl_net, implicit_net, lr, weight_decay, grad_clip, dict(loss_kwargs or {}, static_shapes=True))
        self.optimizer = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay, capturable=True)
        self.graph = None

    def _eager(self, pcl_input, points_query, implicit_target):
        loss = self.forward_loss(pcl_input, points_query, implicit_target)
        loss.backward()
        allreduce_gradients(self.params)
        if self.grad_clip:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
        self.optimizer.step()
        return loss.detach()

    def capture(self, pcl_input, points_query, implicit_target, warmup=2):
        """Warm-up steps on a side stream (they DO update the parameters), then the capture."""
        self.static = (pcl_input.clone(), points_query.clone(), implicit_target.clone())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        losses = []
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.optimizer.zero_grad(set_to_none=True)
                losses.append(self._eager(*self.static))
        cur.wait_stream(side)
        ops.check_pending()
        self.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager(*self.static)
        return losses

    def __call__(self, pcl_input, points_query, implicit_target):
        assert self.graph is not None, 'call capture(...) first'
        for dst, src in zip(self.static, (pcl_input, points_query, implicit_target)):
            if dst is not src:
                dst.copy_(src)
        self.graph.replay()
        return self.static_loss
