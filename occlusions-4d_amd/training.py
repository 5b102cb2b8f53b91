"""Training step on the HIP library (BASELINE config 5).

The reference's train loop (train.py:38-159) wraps a pipeline whose call arity does not match the
published models, so the step is restated against model/ signatures: encode the
point-cloud video, decode supervision query points of every target frame, sum the implicit
losses of loss.py (density BCE :50-64, colour :66-154 (L1 on RGB, or the hsv / bins classification forms), segmentation CE :156-173, tracking
BCE :175-194), backward through occlusions4d_amd.autograd, all-reduce gradients across ranks
(one process per GPU, RCCL) instead of nn.DataParallel, clip, AdamW step.

torch supplies the tape, the loss reductions, the optimiser and torch.distributed; every network
forward and backward kernel is libocc4d.so.  Query points and their targets are inputs of the step;
the training-time point sampler that draws them (utils/geometry.py:578-1105, rank 2 of 8(f)) is
geometry.GuidedImplicitPointSampler (bench_train.py --sampler runs it inside the step).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import autograd, kernels, ops
from .point_transformer_layer import invalidate_weight_caches


FUSED_LOSS = os.environ.get('OCC4D_FUSED_LOSS', '1') == '1'      # density + segmentation terms as one library call (csrc/loss.hip)


def _masked_mean(values, mask):
    """mean of values[mask] without data-dependent shapes (capturable in a hipGraph); same value up to fp32
    summation order."""
    m = mask.to(values.dtype)
    while m.dim() < values.dim():
        m = m[..., None]
    return (values * m).sum() / (m.sum() * (values.numel() // mask.numel()))


def squash_for_loss(implicit_output, color_mode):
    """The pre-loss squashing of the training pipeline (pipeline.py:198-212): density stays a logit (BCE with
    logits follows); 'rgb' -> sigmoid of channels 1:4, 'rgb_nosigmoid' -> clamp to [0, 1] (so no gradient
    flows through colours outside [0, 1]), 'hsv' -> clamp of (S, V), 'bins' -> nothing.  Out of place."""
    if color_mode == 'rgb':
        mid = torch.sigmoid(implicit_output[..., 1:4])
    elif color_mode == 'rgb_nosigmoid':
        mid = torch.clamp(implicit_output[..., 1:4], min=0.0, max=1.0)
    elif color_mode == 'hsv':
        return torch.cat([implicit_output[..., :13], torch.clamp(implicit_output[..., 13:15], min=0.0, max=1.0),
                          implicit_output[..., 15:]], dim=-1)
    elif color_mode == 'bins':
        return implicit_output
    else:
        raise ValueError('Unknown color_mode: ' + str(color_mode))
    return torch.cat([implicit_output[..., :1], mid, implicit_output[..., 4:]], dim=-1)


TRACK_IDX = {'rgb': 4, 'rgb_nosigmoid': 4, 'hsv': 15, 'bins': 10}      # utils.get_track_idx (utils/utils.py:204-224)


def rgb_to_hsv(rgb, epsilon=1e-10):
    """(N, 3) -> (N, 3) hue in degrees, saturation, value: the reference's own arithmetic (utils/utils.py:169-191)."""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    max_rgb = rgb.max(1)[0]
    min_rgb, argmin = rgb.min(1)
    max_min = max_rgb - min_rgb + epsilon
    h1 = 60.0 * (g - r) / max_min + 60.0
    h2 = 60.0 * (b - g) / max_min + 180.0
    h3 = 60.0 * (r - b) / max_min + 300.0
    h = torch.stack((h2, h3, h1), dim=0).gather(0, argmin[None])[0]
    return torch.stack((h, max_min / (max_rgb + epsilon), max_rgb), dim=1)


def _color_term(o, y, keep, color_mode, static_shapes):
    """implicit_color_loss (loss.py:66-154) of one (example, frame): o (N, G) squashed outputs, y (N, 6) targets, keep
    (N) = solid AND colour available.  static_shapes: masked means instead of boolean indexing (capturable)."""
    if color_mode in ('rgb', 'rgb_nosigmoid'):                                   # :78-83: L1 on (R, G, B)
        if static_shapes:
            return _masked_mean((o[:, 1:4] - y[:, 1:4]).abs(), keep)
        return F.l1_loss(o[keep, 1:4], y[keep, 1:4])
    if not static_shapes:
        o, y = o[keep], y[keep]
        keep = torch.ones(o.shape[0], dtype=torch.bool, device=o.device)
    hsv = rgb_to_hsv(y[:, 1:4])                      # (rows outside `keep` carry -1 colours: finite junk, masked out)
    sat, val = hsv[:, 1], hsv[:, 2]

    def bins(n):                                                                 # :93-97 / :121-125
        hue = torch.round(hsv[:, 0] / 360.0 * n).to(torch.int64)
        hue = torch.where(hue == n, torch.zeros_like(hue), hue)
        return hue.clamp(0, n - 1) if static_shapes else hue

    def mean(values, mask):
        if static_shapes:
            m = mask.to(values.dtype)
            return (values * m).sum() / m.sum().clamp(min=1.0)
        return values[mask].mean()
    if color_mode == 'hsv':                                                      # :85-114
        hue = bins(12)
        hm = keep & (sat >= 0.2) & (val >= 0.2)        # hue is not supervised where it is too bland / too dark
        ce = F.cross_entropy(o[:, 1:13], hue, reduction='none')
        if static_shapes:
            loss_hue = torch.where(hm.sum() >= 16, mean(ce, hm) / 2.0, ce.new_zeros(()))
        else:
            loss_hue = ce[hm].mean() / 2.0 if int(hm.sum()) >= 16 else 0.0
        return (loss_hue + mean((o[:, 13] - sat).abs(), keep) + mean((o[:, 14] - val).abs(), keep)) / 3.0
    assert color_mode == 'bins'                                                  # :116-149
    target = bins(6)
    bland = (sat < 0.3) | (val < 0.3)
    target = torch.where((val < 0.2) & bland, torch.full_like(target, 6), target)
    target = torch.where((0.2 <= val) & (val < 0.6) & bland, torch.full_like(target, 7), target)
    target = torch.where((0.6 <= val) & bland, torch.full_like(target, 8), target)
    return mean(F.cross_entropy(o[:, 1:10], target, reduction='none'), keep) / 3.0


def implicit_loss(implicit_output, implicit_target, density_lw=1.0, color_lw=0.0, segmentation_lw=0.0,
                  tracking_lw=0.0, color_mode='rgb', semantic_classes=13, static_shapes=False, squashed=False):
    """loss.MyLosses.per_example + entire_batch (loss.py:200-294) on the RAW decoder outputs.

    implicit_output (T,N,G) or (T,B,N,G) logits (density, R, G, B, mark_track, segm?); implicit_target
    (T,[B,]N,6) with (density, R, G, B, mark_track, segm).  As the reference: the outputs are squashed first
    (squash_for_loss, pipeline.py:198-212; skipped when `squashed`), every term is the mean over its supervised
    points of ONE (example, frame) -- density BCE on all points (:50-64); colour L1 where density >= 0.1 AND
    colour available, target[..., 1] >= 0 (:72-83); segmentation CE where label >= 0 (:156-173); tracking BCE
    where density >= 0.1 AND mark_track >= 0 (:175-194) -- the per-(example, frame) values are averaged
    (:243-250) and summed with their weights (:276-277).
    static_shapes=True computes the masked means by weighting instead of boolean indexing (no host sync, no
    data-dependent shapes): the form a captured step needs, and the one the eager step uses to avoid its host reads."""
    if implicit_output.dim() == 3:
        implicit_output, implicit_target = implicit_output[:, None], implicit_target[:, None]
    if not squashed and color_lw > 0.0:
        implicit_output = squash_for_loss(implicit_output, color_mode)
    if color_mode not in TRACK_IDX:
        raise ValueError('Unknown color_mode: ' + str(color_mode))
    track_idx = TRACK_IDX[color_mode]    # the tracking logit sits behind the colour channels (utils.get_track_idx)
    if (FUSED_LOSS and implicit_output.is_cuda and color_lw == 0.0 and tracking_lw == 0.0 and not ops._lib.is_twin()
            and implicit_output.dtype == torch.float32 and implicit_target.dtype == torch.float32
            and semantic_classes < implicit_output.shape[-1]):
        # the two terms the published configurations weight, value and gradient, as one library call (csrc/loss.hip, round 6:
        # ~60 element-wise launches and 6.6 ms of host time per step otherwise); same means over the same points
        (nf, nb, n, g) = implicit_output.shape
        return autograd.ImplicitLossFn.apply(implicit_output.reshape(nf * nb, n, g).contiguous(),
                                             implicit_target.reshape(nf * nb, n, -1).contiguous(), int(semantic_classes),
                                             float(density_lw), float(segmentation_lw))
    total = implicit_output.new_zeros(())
    (nf, nb) = implicit_output.shape[:2]
    cells = nf * nb
    for t in range(nf):
        for b in range(nb):
            o, y = implicit_output[t, b], implicit_target[t, b]
            solid = y[:, 0] >= 0.1
            if density_lw > 0.0:
                total = total + density_lw * F.binary_cross_entropy_with_logits(o[:, 0], y[:, 0]) / cells
            if color_lw > 0.0:
                term = _color_term(o, y, solid & (y[:, 1] >= 0.0), color_mode, static_shapes)
                total = total + color_lw * term / cells
            if segmentation_lw > 0.0:
                lab = y[:, -1].to(torch.int64)
                keep = lab >= 0
                if static_shapes:
                    ce = F.cross_entropy(o[:, -semantic_classes:], lab.clamp(min=0), reduction='none')
                    term = _masked_mean(ce, keep)
                else:
                    term = F.cross_entropy(o[keep][:, -semantic_classes:], lab[keep])
                total = total + segmentation_lw * term / cells
            if tracking_lw > 0.0:
                keep = solid & (y[:, 4] >= 0.0)
                if static_shapes:
                    bce = F.binary_cross_entropy_with_logits(o[:, track_idx], y[:, 4].clamp(min=0.0),
                                                             reduction='none')
                    term = _masked_mean(bce, keep)
                else:
                    term = F.binary_cross_entropy_with_logits(o[keep, track_idx], y[keep, 4])
                total = total + tracking_lw * term / cells
    return total


class Participation:
    """Which parameters some rank produced a gradient for, as agreed in the last eager step of ONE training-step object
    (the mask lives on the TrainStep, not in a module-level table keyed by id(): ids are recycled after a model is
    freed).  `used` is None until an eager agreement exists."""

    def __init__(self):
        self.used = None

    def clear(self):
        """Forget the agreement: the next eager step of EVERY rank agrees again (call it on all ranks at the same point
        of the program -- an epoch boundary, after freezing / unfreezing parameters)."""
        self.used = None


def allreduce_gradients(params, world=None, participation=None):
    """Average gradients over ranks with ONE flat all-reduce (28.8 MB for the 7.21 M parameters:
    latency/bandwidth of a single bucket; xGMI is point to point, so few large messages).  `participation`: the
    caller's Participation record; a captured step REQUIRES one that an eager step has filled."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = world or dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    # Which parameters take part is agreed first (one small MAX all-reduce of a has-gradient mask): a parameter
    # without a gradient on THIS rank must still be reduced (as zeros) when another rank produced one -- every rank
    # reduces the same flat layout -- but one that NO rank used keeps grad = None, so that the optimiser skips it
    # (no weight decay, no moment update), exactly as in a single-process run and as under the reference's
    # nn.DataParallel (train.py:305).
    # The agreement needs one host read (a stall of the launching thread, and forbidden inside a stream capture), so it is
    # made ONCE per Participation record -- the first eager step -- and reused by every later step, eager or captured,
    # until the caller clears it (round 5; round 4 agreed again in every eager step).  All ranks run the same program, so
    # they agree / reuse at the same steps.  The set of parameters that receive gradients is a property of the model
    # graph (which heads and losses are switched on), not of the data; a parameter that turns up with a gradient after
    # the agreement excluded it is a program error and raises instead of silently skipping its reduction.
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    if participation is not None and participation.used is not None and len(participation.used) == len(params):
        used = participation.used
        late = [i for i, (p, u) in enumerate(zip(params, used)) if p.grad is not None and not u]
        if late:
            raise RuntimeError('allreduce_gradients: %d parameter(s) received a gradient after the ranks agreed that nobody '
                               'uses them; call Participation.clear() on every rank when the set of trained parameters '
                               'changes' % len(late))
    elif capturing:
        raise AssertionError('a captured step needs the participation mask of a preceding eager step '
                             '(run one eager step before capturing)')
    else:
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=params[0].device)
        dist.all_reduce(has, op=dist.ReduceOp.MAX)
        used = [bool(v) for v in has.tolist()]
        if participation is not None:
            participation.used = used
    grads = []
    for p, u in zip(params, used):
        if u:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
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


class SideStreamSampler:
    """Draws the supervision points of the NEXT training step on a side stream while the current step runs.

    The reference samples inside the step on the training stream (pipeline.py:118-161 calls
    GuidedImplicitPointSampler per target frame before the forward pass); the points depend on the target frames and on
    the random generators only, so they can be drawn one step ahead like any dataloader work.  The sampling stream
    never waits for the main stream (the step queued there runs beside it); `take()` makes the main stream wait for the
    draw's completion event and hands the tensors over (record_stream: the allocator must not recycle them under the
    consumer).  The random stream is the one a serial loop would consume: same points, same losses."""

    def __init__(self, sampler, n_frames):
        self.sampler, self.n_frames = sampler, n_frames
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())     # (inputs uploaded so far on the main stream)
        self._pending = None

    def draw(self, frames, sizes, valo_ids, num_valo_ids, ready=None):
        """Queues one step's draw: (queries (T, N, 4), targets (T, N, 6)) of batch element 0, as the bench consumes them.
        The inputs must be COMPLETE on the device before the side stream reads them: either they were uploaded before
        this object was built, or the caller passes `ready` -- an event recorded on the producing stream right after the
        upload (as PointCompletionNetV3.prefetch_geometry takes it): the side stream waits for that event only, not for
        the step queued behind it.  Without `ready` the side stream waits for everything queued on the current stream
        so far (safe, but it then also waits for a step already queued there)."""
        if ready is not None:
            self.stream.wait_event(ready)
        else:
            self.stream.wait_stream(torch.cuda.current_stream())
        for t_ in list(frames) + list(sizes) + [valo_ids, num_valo_ids]:
            if torch.is_tensor(t_) and t_.is_cuda:
                t_.record_stream(self.stream)      # the allocator must not recycle an input under the side stream
        with torch.cuda.stream(self.stream):
            qs, ts = [], []
            for t in range(self.n_frames):
                (si, ai, st, at, _, _) = self.sampler(frames, sizes, valo_ids, num_valo_ids, t)
                qs.append(torch.cat([si, ai], dim=1)[0])
                ts.append(torch.cat([st, at], dim=1)[0])
            q, tgt = torch.stack(qs), torch.stack(ts)
            done = torch.cuda.Event()
            done.record()
        self._pending = (q, tgt, done)

    def take(self):
        """The pending draw, ordered before everything queued on the current stream from here on."""
        assert self._pending is not None, 'draw() first'
        (q, tgt, done), self._pending = self._pending, None
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        q.record_stream(cur)
        tgt.record_stream(cur)
        return q, tgt


class FusedClipAdamW:
    """clip_grad_norm_(max_norm) + torch.optim.AdamW.step() for all parameters as ONE library call of three launches and no
    host read (occ4d_adamw_clip_f32, csrc/optim.hip; train.py:107-109, 287-293).

    The parameters become views of one flat fp32 buffer (`p.data` is re-pointed once, here; values unchanged) and the two
    moment buffers are flat beside it; gradients stay wherever the backward pass left them -- a table of their addresses
    (pinned host array -> one small asynchronous upload) is the only per-step host work: ~0.3 ms against ~15 ms for the
    ~150-tensor foreach path of torch (profiles/r05_train_phases.txt).  A parameter whose .grad is None is skipped
    entirely (no decay, no moment update), as torch does.  The gradients are NOT scaled in memory: the clip coefficient
    is applied inside the update (`last_norm` / `last_coef` stay on the device for whoever wants to log them)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params]
        assert self.params and all(p.is_cuda and p.dtype == torch.float32 for p in self.params), 'CUDA fp32 parameters'
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.steps = 0
        self.counts = [0] * len(self.params)                                # updates per parameter (torch's state['step'])
        dev = self.params[0].device
        numels = [p.numel() for p in self.params]
        offsets, total = [], 0
        for n in numels:
            offsets.append(total)
            total += (n + 3) // 4 * 4                                   # (16-byte aligned views)
        self.flat = torch.zeros((total,), dtype=torch.float32, device=dev)
        for p, off, n in zip(self.params, offsets, numels):
            view = self.flat[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        chunk = int(ops._lib.lib().occ4d_adamw_chunk())
        ct, cs = [], []
        for t, n in enumerate(numels):
            for lo in range(0, n, chunk):
                ct.append(t)
                cs.append(lo)
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)      # noqa: E731
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)      # noqa: E731
        self._offsets, self._numels, self._chunk_tensor, self._chunk_start = i64(offsets), i64(numels), i32(ct), i32(cs)
        self._n_chunks = len(ct)
        # the address table is uploaded asynchronously: a ring of pinned staging arrays, each reused only after the upload
        # that read it has run (the host may be several steps ahead of the device)
        self._staging = [[torch.zeros((2 * len(self.params),), dtype=torch.int64, pin_memory=True), None] for _ in range(4)]
        self._grad_ptrs = torch.zeros((2 * len(self.params),), dtype=torch.int64, device=dev)
        self._ws = torch.zeros((self._n_chunks + 2,), dtype=torch.float32, device=dev)

    @property
    def last_norm(self):
        return self._ws[self._n_chunks]

    @property
    def last_coef(self):
        return self._ws[self._n_chunks + 1]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self, max_norm=None):
        self.steps += 1
        slot = self._staging[self.steps % len(self._staging)]
        if slot[1] is not None:
            slot[1].synchronize()
        b1, b2 = self.betas
        T = len(self.params)
        table = slot[0].numpy()                                             # [0, T): addresses; [T, 2 T): (bias1, sqrt(bias2)) pairs
        bias = table[T:].view(np.float32).reshape(T, 2)
        keep = []                                                           # (non-contiguous gradients: packed copies, kept alive)
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                table[i] = 0
                continue
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            assert g.dtype == torch.float32 and g.is_cuda
            table[i] = g.data_ptr()
            k = self.counts[i] = self.counts[i] + 1
            bias[i, 0] = 1.0 - b1 ** k
            bias[i, 1] = math.sqrt(1.0 - b2 ** k)
        self._grad_ptrs.copy_(slot[0], non_blocking=True)
        L = ops._lib
        L.check(L.lib().occ4d_adamw_clip_f32(
            ops._ptr(self.flat), ops._ptr(self.exp_avg), ops._ptr(self.exp_avg_sq), ops._ptr(self._grad_ptrs),
            ops._ptr(self._offsets), ops._ptr(self._numels), len(self.params), ops._ptr(self._chunk_tensor),
            ops._ptr(self._chunk_start), self._n_chunks, self.lr, b1, b2, self.eps, self.weight_decay,
            float(max_norm) if max_norm else 0.0, ops._ptr(self._ws), ops._stream()))
        slot[1] = torch.cuda.Event()
        slot[1].record()
        del keep

    def state_dict(self):
        return dict(step=self.steps, counts=list(self.counts), exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)

    def load_state_dict(self, sd):
        self.steps = int(sd['step'])
        self.counts = list(sd.get('counts', [self.steps] * len(self.params)))
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])


class TrainStep:
    """One optimisation step: forward (encoder + decoder per target frame), losses, backward,
    gradient all-reduce, clip (train.py:107-109, max norm 0.2), optimiser step."""

    def __init__(self, pcl_net, implicit_net, lr=1e-3, weight_decay=1e-2, grad_clip=0.2, loss_kwargs=None,
                 kernel_selection=None, fused_optimizer=True):
        """`kernel_selection`: dict of kernels.Selection fields this step's forward runs under (e.g.
        dict(train_precision='bf16x6', logit_precision='bf16x6', checkpoint_attention=False)); the backward pass follows
        it through the autograd Functions.  None = the calling thread's scope."""
        self.kernel_selection = dict(kernel_selection or {})
        self.pcl_net, self.implicit_net = pcl_net, implicit_net
        self.params = list(pcl_net.parameters()) + list(implicit_net.parameters())
        # clip + AdamW as one library call over flat buffers (round 6); fused_optimizer=False: torch's own two calls
        self.fused = bool(fused_optimizer) and all(p.is_cuda for p in self.params)
        if self.fused:
            self.optimizer = FusedClipAdamW(self.params, lr=lr, weight_decay=weight_decay)
        else:
            self.optimizer = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        self.grad_clip = grad_clip
        self.loss_kwargs = loss_kwargs or {}
        self.participation = Participation()

    batch_frames = True

    def forward_loss(self, pcl_input, points_query, implicit_target):
        """pcl_input (1,N,8); points_query (T,Nq,4); implicit_target (T,Nq,6) -> scalar loss.
        The reference decodes the T target frames one after the other (pipeline.py:170-212); the decoder is pointwise
        in the queries (the frame is their 4th coordinate), so batch_frames runs all T * Nq of them through ONE decoder
        call: same outputs, GEMMs of 68 812 instead of 17 203 rows (269 instead of 135 row tiles for 256 CUs) and a
        quarter of the launches."""
        with kernels.use(**self.kernel_selection):
            return self._forward_loss(pcl_input, points_query, implicit_target)

    def _forward_loss(self, pcl_input, points_query, implicit_target):
        (pcl_abstract, features_global, _) = self.pcl_net(pcl_input, False)
        T_, Nq = points_query.shape[:2]
        if self.batch_frames:
            out = self.implicit_net(points_query.reshape(T_ * Nq, points_query.shape[-1]), pcl_abstract[0],
                                    features_global[0], None)[0].reshape(T_, Nq, -1)
        else:
            out = torch.stack([self.implicit_net(points_query[t], pcl_abstract[0], features_global[0], None)[0]
                               for t in range(T_)])
        return implicit_loss(out, implicit_target, **self.loss_kwargs)

    def __call__(self, pcl_input, points_query, implicit_target, next_pcl_input=None):
        """`next_pcl_input` (optional): the NEXT step's point cloud, already resident.  Its farthest-point chain and
        kNNs (coordinates only, no weights) are issued on the encoder's geometry stream as soon as this step's forward
        is launched, so they run under this step's backward (PointCompletionNetV3.prefetch_geometry)."""
        ops.check_pending(wait=False)          # status of earlier steps' cooperative FPS launches (no stall)
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.forward_loss(pcl_input, points_query, implicit_target)
        if next_pcl_input is not None:
            self.pcl_net.prefetch_geometry(next_pcl_input)
        with autograd.gradient_overlap():      # parameter gradients beside the data-gradient chain, joined on exit
            loss.backward()
        allreduce_gradients(self.params, participation=self.participation)
        if self.fused:
            self.optimizer.step(max_norm=self.grad_clip)
        else:
            if self.grad_clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip)
            self.optimizer.step()
        invalidate_weight_caches()             # merged inference matrices / per-scene tables are stale now
        return loss.detach()
