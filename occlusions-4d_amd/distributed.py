"""Query-sharded inference over the GPUs of one node (one process per GPU, RCCL).

The reference is single-GPU at inference and nn.DataParallel for training (SURVEY.md
§2.1); nothing is ported.  Given (pcl_abstract, features_global) every query point is
independent (no cross-query op in model/implicit.py:271-445), so the path shards with one
exchange step: rank 0 encodes the clip and broadcasts ONE packed buffer -- the abstract cloud
(M x 291 fp32, 0.6-2.5 MB) followed by the global embedding (512 B) -- and each rank then
decodes a contiguous slice of the query grid.  Outputs stay sharded unless `gather=True`
(all_gather of (N/R) x G fp32).  Both messages are latency-bound on xGMI; no collective sits
inside the decode loop.  For a stream of clips `ClipPipeline` issues the encode + broadcast of
clip i + 1 beside the decode of clip i (the default schedule of `bench.py --gpus N`, N > 1).
"""
import torch
import torch.distributed as dist

from . import inference
from . import ops


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n queries owned by `rank` (ceil split, last ranks may be short)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def packed_layout(abstract_shape, global_dim):
    """(offset of the global embedding, total floats) of the exchange buffer: the abstract cloud's rows first, the
    embedding behind them on a 16-byte boundary (the decoder's vector loads)."""
    n_abs = int(abstract_shape[0]) * int(abstract_shape[1])
    off = (n_abs + 3) // 4 * 4
    return off, off + int(global_dim)


def encode_and_share(pcl_input, pcl_net, abstract_shape, global_dim, device, src=0, timing=None):
    """Rank `src` runs the encoder; everyone receives (pcl_abstract (M,3+E), features_global (D)): views of ONE packed
    buffer that travels in ONE broadcast (round 6; two back-to-back collectives before -- each is latency, not bytes).
    `timing` (optional dict): filled with HIP events on the current stream -- 'encode' = (start, end) around the encoder
    on rank `src`, 'broadcast' = (start, end) around the broadcast -- for the bench's encode_ms / broadcast_ms."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0

    def mark():
        if timing is None or not torch.cuda.is_available():
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
    t0 = mark()
    if world == 1:                                   # nothing to exchange: the encoder's own tensors
        (pcl_abstract, features_global, _) = pcl_net(pcl_input, False)
        pcl_abstract = pcl_abstract.squeeze(0).contiguous()
        features_global = features_global.squeeze(0).contiguous()
        assert tuple(pcl_abstract.shape) == tuple(abstract_shape)
        t1 = t2 = mark()
    else:
        off, total = packed_layout(abstract_shape, global_dim)
        packed = torch.empty((total,), dtype=torch.float32, device=device)
        pcl_abstract = packed[:abstract_shape[0] * abstract_shape[1]].view(*abstract_shape)
        features_global = packed[off:off + global_dim]
        if rank == src:
            (enc_abstract, enc_global, _) = pcl_net(pcl_input, False)
            assert tuple(enc_abstract.shape[1:]) == tuple(abstract_shape)
            pack = (lambda dst, src_: ops.copy_rows(src_, out=dst)) if pcl_abstract.is_cuda else (lambda dst, src_: dst.copy_(src_))
            pack(pcl_abstract, enc_abstract.squeeze(0))
            pack(features_global.view(1, -1), enc_global.reshape(1, -1))
        t1 = mark()
        dist.broadcast(packed, src=src)
        t2 = mark()
    if timing is not None:
        timing['encode'], timing['broadcast'] = (t0, t1), (t1, t2)
    return pcl_abstract, features_global


def abstract_shape(pcl_net, n_points):
    """(M, 3+E) of the encoder output for an n_points input (ceil(N/factor) per level,
    model/modules.py:126; multi-level concat, model/model.py:224-228)."""
    sizes = [n_points]
    for _ in range(pcl_net.down_blocks):
        sizes.append(-(-sizes[-1] // pcl_net.transition_factor))
    m = sum(sizes[len(sizes) - pcl_net.abstract_levels:])
    return (m, 3 + pcl_net.d_feat * 2 ** pcl_net.down_blocks)


def sharded_inference(pcl_input, points_query, pcl_net, implicit_net, batch_size, color_mode,
                      predict_segmentation=False, track_mode='none', semantic_classes=13, gather=False,
                      squash=None, encoded=None, timing=None):
    """pcl_input (1,N,8) and points_query (Nq,4) are CUDA tensors present on every rank (the query
    grid is deterministic, every rank builds it).  Returns (local_output (n_local,G), (lo, hi)) or the
    gathered (Nq,G) tensor when gather=True.  `squash(out, codes)` applies the per-channel post-ops in
    place (default: the HIP kernel; the gloo/CPU test of the sharding logic injects a CPU one)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = points_query.device
    if encoded is None:
        shape = abstract_shape(pcl_net, pcl_input.shape[1])
        pcl_abstract, features_global = encode_and_share(pcl_input, pcl_net, shape, pcl_net.global_dim, device,
                                                         timing=timing)
    else:
        pcl_abstract, features_global = encoded
    lo, hi = shard_bounds(points_query.shape[0], rank, world)
    out = torch.empty((hi - lo, implicit_net.d_out), dtype=torch.float32, device=device)
    if points_query.is_cuda:
        inference.decode_batches(implicit_net, points_query, lo, hi, batch_size, pcl_abstract, features_global, out)
    else:   # CPU stand-ins (gloo test of the sharding logic): plain loop, no streams
        for b in range(lo, hi, batch_size):
            e = min(hi, b + batch_size)
            (o, _) = implicit_net(points_query[b:e], pcl_abstract, features_global, None)
            out[b - lo:e - lo] = o
    (squash or ops.squash)(out, inference.squash_codes(implicit_net.d_out, color_mode, predict_segmentation,
                                                       track_mode, semantic_classes))
    if not gather:
        return out, (lo, hi)
    per = (points_query.shape[0] + world - 1) // world
    padded = torch.zeros((per, implicit_net.d_out), dtype=torch.float32, device=device)
    padded[:hi - lo] = out
    parts = [torch.empty_like(padded) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, padded)
    else:
        parts = [padded]
    return torch.cat(parts)[:points_query.shape[0]]


class ClipPipeline:
    """Throughput mode for a stream of clips (one process per GPU): the encode (+ broadcast) of clip i + 1 is issued
    on a side stream while clip i decodes.  The encoder's critical path is the farthest-point-sampling chain -- one
    workgroup, ~11 ms of dependent steps -- which occupies ONE compute unit; next to the MFMA-bound decode of the
    previous clip it is almost free.  Every clip is still encoded and decoded in full; only the order of issue
    changes.  Usage: submit(clip0); then per clip: enc = take(); submit(next clip); out = decode(enc, queries)."""

    def __init__(self, pcl_net, implicit_net, batch_size, color_mode, predict_segmentation=False, track_mode='none',
                 semantic_classes=13, squash=None):
        self.pcl_net, self.implicit_net = pcl_net, implicit_net
        self.args = (batch_size, color_mode, predict_segmentation, track_mode, semantic_classes)
        self.stream = None                                   # created with the first CUDA clip (CPU stand-ins: in line)
        self.pending = None
        self.squash = squash

    def submit(self, pcl_input):
        shape = abstract_shape(self.pcl_net, pcl_input.shape[1])
        if not pcl_input.is_cuda:                            # (gloo / CPU test of the schedule: same order of issue, no streams)
            enc = encode_and_share(pcl_input, self.pcl_net, shape, self.pcl_net.global_dim, pcl_input.device)
            self.pending = (enc, None, pcl_input)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=pcl_input.device)
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)                       # pcl_input was produced on the caller's stream
        with torch.cuda.stream(self.stream):
            enc = encode_and_share(pcl_input, self.pcl_net, shape, self.pcl_net.global_dim, pcl_input.device)
            ev = torch.cuda.Event()
            ev.record()
        self.pending = (enc, ev, pcl_input)

    def take(self):
        (enc, ev, pcl_input) = self.pending
        self.pending = None
        if ev is not None:
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for t in enc:
                t.record_stream(main)
        return enc, pcl_input

    def decode(self, taken, points_query, gather=False):
        (enc, pcl_input) = taken
        (batch_size, color_mode, seg, track, classes) = self.args
        return sharded_inference(pcl_input, points_query, self.pcl_net, self.implicit_net, batch_size, color_mode, seg,
                                 track, classes, encoded=enc, gather=gather, squash=self.squash)

    def run(self, clips, points_query, gather=False):
        """The whole schedule over an iterable of clips: yields one decode result per clip, in order."""
        it = iter(clips)
        try:
            self.submit(next(it))
        except StopIteration:
            return
        while self.pending is not None:
            taken = self.take()
            nxt = next(it, None)
            if nxt is not None:
                self.submit(nxt)
            yield self.decode(taken, points_query, gather=gather)
