"""Point-transformer encoder on the HIP library.

Interface mirror of the reference's model/model.py ``PointCompletionNetV3``
(:12-233): constructor kwargs (the checkpoint's ``pcl_args``), parameter names
(pre_mlp, blocks.N, global_mlp, abstract_skip_mlps) and
``forward(pcl, return_intermediate) -> (pcl_out, x_global, layer_coords)``.
Only the published inference configuration is built: enable_decoder=False,
skip_connections=False (train.py:216-224); the UpTransition branch is out of scope.
"""
import torch

from . import modules
from . import ops
from . import autograd
from .point_transformer_layer import needs_grad


class PointCompletionNetV3(torch.nn.Module):

    def __init__(self, mixed_precision=False, n_input=4096, n_output=1024, d_in=6, d_out=6,
                 d_feat=32, down_blocks=3, up_blocks=2, transition_factor=4,
                 pt_num_neighbors=16, pt_norm_type='none', down_neighbors=8, abstract_levels=1,
                 skip_connections=False, enable_decoder=False, output_featurized=True,
                 output_global_emb=True, global_dim=512, fps_random_start=True):
        super().__init__()
        if enable_decoder or skip_connections:
            raise NotImplementedError('the UpTransition decoder branch is unused by every published '
                                      'configuration (train.py:223) and is out of scope')
        if mixed_precision:
            raise NotImplementedError('fp32 only (the reference default, args.py:55)')
        for k, v in dict(locals()).items():
            if k not in ('self', '__class__'):
                setattr(self, k, v)
        dim = d_feat
        self.pre_mlp = torch.nn.Sequential(torch.nn.Linear(d_in, dim), torch.nn.ReLU(),
                                           torch.nn.Linear(dim, dim))
        blocks = []
        for _ in range(down_blocks):
            blocks.append(modules.PointTransformerBlock(dim, dim, dim, num_neighbors=pt_num_neighbors))
            blocks.append(modules.DownTransition(dim, dim * 2, factor=transition_factor, knn_k=down_neighbors,
                                                 norm_type=pt_norm_type, fps_random_start=fps_random_start))
            dim *= 2
        blocks.append(modules.PointTransformerBlock(dim, dim, dim, num_neighbors=pt_num_neighbors))
        self.center_block_idx = len(blocks) - 1
        if output_global_emb:
            self.global_mlp = torch.nn.Sequential(torch.nn.Linear(dim, global_dim), torch.nn.ReLU(),
                                                  torch.nn.Linear(global_dim, global_dim))
        if abstract_levels > 1:
            self.abstract_skip_mlps = torch.nn.ModuleList(
                [torch.nn.Linear(dim // int(2 ** (abstract_levels - 1 - lv)), dim)
                 for lv in range(abstract_levels - 1)])
        self.blocks = torch.nn.ModuleList(blocks)

    def _geometry_chain(self, pos, full=False, ready=None):
        """FPS -> sub-cloud of every DownTransition, for all levels, enqueued on a side stream: they depend on
        coordinates only, and the FPS steps are a ~10 ms single-CU dependent chain that would otherwise serialise the
        whole encode.  Returns {block index: (per-batch geometry, clouds, event)}; the main stream waits on the event
        right before the block needs it.  full=True (prefetch_geometry) also runs every kNN of the encoder there: the
        pooling neighbours of the DownTransitions and the self-kNN of the PointTransformerBlocks.  `ready`: an event
        after which `pos` is complete; without it the side stream waits for everything queued on the current stream."""
        main = torch.cuda.current_stream()
        if self._geom_stream is None:
            self._geom_stream = torch.cuda.Stream()
        side = self._geom_stream
        if ready is not None:
            side.wait_event(ready)
        else:
            side.wait_stream(main)
        out = {}
        with torch.cuda.stream(side):
            cur = [pos[b].contiguous() for b in range(pos.shape[0])]
            nested = [modules.NestedFps() for _ in cur]      # (the levels' farthest-point subsets are prefixes of level 0's)
            for c in cur:
                # allocated on the side stream, read by the pooling kNN on the main stream: without this the block
                # could be recycled by a later side-stream allocation while that kNN is still queued
                c.record_stream(main)
            for i, block in enumerate(self.blocks):
                if isinstance(block, modules.DownTransition):
                    # only the FPS subsets chain on the side stream; the down-kNN of a level (which the next level's
                    # FPS does not need) is issued on the main stream when the level is consumed -- queued behind the
                    # FPS it used to delay the whole chain by 0.45 ms per encode
                    g = [block.sample(c, nested=nf) for c, nf in zip(cur, nested)]
                    if full:
                        g = [(inds, p_sub, block.neighbours(p_sub, c)) for (inds, p_sub), c in zip(g, cur)]
                    for tup in g:               # produced on `side`, consumed on `main`
                        for t in tup:
                            t.record_stream(main)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    out[i] = (g, cur, ev)
                    cur = [t[1] for t in g]
                elif full:
                    idx = [ops.knn(c, c, block.num_neighbors, metric=0) for c in cur]
                    for t in idx:
                        t.record_stream(main)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    out[i] = (idx, cur, ev)
        return out

    def prefetch_geometry(self, pcl, ready=None):
        """Issues the coordinate-only part of forward(pcl) -- the FPS chain and every kNN of the encoder -- NOW, on the
        geometry stream; the next forward() called with this very tensor (same storage, same version) picks the
        results up instead of computing them.  A training loop calls it for batch i + 1 once batch i's forward is
        launched (training.TrainStep does, given `next_pcl_input`): the 23 ms FPS chain of a 28672-point cloud then runs
        under batch i's backward instead of in front of batch i + 1's forward -- the reference hides the same work in
        its dataloader workers (utils/geometry.py:353-364).  With fps_random_start the start indices are drawn from
        torch's CPU generator at this call rather than inside forward().  `ready` (optional torch.cuda.Event): `pcl` is
        complete once it has fired; by default the geometry stream waits for all work queued on the current stream.
        Returns the geometry (also kept for the next forward)."""
        pos = pcl[..., :3].detach()
        geom = self._geometry_chain(pos, full=True, ready=ready)
        # the tensor OBJECT is kept (alive, so its address cannot be recycled by another cloud) next to the
        # (storage, version, shape) key; forward() requires both to match
        self._prefetched = (self.geometry_key(pcl), geom, pcl)
        return geom

    @staticmethod
    def geometry_key(pcl):
        return (pcl.data_ptr(), pcl._version, tuple(pcl.shape))

    _prefetched = None

    _geom_stream = None

    def forward(self, pcl, return_intermediate):
        """pcl (B,N,d_in) rows (x,y,z,...) -> (pcl_out (B,M,3+D) | None, x_global (B,F) | None,
        layer_coords list | None)."""
        train = needs_grad(self, pcl)
        B = pcl.shape[0]
        pos = pcl[..., :3].detach()
        layer_coords = [pos, pos] if return_intermediate else None
        l0, l2 = self.pre_mlp[0], self.pre_mlp[2]
        if train:
            x = ops.stack_batch([autograd.linear(autograd.linear(pcl[b], l0, relu_out=True), l2) for b in range(B)])
        else:
            x = ops.stack_batch([ops.linear(ops.linear(pcl[b], l0.weight, l0.bias, relu_out=True), l2.weight, l2.bias)
                             for b in range(B)])
        skips = []
        x_global = None
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[2] is pcl and pre[0] == self.geometry_key(pcl):
            geom = pre[1]
        else:
            geom = self._geometry_chain(pos)
        for i, block in enumerate(self.blocks):
            if isinstance(block, modules.DownTransition):
                g, clouds, ev = geom[i]
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                if len(g[0]) == 2:
                    g = [(inds, p_sub, block.neighbours(p_sub, c)) for (inds, p_sub), c in zip(g, clouds)]
                (x, pos) = block(x, pos, geometry=g)
            elif i in geom:
                idx, _, ev = geom[i]
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                (x, pos) = block(x, pos, knn_idx=idx)
            else:
                (x, pos) = block(x, pos)
            if self.output_global_emb and i == self.center_block_idx:
                g0, g2 = self.global_mlp[0], self.global_mlp[2]
                if train:
                    x_global = ops.stack_batch([
                        autograd.linear(autograd.linear(autograd.MeanRowsFn.apply(x[b])[None], g0, relu_out=True),
                                        g2)[0] for b in range(B)])
                else:
                    x_global = ops.stack_batch([
                        ops.linear(ops.linear(ops.mean_rows(x[b])[None], g0.weight, g0.bias, relu_out=True),
                                   g2.weight, g2.bias)[0] for b in range(B)])
            if return_intermediate:
                layer_coords.append(pos)
            if self.abstract_levels > 1 and isinstance(block, modules.DownTransition):
                for j, skip in enumerate(self.abstract_skip_mlps):
                    if skip.in_features == x.shape[-1]:
                        if train:
                            y = ops.stack_batch([autograd.linear(x[b], skip) for b in range(B)])
                            y = torch.cat([y[..., :-1], torch.full_like(y[..., :1], j + 1.0)], dim=-1)
                        else:
                            y = ops.stack_batch([ops.linear(x[b], skip.weight, skip.bias) for b in range(B)])
                            y[..., -1] = j + 1.0
                        skips.append(torch.cat([pos, y], dim=-1))
        if self.output_featurized:
            pcl_out = torch.cat([pos, x], dim=-1)
            if self.abstract_levels > 1:
                if train:    # (no in-place write on a taped tensor)
                    pcl_out = torch.cat([pcl_out[..., :-1],
                                         torch.full_like(pcl_out[..., :1], float(self.abstract_levels))], dim=-1)
                else:
                    pcl_out[..., -1] = self.abstract_levels
                assert len(skips) == self.abstract_levels - 1
                pcl_out = torch.cat([torch.cat(skips, dim=1), pcl_out], dim=1)
        else:
            pcl_out = None
        return (pcl_out, x_global, layer_coords)
