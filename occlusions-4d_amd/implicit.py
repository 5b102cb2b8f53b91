"""Cross-attention local-implicit decoder on the HIP library.

Interface mirror of the reference's model/implicit.py: ``positional_encode`` (:20-43),
``ResnetBlockFC`` (:68-101), ``ResnetFC`` (:104-208) and ``LocalPclResnetFC`` (:211-445)
with the reference's constructor kwargs (the checkpoint's ``implicit_args``), parameter
names (lin_in, lin_out, lin_z.N, blocks.N.fc_{0,1}, pt_blocks.N.*) and
``forward(points_query, points_abstract, features_global, features_abstract)
-> (output, penult)``.

Exact-in-R refactorings used: (ii) lin_z[i](features_query) splits into a
per-scene constant (global half + bias) plus the inverse-distance interpolation of the
per-abstract-point table W_z^local f_j, because the interpolation is linear; per-scene
tables are computed once per abstract cloud, not once per call.
"""
import torch

from . import geometry  # noqa: F401  (same module graph as the reference)
from . import modules
from . import ops
from . import autograd
from . import point_transformer_layer as ptl
from .point_transformer_layer import needs_grad, weights_epoch

_QUERY_CHUNK = 32768


def positional_encode(points, base_frequency, num_powers):
    """(..., C) -> (..., C*(2F+1)): [p, sin(p w_0), cos(p w_0), ..., sin(p w_{F-1}), cos(p w_{F-1})],
    w_i = 2 pi base 2^i."""
    flat = points.reshape(-1, points.shape[-1])
    enc = ops.posenc(flat, num_powers, base_frequency)
    return enc.reshape(*points.shape[:-1], enc.shape[-1])


def _check_activation(name):
    if name == 'relu':
        return
    if name == 'swish':
        raise NotImplementedError("activation 'swish' is not used by any published configuration; "
                                  "the fused kernels implement 'relu'")
    raise ValueError('Unknown activation: ' + str(name))


class ResnetBlockFC(torch.nn.Module):
    """x + fc_1(act(fc_0(act(x)))) (pre-activation residual block, DVR style)."""

    def __init__(self, d_in=64, d_hidden=256, d_out=64, activation='relu'):
        super().__init__()
        _check_activation(activation)
        self.d_in, self.d_hidden, self.d_out = d_in, d_hidden, d_out
        self.fc_0 = torch.nn.Linear(d_in, d_hidden, bias=True)
        self.fc_1 = torch.nn.Linear(d_hidden, d_out, bias=True)
        self.shortcut = None if d_in == d_out else torch.nn.Linear(d_in, d_out, bias=False)

    def forward(self, x):
        flat = x.reshape(-1, x.shape[-1])
        if needs_grad(self, x):
            return self._run_train(flat).reshape(*x.shape[:-1], self.d_out)
        return self._run(flat).reshape(*x.shape[:-1], self.d_out)

    def _run_train(self, x):
        h = autograd.linear(x, self.fc_0, relu_in=True)
        xs = x if self.shortcut is None else autograd.linear(x, self.shortcut)
        return autograd.linear(h, self.fc_1, relu_in=True, residual=xs)

    def _run(self, x, inplace=False):
        if self.shortcut is None and ptl.USE_TRUNK_KERNELS and self.d_in == self.d_hidden == self.d_out:
            # both layers in one kernel, the (n, d_hidden) intermediate never leaves the registers (csrc/trunk.hip)
            w0p, w1p = ptl.trunk_pack(self.fc_0.weight), ptl.trunk_pack(self.fc_1.weight, 'cols')
            if w0p is not None and w1p is not None:
                return ops.resblock(x, w0p, self.fc_0.bias, w1p, self.fc_1.bias, out=x if inplace else None)
        h = ops.linear(x, self.fc_0.weight, self.fc_0.bias, relu_in=True)
        if self.shortcut is None:
            return ops.linear(h, self.fc_1.weight, self.fc_1.bias, relu_in=True, residual=x,
                              out=x if inplace else None)
        xs = ops.linear(x, self.shortcut.weight)
        return ops.linear(h, self.fc_1.weight, self.fc_1.bias, relu_in=True, residual=xs, out=xs)


class ResnetFC(torch.nn.Module):
    """pixelNeRF-style conditioned residual MLP (global or per-point latent)."""

    def __init__(self, mixed_precision=False, d_in=4, d_hidden=256, d_out=64, d_latent=256,
                 n_blocks=5, pos_encoding_freqs=0, activation='relu'):
        super().__init__()
        _check_activation(activation)
        if mixed_precision:
            raise NotImplementedError('fp32 only (the reference default, args.py:55)')
        self.mixed_precision = mixed_precision
        self.d_in, self.d_hidden, self.d_out, self.d_latent = d_in, d_hidden, d_out, d_latent
        self.n_blocks = n_blocks
        self.pos_encoding_freqs = pos_encoding_freqs
        self.actual_d_in = d_in * (pos_encoding_freqs * 2 + 1) if pos_encoding_freqs > 0 else d_in
        if self.actual_d_in > 0:
            self.lin_in = torch.nn.Linear(self.actual_d_in, d_hidden, bias=True)
        self.lin_out = torch.nn.Linear(d_hidden, d_out, bias=True)
        self.blocks = torch.nn.ModuleList(
            [ResnetBlockFC(d_hidden, d_hidden, d_hidden, activation=activation) for _ in range(n_blocks)])
        if d_latent > 0:
            self.lin_z = torch.nn.ModuleList(
                [torch.nn.Linear(d_latent, d_hidden, bias=True) for _ in range(n_blocks)])

    def forward(self, points, features):
        return self.do_forward(points, features)

    def _embed(self, pts):
        """(n, d_in) -> (n, H): Fourier features + lin_in."""
        if self.pos_encoding_freqs > 0:
            pts = ops.posenc(pts, self.pos_encoding_freqs, 0.1)
        return ops.linear(pts, self.lin_in.weight, self.lin_in.bias)

    def do_forward(self, points, features):
        """points (B,N,d_in) or (N,d_in); features (B,D) or (B,N,D) -> (output (B,N,G), penult (B,N,H))."""
        if needs_grad(self, points, features):
            raise NotImplementedError('training is implemented for the published configuration '
                                      "(local_mode='attention'); ResnetFC.do_forward is inference-only")
        no_batch = points.dim() == 2
        if no_batch:
            points, features = points[None], features[None]
        assert points.shape[0] == features.shape[0]
        assert points.shape[-1] == self.d_in and features.shape[-1] == self.d_latent
        assert self.d_in > 0
        outs, pens = [], []
        for b in range(points.shape[0]):
            x = self._embed(points[b])
            f = features[b]
            for i in range(self.n_blocks):
                if self.d_latent > 0:
                    lz = self.lin_z[i]
                    if f.dim() == 1:
                        # one global embedding steers every point: x += (W f + b), a row constant
                        z = ops.linear(f[None].contiguous(), lz.weight, lz.bias)[0]
                        n = x.shape[0]
                        ops.interp_add(x, z, x.new_zeros((1, self.d_hidden)),
                                       torch.zeros((n, 1), dtype=torch.int32, device=x.device),
                                       x.new_zeros((n, 1)))
                    else:
                        assert f.shape[0] == x.shape[0]
                        x = ops.linear(f, lz.weight, lz.bias, residual=x, out=x)
                x = self.blocks[i]._run(x, inplace=True)
            pens.append(x)
            outs.append(ops.linear(x, self.lin_out.weight, self.lin_out.bias, relu_in=True))
        output, penult = ops.stack_batch(outs), ops.stack_batch(pens)
        if no_batch:
            output, penult = output[0], penult[0]
        return (output, penult)


class LocalPclResnetFC(ResnetFC):
    """ResnetFC + local feature interpolation + query-to-abstract vector cross-attention."""

    def __init__(self, num_local_features=0, local_mode='attention', d_latent_local=64,
                 cross_attn_neighbors=12, cross_attn_layers=1, cr_attn_type='cccccccccc', **kwargs):
        self.num_local_features = num_local_features
        self.local_mode = local_mode
        self.d_latent_local = d_latent_local
        self.cross_attn_neighbors = cross_attn_neighbors
        self.cross_attn_layers = cross_attn_layers
        self.cr_attn_type = cr_attn_type
        super().__init__(**kwargs)
        if local_mode == 'attention':
            blocks, use_at = [], []
            for i in range(cross_attn_layers):
                if cr_attn_type[i] == 'c':
                    blocks.append(modules.PointTransformerBlock(
                        self.d_latent, self.d_latent, self.d_latent, num_neighbors=cross_attn_neighbors,
                        d_hidden_abstract=d_latent_local))
                elif cr_attn_type[i] == 's':
                    raise NotImplementedError()
                else:
                    raise ValueError()
                use_at.append(int((i + 1) * self.n_blocks / (cross_attn_layers + 1)))
            self.pt_blocks = torch.nn.ModuleList(blocks)
            self.use_pt_inds = {j: i for i, j in enumerate(use_at)}
        self._scene = None

    # -- per-scene precompute ---------------------------------------------------------
    def _weights_key(self):
        return (weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in self.lin_z.parameters())

    def prepare_scene(self, points_abstract, features_global, features_abstract=None):
        """Per-scene tables: abstract xyz / features made contiguous, Z = F @ [Wz_0^loc; ..]^T (M, n_blocks*H),
        c = [Wz_i^glob g + b_i] (n_blocks*H).  Cached on the identity (+version) of the tensors the
        caller passes, which are kept alive here so their storage cannot be recycled."""
        key = (id(points_abstract), points_abstract._version, id(features_global), features_global._version,
               id(features_abstract), None if features_abstract is None else features_abstract._version,
               self._weights_key())
        sc = self._scene
        if sc is not None and sc['key'] == key and sc['owners'][0] is points_abstract \
                and sc['owners'][1] is features_global and sc['owners'][2] is features_abstract:
            return sc
        pa, fa, fg = points_abstract, features_abstract, features_global
        if fa is None:
            fa = pa[..., 3:]
            pa = pa[..., :3]
        if pa.dim() == 3:
            assert pa.shape[0] == 1, 'LocalPclResnetFC supports B == 1 only (model/implicit.py:317)'
            pa, fa = pa[0], fa[0]
        if fg.dim() == 2:
            assert fg.shape[0] == 1
            fg = fg[0]
        assert pa.shape[0] == fa.shape[0]
        assert fa.shape[-1] == self.d_latent_local
        dg = self.d_latent - self.d_latent_local
        assert fg.shape[-1] == dg
        pa = pa.contiguous()
        fa = fa.contiguous()
        wl = torch.cat([lz.weight[:, dg:] for lz in self.lin_z], dim=0).contiguous()
        wg = torch.cat([lz.weight[:, :dg] for lz in self.lin_z], dim=0).contiguous()
        bz = torch.cat([lz.bias for lz in self.lin_z], dim=0).contiguous()
        sc = dict(key=key, owners=(points_abstract, features_global, features_abstract), xyz=pa, feats=fa,
                  ztab=ops.linear(fa, wl), zconst=ops.linear(fg[None].contiguous(), wg, bz)[0])
        self._scene = sc
        return sc

    # -- forward ----------------------------------------------------------------------
    def forward(self, points_query, points_abstract, features_global, features_abstract):
        """points_query (B,N,4) or (N,4); points_abstract (B,M,3) (or (B,M,3+E) with
        features_abstract None); features_global (B,D); features_abstract (B,M,E).
        B must be 1.  Returns (output (B,N,G), penult (B,N,H)) (no batch dim if none came in)."""
        if needs_grad(self, points_abstract, features_global, features_abstract):
            return self._forward_train(points_query, points_abstract, features_global, features_abstract)
        if self.num_local_features <= 0:
            return super().do_forward(points_query, features_global)
        if self.local_mode == 'function':
            raise NotImplementedError()
        if self.local_mode not in ('feature', 'attention'):
            raise ValueError()
        no_batch = points_query.dim() == 2
        if not no_batch:
            assert points_query.shape[0] == 1, 'LocalPclResnetFC supports B == 1 only'
            assert points_abstract.shape[0] == 1 and features_global.shape[0] == 1
        q = points_query if no_batch else points_query[0]
        assert q.shape[-1] == self.d_in
        sc = self.prepare_scene(points_abstract, features_global, features_abstract)
        if self.local_mode == 'feature':
            output, penult = self._forward_feature(q, sc)
        else:
            output, penult = self._forward_attention(q, sc, sc['owners'])
        if not no_batch:
            output, penult = output[None], penult[None]
        return (output, penult)

    # -- trunk chains (occ4d_trunk_chain_f32): the blocks between two cross-attention layers as one kernel -----------
    def _chain_plan(self):
        """Segments of the trunk for local_mode 'attention': consecutive residual blocks up to (and including the query
        projection of) the next PointTransformerBlock, or up to lin_out; per segment the flat weight stream and the
        stage counts.  Cached while the parameters (storage, version, weights epoch) are unchanged; None when the
        row-resident kernels do not apply to this configuration."""
        if not (ptl.USE_TRUNK_KERNELS and ptl.USE_TRUNK4 and ptl.USE_TRUNK_CHAIN and self.d_hidden == ops.TRUNK_WIDTH
                and self.d_latent > 0 and all(b.shortcut is None and b.d_hidden == self.d_hidden for b in self.blocks)
                and self.lin_out.weight.is_cuda):
            return None
        key = (weights_epoch(), ptl.USE_ATTN16P) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        hit = getattr(self, '_chain_cache', None)
        if hit is not None and hit[0] == key:
            return hit[1]
        plan, first = [], 0
        for i in range(self.n_blocks):
            last = i == self.n_blocks - 1
            if i not in self.use_pt_inds and not last:
                continue
            blocks = list(range(first, i + 1))
            first = i + 1
            if 2 * len(blocks) + 2 > 12:          # (OCC4D_CHAIN_MAX_OPS)
                return None
            parts = [('resblock', self.blocks[j].fc_0.weight, self.blocks[j].fc_1.weight) for j in blocks]
            seg = dict(blocks=blocks, pt=self.use_pt_inds.get(i), tail=False)
            if seg['pt'] is not None:
                blk = self.pt_blocks[seg['pt']]
                m = blk.layer2.merged_weights(blk.layer1)
                if m['wq'].shape[1] != ops.TRUNK_WIDTH:
                    return None
                parts.append(('linear', m['wq']))
                seg['n_aq'], seg['bias'] = m['wq'].shape[0], m['bq']
                plan.append(seg)
                if last:                            # a cross-attention layer after the last block: lin_out on its own
                    plan.append(dict(blocks=[], pt=None, tail=True))
                    parts_tail = [('linear', self.lin_out.weight)]
                    plan[-1]['stream'], plan[-1]['counts'] = ops.pack_chain_stream(parts_tail)
                    plan[-1]['bias'] = ops.pad_bias(self.lin_out.bias, plan[-1]['counts'][-1])
            else:
                parts.append(('linear', self.lin_out.weight))
                seg['tail'], seg['bias'] = True, self.lin_out.bias
                plan.append(seg)
            seg['stream'], seg['counts'] = ops.pack_chain_stream(parts)
            seg['bias'] = ops.pad_bias(seg['bias'], seg['counts'][-1])
        self._chain_cache = (key, plan)
        return plan

    def _run_chain(self, seg, x, interp, out_rows, aq=None):
        """One segment on the rows of x (in place): returns x (also the penultimate activation after the tail)."""
        H = self.d_hidden
        prog = []
        for j in seg['blocks']:
            prog += [('interp', j * H), ('resblock', self.blocks[j].fc_0.bias, self.blocks[j].fc_1.bias)]
        ns = seg['counts'][-1]
        if seg['tail']:
            prog += [('store', x)] if seg['blocks'] else []
            prog.append(('linear', seg['bias'], ns, self.d_out, True, out_rows))
        else:
            prog += [('linear', seg['bias'], ns, seg['n_aq'], False, aq), ('store', x)]
        ops.trunk_chain(x, seg['stream'], prog, interp=interp if seg['blocks'] else None)
        return x

    def _interp(self, q, sc):
        idx, dist = ops.knn(q, sc['xyz'], self.num_local_features, metric=1, return_dist=True)
        return idx, ops.interp_weights(dist)

    def _forward_attention(self, q_all, sc, owner):
        n = q_all.shape[0]
        H = self.d_hidden
        out = torch.empty((n, self.d_out), dtype=torch.float32, device=q_all.device)
        single = 0 < n <= _QUERY_CHUNK        # one chunk: the trunk activation IS penult (no 54 MB copy)
        pen = None if single else torch.empty((n, H), dtype=torch.float32, device=q_all.device)
        xyz, feats = sc['xyz'], sc['feats']
        for lo in range(0, n, _QUERY_CHUNK):
            q = q_all[lo:lo + _QUERY_CHUNK]
            idx8, w8 = self._interp(q, sc)
            # every cross-attention layer attends from the same query xyz to the same abstract xyz with the same K:
            # one kNN_torch (model/point_transformer_layer.py:167) serves them all
            idx_att = ops.knn(q[:, :3], xyz, self.cross_attn_neighbors, metric=0)[None] if self.use_pt_inds else None
            x = self._embed(q)
            plan = self._chain_plan()
            if plan is not None:
                # the trunk between two cross-attention layers is one kernel: [x += lin_z term; block] ..., then the merged
                # query projection of the next PointTransformerBlock (or lin_out) while the rows are still in registers
                interp = (sc['zconst'], sc['ztab'], idx8, w8)
                for seg in plan:
                    aq = None
                    if seg['pt'] is not None:
                        aq = torch.empty((x.shape[0], seg['n_aq']), dtype=torch.float32, device=x.device)
                    self._run_chain(seg, x, interp, out[lo:lo + _QUERY_CHUNK], aq)
                    if seg['pt'] is not None:
                        blk = self.pt_blocks[seg['pt']]
                        x = blk(x[None], q[None, :, :3], x2=feats[None], p2=xyz[None], scene_owner=owner,
                                knn_idx=idx_att, aq_pre=aq[None])[0][0]
                if single:
                    pen = x
                else:
                    pen[lo:lo + _QUERY_CHUNK] = x
                continue
            for i in range(self.n_blocks):
                ops.interp_add(x, sc['zconst'][i * H:(i + 1) * H], sc['ztab'][:, i * H:(i + 1) * H], idx8, w8)
                x = self.blocks[i]._run(x, inplace=True)
                if i in self.use_pt_inds:
                    blk = self.pt_blocks[self.use_pt_inds[i]]
                    x = blk(x[None], q[None, :, :3], x2=feats[None], p2=xyz[None], scene_owner=owner,
                            knn_idx=idx_att)[0][0]
            if single:
                pen = x
            else:
                pen[lo:lo + _QUERY_CHUNK] = x
            ops.linear(x, self.lin_out.weight, self.lin_out.bias, relu_in=True, out=out[lo:lo + _QUERY_CHUNK])
        return out, pen

    def _forward_feature(self, q_all, sc):
        n = q_all.shape[0]
        H = self.d_hidden
        out = torch.empty((n, self.d_out), dtype=torch.float32, device=q_all.device)
        pen = torch.empty((n, H), dtype=torch.float32, device=q_all.device)
        for lo in range(0, n, _QUERY_CHUNK):
            q = q_all[lo:lo + _QUERY_CHUNK]
            idx8, w8 = self._interp(q, sc)
            x = self._embed(q)
            for i in range(self.n_blocks):
                ops.interp_add(x, sc['zconst'][i * H:(i + 1) * H], sc['ztab'][:, i * H:(i + 1) * H], idx8, w8)
                x = self.blocks[i]._run(x, inplace=True)
            pen[lo:lo + _QUERY_CHUNK] = x
            ops.linear(x, self.lin_out.weight, self.lin_out.bias, relu_in=True, out=out[lo:lo + _QUERY_CHUNK])
        return out, pen

    # -- training path (as-written op order, differentiable kernels) ------------------------
    def _forward_train(self, points_query, points_abstract, features_global, features_abstract):
        """Same contract as forward(); every op is a occlusions4d_amd.autograd Function, so gradients
        reach this module's parameters and, through features_abstract / features_global, the encoder."""
        assert self.local_mode == 'attention' and self.num_local_features > 0, \
            "training is implemented for the published configuration (local_mode='attention')"
        no_batch = points_query.dim() == 2
        q = points_query if no_batch else points_query[0]
        pa, fa, fg = points_abstract, features_abstract, features_global
        if fa is None:
            fa, pa = pa[..., 3:], pa[..., :3]
        if pa.dim() == 3:
            assert pa.shape[0] == 1, 'LocalPclResnetFC supports B == 1 only (model/implicit.py:317)'
            pa, fa = pa[0], fa[0]
        if fg.dim() == 2:
            fg = fg[0]
        pa = pa.detach().contiguous()
        fa = fa.contiguous()
        n = q.shape[0]
        dg = self.d_latent - self.d_latent_local
        idx8, dist = ops.knn(q, pa, self.num_local_features, metric=1, return_dist=True)
        w8 = ops.interp_weights(dist)
        f_local = autograd.InterpFn.apply(fa, idx8, w8)                                   # (n, E)
        f_query = torch.cat([fg[None, :].expand(n, dg), f_local], dim=-1)                 # (n, D)
        x = autograd.linear(ops.posenc(q, self.pos_encoding_freqs, 0.1), self.lin_in)
        qxyz = q[:, :3].detach()
        idx_att = None
        for i in range(self.n_blocks):
            x = autograd.linear(f_query, self.lin_z[i], residual=x)
            x = self.blocks[i]._run_train(x)
            if i in self.use_pt_inds:
                blk = self.pt_blocks[self.use_pt_inds[i]]
                y = autograd.linear(x, blk.layer1)
                if idx_att is None:       # one kNN_torch for all cross-attention layers (same xyz on both sides, same K)
                    idx_att = ops.knn(qxyz, pa, self.cross_attn_neighbors, metric=0)
                agg = blk.layer2._forward(y[None], qxyz[None], fa[None], pa[None], pre=None, scene_owner=None,
                                          knn_idx=idx_att[None])[0]
                x = autograd.linear(agg, blk.layer3, residual=x)
        output = autograd.linear(x, self.lin_out, relu_in=True)
        penult = x
        if not no_batch:
            output, penult = output[None], penult[None]
        return (output, penult)
