"""Vector-attention layer on the HIP library.

Interface mirror of the reference's model/point_transformer_layer.py
(``kNN_torch`` :76, ``index_points`` :102, ``PointTransformerLayer`` :116-183):
same constructor arguments, parameter names (to_q/to_k/to_v, pos_mlp.{0,2},
attn_mlp.{0,2}) and forward signature, so reference checkpoints load unchanged.

What runs instead of the reference's ATen ops
---------------------------------------------
* kNN: streaming top-k kernel (occ4d_knn_f32), no N x M matrix, bit-exact indices.
* attn_mlp[0] is linear before its ReLU, so (exact in R, DESIGN.md "refactoring (i)")
      W1 (q_i - k_j + pe_ij) + b1 = (W1 Wq) y_i - (W1 Wk) x2_j + (W1 P2) r_ij + (W1 c2 + b1)
  with r_ij = relu(P1 (p_i - p_j) + c1).  The per-pair 1st GEMM shrinks from K = D to
  K = 32; (W1 Wq) y and (W1 Wk) x2 are per-point GEMMs whose rows are gathered in the
  pair GEMM's epilogue.  Merged weights are formed in fp64 and rounded once.
* softmax over the K neighbours per channel + weighted sum: one fused kernel.
"""
import os

import torch
from torch import nn

from . import autograd
from . import ops

_PAIR_CHUNK = 32768      # query rows per pass (bounds the (rows*K, 2D) workspace)
USE_FUSED_ATTENTION = True   # tests flip this to cover the unfused kernel chain as well
USE_ATTN16 = True            # d = 416: second-generation fused attention kernel (csrc/crossattn16.hip); False = crossattn.hip
# d = 416: third generation, two 4-wave workgroups per CU out of phase (csrc/crossattn16p.hip); off = crossattn16.hip
USE_ATTN16P = os.environ.get('OCC4D_ATTN16P', '1') != '0'
USE_TRUNK_KERNELS = True     # row-resident fused trunk kernels (csrc/trunk*.hip) where the shapes allow; False = generic Linear
# Opt-in variants, both measured SLOWER end to end than the default (csrc/trunk.hip, one 8-wave workgroup per CU) and
# kept as tested alternatives (bench.py, 20 steps, 2 decode streams: default 122.5-122.8 ms / step; OCC4D_TRUNK4=1
# 123.7; OCC4D_TRUNK4=1 OCC4D_TRUNK_CHAIN=1 123.8-124.4):
# half-CU re-cut of the trunk kernels (csrc/trunk4.hip: 4-wave workgroups, two per CU)
USE_TRUNK4 = os.environ.get('OCC4D_TRUNK4', '0') != '0'
# decoder trunk between two cross-attention layers as ONE kernel with the activation resident in registers
# (occ4d_trunk_chain_f32, csrc/trunk4.hip; needs USE_TRUNK4); off = one kernel per layer
USE_TRUNK_CHAIN = os.environ.get('OCC4D_TRUNK_CHAIN', '0') != '0'
# 'f32' (default): every GEMM exact fp32 on v_mfma_f32_32x32x2_f32.  'bf16x3': the attention-logit GEMM of the
# fused kernel on split-bf16 MFMAs (occ4d_pt_cross_attn_bf16x3_f32); everything else unchanged.  Opt-in.
LOGIT_PRECISION = os.environ.get('OCC4D_LOGIT_PRECISION', 'f32')


# Derived-weight caches (merged matrices, per-scene tables, bf16 packs) are keyed on (data_ptr, _version) of the
# parameters PLUS this epoch.  A parameter's _version does not move when it is updated outside Python's view: a
# captured hipGraph replay (training.GraphedTrainStep), `p.data.add_()`, a raw-pointer write.  Whoever updates
# parameters that way calls invalidate_weight_caches(); TrainStep / GraphedTrainStep do after every step.
_WEIGHTS_EPOCH = [0]


def invalidate_weight_caches():
    """Forces every derived-weight cache of every module to be rebuilt on its next use."""
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


def trunk_pack(weight, kind='rows'):
    """Stage-packed copy of a 416-input weight for the row-resident trunk kernels (ops.pack_trunk_rows / _cols),
    cached on the tensor object while (storage, version, weights epoch) are unchanged; None when the kernels do
    not apply to this shape."""
    if weight.shape[1] != ops.TRUNK_WIDTH or weight.shape[0] % 32 != 0 or not weight.is_cuda:
        return None
    if kind == 'cols' and weight.shape[0] != ops.TRUNK_WIDTH:
        return None
    key = (weights_epoch(), weight.data_ptr(), weight._version, kind, USE_TRUNK4)
    hit = getattr(weight, '_occ4d_trunk_pack', None)
    if hit is not None and hit[0] == key:
        return hit[1]
    if USE_TRUNK4:
        packed = ops.pack_trunk4_rows(weight) if kind == 'rows' else ops.pack_trunk4_cols(weight)
    else:
        packed = ops.pack_trunk_rows(weight) if kind == 'rows' else ops.pack_trunk_cols(weight)
    weight._occ4d_trunk_pack = (key, packed)
    return packed


CHECKPOINT_ATTENTION = True   # training: cross-attention layers recompute their pair tensors in backward (below)
# training with CHECKPOINT_ATTENTION off: 'merged' stores the pair tensors of the MERGED form (forward_train_merged: the
# per-pair first GEMM has K = 32), 'as_written' those of the reference's op order (forward_train)
STORED_ATTENTION_FORM = os.environ.get('OCC4D_STORED_ATTENTION_FORM', 'merged')
# queries per recompute chunk in backward: bounds the (chunk * K, 2D) workspace.  Measured on BASELINE config 5
# (bench_train.py, ms per step eager / replayed, peak memory): 4096: 184 / 179, 4.4 GB; 8192: 173 / 170, 5.2 GB;
# 16384: - / 161, 6.9 GB; 32768: 159 / 157, 10.2 GB (stored pair tensors: 167 / 163, 24.2 GB) -- small chunks run the
# pair GEMMs at 57 K rows and repeat the accumulation of every weight gradient per chunk.
_CHECKPOINT_CHUNK = int(os.environ.get('OCC4D_CHECKPOINT_CHUNK', '32768'))


def _grad_or_none(outputs, inputs, grad_outputs):
    """torch.autograd.grad over the inputs that require grad; None for the others (frozen parameters)."""
    if torch.is_tensor(outputs):
        outputs, grad_outputs = [outputs], [grad_outputs]
    pairs = [(o, g) for o, g in zip(outputs, grad_outputs) if o.requires_grad]
    live = [t for t in inputs if t.requires_grad]
    if not pairs or not live:
        return tuple(None for _ in inputs)
    got = iter(torch.autograd.grad([o for o, _ in pairs], live, [g for _, g in pairs], allow_unused=True))
    return tuple(next(got) if t.requires_grad else None for t in inputs)


class _CheckpointedAttention(torch.autograd.Function):
    """Training-time vector attention without stored pair tensors ("recompute-in-backward to
    avoid storing (N_q, K, 832)").  forward = the fused inference kernel: nothing of size (N*K, .) is written.
    backward walks the queries in chunks and, per chunk, recomputes the layer in its MERGED form (refactoring (i) of
    DESIGN.md 4: the per-pair first GEMM has K = 32 instead of K = D) on the differentiable kernels, back-propagates
    the chunk's output gradient and adds up the gradients of the merged matrices / key-value tables; one small
    autograd pass then carries those to the original parameters (the merged matrices are differentiable fp64
    products of them).  Against the stored-activation path this removes the per-pair W1 GEMMs of forward and
    backward as well as the stored (N*K, 2D) tensors."""

    calls = 0        # (tests: how many layers went through this path)

    @staticmethod
    def forward(ctx, layer, x, pos, x2, pos2, idx, *params):
        _CheckpointedAttention.calls += 1
        with torch.no_grad():
            agg = layer._forward_one(x, pos, x2, pos2, None, None, knn_idx=idx)
        ctx.layer = layer
        ctx.save_for_backward(x, pos, x2, pos2, idx)
        return agg

    @staticmethod
    def backward(ctx, g):
        layer = ctx.layer
        (x, pos, x2, pos2, idx) = ctx.saved_tensors
        L = autograd.LinearFn.apply
        f64 = torch.float64
        g = g.contiguous()
        P1, c1 = layer.pos_mlp[0].weight, layer.pos_mlp[0].bias
        P2, c2 = layer.pos_mlp[2].weight, layer.pos_mlp[2].bias
        W1, b1 = layer.attn_mlp[0].weight, layer.attn_mlp[0].bias
        W2, b2 = layer.attn_mlp[2].weight, layer.attn_mlp[2].bias
        Wq, Wk, Wv = layer.to_q.weight, layer.to_k.weight, layer.to_v.weight
        with torch.enable_grad():
            # merged matrices as differentiable functions of the parameters (fp64 products, rounded once)
            W1d = W1.to(f64)
            mm = autograd.matmul64
            merged = [mm(W1d, Wq.to(f64)).float(), (mm(W1d, c2.to(f64)) + b1.to(f64)).float(),
                      mm(W1d, Wk.to(f64)).float(), mm(W1d, P2.to(f64)).float()]
            wq_l, bq_l, wk_l, wp_l = (m.detach().requires_grad_(True) for m in merged)
            x2d = x2.detach().requires_grad_(True)
            kt = L(x2d, wk_l, None, False, False, None)                    # (M, 2D), once per backward
            vt = L(x2d, Wv, None, False, False, None)                      # (M, D)
            kt_l, vt_l = kt.detach().requires_grad_(True), vt.detach().requires_grad_(True)
            # (detached leaf copies, so that frozen parameters do not break the per-chunk autograd.grad)
            P1, c1, P2l, c2l, W2, b2 = (t.detach().requires_grad_(True) for t in (P1, c1, P2, c2, W2, b2))
            leaves = [kt_l, vt_l, wq_l, bq_l, wp_l, P1, c1, P2l, c2l, W2, b2]
            sums = [torch.zeros_like(t) for t in leaves]
            gx = torch.empty_like(x)
            for lo in range(0, x.shape[0], _CHECKPOINT_CHUNK):
                hi = min(x.shape[0], lo + _CHECKPOINT_CHUNK)
                xc = x[lo:hi].detach().requires_grad_(True)
                ic = idx[lo:hi].contiguous()
                aq = L(xc, wq_l, bq_l, False, False, None)                                      # (c, 2D)
                r = autograd.PosHiddenFn.apply(pos[lo:hi].contiguous(), pos2, ic, P1, c1)      # (c*K, 32)
                if autograd.pair_mlp_fused_ok(aq, r, ic):
                    # a = aq_i - kt_j + Wp r, logits = W2 relu(a), pe = P2 r + c2 from one kernel
                    logits, pe = autograd.PairMlpFn.apply(aq, kt_l, r, wp_l, W2, b2, P2l, c2l, ic)
                else:
                    a = autograd.AttnInLinearFn.apply(aq, kt_l, r, wp_l, ic)                     # aq_i - kt_j + Wp r
                    logits = L(a, W2, b2, True, False, None)                                    # W2 relu(.) + b2
                    pe = L(r, P2l, c2l, False, False, None)
                out = autograd.SoftmaxAggGradOnlyFn.apply(logits, vt_l, pe, ic)      # (value unused: no launch)
                grads = torch.autograd.grad(out, [xc] + leaves, g[lo:hi])
                gx[lo:hi] = grads[0]
                for acc, gr in zip(sums, grads[1:]):
                    acc += gr
            (g_kt, g_vt, g_wq, g_bq, g_wp, g_P1, g_c1, g_P2, g_c2, g_W2, g_b2) = sums
            # key / value tables -> abstract features, Wk' and Wv
            (gx2a, g_wk) = torch.autograd.grad(kt, [x2d, wk_l], g_kt)
            (gx2b, g_Wv) = _grad_or_none(vt, [x2d, Wv], g_vt)
            # merged matrices -> original parameters (frozen ones are skipped: autograd.grad rejects them)
            (d_W1, d_b1, d_Wq, d_Wk, d_P2, d_c2) = _grad_or_none(merged, [W1, b1, Wq, Wk, P2, c2], [g_wq, g_bq, g_wk, g_wp])
        add = lambda u, v: u if v is None else (v if u is None else u + v)   # noqa: E731
        lp, la = layer.pos_mlp, layer.attn_mlp
        by_param = {id(Wq): d_Wq, id(Wk): d_Wk, id(Wv): g_Wv, id(lp[0].weight): g_P1, id(lp[0].bias): g_c1,
                    id(P2): add(g_P2, d_P2), id(c2): add(g_c2, d_c2), id(W1): d_W1, id(b1): d_b1,
                    id(la[2].weight): g_W2, id(la[2].bias): g_b2}
        return (None, gx, None, gx2a + gx2b, None, None) + tuple(by_param[id(p)] for p in layer.parameters())


def needs_grad(module, *tensors):
    """True when the call must be recorded for autograd: the training path (occlusions4d_amd.autograd,
    unfused differentiable kernels in the reference's as-written op order) is taken instead of the
    fused inference kernels."""
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and t.requires_grad for t in tensors):
        return True
    return module is not None and any(p.requires_grad for p in module.parameters())


def square_distance(src, dst):
    """(B,N,C),(B,M,C) -> (B,N,M) squared distances.  Convenience mirror of the reference
    helper (:16-30); the hot path never materialises this matrix (see kNN_torch)."""
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def kNN_torch(query, dataset, k):
    """(B,N0,3),(B,N1,3) -> (B,N0,k) int64 nearest-neighbour indices, nearest first, ties to
    the lowest index; arithmetic identical to the reference's square_distance + argsort."""
    assert query.dim() == 3 and dataset.dim() == 3, "Input tensors should be 3D."
    assert query.shape[0] == dataset.shape[0], "Input tensors should have same batch size."
    assert query.shape[2] == dataset.shape[2], "Input tensors should have same dimension."
    return ops.stack_batch([ops.knn(query[b], dataset[b], k, metric=0, int64=True)
                        for b in range(query.shape[0])])


def index_points(points, idx):
    """(B,N,C),(B,S[,K]) -> (B,S[,K],C) row gather on the device kernel."""
    B = idx.shape[0]
    out = [ops.gather_rows(points[b], idx[b].reshape(-1).to(torch.int32)) for b in range(B)]
    return ops.stack_batch(out).reshape(*idx.shape, points.shape[-1])


class PointTransformerLayer(nn.Module):
    def __init__(self, dim, pos_mlp_hidden_dim=32, attn_mlp_hidden_mult=2, num_neighbors=16, dim2=None):
        super().__init__()
        self.num_neighbors = num_neighbors
        dim2 = dim if dim2 is None else dim2
        self.dim, self.dim2 = dim, dim2
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(dim2, dim, bias=False)
        self.to_v = nn.Linear(dim2, dim, bias=False)
        self.pos_mlp = nn.Sequential(nn.Linear(3, pos_mlp_hidden_dim), nn.ReLU(),
                                     nn.Linear(pos_mlp_hidden_dim, dim))
        self.attn_mlp = nn.Sequential(nn.Linear(dim, dim * attn_mlp_hidden_mult), nn.ReLU(),
                                      nn.Linear(dim * attn_mlp_hidden_mult, dim))
        self._merged = {}
        self._scene = None

    # -- derived weights ---------------------------------------------------------------
    def _params_key(self, pre):
        ps = list(self.parameters()) + ([pre.weight, pre.bias] if pre is not None else [])
        return (weights_epoch(), LOGIT_PRECISION, USE_ATTN16P, USE_TRUNK4) + tuple((p.data_ptr(), p._version) for p in ps)

    def merged_weights(self, pre=None):
        """fp64-merged matrices of refactoring (i); `pre` is an optional nn.Linear applied to
        the query features just before this layer (PointTransformerBlock.layer1), folded in."""
        key = self._params_key(pre)
        hit = self._merged.get(pre is not None)
        if hit is not None and hit[0] == key:
            return hit[1]
        f64 = torch.float64
        W1 = self.attn_mlp[0].weight.detach().to(f64)
        b1 = self.attn_mlp[0].bias.detach().to(f64)
        P2 = self.pos_mlp[2].weight.detach().to(f64)
        c2 = self.pos_mlp[2].bias.detach().to(f64)
        mm = autograd.matmul64               # fp64 products on the library (occ4d_matmul_f64)
        wq = mm(W1, self.to_q.weight.detach().to(f64))
        bq = mm(W1, c2) + b1
        if pre is not None:
            bq = bq + mm(wq, pre.bias.detach().to(f64))
            wq = mm(wq, pre.weight.detach().to(f64))
        m = dict(
            wq=wq.float().contiguous(), bq=bq.float().contiguous(),
            wk=mm(W1, self.to_k.weight.detach().to(f64)).float().contiguous(),
            wp=mm(W1, P2).float().contiguous())
        m['wq_packed'] = trunk_pack(m['wq'])          # (2D, 416) query projection on the row-resident kernel
        if self.dim == 416 and self.pos_mlp[0].out_features == 32 and self.attn_mlp[2].weight.is_cuda:
            pack = ops.pack_attn16p_stream if USE_ATTN16P else ops.pack_attn16_stream
            m['attn16p' if USE_ATTN16P else 'attn16_stream'] = pack(
                self.attn_mlp[2].weight, self.attn_mlp[2].bias, m['wp'], self.pos_mlp[2].weight, self.pos_mlp[2].bias)
        m['w2_bf16x3'] = m['wp_bf16x3'] = None
        if LOGIT_PRECISION == 'bf16x3' and self.attn_mlp[2].weight.shape[1] % 32 == 0:     # (opt-in mode only)
            m['w2_bf16x3'] = ops.pack_w2_bf16x3(self.attn_mlp[2].weight)
            m['wp_bf16x3'] = ops.pack_w2_bf16x3(m['wp']) if m['wp'].shape[1] == 32 else None
        self._merged[pre is not None] = (key, m)
        return m

    def scene_tables(self, x2, owner=None):
        """Per-scene tables (W1 Wk) x2 and Wv x2.  Cached while `owner` (the tensor object, or tuple of
        tensor objects, the caller keeps alive for the scene), the feature tensor x2's storage / version and the
        weights are unchanged -- the reference recomputes them on every forward call."""
        owners = owner if isinstance(owner, tuple) else (owner,)
        key = (tuple((id(o), None if o is None else o._version) for o in owners),
               (x2.data_ptr(), x2._version, tuple(x2.shape)), self._params_key(None))
        if owner is not None and self._scene is not None and self._scene[0] == key \
                and all(a is b for a, b in zip(self._scene[1], owners)):
            return self._scene[2]
        m = self.merged_weights(None)
        # third entry: the value table with pos_mlp[2].bias folded in (v_j + pe_ij = (Wv f_j + c2) + P2 r_ij), which
        # is what the paired-workgroup kernel reads (its GEMM3 then starts from 0: no VALU instruction for the bias)
        tabs = (ops.linear(x2, m['wk']), ops.linear(x2, self.to_v.weight),
                ops.linear(x2, self.to_v.weight, self.pos_mlp[2].bias))
        if owner is not None:
            self._scene = (key, owners, tabs)   # holds the owners alive: their addresses cannot be recycled
        return tabs

    # -- forward -----------------------------------------------------------------------
    def forward(self, x, pos, x2=None, pos2=None):
        """x (B,N,D), pos (B,N,3) [, x2 (B,M,D2), pos2 (B,M,3)] -> agg (B,N,D)."""
        return self._forward(x, pos, x2, pos2, pre=None, scene_owner=None)

    def _forward(self, x, pos, x2, pos2, pre, scene_owner, knn_idx=None, aq_pre=None):
        if needs_grad(self, x, x2) or (pre is not None and needs_grad(pre)):
            out = []
            for b in range(x.shape[0]):
                y = x[b] if pre is None else autograd.linear(x[b], pre)
                if (CHECKPOINT_ATTENTION and x2 is not None and self.dim in ops.FUSED_ATTN_DIMS
                        and self.num_neighbors <= ops.FUSED_ATTN_MAX_K and self.pos_mlp[0].out_features == 32):
                    idx = knn_idx[b] if knn_idx is not None else ops.knn(pos[b].detach(), pos2[b].detach(),
                                                                          self.num_neighbors, metric=0)
                    out.append(_CheckpointedAttention.apply(self, y, pos[b].detach(), x2[b], pos2[b].detach(), idx,
                                                            *self.parameters()))
                    continue
                if (STORED_ATTENTION_FORM == 'merged' and x2 is not None and self.pos_mlp[0].out_features == 32
                        and self.attn_mlp[0].in_features == self.dim):
                    out.append(self.forward_train_merged(y, pos[b].detach(), x2[b], pos2[b].detach(),
                                                         idx=None if knn_idx is None else knn_idx[b]))
                    continue
                out.append(self.forward_train(y, pos[b], None if x2 is None else x2[b],
                                              None if pos2 is None else pos2[b],
                                              idx=None if knn_idx is None else knn_idx[b]))
            return ops.stack_batch(out)
        out = []
        for b in range(x.shape[0]):
            xb2 = None if x2 is None else x2[b]
            pb2 = None if pos2 is None else pos2[b]
            out.append(self._forward_one(x[b], pos[b], xb2, pb2, pre, scene_owner,
                                         None if knn_idx is None else knn_idx[b],
                                         None if aq_pre is None else aq_pre[b]))
        return ops.stack_batch(out)

    def forward_train_merged(self, x, pos, x2, pos2, idx=None):
        """Differentiable cross-attention for one cloud in the MERGED form of DESIGN.md 4 (i): the query / key halves of
        attn_mlp[0] become per-point tables (W1 Wq) x + (W1 c2 + b1) and (W1 Wk) x2, the positional half a K = 32 GEMM
        on the pair hidden units, so that the only per-pair GEMM with a wide contraction is attn_mlp[2].  The merged
        matrices are differentiable fp64 products of the parameters (rounded once); every pair tensor is stored for
        backward.  Same ops as the recompute path (_CheckpointedAttention.backward), without the second forward."""
        L = autograd.LinearFn.apply
        f64 = torch.float64
        if idx is None:
            idx = ops.knn(pos, pos2, self.num_neighbors, metric=0)
        P1, c1 = self.pos_mlp[0].weight, self.pos_mlp[0].bias
        P2, c2 = self.pos_mlp[2].weight, self.pos_mlp[2].bias
        W1, b1 = self.attn_mlp[0].weight, self.attn_mlp[0].bias
        W2, b2 = self.attn_mlp[2].weight, self.attn_mlp[2].bias
        W1d = W1.to(f64)
        mm = autograd.matmul64
        wq = mm(W1d, self.to_q.weight.to(f64)).float()
        bq = (mm(W1d, c2.to(f64)) + b1.to(f64)).float()
        wk = mm(W1d, self.to_k.weight.to(f64)).float()
        wp = mm(W1d, P2.to(f64)).float()
        kt = L(x2, wk, None, False, False, None)                       # (M, 2D)
        vt = L(x2, self.to_v.weight, None, False, False, None)         # (M, D)
        aq = L(x, wq, bq, False, False, None)                          # (N, 2D)
        r = autograd.PosHiddenFn.apply(pos, pos2, idx, P1, c1)         # (N*K, 32)
        if autograd.pair_mlp_fused_ok(aq, r, idx):
            logits, pe = autograd.PairMlpFn.apply(aq, kt, r, wp, W2, b2, P2, c2, idx)
        else:
            a = autograd.AttnInLinearFn.apply(aq, kt, r, wp, idx)                       # aq_i - kt_j + Wp r
            logits = L(a, W2, b2, True, False, None)                   # W2 relu(.) + b2
            pe = L(r, P2, c2, False, False, None)
        return autograd.SoftmaxAggFn.apply(logits, vt, pe, idx)

    def forward_train(self, x, pos, x2=None, pos2=None, idx=None):
        """Differentiable forward for one cloud, as written in the reference (:167-179): x (N,D).  `idx`: the
        (N,K) int32 neighbour lists when the caller already has them."""
        K = self.num_neighbors
        if x2 is None:
            x2, pos2 = x, pos
        if idx is None:
            idx = ops.knn(pos.detach(), pos2.detach(), K, metric=0)
        q = autograd.linear(x, self.to_q)
        kf = autograd.linear(x2, self.to_k)
        vf = autograd.linear(x2, self.to_v)
        r = autograd.PosHiddenFn.apply(pos.detach(), pos2.detach(), idx, self.pos_mlp[0].weight, self.pos_mlp[0].bias)
        pe = autograd.linear(r, self.pos_mlp[2])
        a = autograd.AttnInFn.apply(q, kf, pe, idx)
        h = autograd.linear(a, self.attn_mlp[0], relu_out=True)
        logits = autograd.linear(h, self.attn_mlp[2])
        return autograd.SoftmaxAggFn.apply(logits, vf, pe, idx)

    def _forward_one(self, x, pos, x2, pos2, pre, scene_owner, knn_idx=None, aq_pre=None):
        """`aq_pre` (n, 2D): the merged query projection (W1 Wq L1) x + bias when the caller already has it (the decoder's
        trunk chain writes it while the activation is still in registers)."""
        K = self.num_neighbors
        if x2 is None:
            # self-attention: queries, keys and values all come from the (post-`pre`) features
            y = x if pre is None else ops.linear(x, pre.weight, pre.bias)
            m = self.merged_weights(None)
            kt, vt = ops.linear(y, m['wk']), ops.linear(y, self.to_v.weight)
            vtc = ops.linear(y, self.to_v.weight, self.pos_mlp[2].bias) if m.get('attn16p') is not None else None
            aq_all = ops.linear(y, m['wq'], m['bq'])
            pos2 = pos
        else:
            m = self.merged_weights(pre)
            kt, vt, vtc = self.scene_tables(x2, owner=scene_owner)
            aq_all = None
        n = x.shape[0]
        agg = torch.empty((n, self.dim), dtype=torch.float32, device=x.device)
        P1, c1 = self.pos_mlp[0].weight, self.pos_mlp[0].bias
        P2, c2 = self.pos_mlp[2].weight, self.pos_mlp[2].bias
        W2, b2 = self.attn_mlp[2].weight, self.attn_mlp[2].bias
        for lo in range(0, n, _PAIR_CHUNK):
            hi = min(n, lo + _PAIR_CHUNK)
            if knn_idx is not None:      # the caller already has kNN_torch(pos, pos2, K) (shared by the decoder's layers)
                idx = knn_idx[lo:hi]
            else:
                idx = ops.knn(pos[lo:hi], pos2, K, metric=0)                    # (c,K) int32
            if aq_pre is not None:
                aq = aq_pre[lo:hi]
            elif aq_all is not None:
                aq = aq_all[lo:hi]
            elif m.get('wq_packed') is not None and USE_TRUNK_KERNELS:
                aq = ops.rowlin(x[lo:hi], m['wq_packed'], m['bq'], m['wq'].shape[0])
            else:
                aq = ops.linear(x[lo:hi], m['wq'], m['bq'])
            if (self.dim in ops.FUSED_ATTN_DIMS and K <= ops.FUSED_ATTN_MAX_K
                    and self.pos_mlp[0].out_features == 32 and USE_FUSED_ATTENTION):
                assert LOGIT_PRECISION in ('f32', 'bf16x3'), LOGIT_PRECISION
                if LOGIT_PRECISION == 'f32' and USE_ATTN16 and USE_ATTN16P and m.get('attn16p') is not None:
                    ops.pt_cross_attn16p(aq, pos[lo:hi], pos2, idx, kt, vtc, P1, c1, m['attn16p'], out=agg[lo:hi])
                    continue
                if LOGIT_PRECISION == 'f32' and USE_ATTN16 and m.get('attn16_stream') is not None:
                    ops.pt_cross_attn16(aq, pos[lo:hi], pos2, idx, kt, vt, P1, c1, m['attn16_stream'], out=agg[lo:hi])
                    continue
                ops.pt_cross_attn(aq, pos[lo:hi], pos2, idx, kt, vt, P1, c1, m['wp'], W2, b2, P2, c2,
                                  out=agg[lo:hi],
                                  w2_packed=m['w2_bf16x3'] if LOGIT_PRECISION == 'bf16x3' else None,
                                  wp_packed=m['wp_bf16x3'] if LOGIT_PRECISION == 'bf16x3' else None)
                continue
            r = ops.pt_pos_hidden(pos[lo:hi], pos2, idx, P1, c1)                # (c*K,32)
            h = ops.linear(r, m['wp'], relu_out=True, add_rows=aq, add_div=K,
                           sub_rows=kt, sub_idx=idx.view(-1))                   # (c*K,2D)
            logits = ops.linear(h, W2, b2)                                      # (c*K,D)
            del h
            pe = ops.linear(r, P2, c2)                                          # (c*K,D)
            ops.pt_softmax_agg(logits, vt, pe, idx, out=agg[lo:hi])
        return agg
