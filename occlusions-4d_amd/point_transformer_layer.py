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
from . import kernels
from . import ops

# Which kernel variants / which arithmetic a call runs on is a per-module, per-call choice (kernels.py): process
# defaults seeded from the OCC4D_* environment variables, a thread-local `with kernels.use(...)` scope, a module's own
# `kernel_selection`.  There are no module-level switches to assign (round 6; rounds 1-5 had USE_*, LOGIT_PRECISION,
# TRUNK_PRECISION, ... globals here, which `nn.DataParallel`'s one-thread-per-GPU forwards, train.py:305, would share).


def path_flags(module=None):
    """The OCC4D_PATH_* flags (include/occ4d.h) of a call into `module` under the current thread's selection."""
    return kernels.current(module).flags()


# Derived-weight caches (merged matrices, per-scene tables, bf16 packs) are keyed on (data_ptr, _version) of the
# parameters PLUS this epoch.  A parameter's _version does not move when it is updated outside Python's view: a
# captured hipGraph replay, `p.data.add_()`, a raw-pointer write.  Whoever updates parameters that way calls
# invalidate_weight_caches(); TrainStep does after every step.
_WEIGHTS_EPOCH = [0]


def invalidate_weight_caches():
    """Forces every derived-weight cache of every module to be rebuilt on its next use."""
    _WEIGHTS_EPOCH[0] += 1


def weights_epoch():
    return _WEIGHTS_EPOCH[0]


# Training switches (kernels.Selection): checkpoint_attention -- cross-attention layers recompute their pair tensors in
# backward (below); stored_attention_form -- with the recompute off, 'merged' stores the pair tensors of the MERGED form
# (forward_train_merged: the per-pair first GEMM has K = 32), 'as_written' those of the reference's op order
# (forward_train); checkpoint_chunk -- queries per recompute chunk in backward: bounds the (chunk * K, 2D) workspace.
# Measured on BASELINE config 5 (bench_train.py, ms per step eager / replayed, peak memory): 4096: 184 / 179, 4.4 GB;
# 8192: 173 / 170, 5.2 GB; 16384: - / 161, 6.9 GB; 32768: 159 / 157, 10.2 GB (stored pair tensors: 167 / 163, 24.2 GB) --
# small chunks run the pair GEMMs at 57 K rows and repeat the accumulation of every weight gradient per chunk.


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


def _merged_to_parameters(W1, b1, Wq, Wk, P2, c2, g_wq, g_bq, g_wk, g_wp):
    """Gradients of the parameters behind the merged matrices of DESIGN.md 4 (i)
        wq = W1 Wq,  bq = W1 c2 + b1,  wk = W1 Wk,  wp = W1 P2        (fp64 products, rounded once)
    from the gradients of the merged matrices: what autograd computes for those products (matmul64 / the casts), written
    out -- plain launches on the CURRENT stream (an autograd pass would run each node on the stream of its forward), fp64
    sums rounded once.  None for a frozen parameter.  Order: (W1, b1, Wq, Wk, P2, c2)."""
    f64 = torch.float64
    mm = ops.matmul_f64
    gq, gb, gk, gp = (t.to(f64) for t in (g_wq, g_bq, g_wk, g_wp))
    W1d = W1.detach().to(f64)
    d_W1 = None
    if W1.requires_grad:
        d_W1 = (mm(gq, Wq.detach().to(f64).t()) + gb[:, None] * c2.detach().to(f64)[None, :]
                + mm(gk, Wk.detach().to(f64).t()) + mm(gp, P2.detach().to(f64).t())).float()
    d_b1 = g_bq if b1.requires_grad else None
    W1t = W1d.t()
    d_Wq = mm(W1t, gq).float() if Wq.requires_grad else None
    d_Wk = mm(W1t, gk).float() if Wk.requires_grad else None
    d_P2 = mm(W1t, gp).float() if P2.requires_grad else None
    d_c2 = mm(W1t, gb[:, None])[:, 0].float() if c2.requires_grad else None
    return (d_W1, d_b1, d_Wq, d_Wk, d_P2, d_c2)


@kernels.carries_selection
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
        ctx.kept = None
        with torch.no_grad():
            mode = kernels.scope().store_pairs
            logits = pair = None
            if mode != 'none' and layer.logits_storable():
                # the forward kernel leaves its logits in HBM (1664 B per pair): backward reads them instead of running
                # the 832 -> 416 pair GEMM a second time; 'all': a and pe as well, nothing is recomputed
                pairs = x.shape[0] * layer.num_neighbors
                new = lambda width: torch.empty((pairs, width), dtype=torch.float32, device=x.device)   # noqa: E731
                logits = new(layer.dim)
                if mode == 'all':
                    pair = (new(2 * layer.dim), new(layer.dim))
                ctx.kept = (logits, pair)
            agg = layer._forward_one(x, pos, x2, pos2, None, None, knn_idx=idx, logits_out=logits, pair_out=pair)
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
        lp0, la2 = layer.pos_mlp[0], layer.attn_mlp[2]
        with torch.enable_grad():
            # merged matrices as differentiable functions of the parameters (fp64 products, rounded once)
            W1d = W1.to(f64)
            mm = autograd.matmul64
            merged = [mm(W1d, Wq.to(f64)).float(), (mm(W1d, c2.to(f64)) + b1.to(f64)).float(),
                      mm(W1d, Wk.to(f64)).float(), mm(W1d, P2.to(f64)).float()]
            wq_l, bq_l, wk_l, wp_l = (m.detach().requires_grad_(True) for m in merged)
            x2d = x2.detach().requires_grad_(True)
            kt = L(x2d, wk_l, None, False, False, None, None)                    # (M, 2D), once per backward
            vt = L(x2d, Wv, None, False, False, None, None)                      # (M, D)
            kt_l, vt_l = kt.detach().requires_grad_(True), vt.detach().requires_grad_(True)
            # (detached leaf copies, so that frozen parameters do not break the per-chunk autograd.grad)
            P1, c1, P2l, c2l, W2, b2 = (t.detach().requires_grad_(True) for t in (P1, c1, P2, c2, W2, b2))
            leaves = [kt_l, vt_l, wp_l, P1, c1, P2l, c2l, W2, b2]
            # the query projection for ALL queries at once, outside the chunk loop (it holds no pair tensor: (n, 2D)): one
            # Linear, one data gradient and one weight gradient over 68812 rows instead of three of each over 22976 rows,
            # which fill two thirds of a dispatch round (83 -> 105 TFLOP/s on these launches)
            x_all = x.detach().requires_grad_(True)
            aq_all = L(x_all, wq_l, bq_l, False, False, None, None)                                   # (n, 2D)
            g_aq = torch.empty_like(aq_all)
            # chunks of EQUAL size, at most `checkpoint_chunk` queries each (a multiple of 64: whole 128-pair-row workgroups
            # of the fused pair kernel): 68812 queries = 3 x 22976 instead of 2 x 32768 + 3276 -- the short chunk ran every
            # kernel of the backward at a fraction of its rate (94.8 -> 94.0 ms per step, 9.5 -> 7.7 GB peak)
            n_chunks = max(1, -(-x.shape[0] // kernels.scope().checkpoint_chunk))
            step = -(-x.shape[0] // (64 * n_chunks)) * 64
            # The gradients of the leaves (merged matrices, tables, pos-MLP) are sums over the chunks that nothing reads
            # before the loop ends: they collect in sinks (autograd.gradient_sinks: added up on the parameter-gradient
            # stream beside the chunks' data-gradient chain; autograd.grad hands back None for them)
            with autograd.gradient_sinks(leaves + [wq_l, bq_l, wk_l]) as sink:
                for lo in range(0, x.shape[0], step):
                    hi = min(x.shape[0], lo + step)
                    ic = idx[lo:hi].contiguous()
                    aq = aq_all[lo:hi].detach().requires_grad_(True)                                # (c, 2D)
                    r = autograd.PosHiddenFn.apply(pos[lo:hi].contiguous(), pos2, ic, P1, c1)      # (c*K, 32)
                    if autograd.pair_mlp_fused_ok(aq, r, ic):
                        # a = aq_i - kt_j + Wp r, logits = W2 relu(a), pe = P2 r + c2 from one kernel (the logits: the
                        # forward's own, when it kept them)
                        K = ic.shape[1]
                        kept = None
                        if ctx.kept is not None:
                            (kl, pair), rows = ctx.kept, slice(lo * K, hi * K)
                            kept = (None if pair is None else pair[0][rows], kl[rows], None if pair is None else pair[1][rows])
                        logits, pe = autograd.PairMlpFn.apply(aq, kt_l, r, wp_l, W2, b2, P2l, c2l, ic, kept)
                    else:
                        a = autograd.AttnInLinearFn.apply(aq, kt_l, r, wp_l, ic)                     # aq_i - kt_j + Wp r
                        logits = L(a, W2, b2, True, False, None, None)                                    # W2 relu(.) + b2
                        pe = L(r, P2l, c2l, False, False, None, None)
                    out = autograd.SoftmaxAggGradOnlyFn.apply(logits, vt_l, pe, ic)      # (value unused: no launch)
                    grads = torch.autograd.grad(out, [aq] + leaves, g[lo:hi], allow_unused=True)
                    assert all(gr is None for gr in grads[1:]), 'a leaf gradient bypassed its sink'
                    g_aq[lo:hi] = grads[0]
                (gx, _, _) = torch.autograd.grad(aq_all, [x_all, wq_l, bq_l], g_aq, allow_unused=True)
                # key / value tables -> abstract features, Wk' and Wv  (the table gradients are complete: sums() joins)
                (g_kt, g_vt) = sink.sums()[:2]
                (gx2a, _) = torch.autograd.grad(kt, [x2d, wk_l], g_kt, allow_unused=True)
                (gx2b, g_Wv) = _grad_or_none(vt, [x2d, Wv], g_vt)
                (g_kt, g_vt, g_wp, g_P1, g_c1, g_P2, g_c2, g_W2, g_b2, g_wq, g_bq, g_wk) = sink.sums()
            # merged matrices -> original parameters (frozen ones are skipped: autograd.grad rejects them).  Every input of
            # this small fp64 pass is a sum the parameter-gradient stream produced and every output a parameter gradient:
            # the whole pass is deposited there (autograd._deposit: inside gradient_overlap() it leaves the data-gradient
            # chain; the gradients then reach .grad through the overlap's sums, and this Function reports None for them)
            targets = (W1, b1, Wq, Wk, P2, c2, P2, c2, lp0.weight, lp0.bias, la2.weight, la2.bias)

            def to_parameters():
                return _merged_to_parameters(W1, b1, Wq, Wk, P2, c2, g_wq, g_bq, g_wk, g_wp) + \
                    (g_P2, g_c2, g_P1, g_c1, g_W2, g_b2)
            res = autograd._deposit(tuple(t if t.requires_grad else None for t in targets), to_parameters,
                                    g_wq, g_bq, g_wk, g_wp, g_P2, g_c2, g_P1, g_c1, g_W2, g_b2)
            (d_W1, d_b1, d_Wq, d_Wk, d_P2, d_c2, g_P2, g_c2, g_P1, g_c1, g_W2, g_b2) = res
        add = lambda u, v: u if v is None else (v if u is None else u + v)   # noqa: E731
        by_param = {id(Wq): d_Wq, id(Wk): d_Wk, id(Wv): g_Wv, id(lp0.weight): g_P1, id(lp0.bias): g_c1,
                    id(P2): add(g_P2, d_P2), id(c2): add(g_c2, d_c2), id(W1): d_W1, id(b1): d_b1,
                    id(la2.weight): g_W2, id(la2.bias): g_b2}
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


class PointTransformerLayer(nn.Module, kernels.HasKernelSelection):
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
        self._path = {}

    # -- derived weights ---------------------------------------------------------------
    def _params_key(self, *extra):
        ps = list(self.parameters()) + [p for m in extra if m is not None for p in (m.weight, m.bias)]
        return (weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in ps)

    def path_weights(self, cross, pre=None, post=None, flags=None):
        """occ4d_pt_layer_weights over this layer's parameters in the reference's layout (+ `pre` = the block's layer1,
        `post` = its layer3), and the library's prepared buffer for them (merged matrices of refactoring (i), formed in
        fp64 and rounded once, and the stage-packed weight streams: occ4d_pt_layer_prepare_f32).  Cached while the
        parameters (storage, version, weights epoch) and the kernel-selection flags are unchanged.  `flags`: the owning
        module's selection (a decoder prepares its cross layers under ITS flags); None = this layer's own.  One cache
        entry per flag value: two threads running this layer under different selections do not evict each other."""
        if flags is None:
            flags = path_flags(self)
        slot = (bool(cross), pre is not None, post is not None, flags)
        key = (flags,) + self._params_key(pre, post)
        hit = self._path.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2], flags
        L = ops._lib
        keep = []

        def dp(t):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            ops._dev(t, name='parameter')                  # (RuntimeError for CPU tensors: there is no fallback path)
            keep.append(t)
            return t.data_ptr()
        w = L.PtLayerWeights(dim=self.dim, dim2=self.dim2 if cross else self.dim, pos_hidden=self.pos_mlp[0].out_features,
                             cross=int(bool(cross)), d_in=pre.in_features if pre is not None else self.dim,
                             d_out=post.out_features if post is not None else self.dim)
        w.to_q, w.to_k, w.to_v = dp(self.to_q.weight), dp(self.to_k.weight), dp(self.to_v.weight)
        w.pos0_w, w.pos0_b = dp(self.pos_mlp[0].weight), dp(self.pos_mlp[0].bias)
        w.pos2_w, w.pos2_b = dp(self.pos_mlp[2].weight), dp(self.pos_mlp[2].bias)
        w.attn0_w, w.attn0_b = dp(self.attn_mlp[0].weight), dp(self.attn_mlp[0].bias)
        w.attn2_w, w.attn2_b = dp(self.attn_mlp[2].weight), dp(self.attn_mlp[2].bias)
        if pre is not None:
            w.pre_w, w.pre_b = dp(pre.weight), dp(pre.bias)
        if post is not None:
            w.post_w, w.post_b = dp(post.weight), dp(post.bias)
        w._keep = keep
        prepared = ops.pt_layer_prepare(w, flags, self.to_q.weight.device)
        self._path[slot] = (key, w, prepared)
        return w, prepared, flags

    def merged_weights(self, pre=None):
        """The fp64-merged matrices of refactoring (i) as the LIBRARY forms them (views of the prepared buffer of
        occ4d_pt_layer_prepare_f32, layout of csrc/path.hip): wq = (W1 Wq [L1]), bq = W1 c2 + b1 [+ (W1 Wq) l1_b],
        wk = W1 Wk, wp = W1 P2.  For inspection and tests (tests/test_gpu_contracts.py compares them with the as-written
        expression in fp64); the forward passes never read them from Python."""
        w, prepared, _ = self.path_weights(cross=True, pre=pre)
        D, D2, h = self.dim, self.dim2, self.pos_mlp[0].out_features
        kq = pre.in_features if pre is not None else D
        up = lambda n: (n + 63) // 64 * 64          # noqa: E731   (ALIGN of csrc/path.hip)
        o, out = 0, {}
        for name, shape in (('wq', (2 * D, kq)), ('bq', (2 * D,)), ('wk', (2 * D, D2)), ('wp', (2 * D, h))):
            n = 1
            for v in shape:
                n *= v
            out[name] = prepared[o:o + n].view(*shape)
            o += up(n)
        return out

    # -- forward -----------------------------------------------------------------------
    def forward(self, x, pos, x2=None, pos2=None):
        """x (B,N,D), pos (B,N,3) [, x2 (B,M,D2), pos2 (B,M,3)] -> agg (B,N,D)."""
        with kernels.use(kernels.current(self)):
            return self._forward(x, pos, x2, pos2, pre=None)

    def _forward(self, x, pos, x2, pos2, pre, knn_idx=None, post=None):
        """`pre` / `post`: the Linear layers of the PointTransformerBlock around this layer (layer1, layer3 + residual).
        Inference: ONE library call per cloud runs the whole block (occ4d_pt_layer_fwd_f32) and the block's output is
        returned.  Training (autograd recording): the differentiable kernels; `post` is left to the caller."""
        if needs_grad(self, x, x2) or (pre is not None and needs_grad(pre)) or (post is not None and needs_grad(post)):
            out = []
            for b in range(x.shape[0]):
                y = x[b] if pre is None else autograd.linear(x[b], pre)
                sel = kernels.current(self)
                if (sel.checkpoint_attention and x2 is not None and self.dim in ops.FUSED_ATTN_DIMS
                        and self.num_neighbors <= ops.FUSED_ATTN_MAX_K and self.pos_mlp[0].out_features == 32):
                    idx = knn_idx[b] if knn_idx is not None else ops.knn(pos[b].detach(), pos2[b].detach(),
                                                                          self.num_neighbors, metric=0)
                    out.append(_CheckpointedAttention.apply(self, y, pos[b].detach(), x2[b], pos2[b].detach(), idx,
                                                            *self.parameters()))
                    continue
                if (sel.stored_attention_form == 'merged' and x2 is not None and self.pos_mlp[0].out_features == 32
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
            out.append(self._forward_one(x[b], pos[b], None if x2 is None else x2[b], None if pos2 is None else pos2[b],
                                         pre, post, None if knn_idx is None else knn_idx[b]))
        return ops.stack_batch(out)

    def _forward_one(self, x, pos, x2, pos2, pre=None, post=None, knn_idx=None, logits_out=None, pair_out=None):
        """Inference forward of one cloud through the library's path-level entry point (`logits_out`: the training forward
        of _CheckpointedAttention, ops.pt_layer_fwd)."""
        w, prepared, flags = self.path_weights(cross=x2 is not None, pre=pre, post=post)
        if knn_idx is not None and knn_idx.dtype != torch.int32:
            knn_idx = knn_idx.to(torch.int32)
        return ops.pt_layer_fwd(w, prepared, x, pos, x2, pos2, self.num_neighbors, flags, knn_idx=knn_idx,
                                logits_out=logits_out, pair_out=pair_out)

    def logits_storable(self):
        """Can this layer's fused forward leave its logits in HBM (ops.logits_storable: cross attention, dim 416, k <= 14,
        the fp32 or the bf16 x 3 kernel of the current selection)?"""
        f, L = path_flags(self), ops._lib
        off = L.PATH_UNFUSED | L.PATH_SPLIT_F16 | L.PATH_FIRST_GEN
        return self.dim == ops.TRUNK_WIDTH and self.num_neighbors <= ops.FUSED_ATTN_MAX_K and not (f & off)

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
        kt = L(x2, wk, None, False, False, None, None)                       # (M, 2D)
        vt = L(x2, self.to_v.weight, None, False, False, None, None)         # (M, D)
        aq = L(x, wq, bq, False, False, None, None)                          # (N, 2D)
        r = autograd.PosHiddenFn.apply(pos, pos2, idx, P1, c1)         # (N*K, 32)
        if autograd.pair_mlp_fused_ok(aq, r, idx):
            logits, pe = autograd.PairMlpFn.apply(aq, kt, r, wp, W2, b2, P2, c2, idx)
        else:
            a = autograd.AttnInLinearFn.apply(aq, kt, r, wp, idx)                       # aq_i - kt_j + Wp r
            logits = L(a, W2, b2, True, False, None, None)                   # W2 relu(.) + b2
            pe = L(r, P2, c2, False, False, None, None)
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
