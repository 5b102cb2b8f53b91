"""Differentiable wrappers (torch.autograd.Function) over the HIP kernels: the training path.

SURVEY.md 8(f) rank 1.  The reference trains through torch autograd over ATen ops
(train.py:101-118); here every forward AND backward computation runs in libocc4d.so.  torch
supplies the tape, tensor glue (cat / slicing) and the optimiser only.  The training forward
uses the as-written op order of model/*.py (no weight merging), built from the unfused kernels;
kNN / FPS carry no gradient (they depend on coordinates only).
"""
import os

import torch
from torch.autograd import Function

from . import ops

# Training-path Linear layers with a 416-wide operand (the decoder trunk, the 416 -> 832 / 832 <- 416 layers of the
# attention MLP and their data gradients) run on the row-resident kernel of the inference trunk (occ4d_rowlin_f32:
# 0.72-0.79 of the fp32 MFMA peak against 0.46-0.65 for the generic kernel at these shapes, profiles/train_shapes.py).
# The stage-packed copy of a weight (and of its transpose, for the data gradient) is rebuilt when the weight changes.
ROWLIN_IN_TRAINING = True
# ... in its half-CU re-cut (csrc/trunk4.hip: 64-row workgroups, two per CU).  The training row counts are not whole
# dispatch rounds of the 8-wave kernel (68812 query rows = 2.1 rounds of 256 x 128 rows: the third round runs 26
# workgroups); a half-CU workgroup that has its CU to itself in the last round runs faster, and at whole rounds the two
# are level: 232 vs 301 us at 68812 rows, 276 vs 302 us at 98304 (profiles/time_rowlin_tail.py).
ROWLIN_HALF_CU = os.environ.get('OCC4D_TRAIN_ROWLIN_HALF_CU', '1') == '1'
_PACKS = {}           # stage-packed copies of nn.Parameters only (small LRU); transient leaves are packed uncached
_PACKS_MAX = 64
_ZEROS = {}           # zero bias vectors, kept apart from the packs and created eagerly (never inside a capture)


def _packed(w, transposed):
    from . import point_transformer_layer as ptl
    src = w.detach().t().contiguous() if transposed else w.detach()
    if not isinstance(w, torch.nn.Parameter):
        # a transient leaf (the merged matrices rebuilt by every _CheckpointedAttention.backward): caching it would only
        # pin the leaf and its packed copy (a few MB each) until the table is cleared -- it can never hit again
        return ops.pack_trunk4_rows(src) if ROWLIN_HALF_CU else ops.pack_trunk_rows(src)
    key = (id(w), bool(transposed))
    tag = (w.data_ptr(), w._version, tuple(w.shape), ptl.weights_epoch(), ROWLIN_HALF_CU)
    hit = _PACKS.get(key)
    if hit is not None and hit[0] is w and hit[1] == tag:
        _PACKS[key] = _PACKS.pop(key)              # most recently used last
        return hit[2]
    packed = ops.pack_trunk4_rows(src) if ROWLIN_HALF_CU else ops.pack_trunk_rows(src)
    _PACKS.pop(key, None)
    while len(_PACKS) >= _PACKS_MAX:
        _PACKS.pop(next(iter(_PACKS)))             # least recently used first
    _PACKS[key] = (w, tag, packed)                 # (w kept alive: its id cannot be recycled under this key)
    return packed


def _zeros(n, device):
    key = (n, str(device))
    z = _ZEROS.get(key)
    if z is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # a buffer first created inside a capture would be a captured memset that has not run yet: an eager use
            # before the first replay would read garbage as the bias.  Capture warm-up steps create every size eagerly.
            raise RuntimeError('zero-bias buffer of %d floats requested for the first time inside a stream capture; '
                               'run one eager step before capturing' % n)
        z = _ZEROS[key] = torch.zeros((n,), dtype=torch.float32, device=device)
    return z


def _linear_fwd(x, w, b, relu_in=False, relu_out=False, residual=None, transposed=False, mask=None):
    """[relu]([relu](x) W'^T + b) + residual with W' = w (or w^T when `transposed`: the data gradient); `mask`: the
    result is zeroed where mask <= 0 (the ReLU of a relu_in layer applied to its data gradient)."""
    n_out, k = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    if (ROWLIN_IN_TRAINING and k == ops.TRUNK_WIDTH and n_out % 32 == 0 and not relu_out and x.shape[0] >= 1024
            and x.is_contiguous() and (residual is None or residual.is_contiguous())):
        bias = b if b is not None else _zeros(n_out, x.device)
        if mask is not None and not mask.is_contiguous():
            mask = mask.contiguous()
        return ops.rowlin(x, _packed(w, transposed), bias, n_out, relu_in=relu_in, residual=residual, mask=mask)
    y = ops.linear(x, w.t().contiguous() if transposed else w, b, relu_in=relu_in, relu_out=relu_out,
                   residual=residual)
    return y if mask is None else ops.relu_mask(y, mask)


class MatmulF64Fn(Function):
    """c = a @ b in fp64 on the library (the merged weights of DESIGN.md 4 (i) and, in training, their gradients back
    to the original parameters); b may be a vector."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.vec = b.dim() == 1
        b2 = b[:, None] if ctx.vec else b
        ctx.save_for_backward(a, b2)
        c = ops.matmul_f64(a, b2)
        return c[:, 0] if ctx.vec else c

    @staticmethod
    def backward(ctx, dc):
        a, b2 = ctx.saved_tensors
        dc2 = dc[:, None] if ctx.vec else dc
        da = ops.matmul_f64(dc2, b2.t()) if ctx.needs_input_grad[0] else None
        db = ops.matmul_f64(a.t(), dc2) if ctx.needs_input_grad[1] else None
        if db is not None and ctx.vec:
            db = db[:, 0]
        return da, db


def matmul64(a, b):
    """fp64 product of the merged-weight algebra.  CUDA tensors: the library kernel.  CPU tensors only occur when the
    algebra itself is checked on the host (tests/test_host.py builds a module on the CPU and compares the merged
    matrices with the as-written expression); such a module cannot run a forward pass -- every op rejects CPU tensors."""
    if not a.is_cuda:
        return a @ b
    return MatmulF64Fn.apply(a, b)


class LinearFn(Function):
    """y = [relu]( [relu](x) W^T + b ) + residual   (never relu_out together with residual)."""

    @staticmethod
    def forward(ctx, x, w, b, relu_in, relu_out, residual):
        assert not (relu_out and residual is not None)
        y = _linear_fwd(x, w, b, relu_in=relu_in, relu_out=relu_out, residual=residual)
        ctx.flags = (relu_in, relu_out, b is not None, residual is not None)
        ctx.save_for_backward(x, w, y if relu_out else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        relu_in, relu_out, has_b, has_res = ctx.flags
        dy = dy.contiguous()
        g = ops.relu_mask(dy, y) if relu_out else dy
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = _linear_fwd(g, w, None, transposed=True, mask=x if relu_in else None)
        want_db = has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            # one kernel: dW = g^T [relu](x) on the MFMA, db = column sums of the g tiles it stages
            res = ops.linear_wgrad(g, x, bias=want_db, relu_x=relu_in)
            (dw, db) = res if want_db else (res, None)
        elif want_db:
            db = ops.colsum(g)
        if has_res and ctx.needs_input_grad[5]:
            dres = dy
        return dx, dw, db, None, None, dres


def linear(x, lin, relu_in=False, relu_out=False, residual=None):
    """x (n,K) through an nn.Linear's parameters."""
    return LinearFn.apply(x, lin.weight, lin.bias, relu_in, relu_out, residual)


class PosHiddenFn(Function):
    """r = relu(P1 (pos_i - pos2_j) + c1); gradients to P1, c1 only (coordinates are data)."""

    @staticmethod
    def forward(ctx, pos, pos2, idx, P1, c1):
        r = ops.pt_pos_hidden(pos, pos2, idx, P1, c1)
        ctx.save_for_backward(pos, pos2, idx, r)
        return r

    @staticmethod
    def backward(ctx, gr):
        pos, pos2, idx, r = ctx.saved_tensors
        dP1, dc1 = ops.pt_pos_hidden_bwd(pos, pos2, idx, r, gr.contiguous())
        return None, None, None, dP1, dc1


class AttnInFn(Function):
    """a[p] = q[i] - kf[idx[p]] + pe[p]."""

    @staticmethod
    def forward(ctx, q, kf, pe, idx):
        ctx.save_for_backward(idx)
        ctx.m = kf.shape[0]
        return ops.pt_attn_in(q, kf, pe, idx)

    @staticmethod
    def backward(ctx, da):
        (idx,) = ctx.saved_tensors
        da = da.contiguous()
        k = idx.shape[1]
        return ops.segment_sum(da, k), ops.scatter_add_rows(da, idx, ctx.m, scale=-1.0), da, None


class AttnInLinearFn(Function):
    """a[p] = q[i] - kf[idx[p]] + r[p] Wp^T  (p = i k + j): AttnInFn fused with the K = 32 Linear that feeds it -- the
    (N k, 2D) product Wp r is never written on its own and the gather / subtract pass over it disappears (the generic
    Linear kernel adds the gathered rows in its epilogue)."""

    @staticmethod
    def forward(ctx, q, kf, r, wp, idx):
        k = idx.shape[1]
        a = ops.linear(r, wp, add_rows=q, add_div=k, sub_rows=kf, sub_idx=idx.reshape(-1))
        ctx.save_for_backward(r, wp, idx)
        ctx.m = kf.shape[0]
        return a

    @staticmethod
    def backward(ctx, da):
        r, wp, idx = ctx.saved_tensors
        da = da.contiguous()
        k = idx.shape[1]
        dq = ops.segment_sum(da, k) if ctx.needs_input_grad[0] else None
        dkf = ops.scatter_add_rows(da, idx, ctx.m, scale=-1.0) if ctx.needs_input_grad[1] else None
        dr = _linear_fwd(da, wp, None, transposed=True) if ctx.needs_input_grad[2] else None
        dwp = ops.linear_wgrad(da, r) if ctx.needs_input_grad[3] else None
        return dq, dkf, dr, dwp, None


PAIR_MLP_FUSED = os.environ.get('OCC4D_PAIR_MLP', '1') == '1'


def pair_mlp_fused_ok(aq, r, idx):
    """The fused pair-tensor kernel is built for d = 416 and 32 positional hidden units, with 32-bit row offsets."""
    d = ops.TRUNK_WIDTH
    return (PAIR_MLP_FUSED and aq.shape[1] == 2 * d and r.shape[1] == 32
            and idx.numel() * 2 * d * 4 < 2 ** 32 and aq.shape[0] * aq.stride(0) * 4 < 2 ** 32)


class PairMlpFn(Function):
    """(logits, pe) of the merged-form layer from ONE kernel (ops.pt_pair_mlp), for the chain
        a = aq_i - kt_j + Wp r;  logits = W2 relu(a) [+ b2];  pe = P2 r + c2
    (AttnInLinearFn -> LinearFn(relu_in) -> LinearFn in separate launches otherwise).  b2 is not added: the logits only
    feed the softmax over the neighbour axis, where a per-channel constant cancels; its gradient (the column sums of
    dlogits, zero up to rounding) is still returned.  backward = the backward passes of those three Functions."""

    _stream = None       # (W2, wp, P2, their versions, packed stream) of the last call: the recompute walks the queries
                         # in chunks with the SAME weight tensors (identity + version: held here, so no address is recycled)

    @staticmethod
    def _packed_stream(W2, b2, wp, P2, c2):
        hit = PairMlpFn._stream
        if (hit is not None and hit[0] is W2 and hit[1] is wp and hit[2] is P2
                and hit[3] == (W2._version, wp._version, P2._version)):
            return hit[4]
        stream = ops.pack_attn16p_stream(W2, b2, wp, P2, c2)
        PairMlpFn._stream = (W2, wp, P2, (W2._version, wp._version, P2._version), stream)
        return stream

    @staticmethod
    def forward(ctx, aq, kt, r, wp, W2, b2, P2, c2, idx):
        stream = PairMlpFn._packed_stream(W2, b2, wp, P2, c2)
        a, logits, pe = ops.pt_pair_mlp(aq, kt, r, idx, c2, stream)
        ctx.save_for_backward(a, r, wp, W2, P2, idx)
        ctx.m = kt.shape[0]
        return logits, pe

    @staticmethod
    def backward(ctx, dlogits, dpe):
        a, r, wp, W2, P2, idx = ctx.saved_tensors
        need = ctx.needs_input_grad
        k = idx.shape[1]
        dlogits, dpe = dlogits.contiguous(), dpe.contiguous()
        dW2 = db2 = dP2 = dc2 = dwp = daq = dkt = dr = None
        if need[4] or need[5]:
            if need[4]:
                res = ops.linear_wgrad(dlogits, a, bias=bool(need[5]), relu_x=True)
                (dW2, db2) = res if need[5] else (res, None)
            else:
                db2 = ops.colsum(dlogits)
        da = _linear_fwd(dlogits, W2, None, transposed=True, mask=a)          # (x > 0) . (g W2)
        if need[0]:
            daq = ops.segment_sum(da, k)
        if need[1]:
            dkt = ops.scatter_add_rows(da, idx, ctx.m, scale=-1.0)
        if need[3]:
            dwp = ops.linear_wgrad(da, r)
        if need[6] or need[7]:
            if need[6]:
                res = ops.linear_wgrad(dpe, r, bias=bool(need[7]))
                (dP2, dc2) = res if need[7] else (res, None)
            else:
                dc2 = ops.colsum(dpe)
        if need[2]:
            dr = _linear_fwd(da, wp, None, transposed=True)
            dr += _linear_fwd(dpe, P2, None, transposed=True)
        return daq, dkt, dr, dwp, dW2, db2, dP2, dc2, None


class SoftmaxAggFn(Function):
    """agg[i] = sum_j softmax_j(logits / sqrt(d)) * (v[idx] + pe)."""

    @staticmethod
    def forward(ctx, logits, v, pe, idx):
        ctx.save_for_backward(logits, v, pe, idx)
        return ops.pt_softmax_agg(logits, v, pe, idx)

    @staticmethod
    def backward(ctx, dagg):
        logits, v, pe, idx = ctx.saved_tensors
        dlogits, dpe, dv = ops.pt_softmax_agg_bwd(logits, v, pe, idx, dagg.contiguous())
        return dlogits, dv, dpe, None


class SoftmaxAggGradOnlyFn(Function):
    """SoftmaxAggFn for a caller that only differentiates: the recompute-in-backward path rebuilds the pair tensors to
    back-propagate a GIVEN output gradient and never looks at the output value (the forward pass already produced it
    with the fused kernel), so forward returns an uninitialised tensor of the right shape and launches nothing."""

    @staticmethod
    def forward(ctx, logits, v, pe, idx):
        ctx.save_for_backward(logits, v, pe, idx)
        return logits.new_empty((idx.shape[0], v.shape[1]))

    backward = SoftmaxAggFn.backward


class MaxPoolGatherFn(Function):
    @staticmethod
    def forward(ctx, y, idx):
        ctx.save_for_backward(y, idx)
        return ops.maxpool_gather(y, idx)

    @staticmethod
    def backward(ctx, dz):
        y, idx = ctx.saved_tensors
        return ops.maxpool_gather_bwd(y, idx, dz.contiguous()), None


class LayerNormReluFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        y = ops.layernorm(x, gamma, beta, eps=eps, relu=True)
        ctx.eps = eps
        ctx.save_for_backward(x, gamma, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, y = ctx.saved_tensors
        g = ops.relu_mask(dy.contiguous(), y)
        dx, dgamma, dbeta = ops.layernorm_bwd(x, gamma, g, ctx.eps)
        return dx, dgamma, dbeta, None


class BatchNormReluFn(Function):
    """BatchNorm1d in training mode + ReLU (model/modules.py:98-102,152).  Returns (out, batch mean, biased batch
    variance); the statistics are outputs without gradient (the module updates its running statistics from them)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, eps):
        out, mean, var = ops.bn_train_fwd(y, gamma, beta, eps)
        ctx.eps = eps
        ctx.save_for_backward(y, gamma, out, mean, var)
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        y, gamma, out, mean, var = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_train_bwd(y, g.contiguous(), out, mean, var, gamma, ctx.eps)
        return dx, dgamma, dbeta, None


class SwishFn(Function):
    """x sigmoid(x) (model/implicit.py:46-64, the reference's other activation) with its analytic backward."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.swish(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.swish_bwd(g.contiguous(), x)


def act_linear(x, lin, activation, residual=None):
    """lin(activation(x)) [+ residual] for the training path: 'relu' folds into the Linear kernels (relu_in), 'swish' is an
    element-wise pass in front of a plain Linear."""
    if activation == 'relu':
        return linear(x, lin, relu_in=True, residual=residual)
    if activation == 'swish':
        return linear(SwishFn.apply(x), lin, residual=residual)
    raise ValueError('Unknown activation: ' + str(activation))


class MeanRowsFn(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.n = x.shape[0]
        return ops.mean_rows(x)

    @staticmethod
    def backward(ctx, dout):
        return ops.broadcast_rows(dout.contiguous(), ctx.n, 1.0 / ctx.n)


class ExpandRowsFn(Function):
    """v (d) -> (n, d), every row = v: `v[None].expand(n, d)` with a backward on the library's column-sum kernels.
    torch's own backward of expand is a multi-block reduction whose completion counters are cleared by a cudaMemsetAsync:
    inside a captured step that is a MEMSET NODE, and on this HIP runtime memset nodes are not ordered reliably against
    the kernel nodes around them (profiles/r04_graph_replay_probe.txt: the replayed step computed a wrong gradient for the
    global embedding whenever an eager kernel had run since the last device-wide synchronisation)."""

    @staticmethod
    def forward(ctx, v, n):
        return ops.broadcast_rows(v, n, 1.0)

    @staticmethod
    def backward(ctx, dout):
        return ops.colsum(dout), None


class InterpFn(Function):
    """y[i] = sum_j w[i,j] table[idx[i,j]]  (inverse-distance feature interpolation)."""

    @staticmethod
    def forward(ctx, table, idx, w):
        ctx.save_for_backward(idx, w)
        ctx.m = table.shape[0]
        y = torch.zeros((idx.shape[0], table.shape[1]), dtype=torch.float32, device=table.device)
        return ops.interp_add(y, None, table, idx, w)

    @staticmethod
    def backward(ctx, dy):
        idx, w = ctx.saved_tensors
        return ops.interp_bwd(dy.contiguous(), idx, w, ctx.m), None, None
