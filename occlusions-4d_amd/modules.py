"""Encoder building blocks on the HIP library.

Interface mirror of the reference's model/modules.py: ``PointTransformerBlock``
(:18-67) and ``DownTransition`` (:70-163) with the reference's constructor
arguments, parameter names (layer1/layer2/layer3, mlp.0[/mlp.1]) and forward
signatures.  ``UpTransition`` (:166-289) is not instantiated by any published
config (enable_decoder=False, train.py:223) and is out of scope.
"""
import numpy as np
import torch

from . import ops
from . import point_transformer_layer
from . import autograd
from .point_transformer_layer import needs_grad


class PointTransformerBlock(torch.nn.Module):
    """z = x + layer3(PointTransformerLayer(layer1(x), p[, x2, p2]))."""

    def __init__(self, d_in, d_hidden, d_out, num_neighbors=16, d_hidden_abstract=None):
        super().__init__()
        self.d_in, self.d_hidden, self.d_out = d_in, d_hidden, d_out
        self.num_neighbors = num_neighbors
        self.layer1 = torch.nn.Linear(d_in, d_hidden)
        self.layer2 = point_transformer_layer.PointTransformerLayer(
            d_hidden, pos_mlp_hidden_dim=32, attn_mlp_hidden_mult=2, num_neighbors=num_neighbors,
            dim2=d_hidden_abstract)
        self.layer3 = torch.nn.Linear(d_hidden, d_out)

    def forward(self, x, p, x2=None, p2=None, scene_owner=None, knn_idx=None, aq_pre=None):
        """x (B,N,d_in), p (B,N,3) [, x2 (B,M,E), p2 (B,M,3)] -> (z (B,N,d_out), p).
        `scene_owner` (extension, optional): tensor object identifying the abstract cloud so
        its key/value tables are computed once per scene instead of once per call.  `knn_idx` (extension, optional):
        (B,N,num_neighbors) int32 result of kNN_torch(p, p2) when the caller already has it.  `aq_pre` (extension,
        optional): (B,N,2 d_hidden) merged query projection of layer1 + layer2 when the caller already has it."""
        assert x.shape[:2] == p.shape[:2]
        if x2 is not None:
            assert x2.shape[:2] == p2.shape[:2]
        # layer1 is folded into the query-side merged matrix (cross) or applied once (self)
        agg = self.layer2._forward(x, p, x2, p2, pre=self.layer1, scene_owner=scene_owner, knn_idx=knn_idx,
                                   aq_pre=aq_pre)
        if needs_grad(self, x, x2):
            z = ops.stack_batch([autograd.linear(agg[b], self.layer3, residual=x[b]) for b in range(x.shape[0])])
        else:
            w3p = point_transformer_layer.trunk_pack(self.layer3.weight) \
                if point_transformer_layer.USE_TRUNK_KERNELS and self.d_out == self.d_in else None
            if w3p is not None:
                z = ops.stack_batch([ops.rowlin(agg[b], w3p, self.layer3.bias, self.d_out, residual=x[b])
                                     for b in range(x.shape[0])])
            else:
                z = ops.stack_batch([ops.linear(agg[b], self.layer3.weight, self.layer3.bias, residual=x[b])
                                     for b in range(x.shape[0])])
        return (z, p)


class DownTransition(torch.nn.Module):
    """Farthest point sampling + kNN + Linear[/LayerNorm]/ReLU on all points + K-way max pool."""

    def __init__(self, d_in, d_out, factor=2, knn_k=8, norm_type='none', fps_random_start=True):
        super().__init__()
        self.d_in, self.d_out, self.factor, self.knn_k = d_in, d_out, factor, knn_k
        self.norm_type = norm_type
        self.fps_random_start = fps_random_start
        if norm_type == 'none':
            self.mlp = torch.nn.Sequential(torch.nn.Linear(d_in, d_out), torch.nn.ReLU())
        elif norm_type == 'layer':
            self.mlp = torch.nn.Sequential(torch.nn.Linear(d_in, d_out), torch.nn.LayerNorm(d_out),
                                           torch.nn.ReLU())
        elif norm_type == 'batch':
            raise NotImplementedError("norm_type 'batch' is unused by every published configuration")
        else:
            raise ValueError()

    def sample(self, p):
        """Farthest-point subset of ONE cloud p (N,3): (ascending indices (n_new) int32, their coordinates (n_new,3)).
        Depends on coordinates only: the encoder runs the three levels of this dependent chain (the FPS steps are
        one long chain on a single CU) back to back on a side stream (model.py)."""
        n_new = int(np.ceil(p.shape[0] / self.factor))
        # torch_cluster draws the first sample at random when random_start (training default); the reference
        # forces False at test time (eval/inference.py:59).  The draw uses torch's global CPU generator.
        start = int(torch.randint(p.shape[0], (1,)).item()) if self.fps_random_start else 0
        inds = ops.fps_auto(p, n_new, start=start)                     # ascending int32
        return (inds, ops.gather_rows(p, inds))                        # (n_new), (n_new,3)

    def neighbours(self, p_sub, p):
        """The knn_k nearest full-cloud points of every sampled point: (n_new,k) int32."""
        return ops.knn(p_sub, p, self.knn_k, metric=0)

    def geometry(self, p):
        """Feature-independent half of forward() for ONE cloud p (N,3): (inds, p_sub, nn_idx)."""
        (inds, p_sub) = self.sample(p)
        return (inds, p_sub, self.neighbours(p_sub, p))

    def forward(self, x, p, geometry=None):
        """x (B,N,d_in), p (B,N,3) -> (z (B,ceil(N/factor),d_out), p_sub (B,ceil(N/factor),3)).
        `geometry` (extension, optional): per-batch-element results of self.geometry(p[b])."""
        assert x.shape[:2] == p.shape[:2]
        (B, N, d_in) = x.shape
        lin = self.mlp[0]
        train = needs_grad(self, x)
        zs, ps = [], []
        for b in range(B):
            (inds, p_sub, nn_idx) = geometry[b] if geometry is not None else self.geometry(p[b].detach())
            if train:
                if self.norm_type == 'layer':
                    ln = self.mlp[1]
                    y = autograd.LayerNormReluFn.apply(autograd.linear(x[b], lin), ln.weight, ln.bias, ln.eps)
                else:
                    y = autograd.linear(x[b], lin, relu_out=True)
                zs.append(autograd.MaxPoolGatherFn.apply(y, nn_idx))
                ps.append(p_sub)
                continue
            if self.norm_type == 'layer':
                y = ops.linear(x[b], lin.weight, lin.bias)
                ln = self.mlp[1]
                ops.layernorm(y, ln.weight, ln.bias, eps=ln.eps, relu=True, out=y)
            else:
                y = ops.linear(x[b], lin.weight, lin.bias, relu_out=True)
            zs.append(ops.maxpool_gather(y, nn_idx))
            ps.append(p_sub)
        return (ops.stack_batch(zs), ops.stack_batch(ps))
