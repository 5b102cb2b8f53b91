"""Encoder building blocks on the HIP library.

Interface mirror of the reference's model/modules.py: ``PointTransformerBlock``
(:18-67) and ``DownTransition`` (:70-163) with the reference's constructor
arguments, parameter names (layer1/layer2/layer3, mlp.0[/mlp.1]) and forward
signatures.  ``UpTransition`` (:166-289) is not instantiated by any published
config (enable_decoder=False, train.py:223) and is out of scope.
"""
import os

import numpy as np
import torch

from . import kernels
from . import ops
from . import point_transformer_layer
from . import autograd
from .point_transformer_layer import needs_grad


class PointTransformerBlock(torch.nn.Module, kernels.HasKernelSelection):
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

    def forward(self, x, p, x2=None, p2=None, knn_idx=None):
        """x (B,N,d_in), p (B,N,3) [, x2 (B,M,E), p2 (B,M,3)] -> (z (B,N,d_out), p).
        `knn_idx` (extension, optional): (B,N,num_neighbors) int32 result of kNN_torch(p, p2) when the caller already
        has it.  Inference: the whole block -- layer1, the vector attention, layer3 + residual -- is ONE call of the
        library's path-level entry point per cloud (occ4d_pt_layer_fwd_f32)."""
        with kernels.use(kernels.current(self)):
            return self._forward(x, p, x2, p2, knn_idx)

    def _forward(self, x, p, x2, p2, knn_idx):
        assert x.shape[:2] == p.shape[:2]
        if x2 is not None:
            assert x2.shape[:2] == p2.shape[:2]
        if needs_grad(self, x, x2):
            agg = self.layer2._forward(x, p, x2, p2, pre=self.layer1, knn_idx=knn_idx)
            z = ops.stack_batch([autograd.linear(agg[b], self.layer3, residual=x[b]) for b in range(x.shape[0])])
            return (z, p)
        assert self.d_out == self.d_in, 'PointTransformerBlock adds its input to layer3\'s output (model/modules.py:66)'
        z = self.layer2._forward(x, p, x2, p2, pre=self.layer1, knn_idx=knn_idx, post=self.layer3)
        return (z, p)


# Exact-in-R refactoring (iv): the farthest-point subsets of consecutive DownTransitions are NESTED.  With
# random_start=False every level starts at its point 0, which is the original point 0 at every level (it is always
# selected and the subsets are sorted by index), so level l + 1 runs the greedy rule on S_l = the first m_l picks of
# level 0.  A pick of the full-cloud run maximises the running min-distance over ALL points and lies in S_l, hence it also
# maximises over S_l, and the lowest-index tie rule picks the same point in both (the restricted maximiser set is the
# full one intersected with S_l and contains the full run's winner, its minimum).  By induction the first m_(l+1) picks
# of level 0's selection ORDER are level l + 1's subset: the FPS launches of levels 1, 2 (0.90 + 0.27 of the 4.07 ms
# chain at 14336 points) are replaced by a prefix of level 0's order, bit for bit (tests/test_gpu_parity.py, G5).
NESTED_FPS = os.environ.get('OCC4D_NESTED_FPS', '1') != '0'


class NestedFps:
    """Selection order of the first deterministic FPS of a chain of DownTransitions over ONE cloud, and the original
    (level-0) index of every point of the current level's cloud."""

    def __init__(self):
        self.order = None          # (m_0) int32 original indices in selection order
        self.orig = None           # (n_l) int32 ascending original indices of the current cloud's points

    def begin(self, order, inds_sorted):
        self.order, self.orig = order, inds_sorted

    def usable(self, n_points, m):
        return (self.order is not None and self.orig.shape[0] == n_points and m <= self.order.shape[0]
                and n_points <= 32768)

    def next_level(self, m):
        """Ascending int32 positions (in the current cloud) of the next level's subset = the first m picks: one small
        library kernel (binary search of every pick in the sorted `orig` + bit-set compaction; round 5 -- the
        searchsorted / sort / index sequence of round 4 was the last ATen work on the encode path)."""
        inds, self.orig = ops.nested_fps_level(self.order, self.orig, m)
        return inds


POOL_FROM_SELF_KNN = os.environ.get('OCC4D_POOL_FROM_SELF_KNN', '1') != '0'      # (0: one kNN launch per DownTransition)


def pool_neighbours_from_self_knn(self_idx, inds, k):
    """The pooling neighbours of a DownTransition from the self-kNN lists of the block before it (exact): the k nearest
    full-cloud points of a SAMPLED point (torch_cluster.knn(x=p, y=p_sub, k), model/modules.py:142-146) are the first k
    entries of that point's own nearest-first list over the same cloud (kNN_torch(p, p, K), K >= k,
    model/point_transformer_layer.py:167) -- same distance expression ((dx*dx + dy*dy) + dz*dz, same tie rule (distance,
    then lowest index: a strict total order, so the top k is a prefix of the top K), and p_sub's coordinates are copies of
    p's.  self_idx (N, K) int32, inds (n_new) int32 -> (n_new, k) int32: a row gather instead of a kNN launch (the int32
    bit patterns travel through the fp32 gather kernel untouched)."""
    assert self_idx.dtype == torch.int32 and self_idx.is_contiguous() and self_idx.shape[1] >= k
    return ops.gather_rows(self_idx.view(torch.float32), inds, cols=k).view(torch.int32)


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
            # (model/modules.py:98-102: eps 1e-3, default momentum)  eval mode: running statistics; training mode (round 5):
            # batch statistics over the B N rows of the level + running-statistics update (csrc/batchnorm.hip)
            self.mlp = torch.nn.Sequential(torch.nn.Linear(d_in, d_out), torch.nn.BatchNorm1d(d_out, eps=1e-3),
                                           torch.nn.ReLU())
        else:
            raise ValueError()

    def sample(self, p, nested=None):
        """Farthest-point subset of ONE cloud p (N,3): (ascending indices (n_new) int32, their coordinates (n_new,3)).
        Depends on coordinates only: the encoder runs the three levels of this dependent chain (the FPS steps are
        one long chain on a single CU) back to back on a side stream (model.py).  `nested`: the NestedFps of the chain
        this cloud belongs to; with a deterministic start the subset then comes from the first level's selection order."""
        n_new = int(np.ceil(p.shape[0] / self.factor))
        if nested is not None and NESTED_FPS and not self.fps_random_start:
            if nested.usable(p.shape[0], n_new):
                inds = nested.next_level(n_new)
                return (inds, ops.gather_rows(p, inds))
            inds, order = ops.fps_auto(p, n_new, start=0, return_order=True)
            nested.begin(order, inds)
            return (inds, ops.gather_rows(p, inds))
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
        if self.norm_type == 'batch' and self.mlp[1].training:
            return self._forward_batchnorm_training(x, p, geometry, train)
        if self.norm_type == 'batch' and train:
            raise NotImplementedError("DownTransition(norm_type='batch'): gradients through an eval-mode BatchNorm (frozen "
                                      "running statistics) are not implemented; train with .train() or run without grad")
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
            # Linear [+ LayerNorm | BatchNorm (eval)] + ReLU on ALL points, then the K-way max pool: one library call
            nm = self.mlp[1]
            if self.norm_type == 'layer':
                z = ops.down_pool_fwd(x[b], lin.weight, lin.bias, nn_idx, norm=1, gamma=nm.weight, beta=nm.bias, eps=nm.eps)
            elif self.norm_type == 'batch':
                z = ops.down_pool_fwd(x[b], lin.weight, lin.bias, nn_idx, norm=2, gamma=nm.weight, beta=nm.bias,
                                      mean=nm.running_mean, var=nm.running_var, eps=nm.eps)
            else:
                z = ops.down_pool_fwd(x[b], lin.weight, lin.bias, nn_idx, norm=0)
            zs.append(z)
            ps.append(p_sub)
        return (ops.stack_batch(zs), ops.stack_batch(ps))

    def _forward_batchnorm_training(self, x, p, geometry, train):
        """BatchNorm1d in training mode (model/modules.py:98-102, 152: the MLP runs on the flat (B N, d_in) rows, so the
        statistics are those of ALL rows of the level): y = Linear(x); batch mean / biased variance; ReLU(BN(y)); the
        running statistics move by `momentum` towards the batch mean and the UNBIASED batch variance, as torch does."""
        (B, N, _) = x.shape
        lin, bn = self.mlp[0], self.mlp[1]
        geom = [geometry[b] if geometry is not None else self.geometry(p[b].detach()) for b in range(B)]
        ys = [autograd.linear(x[b], lin) if train else ops.linear(x[b], lin.weight, lin.bias) for b in range(B)]
        y_all = ys[0] if B == 1 else torch.cat(ys, dim=0)
        if train:
            out, mean, var = autograd.BatchNormReluFn.apply(y_all, bn.weight, bn.bias, bn.eps)
        else:
            out, mean, var = ops.bn_train_fwd(y_all, bn.weight, bn.bias, bn.eps)
        if bn.track_running_stats:
            with torch.no_grad():
                rows = y_all.shape[0]
                bn.num_batches_tracked += 1
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
                bn.running_var.mul_(1.0 - m).add_(var, alpha=m * rows / max(rows - 1, 1))
        zs, ps = [], []
        for b in range(B):
            (inds, p_sub, nn_idx) = geom[b]
            y_b = out[b * N:(b + 1) * N]
            zs.append(autograd.MaxPoolGatherFn.apply(y_b, nn_idx) if train else ops.maxpool_gather(y_b, nn_idx))
            ps.append(p_sub)
        return (ops.stack_batch(zs), ops.stack_batch(ps))
