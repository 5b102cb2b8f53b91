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
from .point_transformer_layer import _no_autograd


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

    def forward(self, pcl, return_intermediate):
        """pcl (B,N,d_in) rows (x,y,z,...) -> (pcl_out (B,M,3+D) | None, x_global (B,F) | None,
        layer_coords list | None)."""
        _no_autograd(pcl)
        B = pcl.shape[0]
        pos = pcl[..., :3]
        layer_coords = [pos, pos] if return_intermediate else None
        l0, l2 = self.pre_mlp[0], self.pre_mlp[2]
        x = torch.stack([ops.linear(ops.linear(pcl[b], l0.weight, l0.bias, relu_out=True), l2.weight, l2.bias)
                         for b in range(B)])
        skips = []
        x_global = None
        for i, block in enumerate(self.blocks):
            (x, pos) = block(x, pos)
            if self.output_global_emb and i == self.center_block_idx:
                g0, g2 = self.global_mlp[0], self.global_mlp[2]
                x_global = torch.stack([
                    ops.linear(ops.linear(ops.mean_rows(x[b])[None], g0.weight, g0.bias, relu_out=True),
                               g2.weight, g2.bias)[0] for b in range(B)])
            if return_intermediate:
                layer_coords.append(pos)
            if self.abstract_levels > 1 and isinstance(block, modules.DownTransition):
                for j, skip in enumerate(self.abstract_skip_mlps):
                    if skip.in_features == x.shape[-1]:
                        y = torch.stack([ops.linear(x[b], skip.weight, skip.bias) for b in range(B)])
                        y[..., -1] = j + 1.0
                        skips.append(torch.cat([pos, y], dim=-1))
        if self.output_featurized:
            pcl_out = torch.cat([pos, x], dim=-1)
            if self.abstract_levels > 1:
                pcl_out[..., -1] = self.abstract_levels
                assert len(skips) == self.abstract_levels - 1
                pcl_out = torch.cat([torch.cat(skips, dim=1), pcl_out], dim=1)
        else:
            pcl_out = None
        return (pcl_out, x_global, layer_coords)
