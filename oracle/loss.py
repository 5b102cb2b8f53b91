"""CPU restatement of the reference's training losses.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows pipeline.py:198-212 (pre-loss squashing inside MyTrainPipeline.handle_frame) and loss.py:50-294
(MyLosses: implicit_density_loss :50-64, implicit_color_loss :66-154 for the rgb / rgb_nosigmoid modes,
implicit_segm_loss :156-173, implicit_track_loss :175-194, per_example :196-252, entire_batch :254-294).
Pinned by tests/golden/g14_loss_*.npz, which oracle/gen_golden.py produced by running the reference's own
handle_frame / per_example / entire_batch on the same seeded tensors.
"""
import torch
import torch.nn.functional as F


def squash(raw, color_mode):
    """pipeline.py:198-212: density / mark_track / segmentation stay logits."""
    out = raw.clone()
    if color_mode == 'rgb':
        out[..., 1:4] = torch.sigmoid(raw[..., 1:4])
    elif color_mode == 'rgb_nosigmoid':
        out[..., 1:4] = torch.clamp(raw[..., 1:4], min=0.0, max=1.0)
    elif color_mode == 'hsv':
        out[..., 13:15] = torch.clamp(raw[..., 13:15], min=0.0, max=1.0)
    return out


def density_term(o, y):                                           # loss.py:50-64
    return F.binary_cross_entropy_with_logits(o[..., 0], y[..., 0])


def color_term(o, y):                                             # loss.py:66-83 (rgb, rgb_nosigmoid)
    keep = torch.logical_and(y[..., 0] >= 0.1, y[..., 1] >= 0.0)
    return F.l1_loss(o[keep][..., 1:4], y[keep][..., 1:4])


def segm_term(o, y, semantic_classes):                            # loss.py:156-173
    lab = y[..., -1].type(torch.int64)
    keep = lab >= 0
    return F.cross_entropy(o[..., -semantic_classes:][keep], lab[keep])


def track_term(o, y, track_idx=4):                                # loss.py:175-194
    keep = torch.logical_and(y[..., 0] >= 0.1, y[..., 4] >= 0.0)
    return F.binary_cross_entropy_with_logits(o[keep][..., track_idx], y[keep][..., 4])


def training_loss(raw, target, density_lw, color_lw, segmentation_lw, tracking_lw, color_mode, semantic_classes=13):
    """raw (T,B,N,G) decoder outputs, target (T,B,N,6) -> (total, [rgb, dens, segm, track] means)."""
    assert color_mode in ('rgb', 'rgb_nosigmoid'), 'only the published colour modes are restated'
    (T, B) = raw.shape[:2]
    lists = {k: [] for k in ('rgb', 'dens', 'segm', 'track')}
    for i in range(B):                                            # loss.py:222-241: example-major, frame-minor
        for t in range(T):
            o = squash(raw[t, i:i + 1], color_mode)
            y = target[t, i:i + 1]
            if density_lw > 0.0:
                lists['dens'].append(density_term(o, y))
            if color_lw > 0.0:
                lists['rgb'].append(color_term(o, y))
            if segmentation_lw > 0.0:
                lists['segm'].append(segm_term(o, y, semantic_classes))
            if tracking_lw > 0.0:
                lists['track'].append(track_term(o, y))
    means = {k: (torch.mean(torch.stack(v)) if v else 0.0) for k, v in lists.items()}     # loss.py:243-250
    total = (means['rgb'] * color_lw + means['dens'] * density_lw + means['segm'] * segmentation_lw
             + means['track'] * tracking_lw)                                              # loss.py:276-277
    return total, [means['rgb'], means['dens'], means['segm'], means['track']]
