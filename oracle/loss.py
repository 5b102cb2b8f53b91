"""CPU restatement of the reference's training losses.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows pipeline.py:198-212 (pre-loss squashing inside MyTrainPipeline.handle_frame) and loss.py:50-294
(MyLosses: implicit_density_loss :50-64, implicit_color_loss :66-154 for all four colour modes,
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


def rgb_to_hsv(rgb, epsilon=1e-10):                               # utils/utils.py:169-191
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    max_rgb = rgb.max(1)[0]
    min_rgb, argmin = rgb.min(1)
    max_min = max_rgb - min_rgb + epsilon
    h1 = 60.0 * (g - r) / max_min + 60.0
    h2 = 60.0 * (b - g) / max_min + 180.0
    h3 = 60.0 * (r - b) / max_min + 300.0
    h = torch.stack((h2, h3, h1), dim=0).gather(0, argmin[None])[0]
    return torch.stack((h, max_min / (max_rgb + epsilon), max_rgb), dim=1)


def _hue_bins(hsv, n):                                            # loss.py:93-97 / :121-125
    hue = torch.round(hsv[..., 0] / 360.0 * n).type(torch.int64)
    hue[hue == n] = 0
    return hue


def color_term(o, y, color_mode='rgb'):                           # loss.py:66-149
    keep = torch.logical_and(y[..., 0] >= 0.1, y[..., 1] >= 0.0)
    o, y = o[keep], y[keep]
    if color_mode in ('rgb', 'rgb_nosigmoid'):                    # :78-83
        return F.l1_loss(o[..., 1:4], y[..., 1:4])
    hsv = rgb_to_hsv(y[..., 1:4])
    sat, val = hsv[..., 1], hsv[..., 2]
    if color_mode == 'hsv':                                       # :85-114
        hue = _hue_bins(hsv, 12)
        m = torch.logical_and(sat >= 0.2, val >= 0.2)
        loss_hue = F.cross_entropy(o[..., 1:13][m], hue[m]) / 2.0 if m.sum() >= 16 else 0.0
        return (loss_hue + F.l1_loss(o[..., 13], sat) + F.l1_loss(o[..., 14], val)) / 3.0
    assert color_mode == 'bins'                                   # :116-149
    target = _hue_bins(hsv, 6)
    bland = torch.logical_or(sat < 0.3, val < 0.3)
    target[torch.logical_and(val < 0.2, bland)] = 6
    target[torch.logical_and(torch.logical_and(0.2 <= val, val < 0.6), bland)] = 7
    target[torch.logical_and(0.6 <= val, bland)] = 8
    return F.cross_entropy(o[..., 1:10], target) / 3.0


TRACK_IDX = {'rgb': 4, 'rgb_nosigmoid': 4, 'hsv': 15, 'bins': 10}    # utils/utils.py:204-224


def segm_term(o, y, semantic_classes):                            # loss.py:156-173
    lab = y[..., -1].type(torch.int64)
    keep = lab >= 0
    return F.cross_entropy(o[..., -semantic_classes:][keep], lab[keep])


def track_term(o, y, track_idx=4):                                # loss.py:175-194
    keep = torch.logical_and(y[..., 0] >= 0.1, y[..., 4] >= 0.0)
    return F.binary_cross_entropy_with_logits(o[keep][..., track_idx], y[keep][..., 4])


def training_loss(raw, target, density_lw, color_lw, segmentation_lw, tracking_lw, color_mode, semantic_classes=13):
    """raw (T,B,N,G) decoder outputs, target (T,B,N,6) -> (total, [rgb, dens, segm, track] means)."""
    assert color_mode in TRACK_IDX
    (T, B) = raw.shape[:2]
    lists = {k: [] for k in ('rgb', 'dens', 'segm', 'track')}
    for i in range(B):                                            # loss.py:222-241: example-major, frame-minor
        for t in range(T):
            o = squash(raw[t, i:i + 1], color_mode)
            y = target[t, i:i + 1]
            if density_lw > 0.0:
                lists['dens'].append(density_term(o, y))
            if color_lw > 0.0:
                lists['rgb'].append(color_term(o, y, color_mode))
            if segmentation_lw > 0.0:
                lists['segm'].append(segm_term(o, y, semantic_classes))
            if tracking_lw > 0.0:
                lists['track'].append(track_term(o, y, TRACK_IDX[color_mode]))
    means = {k: (torch.mean(torch.stack(v)) if v else 0.0) for k, v in lists.items()}     # loss.py:243-250
    total = (means['rgb'] * color_lw + means['dens'] * density_lw + means['segm'] * segmentation_lw
             + means['track'] * tracking_lw)                                              # loss.py:276-277
    return total, [means['rgb'], means['dens'], means['segm'], means['track']]
