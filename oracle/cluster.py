"""Restated semantics of ``torch_cluster.fps`` / ``torch_cluster.knn`` (CPU, torch).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
calls these at model/modules.py:133-134 (fps) and :142-143 (knn); the library
(rusty1s/pytorch_cluster, version unpinned) is not vendored under
/root/reference and is not installed here, so what follows restates its
documented behaviour:

* fps(src, batch, ratio, random_start): per batch element, start at local
  index 0 (random_start=False) or at ``torch.randint(n, (1,))`` from torch's global
  CPU generator (random_start=True; the library's own draw is not reproducible
  offline), repeatedly take the argmax (first maximal index) of the running
  minimum of squared Euclidean distances to the already-selected set;
  ``ceil(ratio * n)`` samples; indices are global (flattened) and in selection
  order (the reference sorts them afterwards, model/modules.py:135).
* knn(x, y, k, batch_x, batch_y): for each row of ``y`` the ``k`` nearest rows
  of ``x`` (same batch element) by squared Euclidean distance, returned as a
  (2, k*|y|) index tensor [y index; x index], grouped by y, nearest first.
"""

import torch


def _sqdist_to(p, i):
    # (p - p[i])**2 summed over xyz in coordinate order, plain fp32 (no FMA).
    d = p - p[i]
    d = d * d
    return (d[:, 0] + d[:, 1]) + d[:, 2]


def fps(src, batch=None, ratio=0.5, random_start=True):
    src = src.detach()
    n_total = src.shape[0]
    if batch is None:
        batch = torch.zeros(n_total, dtype=torch.long)
    out = []
    for b in range(int(batch.max().item()) + 1 if n_total else 0):
        sel = (batch == b).nonzero()[:, 0]
        start = int(sel[0])
        p = src[sel]
        n = p.shape[0]
        # sample count as the library computes it: float32(n) * float32(ratio), ceil
        m = int(torch.ceil(torch.tensor(float(n), dtype=torch.float32)
                           * torch.tensor(ratio, dtype=torch.float32)).item())
        first = int(torch.randint(n, (1,)).item()) if random_start else 0
        chosen = torch.empty(m, dtype=torch.long)
        chosen[0] = first
        mind = _sqdist_to(p, first)
        for s in range(1, m):
            nxt = int(torch.argmax(mind))
            chosen[s] = nxt
            mind = torch.minimum(mind, _sqdist_to(p, nxt))
        out.append(chosen + start)
    return torch.cat(out) if out else torch.empty(0, dtype=torch.long)


def knn(x, y, k, batch_x=None, batch_y=None):
    x = x.detach()
    y = y.detach()
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    rows, cols = [], []
    for b in range(int(batch_y.max().item()) + 1 if y.shape[0] else 0):
        xi = (batch_x == b).nonzero()[:, 0]
        yi = (batch_y == b).nonzero()[:, 0]
        xb, yb = x[xi], y[yi]
        for lo in range(0, yb.shape[0], 1024):
            d = yb[lo:lo + 1024, None, :] - xb[None, :, :]
            d = d * d
            d = (d[..., 0] + d[..., 1]) + d[..., 2]
            # stable sort => lowest index wins among equal distances
            nn = torch.sort(d, dim=1, stable=True)[1][:, :k]
            rows.append(yi[lo:lo + 1024, None].expand(-1, k).reshape(-1))
            cols.append(xi[nn].reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)])
