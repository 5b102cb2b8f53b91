"""Compares two DUMP files of graph_replay_probe.py: per replay, the largest gradient / parameter difference relative to
the tensor's largest entry, and the parameters that moved differently by more than half a learning-rate step."""
import sys
import torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
names = a['names']
for it, ((pa, ga), (pb, gb)) in enumerate(zip(a['dump'], b['dump'])):
    rg = max(float((x - y).abs().max() / x.abs().max().clamp(min=1e-30)) for x, y in zip(ga, gb) if x is not None)
    dp = [(float((x - y).abs().max()), n) for n, x, y in zip(names, pa, pb)]
    big = sorted(d for d in dp if d[0] > 5e-4)
    flips = sum(int(((x - y).abs() > 5e-4).sum()) for x, y in zip(pa, pb))
    total = sum(x.numel() for x in pa)
    print('after replay %d: max relative gradient difference %.2e; largest parameter difference %.2e; %d of %d parameter '
          'entries differ by more than 5e-4 (lr = 1e-3), in %d tensors' % (it + 1, rg, max(d[0] for d in dp), flips, total, len(big)))
    for d, n in big[-6:]:
        print('      %-60s %.2e' % (n, d))
