"""Compares the gradients of two DUMP2 files of graph_replay_probe.py: per tensor y ~ alpha x (alpha = <x,y>/<x,x>) and the
relative residual |y - alpha x| / |y|."""
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for it in sorted(a['grads']):
    print('gradients of replay', it + 1)
    rows = []
    for n, x, y in zip(a['names'], a['grads'][it], b['grads'][it]):
        x, y = x.double().flatten(), y.double().flatten()
        al = float((x @ y) / (x @ x).clamp(min=1e-300))
        res = float((y - al * x).norm() / y.norm().clamp(min=1e-300))
        rows.append((n, al, res, float(x.norm()), float(y.norm())))
    for n, al, res, nx, ny in rows:
        if 'attn_mlp.2.bias' in n:
            continue
        print('   %-46s alpha %8.5f  residual %.2e  |x| %.3e |y| %.3e' % (n, al, res, nx, ny))
