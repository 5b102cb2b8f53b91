"""Round-4 probe of GraphedTrainStep at BASELINE config 5 size: what happens to the loss trajectory of the replayed step
when host code does something between two replays.  VARIANT=plain|sync|twinnets|emptyonly|itemonly|poison|poison2 (RANGE=lo,hi)
|absonly|maxonly|readone|readonly|manysync|compare|compare_nograd.  Results: profiles/r04_graph_replay_probe.txt."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk
N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 1830
dev = torch.device('cuda:0')
pa, ia, inf = pk.configs.model_args('carla', N_POINTS)
esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
def nets():
    enc = pk.model.PointCompletionNetV3(**pa).to(dev).train(); dec = pk.implicit.LocalPclResnetFC(**ia).to(dev).train()
    enc.load_state_dict(esd); dec.load_state_dict(dsd)
    return enc, dec
pcl = pk.configs.synthetic_pcl('carla', N_POINTS, 12, SEED).to(dev)
rng = np.random.default_rng(SEED + 100)
q = np.concatenate([rng.uniform([0, -16, -1], [40, 16, 6.4], size=(FRAMES, QUERIES, 3)), np.broadcast_to(np.arange(FRAMES, dtype=np.float64)[:, None, None], (FRAMES, QUERIES, 1))], -1)
target = np.concatenate([rng.integers(0, 2, size=(FRAMES, QUERIES, 1)), rng.uniform(size=(FRAMES, QUERIES, 3)), np.zeros((FRAMES, QUERIES, 1)), rng.integers(-1, 13, size=(FRAMES, QUERIES, 1))], -1)
q = torch.from_numpy(q.astype(np.float32)).to(dev); target = torch.from_numpy(target.astype(np.float32)).to(dev)
lkw = dict(density_lw=1.0, segmentation_lw=0.6)
V = os.environ.get('VARIANT', 'plain')
e1, d1 = nets()
if V in ('twinnets', 'compare', 'compare_nograd'):
    e2, d2 = nets()
g = pk.training.GraphedTrainStep(e1, d1, lr=1e-3, grad_clip=0.2, loss_kwargs=lkw)
g.capture(pcl, q, target)
Z = torch.ones(1000, device=dev)
out = []
for it in range(6):
    out.append(round(float(g(pcl, q, target, next_pcl_input=pcl)), 4))
    if V == 'sync':
        torch.cuda.synchronize()
    if V == 'poison':
        torch.cuda.synchronize()
        junk = [torch.full((n,), float('nan'), device=dev) for n in (1 << 28, 1 << 24, 1 << 20, 1 << 16, 1 << 12, 1 << 8) for _ in range(3)]
        torch.cuda.synchronize()
        del junk
    if V == 'compare_nograd':
        torch.cuda.synchronize()
        with torch.no_grad():
            s = sum(float((a - b).abs().max()) for a, b in zip(g.params, list(e2.parameters()) + list(d2.parameters())))
    if V == 'manysync':
        torch.cuda.synchronize()
        z = torch.ones(1000, device=dev)
        s = sum(float(z.abs().max()) for _ in range(150))
    if V == 'readgrad':
        torch.cuda.synchronize()
        with torch.no_grad():
            s = sum(float(a.grad.abs().max()) for a in g.params if a.grad is not None)
    if V == 'readone':
        torch.cuda.synchronize()
        with torch.no_grad():
            s = float(g.params[0].abs().max())
    if V == 'absonly':
        torch.cuda.synchronize(); zz = Z.abs()
    if V == 'itemonly':
        torch.cuda.synchronize(); s = float(Z[0])
    if V == 'maxonly':
        torch.cuda.synchronize(); zz = Z.max()
    if V == 'emptyonly':
        torch.cuda.synchronize(); zz = torch.empty(1000, device=dev)
    if V.startswith('poison2'):
        torch.cuda.synchronize()
        lo, hi = [int(v) for v in os.environ.get('RANGE', '512,65536').split(',')]
        junk = []
        for nbytes in range(hi, lo - 1, -512):
            for _ in range(3):
                junk.append(torch.full((nbytes // 4,), float('nan'), device=dev))
        torch.cuda.synchronize()
        del junk
    if V == 'readonly':
        torch.cuda.synchronize()
        with torch.no_grad():
            s = sum(float(a.abs().max()) for a in g.params)
    if V == 'compare':
        torch.cuda.synchronize()
        s = sum(float((a - b).abs().max()) for a, b in zip(g.params, list(e2.parameters()) + list(d2.parameters())))
print(V, out)
