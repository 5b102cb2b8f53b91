"""Is the loss trajectory of the EAGER training step (BASELINE config 5) sensitive to an unrelated kernel between steps?
VARIANT=plain|absonly; OCC4D_DETERMINISTIC=0|1.  (Counterpart of graph_replay_probe.py: there the question was whether
the sensitivity belongs to the hipGraph replay.)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk
N_POINTS, FRAMES, QUERIES, SEED = 28672, 4, 17203, 1830
dev = torch.device('cuda:0')
pa, ia, inf = pk.configs.model_args('carla', N_POINTS)
esd, dsd = pk.configs.synthetic_weights(pa, ia, SEED)
enc = pk.model.PointCompletionNetV3(**pa).to(dev).train(); dec = pk.implicit.LocalPclResnetFC(**ia).to(dev).train()
enc.load_state_dict(esd); dec.load_state_dict(dsd)
pcl = pk.configs.synthetic_pcl('carla', N_POINTS, 12, SEED).to(dev)
rng = np.random.default_rng(SEED + 100)
q = np.concatenate([rng.uniform([0, -16, -1], [40, 16, 6.4], size=(FRAMES, QUERIES, 3)), np.broadcast_to(np.arange(FRAMES, dtype=np.float64)[:, None, None], (FRAMES, QUERIES, 1))], -1)
target = np.concatenate([rng.integers(0, 2, size=(FRAMES, QUERIES, 1)), rng.uniform(size=(FRAMES, QUERIES, 3)), np.zeros((FRAMES, QUERIES, 1)), rng.integers(-1, 13, size=(FRAMES, QUERIES, 1))], -1)
q = torch.from_numpy(q.astype(np.float32)).to(dev); target = torch.from_numpy(target.astype(np.float32)).to(dev)
torch.manual_seed(5)
step = pk.training.TrainStep(enc, dec, lr=1e-3, grad_clip=0.2, loss_kwargs=dict(density_lw=1.0, segmentation_lw=0.6, static_shapes=True))
step.batch_frames = True
V = os.environ.get('VARIANT', 'plain')
Z = torch.ones(1000, device=dev)
out = []
for it in range(8):
    out.append(round(float(step(pcl, q, target, next_pcl_input=pcl)), 5))
    if V == 'absonly':
        torch.cuda.synchronize(); zz = Z.abs()
print(V, 'deterministic' if pk.ops.DETERMINISTIC else 'atomics', out)
