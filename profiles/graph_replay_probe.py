"""Round-4 probe of GraphedTrainStep at BASELINE config 5 size: what happens to the loss trajectory of the replayed step
when host code does something between two replays.  VARIANT=plain|sync|twinnets|emptyonly|itemonly|poison|poison2 (RANGE=lo,hi)
|absonly|maxonly|readone|readonly|manysync|compare|compare_nograd; PRELOAD=1 runs abs / max once before the capture.  Results: profiles/r04_graph_replay_probe.txt."""
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
if os.environ.get('PRELOAD') == '1':      # the kernels the variants launch, used once BEFORE the capture (their code object is loaded)
    z0 = torch.ones(1000, device=dev)
    s0 = float(z0.abs().max())
    torch.cuda.synchronize()
g = pk.training.GraphedTrainStep(e1, d1, lr=1e-3, grad_clip=0.2, loss_kwargs=lkw)
if os.environ.get('KERNEL_COPIES') == '1':     # stage the geometry into the static buffers with kernels instead of copy_ (memcpy)
    def _load_geometry(pcl_input, self=g):
        key = self.pcl_net.geometry_key(pcl_input)
        nxt, self._geom_next = self._geom_next, None
        geom = nxt[1] if (nxt is not None and nxt[2] is pcl_input and nxt[0] == key) else \
            self.pcl_net._geometry_chain(pcl_input[..., :3].detach(), full=True)
        self.pcl_net._prefetched = None
        cur = torch.cuda.current_stream()
        for i, entry in geom.items():
            cur.wait_event(entry[2])
            for dst, src in zip(self._geom_tensors(self._geom_static[i]), self._geom_tensors(entry)):
                torch.add(src, 0, out=dst)
    g._load_geometry = _load_geometry
g.capture(pcl, q, target)
if os.environ.get('SYNC_BEFORE_REPLAY'):       # host waits for the eager staging work of a call before it launches the graph
    mode = os.environ['SYNC_BEFORE_REPLAY']
    real_replay = g.graph.replay
    def replay():
        if os.environ.get('VERBOSE'):
            import zlib
            torch.cuda.synchronize()
            sums = []
            for i in sorted(g._geom_static):
                for t in g._geom_tensors(g._geom_static[i]):
                    sums.append(zlib.crc32(t.cpu().numpy().tobytes()) & 0xffff)
            print('   static geometry before this replay (crc16 per tensor):', sums)
        if mode == 'stream':
            torch.cuda.current_stream().synchronize()
        elif mode == 'device':
            torch.cuda.synchronize()
        elif mode == 'event':
            e = torch.cuda.Event(); e.record(); torch.cuda.current_stream().wait_event(e)
        real_replay()
    g.graph.replay = replay
Z = torch.ones(1000, device=dev)
out = []
if V == 'nosync':        # replays issued without waiting for the previous one: losses leave through async copies into pinned memory
    pinned = torch.empty(6, pin_memory=True)
    for it in range(6):
        pinned[it:it + 1].copy_(g(pcl, q, target, next_pcl_input=pcl).reshape(1), non_blocking=True)
    torch.cuda.synchronize()
    print(V, [round(float(v), 4) for v in pinned])
    sys.exit(0)
def check_static_geometry():
    """Is the geometry in the static buffers (what the NEXT replay... no: what the LAST replay read) self-consistent?
    sub-cloud = cloud[inds]; pooling lists = kNN(sub-cloud -> cloud); self lists = kNN(cloud -> cloud)."""
    torch.cuda.synchronize()
    bad = []
    cloud = g.static[0][0, :, :3].contiguous()
    for i, block in enumerate(e1.blocks):
        entry = g._geom_static[i][0]
        if isinstance(block, pk.modules.DownTransition):
            inds, p_sub, nn = entry[0]
            if not torch.equal(p_sub, cloud[inds.long()]):
                bad.append('block %d: p_sub != cloud[inds] (%d rows differ)' % (i, int((p_sub != cloud[inds.long()]).any(dim=1).sum())))
            ref = pk.ops.knn(p_sub, cloud, block.knn_k, metric=0)
            if not torch.equal(nn, ref):
                bad.append('block %d: pooling lists differ in %d rows' % (i, int((nn != ref).any(dim=1).sum())))
            if not bool((inds[1:] > inds[:-1]).all()):
                bad.append('block %d: inds not ascending' % i)
            cloud = p_sub
        else:
            ref = pk.ops.knn(cloud, cloud, block.num_neighbors, metric=0)
            if not torch.equal(entry[0], ref):
                bad.append('block %d: self lists differ in %d rows' % (i, int((entry[0] != ref).any(dim=1).sum())))
    torch.cuda.synchronize()
    return bad


DUMP = os.environ.get('DUMP')
DUMP2 = os.environ.get('DUMP2')       # gradients of replay 2 and 3 through asynchronous copies into pinned memory (no kernel, no sync)
pin = {}
names = [n for n, _ in list(e1.named_parameters()) + list(d1.named_parameters())]
dump = []
for it in range(6):
    out.append(round(float(g(pcl, q, target, next_pcl_input=pcl)), 4))
    if os.environ.get('ADDR') == '1':      # where the allocator put the geometry prefetched for the next step
        import zlib
        ptrs = [t.data_ptr() for i in sorted(g._geom_next[1]) for t in g._geom_tensors(g._geom_next[1][i])]
        print('   after call', it + 1, 'prefetched geometry:', len(ptrs), 'tensors, address checksum', zlib.crc32(str(ptrs).encode()) & 0xffff,
              'first', hex(ptrs[0]), 'reserved MB', torch.cuda.memory_reserved() >> 20)
    if DUMP2 and it in (1, 2):
        pin[it] = [torch.empty(p.grad.shape, pin_memory=True).copy_(p.grad, non_blocking=True) for p in g.params]
    if DUMP:          # (D2H copies only: no kernel) parameter and gradient snapshots after this replay
        torch.cuda.synchronize()
        dump.append(([p.detach().cpu().clone() for p in g.params], [None if p.grad is None else p.grad.detach().cpu().clone() for p in g.params]))
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
    if V == 'absonly_check':
        print('   static geometry after replay', it + 1, check_static_geometry() or 'consistent')
        zz = Z.abs()
    if V == 'plain_check':
        print('   static geometry after replay', it + 1, check_static_geometry() or 'consistent')
    if V == 'abs_then_sync':
        torch.cuda.synchronize(); zz = Z.abs(); torch.cuda.synchronize()
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
if DUMP:
    torch.save({'names': names, 'dump': dump}, DUMP)
if DUMP2:
    torch.cuda.synchronize()
    torch.save({'names': names, 'grads': {k: [t.clone() for t in v] for k, v in pin.items()}}, DUMP2)
