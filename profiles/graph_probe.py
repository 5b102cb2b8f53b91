import sys, time
sys.path.insert(0, '/root/repo')
import torch
import occlusions4d_amd as pk

def run(kind, n, vl, num_sample, batch):
    pa, ia, inf = pk.configs.model_args(kind, n)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, 1830)
    enc = pk.model.PointCompletionNetV3(**pa).cuda().eval(); enc.load_state_dict(esd)
    dec = pk.implicit.LocalPclResnetFC(**ia).cuda().eval(); dec.load_state_dict(dsd)
    pcl = pk.configs.synthetic_pcl(kind, n, vl, 1830).cuda()
    q = pk.geometry.sample_implicit_points_blind_device(num_sample, inf['min_z'], inf['cube_bounds'], 3, inf['data_kind'], 4, 'grid', 'cuda')
    f = lambda: pk.inference.infer_device(pcl, q, enc, dec, batch, inf['color_mode'], inf['predict_segmentation'], 'none', 13)
    with torch.no_grad():
        ref = f()['implicit_output'].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 10
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): f()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = f()
        g.replay(); torch.cuda.synchronize()
        print(kind, n, 'equal:', bool(torch.equal(out['implicit_output'], ref)), float((out['implicit_output'] - ref).abs().max()))
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 10
    print('%s n=%d q=%d: eager %.3f ms, graph %.3f ms' % (kind, n, q.shape[0], eager * 1e3, graph * 1e3), flush=True)

run('greater', 2048, 4, 8192, 4096)
run('greater', 14336, 12, 524288, 32768)
