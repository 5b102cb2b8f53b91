"""Where the perform_inference call (host numpy in / out) spends its time, call by call (VERDICT r5 weak 2:
125.6 ms in round 4 -> 144.4 ms in round 5 on the driver's line).

    python profiles/host_boundary_probe.py [calls]

Prints per call: wall ms, pinned host bytes newly reserved by torch's caching host allocator, device allocator
reserved-bytes delta, the number of decoder weight / scene preparations (ops.decoder_prepare / decoder_prepare_scene
call counts), and a phase split of one call (H2D + grid, encode, decode, split, D2H wait).  First as bench.py called
it in round 5 (after the precision switches were flipped forth and back), then from a fresh state with `res` dropped
between calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import occlusions4d_amd as pk  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    kind = 'greater'
    dev = torch.device('cuda', 0)
    pa, ia, inf = pk.configs.model_args(kind, 14336)
    esd, dsd = pk.configs.synthetic_weights(pa, ia, 1830)
    pcl_cpu = pk.configs.synthetic_pcl(kind, 14336, 12, 1830)
    enc = pk.model.PointCompletionNetV3(**pa).to(dev).eval()
    dec = pk.implicit.LocalPclResnetFC(**ia).to(dev).eval()
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)

    counts = dict(prepare=0, scene=0)
    real_prepare, real_scene = pk.ops.decoder_prepare, pk.ops.decoder_prepare_scene

    def cp(*a, **k):
        counts['prepare'] += 1
        return real_prepare(*a, **k)

    def cs(*a, **k):
        counts['scene'] += 1
        return real_scene(*a, **k)
    pk.ops.decoder_prepare, pk.ops.decoder_prepare_scene = cp, cs

    def call():
        return pk.inference.perform_inference(
            pcl_cpu.clone(), None, None, [enc, dec], dev, 'if', inf['min_z'], inf['cube_bounds'], inf['color_mode'], 3,
            None, sample_implicit=True, num_sample=524288, point_sample_mode='grid', batch_size=32768,
            predict_segmentation=inf['predict_segmentation'], track_mode='none', semantic_classes=13,
            density_threshold=0.5, data_kind=inf['data_kind'], cube_mode=inf['cube_mode'], compress_air=True)

    def host_reserved():
        try:
            st = torch.cuda.host_memory_stats()
            return st.get('reserved_bytes.current', st.get('allocated_bytes.current', -1))
        except Exception:
            return -1

    def run(label, n, keep):
        print('--', label)
        res = None
        for i in range(n):
            c0 = dict(counts)
            h0, d0 = host_reserved(), torch.cuda.memory_reserved()
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = call()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t)
            if keep:
                res = r
            else:
                del r
            print('call %d: %7.2f ms  pinned +%.1f MB  device +%.1f MB  prepare %d  scene %d'
                  % (i, ms, (host_reserved() - h0) / 1e6, (torch.cuda.memory_reserved() - d0) / 1e6,
                     counts['prepare'] - c0['prepare'], counts['scene'] - c0['scene']), flush=True)
        return res

    with torch.no_grad():
        # device-resident step first, as bench.py has run many of them before the host-boundary leg
        pcl = pcl_cpu.to(dev)
        q = pk.geometry.sample_implicit_points_blind_device(524288, inf['min_z'], inf['cube_bounds'], 3, inf['data_kind'],
                                                            inf['cube_mode'], 'grid', dev)
        for _ in range(3):
            pk.distributed.sharded_inference(pcl, q, enc, dec, 32768, inf['color_mode'], inf['predict_segmentation'],
                                             'none', 13)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            pk.distributed.sharded_inference(pcl, q, enc, dec, 32768, inf['color_mode'], inf['predict_segmentation'],
                                             'none', 13)
        torch.cuda.synchronize()
        print('device-resident step: %.2f ms' % (1e3 * (time.perf_counter() - t) / 5))
        run('as bench.py holds `res` across calls (round-5 leg)', calls, keep=True)
        run('results dropped between calls', calls, keep=False)


if __name__ == '__main__':
    main()
