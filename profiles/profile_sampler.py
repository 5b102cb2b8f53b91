"""Where the guided point sampler's time goes (one config-5 frame set, host profile + GPU time)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

FRAMES, QUERIES = 4, 17203
frames, sizes, valo, num_valo = pk.configs.synthetic_target_frames('carla', 57344, FRAMES, 2030)
frames = [f.cuda() for f in frames]
sizes = [z.cuda() for z in sizes]
valo, num_valo = valo.cuda(), num_valo.cuda()
sampler = pk.geometry.GuidedImplicitPointSampler(
    None, min_z=-1.0, cube_bounds=16.0, point_occupancy_radius=0.2, num_solid=7168, num_air=QUERIES - 7168,
    predict_segmentation=True, semantic_classes=13, data_kind='carla', point_sample_bias='low_moving_vehped_sembal', cube_mode=4)
np.random.seed(1)
torch.manual_seed(1)


def run():
    for t in range(FRAMES):
        sampler(frames, sizes, valo, num_valo, t)
    torch.cuda.synchronize()


for _ in range(3):
    run()
t0 = time.perf_counter()
for _ in range(5):
    run()
print('sampler, %d frames: %.2f ms per step' % (FRAMES, 1e3 * (time.perf_counter() - t0) / 5))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print('GPU-side elapsed (events around one run): %.2f ms' % e0.elapsed_time(e1))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    run()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
