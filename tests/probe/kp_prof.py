"""torch.profiler view of one KeypointExtractor forward on the 1 024-point grasp cloud of C2 (bench.py's `keypoint_grasp`): device time against
wall time, and which host-side ops launch the most device work items.  python tests/probe/kp_prof.py [n_grasp]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.keypoint_extractor import KeypointExtractor
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kp = KeypointExtractor(**synthetic.keypoint_extractor_kwargs(bbox=None), deterministic=True).to(dev)
x = torch.from_numpy(np.ascontiguousarray(synthetic.make_grasp(n, seed=0), dtype=np.float32)).to(dev)
pcd = FeaturedPoints(x=x, f=torch.rand(n, 3, device=dev), b=torch.zeros(n, dtype=torch.long, device=dev), w=None)
for _ in range(3):
    kp(pcd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    kp(pcd)
torch.cuda.synchronize()
print(f"KeypointExtractor {n} pts: {(time.perf_counter() - t0) * 100:.2f} ms / forward")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    kp(pcd)
    torch.cuda.synchronize()
ka = prof.key_averages()
dev_us = sum(e.self_device_time_total for e in ka)
n_kernels = sum(e.count for e in ka if e.self_device_time_total > 0 and e.self_cpu_time_total == 0)
print(f"device time {dev_us / 1e3:.2f} ms in {n_kernels} device work items")
print(ka.table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=70))
