"""torch.profiler view of one UNet forward on the 16 384-point scene: which aten ops (host glue between the HIP kernels) launch device work"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.unet import UnetFeatureExtractor
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = sys.argv[2] if len(sys.argv) > 2 else "panda_lowres"
m = UnetFeatureExtractor(**synthetic.unet_kwargs(kind), deterministic=True).to(dev)
x = torch.from_numpy(synthetic.make_scene(n, seed=0).astype(np.float32)).to(dev)
pcd = FeaturedPoints(x=x, f=torch.rand(n, 3, device=dev), b=torch.zeros(n, dtype=torch.long, device=dev), w=None)
for _ in range(3):
    m(pcd)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m(pcd)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count" if len(sys.argv) > 3 else "cuda_time_total", row_limit=30, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=25, max_name_column_width=50, max_src_column_width=90))
