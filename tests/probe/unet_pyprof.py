"""cProfile of the UNet forward on a 4 096-point scene (host-side overhead hunt)"""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffusion_edf_amd import synthetic
from diffusion_edf_amd.gnn_data import FeaturedPoints
from diffusion_edf_amd.unet import UnetFeatureExtractor
dev = torch.device("cuda:0")
m = UnetFeatureExtractor(**synthetic.unet_kwargs("panda_lowres"), deterministic=True).to(dev)
x = torch.from_numpy(synthetic.make_scene(4096, seed=0).astype(np.float32)).to(dev)
pcd = FeaturedPoints(x=x, f=torch.rand(4096, 3, device=dev), b=torch.zeros(4096, dtype=torch.long, device=dev), w=None)
for _ in range(3):
    m(pcd)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m(pcd)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
