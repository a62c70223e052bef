"""K11 (nerfail_gauss_bwd_views) alone on the bench's 8-view batch: ms per call by kernel-level HIP events."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import bench
dev = torch.device('cuda:0')
wi, ori, s_init = bench._attack_inputs(dev, 8, seed=0)
G = torch.randn((8, bench.H, bench.W, 4), device=dev)
r = bench.gauss_kernel_rooflines(dev, wi, ori, s_init, G)
print({k: round(v['ms_per_call'], 4) for k, v in r.items() if isinstance(v, dict)})
