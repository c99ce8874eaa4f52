"""20 PointNet++ forwards (N = 2048, B = 1) for rocprofv3: python scripts/pn2_forward_only.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nirrt_star_amd import png_wrapper
ck = png_wrapper.checkpoint_path('/tmp/nirrt_ck', 2)
if not os.path.exists(ck):
    png_wrapper.make_synthetic_checkpoint(ck)
w = png_wrapper.PNGWrapper(root_dir='/tmp/nirrt_ck', device='cuda')
x = torch.rand(1, 6, 2048, device='cuda')
with torch.no_grad():
    for _ in range(20):
        w.model(x)
torch.cuda.synchronize()
