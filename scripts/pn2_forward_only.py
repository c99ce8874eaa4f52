"""PointNet++ forwards (N = 2048) for rocprofv3 / timing: python scripts/pn2_forward_only.py [B] [reps]
prints ms per forward and the fp32 GEMM rate it implies (2.76 GFLOP per cloud)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nirrt_star_amd import png_wrapper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ck = png_wrapper.checkpoint_path('/tmp/nirrt_ck', 2)
if not os.path.exists(ck):
    png_wrapper.make_synthetic_checkpoint(ck, device="cuda")
w = png_wrapper.PNGWrapper(root_dir='/tmp/nirrt_ck', device='cuda')
torch.manual_seed(0)
x = torch.rand(B, 6, 2048, device='cuda')
with torch.no_grad():
    for _ in range(3):
        w.model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        w.model(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("B=%d: %.3f ms/forward = %.3f ms/cloud, %.2f TFLOP/s of MLP GEMMs" % (B, dt * 1e3, dt * 1e3 / B, 2.76e9 * B / dt / 1e12))
