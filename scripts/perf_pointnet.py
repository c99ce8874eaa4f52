"""PointNet++ forward latency on the GPU: python scripts/perf_pointnet.py"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from nirrt_star_amd import png_wrapper
ck = png_wrapper.checkpoint_path('/tmp/nirrt_ck', 2)
if not os.path.exists(ck):
    png_wrapper.make_synthetic_checkpoint(ck)
w = png_wrapper.PNGWrapper(root_dir='/tmp/nirrt_ck', device='cuda')
rs = np.random.RandomState(0)
pc = rs.uniform(0, 224, size=(2048, 2)).astype(np.float32)
s = (np.linalg.norm(pc - pc[0], axis=1) < 10).astype(np.float32)
g = (np.linalg.norm(pc - pc[1], axis=1) < 10).astype(np.float32)
for _ in range(3):
    w.classify_path_points(pc, s, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    w.classify_path_points(pc, s, g)
torch.cuda.synchronize()
print("classify_path_points: %.2f ms/call" % ((time.perf_counter() - t0) / 20 * 1e3))
x = torch.rand(1, 6, 2048, device='cuda')
with torch.no_grad():
    for _ in range(3):
        w.model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w.model(x)
    torch.cuda.synchronize()
    print("model forward only: %.2f ms/call (2.76 GFLOP -> %.2f TFLOP/s)" % ((time.perf_counter() - t0) / 20 * 1e3, 2.76e9 / ((time.perf_counter() - t0) / 20) / 1e12))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        w.model(x)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
