"""Times the fused set-abstraction kernel on the four PointNet++ MSG branches it serves (SA1, SA2; B clouds of 2048 points):
   python scripts/sa_mlp_bench.py [B]  -> ms per launch, TFLOP/s of the branch's real (unpadded) MLP flops, % of 157.3 TFLOP/s"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from nirrt_star_amd import pointops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BR = [("sa1.0", 2048, 1024, 16, 6, [16, 16, 32]), ("sa1.1", 2048, 1024, 32, 6, [32, 32, 64]),
      ("sa2.0", 1024, 256, 16, 96, [64, 64, 128]), ("sa2.1", 1024, 256, 32, 96, [64, 96, 128])]
torch.manual_seed(0)
for name, N, S, K, C, widths in BR:
    feats = torch.randn(B, N, C, device="cuda")
    xyz = torch.rand(B, N, 3, device="cuda")
    new_xyz = xyz[:, :S].contiguous()
    gidx = torch.randint(0, N, (B, S, K), device="cuda")
    layers, c = [], C + 3
    for w in widths:
        layers.append((torch.randn(w, c, device="cuda") * 0.1, torch.randn(w, device="cuda") * 0.1))
        c = w
    pack = pointops.sa_mlp_pack(layers, C + 3, torch.device("cuda"))
    out = torch.empty(B, S, widths[-1], device="cuda")
    for _ in range(3):
        assert pointops.sa_mlp(feats, xyz, new_xyz, gidx, pack, out, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        pointops.sa_mlp(feats, xyz, new_xyz, gidx, pack, out, 0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl, c = 0, C + 3
    for w in widths:
        fl += 2 * c * w
        c = w
    fl *= B * S * K
    print("%s B=%d: %.3f ms  %.1f TFLOP/s  %.1f%% of fp32 MFMA peak" % (name, B, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100))
