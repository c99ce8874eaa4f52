import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from nirrt_star_amd import png_wrapper
from nirrt_star_amd.pointnet2 import pc_normalize
ck = png_wrapper.checkpoint_path('/tmp/nirrt_ck', 2)
if not os.path.exists(ck):
    png_wrapper.make_synthetic_checkpoint(ck)
w = png_wrapper.PNGWrapper(root_dir='/tmp/nirrt_ck', device='cuda')
rs = np.random.RandomState(0)
pc = rs.uniform(0, 224, size=(2048, 2)).astype(np.float32)
s = (np.linalg.norm(pc - pc[0], axis=1) < 10).astype(np.float32)
g = (np.linalg.norm(pc - pc[1], axis=1) < 10).astype(np.float32)
def T(name, fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print("%-40s %.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3)); return r
pc3 = np.concatenate((pc, np.zeros((2048, 1), np.float32)), axis=1)
T("pc_normalize", lambda: pc_normalize(pc3))
xyz = T("from_numpy.to(cuda)", lambda: torch.from_numpy(pc_normalize(pc3)).to('cuda'))
free = 1 - (s + g).astype(bool)
feat = T("features to cuda", lambda: torch.from_numpy(np.stack((s, g, free.astype(np.float32)), axis=-1)).to('cuda'))
x = torch.cat([xyz, feat], dim=1).permute(1, 0).unsqueeze(0).float()
with torch.no_grad():
    out = T("model(x)", lambda: w.model(x))
    T("seg.to(cpu)", lambda: out[0].detach().to('cpu'))
    seg = out[0].detach().to('cpu')
    T("argmax+softmax cpu", lambda: (np.argmax(seg.numpy(), 2)[0], torch.softmax(seg, dim=-1)[0, :, 1].numpy()))
T("classify_path_points", lambda: w.classify_path_points(pc, s, g))
torch.set_num_threads(1)
T("classify_path_points (1 torch thread)", lambda: w.classify_path_points(pc, s, g))
