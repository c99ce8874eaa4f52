"""Down-sampling of guidance clouds: the register-resident kernel (k_fps_f64_reg, round 6) against the L2-streaming one (k_fps_f64,
NIRRT_FPS64_REG=0) - identical survivors, time per refresh of n clouds.      python scripts/perf_fps64.py [n_clouds ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nirrt_star_amd import pointops  # noqa: E402


def jobs_3d(n, words, n_raw, whole):
    t = pointops.cloud_job_table(n)
    t["words"] = [words.data_ptr() + 4 * words.shape[1] * k for k in range(n)]
    t["mode"] = 2 if whole else 3
    a = t["a"]
    if whole:
        a[:, 0:3] = 0.0
        a[:, 3:6] = [50.0, 40.0, 30.0]
    else:
        rng = np.random.RandomState(3)
        for k in range(n):
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            a[k, 0:9] = (q * np.array([22.0, 9.0, 9.0])).reshape(9)
        a[:, 9:12] = [25.0, 20.0, 15.0]
        a[:, 12:15] = 0.0
        a[:, 15:18] = [50.0, 40.0, 30.0]
    return t


def jobs_2d(n, words, n_raw, tab, w, h):
    t = pointops.cloud_job_table(n)
    t["words"] = [words.data_ptr() + 4 * words.shape[1] * k for k in range(n)]
    t["mode"] = 0
    t["free_tab"] = tab.data_ptr()
    t["w"], t["h"] = w, h
    t["a"][:, 0], t["a"][:, 1] = float(w), float(h)
    return t


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [16, 64, 256, 512]
    dev = torch.device("cuda")
    n_raw, n_points = 10240, 2048
    g = torch.Generator(device="cpu").manual_seed(5)
    tab = torch.ones((225, 225), dtype=torch.uint8, device=dev)
    for n in sizes:
        for label, dim in (("3D whole box", 3), ("3D ellipsoid", 3), ("2D whole image", 2)):
            nw = n_raw * 2 * (3 if dim == 3 else 2)
            words = torch.randint(0, 2 ** 31, (n, nw), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
            jobs = jobs_3d(n, words, n_raw, label == "3D whole box") if dim == 3 else jobs_2d(n, words, n_raw, tab, 224, 224)
            res = {}
            for reg in ("0", "1"):
                os.environ["NIRRT_FPS64_REG"] = reg
                out = torch.zeros((n, n_points, 3), dtype=torch.float64, device=dev)
                pointops.guidance_clouds(jobs, n_raw, n_points, out)      # warm
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    n_cand, n_out = pointops.guidance_clouds(jobs, n_raw, n_points, out)
                torch.cuda.synchronize()
                res[reg] = ((time.perf_counter() - t0) / reps, out.cpu().numpy(), n_cand.copy(), n_out.copy())
            same = np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][2], res["1"][2]) and np.array_equal(res["0"][3], res["1"][3])
            print("%4d clouds %-15s candidates %5d..%5d  streaming %7.2f ms  registers %7.2f ms  identical %s" % (
                n, label, res["0"][2].min(), res["0"][2].max(), res["0"][0] * 1e3, res["1"][0] * 1e3, same), flush=True)
            assert same


if __name__ == "__main__":
    main()
