import sys; sys.path.insert(0,'.')
import numpy as np
from nirrt_star_amd import _hip, worlds
from oracle import oracle as orc
pr = worlds.problem_2d(worlds.random_world_2d(0, "b30"), 0)
iters = 400
rng = np.random.default_rng(0)
samples = rng.uniform(3, 221, size=(iters, 2))
t = _hip.HipTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env"], device_id=0)
o = orc.OracleTree(2, iters, pr["x_start"], pr["x_goal"], 10, pr["search_radius"], 3, pr["env_dict"])
for k,q in enumerate(samples):
    r = t.step(q, _hip.F_IRRT)
    ro = o.step(q, True)
    if not (r.nearest_idx == ro.nearest_idx and r.n == ro.n and r.n_near == ro.n_near and r.n_rewired==ro.n_rewired and r.collided==ro.collided):
        print("iter",k,"q",q)
        for f in ("collided","inserted","nearest_idx","new_idx","n_near","reparented","n_rewired","in_goal","n"):
            print(" ",f,getattr(r,f),getattr(ro,f))
        print(" node_new", list(r.node_new), list(ro.node_new))
        break
