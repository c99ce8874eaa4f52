import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {k: z[k] for k in z.files}
    for k in list(out):
        if k == "env" or k.endswith("_env"):
            out[k] = json.loads(str(out[k]))
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def make_oracle_tree(orc, g, iter_max=None):
    """OracleTree for a run_* / random_* fixture dict."""
    return orc.OracleTree(int(g["dim"]), int(iter_max if iter_max is not None else g["iter_max"]),
                          g["x_start"], g["x_goal"], float(g["step_len"]), float(g["search_radius"]),
                          float(g["clearance"]), g["env"])
