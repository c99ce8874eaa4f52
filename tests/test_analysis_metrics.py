"""CPU: the evaluation metrics of nirrt_star_amd.analysis vs the reference's own analysis script run on the same synthetic
result pickles (tests/golden/analysis_ref.json, written by make_golden.py from /root/reference)."""
import json
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

KEYS = {'rrt_star-none': 'rrt', 'irrt_star-none': 'irrt', 'nrrt_star-pointnet2': 'nrrt_png', 'nrrt_star-unet': 'nrrt_gng',
        'nrrt_star-c-bfs-pointnet2': 'nrrt_png_c', 'nirrt_star-pointnet2': 'nirrt_png', 'nirrt_star-c-bfs-pointnet2': 'nirrt_png_c'}


def test_metrics_match_the_reference_script(tmp_path):
    import analysis_inputs
    from nirrt_star_amd import analysis
    with open(os.path.join(HERE, "golden", "analysis_ref.json")) as f:
        exp = json.load(f)
    data = analysis_inputs.make_inputs()
    folder = tmp_path / "results" / "evaluation" / "2d"
    folder.mkdir(parents=True)
    for stem, lst in data.items():
        with open(folder / ("random_2d-%s-%d.pickle" % (stem, exp["n"])), "wb") as f:
            pickle.dump(lst, f)
    res = analysis.load_results(str(folder), "random_2d", exp["n"])
    assert sorted(res) == sorted(KEYS.values())
    out = analysis.analyse(res, exp["n"])
    for m in KEYS.values():
        assert out[m]["first_solution_iterations"] == exp["first_solution"][m]
        assert np.array_equal(np.array(out[m]["path_cost_mean"]), np.array(exp["path_cost_mean"][m]))   # same float64 ops


def test_short_lists_use_their_last_entry():
    from nirrt_star_amd import analysis
    r = [{"result": [np.inf, 10.0, 9.0]}]
    rrt = [{"result": [np.inf, np.inf, 20.0]}]
    ratios = analysis.path_cost_ratios(r, rrt, [0, 1, 250])
    assert ratios == {0: [0.5], 1: [0.45], 250: [0.45]}
