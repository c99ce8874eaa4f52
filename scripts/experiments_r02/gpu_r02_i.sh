#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_hip_edges.py tests/test_hip_parity.py tests/test_hip_sampling.py tests/test_hip_grid.py -m gpu -x -q 2>&1 | tail -3
bash scripts/gpu_sweep.sh i
