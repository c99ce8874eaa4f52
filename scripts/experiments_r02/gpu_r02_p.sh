#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python scripts/perf_predict.py 5000 2>&1 | tail -14
timeout 600 python -m pytest tests/test_pointnet2.py tests/test_nirrt_batch_gpu.py -m gpu -x -q 2>&1 | tail -4
for B in 1; do timeout 300 python scripts/perf_pointnet.py 2>&1 | head -3; done
