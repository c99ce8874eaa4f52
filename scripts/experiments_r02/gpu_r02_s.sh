#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "host_steer" 2>&1 | tail -40
