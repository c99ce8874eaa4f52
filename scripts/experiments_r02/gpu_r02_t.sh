#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/perf_outliers.py 4096 50000 4096-8191 2>&1 | tail -9
