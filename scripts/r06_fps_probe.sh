#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r06; mkdir -p $O
python -m pytest $R/tests/test_pointnet2.py $R/tests/test_guidance_device_gpu.py $R/tests/test_guidance_fixtures.py -m gpu -x -q 2>&1 | tail -15 | tee $O/fps_tests.txt
python $R/scripts/perf_fps64.py 16 128 256 512 2>&1 | grep -v amdgpu.ids | tee $O/fps64_perf.txt
