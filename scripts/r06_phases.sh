#!/bin/bash
# per-phase device clock of the loop with the -DNIRRT_PROFILE build (scripts/build_variant.sh prof -DNIRRT_PROFILE): IRRT* and RRT* 2D
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/r06
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so python $R/scripts/perf_phases.py 2>&1 | tee $R/gpurun_out/r06/phases_irrt2d.txt
NIRRT_HIP_SO=$R/nirrt_star_amd/libnirrt_hip_prof.so python $R/scripts/perf_phases.py --algo rrt --world b30 2>&1 | tee $R/gpurun_out/r06/phases_rrt2d.txt
