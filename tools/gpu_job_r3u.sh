#!/bin/bash
# round 3, step u: the device SAH builder -- same-tree check, builders side by side, phases, kernel times, OBJ load with it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 5 200 python3 tools/sahdev_check.py > $O/r3u_same_tree.txt 2>&1 &&
timeout -k 5 300 python3 tools/bvh_build_bench.py mesh100k mesh1M > $O/r3u_builders.txt 2>&1 &&
PTAMD_TRACE=1 timeout -k 5 100 python3 tools/sahdev_profile.py 1000000 256 > $O/r3u_phases.txt 2>&1 &&
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/r3u_prof -o p -- python3 tools/sahdev_profile.py 1000000 256 > $O/r3u_prof.log 2>&1 &&
python3 tools/rocpd_kernels.py $O/r3u_prof/p_results.db k_stage_area > $O/r3u_kernels.txt 2>&1 &&
PTAMD_TRACE=1 timeout -k 5 200 python3 tools/obj_load_time.py 1000000 > $O/r3u_obj_load.txt 2>&1
echo "rc $?"
tail -3 $O/r3u_same_tree.txt; cat $O/r3u_builders.txt; tail -25 $O/r3u_phases.txt; cat $O/r3u_kernels.txt; tail -12 $O/r3u_obj_load.txt
