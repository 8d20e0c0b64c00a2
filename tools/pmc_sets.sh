#!/bin/bash
# usage: tools/pmc_sets.sh <tag> "<set 1>;<set 2>;..." <prof_render args...>   -- one rocprofv3 --pmc pass per counter set
# (the general form of pmc_passes.sh; never together with the runtime traces).  Output: gpurun_out/pmc_<tag>_<i>/ and
# gpurun_out/pmc_<tag>_summary.txt (mean per kernel and counter + kernel durations)
set -e
tag=$1; sets=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$i -o p --output-format csv -- python3 tools/prof_render.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  tail -1 gpurun_out/pmc_${tag}_$i.log
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_[0-9]* > gpurun_out/pmc_${tag}_summary.txt
python3 tools/trace_summary.py gpurun_out/pmc_${tag}_1 >> gpurun_out/pmc_${tag}_summary.txt
cat gpurun_out/pmc_${tag}_summary.txt
