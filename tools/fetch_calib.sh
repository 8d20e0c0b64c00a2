#!/bin/bash
# FETCH_SIZE calibration for 64-byte per-lane gathers (tools/micro/fetch_calib.hip): one rocprofv3 --pmc pass per counter set, then
# per kernel the counters of its LAST launch against the bytes it is known to have asked for.  Output: gpurun_out/<tag>/fetch_calib.txt
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
tag=${1:-calib}; out=gpurun_out/$tag; mkdir -p $out
rocprofv3 -L > $out/counters_available.txt 2>&1
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/pass_$i -o p --output-format csv -- ./tools/micro/fetch_calib > $out/pass_$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 tools/fetch_calib_summary.py $out > $out/fetch_calib.txt 2>&1
cat $out/fetch_calib.txt
