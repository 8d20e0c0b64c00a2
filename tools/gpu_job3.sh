#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "schedules or config1 or chained or node_paths or split_api or ragged" > gpurun_out/t3.log 2>&1; rc=$?
tail -15 gpurun_out/t3.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit 1; fi
timeout -k 10 900 python tools/sweep_suspend.py > gpurun_out/sweep3.log 2>&1
cat gpurun_out/sweep3.log
