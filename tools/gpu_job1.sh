#!/bin/bash
# tests, then the sweep with the default library and with the A/B library (round-1 slab test on the global path)
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t1.log 2>&1; rc=$?
tail -5 gpurun_out/t1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit 1; fi
timeout -k 10 600 python tools/sweep.py --count > gpurun_out/sweep1.log 2>&1; rc=$?
cat gpurun_out/sweep1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "sweep timed out"; exit 1; fi
PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_oldslab.so timeout -k 10 600 python tools/sweep.py --what cb,mesh100k,mesh1m > gpurun_out/sweep1_oldslab.log 2>&1
cat gpurun_out/sweep1_oldslab.log
