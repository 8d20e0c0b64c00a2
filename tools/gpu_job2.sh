#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t2.log 2>&1; rc=$?
tail -15 gpurun_out/t2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out"; exit 1; fi
timeout -k 10 600 python tools/sweep.py --what mesh100k,mesh1m --count > gpurun_out/sweep2.log 2>&1; rc=$?
cat gpurun_out/sweep2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "sweep timed out"; exit 1; fi
mkdir -p gpurun_out/img
timeout -k 10 300 python tools/make_image.py gpurun_out/img/c1_cornell_256x256_b4_s16 W=256 H=256 bounces=4 spp=16 &&
timeout -k 10 300 python tools/make_image.py gpurun_out/img/cornell_256x256_b8_s1024 W=256 H=256 bounces=8 spp=1024 &&
timeout -k 10 300 python tools/make_image.py gpurun_out/img/mesh100k_320x180_b8_s256 scene=mesh100k W=320 H=180 bounces=8 spp=256
rm -f gpurun_out/img/*.pfm
