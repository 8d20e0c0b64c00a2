#!/bin/bash
# usage: tools/gpu_job.sh <tag> [tests] [ab "<names>"] [rates]   -- the round's standing measurement job, pieces by keyword:
#   tests  the GPU test suite (-m gpu), log under gpurun_out/<tag>/tests.log
#   abtests <lib> <-k expr>   GPU tests matching the expression against an A/B library
#   ab     mesh / Cornell rates with the default library and with every A/B library named (make ab AB=...)
# Everything is written under gpurun_out/<tag>/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
while [ $# -gt 0 ]; do
  case $1 in
    tests) timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -4 $out/tests.log; [ $rc -ne 0 ] && exit 1;;
    micro) ./tools/micro/exec_ops > $out/exec_ops.txt 2>&1; tail -12 $out/exec_ops.txt;;
    abtests) shift; lib=$1; shift; kexpr=$1
        PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > $out/tests_$lib.log 2>&1; rc=$?; tail -3 $out/tests_$lib.log; [ $rc -ne 0 ] && exit 1;;
    ab) shift; libs="default $1"
        for rep in 1 2; do for lib in $libs; do
          if [ $lib = default ]; then unset PTAMD_LIB; else export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; fi
          echo "== lib $lib" >> $out/ab.txt
          for sc in "scene=mesh100k spp=16 reps=3" "scene=mesh1m spp=8 bounces=16 reps=3" "scene=cornell spp=64 reps=3"; do
            timeout -k 10 300 python tools/prof_render.py $sc >> $out/ab.txt 2>&1 || exit 1
          done
        done; done; unset PTAMD_LIB; cat $out/ab.txt;;
  esac
  shift
done
