#!/bin/bash
# usage: tools/gpu_job.sh <tag> [tests] [ab "<names>"] [rates]   -- the round's standing measurement job, pieces by keyword:
#   tests  the GPU test suite (-m gpu), log under gpurun_out/<tag>/tests.log
#   abtests <lib> <-k expr>   GPU tests matching the expression against an A/B library
#   bench  bench.py, then bench.py --steps 5 under rocprofv3 --kernel-trace --stats
#   pmc    the rocprofv3 --pmc passes of the six tracked launch shapes -> gpurun_out/counters.json (copy it to profiles/)
#   sweep  tools/sweep.py --count on every BASELINE config
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
    bench) timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err; rc=$?; tail -2 $out/bench.err; cut -c1-300 $out/bench.json; [ $rc -ne 0 ] && { echo "bench failed rc=$rc"; exit 1; }
        rocprofv3 --kernel-trace --stats -d $out/bench_rocprof -o b --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline --no-variants --no-configs > $out/bench_rocprof.json 2> $out/bench_rocprof.err
        cut -c1-200 $out/bench_rocprof.json; head -4 $out/bench_rocprof/b_kernel_stats.csv;;
    pmc) # the PMC passes behind profiles/counters.json (one counter set per pass; never with the runtime traces)
        rm -f gpurun_out/counters.json; cp profiles/counters.json gpurun_out/counters.json 2>/dev/null
        tools/pmc_passes.sh ${tag}_cb scene=cornell spp=64 reps=2 > $out/pmc_cb.log 2>&1 && python3 tools/pmc_record.py cornell_1920x1080_b8_spp64 gpurun_out/counters.json gpurun_out/pmc_${tag}_cb_[0-9] > $out/pmc_cb_record.log 2>&1; tail -24 $out/pmc_cb_record.log
        tools/pmc_passes.sh ${tag}_mesh100k scene=mesh100k spp=64 reps=2 > $out/pmc_mesh100k.log 2>&1 && python3 tools/pmc_record.py mesh100k_1920x1080_b8_spp64 gpurun_out/counters.json gpurun_out/pmc_${tag}_mesh100k_[0-9] > $out/pmc_mesh100k_record.log 2>&1; tail -24 $out/pmc_mesh100k_record.log
        tools/pmc_passes.sh ${tag}_mesh1m scene=mesh1m spp=64 bounces=16 reps=2 > $out/pmc_mesh1m.log 2>&1 && python3 tools/pmc_record.py mesh1m_1920x1080_b16_spp64 gpurun_out/counters.json gpurun_out/pmc_${tag}_mesh1m_[0-9] > $out/pmc_mesh1m_record.log 2>&1; tail -24 $out/pmc_mesh1m_record.log
        tools/pmc_passes.sh ${tag}_c1 scene=cornell W=256 H=256 bounces=4 spp=16 reps=8 > $out/pmc_c1.log 2>&1 && python3 tools/pmc_record.py cornell_256x256_b4_spp16 gpurun_out/counters.json gpurun_out/pmc_${tag}_c1_[0-9] > $out/pmc_c1_record.log 2>&1; tail -6 $out/pmc_c1_record.log
        tools/pmc_passes.sh ${tag}_c4 scene=cornell W=3840 H=2160 spp=16 reps=2 > $out/pmc_c4.log 2>&1 && python3 tools/pmc_record.py cornell_3840x2160_b8_spp16 gpurun_out/counters.json gpurun_out/pmc_${tag}_c4_[0-9] > $out/pmc_c4_record.log 2>&1; tail -6 $out/pmc_c4_record.log
        tools/pmc_passes.sh ${tag}_wf scene=cornell spp=4 reps=2 variant=1 wf_streams=1 > $out/pmc_wf.log 2>&1 && python3 tools/pmc_record.py wavefront_cornell_1920x1080_b8_spp1 gpurun_out/counters.json gpurun_out/pmc_${tag}_wf_[0-9] > $out/pmc_wf_record.log 2>&1; tail -24 $out/pmc_wf_record.log;;
    sweep) timeout -k 10 600 python tools/sweep.py --count --what cb,c1c4,mesh100k,mesh1m > $out/sweep_all.txt 2>&1; grep -v "^$" $out/sweep_all.txt | cut -c1-330;;
  esac
  shift
done
