#!/bin/bash
# bench + rocprofv3 kernel stats of the same command + PMC passes of the three scenes with the current kernel
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err; rc=$?
cat gpurun_out/bench_r2.json; tail -3 gpurun_out/bench_r2.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit 1; fi
rocprofv3 --kernel-trace --stats -d gpurun_out/bench_r2_rocprof -o b --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline --no-variants > gpurun_out/bench_r2_rocprof.json 2> gpurun_out/bench_r2_rocprof.err
cat gpurun_out/bench_r2_rocprof.json
tools/pmc_passes.sh cb scene=cornell spp=64 reps=2 > gpurun_out/pmc_cb.log 2>&1 && python3 tools/pmc_record.py cornell_1920x1080_b8_spp64 gpurun_out/counters.json gpurun_out/pmc_cb_[0-9] > gpurun_out/pmc_cb_record.log 2>&1
tail -25 gpurun_out/pmc_cb_record.log
tools/pmc_passes.sh mesh100k scene=mesh100k spp=16 reps=2 > gpurun_out/pmc_mesh100k.log 2>&1 && python3 tools/pmc_record.py mesh100k_1920x1080_b8_spp16 gpurun_out/counters.json gpurun_out/pmc_mesh100k_[0-9] > gpurun_out/pmc_mesh100k_record.log 2>&1
tools/pmc_passes.sh mesh1m scene=mesh1m spp=8 bounces=16 reps=2 > gpurun_out/pmc_mesh1m.log 2>&1 && python3 tools/pmc_record.py mesh1m_1920x1080_b16_spp8 gpurun_out/counters.json gpurun_out/pmc_mesh1m_[0-9] > gpurun_out/pmc_mesh1m_record.log 2>&1
tail -4 gpurun_out/pmc_mesh1m_record.log
