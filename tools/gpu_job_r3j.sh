#!/bin/bash
# round 3, job j: the measurement job behind profiles/counters.json and profiles/r03/j_*: full GPU test suite, bench.py,
# the same command under rocprofv3 --kernel-trace --stats, PMC passes (megakernel on three scenes, wavefront on the Cornell
# box), tile-cost histogram, OBJ load times
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; tail -3 gpurun_out/r3j_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err; rc=$?
cut -c1-400 gpurun_out/r3j_bench.json; tail -3 gpurun_out/r3j_bench.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit 1; fi
rocprofv3 --kernel-trace --stats -d gpurun_out/r3j_bench_rocprof -o b --output-format csv -- python3 bench.py --steps 5 --no-cpu-baseline --no-variants > gpurun_out/r3j_bench_rocprof.json 2> gpurun_out/r3j_bench_rocprof.err
cut -c1-300 gpurun_out/r3j_bench_rocprof.json
rm -f gpurun_out/counters.json; cp profiles/counters.json gpurun_out/counters.json 2>/dev/null
tools/pmc_passes.sh r3j_cb scene=cornell spp=64 reps=2 > gpurun_out/r3j_pmc_cb.log 2>&1 && python3 tools/pmc_record.py cornell_1920x1080_b8_spp64 gpurun_out/counters.json gpurun_out/pmc_r3j_cb_[0-9] > gpurun_out/r3j_pmc_cb_record.log 2>&1
tail -22 gpurun_out/r3j_pmc_cb_record.log
tools/pmc_passes.sh r3j_mesh100k scene=mesh100k spp=16 reps=2 > gpurun_out/r3j_pmc_mesh100k.log 2>&1 && python3 tools/pmc_record.py mesh100k_1920x1080_b8_spp16 gpurun_out/counters.json gpurun_out/pmc_r3j_mesh100k_[0-9] > gpurun_out/r3j_pmc_mesh100k_record.log 2>&1
tools/pmc_passes.sh r3j_mesh1m scene=mesh1m spp=8 bounces=16 reps=2 > gpurun_out/r3j_pmc_mesh1m.log 2>&1 && python3 tools/pmc_record.py mesh1m_1920x1080_b16_spp8 gpurun_out/counters.json gpurun_out/pmc_r3j_mesh1m_[0-9] > gpurun_out/r3j_pmc_mesh1m_record.log 2>&1
tools/pmc_passes.sh r3j_wf scene=cornell spp=4 reps=2 variant=1 > gpurun_out/r3j_pmc_wf.log 2>&1 && python3 tools/pmc_record.py wavefront_cornell_1920x1080_b8_spp1 gpurun_out/counters.json gpurun_out/pmc_r3j_wf_[0-9] > gpurun_out/r3j_pmc_wf_record.log 2>&1
tail -22 gpurun_out/r3j_pmc_wf_record.log
timeout -k 10 300 python3 tools/tile_cost_histogram.py > gpurun_out/r3j_tile_cost.txt 2>&1; cat gpurun_out/r3j_tile_cost.txt
PTAMD_TRACE=1 timeout -k 10 300 python3 tools/obj_load_time.py 1000000 > gpurun_out/r3j_obj_load.txt 2>&1; tail -14 gpurun_out/r3j_obj_load.txt
echo done1
timeout -k 10 600 python3 tools/scaling_probe2.py 3840 2160 > gpurun_out/r3j_probe_4k.txt 2>&1; cat gpurun_out/r3j_probe_4k.txt
timeout -k 10 300 python3 tools/one_spp.py > gpurun_out/r3j_one_spp.txt 2>&1; cat gpurun_out/r3j_one_spp.txt
echo done2
