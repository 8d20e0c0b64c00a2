cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out/r4t; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "mesh or wide or config or adversarial or register or closest or wavefront" > $out/tests.log 2>&1; rc=$?; tail -3 $out/tests.log; [ $rc -ne 0 ] && exit 1
tools/gpu_job.sh r4t ab "nest"
timeout -k 10 300 python tools/sweep_phase.py mesh100k,mesh1m --nm 6 --lm 0,2,4,8,12 --reps 4 > $out/leafmin.txt 2>&1; cat $out/leafmin.txt
