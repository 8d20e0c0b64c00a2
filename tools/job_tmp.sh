cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out/r5s; mkdir -p $out
for i in 1 2; do timeout -k 10 400 python bench.py --steps 4 --no-cpu-baseline > $out/bench$i.json 2> $out/bench$i.err; python - <<PY
import json
b=json.load(open('$out/bench$i.json'))
print(round(b['value'],1), [(c['config'], round(c['msamples_per_s'],1), round(c['kernel']['mean_launch_ms'],2)) for c in b['configs']])
PY
done
