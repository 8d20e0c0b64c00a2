#!/bin/bash
# usage: tools/pmc_lds.sh <tag> <prof_render args...>  -- LDS-array counters of the render kernel (one pass)
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmclds_${tag} -o p --output-format csv -- python3 tools/prof_render.py "$@" > gpurun_out/pmclds_${tag}.log 2>&1
tail -2 gpurun_out/pmclds_${tag}.log
python3 - <<PY
import csv, glob
from collections import defaultdict
v = defaultdict(list)
for f in glob.glob("gpurun_out/pmclds_${tag}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, a in sorted(v.items()):
    print("%-24s %s" % (k, " ".join("%.4g" % x for x in a)))
PY
