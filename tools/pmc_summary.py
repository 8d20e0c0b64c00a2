"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name prefix."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render" in r["Kernel_Name"] or "wf_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-60s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
