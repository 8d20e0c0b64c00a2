"""Per-kernel totals from a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import sys
from collections import defaultdict

tot = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-50:]
            tot[k][0] += 1
            tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
allt = sum(v[1] for v in tot.values())
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-52s calls=%6d total=%10.1f us  avg=%9.2f us  %5.1f%%" % (k, n, us, us / n, 100 * us / allt))
