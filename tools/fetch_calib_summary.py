"""Summary of tools/fetch_calib.sh: per kernel of tools/micro/fetch_calib.hip the counters of its last launch against the bytes it
is known to request (blocks x 256 lanes x 256 records x 64 B for gather64<MiB>; 2 GiB for stream16)."""
import csv
import glob
import re
import sys
from collections import defaultdict

out = sys.argv[1]
last = defaultdict(dict)      # kernel -> counter -> value of the last dispatch
for f in sorted(glob.glob(out + "/pass_*/**/*_counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = defaultdict(lambda: defaultdict(list))
    for r in rows:
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, cs in per.items():
        for c, vals in cs.items():
            last[k][c] = sorted(vals)[-1][1]
print(open(sorted(glob.glob(out + "/pass_1.log"))[0]).read().strip())
print()
print("%-22s %12s %14s %9s | %12s %12s %9s | %s" % ("kernel", "asked MB", "FETCH_SIZE MB", "asked/F", "TCC_MISS", "miss x 64 MB", "asked/m64", "other counters"))
for k in sorted(last, key=lambda s: [int(x) for x in re.findall(r"\d+", s)]):
    c = last[k]
    m = re.search(r"gather64<(\d+)>", k)
    asked = 256 * 16 * 256 * 256 * 64 / 1e6 if m else 2048 * 1.048576
    fs = c.get("FETCH_SIZE", float("nan")) * 1024 / 1e6
    miss = c.get("TCC_MISS_sum", float("nan"))
    rest = " ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items()) if n not in ("FETCH_SIZE", "TCC_MISS_sum"))
    print("%-22s %12.1f %14.1f %9.3f | %12.4g %12.1f %9.3f | %s" % (k[-22:], asked, fs, asked / fs if fs == fs and fs else float("nan"), miss, miss * 64 / 1e6, asked / (miss * 64 / 1e6) if miss == miss and miss else float("nan"), rest))
