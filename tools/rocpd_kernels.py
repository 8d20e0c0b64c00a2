"""Per-kernel totals of a rocprofv3 run kept in rocpd (sqlite) form: python3 tools/rocpd_kernels.py <results.db> [first-kernel-substring]
With a substring, only the launches from the LAST launch of a kernel with that name on are counted (one build / one frame)."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
if len(sys.argv) > 2:
    at = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
    rows = rows[at[-1]:]
agg = collections.OrderedDict()
for name, s, e in rows:
    m = re.search(r"(k_\w+|wf_\w+)", name)
    short = m.group(1) if m else name[:48]
    a = agg.setdefault(short, [0, 0.0, []])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2].append((e - s) / 1e3)
total = 0.0
for k, (c, t, l) in agg.items():
    print("%-44s %4d calls %9.1f us   %s" % (k, c, t, " ".join("%.0f" % x for x in l[:16])))
    total += t
print("kernel time %.1f us, first start to last end %.1f us" % (total, (rows[-1][2] - rows[0][1]) / 1e3))
