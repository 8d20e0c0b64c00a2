"""Per basic block of one kernel (hipcc -S output): instruction counts by class and the scratch traffic, so that
spill code inside the traversal loops can be told from spill code around them.
usage: python tools/isa_blocks.py file.s mangled_kernel_substring [--all]"""
import re
import sys

text = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % re.escape(sub), text, re.S | re.M)
body = m.group(2)
parts = re.split(r"^(\.LBB\d+_\d+):.*$", body, flags=re.M)
blocks = [("entry", parts[0])] + [(parts[i], parts[i + 1]) for i in range(1, len(parts), 2)]
order = {name: i for i, (name, _) in enumerate(blocks)}
print("kernel", m.group(1), "blocks", len(blocks))
tot_ld = tot_st = 0
rows = []
for i, (name, b) in enumerate(blocks):
    ops = [mm.group(1) for l in b.splitlines() for mm in [re.match(r"^\s+([a-z_0-9]+)", l)] if mm and not l.strip().startswith((";", "."))]
    n = len(ops)
    sl = sum(o.startswith("scratch_load") for o in ops)
    ss = sum(o.startswith("scratch_store") for o in ops)
    gl = sum(o.startswith(("global_load", "flat_load", "buffer_load")) for o in ops)
    gs = sum(o.startswith(("global_store", "flat_store", "buffer_store")) for o in ops)
    ds = sum(o.startswith("ds_") for o in ops)
    va = sum(o.startswith("v_") for o in ops)
    targets = re.findall(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", b)
    back = [t for t in targets if order.get(t, 1 << 30) <= i]
    tot_ld += sl
    tot_st += ss
    rows.append((name, n, va, ds, gl, gs, sl, ss, back))
for r in rows:
    if "--all" in sys.argv or r[6] or r[7] or r[8] or r[4] >= 3:
        print("%-12s insts %4d valu %4d lds %3d gload %2d gstore %2d | scratch ld %2d st %2d %s" % (r[:8] + ("<- loop back to " + ",".join(r[8]) if r[8] else "",)))
print("scratch loads %d stores %d (static)" % (tot_ld, tot_st))
