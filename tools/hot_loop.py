"""Instruction mix of the basic blocks of a kernel (from hipcc -S output): finds the node-visit block (the one
with the most ds_read_b64) and the triangle block, prints their instruction counts by class.
usage: python tools/hot_loop.py file.s mangled_kernel_substring"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
sub = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % re.escape(sub), text, re.S | re.M)
body = m.group(2)
blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
print("kernel", m.group(1), "blocks", len(blocks), "instructions", sum(1 for l in body.splitlines() if re.match(r"^\s+[vsdgb]\w+", l)))


def mix(b):
    c = Counter()
    for l in b.splitlines():
        mm = re.match(r"^\s+([a-z_0-9]+)", l)
        if not mm or l.strip().startswith((";", ".")):
            continue
        op = mm.group(1)
        k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "flat_", "buffer_")) else "other"
        c[k] += 1
    return c


scored = sorted(((b.count("ds_read_b64") + b.count("ds_read2_b64") * 2, i) for i, b in enumerate(blocks)), reverse=True)
for score, i in scored[:2]:
    print("block %d (ds_read_b64 x%d):" % (i, score), dict(mix(blocks[i])))
tot = mix(body)
print("whole kernel:", dict(tot))
