"""Static guard against the hang class of round 2 (profiles/r02/v_*): a wave-uniform loop-carried value (the persistent
loop's work item, the wavefront kernel's ray-range cursor) that lives in a VGPR can be spilled lane by lane under a
partial exec mask; lane 0 then reads back a stale slot and publishes a wrong pass number (consumers wait for ever).
The source keeps these values in scalar registers (readfirstlane at every redefinition); this script checks, in the
ISA of EVERY k_render / wf_intersect instance (hipcc -S output), that the compiler did:

  1. the result of every RETURNING global atomic on a wave-uniform address (scalar base: work fetch, exit count) is read
     by v_readfirstlane_b32 before anything else touches that VGPR (atomics with per-lane 64-bit addresses -- the three
     lanes that reserve queue space in wf_intersect -- return per-lane values and are not meant);
  2. the data operand of every agent-scope store (global_store_dword ... sc1: tile_done[] publish, counter resets) was
     moved from a scalar register or an immediate in the SAME basic block -- no long-lived VGPR copy of a pass number;
  3. no VGPR that is a plain copy of a scalar register (v_mov_b32 vX, sY) is stored to scratch in the block it was
     copied in AND read back from that slot as a wave-uniform (v_readfirstlane_b32, or the data of an agent-scope store): a
     spilled copy of a wave-uniform value is exactly the round-2 bug.  (A per-lane variable that only starts from a scalar --
     the suspend schedule's sample counter -- may be initialised that way.)

usage: python tools/check_isa.py file.s [file.s ...]   -> exit 1 and a report if any instance fails."""
import re
import sys

KERNELS = re.compile(r"^(_Z\w*(?:k_render|wf_intersect)\w*):", re.M)


def instances(text):
    for m in KERNELS.finditer(text):
        end = text.index("s_endpgm", m.end())
        yield m.group(1), text[m.end():end]


def insts(body):
    out = []
    for line in body.splitlines():
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) and not t.startswith(".LBB"):
            continue
        out.append(t.split(";")[0].strip())
    return out


def regs_of(tok):
    """v3 -> {3}; v[2:3] -> {2, 3}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(inst):
    parts = inst.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


def uniform_use_of_slot(L, slot):
    """index of a v_readfirstlane_b32 (or the data of an agent-scope store) fed by a reload of scratch slot `slot`, or None"""
    for i, ins in enumerate(L):
        if not ins.startswith("scratch_load") or not re.search(r"offset:%s\b" % slot, ins):
            continue
        dst = set().union(*[regs_of(o) for o in operands(ins)[:1]])
        for j in range(i + 1, min(len(L), i + 200)):
            nxt = L[j]
            if nxt.startswith(".LBB"):
                break
            nops = operands(nxt)
            if not nops:
                continue
            op = nxt.split()[0]
            if op == "v_readfirstlane_b32" and len(nops) > 1 and regs_of(nops[1]) & dst:
                return j
            if op == "global_store_dword" and re.search(r"\bsc1\b", nxt) and len(nops) > 1 and regs_of(nops[1]) & dst:
                return j
            if not op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "s_", "v_cmp")) and regs_of(nops[0]) & dst:
                dst = dst - regs_of(nops[0])           # redefined
                if not dst:
                    break
    return None


def check(name, body):
    errs = []
    L = insts(body)
    n_atomic = n_pub = 0
    for i, ins in enumerate(L):
        op = ins.split()[0]
        ops = operands(ins)
        # ---- 1. returning atomics: global_atomic_<op> vDst, vAddr, vData, s[..] ... sc0
        if op.startswith("global_atomic") and re.search(r"\bsc0\b", ins) and len(ops) >= 4 and regs_of(ops[0]) and re.match(r"s\[", ops[3]):
            n_atomic += 1
            dst = regs_of(ops[0])
            ok = False
            for nxt in L[i + 1:i + 40]:
                if nxt.startswith(".LBB") or nxt.split()[0].startswith(("s_", "buffer_inv", "buffer_wbl2")):
                    continue
                nops = operands(nxt)
                touched = set().union(*[regs_of(o) for o in nops]) if nops else set()
                if not (touched & dst):
                    continue
                ok = nxt.split()[0] == "v_readfirstlane_b32" and regs_of(nops[1]) <= dst
                break
            if not ok:
                errs.append("returning atomic at #%d (%s): result is not read by v_readfirstlane first" % (i, ins))
        # ---- 2. agent-scope stores
        if op == "global_store_dword" and re.search(r"\bsc1\b", ins):
            n_pub += 1
            data = regs_of(ops[1]) if len(ops) > 1 else set()
            ok = False
            for prv in reversed(L[max(0, i - 60):i]):
                if prv.startswith(".LBB"):
                    break
                if prv.split()[0].startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "s_")):
                    continue                      # defines no VGPR
                pops = operands(prv)
                if pops and regs_of(pops[0]) & data:
                    ok = prv.split()[0] in ("v_mov_b32_e32", "v_mov_b32") and (re.fullmatch(r"s\d+|-?\d+|0x[0-9a-f]+|vcc_lo|vcc_hi", pops[1]) is not None)
                    break
            if not ok:
                errs.append("agent-scope store at #%d (%s): data is not a fresh copy of a scalar register" % (i, ins))
    # ---- 3. spilled copies of scalars, per basic block
    copies = {}
    for i, ins in enumerate(L):
        if ins.startswith(".LBB"):
            copies = {}
            continue
        op = ins.split()[0]
        ops = operands(ins)
        if op.startswith("scratch_store"):
            for o in ops:
                for r in regs_of(o):
                    if r in copies:
                        # a per-lane variable that merely STARTS from a scalar (the suspend schedule's sample counter = the work
                        # item's first sample) is stored like this too.  What made round 2 hang is the way back: the slot reloaded
                        # and taken for wave-uniform again.
                        m = re.search(r"offset:(\d+)", ins)
                        slot = m.group(1) if m else "0"
                        use = uniform_use_of_slot(L, slot)
                        if use is not None:
                            errs.append("scratch store at #%d (%s): v%d is a copy of %s made in this block, and the slot is read back as a wave-uniform at #%d (%s)" % (
                                i, ins, r, copies[r], use, L[use]))
            continue
        if ops:
            for r in regs_of(ops[0]):
                copies.pop(r, None)
            if op in ("v_mov_b32_e32", "v_mov_b32") and re.fullmatch(r"s\d+", ops[1]):
                for r in regs_of(ops[0]):
                    copies[r] = ops[1]
    return errs, n_atomic, n_pub


def main():
    bad = 0
    total = 0
    for path in sys.argv[1:]:
        text = open(path).read()
        for name, body in instances(text):
            total += 1
            errs, na, npub = check(name, body)
            if errs:
                bad += 1
                print("FAIL %s" % name)
                for e in errs:
                    print("     " + e)
            else:
                print("ok   %-90s returning atomics %d, agent-scope stores %d" % (name, na, npub))
    print("%d kernel instances checked, %d failed" % (total, bad))
    if total == 0:
        print("no k_render / wf_intersect instance found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
