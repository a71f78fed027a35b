"""Static instruction mix of a kernel between its phase marks (s_memrealtime of the profiling builds).

usage: python tools/isa_phase_count.py /tmp/isa/pve.s '_Z9k_rolloutILi128ELi4ELb1E'
Basic blocks are listed with their instruction classes so that rarely executed blocks (tie paths, exact XY) can be told
from the straight-line code."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            return "valu_f64"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op == "s_memrealtime":
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(key))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    phase = 0
    per_phase = [Counter()]
    blocks = []          # (phase, label, Counter)
    cur = ("entry", Counter())
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append((phase,) + cur)
                cur = (s.split(":")[0], Counter())
            continue
        m = re.match(r"^\.?LBB\d+_\d+:", s)
        if m:
            blocks.append((phase,) + cur)
            cur = (s.split(":")[0], Counter())
            continue
        op = s.split()[0]
        cls = classify(op)
        per_phase[-1][cls] += 1
        cur[1][cls] += 1
        if op == "s_memrealtime":
            blocks.append((phase,) + cur)
            cur = ("after_mark%d" % phase, Counter())
            phase += 1
            per_phase.append(Counter())
    blocks.append((phase,) + cur)
    cols = ["valu", "valu_f64", "salu", "lds", "vmem", "smem", "branch", "wait", "barrier"]
    print("%-6s" % "phase" + "".join("%9s" % c for c in cols) + "    total")
    tot = Counter()
    for i, c in enumerate(per_phase):
        print("%-6d" % i + "".join("%9d" % c[k] for k in cols) + "%9d" % sum(c.values()))
        tot.update(c)
    print("%-6s" % "all" + "".join("%9d" % tot[k] for k in cols) + "%9d" % sum(tot.values()))
    if len(sys.argv) > 3:
        for ph, label, c in blocks:
            n = sum(c.values())
            if n >= int(sys.argv[3]):
                print("phase %2d %-14s n=%4d  " % (ph, label, n) + " ".join("%s=%d" % (k, c[k]) for k in cols if c[k]))


main()
