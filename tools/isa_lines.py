"""Static instruction map of one kernel from a `-gline-tables-only -save-temps` assembly file.

    python tools/isa_lines.py /tmp/g/bp_capi-hip-amdgcn-amd-amdhsa-gfx950.s _Z20k_physics_step_sched [--blocks]

Prints, per source line (file:line of the innermost `.loc`), the number of VALU / SALU / branch / LDS / VMEM / SMEM / wait
instructions the compiler emitted for it, and the same summed per phase of substep() (phases are line ranges of
bp_physics.hpp, kept in PHASES below).  Static counts only: the dynamic weights come from tools/pcsample.sh.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime") or op.startswith("s_store"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def parse(path, kernel):
    files = {}
    rows = []          # (file, line, class, op, text)
    inside = False
    cur = ("?", 0)
    label = None
    for raw in open(path):
        line = raw.rstrip("\n")
        s = line.strip()
        m = re.match(r"\.file\s+(\d+)\s+(?:\"([^\"]*)\"\s+)?\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if not inside:
            if re.match(r"^%s\w*:" % re.escape(kernel), line):
                inside = True
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section") and rows:
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(".") and not re.match(r"^\.LBB", s) or s.startswith(";"):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            label = m.group(1)
            rows.append((cur[0], cur[1], "label", label, s))
            continue
        op = s.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            rows.append((cur[0], cur[1], classify(op), op, s))
    return rows


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    rows = parse(path, kernel)
    per = collections.defaultdict(collections.Counter)
    tot = collections.Counter()
    for f, l, c, op, _ in rows:
        if c == "label":
            continue
        per[(f, l)][c] += 1
        tot[c] += 1
    print("static instructions of %s: %s  total %d" % (kernel, dict(tot), sum(tot.values())))
    byfile = collections.defaultdict(collections.Counter)
    for (f, l), c in per.items():
        byfile[f].update(c)
    for f, c in byfile.items():
        print("  %-22s %s" % (f, dict(c)))
    if "--lines" in sys.argv:
        for (f, l), c in sorted(per.items()):
            print("%s:%d %s" % (f, l, dict(c)))


if __name__ == "__main__":
    main()
