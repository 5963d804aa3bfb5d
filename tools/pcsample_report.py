"""Reduce rocprofv3 PC-sampling CSVs (tools/pcsample.sh) to a histogram small enough to travel back from the GPU box.

    python tools/pcsample_report.py gpurun_out/pcs_TAG        -> gpurun_out/pcs_TAG/hist.json (+ head.txt with the raw header / first rows)

hist.json: {"columns": [...], "n": samples, "rows": [[count, instruction, comment, issued, type, stall], ...]} for the samples of the
dispatches whose kernel name contains BP_PCS_KERNEL (default k_physics_step).  Raw CSVs above 8 MB are deleted afterwards.
"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
want = os.environ.get("BP_PCS_KERNEL", "k_physics_step")
files = sorted(glob.glob(os.path.join(out, "raw", "**", "*.csv"), recursive=True))
head = open(os.path.join(out, "head.txt"), "w")
disp = {}
for f in files:
    head.write("== %s (%d bytes)\n" % (f, os.path.getsize(f)))
    with open(f) as fh:
        for i, line in enumerate(fh):
            if i >= 6:
                break
            head.write(line)
    if f.endswith("kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            disp[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
head.close()
hist = collections.Counter()
cols = None
n = 0
for f in files:
    if "pc_sampling" not in os.path.basename(f):
        continue
    rd = csv.DictReader(open(f))
    cols = rd.fieldnames
    for r in rd:
        k = disp.get(r.get("Dispatch_Id"), "")
        if disp and want not in k:
            continue
        n += 1
        hist[(r.get("Instruction", ""), r.get("Instruction_Comment", ""), r.get("Wave_Issued_Instruction", ""), r.get("Instruction_Type", ""),
              r.get("Stall_Reason", ""))] += 1
rows = [[c] + list(k) for k, c in hist.most_common()]
json.dump({"columns": cols, "n": n, "kernel": want, "rows": rows}, open(os.path.join(out, "hist.json"), "w"))
print("pc samples of %s: %d in %d distinct (instruction, line, issued, type, stall) rows; columns %s" % (want, n, len(rows), cols))
for f in files:
    if os.path.getsize(f) > 8 << 20:
        os.remove(f)
