"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel-name substring, last dispatch's counters."""
import collections
import csv
import sys

path, needle = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    if needle in r["Kernel_Name"]:
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
d = sorted(agg)[-1]
c = agg[d]
for k in sorted(c):
    print(f"{k:32s} {c[k]:.4g}")
wc = c.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if k in c:
            print(f"  {k}/WAVE_CYCLES = {c[k] / wc:.3f}")
if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"  mfma pipe busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} (of 1024 SIMDs x {cyc:.3g} cycles)")
