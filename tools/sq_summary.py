#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass -> CSV with derived columns.

    python tools/sq_summary.py gpurun_out/r02j/sq/sq_counter_collection.csv > profiles/r02_j_sq_summary.csv

SQ_INSTS_VALU counts wave instructions; a wave64 VALU instruction occupies its SIMD (16 lanes) for 4 cycles, so
`valu_floor_us` = INSTS_VALU * 4 / (1024 SIMDs * 2.4 GHz) is the time the kernel's arithmetic alone needs on a full chip."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void smac::", "")
    if not k.startswith("k_"):
        continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
print("kernel,launches,us_under_pmc," + ",".join(cols) + ",valu_per_wave,lds_per_wave,valu_floor_us,wait_any_frac,wait_inst_frac")
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    m = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else float("nan")) for c in cols}
    waves = m["SQ_WAVES"] or float("nan")
    n = len(acc[k]["SQ_WAVES"]) or 1
    print(f"{k},{n},{sum(dur[k]) / len(dur[k]):.1f}," + ",".join(f"{m[c]:.0f}" for c in cols) +
          f",{m['SQ_INSTS_VALU'] / waves:.0f},{m['SQ_INSTS_LDS'] / waves:.0f},{m['SQ_INSTS_VALU'] * 4 / (1024 * 2.4e3):.1f},"
          f"{m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f},{m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}")
