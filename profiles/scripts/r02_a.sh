#!/bin/bash
# round 2, GPU session A: XCD placement probe, precision probes (default vs f64 constitutive), A/B timing, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
./tools/microbench/xcc_probe > $O/xcc.txt 2>&1
make -C oracle -s
timeout -k 10 400 python tools/prec_probe.py --precision float32 --out $O/prec_f32_default.json > $O/prec_f32_default.log 2>&1
SMAC_LIB=$PWD/softmac_amd/lib/libsoftmac_hip_cf64.so timeout -k 10 400 python tools/prec_probe.py --precision float32 --out $O/prec_f32_cf64.json > $O/prec_f32_cf64.log 2>&1
timeout -k 10 400 python tools/prec_probe.py --precision float64 --out $O/prec_f64.json > $O/prec_f64.log 2>&1
bash tools/ab.sh softmac_amd/lib/libsoftmac_hip.so softmac_amd/lib/libsoftmac_hip_cf64.so > $O/ab.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o sq -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r02a"
files = glob.glob(O + "/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
with open(O + "/sq_summary.csv", "w") as fh:
    names = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
    fh.write("kernel,launches," + ",".join(n_ + "_per_launch" for n_ in names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
        if n[k]: fh.write(k + f",{n[k]}," + ",".join(f"{acc[k][c]/n[k]:.4g}" for c in names) + "\n")
print(open(O + "/sq_summary.csv").read()[:3000])
PY
cat $O/xcc.txt; cat $O/ab.txt; grep -v "^ " $O/prec_f32_default.log; grep -v "^ " $O/prec_f32_cf64.log; grep -v "^ " $O/prec_f64.log
