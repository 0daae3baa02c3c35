#!/bin/bash
# round 5: FETCH_SIZE / WRITE_SIZE calibration on known byte counts in this code's access shapes (tools/microbench/fetch_calib.hip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05cal; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/fetch_calib tools/microbench/fetch_calib.hip || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $O/fetch_calib > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $O/fetch_calib > $O/write.log 2>&1
python3 - $O <<'PY'
import collections, csv, glob, sys
O = sys.argv[1]
known = {"k_read16": 12 * 24 * (1 << 20) * 4, "k_read4": 24 * (1 << 20) * 4, "k_write4": 24 * (1 << 20) * 4, "k_write16": 12 * 24 * (1 << 20) * 4}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for path in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        b = known.get(k)
        if b:
            mean = sum(v) / len(v) * 1024
            print(f"{counter:10s} {k:10s} launches {len(v):4d}  counter {mean / 1e6:9.2f} MB  known {b / 1e6:9.2f} MB  counter / known = {mean / b:.3f}")
PY
