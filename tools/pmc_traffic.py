#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as
MI355X_MICROARCH.md prescribes) -> profiles/traffic_latest.json (read by bench.py for roofline.traffic).

gfx950 correction (same guide): FETCH_SIZE tallies 128-B read requests at 64 B, i.e. reports 1/2 of the bytes of a
coalesced stream.  The guide states that for 16-B-per-lane loads; CALIBRATED in round 5 on known byte counts in this code's
access shapes (tools/microbench/fetch_calib.hip, profiles/r05_fetch_calib.txt): FETCH_SIZE / bytes = 0.500 for 16 B per lane
AND for the particle kernels' 4-B-per-lane SoA row reads (24 rows of 1M floats, working set beyond the Infinity Cache);
WRITE_SIZE / bytes = 1.000 for both store shapes.  So bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, no further factor.
(Round 4 had read 0.56-0.58 off the particle kernels themselves: that was their re-read traffic, not the counter.)"""
import collections
import csv
import json
import sys

NAMES = {"k_p2g<float, true, false>": "p2g", "k_p2g_grad<float, false, true>": "p2g_grad", "k_g2p<float>": "g2p", "k_g2p_grad<float, false>": "g2p_grad",
         "k_grid_op<float, false>": "grid_op", "k_grid_op_grad<float, false>": "grid_op_grad", "k_clear_active<float>": "clear_grid",
         "k_contact_hits<float>": "contact", "k_contact_grad<float>": "contact_grad", "k_reduce_aout<float>": "reduce_agvout",
         "k_grid_save<float>": "grid_checkpoint", "k_p2g_g2p_grad<float, false>": "p2g_g2p_grad", "k_reduce_grid_grad<float>": "reduce_agvout",
         "k_contact_hits<float, false>": "contact", "k_contact_grad<float, true, false>": "contact_grad", "k_grid_restore<float>": "grid_restore",
         # round 3: the checkpoint save rides in k_g2p's launch, the next restore in the grid-adjoint reduction's
         "k_g2p<float, false>": "g2p", "k_g2p<float, true>": "g2p", "k_reduce_grid_grad_ahead<float>": "reduce_agvout",
         # round 4: the two-entry material table added a template parameter (false = one material, the benchmarked instantiations)
         "k_p2g<float, true, false, false>": "p2g", "k_p2g_grad<float, false, true, false>": "p2g_grad",
         # round 4: G2P of substep f + P2G of substep f+1 in one launch (with the checkpoint save in its first workgroups)
         "k_g2p_p2g<float, true>": "g2p_p2g", "k_g2p_p2g<float, false>": "g2p_p2g",
         # (k_grid_op got a third template parameter: the slab loop's instantiation packs / adds the shared planes itself)
         "k_grid_op<float, false, false>": "grid_op",
         # round 5: the slab loop's instantiations pack / add the shared planes of grid_v_out.grad themselves (a template parameter more)
         "k_reduce_aout<float, false>": "reduce_agvout", "k_grid_op_grad<float, false, false>": "grid_op_grad"}


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void smac::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_by_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_by_kernel(sys.argv[2], "WRITE_SIZE")
out, rows = {}, []
for k, short in NAMES.items():
    if k in fetch and k in write:
        b = (2 * fetch[k] + write[k]) * 1024
        out[short] = b
        rows.append((short, fetch[k], write[k], b))
# bench.py quotes these numbers only while the kernel sources are the ones they were measured on
import hashlib, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha1()
for name in ("smac_kernels.hpp", "smac_math.hpp", "smac_sort.hpp", "softmac_hip.hip", "smac_cloth.hpp", "smac_cloth_kernels.hpp", "smac_migrate.hpp", "smac_comm.hpp"):   # = bench.py:kernel_sources_sha1
    h.update(open(os.path.join(root, "softmac_amd", "csrc", name), "rb").read())
out["kernel_sources_sha1"] = h.hexdigest()
try:
    out["measured"] = "PMC passes on commit " + subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() + \
        (" (" + sys.argv[4] + ")" if len(sys.argv) > 4 else "")
except Exception:
    out["measured"] = sys.argv[4] if len(sys.argv) > 4 else "unknown"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_per_launch(2F+W)")
for r in rows:
    print("%s,%.1f,%.1f,%.0f" % r)
