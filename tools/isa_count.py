"""Instruction census of kernels in a hipcc -S listing: python tools/isa_count.py <listing.s> <name substring> ...
Per kernel whose mangled name holds a substring: instructions in all, f64 / f32 VALU, f64 division / square-root sequence members, LDS, global memory, scalar."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
subs = sys.argv[2:]
parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if not any(s in name for s in subs):
        continue
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    f64 = sum("f64" in x for x in ins)
    f32 = sum("f32" in x for x in ins)
    slow = sum(x.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")) for x in ins)
    print(f"{dn[:90]}: {len(ins)} instructions, f64 {f64} (division / sqrt members {slow}), f32 {f32}, v_* {sum(x.startswith('v_') for x in ins)}, "
          f"ds_* {sum(x.startswith('ds_') for x in ins)}, global/buffer/flat {sum(x.startswith(('global_', 'buffer_', 'flat_')) for x in ins)}, s_* {sum(x.startswith('s_') for x in ins)}")
