#!/usr/bin/env python3
"""Where the scratch (spill) traffic of a kernel sits relative to its matrix-instruction loops:
python tools/asm_loops.py quiver_amd/csrc/qv_batched.hip [name-substring]
Compiles to gfx950 assembly and prints, per kernel, every basic block that holds matrix instructions with its global loads,
LDS ops and scratch loads/stores (a scratch access inside such a block sits in the same in-order vmcnt queue as the row requests)."""
import re
import subprocess
import sys

src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
asm = "/tmp/asm_loops.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, src],
                      stderr=subprocess.DEVNULL)
text = open(asm).read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN2qv\S+):\s", text, flags=re.M)]
for i, (pos, name) in enumerate(starts):
    body = text[pos: starts[i + 1][0] if i + 1 < len(starts) else len(text)]
    body = body.split(".section")[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("void qv::", "")
    if pat not in dn:
        continue
    blocks, cur, label = [], [], "entry"
    for line in body.split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append((label, cur)); cur, label = [], line.split(":")[0]
        else:
            cur.append(line)
    blocks.append((label, cur))
    tot = sum(1 for l in body.split("\n") if re.search(r"\bscratch_(load|store)", l))
    print("%s: %d scratch instructions in all" % (dn, tot))
    for lab, b in blocks:
        m = sum(1 for l in b if "v_mfma" in l)
        if m == 0:
            continue
        print("    %-12s mfma %3d  global_load %3d  ds_read %3d ds_write %3d  scratch_load %3d scratch_store %3d  v_accvgpr %3d  s_waitcnt vmcnt %d" % (
            lab, m, sum(1 for l in b if "global_load" in l), sum(1 for l in b if "ds_read" in l or "ds_load" in l), sum(1 for l in b if "ds_write" in l or "ds_store" in l),
            sum(1 for l in b if "scratch_load" in l), sum(1 for l in b if "scratch_store" in l), sum(1 for l in b if "v_accvgpr" in l),
            sum(1 for l in b if "s_waitcnt" in l and "vmcnt" in l)))
    if len(sys.argv) > 3:                                            # third argument: also list every block that touches scratch
        for lab, b in blocks:
            sl = sum(1 for l in b if "scratch_load" in l); ss = sum(1 for l in b if "scratch_store" in l)
            if sl + ss:
                n = sum(1 for l in b if l.startswith("\t") and not l.strip().startswith((";", ".")))
                print("      scratch: %-12s instrs %4d  loads %3d stores %3d  ds %3d  valu %4d" % (lab, n, sl, ss, sum(1 for l in b if "\tds_" in l), sum(1 for l in b if re.match(r"\tv_", l))))
