#!/usr/bin/env python3
"""Which kernels libqv.so ships, and which of them the oracle-checked GPU tests launch.

  python tools/kernel_coverage.py list [libqv.so]                 the gfx950 kernels of the library (no GPU needed)
  python tools/kernel_coverage.py check <trace-dir-or-csv ...>    compare with the kernel names of rocprofv3 --kernel-trace runs
                                                                  (--output-format csv) and FAIL (exit 1) on a shipped kernel no run launched

The shipped list comes from the library itself: its .hip_fatbin section is a sequence of clang offload bundles (one per translation
unit); the hipv4-amdgcn-amd-amdhsa--gfx950 entry of each is an ELF code object whose kernels are the symbols with a `.kd` (kernel
descriptor) twin.  Names are demangled with c++filt and cut at the argument list, which is also what rocprofv3 reports.

The traces come from `tools/run_kernel_coverage.sh` (pytest -m gpu under rocprofv3 --kernel-trace, the program directly after `--`;
test files in a few chunks so that one trace stays small)."""
import csv
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        pos = data.find(magic, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", data, pos + len(magic))
        p = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos += len(magic)


def short(name):
    """demangled name without return type, namespace and argument list: k_flat_scan<0, 16, true>"""
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut].strip()
    if name.startswith("void "):
        name = name[5:]
    return name.replace("qv::", "").replace("(anonymous namespace)::", "")


def shipped(so):
    names = set()
    for co in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", f.name], capture_output=True, text=True, check=True).stdout
        kds = [ln.split()[-1][:-3] for ln in out.splitlines() if ln.rstrip().endswith(".kd")]
        if kds:
            dem = subprocess.run(["c++filt"], input="\n".join(kds), capture_output=True, text=True, check=True).stdout.splitlines()
            names.update(short(d) for d in dem)
    return names


def launched(paths):
    names, files = {}, []
    for p in paths:
        if os.path.isdir(p):
            files += glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
        else:
            files.append(p)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                n = short(r["Kernel_Name"])
                names[n] = names.get(n, 0) + 1
    return names, files


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in ("list", "check"):
        sys.exit(__doc__)
    if sys.argv[1] == "list":
        so = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "quiver_amd", "lib", "libqv.so")
        ks = sorted(shipped(so))
        print("\n".join(ks))
        fam = sorted({re.sub(r"<.*", "", k) for k in ks})
        sys.stderr.write("%d kernel instantiations of %d templates in %s\n" % (len(ks), len(fam), so))
        return
    so = os.path.join(ROOT, "quiver_amd", "lib", "libqv.so")
    ks = shipped(so)
    got, files = launched(sys.argv[2:])
    mine = {k: v for k, v in got.items() if k in ks}
    missing = sorted(ks - set(got))
    fam = sorted({re.sub(r"<.*", "", k) for k in ks})
    print("libqv.so ships %d kernel instantiations of %d templates; %d trace files, %d launches of %d distinct shipped kernels" % (
        len(ks), len(fam), len(files), sum(mine.values()), len(mine)))
    print("shipped kernels never launched by the traced runs: %d" % len(missing))
    for k in missing:
        print("  UNLAUNCHED  " + k)
    print("launch counts:")
    for k in sorted(mine):
        print("  %8d  %s" % (mine[k], k))
    sys.exit(1 if missing else 0)


if __name__ == "__main__":
    main()
