#!/usr/bin/env python3
"""profiles/rNN_10Mx768_pmc.json from the counter means tools/pmc_kernel.sh printed for the headline scan kernel:
  python tools/make_pmc_json.py gpurun_out/r06_pmc_flat_scan.txt r06 > profiles/r06_10Mx768_pmc.json
The file records the SHA-256 of the kernel's sources as they were when the counters were taken; bench.py reports `roofline.traffic`
from it only while that still matches (bench.pmc_traffic)."""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("quiver_amd/csrc/qv_scan.hip", "quiver_amd/csrc/qv_kernels.h", "quiver_amd/csrc/qv_device.h")


def source_hash():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    txt, tag = open(sys.argv[1]).read(), sys.argv[2]
    m = {k: (float(v), int(n)) for k, v, n in re.findall(r"^(\S+)\s+mean (\S+) over (\d+) launches", txt, re.M)}
    fetch, write = m["FETCH_SIZE"], m["WRITE_SIZE"]
    print(json.dumps({
        "kernel": "k_flat_scan", "launches": fetch[1], "FETCH_SIZE_KB_per_launch": fetch[0], "WRITE_SIZE_KB_per_launch": write[0],
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE exact",
        "hbm_bytes_per_launch": 2 * fetch[0] * 1024 + write[0] * 1024,
        "kernel_sources": list(SOURCES), "kernel_sources_sha256": source_hash(),
        "command": "tools/run_round_pmc.sh %s: tools/pmc_kernel.sh k_flat_scan ... 'FETCH_SIZE/WRITE_SIZE' -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also (each counter its own rocprofv3 --kernel-trace --pmc pass)" % tag,
    }, indent=1))
