#!/usr/bin/env python3
"""The dispatch table of DESIGN.md §4.1, from the kernels a call actually launches.

  python tools/dispatch_table.py shape <name>          run ONE shape (under `rocprofv3 --kernel-trace --output-format csv -d <dir>`):
                                                       set-up, a marker (qv_index_get_row -> k_fetch_row), the call, the marker again
  python tools/dispatch_table.py table <dir>           every <dir>/<shape>/**/*kernel_trace.csv -> a markdown table: shape -> the kernels
                                                       launched between the two markers, in launch order (repeats collapsed)
tools/run_dispatch_table.sh drives both on the GPU box and writes profiles/rNN_dispatch_table.md."""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# name -> (what, metric, rows, dim, nq, k, kind)
SHAPES = {
    "flat_1q_k10_10kx128":   ("qv_index_search", "cosine", 10_000, 128, 1, 10, "flat"),
    "flat_1q_k10_30kx768":   ("qv_index_search", "cosine", 30_000, 768, 1, 10, "flat"),
    "flat_4q_k10_30kx768":   ("qv_index_search", "cosine", 30_000, 768, 4, 10, "flat"),
    "flat_1q_k10_1Mx768":    ("qv_index_search", "cosine", 1_000_000, 768, 1, 10, "flat"),
    "flat_1q_k100_1Mx768":   ("qv_index_search", "cosine", 1_000_000, 768, 1, 100, "flat"),
    "flat_1q_k1000_1Mx768":  ("qv_index_search", "cosine", 1_000_000, 768, 1, 1000, "flat"),
    "flat_1q_k20000_1Mx768": ("qv_index_search", "cosine", 1_000_000, 768, 1, 20000, "flat"),
    "flat_4q_k10_1Mx768":    ("qv_index_search", "cosine", 1_000_000, 768, 4, 10, "flat"),
    "flat_8q_k10_1Mx768":    ("qv_index_search", "cosine", 1_000_000, 768, 8, 10, "flat"),
    "flat_16q_k10_1Mx768":   ("qv_index_search", "cosine", 1_000_000, 768, 16, 10, "flat"),
    "flat_64q_k10_1Mx768":   ("qv_index_search", "cosine", 1_000_000, 768, 64, 10, "flat"),
    "flat_256q_k10_1Mx768":  ("qv_index_search", "cosine", 1_000_000, 768, 256, 10, "flat"),
    "flat_256q_k100_1Mx768": ("qv_index_search", "cosine", 1_000_000, 768, 256, 100, "flat"),
    "flat_256q_k1000_1Mx768": ("qv_index_search", "cosine", 1_000_000, 768, 256, 1000, "flat"),
    "flat_256q_k10_1Mx768_bf16rows": ("qv_index_search, QV_FLAG_BF16_ROWS", "cosine", 1_000_000, 768, 256, 10, "flat_bf16"),
    "flat_256q_k10_1Mx768_fp32": ("qv_index_search, qv_index_set_filter(fp32)", "dot_product", 1_000_000, 768, 256, 10, "flat_fp32"),
    "flat_256q_k10_1Mx2048": ("qv_index_search", "cosine", 400_000, 2048, 256, 10, "flat"),
    "flat_256q_k10_1Mx200":  ("qv_index_search", "euclidean", 1_000_000, 200, 256, 10, "flat"),
    "flat_16q_k10_1Mx768_filter_off": ("qv_index_search, filter off", "cosine", 1_000_000, 768, 16, 10, "flat_off"),
    "flat_16q_k10_1Mx768_l1": ("qv_index_search", "l1", 1_000_000, 768, 16, 10, "flat"),
    "graph_1q_ef128_200kx128":    ("qv_graph_search", "cosine", 200_000, 128, 1, 10, "graph"),
    "graph_200q_ef128_200kx128":  ("qv_graph_search", "cosine", 200_000, 128, 200, 10, "graph"),
    "graph_600q_ef128_200kx128":  ("qv_graph_search", "cosine", 200_000, 128, 600, 10, "graph"),
    "graph_4096q_ef128_200kx128": ("qv_graph_search", "cosine", 200_000, 128, 4096, 10, "graph"),
    "graph_4096q_ef512_200kx128": ("qv_graph_search (efSearch 512)", "cosine", 200_000, 128, 4096, 10, "graph512"),
}


def run_shape(name):
    import numpy as np
    import quiver_amd
    from quiver_amd.device_index import DeviceGraph, random_levels
    from tests import _oracle as O
    what, metric, n, dim, nq, k, kind = SHAPES[name]
    idx = quiver_amd.DeviceIndex(dim, metric, rowmajor=kind.startswith("graph"), bf16_rows=(kind == "flat_bf16"))
    idx.reserve(n); idx.add_synthetic(20260424, 0, n)
    qs = O.gen_rows(20260425, 0, nq, dim)
    if kind == "flat_fp32": idx.set_filter("fp32")
    if kind == "flat_off": idx.set_filter("off")
    if kind.startswith("graph"):
        g = DeviceGraph.build(idx, random_levels(n, 1, 1), m=16, max_m0=32, ef_construction=100)
        ef = 512 if kind == "graph512" else 128
        g.search(qs, k, ef)                      # warm
        idx.get_row(0); g.search(qs, k, ef); idx.get_row(0)
    else:
        idx.search(qs, k)
        idx.get_row(0); idx.search(qs, k); idx.get_row(0)


def table(root):
    print("| call | metric | rows × dim | queries | k | kernels launched, in order |\n|---|---|---|---|---|---|")
    for name, (what, metric, n, dim, nq, k, kind) in SHAPES.items():
        files = glob.glob(os.path.join(root, name, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            print("| `%s` | %s | %d × %d | %d | %d | (no trace) |" % (what, metric, n, dim, nq, k)); continue
        rows = sorted((r for f in files for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
        names = [re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("qv::", "") for r in rows]
        marks = [i for i, x in enumerate(names) if x.startswith("k_fetch_row")]
        seq = names[marks[-2] + 1:marks[-1]] if len(marks) >= 2 else names
        out = []
        for x in seq:
            if out and out[-1][0] == x: out[-1][1] += 1
            else: out.append([x, 1])
        print("| `%s` | %s | %d × %d | %d | %d | %s |" % (what, metric, n, dim, nq, k, " → ".join("`%s`%s" % (a, "" if c == 1 else " ×%d" % c) for a, c in out)))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "shape": run_shape(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "table": table(sys.argv[2])
    elif len(sys.argv) >= 2 and sys.argv[1] == "names": print(" ".join(SHAPES))
    else: sys.exit(__doc__)
