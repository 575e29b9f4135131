"""summarise QV_TRACE lines of qv_graph_search from stdin: pass 1 / pass 2 durations"""
import re, sys
import numpy as np
p1, p2 = [], []
for line in sys.stdin:
    m = re.search(r"pass 1 \+ download ([\d.]+) ms \((\d+) flagged", line)
    if m: p1.append(float(m.group(1)))
    m = re.search(r"pass 2 .*download ([\d.]+) ms", line)
    if m: p2.append(float(m.group(1)))
for name, v in (("pass1", p1), ("pass2", p2)):
    if v:
        v = np.array(v)
        print(name, "n=%d mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f ms" % (v.size, v.mean(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
