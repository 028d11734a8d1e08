"""Summarise a rocprofv3 kernel_trace.csv: per kernel (and per launch grid for the GEMM) count,
mean duration, total, share.  Usage: python tools/trace_summary.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    short = re.sub(r"^void ", "", short)
    short = short.split("(")[0][:70]
    if "gemm_f32" in n:
        key = "%s grid(%d,%s,%s)" % (short, int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"], r["Grid_Size_Z"])
    else:
        key = short
    g[key].append(dur)
tot = sum(sum(v) for v in g.values())
print("total kernel time %.3f ms over %d dispatches" % (tot / 1e6, len(rows)))
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print("%-86s n=%6d avg=%8.2f us total=%8.3f ms %5.1f%%" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
# concurrency: union of busy intervals vs sum of durations (1.0 = fully serial)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, cur_s, cur_e = 0, None, None
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None:
    busy += cur_e - cur_s
print("sum of kernel durations %.3f ms, union of busy time %.3f ms, overlap factor %.2f" % (tot / 1e6, busy / 1e6, tot / max(busy, 1)))
