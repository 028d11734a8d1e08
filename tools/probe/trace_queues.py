"""Per-queue view of a rocprofv3 kernel trace of graph replays: how much of the side queues' kernel
time overlaps kernels of the busiest queue, and a short interleaved timeline."""
import csv, sys, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r.get("Stream_Id") == "0"] or rows   # graph replays are attributed to stream 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
c = collections.Counter(r["Queue_Id"] for r in rows)
print("dispatches per queue:", dict(c))
mainq = c.most_common(1)[0][0]
main = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Queue_Id"] == mainq]
starts = [s for s, _ in main]
tot = ov = 0
for r in rows:
    if r["Queue_Id"] == mainq:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    i = max(bisect.bisect_left(starts, s) - 1, 0)
    while i < len(main) and main[i][0] < e:
        ov += max(0, min(e, main[i][1]) - max(s, main[i][0]))
        i += 1
print("side-queue kernel time %.3f ms, of which overlapped with queue %s kernels: %.3f ms (%.0f%%)" % (tot / 1e6, mainq, ov / 1e6, 100.0 * ov / max(tot, 1)))
t0 = int(rows[len(rows) // 2]["Start_Timestamp"])
print("timeline sample (us from t0):")
for r in rows[len(rows) // 2: len(rows) // 2 + int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2 + 60]:
    print("q%s %9.1f -> %9.1f  %s" % (r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:50]))
