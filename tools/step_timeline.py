"""One captured step as a list of dispatches: from a rocprofv3 kernel_trace.csv of `bench.py --no-extras --no-roofline`,
print the dispatches of the LAST graph replay in order (start offset, duration, gap to the previous end, grid, kernel).
Usage: python tools/step_timeline.py <kernel_trace.csv> [dispatches_per_step]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay: find the period by locating the last occurrence of the first kernel of a step (the counter add)
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not per:
    # period = distance between the last two occurrences of the sine_pos kernel (once per step)
    idx = [i for i, x in enumerate(names) if "sine_pos_kernel" in x]
    per = idx[-1] - idx[-2]
    start = idx[-2]
else:
    start = n - per
seg = rows[start - 8:start - 8 + per]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    short = re.sub(r"^void ", "", short).split("(")[0][:60]
    grid = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    print("%9.1f us  dur %7.2f  gap %6.2f  wg %5d x %-4s z%-2s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, grid,
          r["Workgroup_Size_X"], r["Grid_Size_Z"], short))
    prev_end = e
    tot += e - s
print("dispatches %d, sum of durations %.1f us, span %.1f us" % (len(seg), tot / 1e3, (prev_end - t0) / 1e3))
