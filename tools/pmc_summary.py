"""Summarise a rocprofv3 counter_collection.csv: per kernel, dispatch count and the sum / mean of
every counter.  Usage: python tools/pmc_summary.py <counter_collection.csv> [top_n]"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n).split("(")[0][:70]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[n][r["Counter_Name"]] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = collections.defaultdict(float)
for n in agg:
    for c, v in agg[n].items():
        tot[c] += v
for c, v in tot.items():
    print("TOTAL %-14s %.6e" % (c, v))
for n in sorted(agg, key=lambda k: -sum(agg[k].values()))[:top]:
    for c, v in agg[n].items():
        print("%-72s %-12s n=%6d sum=%.4e mean=%.4e" % (n, c, cnt[n][c], v, v / cnt[n][c]))
