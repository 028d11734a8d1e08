"""MFMA utilisation per kernel from a rocprofv3 pass with SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES (per-dispatch sums
over the chip): util = MFMA-busy cycles / CU-busy cycles.  Unit caveat (MI355X_MICROARCH.md, cycle constants): the
MFMA counter counts cycles, the SQ busy counters count quad-cycles on gfx950 -- the ratio printed applies the factor 4
(util = mfma / (4 * busy)); the raw sums are printed next to it.  usage: pmc_mfma.py <counter_collection.csv> [top_n]"""
import collections, csv, re, sys
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"^void ", "", k).split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        n[k] += 1
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tm = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for a in agg.values()); tb = sum(a.get("SQ_BUSY_CU_CYCLES", 0) for a in agg.values())
print("ALL KERNELS: MFMA busy %.4e cycles, CU busy %.4e quad-cycles -> MFMA utilisation %.1f %%" % (tm, tb, 100 * tm / max(4 * tb, 1)))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CU_CYCLES", 0))[:top]:
    m, b = agg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), agg[k].get("SQ_BUSY_CU_CYCLES", 0.0)
    print("%-62s n=%5d  mfma %.3e  cu_busy %.3e  util %5.1f %%" % (k, n[k], m, b, 100 * m / max(4 * b, 1)))
