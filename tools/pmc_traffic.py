"""HBM traffic of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, in
separate runs: they do not fit one pass on gfx950).  Units and corrections per
MI355X_MICROARCH.md "HBM": both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide coalesced reads (128-B requests tallied at 64 B), so it is doubled; WRITE_SIZE is
exact.  The number of training steps the profiled command executed is counted from the trace itself:
crit_tail_kernel (the criterion's finishing workgroup) runs exactly once per step (eager warm-up, capture, replays and the PCIe-inclusive
leg of bench.py all included); the 4th argument is only the fallback when that kernel is absent.
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps] [bench.json] [tag]
(bench.json of the same round: its roofline.launches_per_step is recorded so bench.py can tell a stale file)"""
import csv, json, sys
csv.field_size_limit(1 << 30)


ALL = {}


def gemm_sum(path, counter):
    tot, n, steps, every = 0.0, 0, 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        every += float(r["Counter_Value"])
        if "gemm_" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
        elif "crit_tail_kernel" in r["Kernel_Name"]:
            steps += 1
    ALL[counter] = every
    return tot, n, steps


f, nf, sf = gemm_sum(sys.argv[1], "FETCH_SIZE")
w, nw, sw = gemm_sum(sys.argv[2], "WRITE_SIZE")
assert sf == sw, (sf, sw)
steps = sf if sf > 0 else int(sys.argv[4])
out = {
    "steps_profiled": steps, "gemm_kernel_launches_per_step": nf / steps,
    "hbm_bytes_per_step": (2.0 * f + w) * 1024.0 / steps,
    "kernels": "every kernel whose name contains gemm_ (gemm_wstage / gemm_lds64 / gemm_frag / gemm_f32)",
    "launches_fetch_pass": nf, "launches_write_pass": nw,
    "fetch_kib_raw_per_launch": f / max(nf, 1), "write_kib_per_launch": w / max(nw, 1),
    "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE x1, KiB -> bytes",
    "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
    # every kernel of the step (same correction): the whole-step HBM-side traffic next to SURVEY 8d's algorithmic bytes
    "step_hbm_bytes_all_kernels": (2.0 * ALL["FETCH_SIZE"] + ALL["WRITE_SIZE"]) * 1024.0 / steps,
}
if len(sys.argv) > 5:
    try:
        line = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1])
        out["launch_calls_per_step"] = line["roofline"]["launches_per_step"]
    except Exception as e:  # noqa: BLE001
        out["launch_calls_per_step"] = None
if len(sys.argv) > 6:
    out["profile"] = sys.argv[6]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
