"""Turns a rocprofv3 --pmc pass of SQ counters over the BAL-scale profiling target into a per-kernel issue table:
where a wave's cycles go (parked on a wait or barrier / stalled at issue / issuing) and how many vector instructions
it executes.  The SQ cycle counters are in quad-cycles (x4 = shader cycles).

Usage: python scripts/summarize_sq.py <pmc directory> <out.json> [workload key]"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hipba::", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    out = {}
    print("%-28s %6s %9s %9s %7s %7s %7s %7s %9s" % ("kernel", "calls", "waves", "cyc/wave", "parked", "stalled", "issuing", "valu", "VALU/wave"))
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        waves, wc = c.get("SQ_WAVES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
        if waves <= 0 or wc <= 0:
            continue
        n = len(calls[name])
        e = {"calls": n, "waves_per_call": waves / n, "cycles_per_wave": 4.0 * wc / waves,
             "parked_frac": c.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
             "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "valu_frac": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
             "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0.0) / waves, "salu_insts_per_wave": c.get("SQ_INSTS_SALU", 0.0) / waves}
        out[name] = e
        print("%-28s %6d %9.0f %9.0f %6.1f%% %6.1f%% %6.1f%% %6.1f%% %9.0f" % (
            name[:28], n, e["waves_per_call"], e["cycles_per_wave"], 100 * e["parked_frac"], 100 * e["issue_stall_frac"],
            100 * e["issuing_frac"], 100 * e["valu_frac"], e["valu_insts_per_wave"]))
    # stamped with the kernel sources and the workload the counters belong to: bench.py ignores the file otherwise
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_fingerprint
    doc = {"csrc_sha256": csrc_fingerprint(), "workload": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": out}
    json.dump(doc, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
