"""Aggregate rocprofv3 --pmc counter_collection CSVs (one directory per pass) into a per-kernel mean table."""
import collections, csv, glob, json, sys
out = {}
for d in sorted(glob.glob(sys.argv[1] + "/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        for (kern, ctr), (n, s) in agg.items():
            out.setdefault(kern, {})[ctr] = dict(launches=n, mean=s / n)
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for kern in out:
    if kern.startswith(("pn_", "void gemm", "clip_adam", "gae", "void sa_", "void fps")):
        print(kern, {c: round(v["mean"], 1) for c, v in out[kern].items()})
