"""Per-(kernel, grid) launch statistics from a rocprofv3 --kernel-trace CSV: the same kernel name is launched at different
sizes (the rollout forward at 4096 clouds, the training forward at 2048) and next to other work (two-stream runs), so
the per-name averages of --stats do not line up with bench.py's HIP-event mean of the timed launches; these do.
usage: python tools/trace_summary.py <kernel_trace.csv> <out.csv> [top_n]"""
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault((name, grid, wg), []).append(d)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "work_groups", "threads_per_group", "launches", "total_ms", "mean_us", "median_us", "min_us", "max_us"])
    for (name, grid, wg), ds in rows:
        ds = sorted(ds)
        w.writerow([name, grid, wg, len(ds), round(sum(ds) / 1e3, 3), round(sum(ds) / len(ds), 1), round(ds[len(ds) // 2], 1),
                    round(ds[0], 1), round(ds[-1], 1)])
