"""Which Python lines of the product path launch ATen kernels (copies, fills, sorts ...)?  One learner step under torch.profiler with
stacks; prints device time per (op, innermost partmanip_amd frame).  usage: python tools/aten_sites.py vision_pn2 | sparse_unet"""
import sys
import collections
import torch
sys.path.insert(0, '.')
import bench                                                                        # noqa: E402
from torch.profiler import profile, ProfilerActivity                                # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "vision_pn2"
dev = "cuda:0"
if what == "sparse_unet":
    import argparse
    from partmanip_amd.algorithms import dagger as _dagger
    runs, orig_update = [], _dagger.update

    def spy(self, it):                                  # bench.run_dagger builds the runner and warms it up: keep a handle on it
        if not runs:
            runs.append(self)
        return orig_update(self, it)
    _dagger.update = spy
    bench.run_dagger(argparse.Namespace(student="sparse_unet", points=4096, warmup=1, steps=1, no_cpu_baseline=True), dev, 0, 1)
    _dagger.update = orig_update
    run = runs[0]
    step = lambda: run.update(1)
else:
    w = dict(bench.WORKLOADS[what])
    run, ac, st, last_values, step = bench.build_runner(w, bench.make_cfg(w, dev), dev, 0)
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    site = "?"
    for fr in ev.stack or []:
        if "partmanip_amd" in fr or "bench.py" in fr:
            site = fr.strip()
            break
    a = agg[(ev.name, site)]
    a[0] += ev.self_device_time_total
    a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"ATen device time in one step: {tot / 1e3:.2f} ms")
for (name, site), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{us / 1e3:9.3f} ms {n:6d}x  {name:28s} {site[-110:]}")
