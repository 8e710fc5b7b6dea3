"""Which Python lines of the product path call the tensor library (copies, fills, index ops ...)?  One learner step with the usual
suspects wrapped; prints bytes touched per (op, innermost partmanip_amd frame).  usage: python tools/aten_sites.py vision_pn2 | sparse_unet"""
import sys
import collections
import torch
sys.path.insert(0, '.')
import bench                                                                        # noqa: E402

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
# ---- who calls the tensor library?  torch.profiler gives no Python stacks on this stack, so the usual suspects are wrapped in Python
# and every call is charged to the innermost partmanip_amd / bench frame with the bytes it moves
import traceback
sites = collections.defaultdict(lambda: [0, 0])


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "partmanip_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{fr.filename.split('/repo/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def _wrap_method(name, nbytes):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        r = orig(self, *a, **k)
        if self.is_cuda:
            e = sites[(name, _site())]
            e[0] += nbytes(self, r)
            e[1] += 1
        return r
    setattr(torch.Tensor, name, w)
    return orig


def _wrap_fn(name):
    orig = getattr(torch, name)

    def w(*a, **k):
        r = orig(*a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda:
            e = sites[("torch." + name, _site())]
            e[0] += r.numel() * r.element_size()
            e[1] += 1
        return r
    setattr(torch, name, w)
    return orig


undo = [(torch.Tensor, n, _wrap_method(n, f)) for n, f in (
    ("copy_", lambda s_, r: s_.numel() * s_.element_size()), ("zero_", lambda s_, r: s_.numel() * s_.element_size()),
    ("fill_", lambda s_, r: s_.numel() * s_.element_size()), ("clone", lambda s_, r: s_.numel() * s_.element_size()),
    ("contiguous", lambda s_, r: 0 if r.data_ptr() == s_.data_ptr() else r.numel() * r.element_size()),
    ("index_select", lambda s_, r: r.numel() * r.element_size()), ("to", lambda s_, r: 0 if r is s_ else r.numel() * r.element_size()))]
undo += [(torch, n, _wrap_fn(n)) for n in ("zeros", "cat", "full", "arange", "ones", "stack", "zeros_like")]
step()
torch.cuda.synchronize()
for mod, n, o in undo:
    setattr(mod, n, o)
tot = sum(v[0] for v in sites.values())
print(f"tensor-library calls in one step: {sum(v[1] for v in sites.values())}, {tot / 1e6:.1f} MB touched")
for (name, site), (b, n) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{b / 1e6:10.2f} MB {n:6d}x  {name:18s} {site}")
