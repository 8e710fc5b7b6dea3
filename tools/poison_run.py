"""Run pytest in-process after filling (and releasing to torch's caching allocator) a large part of HBM with a poison pattern:
a kernel that reads memory it never wrote then sees garbage instead of the zeros of a fresh allocation.
usage: python tools/poison_run.py <GB> <value> <pytest args...>"""
import sys, torch, pytest
gb, val = int(sys.argv[1]), float(sys.argv[2])
bufs = [torch.full((1 << 28,), val, device="cuda:0") for _ in range(gb)]      # 1 GiB each
torch.cuda.synchronize()
del bufs
sys.exit(pytest.main(sys.argv[3:]))
