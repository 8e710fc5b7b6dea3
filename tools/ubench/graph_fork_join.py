"""What a fork / join costs INSIDE a captured hipGraph on MI355X: a dependent chain of small kernels, replayed from
a graph, against the same chain in which every step forks one more small kernel onto a second stream and joins it again
(event record / wait during capture = DAG edges).  usage: python tools/ubench/graph_fork_join.py"""
import sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
D = 'cuda:0'
REP = 40
x, y, z = torch.zeros(2048, 512, device=D), torch.zeros(2048, 512, device=D), torch.zeros(2048, 512, device=D)
side = torch.cuda.Stream()


def timed(body, n=10):
    body()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        for _ in range(REP):
            body()
    torch.cuda.current_stream().wait_stream(cap)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * REP) * 1e3


def chain():
    ops.action_activation(x, y, 1.0, True)
    ops.action_activation(y, x, 1.0, True)


def forked():
    cur = torch.cuda.current_stream()
    ops.action_activation(x, y, 1.0, True)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ops.action_activation(y, z, 1.0, True)          # independent of the next kernel of the chain
    ops.action_activation(y, x, 1.0, True)
    cur.wait_stream(side)


t0, t1 = timed(chain), timed(forked)
print(f"two dependent 4 MB kernels per step: {t0:.1f} us; + one forked kernel and its join: {t1:.1f} us per step "
      f"(the forked kernel alone is ~{t0 / 2:.1f} us of work that could overlap)")
