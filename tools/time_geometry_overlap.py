import sys, torch, time
sys.path.insert(0, '.')
from partmanip_amd.algo_utils import ActorCritic
from partmanip_amd.feeder import FeederEnv
from partmanip_amd import ops
DEV='cuda:0'; B=2048; P=4096
env = FeederEnv(B, {"depth_sparse": 4 * P, "proprio_state": 0}, 10, DEV, seed=7, point_num=P)
x = env.reset()["depth_sparse"]
net = dict(name="SparseUNet", activation="tanh", point_num=P, grid=50)
torch.manual_seed(0)
ac = ActorCritic(4 * P, 10, dict(action_std=0.1, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
ac.flat()
dy = torch.randn(B, 10, device=DEV)
def ev(): e=torch.cuda.Event(enable_timing=True); e.record(); return e
for _ in range(2):
    ac.actor.hip_forward(x); ac.actor.hip_backward(dy)
torch.cuda.synchronize()
a=ev(); 
for _ in range(3): g=ac.actor.geometry(x)
b=ev(); torch.cuda.synchronize(); print("geometry ms", a.elapsed_time(b)/3)
a=ev()
for _ in range(3): ac.actor.hip_forward(x)
b=ev(); torch.cuda.synchronize(); print("fwd inline ms", a.elapsed_time(b)/3)
gs=[ac.actor.geometry(x) for _ in range(3)]
torch.cuda.synchronize()
a=ev()
for g in gs:
    ac.actor.take_geometry(g); ac.actor.hip_forward(x)
b=ev(); torch.cuda.synchronize(); print("fwd with tables ms", a.elapsed_time(b)/3)
# overlapped: geometry on side stream while fwd+bwd on main
side=torch.cuda.Stream()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(3):
    ac.actor.take_geometry(gs[0]); ac.actor.hip_forward(x); ac.actor.hip_backward(dy)
torch.cuda.synchronize(); t1=time.perf_counter(); print("fwd+bwd (tables ready) ms", (t1-t0)/3*1e3)
t0=time.perf_counter()
for _ in range(3):
    ac.actor.take_geometry(gs[0]); ac.actor.hip_forward(x); ac.actor.hip_backward(dy)
    with torch.cuda.stream(side):
        g2=ac.actor.geometry(x)
torch.cuda.synchronize(); t1=time.perf_counter(); print("fwd+bwd with geometry on side stream ms", (t1-t0)/3*1e3)
t0=time.perf_counter()
for _ in range(3):
    ac.actor.hip_forward(x); ac.actor.hip_backward(dy)
torch.cuda.synchronize(); t1=time.perf_counter(); print("fwd+bwd inline geometry ms", (t1-t0)/3*1e3)
