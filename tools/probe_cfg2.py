"""Is the parameter-space distance between the HIP learner and the fp32 CPU oracle after cfg 2's 1280 + 1280 Adam steps a
defect or the amplification of fp32 round-off?  Third party: the SAME oracle in fp64.  Prints, per tensor,
||a - b|| / ||b - init|| for (hip, o32), (hip, o64), (o32, o64).    python tools/probe_cfg2.py [n_updates] [O] [lr]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ref_cpu as R
from tests.golden import cases
from tests.golden.detgen import det_normal
from tests.helpers import t, flat_state, FakeEnv, FakeLogger, per_tensor_update_error
from tests.test_gpu_fullsize import _cfg, _rollout_from_policy, _fill
DEV = "cuda:0"
n_up = int(sys.argv[1]) if len(sys.argv) > 1 else 5
O = int(sys.argv[2]) if len(sys.argv) > 2 else 53
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5
N, T, A = 4096, 128, 10
net = dict(name="MLP", hid_dim=[512, 512, 512], activation="tanh")
sd = cases.actor_critic_state(net, O, A, 0.5, 831)
cfg = _cfg(net, N, T, 8, n_up, lr, "cpu")
torch.set_num_threads(min(os.cpu_count() or 1, 32))
p32 = {k: t(v.copy()) for k, v in sd.items()}
obs = t(det_normal((T, N, O), 8310))
st = _rollout_from_policy(p32, cfg["model"], obs, 8311)
ret, adv = R.gae_returns(st["rewards"], st["values"], st["dones"], st["succs"], st["last_values"], 0.99, 0.95, None, False)
roll = {k: st[k] for k in ("observations", "actions", "values", "actions_log_prob", "mu", "sigma")}
roll["returns"], roll["advantages"] = ret, adv
t0 = time.time(); o32 = R.ppo_update(p32, roll, cfg, 1); print("oracle fp32 s", time.time() - t0, flush=True)
p64 = {k: t(v.copy()).double() for k, v in sd.items()}
roll64 = {k: v.double() for k, v in roll.items()}
t0 = time.time(); o64 = R.ppo_update(p64, roll64, cfg, 1); print("oracle fp64 s", time.time() - t0, flush=True)
from partmanip_amd.algorithms import ppo
res = {}
for tag, kw in (("default", {}), ("nographs", dict(use_graphs=False)), ("nofused", dict(fused_head=False)), ("plain", dict(use_graphs=False, solo_group=False, fused_head=False))):
    with tempfile.TemporaryDirectory() as d:
        run = ppo(FakeEnv(N, {"normal_state": O}, A), _cfg(net, N, T, 8, n_up, lr, DEV), FakeLogger(d))
    run.actor_critic.load_state_dict({k: t(v.copy()) for k, v in sd.items()})
    for k, v in kw.items():
        setattr(run, k, v)
    _fill(run, st)
    run.log_dict = {}; run.curr_iter = 1
    run.learn(st["last_values"].to(DEV)); torch.cuda.synchronize()
    res[tag] = (flat_state(run.actor_critic.state_dict()), dict(run.log_dict))
    del run
f32, f64 = flat_state(p32), np.concatenate([np.asarray(v).reshape(-1) for v in p64.values()])
def table(a, b, name):
    e = per_tensor_update_error(a, b, sd)
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    print(f"--- {name}: q99.9 {np.quantile(d, 0.999) / lr:.3e} lr, max {d.max() / lr:.3e} lr, median {np.median(d) / lr:.3e} lr")
    print("   " + "  ".join(f"{k.replace('model.', '')}:{v[0]:.1e}" for k, v in e.items()))
table(f32, f64, "oracle32 vs oracle64")
for tag, (fl, log) in res.items():
    table(fl, f64, f"hip[{tag}] vs oracle64")
    table(fl, f32, f"hip[{tag}] vs oracle32")
a, b = res["default"][0], res["nographs"][0]
print("default == nographs bitwise:", np.array_equal(a, b), " default == nofused:", np.array_equal(a, res["nofused"][0]),
      " default vs plain max diff / lr:", np.abs(a.astype(np.float64) - res["plain"][0]).max() / lr)
for k in ("Train/surrogate_loss", "Train/kl", "Train/value_function_loss"):
    print(k, "o32", o32["log"][k], "o64", o64["log"][k], "hip", float(res["default"][1][k]))
