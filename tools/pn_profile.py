"""In-kernel timeline of pn_bwd16_kernel (A/B build: tools/build_ab.sh pnprof pointnet_enc.hip -DPN_PROFILE;
PARTMANIP_HIP_LIB=gpurun_ab/pnprof.so python tools/pn_profile.py).  Seven s_memtime stamps per wave and tile for work-groups
0-3 (first cloud, tiles 0-15): loop top | after valu_issue | after the dW2 loop | after the dh1 loop | after the MFMA stage's
epilogue | after valu_finish | after the barrier.  The stamps themselves cost (s_memtime + lgkmcnt(0)); read the shares."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from partmanip_amd.algo_utils import ActorCritic
from partmanip_amd._lib import lib
DEV = 'cuda:0'
import os
BF6 = os.environ.get("PN_PRECISION_BWD", "f32") == "bf16x6"       # the split-bf16 backward: 8 waves, 8 stamps (see its PN_STAMPs)
net = dict(name="PointNet", activation="tanh", max_mean=True, sub_mean=False, precision="f32", save_h2=True,
           precision_bwd="bf16x6" if BF6 else "f32")
torch.manual_seed(0)
ac = ActorCritic(3072, 10, dict(action_std=0.5, action_activate="tanh", clipAction=1.0, network=net)).to(DEV)
ac.flat()
B = 2048
x = (torch.rand(B, 1024, 3, device=DEV) * 2 - 1).reshape(B, -1).contiguous()
dy = torch.randn(B, 10, device=DEV)
for _ in range(3):
    ac.actor.hip_forward(x)
    ac.actor.hip_backward(dy)
torch.cuda.synchronize()
buf = torch.zeros(4 * 16 * 16 * 8, dtype=torch.int64)
lib.pm_debug_pn_prof_read.argtypes = [ctypes.c_void_p]
assert lib.pm_debug_pn_prof_read(buf.data_ptr()) == 0
t = buf.view(4, 16, 16, 8).double()          # [wg][tile][wave][stamp]
names = ["valu_issue", "dW2 loop", "dh1 loop", "mfma epilogue", "valu_finish", "barrier wait"]
NWV, NST = 16, 7
if BF6:
    names = ["valu_issue", "dW2 loop", "dh1 loop", "barrier 1 wait", "dh1 epilogue", "valu_finish", "barrier 2 wait"]
    NWV, NST = 8, 8
    t = t[:, :, :8]
for wg in range(4):
    s = t[wg, 2:15]                          # steady-state tiles
    d = torch.stack([s[..., i + 1] - s[..., i] for i in range(NST - 1)], -1)      # [tile][wave][NST-1]
    tile_len = (t[wg, 3:15, :, 0] - t[wg, 2:14, :, 0]).mean()
    print(f"wg {wg}: tile period {tile_len:.0f} cycles")
    for i, n in enumerate(names):
        print(f"   {n:14s} mean {d[..., i].mean():7.0f}   min-wave {d[..., i].mean(0).min():7.0f}   max-wave {d[..., i].mean(0).max():7.0f}"
              f"   max over waves per tile (mean) {d[..., i].max(1).values.mean():7.0f}")
    # when does the LAST wave reach the barrier relative to the first wave's loop top
    first = s[..., 0].min(1).values
    print("   first loop top -> last wave at barrier: %.0f; -> first wave at barrier: %.0f" %
          ((s[..., NST - 2].max(1).values - first).mean(), (s[..., NST - 2].min(1).values - first).mean()))
wg0 = t[0, 5]
print("wg 0, tile 5, per wave (relative to the earliest stamp):")
base = wg0[:, 0].min()
for w in range(NWV):
    print("   wave %2d: " % w + " ".join("%7.0f" % (wg0[w, i] - base) for i in range(NST)))
