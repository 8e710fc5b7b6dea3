"""In-kernel timeline of gemm2_kernel (A/B build with -DG2_PROFILE: tools/build_ab.sh g2prof gemm2_f32.hip -DG2_PROFILE;
PARTMANIP_HIP_LIB=gpurun_ab/g2prof.so python tools/g2_profile.py [MxKxN ...]).  Four s_memtime stamps per work-group:
entry, after the prologue, after the K loop, after the epilogue (100 MHz constant clock -> 10 ns ticks)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from partmanip_amd import ops
from partmanip_amd._lib import lib
DEV = 'cuda:0'
lib.pm_debug_set_gemm_prof.argtypes = [ctypes.c_void_p]
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(2048, 512, 512)]
for M, K, N in shapes:
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV)
    prof = torch.zeros(65536 * 4 * 4, dtype=torch.int64, device=DEV)
    for _ in range(3):
        ops.linear_fwd(x, w, b, y, ops.ACT_TANH)
    torch.cuda.synchronize()
    lib.pm_debug_set_gemm_prof(prof.data_ptr())
    ops.linear_fwd(x, w, b, y, ops.ACT_TANH)
    torch.cuda.synchronize()
    lib.pm_debug_set_gemm_prof(None)
    ks = prof[65536 * 8:].view(-1, 8)[:, :5].cpu().double()
    ks = ks[(ks > 0).all(1)]
    if ks.shape[0]:
        d = (ks[:, 1:] - ks[:, :-1]).mean(0)
        print('   K-step 4 of wave 0 (cycles between the 5 stamps): %.0f | %.0f | %.0f | %.0f' % tuple(d.tolist()))
    p = prof[:65536 * 4].view(-1, 4).cpu()
    p = p[p[:, 0] != 0].double()
    p = p[(p > 0).all(1)]
    t0 = p[:, 0].min()
    tick = 1.0 / 2.4                                      # ns per tick: __builtin_readcyclecounter counts shader cycles (~2.4 GHz)
    f = lambda v: f"{float(v) * tick / 1e3:7.2f}"
    print(f"{M}x{K}x{N}: {p.shape[0]} work-groups; start skew (last entry - first) {f(p[:, 0].max() - t0)} us; "
          f"prologue {f((p[:, 1] - p[:, 0]).mean())} us, K loop {f((p[:, 2] - p[:, 1]).mean())} us, "
          f"epilogue {f((p[:, 3] - p[:, 2]).mean())} us; first entry -> last exit {f(p[:, 3].max() - t0)} us")
