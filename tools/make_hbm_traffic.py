#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the PMC summaries of a profile set (tools/profile_round.sh -> tools/pmc_run.sh ->
tools/pmc_summary.py).  FETCH_SIZE / WRITE_SIZE are KB per launch; FETCH_SIZE is doubled (gfx950 tallies the 128-byte
requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM section).  GRBM_GUI_ACTIVE sums the 8 XCDs;
SQ_VALU_MFMA_BUSY_CYCLES sums the issued MFMAs' pipe cycles over the 1024 SIMDs.

    python tools/make_hbm_traffic.py <tag> <pmc_enc summary.json> [<pmc_lin summary.json> [<pmc_su summary.json> [<pmc_fps summary.json>]]]
                                     [sa=<summary of tools/time_sa.py>] [su2048=<summary of tools/time_sparse_unet.py 2048>]
                                     [state=<summary of bench.py --workload state --lean --steps 2 --warmup 0>]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("-"))
sys.argv = [a for a in sys.argv if "=" not in a]
tag = sys.argv[1]
enc = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None          # (only fps=... given: update the FPS keys alone)
lin = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else None
su = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else None
fps = json.load(open(sys.argv[5])) if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else None
if "fps" in extra and os.path.exists(extra["fps"]):
    fps = json.load(open(extra["fps"]))
dst = os.path.join(ROOT, "profiles", "hbm_traffic.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}


def find(tab, prefix):
    ks = [k for k in tab if k.split("<")[0].replace("void ", "") == prefix]
    if not ks:
        raise KeyError(prefix)
    return tab[max(ks, key=lambda k: tab[k].get("GRBM_GUI_ACTIVE", {}).get("mean", 0.0))]


def traffic(k):
    return 2.0 * k["FETCH_SIZE"]["mean"] * 1024.0, k["WRITE_SIZE"]["mean"] * 1024.0


def busy(k):
    return k["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0 / (k["GRBM_GUI_ACTIVE"]["mean"] / 8.0)


if enc is not None:
    out["source"] = (f"tools/profile_round.sh {tag} -> tools/pmc_run.sh gpurun_out/{tag}/pmc_enc python tools/time_enc.py (separate "
                     f"rocprofv3 --pmc passes, counters only) -> profiles/round{tag[1]}_{tag[2:]}_pmc_encoder.json; tools/make_hbm_traffic.py")
    out["correction"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies 128-B requests at 64 B); counters are KB"
    for name in ("pn_fwd_kernel", "pn_bwd16_kernel", "pn_bwd_prep_kernel"):
        k = find(enc, name)
        f, w = traffic(k)
        out[f"{name}_fetch_bytes_raw"] = f / 2.0
        out[f"{name}_write_bytes"] = w
        out[f"{name}_bytes_per_launch"] = f + w
        if name != "pn_bwd_prep_kernel":
            out[f"{name}_mfma_busy"] = busy(k)
            out[f"{name}_cycles_per_launch"] = k["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            out[f"{name}_lds_bank_conflict_fraction"] = k["SQ_LDS_BANK_CONFLICT"]["mean"] / max(k["SQ_LDS_IDX_ACTIVE"]["mean"], 1.0)
    side = sum(find(enc, n)["GRBM_GUI_ACTIVE"]["mean"] / 8.0 for n in ("pn_bwd_prep_kernel", "pn_dw3_gather_kernel", "pn_dw3_finish_kernel",
                                                                         "pn_bwd_reduce1_kernel", "pn_bwd_reduce2_kernel"))
    k16 = find(enc, "pn_bwd16_kernel")
    out["pn_bwd16_kernel_mfma_busy_over_backward_call"] = (k16["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0) / (k16["GRBM_GUI_ACTIVE"]["mean"] / 8.0 + side)
if lin is not None:
    for key, kern in (("fwd", "gemm2_dma_kernel<false, false, 1, 1, false, 0>"), ("bwd_data", "gemm2_dma_kernel<false, true, 1, 1, false, 0>"),
                      ("bwd_weight", "gemm2_dma_kernel<true, true, 1, 1, false, 0>")):
        k = lin.get("void " + kern) or lin.get(kern)
        if k is None:
            continue
        f, w = traffic(k)
        out[f"linear_2048x512x512_{key}_bytes_per_launch"] = f + w
        out[f"linear_2048x512x512_{key}_mfma_insts"] = k["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 64.0
    out["linear_2048x512x512_algorithmic_bytes"] = (2 * 2048 * 512 + 512 * 512) * 4
    out["linear_2048x512x512_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_lin python tools/time_gemm.py 2048x512x512 -> "
                                       f"profiles/round{tag[1]}_{tag[2:]}_pmc_linear.json; FETCH doubled (gfx950 correction); each of the 8 XCDs fetches "
                                       "the whole weight matrix into its own L2")
if su is not None:
    f = sum(2.0 * v["FETCH_SIZE"]["mean"] * v["FETCH_SIZE"]["launches"] for v in su.values() if "FETCH_SIZE" in v) * 1024.0
    w = sum(v["WRITE_SIZE"]["mean"] * v["WRITE_SIZE"]["launches"] for v in su.values() if "WRITE_SIZE" in v) * 1024.0
    out["sparse_unet_256_clouds_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_su python tools/time_sparse_unet.py 256 (5 forward + backward "
                                          "passes, ALL kernels of the process): bytes below are per forward + backward pass")
    out["sparse_unet_256_clouds_fetch_bytes_per_pass"] = f / 5.0
    out["sparse_unet_256_clouds_write_bytes_per_pass"] = w / 5.0
if fps is not None:
    k = find(fps, "fps_multi_kernel")
    f, w = traffic(k)
    out["fps_multi_kernel_bytes_per_launch"] = f + w
    out["fps_multi_kernel_fetch_bytes_raw"] = f / 2.0
    out["fps_multi_kernel_cycles_per_launch"] = k["GRBM_GUI_ACTIVE"]["mean"] / 8.0
    out["fps_multi_kernel_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_fps python bench.py --workload depth2pc --no-cpu-baseline: the "
                                    "multi-work-group FPS launch of depth2pc (64 envs x 6 views x 180 x 320), FETCH doubled")
    if "SQ_LDS_BANK_CONFLICT" in k:
        out["fps_multi_kernel_lds_bank_conflict_fraction"] = k["SQ_LDS_BANK_CONFLICT"]["mean"] / max(k["SQ_LDS_IDX_ACTIVE"]["mean"], 1.0)


def total(tab, passes):
    """HBM bytes of ALL kernels of a command / the number of passes it ran (FETCH doubled)."""
    f = sum(2.0 * v["FETCH_SIZE"]["mean"] * v["FETCH_SIZE"]["launches"] for v in tab.values() if "FETCH_SIZE" in v) * 1024.0
    w = sum(v["WRITE_SIZE"]["mean"] * v["WRITE_SIZE"]["launches"] for v in tab.values() if "WRITE_SIZE" in v) * 1024.0
    return f / passes, w / passes


if "sa" in extra and os.path.exists(extra["sa"]):
    sa = json.load(open(extra["sa"]))
    for kname, v in sa.items():
        base = kname.replace("void ", "").split("<")[0]
        if base not in ("sa_fwd_pk_kernel", "sa_bwd_pk_kernel", "sa_fwd_kernel", "sa_bwd_kernel") or "FETCH_SIZE" not in v:
            continue
        dims = [x.strip() for x in kname.split("<")[1].split(",")[:3]]
        key = f"sa_{'fwd' if 'fwd' in base else 'bwd'}_{dims[0]}x{dims[1]}x{dims[2]}"
        f, w = traffic(v)
        out[f"{key}_bytes_per_launch"] = f + w
        out[f"{key}_kernel"] = base
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            out[f"{key}_mfma_busy"] = busy(v)
    out["sa_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_sa python tools/time_sa.py (2048 clouds, the distinct-row kernels): HBM bytes "
                      "per launch, FETCH doubled")
if "ga" in extra and os.path.exists(extra["ga"]):
    ga = json.load(open(extra["ga"]))
    for kname, v in ga.items():
        base = kname.replace("void ", "").split("<")[0].split("(")[0]
        if base in ("ga_fwd_kernel", "ga_bwd_dh_kernel", "ga_dw_gather_kernel", "ga_dw_finish_kernel") and "FETCH_SIZE" in v:
            f, w = traffic(v)
            out[f"{base}_bytes_per_launch"] = f + w
            if base == "ga_fwd_kernel" and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                out["ga_fwd_kernel_mfma_busy"] = busy(v)
    out["ga_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_ga python tools/time_groupall.py (2048 clouds x 64 rows, 256 -> 512): HBM bytes per "
                      "launch, FETCH doubled")
if "su2048" in extra and os.path.exists(extra["su2048"]):
    f, w = total(json.load(open(extra["su2048"])), 5.0)
    out["sparse_unet_2048_clouds_fetch_bytes_per_pass"] = f
    out["sparse_unet_2048_clouds_write_bytes_per_pass"] = w
    out["sparse_unet_2048_clouds_bytes_per_pass"] = f + w
    out["sparse_unet_2048_clouds_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_su2048 python tools/time_sparse_unet.py 2048 (5 forward + "
                                           "backward passes of a 2048-cloud mini-batch, ALL kernels of the process incl. the voxel tables)")
if "state" in extra and os.path.exists(extra["state"]):
    f, w = total(json.load(open(extra["state"])), 2.0)
    out["state_iteration_bytes"] = f + w
    out["state_iteration_fetch_bytes"] = f
    out["state_iteration_write_bytes"] = w
    out["state_iteration_note"] = (f"tools/pmc_run.sh gpurun_out/{tag}/pmc_state env PARTMANIP_GRAPHS=0 python bench.py --workload state --lean "
                                   "--no-cpu-baseline --steps 2 --warmup 0: ALL kernels of the process (two iterations + the rollout "
                                   "forward), divided by 2")
json.dump(out, open(dst, "w"), indent=1)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if not k.endswith("note") and k not in ("source", "correction")})
