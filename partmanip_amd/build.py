"""Build recipe for libpartmanip_hip.so (gfx950 only).  `python -m partmanip_amd.build`.

hipcc cross-compiles without a GPU; the .so is written IN-TREE (partmanip_amd/lib/) so it
travels with the repo snapshot to the GPU box.  Objects are cached per source by mtime.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpartmanip_hip.so")
SOURCES = ["gae.hip", "losses.hip", "adam.hip", "gemm_f32.hip", "gemm2_f32.hip", "pointnet_enc.hip", "pointnet_enc_bf3.hip", "pointnet_enc_bf6.hip", "pointops.hip", "sa_fused.hip", "sa_groupall.hip", "conv3d.hip", "sparse_voxel.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mfma_f32.h"), os.path.join(CSRC, "gemm2.h"), os.path.join(CSRC, "skinny.h"), os.path.join(CSRC, "pointnet_enc_bwd_bf6.h"),
            os.path.join(HERE, "..", "include", "partmanip_hip.h")]
    objs, rebuilt = [], False
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in hdrs):
            cmd = [hipcc, "-x", "hip", "-c", s, "-o", o] + FLAGS
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
