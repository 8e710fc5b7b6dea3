"""A/B build of the library: `python tools/build_ab.py NAME SOURCE.hip -DFLAG[=v] ...` compiles ONE source of csrc/ with extra
flags and links it with the cached objects of the default build into gpurun_ab/NAME.so (select it with PARTMANIP_HIP_LIB)."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partmanip_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
root = os.path.join(os.path.dirname(B.HERE), "gpurun_ab")
os.makedirs(os.path.join(root, "obj"), exist_ok=True)
obj = os.path.join(root, "obj", f"{name}.o")
subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj] + B.FLAGS + flags, check=True)
objs = [obj if s == src else os.path.join(B.LIBDIR, "obj", s.replace(".hip", ".o")) for s in B.SOURCES]
out = os.path.join(root, f"{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs, check=True)
print(out)
