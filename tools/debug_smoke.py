import ctypes, os, sys
sys.path.insert(0, '.')
hip = ctypes.CDLL("libamdhip64.so")
def last(tag):
    print(tag, "hipPeekAtLastError =", hip.hipPeekAtLastError(), flush=True)
import torch
last("after import torch")
os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
from partmanip_amd.build import build as build_lib
last("after import build")
lib = build_lib()
last("after build_lib()")
import partmanip_amd
from partmanip_amd import _lib
last("after import package")
print(_lib.lib.pm_version())
from partmanip_amd.algorithms import ppo, dagger
last("after import algorithms")
print(torch.cuda.is_available())
last("after cuda.is_available")
x = torch.zeros(4, device="cuda:0")
last("after first tensor")
