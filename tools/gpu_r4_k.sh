cd "$GRAFT_REPO_ROOT"
echo "--- A: -c build(); smoke()"; python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | cut -c1-100
echo "--- B: main smoke"; python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-100
echo "--- C: HIP_VISIBLE env"; env | grep -i -E "HIP|ROCR|CUDA_VIS|HSA" 
echo "--- D: -c with PYTORCH_ROCM_ARCH preset + smoke only"; PYTORCH_ROCM_ARCH=gfx950 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-100
echo "--- E: build only then subprocess"; python -c "
import __graft_entry__ as g, subprocess, sys
g.build()
import torch
print('cuda', torch.cuda.is_available(), torch.cuda.device_count())
from partmanip_amd import ops
st=torch.zeros(8,4,1,device='cuda:0'); d=torch.zeros(8,4,1,dtype=torch.bool,device='cuda:0'); last=torch.zeros(4,1,device='cuda:0')
ops.gae_scan(st,st,d,d,last,st.clone(),st.clone(),0.99,0.95,None); torch.cuda.synchronize(); print('gae ok')
" 2>&1 | tail -3 | cut -c1-150
