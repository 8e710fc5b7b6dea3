cd "$GRAFT_REPO_ROOT"
for v in main sa_ab1 sa_ab2 sa_ab4 sa_ab8 sa_ab16 sa_ab32 sa_ab64 main; do
  if [ $v == main ]; then unset PARTMANIP_HIP_LIB; else export PARTMANIP_HIP_LIB=gpurun_ab/$v.so; fi
  echo "== $v"; python tools/time_sa.py 2>&1 | grep "fwd"
done
