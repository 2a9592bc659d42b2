cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_node_v2.txt
: > $O
for rep in 1 2; do
for v in "" v1; do
  echo "== variant '${v}' rep $rep" >> $O
  export CAP_LIB_VARIANT=$v; [ -z "$v" ] && unset CAP_LIB_VARIANT
  if [ $rep = 1 ]; then timeout -k 10 300 python -m pytest "tests/test_sponza_class_gpu.py::test_small_scale_parity" -x -q -m gpu 2>&1 | tail -1 >> $O; fi
  timeout -k 10 300 python tools/hall_stages.py 8 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
  timeout -k 10 300 python tools/hall_stages.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
done; done
cat $O
