cd $GRAFT_REPO_ROOT
O=gpurun_out/sort_ext.txt; : > $O
CAP_SORT_EXT=3 timeout -k 10 300 python -m pytest "tests/test_sponza_class_gpu.py::test_small_scale_parity" tests/test_fallback_kernels_gpu.py -x -q -m gpu 2>&1 | tail -2 >> $O
for v in 0 1 2 3; do
  echo "== CAP_SORT_EXT=$v" >> $O
  CAP_SORT_EXT=$v timeout -k 10 300 python tools/hall_stages.py 8 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
  CAP_SORT_EXT=$v timeout -k 10 300 python tools/hall_stages.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
done
cat $O
