set -o pipefail
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  CAP_LIB_VARIANT=packetv1 python tools/primary_ab.py 1.0 32 2>&1 | tail -1
  CAP_LIB_VARIANT=packetsub python tools/primary_ab.py 1.0 32 2>&1 | tail -1
  python tools/primary_ab.py 1.0 32 2>&1 | tail -1
done
timeout -k 10 900 python -m pytest tests/test_sponza_class_gpu.py tests/test_bvh_gpu.py tests/test_parity_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -5
