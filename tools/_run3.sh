cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_micro
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/t_all.txt 2>&1
tail -30 gpurun_out/t_all.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/valu_cost.hip -o /tmp/valu_cost && timeout -k 10 300 /tmp/valu_cost > gpurun_out/r06_micro/valu_cost.txt 2>&1
head -40 gpurun_out/r06_micro/valu_cost.txt
