cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/t_all.txt 2>&1
tail -8 gpurun_out/t_all.txt
bash tools/prof_all.sh > gpurun_out/prof_all.log 2>&1; tail -2 gpurun_out/prof_all.log; tail -2 gpurun_out/prof_all_progress.log
