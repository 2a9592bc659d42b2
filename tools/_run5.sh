cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/t_all.txt 2>&1
tail -12 gpurun_out/t_all.txt
timeout -k 10 400 python bench.py --only ingest > gpurun_out/b_ingest.json 2>gpurun_out/b_ingest.err; tail -c 1500 gpurun_out/b_ingest.json
timeout -k 10 400 python bench.py --only realtime > gpurun_out/b_realtime.json 2>gpurun_out/b_realtime.err; tail -c 3000 gpurun_out/b_realtime.json
