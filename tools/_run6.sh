cd $GRAFT_REPO_ROOT
O=gpurun_out/nee.txt; : > $O
timeout -k 10 600 python -m pytest tests/test_ext_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu 2>&1 | tail -3 >> $O
for rep in 1 2; do
echo "== product (cull on)" >> $O
timeout -k 10 300 python bench.py --only ext --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['result']; print(d['value'], d['ms_per_step'], d['stage_ms'])" >> $O
echo "== CAP_NO_NEE_PAIR_CULL=1" >> $O
CAP_NO_NEE_PAIR_CULL=1 timeout -k 10 300 python bench.py --only ext --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['result']; print(d['value'], d['ms_per_step'], d['stage_ms'])" >> $O
done
echo "== neecheck variant: config3 (guards must stay 0)" >> $O
CAP_LIB_VARIANT=neecheck timeout -k 10 400 python bench.py --only config3 2>&1 | tail -c 400 >> $O
echo >> $O
CAP_LIB_VARIANT=neecheck timeout -k 10 400 python bench.py --only config5 2>&1 | tail -c 300 >> $O
cat $O
