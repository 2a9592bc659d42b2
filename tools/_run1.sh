set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd capsaicin_amd/csrc && make -B trace8.o EXTRA=-DCAP_W8_COUNT >/dev/null 2>&1 && make EXTRA=-DCAP_W8_COUNT >/dev/null 2>&1)
timeout -k 10 300 python tools/w8_counts.py 8 > gpurun_out/w8_counts_big.json 2>gpurun_out/w8_err.txt
timeout -k 10 300 python tools/w8_counts.py 1 > gpurun_out/w8_counts_tree.json 2>>gpurun_out/w8_err.txt
(cd capsaicin_amd/csrc && make -B trace8.o >/dev/null 2>&1 && make >/dev/null 2>&1)
timeout -k 10 300 python tools/hall_stages.py 8 > gpurun_out/stages_base.txt 2>>gpurun_out/w8_err.txt
timeout -k 10 300 python tools/hall_stages.py 1 >> gpurun_out/stages_base.txt 2>>gpurun_out/w8_err.txt
cat gpurun_out/w8_counts_big.json gpurun_out/w8_counts_tree.json gpurun_out/stages_base.txt
