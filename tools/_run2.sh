cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/sahdev1.txt
: > $O
for leaf in 16 32 64 256; do echo "== leaf $leaf" >> $O; CAP_SAHDEV_LEAF=$leaf timeout -k 10 200 python tools/tree_build_ab.py 4 >> $O 2>&1; done
echo "== others" >> $O
timeout -k 10 300 python tools/tree_build_ab.py 2 3 >> $O 2>&1
echo "== cost split scale 1" >> $O
timeout -k 10 300 python tools/tree_cost_split.py 1 >> $O 2>&1
echo "== big scene" >> $O
for leaf in 32 64; do echo "== leaf $leaf" >> $O; CAP_SAHDEV_LEAF=$leaf timeout -k 10 300 python tools/hall_stages.py 8 8 4 >> $O 2>&1; done
timeout -k 10 300 python tools/hall_stages.py 8 8 0 >> $O 2>&1
grep -v amdgpu.ids $O | cut -c1-400
