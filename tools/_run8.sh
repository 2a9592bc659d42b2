cd $GRAFT_REPO_ROOT
O=gpurun_out/thresholds_r06.txt; : > $O
echo "# camera rays: packet walk (CAP_PRIMARY_WIDE=0) vs wide per-lane kernel (=1), device SAH tree, 8 spp at 1080p" >> $O
for sc in 1 2 4; do for pw in 0 1; do CAP_PRIMARY_WIDE=$pw timeout -k 10 200 python tools/primary_ab.py $sc 8 2>/dev/null >> $O; done; done
echo "# shadow rays: per-chunk kernel (CAP_ANY_REFILL=0) vs lane refill (=1)" >> $O
for sc in 1 4 8; do for ar in 0 1; do CAP_ANY_REFILL=$ar timeout -k 10 200 python tools/primary_ab.py $sc 8 2>/dev/null >> $O; done; done
echo "# closest8 refill threshold on the SAH tree (hall_stages scale 8 / 1)" >> $O
for rf in 8 16 24 32; do echo "== CAP_W8_REFILL=$rf" >> $O; CAP_W8_REFILL=$rf timeout -k 10 200 python tools/hall_stages.py 8 2>/dev/null | cut -c1-200 >> $O; CAP_W8_REFILL=$rf timeout -k 10 200 python tools/hall_stages.py 1 2>/dev/null | cut -c1-200 >> $O; done
cat $O
