#!/bin/bash
# Address-translation counters of one workload's kernels (TCP_UTCL1_*): tools/tlb_pmc.sh big|tree   (run through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
WL=${1:-big}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CAP_NO_TWO_LANES=1
for pass in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
            "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  rm -rf $OUT/prof_tlb
  timeout -k 10 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/prof_tlb -- python3 $ROOT/bench.py --only $WL > $OUT/prof_tlb.log 2>&1 || { tail -5 $OUT/prof_tlb.log; continue; }
  python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/prof_tlb/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cap::", "").replace("cap::", "")[:40]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(d.items(), key=lambda kv: -max(kv[1].values()))[:6]:
    print("%-42s %s" % (k, "  ".join("%s=%.4g" % (a.replace("TCP_UTCL1_", "").replace("_sum", ""), b) for a, b in sorted(c.items()))))
PY
done
