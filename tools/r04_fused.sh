#!/bin/bash
# usage (GPU box): tools/r04_fused.sh <tag>  -- fused vs three-kernel step at N = 1, 8 (emulated)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04e}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overflow or repeated or two_streams or kernel_times or many_repertoires or random_sets" > $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
for f in 1 0; do
 for n in 1 8; do
  timeout 600 python3 bench.py --steps 30 --warmup 5 --cpu-sample -1 --tunable fused_step=$f --tunable work_shard_count=$n --tunable work_shard_index=0 $EXTRA > $O/b_f${f}_n$n.json 2> $O/b_f${f}_n$n.err
  python3 - $O/b_f${f}_n$n.json $f $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("fused=%s shard 1/%s: step %.4f ms probe %.4f rest %.4f parity %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["parity_vs_reference_full_size"]))
PY
 done
done
