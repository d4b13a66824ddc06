#!/bin/bash
# usage (GPU box): tools/r04_third.sh <tag>  -- the fused step: parity first, then bench, shards, whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04c}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overflow or repeated or two_streams or kernel_times or many_repertoires" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
for f in 1 0; do
  timeout 600 python3 bench.py --cpu-sample -1 --tunable fused_step=$f > $O/bench_f$f.json 2> $O/bench_f$f.err
  tail -2 $O/bench_f$f.err
  python3 - $O/bench_f$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("step %.4f ms probe %.4f rest %.4f parity %s incl_layout %.3g dev %s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["parity_vs_reference_full_size"], d["value_incl_layout"], d["device_resident_inputs"]))
PY
done
for n in 2 4 8; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 > $O/w$n.json 2> $O/w$n.err
  python3 - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("work shard 1/%s: step %.4f ms probe %.4f rest %.4f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"]))
PY
done
timeout 600 python3 bench.py --cpu-sample -1 --indels > $O/bench_cfg4.json 2> $O/bench_cfg4.err; python3 -c "
import json,sys
d=json.loads(open('$O/bench_cfg4.json').read().strip().splitlines()[-1]); print('cfg4 step', d['ms_per_step'], d['parity_vs_reference_full_size'])"
timeout 600 python3 bench.py --cpu-sample -1 --self > $O/bench_self.json 2> $O/bench_self.err; python3 -c "
import json,sys
d=json.loads(open('$O/bench_self.json').read().strip().splitlines()[-1]); print('self step', d['ms_per_step'], d['parity_vs_reference_full_size'])"
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
