#!/bin/bash
# usage (GPU box): tools/secondary_workloads.sh <tag>  -- bench lines of the other BASELINE configurations
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; mkdir -p $O
run() {  # name, bench args...
  local name=$1; shift
  timeout 1500 python bench.py --cpu-sample -1 "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; c=d["config"]
    print("%-22s %.4g query/s  step %.3f ms (probe %.3f, resolve %.3f)  variant %s  setup index %.2f s layout %.3f s  checksum %s" % (
      sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], c["layout"]["variant"],
      c["setup_seconds"]["index_build+upload"], c["setup_seconds"]["query_layout+upload"], c["matrix_checksum"][:8]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run cfg2_d0_1m      --refs 1000000 --queries 1000000 --differences 0
run cfg3_indels     --indels
run self_d1         --self
run nt_d1_10m       --nucleotides --ignore-genes --steps 10 --warmup 2
run nt_d2_2m_5m     --nucleotides --ignore-genes --differences 2 --refs 5000000 --queries 2000000 --steps 5 --warmup 2
run nt_d2_2m_100m   --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 2000000 --steps 2 --warmup 1
run cfg5_12m5_100m  --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1
