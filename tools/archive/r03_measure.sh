#!/bin/bash
# the side measurements of DESIGN section 5/6: emulated work shards, self-comparison, nucleotides d=1
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r03m; mkdir -p $O
tools/emulate_work_shards.sh r03m_w 2>&1 | tail -5
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("%-10s step %.4f ms  %.3e q/s  probe %.4f resolve %.4f  positives %d pairs %d  incl_layout %.3e" % (sys.argv[2], d["ms_per_step"], d["value"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], r["pairs_per_launch"], d.get("value_incl_layout") or 0))
PY
}
timeout 300 python bench.py --cpu-sample -1 --self > $O/self.json 2>/dev/null; line $O/self.json self
timeout 300 python bench.py --cpu-sample -1 --nucleotides --ignore-genes > $O/nt1.json 2>/dev/null; line $O/nt1.json nt_d1
timeout 300 python bench.py --cpu-sample -1 > $O/cfg3.json 2>/dev/null; line $O/cfg3.json cfg3
