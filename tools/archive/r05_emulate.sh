#!/bin/bash
# usage (GPU box): tools/r05_emulate.sh <tag> [cfg2|cfg3|cfg4|cfg5|pubd0 ...]
# One rank's share of an N-GPU run at N = 1, 2, 4, 8, measured on ONE GPU (no 8-GPU node is available to the
# builder): cfg3 / cfg4 divide the step's WORK by filter slice (--shard-by work: tunables work_shard_count = N,
# work_shard_index = 0), cfg5 divides the QUERIES (--shard-by queries: a rank holds 1/N of the 100M queries and
# streams the whole filter of 100M references); cfg2 / pubd0 (d = 0, no filter) divide the queries by hash class -- the split bench.py itself makes for them under torch.distributed.run.
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05e}; shift
O=gpurun_out/$tag; mkdir -p $O
what=${@:-cfg3 cfg4 cfg5}
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
r=d["roofline"]
print("%s: step %.4f ms (probe %.4f, rest %.4f) queries on this GPU %d chunks %s parity %s" % (
    sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["queries_this_gpu"],
    d["config"]["layout"].get("chunks"), d["parity_vs_reference_full_size"]))
PY
}
for w in $what; do
  for n in 1 2 4 8; do
    case $w in
      cfg3) a="--steps 30 --warmup 5 --tunable work_shard_count=$n --tunable work_shard_index=0" ;;
      cfg4) a="--indels --steps 30 --warmup 5 --tunable work_shard_count=$n --tunable work_shard_index=0" ;;
      cfg2) a="--differences 0 --refs 1000000 --queries 1000000 --steps 30 --warmup 5 --tunable work_shard_count=$n --tunable work_shard_index=0" ;;
      pubd0) a="--differences 0 --law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120 --steps 10 --warmup 3 --tunable work_shard_count=$n --tunable work_shard_index=0" ;;
      cfg5) a="--nucleotides --ignore-genes --differences 2 --refs 100000000 --queries $((100000000 / n)) --steps 2 --warmup 1" ;;
    esac
    timeout 1800 python3 bench.py --cpu-sample -1 $a > $O/${w}_n$n.json 2> $O/${w}_n$n.err
    line $O/${w}_n$n.json "$w, one rank of $n"
  done
done
