#!/bin/bash
# usage (GPU box): tools/r06_rechash_ab.sh [tag] -- record tiles with the hash in the record (record_tiles = 1, the default
# where every sequence is within 28 residues) against records hashed by the kernel (2) and per-slot arrays (0)
tag=${1:-r06_rh}
O=gpurun_out/$tag; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "adversarial or synthetic_aa or d0 or ragged or long_seq or existence or pairs_list or work_shards or routed or device_resident or full_size_matches or repeated_launches or wrap or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 600 python3 bench.py --cpu-sample -1"
for rt in 1 2 0; do
  $B --tunable record_tiles=$rt > $O/cfg3_rt$rt.json 2> $O/cfg3_rt$rt.err
  $B --indels --tunable record_tiles=$rt > $O/cfg4_rt$rt.json 2> $O/cfg4_rt$rt.err
done
python3 - $O <<'PY' | tee $O/summary.txt
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1])
        k={a:round(b,3) for a,b in j['step_kernels_ms'].items()}
        print("%-18s set %.3f ms  resident %.3f  %s  parity %s %s"%(f.split('/')[-1][:-5], j['ms_per_step'], j['resident_step_ms'], k, j.get('parity_vs_reference_full_size'), j.get('resident_steps_same_matrix')))
    except Exception as e:
        print(f, "FAILED", e)
PY
