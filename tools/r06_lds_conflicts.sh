#!/bin/bash
# usage (GPU box): tools/r06_lds_conflicts.sh [tag]  -- VERDICT r5 item 7: the LDS bank conflicts of 32-byte
# words read at uniformly random offsets (tools/lds_conflicts.hip), timed bare and counted by the hardware.
tag=${1:-r06_ldsc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sz in 58368 32768; do
  $R/tools/bin/lds_conflicts $sz > $O/bare_$sz.json 2> $O/bare_$sz.err
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pmc_$sz -o p --output-format csv -- \
      $R/tools/bin/lds_conflicts $sz > $O/pmc_$sz.log 2>&1
done
find $O -name '*agent_info*' -delete
python3 - $O <<'PY' | tee $O/lds_conflicts.txt
import csv, glob, json, sys, collections, re
O = sys.argv[1]
for sz in (58368, 32768):
    try:
        bare = {f["mode"]: f for f in json.load(open(f"{O}/bare_{sz}.json"))["forms"]}
    except Exception as e:
        print("bare run failed:", e); continue
    ctr = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"{O}/pmc_{sz}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"lds_words_kernel<(\d+)>", r["Kernel_Name"])
            if m:
                ctr[int(m.group(1))][r["Counter_Name"]] += float(r["Counter_Value"])
    print(f"slice of {sz} bytes; 2 workgroups x 16 waves per CU, six words in flight per lane")
    print("%-4s %-58s %8s %12s %12s %9s" % ("mode", "form", "ms", "ns/word/CU", "conflict", "share"))
    for m in sorted(bare):
        c = ctr.get(m, {})
        act, con = c.get("SQ_LDS_IDX_ACTIVE", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        print("%-4d %-58s %8.3f %12.3f %12.3g %9s" % (m, bare[m]["what"][:58], bare[m]["ms"], bare[m]["ns_per_word_per_cu"],
              con, ("%.3f" % (con / act)) if act else "-"))
PY
