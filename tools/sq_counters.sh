#!/bin/bash
# SQ instruction / busy counters of the MFMA chunk kernel (rocprofv3 --pmc, counters only, two passes of 8) for the dense prefix
# and the C4 chunk step: VALU : MFMA instruction ratio, MFMA busy cycles, LDS bank conflicts.  Prints per-kernel means (millions).
# Called by tools/prof_round.sh; output is copied to profiles/<tag>_sq_counters.txt.
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
run() {   # label, command...
  local label=$1; shift
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    rm -rf /tmp/pp; timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -- "$@" > /tmp/pp.log 2>&1
    LABEL="$label" python - <<'PY'
import collections, csv, glob, os
f = glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True)
if not f:
    print(os.environ["LABEL"], "no counters:", open('/tmp/pp.log').read()[-400:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if any(n in r['Kernel_Name'] for n in ('ekv_attn_chunk', 'ekv_attn_wide', 'ekv_attn_resident', 'score_select')):
        acc[r['Kernel_Name'][:72] + ' wg=' + r['Workgroup_Size'] + ' vgpr=' + r['VGPR_Count']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(os.environ["LABEL"], '|', k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, '(millions per launch, mean of', len(next(iter(v.values()))), 'launches)')
PY
  done
}
run "dense prefix 4906 tokens x 8 layers" python $R/tools/bench_prefix.py 4906 8
run "C4 chunk step (S=9994 stride 96: one pass + column-sum pass)" python $R/tools/bench_chunk.py 9994 96 4
export MODE=ppl BUDGET=0.39949283136642936 STREAMING=1 SHAPE=40,40,40
run "configs[4] chunk step (S=10253 stride 96, RoPE-on-read: one pass + column-sum pass)" python $R/tools/bench_chunk.py 10253 96 4
unset MODE BUDGET STREAMING SHAPE
BUDGET=0.3 run "configs[2] chunk step (S=4096 stride 16, 8 KV heads x GQA 4: the logits-resident kernel)" python $R/tools/bench_chunk.py 4096 16 8 8
