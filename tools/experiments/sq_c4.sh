#!/bin/bash
# LDS counters of the configs[4] (RoPE-on-read) chunk step only: [EASYKV_HIP_LIB=variant] tools/experiments/sq_c4.sh
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export MODE=ppl BUDGET=0.39949283136642936 STREAMING=1 SHAPE=40,40,40
rm -rf /tmp/pp; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pp -- python $R/tools/bench_chunk.py 10253 96 4 > /tmp/pp.log 2>&1
python - <<'PY'
import collections, csv, glob
f = glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if 'ekv_attn_wide' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:72]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, '(millions per launch)')
PY
