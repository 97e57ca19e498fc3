#!/bin/bash
# Evidence run of a round on the GPU box (called through gpurun): rocprofv3 kernel trace of the default bench command, then
# FETCH_SIZE / WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: they do not fit one pass; never combined with a trace)
# for the decode bench and for each strided-prefill shape.  Raw outputs -> gpurun_out/prof_$TAG; tools/summarize_prof.py
# condenses them into profiles/${TAG}_*.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
run_pmc() {   # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/p_$name
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p_$name -- "$@" > $OUT/${name}.log 2>&1
  mkdir -p $OUT/$name
  cp $(find /tmp/p_$name -name "*counter_collection.csv" | head -1) $OUT/$name/ 2>/dev/null
}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/bench.py --steps 512 --no-cpu-baseline --no-live-pmc > $OUT/bench_under_trace.log 2>&1
mkdir -p $OUT/trace; cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/trace/
D="python $R/bench.py --no-cpu-baseline --no-live-pmc --steps 16 --warmup 4 --prewarm-s 0.05 --no-prefill --no-boundary"
run_pmc decode_fetch FETCH_SIZE $D
run_pmc decode_write WRITE_SIZE $D
for cfg in "c2 4096 8 24" "s64 4096 64 12" "c4 9994 96 8"; do
  set -- $cfg
  run_pmc chunk_$1_fetch FETCH_SIZE python $R/tools/bench_chunk.py $2 $3 $4
  run_pmc chunk_$1_write WRITE_SIZE python $R/tools/bench_chunk.py $2 $3 $4
done
# BASELINE configs[2]: Mistral GQA (8 KV heads), stride 16, budget 0.3
BUDGET=0.3 run_pmc chunk_c3m_fetch FETCH_SIZE python $R/tools/bench_chunk.py 4096 16 12 8
BUDGET=0.3 run_pmc chunk_c3m_write WRITE_SIZE python $R/tools/bench_chunk.py 4096 16 12 8
# BASELINE configs[4]: Llama2-13B heads, ppl geometry, streaming RoPE-on-read
export MODE=ppl BUDGET=0.39949283136642936 STREAMING=1 SHAPE=40,40,40
run_pmc chunk_c5_fetch FETCH_SIZE python $R/tools/bench_chunk.py 10253 96 6
run_pmc chunk_c5_write WRITE_SIZE python $R/tools/bench_chunk.py 10253 96 6
unset MODE BUDGET STREAMING SHAPE
timeout 600 bash $R/tools/sq_counters.sh > $OUT/sq_counters.txt 2>&1
# cycle stamps of the scorer tail of the wide column-sum pass (a -DEKV_TAIL_PROFILE build of the m2 translation units, made in the build
# container: tools/experiments/build_variant.sh tailprof "-DEKV_TAIL_PROFILE" ekv_attn_wide_d128_m2.hip)
if [ -f $R/easykv_amd/csrc/variants/lib_tailprof.so ]; then
  EASYKV_HIP_LIB=$R/easykv_amd/csrc/variants/lib_tailprof.so timeout 300 python $R/tools/experiments/exp_widetail_prof.py c3 s64 c2 > $OUT/wide_tail_stamps.txt 2>&1
fi
# cycle stamps of the logits-resident chunk step (tools/experiments/build_variant.sh resprof "-DEKR_PROFILE" ekv_attn_resident_d128.hip)
if [ -f $R/easykv_amd/csrc/variants/lib_resprof.so ]; then
  EASYKV_HIP_LIB=$R/easykv_amd/csrc/variants/lib_resprof.so timeout 300 python $R/tools/experiments/exp_resident_prof.py > $OUT/resident_stamps.txt 2>&1
fi
cd $R
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
ls $OUT
