#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:
#     gpurun -- 'bash profiles/collect.sh r03_demo'
#     gpurun -- 'bash profiles/collect.sh r03_c5 --wnlow 333.33 --wnhigh 10000 --wndelt 0.0009667 --wnosamp 1 --layers 150 --lines 10000000'
# (arguments after the tag go to bench.py: the workload)
# Three separate runs (kernel trace; PMC FETCH_SIZE; PMC WRITE_SIZE -- never combined, as
# MI355X_MICROARCH.md prescribes), plus one SQ instruction-mix pass for the walk kernels.  The raw
# CSVs are summarised on the GPU box (profiles/summarize.py) into gpurun_out/<tag>_summary/: copy those
# files into profiles/ to commit them.
TAG=${1:-r03}
shift
WORKLOAD="$@"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras $WORKLOAD"
run() {     # run NAME ARGS... -- one profiled bench run, its CSVs copied out
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout -k 10 ${LIMIT:-300} rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o out -- $BENCH ${STEPS} > $OUT/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $OUT/$name.log; return 1; }
  for f in $(find /tmp/prof_$name -name "*kernel_stats.csv" -o -name "*counter_collection.csv"); do cp $f $OUT/${name}_$(basename $f | sed 's/^out_//'); done
  echo "$name ok"
}
STEPS="" run trace --kernel-trace --stats &&
STEPS="--steps 3 --warmup 2" run fetch --kernel-trace --pmc FETCH_SIZE &&
STEPS="--steps 3 --warmup 2" run write --kernel-trace --pmc WRITE_SIZE &&
STEPS="--steps 3 --warmup 2" run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES &&
STEPS="--steps 3 --warmup 2" run wait --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE &&
STEPS="--steps 3 --warmup 2" run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT &&
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line_under_trace.json
# summarised where the data is; the raw per-dispatch counter CSVs (tens of MB per pass) stay behind
python3 $ROOT/profiles/summarize.py --from $OUT --tag $TAG --dest $ROOT/gpurun_out/${TAG}_summary --cmd "$BENCH" --workload "${WORKLOAD:-bench.py defaults (CH4-demo shape)}" &&
rm -f $OUT/*_counter_collection.csv
ls -la $OUT $ROOT/gpurun_out/${TAG}_summary
