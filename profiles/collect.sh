#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:
#     gpurun -- 'bash profiles/collect.sh r02_v1'
# Three separate runs (kernel trace; PMC FETCH_SIZE; PMC WRITE_SIZE -- never combined, as
# MI355X_MICROARCH.md prescribes), plus one SQ instruction-mix pass for the walk kernels.  The raw
# CSVs land in gpurun_out/<tag>/; `python profiles/summarize.py --tag <tag> --from gpurun_out/<tag>`
# then writes the summaries that are committed here.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras"
run() {     # run NAME ARGS... -- one profiled bench run, its CSVs copied out
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d /tmp/prof_$name -o out -- $BENCH ${STEPS} > $OUT/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $OUT/$name.log; return 1; }
  for f in $(find /tmp/prof_$name -name "*kernel_stats.csv" -o -name "*counter_collection.csv"); do cp $f $OUT/${name}_$(basename $f | sed 's/^out_//'); done
  echo "$name ok"
}
STEPS="" run trace --kernel-trace --stats &&
STEPS="--steps 3 --warmup 2" run fetch --kernel-trace --pmc FETCH_SIZE &&
STEPS="--steps 3 --warmup 2" run write --kernel-trace --pmc WRITE_SIZE &&
STEPS="--steps 3 --warmup 2" run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES &&
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line_under_trace.json
ls -la $OUT
