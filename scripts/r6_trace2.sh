#!/bin/bash
# usage: scripts/r6_trace2.sh <tag> [bench args]: kernel trace of the TWO-lane bench run (env MPSFR_* options pass through),
# per-queue durations / gaps (scripts/r5_trace_gaps.py)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace2_$1; shift
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 400 --warmup 10 --cpu-rows 0 --f64-steps 0 --profile-steps 0 --unpruned-steps 0 --host-steps 0 --native-steps 0 --e2e-steps 0 --prime 200 --min-seconds 0 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
grep '^{' $OUT/log.txt | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print('bench under rocprof: %.3f M PSFs/s' % (b['value']/1e6))"
python3 scripts/r5_trace_gaps.py $OUT > $OUT/trace_gaps.txt
cat $OUT/trace_gaps.txt
find $OUT -name "*kernel_trace.csv" -size +64M -delete
