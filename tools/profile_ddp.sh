#!/bin/bash
# Only the legs of tools/profile_all.sh that need the process group or the plain (uninstrumented) kernel trace:
#   tools/profile_ddp.sh r03   (same output directory as profile_all.sh)
set -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
sed -n '/^# stream overlap/,/^python3 tools\/step_timeline/p;/^# (librccl prints/,/^python3 tools\/kfamily.py $OUT\/kt_ddp/p' tools/profile_all.sh > /tmp/ddp_legs.sh
. /tmp/ddp_legs.sh
rm -f $OUT/kt_plain/k_kernel_trace.csv $OUT/kt_ddp/k_kernel_trace.csv
tail -n 5 $OUT/overlap.txt $OUT/bench_force_ddp*.json
