#!/bin/bash
# rocprofv3 kernel trace of the default bench.py workload -> steady-state per-kernel table and the
# kernel timeline of one step.    usage (on the GPU box): tools/profile_bench.sh <name> [bench args]
# writes gpurun_out/<name>_steady.txt, gpurun_out/<name>_sequence.txt, gpurun_out/<name>_stats.csv
NAME=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-native-side --no-secondary --no-extra-passes "$@" > $R/gpurun_out/${NAME}_run.log 2>&1
T=$(find /tmp/prof_$NAME -name '*kernel_trace.csv' | head -1)
S=$(find /tmp/prof_$NAME -name '*kernel_stats.csv' | head -1)
python3 $R/tools/trace_stats.py $T 5 > $R/gpurun_out/${NAME}_steady.txt
python3 $R/tools/trace_sequence.py $T ${SEQ_MIN_US:-40} > $R/gpurun_out/${NAME}_sequence.txt
head -60 $S > $R/gpurun_out/${NAME}_stats.csv
tail -1 $R/gpurun_out/${NAME}_run.log | cut -c1-400
