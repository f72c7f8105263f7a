#!/bin/bash
# GPU idle gaps inside one steady-state step of the default bench.py workload (rocprofv3 kernel trace
# -> tools/trace_gaps.py).    usage (on the GPU box): tools/profile_gaps.sh   -> gpurun_out/gaps.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-native-side --no-secondary --no-extra-passes --no-events > $R/gpurun_out/gaps_run.log 2>&1
T=$(find /tmp/prof_gaps -name '*kernel_trace.csv' | head -1)
python3 $R/tools/trace_gaps.py $T 15 > $R/gpurun_out/gaps.txt
head -60 $R/gpurun_out/gaps.txt
