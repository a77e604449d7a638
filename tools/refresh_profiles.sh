#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the default bench + two PMC passes.
# usage: bash tools/refresh_profiles.sh TAG [stats|pmc|all]   (outputs under gpurun_out/prof_TAG_*)
TAG=${1:-x}
WHAT=${2:-all}
R=$PWD
cd /tmp && export TMPDIR=/tmp
cd $R
# counter passes serialise every dispatch and print nothing for minutes: keep a heartbeat file moving
( while true; do date >> gpurun_out/heartbeat_${TAG}.txt; sleep 45; done ) &
HB=$!
rc=0
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
rm -rf gpurun_out/prof_${TAG}_stats gpurun_out/prof_${TAG}_c4 gpurun_out/prof_${TAG}_c5
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -o run -- python bench.py --also 0 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_${TAG}_stats.log 2>&1 || rc=1
[ $rc = 0 ] && { timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c4 -o run -- python bench.py --workload c4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_${TAG}_c4.log 2>&1 || rc=1; }
[ $rc = 0 ] && { timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_c5 -o run -- python bench.py --workload c5 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_${TAG}_c5.log 2>&1 || rc=1; }
fi
if [ $rc = 0 ] && { [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; }; then
rm -rf gpurun_out/prof_${TAG}_pmcf gpurun_out/prof_${TAG}_pmcw
timeout -k 10 1000 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmcf -o run -- python bench.py --also 0 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/prof_${TAG}_pmcf.log 2>&1 &&
timeout -k 10 1000 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmcw -o run -- python bench.py --also 0 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/prof_${TAG}_pmcw.log 2>&1 || rc=1
fi
kill $HB
python tools/summarise_profiles.py ${TAG}
rm -f gpurun_out/prof_${TAG}_*/run*kernel_trace.csv gpurun_out/prof_${TAG}_pmc*/run*counter_collection.csv gpurun_out/heartbeat_${TAG}.txt
exit $rc
