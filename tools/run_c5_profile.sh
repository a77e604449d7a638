R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/prof_c5
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/prof_c5 -o run -- python tools/perf_stats_c5.py > gpurun_out/c5.log 2>&1
rm -f gpurun_out/prof_c5/*_trace.csv gpurun_out/prof_c5/*/*_trace.csv
tail -4 gpurun_out/c5.log
