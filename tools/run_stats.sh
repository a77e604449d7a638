# rocprofv3 kernel stats of the default bench (3 steps incl. warmup): per-tile kernel times under load
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/prof_d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -o run -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/prof_d.log 2>&1
rm -f gpurun_out/prof_d/*kernel_trace.csv gpurun_out/prof_d/*/*kernel_trace.csv
tail -1 gpurun_out/prof_d.log | cut -c1-160
