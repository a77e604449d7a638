R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/prof_tr
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tr -o run -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/prof_tr.log 2>&1
python tools/trace_summary.py $(find gpurun_out/prof_tr -name "*kernel_trace.csv") > gpurun_out/trace_summary.txt 2>&1
rm -f $(find gpurun_out/prof_tr -name "*kernel_trace.csv")
