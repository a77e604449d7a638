R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -2 gpurun_out/tg.log
rm -rf gpurun_out/prof_tr
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tr -o run -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/prof_tr.log 2>&1
python tools/trace_summary.py $(find gpurun_out/prof_tr -name "*kernel_trace.csv") > gpurun_out/trace_summary.txt 2>&1
rm -f $(find gpurun_out/prof_tr -name "*kernel_trace.csv")
timeout -k 10 400 python bench.py --cpu-sample 0 > gpurun_out/bd.log 2>&1; tail -1 gpurun_out/bd.log | cut -c1-300
