# parity tests, then a one-worker kernel profile on a 12288^2 image, then the default bench
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -3 gpurun_out/tg.log
rm -rf gpurun_out/prof_w1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w1 -o run -- python bench.py --size 12288 --steps 2 --warmup 1 --cpu-sample 0 --workers 1 > gpurun_out/prof_w1.log 2>&1
rm -f gpurun_out/prof_w1/*kernel_trace.csv gpurun_out/prof_w1/*/*kernel_trace.csv
timeout -k 10 400 python bench.py --cpu-sample 0 > gpurun_out/bd.log 2>&1; tail -1 gpurun_out/bd.log | cut -c1-400
