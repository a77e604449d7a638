R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -3 gpurun_out/tg.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-250
