# round 3: all GPU tests, then the default bench (C3)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tg.log 2>&1; tail -3 gpurun_out/r3_tg.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_default.log 2>&1; tail -1 gpurun_out/r3_bench_default.log | cut -c1-600
