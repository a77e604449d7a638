# round 3: all GPU tests, the default bench (C3), a two-rank rehearsal of `bench.py --gpus 2` (the ranks share
# the one GPU of the box and talk over sockets: not a scaling datum, it exercises the sharded driver's bench line)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_tg.log 2>&1; tail -3 gpurun_out/r3_tg.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_default.log 2>&1; tail -1 gpurun_out/r3_bench_default.log | cut -c1-300
SHEPSEG_COMM=socket SHEPSEG_DEVICE=0 timeout -k 10 500 python bench.py --gpus 2 --steps 1 --warmup 1 --workers 12 --cpu-sample 0 > gpurun_out/r3_two_ranks.log 2>&1; tail -1 gpurun_out/r3_two_ranks.log | cut -c1-1500
