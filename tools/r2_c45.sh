# the new full-size tests (C4, C5), then the three bench workloads
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c4 or c5" > gpurun_out/tg45.log 2>&1; rc=$?; tail -15 gpurun_out/tg45.log
[ $rc = 0 ] || exit 1
timeout -k 10 400 python bench.py --workload c5 --steps 3 > gpurun_out/b_c5.log 2>&1; tail -1 gpurun_out/b_c5.log | cut -c1-1500
timeout -k 10 400 python bench.py --workload c4 --steps 3 --cpu-sample 0 > gpurun_out/b_c4.log 2>&1; tail -1 gpurun_out/b_c4.log | cut -c1-1200
timeout -k 10 400 python bench.py --steps 3 > gpurun_out/b_c3.log 2>&1; tail -1 gpurun_out/b_c3.log | cut -c1-2500
