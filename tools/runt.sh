R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu --durations=5 > gpurun_out/tg.log 2>&1; tail -25 gpurun_out/tg.log
