R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=3 > gpurun_out/tg.log 2>&1; tail -12 gpurun_out/tg.log
