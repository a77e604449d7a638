R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -15 gpurun_out/tg.log
