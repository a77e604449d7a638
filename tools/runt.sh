R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -4 gpurun_out/tg.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
