# full GPU verification: all gpu tests, smoke, default bench (with cpu baseline)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/full_tests.txt
tail -5 gpurun_out/full_tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/bench_default.txt 2>&1; tail -1 gpurun_out/bench_default.txt
