# end-of-round verification: all GPU tests, smoke, the bench workloads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/full_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/full_tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/r2_final.sh
