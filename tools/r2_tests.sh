cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -q -m gpu > gpurun_out/full_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/full_tests.txt
tail -8 gpurun_out/full_tests.txt
