cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fit_elkan.py -x -q -m gpu > gpurun_out/elkan2.txt 2>&1; echo "rc=$?" >> gpurun_out/elkan2.txt
bash tools/r2_elkan_prof.sh >> gpurun_out/elkan2.txt 2>&1
tail -22 gpurun_out/elkan2.txt
