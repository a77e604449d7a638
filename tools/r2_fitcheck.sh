cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fit_elkan.py tests/test_gpu_tile.py -q -m gpu -k "fit or kmeans or elkan" > gpurun_out/fitcheck.txt 2>&1; echo "rc=$?" >> gpurun_out/fitcheck.txt
SHEPSEG_FUZZ_DUMP=gpurun_out timeout -k 10 900 python tests/fuzz_gpu.py 1200 500 more fit >> gpurun_out/fitcheck.txt 2>&1
grep -v "^  \.\.\." gpurun_out/fitcheck.txt | tail -12
