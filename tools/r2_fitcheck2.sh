cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fit_elkan.py tests/test_gpu_fullsize.py -q -m gpu -k "elkan or fit or c3_fullsize" > gpurun_out/fitcheck2.txt 2>&1; echo "rc=$?" >> gpurun_out/fitcheck2.txt
SHEPSEG_FIT_TIMING=1 timeout -k 10 300 python tools/perf_fit.py 2>&1 | grep "kmeans fit: n=" | tail -2 >> gpurun_out/fitcheck2.txt
timeout -k 10 600 python tests/fuzz_gpu.py 600 901 more fit 2>&1 | grep -v "^  \.\.\." | tail -3 >> gpurun_out/fitcheck2.txt
tail -9 gpurun_out/fitcheck2.txt
