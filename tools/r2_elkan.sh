# the Elkan path on the GPU: its tests, the older fit tests, the fit's timing on the benchmark sample
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fit_elkan.py -x -q -m gpu > gpurun_out/elkan.txt 2>&1; echo "rc=$?" >> gpurun_out/elkan.txt
timeout -k 10 900 python -m pytest tests/test_gpu_tile.py -x -q -m gpu -k "fit or kmeans" >> gpurun_out/elkan.txt 2>&1; echo "rc=$?" >> gpurun_out/elkan.txt
for a in auto elkan; do
  SHEPSEG_FIT_ALGO=$a SHEPSEG_FIT_TIMING=1 timeout -k 10 300 python tools/perf_fit.py >> gpurun_out/elkan.txt 2>&1
done
tail -40 gpurun_out/elkan.txt
