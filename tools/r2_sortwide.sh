# A/B of the 4096-item staged sort tile on C5, the statistics tests, and the dump of one fuzz case
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{
for w in 1 0 1 0; do echo "== SHEPSEG_SORT_WIDE=$w"; SHEPSEG_SORT_WIDE=$w timeout -k 10 300 python tools/perf_stats_c5.py | tail -3; done
} > gpurun_out/sortwide.txt 2>&1 &&
timeout -k 10 600 python -m pytest tests/test_gpu_stats.py -x -q -m gpu >> gpurun_out/sortwide.txt 2>&1 &&
{ SHEPSEG_FUZZ_DUMP=gpurun_out timeout -k 10 300 python tests/fuzz_gpu.py 1200 321 more fit >> gpurun_out/sortwide.txt 2>&1; true; }
tail -30 gpurun_out/sortwide.txt
