# rehearsal of the multi-rank path at full C3 size on ONE GPU: two ranks share the device, strips go
# through the socket transport; max_seg_id must equal the single-process run's (2569557 with the reference model of C3)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for m in parallel sequential; do
  SHEPSEG_COMM=socket SHEPSEG_STITCH=$m timeout -k 10 500 python bench.py --gpus 2 --workers 12 --steps 2 --cpu-sample 0 > gpurun_out/two_$m.log 2>&1 || { tail -20 gpurun_out/two_$m.log; exit 1; }
  grep "^{" gpurun_out/two_$m.log | tail -1 | cut -c1-900
done
