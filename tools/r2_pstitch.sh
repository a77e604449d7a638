# the parallel stitch on the GPU: the sharded-driver tests, then C3 through the sharded driver at world 1 in both forms
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rc=0

for m in sequential parallel; do
  SHEPSEG_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 SHEPSEG_STITCH=$m timeout -k 10 400 python bench.py --gpus 1 --steps 3 --cpu-sample 0 > gpurun_out/ps_$m.log 2>&1 || { tail -5 gpurun_out/ps_$m.log; exit 1; }
  tail -1 gpurun_out/ps_$m.log | cut -c1-900
done
