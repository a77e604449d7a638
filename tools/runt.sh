R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tg.log 2>&1; tail -4 gpurun_out/tg.log
SHEPSEG_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-300
