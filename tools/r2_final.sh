# the bench workloads of the round's final code: C3 (with the CPU baseline leg), C4, C5, the .npy-file source
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 500 python bench.py --steps 5 > gpurun_out/f_c3.log 2>&1 || { tail -5 gpurun_out/f_c3.log; exit 1; }
tail -1 gpurun_out/f_c3.log | cut -c1-3000
timeout -k 10 400 python bench.py --workload c4 --steps 3 --cpu-sample 0 > gpurun_out/f_c4.log 2>&1 || { tail -5 gpurun_out/f_c4.log; exit 1; }
tail -1 gpurun_out/f_c4.log | cut -c1-1200
timeout -k 10 400 python bench.py --workload c5 --steps 3 --cpu-sample 0 > gpurun_out/f_c5.log 2>&1 || { tail -5 gpurun_out/f_c5.log; exit 1; }
tail -1 gpurun_out/f_c5.log | cut -c1-1200
SHEPSEG_IO_TIMING=1 timeout -k 10 500 python bench.py --source npy --steps 2 --cpu-sample 0 > gpurun_out/f_npy.log 2>&1 || { tail -5 gpurun_out/f_npy.log; exit 1; }
grep -v "^{" gpurun_out/f_npy.log | tail -12; tail -1 gpurun_out/f_npy.log | cut -c1-1500
