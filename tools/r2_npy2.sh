# .npy pipeline with the files on tmpfs (only when the box has the memory to spare)
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
avail=$(awk '/MemAvailable/ {print int($2/1048576)}' /proc/meminfo); echo "MemAvailable ${avail} GiB"; df -h /dev/shm | tail -1
[ "$avail" -gt 150 ] || { echo "not enough memory for tmpfs files"; exit 0; }
SHEPSEG_IO_TIMING=1 timeout -k 10 500 python bench.py --source npy --scratch /dev/shm --steps 2 --cpu-sample 0 > gpurun_out/f_npy_shm.log 2>&1; rc=$?
rm -f /dev/shm/shepseg_bench_*
[ $rc = 0 ] || { tail -5 gpurun_out/f_npy_shm.log; exit 1; }
grep "io\]" gpurun_out/f_npy_shm.log | tail -5 | tr '\n' ';'; echo
tail -1 gpurun_out/f_npy_shm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_s'], d['config']['host_timers_s'])"
