# the .npy source / sink pipeline: "ENV=..,ENV=.." settings, one run each
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
for spec in "$@"; do
  for kv in $(echo "$spec" | tr ',' ' '); do export $kv; done
  echo "== $spec"
  SHEPSEG_IO_TIMING=1 timeout -k 10 500 python bench.py --source npy --steps 3 --cpu-sample 0 > gpurun_out/f_npy.log 2>&1 || { tail -5 gpurun_out/f_npy.log; exit 1; }
  for kv in $(echo "$spec" | tr ',' ' '); do unset ${kv%%=*}; done
  grep -v "^{" gpurun_out/f_npy.log | grep "io\]" | tail -5 | tr '\n' ';'; echo
  tail -1 gpurun_out/f_npy.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['step_s'], d['config']['host_timers_s'])"
done
