for sb in 64 32 96 48 64; do
SHEPSEG_SMALL_BLOCKS=$sb timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("small_blocks $sb", d["value"], d["ms_per_step"], d["config"]["host_timers_s"], round(d["roofline"]["device_ms_by_kernel"]["k_small_loop"]/144,1))
PY
done
