for cfg in "20 16 16" "24 20 20" "28 24 24" "24 20 16"; do
set -- $cfg
SHEPSEG_SMALL_MAX=$3 GPU_MAX_HW_QUEUES=$1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --workers $2 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("queues $1 workers $2 smallmax $3", d["value"], d["ms_per_step"], d["config"]["host_timers_s"])
PY
done
