for cfg in "20 16" "24 20" "28 24" "24 18"; do
set -- $cfg
GPU_MAX_HW_QUEUES=$1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --workers $2 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("queues $1 workers $2", d["value"], d["ms_per_step"], d["config"]["host_timers_s"])
PY
done
