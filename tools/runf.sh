for q in 16 20 24 16; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("queues $q", d["value"], d["ms_per_step"], d["config"]["host_timers_s"])
PY
done
