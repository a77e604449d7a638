for w in 8 16; do
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --workers $w > gpurun_out/bq.log 2>&1; python - <<PY
import json
d=json.loads(open("gpurun_out/bq.log").read().strip().splitlines()[-1])
print("workers",$w, d["value"], d["ms_per_step"], d["config"]["host_timers_s"], {k[:10]:v for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
