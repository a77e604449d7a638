for sr in 0 1; do
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --simple-recode $sr > gpurun_out/bsr.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bsr.log").read().strip().splitlines()[-1])
print("simple_recode", $sr, d["value"], d["ms_per_step"], d["config"]["host_timers_s"])
PY
done
