for o in lpt rowmajor lpt rowmajor; do
SHEPSEG_TILE_ORDER=$o timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("order $o", d["value"], d["ms_per_step"], d["config"]["host_timers_s"])
PY
done
