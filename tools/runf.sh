for pl in 1 4 16 1 4 16; do
SHEPSEG_SMALL_POLL=$pl timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("poll $pl", d["value"], d["ms_per_step"], d["config"]["step_s"], {k[:8]:round(v/144,1) for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
