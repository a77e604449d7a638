for sm in 6144 10240 12288 16384 10240 6144; do
SHEPSEG_DFS_SMALL=$sm timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("bmw_small $sm", d["value"], d["ms_per_step"], d["config"]["host_timers_s"], {k[:8]:round(v/144,1) for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
