for cfg in "16 16 0" "32 16 1" "32 24 0" "24 12 1" "8 8 0"; do
set -- $cfg
GPU_MAX_HW_QUEUES=$1 SHEPSEG_DFS_FORK=$3 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --workers $2 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("queues $1 workers $2 fork $3", d["value"], d["ms_per_step"], {k[:10]:round(v/144,1) for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
