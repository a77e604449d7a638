for f in 1 0 1 0; do
SHEPSEG_DFS_FORK=$f timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/bf.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bf.log").read().strip().splitlines()[-1])
print("fork $f", d["value"], d["ms_per_step"], {k[:10]:round(v/144,1) for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
