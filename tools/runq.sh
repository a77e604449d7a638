for w in 2 4 6 8 12 16; do
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --workers $w > gpurun_out/bq.log 2>&1; python - <<PY
import json
d=json.loads(open("gpurun_out/bq.log").read().strip().splitlines()[-1])
t=d["config"]["host_timers_s"]
print("workers",$w, d["value"], d["ms_per_step"], "seg_sum %.1f per_tile_ms %.0f" % (t["segmentation"], t["segmentation"]/144*1000), {k[:10]:round(v/144,1) for k,v in d["roofline"]["device_ms_by_kernel"].items()})
PY
done
