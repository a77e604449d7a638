R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o run -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/tl.log 2>&1
f=$(ls gpurun_out/tl/*kernel_trace.csv gpurun_out/tl/*/*kernel_trace.csv 2>/dev/null | head -1)
python tools/timeline.py $f > gpurun_out/timeline.txt 2>&1
rm -rf gpurun_out/tl
tail -1 gpurun_out/tl.log | cut -c1-200
