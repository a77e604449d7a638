# round 3: the fit on the benchmark sample: trace of the Elkan iterations, then timings
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
SHEPSEG_FIT_TRACE=1 timeout -k 10 200 python tools/perf_fit.py > gpurun_out/r3_fit_trace.log 2>&1 || { tail -5 gpurun_out/r3_fit_trace.log; exit 1; }
grep "elkan batch" gpurun_out/r3_fit_trace.log | head -40
SHEPSEG_FIT_TIMING=1 timeout -k 10 200 python tools/perf_fit.py > gpurun_out/r3_fit_timing.log 2>&1; tail -12 gpurun_out/r3_fit_timing.log
