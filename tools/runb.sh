# default bench under a few settings of one environment knob: bash tools/runb.sh VAR v1 v2 ...
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
V=$1; shift
for x in "$@"; do
  echo "== $V=$x"
  env $V=$x timeout -k 10 300 python bench.py --cpu-sample 0 --steps 3 > gpurun_out/rb_$x.log 2>&1
  tail -1 gpurun_out/rb_$x.log | cut -c1-160
done
