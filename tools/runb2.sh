# default bench under "ENV=..,ENV=..:bench args" settings: bash tools/runb2.sh "A=1,B=2:--workers 24" ...
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=$(echo "${spec%%:*}" | tr ',' ' ')
  args="${spec#*:}"
  echo "== $spec"
  env $envs timeout -k 10 300 python bench.py --cpu-sample 0 --steps 3 $args > gpurun_out/rb2_$i.log 2>&1
  tail -1 gpurun_out/rb2_$i.log | cut -c60-160
done
