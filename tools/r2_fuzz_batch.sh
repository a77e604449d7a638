# a long randomised parity run: default kinds, the extra kinds, big rasters
R=$PWD; cd /tmp && export TMPDIR=/tmp; cd $R
: > gpurun_out/fuzz_batch.txt
for sd in $(seq 700 709); do
  timeout -k 10 300 python tests/fuzz_gpu.py 1000 $sd > gpurun_out/fz.log 2>&1
  grep "MISMATCH\|fuzz_gpu seed" gpurun_out/fz.log | cut -c1-300 >> gpurun_out/fuzz_batch.txt
done
for sd in 720 721 722; do
  timeout -k 10 300 python tests/fuzz_gpu.py 1200 $sd more > gpurun_out/fz.log 2>&1
  grep "MISMATCH\|fuzz_gpu seed" gpurun_out/fz.log | cut -c1-300 >> gpurun_out/fuzz_batch.txt
done
timeout -k 10 600 python tests/fuzz_gpu.py 60 723 big > gpurun_out/fz.log 2>&1
grep "MISMATCH\|fuzz_gpu seed" gpurun_out/fz.log | cut -c1-300 >> gpurun_out/fuzz_batch.txt
grep -c "fuzz_gpu seed" gpurun_out/fuzz_batch.txt
