set -e
cp gpurun_lib_new.so multimodaldiscussiontransformer_amd/libmdt_hip.so
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b80.log 2>&1
L=multimodaldiscussiontransformer_amd/libmdt_hip.so
out=gpurun_out/ab80.log; rm -f $out
for v in old new old new old new old new; do
  cp gpurun_lib_$v.so $L
  echo "== $v" >> $out
  timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-selfcheck --no-gemm-timer 2>/dev/null | cut -c1-170 >> $out
done
cp gpurun_lib_new.so $L
