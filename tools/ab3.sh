# In-call comparison of several builds of libmdt_hip.so kept under ab_libs/<name>.so (scratch, git-ignored; delete after the run):
#   gpurun -- 'bash tools/ab3.sh old ln ln_nt'   -> gpurun_out/ab3.log, two alternating rounds of the default bench per arm
export MDT_SKIP_SOURCE_HASH=1
L=multimodaldiscussiontransformer_amd/libmdt_hip.so
cp $L /tmp/lib_keep.so
out=gpurun_out/ab3.log; : > $out
for round in 1 2; do
  for v in "$@"; do
    cp ab_libs/$v.so $L
    echo "== $v" >> $out
    timeout -k 10 300 python bench.py --foreign-library --steps 8 --warmup 3 --no-cpu-baseline --no-selfcheck --no-gemm-timer 2>/dev/null | cut -c1-170 >> $out
  done
done
cp /tmp/lib_keep.so $L
cat $out
