# In-call A/B of two builds of libmdt_hip.so (ab_libs/old.so / ab_libs/new.so — scratch copies, git-ignored, DELETE them after the run: they ship with every gpurun lease): boxes differ by
# several per cent, so both arms must run inside ONE gpurun call.  Usage: gpurun -- 'bash tools/ab_libs.sh [cmd...]'
set -e
export MDT_SKIP_SOURCE_HASH=1   # two builds against one csrc/: the import-time source-hash check is for shipped trees
L=multimodaldiscussiontransformer_amd/libmdt_hip.so
out=gpurun_out/ab_libs.log; rm -f $out
cmd=${@:-python bench.py --foreign-library --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck}
for v in old new old new; do
  cp ab_libs/$v.so $L
  echo "== $v" >> $out
  timeout -k 10 400 $cmd 2>/dev/null | cut -c1-170 >> $out
done
cp ab_libs/new.so $L
