# Per-kernel times in situ for two builds of libmdt_hip.so (ab_libs/old.so / ab_libs/new.so), one-stream accounting:
# (OLD_ENV / NEW_ENV: extra NAME=VALUE settings of an arm; without ab_libs/*.so both arms use the built library)
# rocprofv3 kernel stats of a short bench run with each, into gpurun_out/pab_{old,new}/.  gpurun -- 'bash tools/prof_ab_libs.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MDT_SKIP_SOURCE_HASH=1 MDT_TWO_STREAMS=${MDT_TWO_STREAMS:-0}
L=multimodaldiscussiontransformer_amd/libmdt_hip.so
for v in old new; do
  [ -f ab_libs/$v.so ] && cp ab_libs/$v.so $L
  if [ $v = old ]; then export $OLD_ENV MDT_AB_ARM=old; else export $NEW_ENV MDT_AB_ARM=new; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pab_$v -- python3 bench.py --foreign-library --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer --no-verify-exchange > gpurun_out/pab_$v.log 2>&1 || exit 1
done
[ -f ab_libs/old.so ] && cp ab_libs/old.so $L
exit 0
