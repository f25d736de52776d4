for w in 2048 1536 1280 1024 768 512; do echo "WGS $w"; MDT_LN_BWD_WGS=$w timeout -k 10 120 python tools/ln_bench.py 2>&1 | grep backward; done
