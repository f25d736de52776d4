# rocprofv3 kernel trace of a python tool, summarised per kernel name (median / min / n).  Usage: bash tools/trace_cmd.sh <tag> tools/x.py [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${tag}_prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_prof -- python3 "$@" > gpurun_out/${tag}_trace.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/{tag}_prof/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(f'gpurun_out/{tag}_trace_summary.txt', 'w') as w:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        w.write(f"{k[:110]:110s} n={len(v):5d} med={v[len(v)//2]/1e3:8.1f} us min={v[0]/1e3:8.1f} total={sum(v)/1e6:8.2f} ms\n")
PY
rm -rf gpurun_out/${tag}_prof
