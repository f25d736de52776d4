"""Copy a measurement set produced by tools/run_profile.sh <tag> from gpurun_out/ into profiles/ (named per round)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "round1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", "")
dst = os.path.join(root, "profiles", f"{rnd}_{tag}")
shutil.copy(glob.glob(base + f"{tag}_prof/*/*kernel_stats.csv")[0], dst + "_kernel_stats.csv")
def steps_in(stats_csv):
    """steps in a kernel trace = launches of a kernel that runs exactly once per step (vit_patchify; bert_embed_rows
    for a text-only run)"""
    for r in csv.DictReader(open(stats_csv)):
        if "vit_patchify" in r["Name"] or "bert_embed_rows" in r["Name"]:
            return int(r["Calls"])
    return None


json.dump({"steps_in_trace": steps_in(dst + "_kernel_stats.csv"),
           "note": "divide the TotalDurationNs of a kernel by steps_in_trace for its time per step; the trace covers warm-up, "
                   "timed and roofline-pass steps of the profiled command alike"},
          open(dst + "_kernel_stats_meta.json", "w"), indent=1)
two = glob.glob(base + f"{tag}_prof2/*/*kernel_stats.csv")
if two:
    shutil.copy(two[0], dst + "_kernel_stats_two_streams.csv")
open(dst + "_bench.json", "w").write(open(base + f"{tag}_bench.log").read().strip().split("\n")[-1] + "\n")


def agg(kind, counter):
    f = glob.glob(base + f"{tag}_pmc_{kind}/*/*counter_collection.csv")[0]
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        key = "other"
        for k in ("gemm_f8_w4", "gemm_bf16_w4s", "gemm_bf16_w4p", "gemm_bf16_pp256p", "gemm_bf16_pp256", "gemm_bf16_tile128", "attn_bwd_v5", "attn_bwd_v4", "attn_bwd_v3", "attn_fwd_v2", "layernorm_bwd",
                  "layernorm_fwd", "colsum"):
            if k in n:
                key = k
                break
        d[key][0] += 1
        d[key][1] += float(r["Counter_Value"])
    return d


fe, wr = agg("fetch", "FETCH_SIZE"), agg("write", "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over: python3 bench.py --steps 1 --warmup 1 "
                 "--no-cpu-baseline --no-selfcheck --no-gemm-timer",
       "corrections": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md HBM section); "
                      "WRITE_SIZE (KB) x 1024; fabric-side counters: Infinity-Cache hits are included",
       "kernels": {}}
for k in fe:
    n = fe[k][0]
    fb, wb = fe[k][1] * 1024 * 2, wr[k][1] * 1024
    out["kernels"][k] = dict(launches=n, fetch_bytes_per_launch=round(fb / n), write_bytes_per_launch=round(wb / n),
                             bytes_per_launch=round((fb + wb) / n))
g = [out["kernels"][k] for k in ("gemm_bf16_w4s", "gemm_bf16_w4p", "gemm_bf16_pp256p", "gemm_bf16_pp256", "gemm_bf16_tile128") if k in out["kernels"]]
tl = sum(x["launches"] for x in g)
sys.path.insert(0, root)
try:
    import bench
    out["code_state_hash"] = bench.code_state_hash()       # bench.py quotes these numbers only while the code is this
except Exception as e:  # noqa: BLE001
    out["code_state_hash"] = None
    print("no code hash:", e)
out["gemm_family_bytes_per_launch"] = round(sum(x["bytes_per_launch"] * x["launches"] for x in g) / tl)
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
print("saved", dst + "_*", "gemm family bytes/launch", out["gemm_family_bytes_per_launch"])
