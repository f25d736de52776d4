"""Aggregate the two SQ counter passes of tools/sq_counters.sh <tag> per kernel family → profiles/<round>_<tag>_sq_counters.json.
Fractions are of SQ_WAVE_CYCLES (cycles summed over resident waves): active = issuing an instruction, wait_inst = stalled
on an instruction dependency / counter, wait_any = parked (barrier, s_waitcnt, sleep).  mfma_util_est = wave-level MFMA
instructions x 16 cycles (v_mfma_f32_16x16x32_bf16 at the dense rate: 16 384 flops / 1 024 flops per cycle per SIMD) over
the kernels' SIMD-cycles (duration from the same pass's kernel trace x 1 024 SIMDs x an assumed 2.1 GHz)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "round1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAM = ("gemm_f8_w4", "gemm_bf16_w4s", "gemm_bf16_w4p", "gemm_bf16_pp256p", "gemm_bf16_pp256", "gemm_bf16_tile128", "attn_bwd_v5", "attn_bwd_v4", "attn_bwd_v3", "attn_fwd_v2", "layernorm_bwd", "layernorm_fwd", "colsum")


def durations():
    f = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_sq_b", "*", "*kernel_trace.csv"))[0]
    d = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        fam = next((k for k in FAM if k in r["Kernel_Name"]), None)
        if fam:
            d[fam] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return d


def load(part):
    f = glob.glob(os.path.join(root, "gpurun_out", f"{tag}_sq_{part}", "*", "*counter_collection.csv"))[0]
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        fam = next((k for k in FAM if k in r["Kernel_Name"]), None)
        if fam is None:
            continue
        d[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[fam] += 1
    return d, n


a, na = load("a")
b, _ = load("b")
dur = durations()
out = {}
for fam in a:
    ca, cb = a[fam], b.get(fam, {})
    wc = max(1.0, ca["SQ_WAVE_CYCLES"])
    out[fam] = dict(launches=na[fam],
                    active_frac=round(ca["SQ_ACTIVE_INST_ANY"] / wc, 3), wait_inst_frac=round(ca["SQ_WAIT_INST_ANY"] / wc, 3),
                    wait_any_frac=round(ca["SQ_WAIT_ANY"] / wc, 3), valu_active_frac=round(ca["SQ_ACTIVE_INST_VALU"] / wc, 3),
                    lds_active_frac=round(ca["SQ_ACTIVE_INST_LDS"] / wc, 3),
                    total_us=round(dur[fam] / 1e3),
                    mfma_util_est=round(cb.get("SQ_INSTS_MFMA", 0.0) * 16.0 / max(1.0, dur[fam] * 2.1 * 1024.0), 3),
                    insts_valu=cb.get("SQ_INSTS_VALU"), insts_mfma=cb.get("SQ_INSTS_MFMA"), insts_lds=cb.get("SQ_INSTS_LDS"),
                    insts_vmem=cb.get("SQ_INSTS_VMEM"), insts_salu=cb.get("SQ_INSTS_SALU"),
                    valu_per_mfma=round(cb.get("SQ_INSTS_VALU", 0.0) / cb["SQ_INSTS_MFMA"], 2) if cb.get("SQ_INSTS_MFMA") else None)
dst = os.path.join(root, "profiles", f"{rnd}_{tag}_sq_counters.json")
json.dump(out, open(dst, "w"), indent=1)
for k, v in out.items():
    print(f"{k:20s} launches {v['launches']:4d} active {v['active_frac']:.3f} wait_inst {v['wait_inst_frac']:.3f} parked {v['wait_any_frac']:.3f} "
          f"valu {v['valu_active_frac']:.3f} mfma_util {v['mfma_util_est']:.3f} valu/mfma {v['valu_per_mfma']}")
