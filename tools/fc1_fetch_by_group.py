"""fc1 forward (M x 3072 x 768, bias + GELU + saved derivative: the 8-wave kernel) under different column-group widths of its tile walk:
fabric reads per launch (FETCH_SIZE) next to the launch time — does fetching the operands fewer times make the launch faster?
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fc1_fetch -- python3 tools/fc1_fetch_by_group.py
  python tools/fc1_fetch_by_group.py --report gpurun_out/fc1_fetch
Group g = the tile order sweeps g column tiles of a 256-row panel before moving down the rows; default: 6 of the 12 (XCDs 0-3 the left
half of W, XCDs 4-7 the right one).  FETCH_SIZE KB x 1024 x 2 on gfx950 (MI355X_MICROARCH.md)."""
import csv
import glob
import os
import sys

M, N, K = 106496, 3072, 768
GROUPS = [0, 12, 6, 4, 3, 2, 1]          # 0 = the library's own choice
REPS = 3


def run():
    import torch
    sys.path.insert(0, ".")
    from multimodaldiscussiontransformer_amd import _lib, ops
    bf = torch.bfloat16
    a = torch.randn(M, K, device="cuda", dtype=bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    b = torch.randn(N, device="cuda", dtype=bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    aux = torch.empty(M, N, device="cuda", dtype=bf)
    for g in GROUPS:
        if g:
            os.environ["MDT_GEMM_GROUP"] = str(g)
        else:
            os.environ.pop("MDT_GEMM_GROUP", None)
        _lib.reload_env()
        for _ in range(REPS):
            ops.gemm(a, w, bias=b, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)
        torch.cuda.synchronize()


def report(d):
    cc = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(cc)) if "pp256p" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    kt = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "pp256p" in r["Kernel_Name"]]
    algo = (M * K + N * K) * 2
    print(f"fc1 forward M {M} N {N} K {K}: operands A + W = {algo / 1e6:.0f} MB, outputs 2 x {M * N * 2 / 1e6:.0f} MB; {len(rows)} launches")
    for i, g in enumerate(GROUPS):
        f = [float(r["Counter_Value"]) * 1024 * 2 for r in rows[i * REPS:(i + 1) * REPS]]
        t = dur[i * REPS:(i + 1) * REPS]
        if not f:
            continue
        print(f"  group {g if g else 'default'}: fetched {min(f) / 1e6:7.0f} MB per launch = {min(f) / algo:4.1f} x (A + W); launch {min(t) / 1e3:7.1f} us (under the counter pass)")


if __name__ == "__main__":
    if "--report" in sys.argv:
        report(sys.argv[sys.argv.index("--report") + 1])
    else:
        run()
