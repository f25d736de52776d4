"""Every GEMM shape of one mDT-base block (forward, dgrad, wgrad), two launches each, for a fabric-traffic pass:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_shapes -- python3 tools/gemm_pmc_shapes.py
  python tools/gemm_pmc_shapes.py --report gpurun_out/pmc_shapes
The report pairs the dispatches (in order) with the shapes and prints fetched bytes (FETCH_SIZE KB x 1024 x 2 on gfx950)
against the operand bytes of the launch."""
import csv
import glob
import json
import sys

M = 106496
SHAPES = [  # name, N, K, trans_a, trans_b, extra operand reads (x M x N x 2 B), fp32 split-K output
    ("fwd qkv", 2304, 768, 0, 0, 0, 0), ("fwd o + residual", 768, 768, 0, 0, 1, 0), ("fwd fc1", 3072, 768, 0, 0, 0, 0),
    ("fwd fc2 + residual", 768, 3072, 0, 0, 1, 0),
    ("dgrad qkv", 768, 2304, 0, 1, 0, 0), ("dgrad o", 768, 768, 0, 1, 0, 0), ("dgrad fc1", 768, 3072, 0, 1, 0, 0),
    ("dgrad fc2", 3072, 768, 0, 1, 0, 0),
    ("wgrad qkv", 2304, 768, 1, 1, 0, 1), ("wgrad o", 768, 768, 1, 1, 0, 1), ("wgrad fc1", 3072, 768, 1, 1, 0, 1),
    ("wgrad fc2", 768, 3072, 1, 1, 0, 1),
]
REPS = 2


def run():
    import torch
    sys.path.insert(0, ".")
    from multimodaldiscussiontransformer_amd import ops
    from multimodaldiscussiontransformer_amd.engine import _split_k
    bf = torch.bfloat16
    for name, n, k, ta, tb, extra, wg in SHAPES:
        if wg:      # dW[n, k] += dY[M, n]^T X[M, k]
            dy = torch.randn(M, n, device="cuda", dtype=bf)
            x = torch.randn(M, k, device="cuda", dtype=bf)
            c = torch.zeros(n, k, device="cuda", dtype=torch.float32)
            for _ in range(REPS):
                ops.gemm(dy, x, trans_a=True, trans_b=True, out=c, epilogue=ops.EPI_ATOMIC, split_k=_split_k(n, k, M))
        else:
            a = torch.randn(M, k, device="cuda", dtype=bf)
            b = torch.randn(k, n, device="cuda", dtype=bf) if tb else torch.randn(n, k, device="cuda", dtype=bf)
            out = torch.empty(M, n, device="cuda", dtype=bf)
            res = torch.randn(M, n, device="cuda", dtype=bf) if extra else None
            for _ in range(REPS):
                ops.gemm(a, b, trans_b=bool(tb), out=out, residual=res)
        torch.cuda.synchronize()


def report(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == len(SHAPES) * REPS, (len(rows), len(SHAPES) * REPS)
    out = []
    for i, (name, n, k, ta, tb, extra, wg) in enumerate(SHAPES):
        got = sum(float(r["Counter_Value"]) for r in rows[i * REPS:(i + 1) * REPS]) / REPS * 1024 * 2
        need = 2.0 * (M * n + M * k) if wg else 2.0 * (M * k + n * k) + extra * 2.0 * M * n
        out.append(dict(shape=name, M=M, N=n, K=k, operand_MB=round(need / 1e6), fetched_MB=round(got / 1e6), ratio=round(got / need, 2),
                        kernel=rows[i * REPS]["Kernel_Name"][:60]))
        print(f"{name:20s} N={n:5d} K={k:5d} operands {need / 1e6:7.0f} MB  fetched {got / 1e6:7.0f} MB  x{got / need:.2f}")
    json.dump(out, open(d + "/report.json", "w"), indent=1)


if __name__ == "__main__":
    if "--report" in sys.argv:
        report(sys.argv[sys.argv.index("--report") + 1])
    else:
        run()
