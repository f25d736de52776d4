"""In-kernel cycles per K-tile and clock of the ping-pong GEMM (MDT_GEMM_STAMP=1 diagnostic). GPU box only.
Each shape is launched back to back; read the LAST line per shape (the chip has settled its clock by then)."""
import os
import sys

os.environ["MDT_GEMM_STAMP"] = "1"
import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402

bf = torch.bfloat16
M = 212992
shapes = [(M, 3072, 768, 0, 0, "ffn1 fwd NN"), (M, 768, 3072, 0, 0, "ffn2 fwd NN"), (M, 768, 3072, 0, 1, "ffn1 dgrad NT"),
          (M, 3072, 768, 0, 1, "ffn2 dgrad NT")]
if "--ffn1" in sys.argv:
    shapes = shapes[:1]
for (m, n, k, ta, tb, name) in shapes:
    a = torch.randn(m, k, device="cuda", dtype=bf)
    b = torch.randn(k, n, device="cuda", dtype=bf) if tb else torch.randn(n, k, device="cuda", dtype=bf)
    out = torch.empty(m, n, device="cuda", dtype=bf)
    print(f"== {name}", file=sys.stderr, flush=True)
    for _ in range(40):
        ops.gemm(a, b, trans_b=bool(tb), out=out)
    torch.cuda.synchronize()
if "--ffn1" in sys.argv:
    sys.exit(0)
dy = torch.randn(M, 3072, device="cuda", dtype=bf)
x = torch.randn(M, 768, device="cuda", dtype=bf)
c = torch.zeros(3072, 768, device="cuda", dtype=torch.float32)
print("== ffn1 wgrad TT split7", file=sys.stderr, flush=True)
for _ in range(40):
    ops.gemm(dy, x, trans_a=True, trans_b=True, out=c, epilogue=ops.EPI_ATOMIC, split_k=7)
torch.cuda.synchronize()
