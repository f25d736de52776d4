import os, sys
import torch
sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops
torch.manual_seed(0)
bf = torch.bfloat16
M, N, K = 70000, 768, 128
a = torch.randn(M, K, device="cuda", dtype=bf)
b = (torch.randn(N, K, device="cuda") * 0.2).to(bf)
ref = a.float() @ b.float().t()
out = ops.gemm(a, b).float()
print("max err", (out - ref).abs().max().item())
# which ref column does each out column match (first 64 cols, rows 0..255)?
r = ref[:256, :64]; o = out[:256, :64]
d = (o.t()[:, None, :] - r.t()[None, :, :]).abs().amax(-1)   # [out col, ref col]
print("col map :", d.argmin(1).tolist())
print("col err :", [round(x, 2) for x in d.amin(1).tolist()][:16])
d = (out[:64, :256][:, None, :] - ref[:64, :256][None, :, :]).abs().amax(-1)
print("row map :", d.argmin(1).tolist())
print("row err :", [round(x, 2) for x in d.amin(1).tolist()][:16])
