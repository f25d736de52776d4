"""A/B of the attention backward variants with and without dropout (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

bf = torch.bfloat16
for (nseq, S, name) in [(2048, 104, "bert"), (512, 201, "vit")]:
    H, hd = 12, 64
    qkv = torch.randn(nseq * S, 3 * H * hd, device="cuda", dtype=bf)
    dout = torch.randn(nseq * S, H * hd, device="cuda", dtype=bf)
    for p in (0.0, 0.3):
        out, lse = ops.attention_fwd(qkv, nseq, S, H, drop_p=p, drop_seed=5)
        t = timeit(lambda: ops.attention_fwd(qkv, nseq, S, H, drop_p=p, drop_seed=5))
        line = f"{name} p={p}: fwd {t*1e3:.3f} ms |"
        for v in ("v1", "v2", "v3"):
            os.environ["MDT_ATTN_BWD"] = v
            __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
            t = timeit(lambda: ops.attention_bwd(dout, qkv, out, lse, nseq, S, H, drop_p=p, drop_seed=5))
            line += f" bwd {v} {t*1e3:.3f} ms"
        os.environ.pop("MDT_ATTN_BWD")
        __import__("multimodaldiscussiontransformer_amd._lib", fromlist=["x"]).reload_env()
        print(line, flush=True)
