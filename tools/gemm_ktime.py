"""Per-tile time of the persistent bf16 GEMM against K (fixed overhead per tile + cost per 64-deep K-tile).  GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from multimodaldiscussiontransformer_amd import ops

M = 106496
bf = torch.bfloat16
for N in (768, 3072):
    pts = []
    for K in (256, 512, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda", dtype=bf)
        b = torch.randn(N, K, device="cuda", dtype=bf)
        out = torch.empty(M, N, device="cuda", dtype=bf)
        for _ in range(3):
            ops.gemm(a, b, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.gemm(a, b, out=out)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        tiles = (M // 256) * (N // 256)
        per_tile_us = us / (tiles / 256.0)
        pts.append((K, per_tile_us))
        print(f"N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  per tile {per_tile_us:6.2f} us ({per_tile_us / (K / 64):5.2f} us per K-tile)")
        del a, b, out
    (k0, t0), (k1, t1) = pts[0], pts[-1]
    slope = (t1 - t0) / ((k1 - k0) / 64)
    print(f"   fit: {slope:.3f} us per K-tile (MFMA alone: 0.97 us at 2.1 GHz), fixed {t0 - slope * k0 / 64:.2f} us per tile")
