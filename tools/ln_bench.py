"""LayerNorm kernels at the training step's shapes: time per launch and bytes moved per second against the 8 TB/s of HBM3E
(MI355X_MICROARCH.md).  The four forms a step launches: forward; backward with the residual-path gradient added (x needs grad);
backward that also writes the hidden-dropout'ed gradient of the dense layer behind it and its column sums (the bias gradient) —
every post-LN block's two LayerNorms take that form.  GPU box only:  python tools/ln_bench.py [rows] [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100864
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    bf = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(rows, D, device="cuda", generator=g).to(bf)
    dy = torch.randn(rows, D, device="cuda", generator=g).to(bf)
    add = torch.randn(rows, D, device="cuda", generator=g).to(bf)
    gamma = (1 + 0.1 * torch.randn(D, device="cuda", generator=g)).to(bf)
    beta = (0.1 * torch.randn(D, device="cuda", generator=g)).to(bf)
    # a second set of tensors: alternate, so that a launch does not find its inputs in the 256 MB Infinity Cache from the one before
    x2, dy2, add2 = x.clone(), dy.clone(), add.clone()
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-12)
    dg, db, cs = (torch.zeros(D, dtype=torch.float32, device="cuda") for _ in range(3))
    el = rows * D * 2
    flip = [0]

    def pick():
        flip[0] ^= 1
        return (x, dy, add) if flip[0] else (x2, dy2, add2)

    # yardsticks: what the part gives the simplest streaming kernels at the same size (the vendor's copy and a three-stream add)
    t = timeit(lambda: y.copy_(pick()[0]))
    print(f"yardstick: torch copy_               rows {rows} D {D}: {t * 1e6:7.1f} us  {2 * el / t / 1e12:5.2f} TB/s = {2 * el / t / 8e12:4.2f} of HBM peak (1 read, 1 write)")
    t = timeit(lambda: torch.add(pick()[0], pick()[1], out=y))
    print(f"yardstick: torch add(out=)           rows {rows} D {D}: {t * 1e6:7.1f} us  {3 * el / t / 1e12:5.2f} TB/s = {3 * el / t / 8e12:4.2f} of HBM peak (2 reads, 1 write)")
    t = timeit(lambda: pick()[0].float().sum() if False else torch.sum(pick()[0], dtype=torch.float32))
    print(f"yardstick: torch sum (read only)     rows {rows} D {D}: {t * 1e6:7.1f} us  {el / t / 1e12:5.2f} TB/s = {el / t / 8e12:4.2f} of HBM peak (1 read)")
    t = timeit(lambda: ops.layernorm_fwd(pick()[0], gamma, beta, 1e-12, out=y))
    print(f"forward                              rows {rows} D {D}: {t * 1e6:7.1f} us  {2 * el / t / 1e12:5.2f} TB/s = {2 * el / t / 8e12:4.2f} of HBM peak (x read, y written)")
    dx = torch.empty_like(x)

    def bwd_plain():
        a, b, c = pick()
        ops.layernorm_bwd(b, a, gamma, mean, rstd, add=c, dgamma=dg, dbeta=db, dx=dx)
    t = timeit(bwd_plain)
    print(f"backward + residual gradient         rows {rows} D {D}: {t * 1e6:7.1f} us  {4 * el / t / 1e12:5.2f} TB/s = {4 * el / t / 8e12:4.2f} of HBM peak (dy, x, add read; dx written)")

    def bwd_tail():
        a, b, c = pick()
        ops.layernorm_bwd(b, a, gamma, mean, rstd, dgamma=dg, dbeta=db, dx=dx, drop_p=0.4, drop_seed=7, colsum=cs, want_dropped=True)
    t = timeit(bwd_tail)
    print(f"backward + dropped copy + column sums rows {rows} D {D}: {t * 1e6:7.1f} us  {4 * el / t / 1e12:5.2f} TB/s = {4 * el / t / 8e12:4.2f} of HBM peak (dy, x read; dx, dxd written)")

    def bwd_tail_add():
        a, b, c = pick()
        ops.layernorm_bwd(b, a, gamma, mean, rstd, add=c, dgamma=dg, dbeta=db, dx=dx, drop_p=0.4, drop_seed=7, colsum=cs, want_dropped=True)
    t = timeit(bwd_tail_add)
    print(f"backward + residual + dropped copy   rows {rows} D {D}: {t * 1e6:7.1f} us  {5 * el / t / 1e12:5.2f} TB/s = {5 * el / t / 8e12:4.2f} of HBM peak (dy, x, add read; dx, dxd written)")


if __name__ == "__main__":
    main()
