"""gemm_bf16_solo (4 waves on 128 x 256 tiles, two workgroups per CU: csrc/gemm_solo.hip) against the kernels the step uses today, at
the step's shapes, one call: time per launch and a bit comparison of every output.  GPU box only: python tools/solo_ab.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaldiscussiontransformer_amd import _lib, ops  # noqa: E402

bf = torch.bfloat16


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 106496 + 37
    g = torch.Generator(device="cuda").manual_seed(0)
    cases = [("fc1 forward (GELU + saved derivative)", 3072, 768, "fc1"), ("qkv forward (bias)", 2304, 768, "bias"),
             ("o-proj forward (bias + dropout + residual)", 768, 768, "dense"), ("fc2 forward (bias + dropout + residual)", 768, 3072, "dense")]
    for name, N, K, kind in cases:
        a = torch.randn(M, K, device="cuda", generator=g).to(bf)
        w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(bf)
        b = torch.randn(N, device="cuda", generator=g).to(bf)
        res = torch.randn(M, N, device="cuda", generator=g).to(bf) if kind == "dense" else None
        outs = {}
        line = f"{name:46s} M {M} N {N} K {K}:"
        for solo, skew in ((0, 0), (2, 0), (2, -1), (2, 150), (2, 600), (0, 0)):
            os.environ["MDT_GEMM_SOLO"] = str(solo)
            os.environ["MDT_GEMM_SOLO_SKEW"] = str(skew)
            _lib.reload_env()
            out = torch.empty(M, N, dtype=bf, device="cuda")
            aux = torch.empty(M, N, dtype=bf, device="cuda") if kind == "fc1" else None

            def run():
                if kind == "fc1":
                    ops.gemm(a, w, bias=b, aux=aux, out=out, epilogue=ops.EPI_GELU | ops.EPI_AUX_GRAD)
                elif kind == "bias":
                    ops.gemm(a, w, bias=b, out=out)
                else:
                    ops.gemm(a, w, bias=b, residual=res, out=out, drop_p=0.4, drop_seed=99)
            t = timeit(run)
            line += f"  {('solo skew ' + str(skew)) if solo else 'now'} {t:6.1f} us"
            outs.setdefault(solo, (out.clone(), None if aux is None else aux.clone()))
        same = torch.equal(outs[0][0].view(torch.int16), outs[2][0].view(torch.int16)) and (outs[0][1] is None or torch.equal(outs[0][1].view(torch.int16), outs[2][1].view(torch.int16)))
        print(line + f"  outputs {'bit-identical' if same else 'DIFFER'}", flush=True)
    os.environ.pop("MDT_GEMM_SOLO", None)
    os.environ.pop("MDT_GEMM_SOLO_SKEW", None)
    _lib.reload_env()


if __name__ == "__main__":
    main()
