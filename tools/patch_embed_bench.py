"""ViT patch embedding at the bench step's shape (512 images of 224 x 224, ViT-B/16): the one-launch kernel (csrc/patch_embed.hip,
SURVEY K8) against the three launches it replaces (vit_patchify + gemm + vit_assemble).  GPU box only:
python tools/patch_embed_bench.py [images]"""
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
    return s.elapsed_time(e) / iters * 1e3


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    bf = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    img = torch.randn(I, 3, 224, 224, device="cuda", generator=g)
    w = (torch.randn(768, 768, device="cuda", generator=g) * 0.03).to(bf)
    b, cls = torch.randn(768, device="cuda", generator=g).to(bf), torch.randn(768, device="cuda", generator=g).to(bf)
    pos = torch.randn(197, 768, device="cuda", generator=g).to(bf)
    tok = torch.empty(I * 197, 768, dtype=bf, device="cuda")
    tok3 = torch.empty_like(tok)

    def three():
        cols = ops.vit_patchify(img, 16, bf)
        patches = ops.gemm(cols, w, bias=b)
        ops.vit_assemble(patches, cls, pos, tok3, I, 196, seq_stride=197, off=0)

    t1 = timeit(lambda: ops.vit_patch_embed(img, 16, w, b, cls, pos, tok, seq_stride=197, off=0))
    t3 = timeit(three)
    fl = 2.0 * I * 196 * 768 * 768
    by = img.numel() * 4 + tok.numel() * 2
    print(f"images {I}: one launch {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TFLOP/s, {by / t1 / 1e6:5.2f} TB/s of pixels in + tokens out); "
          f"three launches {t3:7.1f} us; max |diff| {float((tok.float() - tok3.float()).abs().max()):.3e}")


if __name__ == "__main__":
    main()
