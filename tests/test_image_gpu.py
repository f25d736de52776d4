"""SURVEY.md §8f-3, device half of the image front end (``mdt_image_preprocess``, csrc/image.hip): decoded RGB bytes of different
sizes → [n, 3, 224, 224] pixel tensors, against what the reference's ViTImageProcessor call produced for the same images
(tests/golden/discussions/pixel_values.npz: resized bytes + value table, oracle/gen_golden.py case_pixel_values) — byte- and
bit-exact — and through the dataset front end + packer into the model's image branch."""
import os

import numpy as np
import pytest
import torch

from oracle import cases

pytestmark = pytest.mark.gpu


def test_image_preprocess_matches_the_processor_fixture(golden_dir):
    from multimodaldiscussiontransformer_amd import ops
    g = np.load(os.path.join(golden_dir, "discussions", "pixel_values.npz"))
    imgs = cases.pixel_value_inputs(os.path.join(golden_dir, "discussions"))
    pk = ops.PackedImages(imgs, 224)
    out, u8 = ops.image_preprocess(pk, return_bytes=True)
    torch.cuda.synchronize()
    assert np.array_equal(u8.cpu().numpy(), g["resized"])                       # PIL's bytes
    want = np.stack([np.stack([g["lut"][c][g["resized"][i, :, :, c]] for c in range(3)]) for i in range(len(imgs))])
    assert np.array_equal(out.cpu().numpy(), want)                              # = pixel_values of the processor, bit for bit
    assert abs(float(out.double().sum()) - float(g["checksum"][0])) < 1e-6 * want.size
    bf = ops.image_preprocess(pk, dtype=torch.bfloat16)
    assert torch.equal(bf.cpu(), torch.from_numpy(want).bfloat16())


def test_image_preprocess_against_pil_many_sizes():
    Image = pytest.importorskip("PIL.Image")
    from multimodaldiscussiontransformer_amd import ops
    rng = np.random.default_rng(11)
    sizes = [(1, 1), (2, 3), (223, 225), (224, 224), (500, 375), (1200, 900), (37, 1500), (1500, 41)] + \
            [(int(rng.integers(1, 900)), int(rng.integers(1, 900))) for _ in range(24)]
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
    _, u8 = ops.image_preprocess(ops.PackedImages(imgs, 224), return_bytes=True)
    got = u8.cpu().numpy()
    for i, a in enumerate(imgs):
        ref = np.asarray(Image.fromarray(a).resize((224, 224), resample=Image.BILINEAR))
        assert np.array_equal(got[i], ref), sizes[i]
    # another target size (ViT-L/14 uses 224 too; 384-px checkpoints exist)
    _, u8 = ops.image_preprocess(ops.PackedImages(imgs[:6], 384), return_bytes=True)
    for i in range(6):
        assert np.array_equal(u8[i].cpu().numpy(), np.asarray(Image.fromarray(imgs[i]).resize((384, 384), resample=Image.BILINEAR))), sizes[i]


def test_discussions_front_end_on_device_equals_host_path(golden_dir):
    """data/discussions.py with image_preprocess="device": trees carry decoded bytes, the packer uploads them as uint8 and
    runs the kernel; x_images equals what the host (PIL + numpy) path uploads as fp32."""
    from multimodaldiscussiontransformer_amd.data import discussions as D
    from multimodaldiscussiontransformer_amd.data.packer import pack_batch
    root = os.path.join(golden_dir, "discussions")
    tok = D.default_tokenizer(os.path.join(root, "vocab.txt"))
    host = D.HatefulDiscussions(os.path.join(root, "sample.jsonl"), tokenizer=tok, max_length=24, image_root=root, image_size=224)
    dev = D.HatefulDiscussions(os.path.join(root, "sample.jsonl"), tokenizer=tok, max_length=24, image_root=root, image_size=224,
                               image_preprocess="device")
    th = [host[i] for i in range(len(host))]
    td = [dev[i] for i in range(len(dev))]
    th = [t for t in th if t is not None]
    td = [t for t in td if t is not None]
    assert any(t.get("images_u8") for t in td) and all(t["images"] is None for t in td)
    a = pack_batch(th, 5)
    b = pack_batch(td, 5)
    assert a.images is not None and torch.equal(a.images, b.images)
    assert torch.equal(a.batched_data["x_image_indexes"], b.batched_data["x_image_indexes"])
