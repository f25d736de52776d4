"""SURVEY.md §8f-3, host half of the image front end: the resize taps ``mdt_resize_plan`` computes (C ABI, host) make a plain
integer restatement of the two passes reproduce PIL's bilinear resize byte for byte — the reference's ViTImageProcessor call
(experiments/hateful_discussions/datasets/hateful_discussions.py:168-184) — and ``mdt_image_norm_lut`` the processor's
rescale / normalise arithmetic, against the fixture the installed processor produced (tests/golden/discussions/pixel_values.npz,
oracle/gen_golden.py case_pixel_values).  No GPU: the device passes are tests/test_image_gpu.py."""
import os

import numpy as np
import pytest

from oracle import cases


def two_pass(img, out, plan):
    H, W, _ = img.shape
    bh, kh = plan(W, out)
    bv, kv = plan(H, out)
    tmp = np.zeros((H, out, 3), np.uint8)
    for xo in range(out):
        x0, n = bh[xo]
        acc = (img[:, x0:x0 + n, :].astype(np.int64) * kh[xo, :n][None, :, None]).sum(1) + (1 << 21)
        tmp[:, xo, :] = np.clip(acc >> 22, 0, 255)
    o = np.zeros((out, out, 3), np.uint8)
    for yo in range(out):
        y0, n = bv[yo]
        acc = (tmp[y0:y0 + n].astype(np.int64) * kv[yo, :n][:, None, None]).sum(0) + (1 << 21)
        o[yo] = np.clip(acc >> 22, 0, 255)
    return o


def test_resize_plan_reproduces_the_fixture_bytes(golden_dir):
    from multimodaldiscussiontransformer_amd import ops
    g = np.load(os.path.join(golden_dir, "discussions", "pixel_values.npz"))
    imgs = cases.pixel_value_inputs(os.path.join(golden_dir, "discussions"))
    assert [list(a.shape[:2]) for a in imgs] == g["sizes"].tolist()
    for i, a in enumerate(imgs):
        assert np.array_equal(two_pass(a, 224, ops.resize_plan), g["resized"][i]), i
    # taps: an enlarged axis has 3 slots (2 live taps), a 1000 -> 224 reduction 11; every row of weights sums to 2^22 +- rounding
    assert ops.resize_plan(56, 224)[1].shape[1] == 3 and ops.resize_plan(1000, 224)[1].shape[1] == 11
    for size in (40, 224, 1000, 97):
        b, c = ops.resize_plan(size, 224)
        assert np.all(b[:, 0] >= 0) and np.all(b[:, 0] + b[:, 1] <= size) and np.all(b[:, 1] >= 1)
        assert np.all(np.abs(c.sum(1) - (1 << 22)) <= c.shape[1])


def test_resize_plan_against_pil_on_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    from multimodaldiscussiontransformer_amd import ops
    rng = np.random.default_rng(5)
    for _ in range(6):
        H, W = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((224, 224), resample=Image.BILINEAR))
        assert np.array_equal(two_pass(img, 224, ops.resize_plan), ref), (H, W)


def test_norm_lut_is_the_processors_arithmetic(golden_dir):
    from multimodaldiscussiontransformer_amd import ops
    g = np.load(os.path.join(golden_dir, "discussions", "pixel_values.npz"))
    lut = ops.image_norm_lut().reshape(3, 256)
    assert np.array_equal(lut, g["lut"])                       # bit-equal floats: (float)(u * (1 / 255)) - 0.5) / 0.5
    assert lut[0, 0] == -1.0 and lut[2, 255] == 1.0
    other = ops.image_norm_lut(1 / 255, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)).reshape(3, 256)
    want = (np.float32(np.float64(200) * (1 / 255)) - np.float32(0.456)) / np.float32(0.224)
    assert other[1, 200] == want


def test_packed_images_layout():
    from multimodaldiscussiontransformer_amd import ops
    rng = np.random.default_rng(2)
    imgs = [rng.integers(0, 256, s + (3,), dtype=np.uint8) for s in ((40, 56), (300, 500), (40, 56))]
    pk = ops.PackedImages(imgs, 224, pin=False)
    d = pk.desc.numpy()
    assert d[:, 2].tolist() == [40, 300, 40] and d[:, 3].tolist() == [56, 500, 56] and pk.max_h == 300
    assert d[0, 4] == d[2, 4] and d[0, 6] == d[2, 6]            # equal sizes share their tap tables
    assert pk.tmp_bytes == (40 + 300 + 40) * 224 * 3 and pk.pixels.numel() == sum(a.size for a in imgs)
    assert np.array_equal(pk.pixels.numpy()[d[1, 0]:d[1, 0] + imgs[1].size].reshape(300, 500, 3), imgs[1])
    with pytest.raises(Exception):
        ops.PackedImages([np.zeros((4, 4), dtype=np.uint8)])


def test_image_entry_points_validate_their_arguments():
    from multimodaldiscussiontransformer_amd import _lib as L
    b = np.zeros((224, 2), dtype=np.int32)
    c = np.zeros((224, 3), dtype=np.int32)
    assert L.lib.mdt_resize_plan_ksize(56, 224) == 3 and L.lib.mdt_resize_plan_ksize(0, 224) == 0
    assert L.lib.mdt_resize_plan(56, 224, b.ctypes.data, c.ctypes.data, 5) == -1          # wrong ksize
    assert b"ksize" in L.lib.mdt_last_error_string()
    assert L.lib.mdt_resize_plan(56, 224, None, c.ctypes.data, 3) == -1
    assert L.lib.mdt_image_norm_lut(1 / 255, None, None, None) == -1
    # device entry point: refused on the host side before any launch
    assert L.lib.mdt_image_preprocess(None, 1, 10, None, None, None, None, None, 0, None, None, 224) == -1
    assert b"null" in L.lib.mdt_last_error_string()
    assert L.lib.mdt_image_preprocess(None, 0, 0, None, None, None, None, None, 0, None, None, 224) == 0    # empty batch: nothing to do
