"""Output stage ("next" row f1): mask resize + COCO RLE.  CPU tests pin the oracle (round trips, hand vectors, ATen's
bilinear resize); GPU tests compare the device kernels and the postprocess_ytbvis mirror with it."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from stmask_amd.output_utils import rle_counts_to_string


def test_rle_hand_vectors_and_round_trip():
    img = torch.tensor([[0, 1, 1], [1, 1, 0]], dtype=torch.uint8)       # column-major: 0 1 1 1 1 0
    assert oracle.rle_encode(img).tolist() == [1, 4, 1]
    assert oracle.rle_to_string(oracle.rle_encode(img)) == b"141"
    assert oracle.rle_encode(torch.ones(2, 2, dtype=torch.uint8)).tolist() == [0, 4]   # starts with an empty zero run
    assert oracle.rle_encode(torch.zeros(3, 2, dtype=torch.uint8)).tolist() == [6]
    g = torch.Generator().manual_seed(1)
    for h, w in [(7, 5), (90, 160), (64, 64)]:
        img = (torch.rand(h, w, generator=g) > 0.6).to(torch.uint8)
        c = oracle.rle_encode(img)
        assert int(c.sum()) == h * w
        s = oracle.rle_to_string(c)
        assert torch.equal(oracle.rle_from_string(s), c)                 # string packing round trip (deltas, signs)
        assert torch.equal(oracle.rle_decode(c, h, w), img)
        assert rle_counts_to_string(c.tolist()) == s                     # host packer of the product == oracle
    big = torch.tensor([0, 100000, 3, 70000, 1, 2, 900000], dtype=torch.int64)   # multi-chunk counts, negative deltas
    assert torch.equal(oracle.rle_from_string(oracle.rle_to_string(big)), big)
    assert rle_counts_to_string(big.tolist()) == oracle.rle_to_string(big)


def test_library_host_rle_string_packer_equals_oracle():
    """stm_rle_strings_host (the library's host-side maskApi rleToString for a batch of masks: what output_utils.encode_masks calls) against the
    oracle's string packer on ragged rows: empty rows, zero first runs, multi-chunk counts, negative deltas, a row that fills its width."""
    from stmask_amd.output_utils import rle_strings
    g = torch.Generator().manual_seed(5)
    rows = [[], [6], [0, 4], [1, 4, 1], [0, 100000, 3, 70000, 1, 2, 900000]]
    for _ in range(40):
        k = int(torch.randint(1, 60, (1,), generator=g))
        rows.append(torch.randint(0, 3000, (k,), generator=g).tolist())
    width = max(len(r) for r in rows)
    counts = torch.zeros(len(rows), width, dtype=torch.int32)
    for i, r in enumerate(rows):
        counts[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    got = rle_strings(counts, torch.tensor([len(r) for r in rows], dtype=torch.int32))
    for i, r in enumerate(rows):
        assert got[i] == oracle.rle_to_string(torch.tensor(r, dtype=torch.int64)), i


def test_resize_threshold_matches_aten_bilinear():
    g = torch.Generator().manual_seed(2)
    for (mh, mw, ch, cw, oh, ow) in [(24, 40, 22, 40, 90, 160), (96, 160, 90, 160, 360, 640), (12, 20, 12, 20, 50, 37)]:
        m = torch.sigmoid(torch.randn(mh, mw, generator=g) * 3)
        got = oracle.mask_resize_threshold(m, ch, cw, oh, ow)
        ref = (F.interpolate(m[None, None, :ch, :cw], (oh, ow), mode="bilinear", align_corners=False)[0, 0] > 0.5)
        assert (got == ref.to(torch.uint8)).float().mean() > 0.9995      # only pixels within rounding of 0.5 may differ


@pytest.mark.gpu
def test_device_resize_rle_equals_oracle():
    from stmask_amd import ops
    g = torch.Generator().manual_seed(3)
    for (n, mh, mw, ch, cw, oh, ow) in [(5, 24, 40, 22, 40, 90, 160), (3, 96, 160, 90, 160, 360, 640), (2, 96, 160, 96, 160, 720, 1280),
                                         (4, 12, 20, 12, 20, 50, 37)]:
        m = torch.sigmoid(torch.randn(n, mh, mw, generator=g) * 3)
        m[0] = 0.0                                                       # empty mask -> one run
        if n > 1:
            m[1] = 1.0                                                   # full mask -> [0, h*w]
        counts, nr = ops.mask_resize_rle(m.cuda(), ch, cw, oh, ow, 0.5, max_runs=oh * ow + 1)
        for i in range(n):
            ref = oracle.rle_encode(oracle.mask_resize_threshold(m[i], ch, cw, oh, ow))
            k = int(nr[i])
            assert k == len(ref), (i, k, len(ref))
            assert torch.equal(counts[i, :k].cpu().to(torch.int64), ref)
    # overflow is reported, not silently truncated
    noisy = (torch.rand(1, 24, 40, generator=g) > 0.5).float()
    counts, nr = ops.mask_resize_rle(noisy.cuda(), 24, 40, 24, 40, 0.5, max_runs=16)
    assert int(nr[0]) > 16


@pytest.mark.gpu
def test_postprocess_ytbvis_mirror():
    from stmask_amd.output_utils import postprocess_ytbvis
    g = torch.Generator().manual_seed(4)
    n = 6
    det = {"box": torch.rand(n, 4, generator=g).sort(1)[0][:, [0, 1, 2, 3]].cuda(), "score": torch.tensor([.9, .8, .01, .7, .6, .5]).cuda(),
           "class": torch.arange(1, n + 1).cuda(), "mask": torch.sigmoid(torch.randn(n, 96, 160, generator=g) * 3).cuda(),
           "mask_coeff": torch.randn(n, 32, generator=g).cuda(), "box_ids": torch.arange(n).cuda(),
           "proto": torch.zeros(96, 160, 32).cuda()}
    meta = {"ori_shape": (720, 1280, 3), "img_shape": (360, 640, 3), "pad_shape": (384, 640, 3)}
    out = postprocess_ytbvis({"detection": det, "net": None}, meta, score_threshold=0.05)
    s_h = 360 / 384
    c = (det["box"][:, :2] + det["box"][:, 2:]) / 2
    keep = ((det["score"] > 0.05) & ~((c[:, 0] > 1.0) | (c[:, 1] > s_h))).cpu()
    assert len(out["segm"]) == int(keep.sum()) and out["box"].dtype == torch.int64
    kept = torch.nonzero(keep).view(-1)
    for j, i in enumerate(kept.tolist()):
        ref_bits = oracle.mask_resize_threshold(det["mask"][i].cpu(), int(s_h * 96), 160, 720, 1280)
        assert out["segm"][j]["size"] == [720, 1280]
        assert out["segm"][j]["counts"] == oracle.rle_to_string(oracle.rle_encode(ref_bits))
        assert torch.equal(oracle.rle_decode(oracle.rle_from_string(out["segm"][j]["counts"]), 720, 1280), ref_bits)
    # boxes: rescaled by the padding ratio, clamped to the original frame, truncated to integers
    b = det["box"][kept.cuda()].clone()
    b[:, 1::2] /= s_h
    assert (out["box"][:, 0] == (b[:, 0] * 1280).clamp(min=0).long()).all()
    assert (out["box"][:, 3] == (b[:, 3] * 720).clamp(max=720).long()).all()
