"""Pin the CPU oracle against golden vectors produced by the reference's own torch code
(tests/golden/gen_golden.py, run in the build container).  CPU only."""
import pytest
import torch

import oracle
from conftest import ulp_diff

CASES = ["c0_", "c1_", "c2_"]


def test_priors_bit_exact(golden_priors):
    for k, ref in golden_priors.items():
        h, w = [int(v) for v in k[2:].split("x")]
        assert torch.equal(oracle.make_priors(h, w)[0], ref), k


def test_priors_reduced_pyramid(golden_postproc):
    levels = golden_postproc["levels"].tolist()
    pri = torch.cat([oracle.make_priors(h, w) for h, w in levels], 1)[0]
    assert torch.equal(pri, golden_postproc["priors"])


@pytest.mark.parametrize("p", CASES)
def test_decode_within_1ulp_of_reference(golden_postproc, p):
    """box_utils.py:238-283.  torch.exp on the reference CPU path is MKL VML (not reproducible op for op);
    the canonical exp is correctly rounded, hence <= 1 ULP on w/h and a few ULP after the subtraction."""
    g = golden_postproc
    got = oracle.decode(g[p + "loc"], g["priors"])
    ref = g[p + "boxes"]
    exact = (got == ref).float().mean().item()
    assert exact > 0.97, exact
    # x1 = cx - w/2 can cancel, so bound the error by 1 ULP of the box size plus 1 ULP of the coordinate
    wh = (ref[:, 2:] - ref[:, :2]).abs().repeat(1, 2)
    assert ((got - ref).abs() <= 1.2e-7 * (wh + ref.abs()) + 1e-9).all()
    # the w/h themselves (exp * prior) are within 1 ULP
    assert ulp_diff(got[:, 2:] - got[:, :2], ref[:, 2:] - ref[:, :2]).max() <= 4


def test_expf_is_correctly_rounded():
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(3_000_000, generator=g) * 2 - 1) * 12
    assert torch.equal(oracle.expf(x), torch.exp(x.double()).float())
    edge = torch.tensor([0.0, -0.0, 1.0, -1.0, 88.0, -87.0, -103.0, 1e-10, -1e-10])
    assert torch.equal(oracle.expf(edge), torch.exp(edge.double()).float())


@pytest.mark.parametrize("p", CASES)
def test_center_size_jaccard_sanitize_exact(golden_postproc, p):
    g = golden_postproc
    assert torch.equal(oracle.center_size(g[p + "boxes"]), g[p + "center_size"])
    cb = g[p + "cand_box"]
    assert torch.equal(oracle.jaccard(cb[:64], cb[:96]), g[p + "jaccard"])
    assert torch.equal(oracle.sanitize_hw(g[p + "cc_box"], 24, 40), g[p + "sanitize_hw"])


@pytest.mark.parametrize("p", CASES)
def test_candidate_filter_exact(golden_postproc, p):
    g = golden_postproc
    keep = oracle.candidate_filter(g[p + "conf"], 0.05)
    assert torch.equal(keep, g[p + "keep_idx"])
    assert torch.equal(g[p + "conf"][keep], g[p + "cand_conf"])


@pytest.mark.parametrize("p", CASES)
def test_cc_fast_nms_bit_exact(golden_postproc, p):
    """detection_TF.py:85-134 on the reference's own candidate rows."""
    g = golden_postproc
    idx, cls, sc = oracle.cc_fast_nms(g[p + "cand_conf"], g[p + "cand_box"], g[p + "cand_centerness"], 0.5, 200)
    assert len(idx) == len(g[p + "cc_class"]) and len(idx) > 0
    assert torch.equal(g[p + "cand_box"][idx], g[p + "cc_box"])
    assert torch.equal(cls, g[p + "cc_class"])
    assert torch.equal(sc, g[p + "cc_score"])
    assert torch.equal(g[p + "cand_centerness"][idx], g[p + "cc_centerness"])


@pytest.mark.parametrize("p", CASES)
def test_per_class_fast_nms_bit_exact(golden_postproc, p):
    """detection_TF.py:136-204"""
    g = golden_postproc
    idx, cls, sc = oracle.fast_nms(g[p + "cand_conf"], g[p + "cand_box"], g[p + "cand_centerness"], 0.5, 200, 0.05,
                                   100)
    assert len(idx) == len(g[p + "pc_class"]) and len(idx) > 0
    assert torch.equal(sc, g[p + "pc_score"])
    assert torch.equal(cls, g[p + "pc_class"])
    assert torch.equal(g[p + "cand_box"][idx], g[p + "pc_box"])


@pytest.mark.parametrize("p", CASES)
def test_per_class_fast_nms_non_tf_ranks_without_centerness(golden_postproc, p):
    """detection.py:211-261 (row a18): the non-TF Detect.fast_nms takes no centerness -- goldens from the reference's own method
    (tests/golden/gen_golden.py postproc_nontf), and the host mirror's Detect must go the same way."""
    from conftest import load_golden
    g, gn = golden_postproc, load_golden("postproc_nontf.npz")
    idx, cls, sc = oracle.fast_nms(g[p + "cand_conf"], g[p + "cand_box"], None, 0.5, 200, 0.05, 100)
    assert len(idx) == len(gn[p + "pcn_class"]) and len(idx) > 0
    assert torch.equal(sc, gn[p + "pcn_score"]) and torch.equal(cls, gn[p + "pcn_class"])
    assert torch.equal(g[p + "cand_box"][idx], gn[p + "pcn_box"])
    assert not torch.equal(sc[:8], g[p + "pc_score"][:8])             # (the Detect_TF variant scores differently: the two goldens are not the same test)


@pytest.mark.parametrize("p", CASES)
def test_generate_mask_within_tolerance(golden_postproc, p):
    """mask_utils.py:111-128; tolerance 1e-5 abs (reference matmul is fp32 MKL, oracle accumulates in double)."""
    g = golden_postproc
    got = oracle.generate_mask(g[p + "proto"], g[p + "cc_mask_coeff"], g[p + "cc_box"])
    ref = g[p + "masks"]
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 1e-5
    assert torch.equal(got == 0, ref == 0)  # crop region identical
    got2 = oracle.generate_mask(g[p + "proto"], g[p + "cc_mask_coeff"][:5], None)
    assert (got2 - g[p + "masks_nocrop"]).abs().max() < 1e-5


@pytest.mark.parametrize("p", CASES)
def test_mask_iou_exact(golden_postproc, p):
    g = golden_postproc
    m = g[p + "masks"]
    got = oracle.mask_iou(m[: min(20, len(m))], m, 0.5)
    assert torch.equal(got, g[p + "mask_iou"])


def test_crop_boundaries(golden_postproc):
    """box_utils.py:341-364: x1-1 / x2+1 float bounds, swapped corners, boxes touching 0 and 1."""
    g = golden_postproc
    ones = torch.ones(24, 40, 32)
    coeff = torch.full((5, 32), 10.0)  # sigmoid(32*tanh(10)) == 1.0f
    got = oracle.generate_mask(ones, coeff, g["crop_boxes"])
    assert torch.equal(got, g["crop_mask"])


def test_fcb_ali_offsets(golden_fcb_ali):
    """Featurealign.py:46-69.  The dy/dx parts are exact; the exp-dependent part is <= 1 ULP of exp."""
    g = golden_fcb_ali
    for kh, kw in [(3, 3), (3, 5), (5, 3)]:
        got = oracle.fcb_ali_offsets(g["loc"], kh, kw)
        ref = g[f"off_{kh}x{kw}"]
        assert got.shape == ref.shape
        assert (got - ref).abs().max() <= 4e-7 * ref.abs().max()
        assert (got == ref).float().mean() > 0.9


def test_nms_tie_rule_is_stable_descending():
    """torch.sort (detection_TF.py:93) is unstable; canonical rule = lower original index first."""
    boxes = torch.tensor([[0.1, 0.1, 0.3, 0.3], [0.6, 0.6, 0.9, 0.9], [0.1, 0.1, 0.3, 0.3], [0.4, 0.1, 0.5, 0.2]])
    conf = torch.zeros(4, 41)
    conf[:, 3] = 0.5  # four-way tie
    idx, cls, sc = oracle.cc_fast_nms(conf, boxes, None, 0.5, 200)
    assert idx.tolist() == [0, 1, 3] and cls.tolist() == [3, 3, 3]


def test_fast_nms_chain_differs_from_greedy():
    """Fast NMS lets a suppressed box still suppress others (A>B>C: greedy keeps C, Fast NMS drops it)."""
    A = [0.10, 0.10, 0.50, 0.50]
    B = [0.22, 0.10, 0.62, 0.50]   # IoU(A,B) = 0.538
    C = [0.34, 0.10, 0.74, 0.50]   # IoU(B,C) = 0.538, IoU(A,C) = 0.25
    boxes = torch.tensor([A, B, C])
    conf = torch.zeros(3, 41)
    conf[:, 1] = torch.tensor([0.9, 0.8, 0.7])
    idx, _, _ = oracle.cc_fast_nms(conf, boxes, None, 0.5, 200)
    assert idx.tolist() == [0]
    # degenerate zero-area duplicates give NaN IoU -> dropped, like torch.max NaN propagation
    z = torch.tensor([[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]])
    c2 = torch.zeros(2, 41)
    c2[:, 1] = torch.tensor([0.9, 0.8])
    idx, _, _ = oracle.cc_fast_nms(c2, z, None, 0.5, 200)
    assert idx.tolist() == [0]
