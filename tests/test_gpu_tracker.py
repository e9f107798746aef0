"""The tracker-bookkeeping kernels of the batched pipeline (stmask_amd/csrc/tracker.hip) against the torch op chains they
replace (kept as the CPU mirror in oracle/cpu_path.py -- the same chains the reference-golden CPU tests of the pipeline run
on): bit-equal outputs on seeded random tables with ragged clips (empty clips, clips without detections, ties)."""
import pytest
import torch

from oracle import cpu_path as ref
from stmask_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tables(seed, B=5, top_k=20, N=300, mdim=32, edim=128, counts=(3, 0, 20, 7, 1), prev_n=(4, 6, 0, 9, 2), hw=(24, 40)):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    idx = torch.stack([torch.randperm(N, generator=g)[:top_k] for _ in range(B)])
    cls = torch.randint(1, 41, (B, top_k), generator=g)
    score = torch.rand(B, top_k, generator=g)
    c = torch.rand(B, top_k, 2, generator=g)
    wh = torch.rand(B, top_k, 2, generator=g) * 0.4 + 0.02
    box = torch.cat([c - wh / 2, c + wh / 2], -1)
    cnt = torch.tensor(counts, dtype=torch.int32)
    coeff, track, cen = r(B, N, mdim), torch.nn.functional.normalize(r(B, N, edim), dim=-1), torch.tanh(r(B, N, 1))
    Pn = sum(prev_n)
    pc = torch.rand(Pn, 2, generator=g)
    pwh = torch.rand(Pn, 2, generator=g) * 0.4 + 0.02
    prev = {"box": torch.cat([pc - pwh / 2, pc + pwh / 2], -1), "mask_coeff": r(Pn, mdim), "track": torch.nn.functional.normalize(r(Pn, edim), dim=-1),
            "class": torch.randint(1, 41, (Pn,), generator=g), "score": torch.rand(Pn, generator=g), "centerness": torch.rand(Pn, generator=g),
            "mask": torch.rand(Pn, *hw, generator=g) * (torch.rand(Pn, 1, 1, generator=g) > 0.3),
            "clip": torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(prev_n))}
    off = torch.tensor([0] + list(torch.tensor(prev_n).cumsum(0)), dtype=torch.int32)
    return dict(idx=idx, cls=cls, score=score, box=box, cnt=cnt, coeff=coeff, track=track, cen=cen, prev=prev, off=off, B=B, top_k=top_k, Pn=Pn)


def _d(t):
    return t.to(DEV) if torch.is_tensor(t) else t


def test_gather_detections_equals_masked_index_selects():
    t = _tables(1)
    D = int(t["cnt"].sum())
    got = ops.gather_detections(_d(t["idx"]), _d(t["cls"]), _d(t["score"]), _d(t["box"]), _d(t["cnt"]), _d(t["coeff"]), _d(t["track"]), _d(t["cen"]), D)
    want = ref._gather_detections(t["idx"], t["cls"], t["score"], t["box"], t["cnt"], t["coeff"], t["track"], t["cen"], D)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and torch.equal(got[k].cpu(), want[k]), k
    empty = ops.gather_detections(_d(t["idx"]), _d(t["cls"]), _d(t["score"]), _d(t["box"]), _d(torch.zeros_like(t["cnt"])), _d(t["coeff"]),
                                  _d(t["track"]), _d(t["cen"]), 0)
    assert all(v.shape[0] == 0 for v in empty.values())


def test_counts_to_host_carries_the_scores():
    t = _tables(2)
    counts, sc = ops.counts_to_host(_d(t["cnt"]), extra=_d(t["score"]))
    assert counts == t["cnt"].tolist() and torch.equal(torch.from_numpy(sc.copy()), t["score"].reshape(-1))


def test_shift_rois_and_shift_apply_equal_the_torch_chain():
    t = _tables(3)
    p = t["prev"]
    box = p["box"].clone()
    box[0] = torch.tensor([0.9, 0.2, 0.1, 0.8])       # swapped corners
    box[1] = torch.tensor([-0.2, -0.1, 1.3, 1.2])     # outside the frame
    rois = ops.shift_rois(_d(box), _d(p["clip"]), 24, 40).cpu()
    assert torch.equal(rois, ref._shift_rois(box, p["clip"], 24, 40))
    g = torch.Generator().manual_seed(9)
    loc, dc = torch.randn(t["Pn"], 4, generator=g) * 0.7, torch.randn(t["Pn"], 32, generator=g)
    b1, c1, s1 = box.clone(), p["mask_coeff"].clone(), p["score"].clone()
    ref._shift_apply_(loc, dc, b1, c1, s1, 0.95)
    b2, c2, s2 = _d(box.clone()), _d(p["mask_coeff"].clone()), _d(p["score"].clone())
    ops.shift_apply_(_d(loc), _d(dc), b2, c2, s2, 0.95)
    assert torch.equal(b2.cpu(), b1) and torch.equal(c2.cpu(), c1) and torch.equal(s2.cpu(), s1)


@pytest.mark.parametrize("coeffs", [[0, 1, 2, 0], [0.5, 1.0, 2.0, 0.25]])
def test_match_scores_equals_comp_scores_argmax(coeffs):
    t = _tables(4)
    D = int(t["cnt"].sum())
    det = ref._gather_detections(t["idx"], t["cls"], t["score"], t["box"], t["cnt"], t["coeff"], t["track"], t["cen"], D)
    p = t["prev"]
    # make some pairs genuinely close: copies of prev rows among the detections of the same clip, and an exact tie
    det["box"][0], det["track"][0], det["class"][0] = p["box"][1], p["track"][1], p["class"][1]
    p["box"][3], p["track"][3], p["class"][3] = p["box"][2].clone(), p["track"][2].clone(), p["class"][2].clone()
    det["box"][1], det["track"][1], det["class"][1] = p["box"][2], p["track"][2], p["class"][2]
    cos = det["track"] @ p["track"].t()
    g = torch.Generator().manual_seed(5)
    miou = torch.rand(D, t["Pn"], generator=g) * (det["clip"][:, None] == p["clip"][None, :])
    miou[1, 3] = miou[1, 2]
    want = ref._match_scores(cos, miou, det["box"], p["box"], det["score"], det["class"], p["class"], det["clip"], t["off"], coeffs, 0.3)
    got = ops.match_scores(_d(cos), _d(miou), _d(det["box"]), _d(p["box"]), _d(det["score"]), _d(det["class"]), _d(p["class"]), _d(det["clip"]),
                           _d(t["off"]), coeffs, 0.3).cpu()
    assert torch.equal(got, want), (got.tolist(), want.tolist())
    assert int(got[0]) == 2 and int(got[1]) == 3               # the planted matches; the tie goes to the lower row
    # the pipeline's form: the cosine term from the embedding tables inside the kernel (identical rows still tie exactly)
    got_e = ops.match_scores_embed(_d(det["track"]), _d(p["track"]), _d(miou), _d(det["box"]), _d(p["box"]), _d(det["score"]), _d(det["class"]),
                                   _d(p["class"]), _d(det["clip"]), _d(t["off"]), coeffs, 0.3).cpu()
    assert torch.equal(got_e, want), (got_e.tolist(), want.tolist())
    assert torch.equal(ref._match_scores_embed(det["track"], p["track"], miou, det["box"], p["box"], det["score"], det["class"], p["class"],
                                               det["clip"], t["off"], coeffs, 0.3), want)
    assert (got[t["cnt"][0]:t["cnt"][0] + t["cnt"][2]] == 0).all()   # clip 2 has no tracked rows: every detection is new


def test_gather_rows2_equals_cat_index_select():
    t = _tables(6)
    D = int(t["cnt"].sum())
    det = ref._gather_detections(t["idx"], t["cls"], t["score"], t["box"], t["cnt"], t["coeff"], t["track"], t["cen"], D)
    det["mask"] = torch.rand(D, 24, 40)
    p = t["prev"]
    g = torch.Generator().manual_seed(7)
    plan = torch.randint(0, t["Pn"] + D, (t["Pn"] + 11,), generator=g).to(torch.int32)
    keys = ("box", "mask_coeff", "track", "class", "score", "centerness", "mask", "clip")
    want = ref._gather_rows2([p[k] for k in keys], [det[k] for k in keys], plan, t["Pn"])
    got = ops.gather_rows2([_d(p[k]) for k in keys], [_d(det[k]) for k in keys], _d(plan), t["Pn"])
    for k, a, b in zip(keys, got, want):
        assert a.dtype == b.dtype and torch.equal(a.cpu(), b), k
    # an empty tracked table (first detections after frames without any)
    e = [v[:0] for v in (p[k] for k in keys)]
    plan2 = torch.arange(D, dtype=torch.int32)
    got = ops.gather_rows2([_d(v) for v in e], [_d(det[k]) for k in keys], _d(plan2), 0)
    for k, a in zip(keys, got):
        assert torch.equal(a.cpu(), det[k]), k


def test_pack_tracked_equals_keep_rule_and_scatter():
    t = _tables(8, prev_n=(4, 260, 0, 9, 2), top_k=20)       # clip 1 overflows top_k and spans several scan chunks
    p = t["prev"]
    g = torch.Generator().manual_seed(3)
    tm = torch.randint(0, 14, (t["Pn"],), generator=g).to(torch.int32)
    p["mask"][5] = 0.0
    p["mask"][6] = 0.0
    p["mask"][6, 3, 3] = 0.9                                   # exactly one pixel over 0.5: not kept
    p["mask"][7] = 0.0
    p["mask"][7, 23, 38:40] = 0.7                              # two pixels, the last ones of the row: kept
    p["score"][8] = 0.01
    want = ref._pack_tracked(p["mask"], p["score"], tm, t["off"], p["box"], p["class"], p["mask_coeff"], t["B"], 20, 40, 10, 0.05)
    got = ops.pack_tracked(_d(p["mask"]), _d(p["score"]), _d(tm), _d(t["off"]), _d(p["box"]), _d(p["class"]), _d(p["mask_coeff"]), t["B"], 20, 40,
                           10, 0.05).cpu()
    assert torch.equal(got, want)
    assert int((got[1, :, 7] > 0).sum()) == 20 and int((got[2, :, 7] > 0).sum()) == 0
    # the pipeline's form: the pixel count of the keep rule from the masks' bit words (hw = 24 * 40 = 15 words, the last one
    # partly filled), the soft masks are not read
    hw = p["mask"][0].numel()
    words = (hw + 63) // 64
    bits = torch.zeros(t["Pn"], words * 64, dtype=torch.bool)
    bits[:, :hw] = p["mask"].reshape(t["Pn"], -1) > 0.5
    weights = (2 ** torch.arange(63, dtype=torch.int64)).tolist() + [-(2 ** 63)]
    packed = (bits.view(t["Pn"], words, 64).to(torch.int64) * torch.tensor(weights, dtype=torch.int64)).sum(-1)
    got_b = ops.pack_tracked_bits(_d(packed), _d(p["score"]), _d(tm), _d(t["off"]), _d(p["box"]), _d(p["class"]), _d(p["mask_coeff"]), t["B"],
                                  20, 40, 10, 0.05).cpu()
    assert torch.equal(got_b, want)


def test_lincomb_bits_and_mask_iou_bits_equal_the_two_pass_form():
    """stm_lincomb_sigmoid_crop_bits_f32 writes the soft masks AND their (> 0.5) bits in one pass; stm_mask_iou_bits_f32 on those
    bits equals stm_mask_iou_grouped_f32 on the soft masks (which packs them itself), for a pixel count that is not a multiple of
    64 or 256 as well."""
    for (h, w) in [(24, 40), (13, 11), (96, 160)]:
        g = torch.Generator().manual_seed(h)
        protos = torch.relu(torch.randn(3, h, w, 32, generator=g))
        n1, n2 = 19, 33
        c1, c2 = torch.randn(n1, 32, generator=g), torch.randn(n2, 32, generator=g)
        mk = lambda n: torch.cat([torch.rand(n, 2, generator=g) * 0.5, torch.rand(n, 2, generator=g) * 0.5 + 0.5], 1)
        b1, b2 = mk(n1), mk(n2)
        r1 = torch.sort(torch.randint(0, 3, (n1,), generator=g)).values.to(torch.int32)
        r2 = torch.sort(torch.randint(0, 3, (n2,), generator=g)).values.to(torch.int32)
        m1, bits1 = ops.lincomb_sigmoid_crop_bits(_d(protos), _d(c1), _d(b1), _d(r1))
        m2, bits2 = ops.lincomb_sigmoid_crop_bits(_d(protos), _d(c2), _d(b2), _d(r2))
        ref1 = ops.lincomb_sigmoid_crop(_d(protos), _d(c1), _d(b1), apply_tanh=True, row_proto=_d(r1))
        assert torch.equal(m1, ref1)
        # the words: bit k of word j = pixel 64 j + k
        flat = (m1.reshape(n1, -1) > 0.5).cpu()
        words = bits1.cpu()
        for row in (0, n1 - 1):
            got = torch.tensor([(int(words[row, p // 64]) >> (p % 64)) & 1 for p in range(h * w)], dtype=torch.bool)
            assert torch.equal(got, flat[row])
        want = ops.mask_iou(m1, m2, group1=_d(r1), group2=_d(r2))
        got = ops.mask_iou_bits(bits1, bits2, h * w, group1=_d(r1), group2=_d(r2))
        assert torch.equal(got, want)
        assert torch.equal(ops.mask_iou_bits(bits1, bits2, h * w), ops.mask_iou(m1, m2))
