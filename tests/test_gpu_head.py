"""stm_head_assemble_f32 against the reference's own tail of PredictionModule_FC.forward (prediction_head_FC.py:168-195),
restated with the same torch calls (cat over kernel shapes on the channel axis, view, cat over levels; centerness
concatenated along H; tanh; F.normalize) on seeded per-level tensors."""
import pytest
import torch
import torch.nn.functional as F

from stmask_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,sizes", [(2, [(6, 10), (3, 5), (2, 3)]), (8, [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)])])
def test_head_assemble_matches_reference_tail(B, sizes):
    K, P, ncls, mdim, edim = 3, 64, 41, 32, 128
    g = torch.Generator().manual_seed(len(sizes))
    ntot = sum(B * h * w for h, w in sizes)
    small = [torch.randn(ntot, 3 * P, generator=g).cuda() for _ in range(K)]
    trk = [torch.randn(ntot, edim, generator=g).cuda() for _ in range(K)]
    conf, loc, mask, track, cen = ops.head_assemble(small, trk, B, sizes, ncls, mdim, edim, P)
    # the reference, level by level (nhwc tensors per kernel shape k)
    r_conf, r_loc, r_mask, r_trk, r_cen, start = [], [], [], [], [], 0
    for h, w in sizes:
        sl = slice(start, start + B * h * w)
        start += B * h * w
        cf = [s[sl, 0:ncls].view(B, h, w, ncls) for s in small]
        ce = [s[sl, P:P + 1].view(B, h, w, 1) for s in small]
        bb = [s[sl, P + 1:P + 5].view(B, h, w, 4) for s in small]
        mk = [s[sl, 2 * P:2 * P + mdim].view(B, h, w, mdim) for s in small]
        tk = [t[sl].view(B, h, w, edim) for t in trk]
        r_mask.append(torch.cat(mk, dim=-1).view(B, -1, mdim))
        r_loc.append(torch.cat(bb, dim=-1).view(B, -1, 4))
        r_cen.append(torch.tanh(torch.cat(ce, dim=1).view(B, -1, 1)))          # dim=1: the reference's quirk
        r_conf.append(torch.cat(cf, dim=-1).view(B, -1, ncls))
        r_trk.append(F.normalize(torch.cat(tk, dim=-1).view(B, -1, edim), dim=-1))
    assert torch.equal(conf, torch.cat(r_conf, 1)) and torch.equal(loc, torch.cat(r_loc, 1))
    assert torch.equal(mask, torch.cat(r_mask, 1))
    assert (cen - torch.cat(r_cen, 1)).abs().max().item() < 2e-7
    assert (track - torch.cat(r_trk, 1)).abs().max().item() < 2e-7
