"""Row f3 (frame pre-processing).  CPU: the oracle restatement of eval.py:703-717 against independent numpy arithmetic
where that is possible without cv2 (exact 2x down-scale = OpenCV's fixed-point average, identity resize, constants,
normalisation + pad + CHW in numpy float64 as the reference computes them).  GPU: the HIP kernel bit-exact against the
oracle on seeded frames, including non-integer scale factors and up-scaling."""
import numpy as np
import pytest
import torch

import oracle

MEANS = np.array((123.675, 116.28, 103.53))
STD = np.array((58.395, 57.12, 57.375))


def rand_u8(*shape, seed=0):
    return torch.randint(0, 256, shape, generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)


def test_oracle_preprocess_720p_matches_numpy_chain():
    im = rand_u8(2, 720, 1280, 3)
    out = oracle.preprocess_frames(im)
    assert out.shape == (2, 3, 384, 640)
    a = im.numpy().astype(np.int64)
    D = (a[:, :, 0::2] + a[:, :, 1::2]) * 1024                      # horizontal pass, both coefficients 1024
    v = ((((1024 * (D[:, 0::2] >> 4)) >> 16) + ((1024 * (D[:, 1::2] >> 4)) >> 16) + 2) >> 2)
    ref = np.pad((v - MEANS) / STD, ((0, 0), (0, 24), (0, 0), (0, 0)))   # numpy float64, zero pad 360 -> 384
    ref = torch.tensor(ref).permute(0, 3, 1, 2).float()
    assert torch.equal(out, ref)


def test_oracle_preprocess_identity_constant_and_modes():
    im = rand_u8(1, 37, 53, 3, seed=1)
    o = oracle.preprocess_frames(im, size=(53, 37), divisor=1, mode=0)
    assert torch.equal(o[0].permute(1, 2, 0).to(torch.uint8), im[0])          # same size: coefficients (2048, 0)
    const = torch.full((1, 45, 80, 3), 77, dtype=torch.uint8)
    o = oracle.preprocess_frames(const, size=(64, 36), divisor=32, mode=0)
    assert (o[0, :, :36, :64] == 77).all() and (o[0, :, 36:] == 0).all()       # interpolation of a constant; zero pad
    o2 = oracle.preprocess_frames(const, size=(64, 36), divisor=32, mode=2)
    assert torch.equal(o2[0, :, 0, 0], torch.tensor(77 - MEANS).float())
    o3 = oracle.preprocess_frames(const, size=(64, 36), divisor=32, mode=3)
    assert torch.equal(o3[0, :, 0, 0], torch.tensor([77 / 255.0] * 3, dtype=torch.float64).float())


@pytest.mark.gpu
@pytest.mark.parametrize("src,size", [((720, 1280), (640, 360)), ((480, 854), (640, 360)), ((360, 640), (640, 360)),
                                      ((200, 300), (640, 360)), ((1080, 1920), (1280, 720))])
def test_preprocess_kernel_bit_exact(src, size):
    from stmask_amd import ops
    im = rand_u8(3, src[0], src[1], 3, seed=src[0])
    for mode in (0, 1):
        ref = oracle.preprocess_frames(im, size=size, mode=mode)
        out = ops.preprocess_frames(im.cuda(), size=size, mode=mode).cpu()
        assert torch.equal(out, ref), (src, size, mode, (out - ref).abs().max())


@pytest.mark.gpu
def test_preprocess_mirror_meta_and_errors():
    from stmask_amd import ops, preprocess
    from stmask_amd._lib import StmError
    im = rand_u8(2, 720, 1280, 3).cuda()
    x, meta = preprocess.preprocess_eval_frames(im, idx=0)
    assert x.shape == (2, 3, 384, 640) and meta == {"ori_shape": (720, 1280, 3), "img_shape": (360, 640, 3),
                                                    "pad_shape": (384, 640, 3), "frame_id": 0, "is_first": True}
    with pytest.raises(StmError):
        ops.preprocess_frames(im.float())
    with pytest.raises(StmError):
        ops.preprocess_frames(im.cpu())
