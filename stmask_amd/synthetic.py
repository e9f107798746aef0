"""Deterministic synthetic weights / frames.

There is no network for datasets or checkpoints, so benchmarks and parity fixtures use
seeded synthetic data.  Everything here is a pure function of (key name, shape, seed) on
the CPU generator, so the golden-vector generator (which fills the *reference's* modules in
the build container) and the GPU tests (which fill ours) get bit-identical tensors as long
as the state-dict keys agree -- which is itself part of the drop-in contract
(SURVEY.md Appendix B).
"""
import math
import zlib

import torch


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_tensor(key, shape, seed=0, bg_bias=None):
    """Value for one state-dict entry.  Conv/linear weights: N(0, sqrt(2/fan_in)) (keeps activations
    O(1) through ~60 ReLU layers); BN: identity (weight 1, bias 0, mean 0, var 1); biases: U(-0.1, 0.1)."""
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    is_bn = (".bn" in key) or ("downsample.1." in key) or key.startswith("backbone.bn1")
    if is_bn:
        if leaf == "weight":
            # identity BN everywhere except the residual-branch BN: 0.3 keeps the variance of
            # x + f(x) from doubling in each of the 16/33 bottlenecks (activations stay O(1))
            return torch.full(shape, 0.3 if ".bn3." in key else 1.0)
        if leaf == "running_var":
            return torch.ones(shape)
        return torch.zeros(shape)
    if "conv_offset_mask" in key:
        # reference zero-inits these (backbone.py:24-26), which would make the gather degenerate
        if leaf == "weight":
            return torch.randn(shape, generator=g) * 0.01
        b = torch.zeros(shape)
        n_off = shape[0] * 2 // 3
        b[:n_off] = torch.rand(n_off, generator=g) * 4.0 - 2.0
        return b
    if key.endswith("conv_offset.weight"):  # FCB-ada 1x1 offset conv (Featurealign.py:20-25)
        return torch.randn(shape, generator=g) * 0.5
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (math.sqrt(2.0 / max(fan_in, 1)) * _head_gain(key, shape))
    if leaf == "bias":
        return torch.rand(shape, generator=g) * 0.2 - 0.1 + _head_bias(key, shape, BG_BIAS if bg_bias is None else bg_bias)
    return torch.randn(shape, generator=g)


# Output-layer calibration ("detection seeding"): plain random heads give either zero or ALL priors
# as candidates.  Small class logits + a background bias make ~3-8 % of the priors pass the 0.05 threshold in
# spatial clusters, so Fast NMS, lincomb and the tracker see a realistic load (hundreds of candidates, tens of
# detections per frame).
def _is_class_out(key, shape):
    return "conf_layer" in key and len(shape) >= 1 and shape[0] == NUM_CLASSES and "conv_adaption" not in key


NUM_CLASSES = 41


def _head_gain(key, shape):
    if _is_class_out(key, shape):
        # FCB (FeatureAlign) class branch has one more conv + ReLU in front, which narrows the logits
        return 0.6 if ".conv." in key else 0.35
    if "bbox_layer" in key:
        return 0.3
    if "centerness_layer" in key:
        return 0.5
    if key.startswith("TemporalNet.fc"):
        return 0.3
    if key.startswith("fpn.lat_layers"):
        return 0.2  # backbone outputs have second moment ~13: bring P3..P7 back to O(1)
    return 1.0


def _head_bias(key, shape, bg_bias):
    b = torch.zeros(shape)
    if _is_class_out(key, shape):
        b[0] = bg_bias
    elif "centerness_layer" in key:
        b += 1.0
    return b


# 6.2: dense regime used by the small parity fixtures (hundreds of candidates on a 128x192 frame);
# bench.py uses 6.9 at 384x640: ~40 candidates / ~35 detections per frame, tracked set growing to ~100 per 8-frame clip
BG_BIAS = 6.2
BENCH_BG_BIAS = 6.9


def detection_seeding(sd, classes=(1, 7, 13), conf_bump=6.0, centerness_bump=2.0, num_classes=41):
    """Random heads give max class prob ~1/41 < 0.05 and centerness ~0 => zero detections.  Bump a few
    class logits and the centerness bias so post-processing sees hundreds of candidates."""
    for k, v in sd.items():
        if "conf_layer" in k and k.endswith("bias") and v.numel() == num_classes:
            for c in classes:
                v[c] += conf_bump
        if "centerness_layer" in k and k.endswith("bias"):
            v += centerness_bump
    return sd


def fill_state_dict(module, seed=0, seed_detections=False, bg_bias=None):
    """Overwrite every entry of module.state_dict() with its seeded value (in place) and return the dict."""
    sd = module.state_dict()
    new = {k: seeded_tensor(k, v.shape, seed, bg_bias).to(v.dtype) for k, v in sd.items()}
    if seed_detections:
        detection_seeding(new)
    module.load_state_dict(new)
    return new


def synthetic_clip(n_frames, h=384, w=640, seed=0, device="cpu"):
    """x_t = roll(x_0, (2t, 3t)) + 0.05 * N(0,1): real displacement for correlation / TF."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    x0 = torch.randn(3, h, w, generator=g)
    frames = []
    for t in range(n_frames):
        frames.append(torch.roll(x0, shifts=(2 * t, 3 * t), dims=(1, 2)) + 0.05 * torch.randn(3, h, w, generator=g))
    return torch.stack(frames).to(device)
