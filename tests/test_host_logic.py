"""CPU-only tests of the host-side mirror: configs, state-dict compatibility with the reference, priors, synthetic data."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from stmask_amd import synthetic
from stmask_amd.config import CONFIGS, cfg, get_cfg, set_cfg
from stmask_amd.layers import PredictionModule_FC
from stmask_amd.model import STMask

MODEL_CASES = [("STMask_plus_resnet50_config", "r50_fca"), ("STMask_plus_resnet50_ada_config", "r50_ada"),
               ("STMask_plus_resnet50_ali_config", "r50_ali"), ("STMask_plus_base_ali_config", "r101_ali")]


@pytest.mark.parametrize("name,tag", MODEL_CASES)
def test_state_dict_matches_reference(name, tag):
    """Keys and shapes equal the reference model's (captured by gen_golden.py): its checkpoints load unchanged."""
    net = STMask(get_cfg(name))
    sd = net.state_dict()
    g = load_golden(f"model_{tag}.npz")
    assert sorted(sd.keys()) == list(g["state_keys"])
    for k, s in zip(g["state_keys"], g["state_shapes"]):
        assert str(tuple(sd[k].shape)) == s, k


def test_dcn_layer_selection_rule():
    """backbone.py:124,130 with args config.py:288,307: R50 -> 7 DCN blocks, R101 -> 11."""
    def dcn_blocks(name):
        net = STMask(get_cfg(name))
        return [f"{s}.{b}" for s, layer in enumerate(net.backbone.layers) for b, blk in enumerate(layer) if blk.use_dcn]
    assert dcn_blocks("STMask_plus_resnet50_config") == ["1.0", "1.2", "2.0", "2.2", "2.4", "3.0", "3.2"]
    r101 = dcn_blocks("STMask_plus_base_config")
    assert len(r101) == 11 and r101[:2] == ["1.0", "1.3"] and r101[-1] == "3.0"
    assert dcn_blocks("STMask_resnet50_config") == []


def test_priors_match_reference(golden_priors):
    pm = PredictionModule_FC.__new__(PredictionModule_FC)
    pm.pred_aspect_ratios, pm.pred_scales = cfg.pred_aspect_ratios[0], cfg.pred_scales[0]
    for k, ref in golden_priors.items():
        h, w = [int(v) for v in k[2:].split("x")]
        assert torch.equal(pm.make_priors(h, w, "cpu")[0], ref), k
    total = sum(v.shape[0] for v in golden_priors.values())
    assert total == 15345


def test_set_cfg_swaps_default_in_place():
    set_cfg("STMask_plus_resnet50_ada_config")
    assert cfg.use_pred_offset and cfg.use_dcn_class and cfg.backbone_dcn_interval == 2
    set_cfg("STMask_plus_base_config")
    assert not cfg.use_dcn_class and cfg.backbone_layers == [3, 4, 23, 3]
    with pytest.raises(KeyError):
        set_cfg("nope")
    assert set(CONFIGS) >= {"STMask_plus_base_ali_config", "STMask_plus_resnet50_config"}


def test_synthetic_weights_are_pure_functions_of_key():
    a = synthetic.seeded_tensor("backbone.layers.1.0.conv2.weight", (128, 128, 3, 3), 0)
    b = synthetic.seeded_tensor("backbone.layers.1.0.conv2.weight", (128, 128, 3, 3), 0)
    c = synthetic.seeded_tensor("backbone.layers.1.0.conv2.weight", (128, 128, 3, 3), 1)
    assert torch.equal(a, b) and not torch.equal(a, c)
    om = synthetic.seeded_tensor("backbone.layers.1.0.conv2.conv_offset_mask.bias", (27,), 0)
    assert om[:18].abs().max() <= 2 and om[18:].abs().max() == 0  # offsets U(-2,2), mask logits 0
    clip = synthetic.synthetic_clip(3, 32, 48, seed=0)
    assert clip.shape == (3, 3, 32, 48) and torch.equal(clip, synthetic.synthetic_clip(3, 32, 48, seed=0))


def test_model_golden_fixtures_are_nontrivial():
    for _, tag in MODEL_CASES:
        g = load_golden(f"model_{tag}.npz")
        last = int(g["n_frames"]) - 1
        assert g["t0_box"].shape[0] >= 3 and g[f"t{last}_box"].shape[0] >= 3, tag
        assert np.isfinite(g["f0_proto"].numpy()).all()


def test_fused_deform_conv_grid_helper_follows_the_tile_shapes(monkeypatch):
    """ops.deform_conv_fused_tiles mirrors the launcher of csrc/dcn_fused.hip (pick_patch + the tile shape rule): 128-pixel x 128-channel tiles, or
    64 x 256 where the layer has a multiple of 256 output channels (STM_DCN_FUSED_WIDE, default on).  The graph's small-grid rule rests on it."""
    from stmask_amd import ops
    monkeypatch.delenv("STM_DCN_FUSED_WIDE", raising=False)
    assert ops.deform_conv_fused_tiles(32, 48, 80, 128) == 32 * 30                 # 8x16 patches: 6 x 5 per image, one channel tile
    assert ops.deform_conv_fused_tiles(32, 24, 40, 256) == 32 * 15                 # 64-pixel patches (8x8): 3 x 5 per image, one 256-channel tile
    assert ops.deform_conv_fused_tiles(32, 12, 20, 512) == 32 * 4 * 2              # 6x10 patches: 2 x 2 per image, two 256-channel tiles
    assert ops.deform_conv_fused_tiles(1, 3, 5, 128) == 1                          # P7: one partial patch
    monkeypatch.setenv("STM_DCN_FUSED_WIDE", "0")
    assert ops.deform_conv_fused_tiles(32, 24, 40, 256) == 32 * 8 * 2              # 128-pixel patches, two 128-channel tiles: every pixel sampled twice


def test_bench_defaults_of_round_5():
    """The driver runs `python bench.py` bare: the next trunk starts early, the timed region carries no per-launch events, the roofline passes exist."""
    import bench
    a = bench.parse_args([])
    assert a.overlap == "early" and not a.events_in_timed_region and not a.no_sampler_pass and a.gpus == 1
    assert bench.parse_args(["--events-in-timed-region", "--overlap", "late"]).events_in_timed_region


def test_bench_defaults_of_round_6():
    """The bare `python bench.py` also runs the sustained line and one short line per remaining BASELINE configuration."""
    import bench
    names = bench.parse_args([]).extras.split(",")
    assert {"sustained", "config3", "config4", "config5", "clips1", "clips8", "realistic"} <= set(names)
    from benchlib.extras import CONFIG_LINES
    assert [c[0] for c in CONFIG_LINES] == ["config3", "config4", "config5"]
    assert CONFIG_LINES[2][3:6] == (736, 1280, "fp16x1")                      # BASELINE config 5: 720x1280 (padded to /32), fp16 backbone convolutions


def test_rank_cpu_binding_helpers(tmp_path):
    """benchlib.launch: a rank takes its share of the cores local to its GPU's NUMA node."""
    from benchlib import launch
    assert launch.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert launch.parse_cpulist("") == []
    cores = list(range(0, 16)) + list(range(128, 144))                        # physical cores, then their SMT siblings
    slices = [launch.rank_core_slice(cores, i, 4) for i in range(4)]
    assert sorted(sum(slices, [])) == cores and all(len(s) == 8 for s in slices)
    assert slices[1] == [4, 5, 6, 7, 132, 133, 134, 135]                        # a rank's cores come with their own siblings
    assert launch.rank_core_slice(list(range(8)), 1, 2) == [4, 5, 6, 7]
    assert launch.rank_core_slice([5, 6], 3, 4) == [6]                         # fewer cores than ranks: shared, never empty
    d = tmp_path / "0000:c1:00.0"
    d.mkdir()
    (d / "local_cpulist").write_text("96-111,224-239\n")
    assert launch.gpu_local_cpulist(0, 0xc1, 0, sysfs=str(tmp_path)) == list(range(96, 112)) + list(range(224, 240))
    assert launch.gpu_local_cpulist(0, 0xc2, 0, sysfs=str(tmp_path)) == []


def test_pmc_traffic_file_is_refused_when_its_launch_counts_disagree(tmp_path, monkeypatch):
    """bench.py reads the HBM traffic of its kernels from a committed counter file: one that was collected on other launches must not be quoted."""
    import json
    from benchlib import roofline as rf
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "x.json").write_text(json.dumps({"conv_planar": {"traffic_bytes_per_launch": 4.4e8, "launches_per_step": 70.0},
                                             "old": {"traffic_bytes_per_launch": 1.0e8}}))
    monkeypatch.setattr(rf, "ROOT", str(tmp_path))
    t, src, why = rf.pmc_traffic("conv_planar", 70.6, pmc_file="x.json")
    assert t == 440000000 and src == "x.json" and why is None
    t, src, why = rf.pmc_traffic("conv_planar", 81.0, pmc_file="x.json")       # another tile rule / fusion threshold: refused
    assert t is None and "differ" in why
    t, src, why = rf.pmc_traffic("old", 3.0, pmc_file="x.json")                # a file from before the check existed
    assert t is None and "launches_per_step" in why
    t, src, why = rf.pmc_traffic("missing", 3.0, pmc_file="x.json")
    assert t is None and src is None


def test_board_sampler_reads_hwmon_files_and_summarises_windows(tmp_path, monkeypatch):
    """benchlib.extras.BoardSampler (extras.sustained, scripts/power_probe.py): power1_input in microwatts and freq1_input in Hz of the device's hwmon
    directory, sampled on a host thread; summary per named time window; no source -> no thread, an empty summary."""
    import time
    from benchlib import extras
    pw, ck = tmp_path / "power1_input", tmp_path / "freq1_input"
    pw.write_text("1320000000\n")
    ck.write_text("1930000000\n")
    monkeypatch.setattr(extras.BoardSampler, "_find_hwmon", staticmethod(lambda i: {"power_uw": str(pw), "sclk_hz": str(ck)}))
    with extras.BoardSampler(0, 0.005) as smp:
        time.sleep(0.06)
        pw.write_text("1400000000\n")
        time.sleep(0.06)
    s = smp.summary([("late", 0.07, 10.0)])
    assert s["samples"] >= 8 and s["source"].startswith("sysfs hwmon")
    assert s["power_w"]["min"] == 1320.0 and s["power_w"]["max"] == 1400.0 and s["sclk_mhz"]["mean"] == 1930.0
    assert s["late"]["samples"] >= 2 and s["late"]["power_w"]["min"] == 1400.0
    monkeypatch.setattr(extras.BoardSampler, "_find_hwmon", staticmethod(lambda i: {}))
    monkeypatch.setattr("shutil.which", lambda name: None)
    with extras.BoardSampler(0, 0.005) as none:
        pass
    assert none.source is None and none.summary()["samples"] == 0


def test_look_ahead_depth_and_graph_slots_by_batch_size():
    """Round 6: trunk graphs at every batch size; up to LARGE_BATCH clips three trunks run ahead (launch-latency-bound chains), above it two (a 32-clip trunk
    fills the GPU except at its ends); 2 D + 2 graph slots either way; the detection gather's stream comes after the trunk streams."""
    from stmask_amd import pipeline

    class Net:
        cfg = type("C", (), {"temporal_fusion_module": True})()

    P = pipeline.BatchedClipPipeline
    small, large = P(Net(), 8), P(Net(), 32)
    assert small.prefetch_depth == P.PREFETCH_DEPTH == 3 and small.n_graph_slots == P.N_GRAPH_SLOTS == 8
    assert large.prefetch_depth == 2 and large.n_graph_slots == 6
    assert pipeline.trunk_stream_count() == 3


def test_committed_pmc_traffic_file_is_checkable():
    """The counter file bench.py quotes (benchlib.roofline.PMC_FILE) exists, names the run it was collected on and carries launches per step for the three
    kernel groups the line quotes traffic for -- without them bench.py refuses it (test above) and the line's `traffic` would be null."""
    import json
    import os
    from benchlib import roofline as rf
    path = os.path.join(rf.ROOT, "profiles", rf.PMC_FILE)
    assert os.path.exists(path), path
    d = json.load(open(path))
    for grp in ("conv_planar", "dcn_fused", "dcn_sample_planar"):
        assert d[grp]["launches_per_step"] > 0 and d[grp]["traffic_bytes_per_launch"] > 1e6, grp
        t, src, why = rf.pmc_traffic(grp, d[grp]["launches_per_step"])
        assert t == int(d[grp]["traffic_bytes_per_launch"]) and src == rf.PMC_FILE and why is None
    assert "32 clips/GPU" in d["profiled_run"]["workload"] and "STMask_plus_resnet50_config" in d["profiled_run"]["workload"]


def test_sustained_and_config_lines_glue_with_a_fake_runner(monkeypatch):
    """benchlib.extras.sustained_block / config_lines without a GPU: chunking (>= 500 steps AND >= min_seconds, bounded), graph captures kept out of the
    first chunk, frames/s arithmetic, one line per BASELINE configuration with its own arguments, a failing configuration reported instead of raised."""
    import argparse
    import types
    import torch
    from benchlib import extras

    made = []

    class FakeRunner:
        def __init__(self, args, dev, rank, world, clips, planes=None, net=None, max_instances=None):
            if args.config == "STMask_plus_base_ali_config" and args.height == 736:
                raise RuntimeError("out of memory (pretend)")
            self.args, self.clips, self.planes = args, clips, planes
            self.batched, self.tracked_sum, self.tracked_steps, self.steps_run = True, 50.0, 10, 0
            self.gatherer = types.SimpleNamespace(wait=lambda: None)
            self.pipe = types.SimpleNamespace(use_graph=True, _graphs=[], n_graph_slots=3, prefetch_early=True)
            self.net = types.SimpleNamespace(cfg=types.SimpleNamespace(backbone_layers=(3, 4, 6, 3), backbone_dcn_layers=(0, 4, 6, 3), use_pred_offset=True,
                                                                      use_dcn_class=True, temporal_fusion_module=True))
            made.append(self)

        def step(self, t):
            import time
            time.sleep(0.0005)
            self.steps_run += 1
            if len(self.pipe._graphs) < self.pipe.n_graph_slots and self.steps_run > 2:
                self.pipe._graphs.append(object())

        def timed(self, warmup, steps, use_dist=False, collect=False):
            return (0.5, None, None, [] if collect else None)

    monkeypatch.setattr(extras, "Runner", FakeRunner)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(extras.BoardSampler, "_find_hwmon", staticmethod(lambda i: {}))
    monkeypatch.setattr("shutil.which", lambda name: None)
    args = argparse.Namespace(clips=32, warmup=2, steps=20, frames=16, config="STMask_plus_resnet50_config", height=384, width=640, planes="fp16x2", max_instances=0)
    s = extras.sustained_block(args, torch.device("cpu"), 0, 1, None, headline_ms=20.0, min_seconds=0.0, chunk=100, min_chunks=5, max_chunks=6)
    assert s["steps"] == 500 and len(s["frames_per_s_by_100_steps"]) == 5 and s["board"]["samples"] == 0 and s["board"]["source"] is None
    assert made[0].steps_run == 500 + 2 + 3 and len(made[0].pipe._graphs) == 3            # warm-up 2, three more steps until every graph slot exists, then the chunks
    assert abs(s["value"] - 32 * 500 / s["seconds"]) / s["value"] < 0.05 and s["first_100"] == s["frames_per_s_by_100_steps"][0]

    class _Pl:
        FMT, BACKBONE_FMT = 1, None
        set_format = staticmethod(lambda *a: None)
    monkeypatch.setitem(__import__("sys").modules, "stmask_amd.planar", _Pl)
    import stmask_amd
    monkeypatch.setattr(stmask_amd, "planar", _Pl, raising=False)
    made.clear()
    out = extras.config_lines(args, torch.device("cpu"), 0, 1, ["config3", "config4", "config5"], steps=16, warmup=3)
    assert set(out) == {"config3", "config4", "config5"}
    assert out["config3"]["value"] == round(32 * 16 / 0.5, 2) and out["config3"]["planes"] == "fp16x2" and "FCB(ada)" in out["config3"]["workload"]
    assert out["config4"]["clips_per_gpu"] == 32 and made[1].args.config == "STMask_plus_base_ali_config" and made[1].args.height == 384
    assert "error" in out["config5"] and "out of memory" in out["config5"]["error"]        # a failing side line is reported, never raised
    assert args.config == "STMask_plus_resnet50_config" and args.clips == 32                 # the headline's arguments are untouched
