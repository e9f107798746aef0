"""Row f2 (result aggregation): stmask_amd.eval_utils against the reference's own bbox2result_with_id +
results2json_videoseg (layers/eval_utils.py:15-106), captured in tests/golden/results_json.json by
tests/golden/gen_golden.py results_json on the seeded input of tests/golden/synth_results.py."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from synth_results import synth_video_results  # noqa: E402
from stmask_amd import eval_utils  # noqa: E402


def test_results_json_matches_reference(tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "results_json.json")))
    classes = ["c%d" % i for i in range(40)]
    frames = synth_video_results()
    results = [eval_utils.bbox2result_with_id(det, meta, classes) for det, meta in frames]
    assert len(results) == len(gold["per_frame"])
    for r, g in zip(results, gold["per_frame"]):
        assert sorted(str(k) for k in r) == sorted(g)                    # same objects kept (id -1 dropped)
        for k, v in r.items():
            if k in ("video_id", "frame_id"):
                assert v == g[k]
            else:
                e = g[str(k)]
                assert v["bbox"].tolist() == e["bbox"] and int(v["label"]) == e["label"]
                assert float(v["score"]) == e["score"] and v["category"] == e["category"]
    out = tmp_path / "sub" / "results.json"
    records = eval_utils.results2json_videoseg(results, str(out))
    assert json.load(open(out)) == gold["records"]                      # scores, majority categories, RLE lists, Nones
    assert records == gold["records"]


def test_results_json_edge_cases():
    assert eval_utils.video_records([]) == []
    only_meta = [{"video_id": 1, "frame_id": 0}, {"video_id": 1, "frame_id": 1}]
    assert eval_utils.video_records(only_meta) == []                     # frames without objects produce no records
