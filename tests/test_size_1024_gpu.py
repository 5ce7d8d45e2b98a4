"""BASELINE.json configs[3] / configs[4] run at 1024 x 1024 inputs (87 360 anchors, a 256 x 256 routing grid at level 0): DAN (bf16 build) and
DAN-Deform (fp16 build, tests/fp16/cases.py) on ONE 1024 x 1024 image, graph level, against the CPU oracle — the 16-bit path's logits of both
stages, and the fp32 path's logits (1e-4 of scale), stage-1 boxes (1e-4 px-relative) and dynamically routed stage-2 boxes.
The oracle forward takes 10-20 s per graph on the host cores, hence one image."""
import pytest

pytestmark = pytest.mark.gpu


def test_dan_at_1024_logits_boxes_routing(dev):
    import test_eval_f32_gpu as TE
    TE.dan_eval_case(False, 1024, 1024, dev, logits16_tol=0.06)
