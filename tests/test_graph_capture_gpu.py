"""The library is stream-ordered end to end (round 6: no entry point synchronises or reads back), so a whole detection pipeline can be
captured into a hipGraph (torch.cuda.graph) and replayed on new input contents: detector forward on raw uint8 images -> window NMS +
top-K, and the demo's greedy NMS + sub-pixel step on the same score map.  Replays must reproduce the eager results bit for bit."""
import numpy as np
import pytest
import torch

from balf_amd import arch, ops, pipeline
from balf_amd.model import get_model
from balf_amd.utils import synth
from tests.golden import cases

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_detect_and_greedy_pipeline_replays_from_a_graph():
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(synth.synthetic_state_dict(cases.WEIGHT_SEED))
    m = m.eval().to(DEV)
    h, w, k = 480, 640, 1000
    _, _, top, left = arch.padded_hw(h, w)
    imgs = [torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, 10 * j + i) for i in range(2)])).to(DEV) for j in range(3)]

    def run(x):
        idx, score, count, prob = pipeline.detect_batch_u8(m, x, 15, 15, k)
        g = ops.greedy_nms(prob, top, left, h, w, 15, 0.015, 15, 1024, 5)
        return (idx, score, count, prob) + tuple(g)
    static = imgs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # warm-up: weight blobs, range probes, LDS attributes, workspaces
        for _ in range(2):
            run(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = run(static)
    for x in (imgs[1], imgs[2], imgs[0]):
        static.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        want = run(x)
        torch.cuda.synchronize()
        for a, b in zip(outs, want):
            assert torch.equal(a, b)
    assert int(outs[2].min()) == k and int(outs[7].min()) > 100


def test_graphed_detector_matches_eager_calls():
    """pipeline.GraphedDetector: captured for one shape, replayed on three different batches, against the eager detect_batch_u8; a
    wrong shape is refused."""
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(synth.synthetic_state_dict(cases.WEIGHT_SEED))
    m = m.eval().to(DEV)
    h, w, k = 200, 328, 500
    imgs = [torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, 7 * j + i) for i in range(3)])).to(DEV) for j in range(3)]
    det = pipeline.GraphedDetector(m, imgs[0], 15, 15, k)
    for x in (imgs[1], imgs[2], imgs[0], imgs[1]):
        got = [t.clone() for t in det(x)]
        want = pipeline.detect_batch_u8(m, x, 15, 15, k)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    with pytest.raises(ValueError):
        det(imgs[0][:2])
    # RGB input [B,H,W,3]
    rgb = [torch.from_numpy(np.stack([np.stack([synth.synthetic_gray_u8(h, w, 5 * j + i + c) for c in range(3)], axis=-1) for i in range(2)])).to(DEV)
           for j in range(2)]
    det3 = pipeline.GraphedDetector(m, rgb[0], 15, 15, k)
    for x in (rgb[1], rgb[0]):
        got = [t.clone() for t in det3(x)]
        want = pipeline.detect_batch_u8(m, x, 15, 15, k)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b)
