"""Host logic behind the drop-in boundary: get_model's load API, error behaviour, padding helpers.
CPU only; expectations for the loader come from running the reference's own loader
(tests/golden/geometry.json, written by make_golden.py)."""
import json
import logging
import os

import numpy as np
import pytest
import torch

from balf_amd import arch, pipeline
from balf_amd._lib import BalfHipError
from balf_amd.model import get_model
from balf_amd.utils import synth
from balf_amd.utils import test_utils as T
from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
GEO = json.load(open(os.path.join(G, "geometry.json")))


def new_model():
    return get_model.load_model(arch.DEFAULT_MODEL_CFG)


def test_state_dict_is_the_reference_contract():
    m = new_model()
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == GEO["state"]
    assert sum(p.numel() for p in m.parameters()) == 1280859 - 0
    assert isinstance(m, torch.nn.Module) and hasattr(m, "eval") and hasattr(m, "named_parameters")


def test_load_model_reads_only_network_architecture():
    cfg = {"name": "whatever", "network_architecture": dict(arch.DEFAULT_ARCH, out_channels=1)}
    assert len(get_model.load_model(cfg).state_dict()) == 167
    bad = dict(arch.DEFAULT_ARCH); del bad["cell_size"]
    with pytest.raises(KeyError):
        get_model.load_model({"network_architecture": bad})
    with pytest.raises(NotImplementedError):
        get_model.load_model({"network_architecture": dict(arch.DEFAULT_ARCH, en_embed_dims=[3, 16, 32, 64, 128])})


def test_loader_matches_reference_behaviour(tmp_path, capsys):
    sd = synth.synthetic_state_dict(7)
    m = new_model()
    full = tmp_path / "full.pth"
    torch.save({"epoch": 12, "repeatability": 0.5, "model_state": sd, "optimizer_state": None}, full)
    assert list(get_model.load_test_pretrained_model(m, str(full), device="cpu")) == GEO["loader"]["full"]
    assert float(m.state_dict()["down3.conv2.bias"][5]) == GEO["loader"]["full_probe"]
    bare = tmp_path / "bare.pth"
    torch.save({"model_state": sd}, bare)
    assert list(get_model.load_test_pretrained_model(m, str(bare), device="cpu")) == GEO["loader"]["bare"]

    def attempt(tag, mut):
        d = dict(sd); mut(d)
        p = tmp_path / (tag + ".pth")
        torch.save({"model_state": d}, p)
        try:
            get_model.load_test_pretrained_model(m, str(p), device="cpu")
            return "ok"
        except AssertionError:
            return "AssertionError"

    assert attempt("missing_key", lambda d: d.pop("down2.conv2.bias")) == GEO["loader"]["missing_key"]
    assert "Not updated weight down2.conv2.bias" in capsys.readouterr().out
    assert attempt("wrong_shape", lambda d: d.__setitem__("down1.conv.0.weight", torch.zeros(32, 4))) == \
        GEO["loader"]["wrong_shape"]
    assert attempt("extra_key", lambda d: d.__setitem__("not.a.key", torch.zeros(1))) == GEO["loader"]["extra_key"]
    with pytest.raises(FileNotFoundError):
        get_model.load_test_pretrained_model(m, str(tmp_path / "nope.pth"), device="cpu")


def test_load_pretrained_model_logs(tmp_path):
    sd = synth.synthetic_state_dict(8)
    p = tmp_path / "c.pth"
    torch.save({"epoch": 3, "model_state": sd}, p)
    msgs = []

    class L:
        def info(self, s):
            msgs.append(s)

    assert get_model.load_pretrained_model(new_model(), str(p), L(), device="cpu") == (3, 0.0)
    assert any("loaded 167/167" in s for s in msgs) and any(s.startswith("Update weight down1.conv.0.weight") for s in msgs)
    with pytest.raises(FileNotFoundError):
        get_model.load_pretrained_model(new_model(), str(tmp_path / "x.pth"), L())


def test_optimizer_state_round_trip(tmp_path):
    m = new_model()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    p = tmp_path / "o.pth"
    torch.save({"model_state": m.state_dict(), "optimizer_state": opt.state_dict()}, p)
    opt2 = torch.optim.Adam(new_model().parameters(), lr=5e-2)
    get_model.load_test_pretrained_model(new_model(), str(p), optimizer=opt2, device="cpu")
    assert opt2.param_groups[0]["lr"] == 1e-3


def test_forward_refuses_cpu_training_and_bad_shapes():
    m = new_model()
    with pytest.raises(BalfHipError):                    # train mode: inference path only
        m(torch.zeros(1, 3, 64, 64))
    m.eval()
    with pytest.raises(BalfHipError):                    # no CPU fallback
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 64, 64))


def test_padding_helpers_match_reference_geometry():
    for key, g in GEO["pad"].items():
        h, w = map(int, key.split("x"))
        img = np.zeros((h, w, 3)); img[0, 0, 0] = 1.0
        ev = T.make_shape_even(img)
        pd = T.mod_padding_symmetric(ev, factor=64)
        assert list(ev.shape[:2]) == g["even"] and list(pd.shape[:2]) == g["padded"]
        assert list(np.argwhere(pd[:, :, 0] == 1.0)[0]) == g["origin"]
        x = pipeline.pad_batch(img[None])
        assert list(x.shape) == [1, 3] + g["padded"] and x.dtype == torch.float32
    rb = T.remove_borders(np.ones((40, 50), np.float32), 15)
    assert np.array_equal(rb, O.remove_borders(np.ones((40, 50), np.float32), 15))
    assert T.remove_borders(np.ones((20, 50), np.float32), 15).sum() == 0


def test_accelerated_helpers_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(BalfHipError):
        T.apply_nms(np.zeros((8, 8), np.float32), 3)
    with pytest.raises(BalfHipError):
        T.get_point_coordinates(np.zeros((8, 8), np.float32), num_points=3)


def test_shard_range_partitions_the_batch():
    for total in (256, 255, 7, 1):
        for world in (1, 2, 4, 8):
            spans = [pipeline.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


# ---------------- demo path: HardNet container, demo_match mirror ----------------
def test_hardnet_container_matches_reference_state_dict():
    import numpy as np
    from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet
    from balf_amd.utils import synth
    keys = list(np.load(os.path.join(os.path.dirname(__file__), "golden", "hardnet.npz"))["state_keys"])
    m = HardNet()
    assert list(m.state_dict().keys()) == keys                      # recorded from the reference's class
    sd = synth.synthetic_hardnet_state_dict(1)
    m.load_state_dict({"state_dict": sd}["state_dict"])             # demo_match.py:132-133
    assert all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())


def test_hardnet_has_no_cpu_path():
    from balf_amd._lib import BalfHipError
    from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet
    m = HardNet().eval()
    with pytest.raises(BalfHipError):
        m(torch.zeros(2, 1, 32, 32))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 32, 32))
    with pytest.raises(BalfHipError):
        HardNet().train()(torch.zeros(2, 1, 32, 32))


def test_demo_match_mirror_signatures():
    import inspect
    from balf_amd.demo import demo_match
    assert list(inspect.signature(demo_match.detect).parameters) == ["args", "im", "detector", "device"]
    assert list(inspect.signature(demo_match.extract_features).parameters) == [
        "args", "im_rgb", "im_gray", "detector", "descriptor", "device"]
    assert list(inspect.signature(demo_match.extract_matches).parameters) == [
        "args", "im_rgb1", "im_gray1", "im_rgb2", "im_gray2", "detector", "descriptor", "device"]
    a = demo_match.DEFAULT_ARGS          # /root/reference/balf/configs/config.py:44-59
    assert (a.border_size, a.nms_size, a.num_features, a.s_mult, a.patch_size) == (15, 15, 2048, 60, 4)


def test_repeatability_host_helpers_match_reference_golden():
    """check_common_points / select_top_k (index bookkeeping, host side) against results recorded from the
    reference's own function bodies."""
    import numpy as np
    from balf_amd.benchmark_test import repeatability_tools as R
    from tests.golden import cases
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "repeatability.npz"))
    src, _ = cases.repeat_inputs(cases.REPEAT_CASES["train_eval"])
    _, _, ms, _ = cases.eval_inputs(cases.EVAL_CASE)
    kp = np.stack([src[:, 1] * 0.3 + 1, src[:, 0] * 0.3 + 1, src[:, 2], src[:, 3]], axis=1)
    assert np.array_equal(R.check_common_points(kp, ms), g["helpers.common"])
    assert np.array_equal(R.select_top_k(kp, 40), g["helpers.topk"])


def test_packed_weight_cache_key_notices_every_kind_of_change():
    """The per-call key of the packed-weight cache (slot identity + version sum + data pointers over a cached tensor list)
    must change on in-place updates, load_state_dict (copy and assign), dtype / device moves of the model OR of a submodule,
    parameter replacement, `p.data = ...` swaps and vector_to_parameters -- and must NOT change because some
    other module was built or registered a parameter."""
    import torch
    from balf_amd import arch
    from balf_amd.model import get_model
    from balf_amd.utils import synth
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG).eval()
    seen = [m._state_key("cpu")]

    def changed():
        k = m._state_key("cpu")
        assert k not in seen
        seen.append(k)
    assert m._state_key("cpu") == seen[0]                  # stable while nothing changes
    with torch.no_grad():
        m.down3.conv2.bias.add_(1.0)
    changed()
    m.load_state_dict(synth.synthetic_state_dict(1))
    changed()
    m.load_state_dict(synth.synthetic_state_dict(2), assign=True)
    changed()
    m.down1.conv[0].weight = torch.nn.Parameter(torch.zeros(32, 3))
    changed()
    assert m._state_tensors()[0] is m.down1.conv[0].weight
    m.double()
    changed()
    m.float()
    changed()
    # the idioms the round-3 key missed
    m.down2.conv2.weight.data = torch.ones(64, 64)                       # EMA-style swap
    changed()
    m.down1.double()                                                     # _apply on a submodule
    changed()
    m.detector_head.norm.double()                                        # ... one that owns buffers (re-bound in the dict)
    changed()
    assert m._state_tensors()[-3] is m.detector_head.norm.running_mean
    m.float()
    changed()
    ps = list(m.down4.parameters())
    torch.nn.utils.vector_to_parameters(torch.nn.utils.parameters_to_vector(ps) * 0 + 2.0, ps)
    changed()
    # other modules coming and going leave the key (and with it the packed blob and the f16 range verdict) alone
    k = m._state_key("cpu")
    other = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    torch.nn.Linear(3, 3).register_buffer("x", torch.zeros(1))
    assert m._state_key("cpu") == k
    del other
    m.precision = "fp32"
    changed()
    assert len(m._state_tensors()) == 167


def test_data_alias_edits_need_invalidate_packed():
    """an in-place edit THROUGH ``p.data`` is invisible to the cache key (``.data`` is an alias with its own version
    counter) -- pinned here so that the documented remedy, ``invalidate_packed()``, stays the contract: it empties the blob cache
    and the split-f16 verdict, so the next forward packs the edited weights."""
    import torch
    from balf_amd import arch
    from balf_amd.model import get_model
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG).eval()
    k = m._state_key("cpu")
    m.down2.conv2.weight.data.mul_(2.0)
    m.down1.conv[0].bias.data.copy_(torch.ones(32))
    assert m._state_key("cpu") == k                      # the limitation: nothing moved that the key can see
    with torch.no_grad():
        m.down2.conv2.weight.mul_(0.5)                   # the same edit on the parameter itself IS seen
    assert m._state_key("cpu") != k
    blob = m.packed_weights("cpu")                       # (packing is host code: works without a GPU)
    assert m._packed and m.packed_weights("cpu") is blob
    m.down3.conv2.weight.data.mul_(3.0)
    assert m.packed_weights("cpu") is blob               # stale, as documented ...
    m.invalidate_packed()
    assert not m._packed and m._fp16_verdict is None
    fresh = m.packed_weights("cpu")
    assert fresh is not blob and not torch.equal(fresh, blob)   # ... and re-packed from the edited weights after the call

