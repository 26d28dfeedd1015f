"""HardNet descriptor: the HIP path through the C ABI against the reference's golden descriptors
and against the oracle on larger batches."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet      # noqa: E402
from balf_amd.utils import synth                                      # noqa: E402
from oracle import oracle                                             # noqa: E402
from tests.golden import cases                                        # noqa: E402

pytestmark = pytest.mark.gpu
DESC_TOL = 2e-5          # max-abs on unit-norm descriptors; split-f16 operands, fp32 accumulate


@pytest.fixture(scope="module")
def hardnet():
    m = HardNet()
    m.load_state_dict(synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED))
    return m.eval().to("cuda:0")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "hardnet.npz"))


@pytest.mark.parametrize("name", list(cases.HARDNET_CASES))
def test_descriptors_match_reference_goldens(hardnet, gold, name):
    n, seed = cases.HARDNET_CASES[name]
    x = synth.synthetic_patches(n, seed).to("cuda:0")
    with torch.inference_mode():
        d = hardnet(x).cpu().numpy()
    ref = gold[name + ".desc"]
    assert d.shape == ref.shape
    assert np.abs(d - ref).max() < DESC_TOL
    assert np.abs(np.linalg.norm(d, axis=1) - 1.0).max() < 1e-5


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_descriptors_vs_oracle_fp64(hardnet, n):
    """Batch sizes around the 64-patch GEMM tile and the 4096-patch chunk of balf_hardnet_forward."""
    x = synth.synthetic_patches(n, 100 + n)
    sd64 = {k: v.double() for k, v in synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED).items()}
    pick = np.unique(np.concatenate([np.arange(min(n, 70)), np.arange(max(0, n - 70), n)]))
    ref = oracle.hardnet_forward(sd64, x[pick].double()).numpy()
    with torch.inference_mode():
        d = hardnet(x.to("cuda:0")).cpu().numpy()
    assert np.abs(d[pick] - ref).max() < DESC_TOL


def test_batch_invariance_and_determinism(hardnet):
    x = synth.synthetic_patches(300, 9).to("cuda:0")
    with torch.inference_mode():
        a = hardnet(x)
        b = hardnet(x)
        c = hardnet(x[100:101].contiguous())
    assert torch.equal(a, b)
    assert torch.equal(a[100:101], c)


def test_constant_patch_is_finite(hardnet):
    """std = 0: the reference divides by (0 + 1e-7); the normalised patch is all zeros and the descriptor is
    the (normalised) bias response -- finite, unit norm."""
    x = torch.full((3, 1, 32, 32), 0.5, device="cuda:0")
    sd = synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED)
    ref = oracle.hardnet_forward(sd, x.cpu()).numpy()
    with torch.inference_mode():
        d = hardnet(x).cpu().numpy()
    assert np.isfinite(d).all() and np.abs(d - ref).max() < DESC_TOL


def test_rejects_bad_input(hardnet):
    from balf_amd._lib import BalfHipError
    with pytest.raises(ValueError):
        hardnet(torch.zeros(2, 1, 31, 32, device="cuda:0"))
    with pytest.raises(BalfHipError):
        hardnet(torch.zeros(2, 1, 32, 32))


def test_model_moved_under_inference_mode():
    """Parameters created by .to() inside torch.inference_mode() are inference tensors (no version counter)."""
    m = HardNet()
    m.load_state_dict(synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED))
    x = synth.synthetic_patches(4, 1)
    with torch.inference_mode():
        d = m.eval().to("cuda:0")(x.to("cuda:0"))
    assert d.shape == (4, 128)


def test_plain_f16_mode(hardnet):
    """precision='fp16': one f16 product per MAC, single-plane activations.  Descriptors stay within 1e-3 of the fp64
    oracle (measured ~3e-4; cosine > 0.999999) and are deterministic; the default split mode is untouched."""
    x = synth.synthetic_patches(700, 21)
    sd64 = {k: v.double() for k, v in synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED).items()}
    ref = oracle.hardnet_forward(sd64, x.double()).numpy()
    m = HardNet()
    m.load_state_dict(synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED))
    m.precision = "fp16"
    m = m.eval().to("cuda:0")
    with torch.inference_mode():
        d = m(x.to("cuda:0"))
        d2 = m(x.to("cuda:0"))
        split = hardnet(x.to("cuda:0"))
    assert torch.equal(d, d2)
    d = d.cpu().numpy()
    assert np.abs(d - ref).max() < 1e-3
    assert (d * ref).sum(axis=1).min() > 0.99999
    assert np.abs(split.cpu().numpy() - ref).max() < DESC_TOL
    m.precision = "bf16"
    with pytest.raises(ValueError):
        m(x[:2].to("cuda:0"))


def test_masked_slots_are_exact_zeros_even_over_poisoned_workspace(hardnet):
    """Unused slots read stale workspace memory inside a partially used GEMM tile; whatever is there (NaN here), their
    descriptors must be exact zeros and the used slots unaffected."""
    from balf_amd import ops
    b, k = 3, 200
    x = synth.synthetic_patches(b * k, 5).view(b, k, 1, 32, 32).to("cuda:0")
    count = torch.tensor([200, 37, 0], dtype=torch.int32)
    with torch.inference_mode():
        full = hardnet(x.view(b * k, 1, 32, 32)).view(b, k, 128)
        ws = ops._workspace("hardnet", x.device, 1)
        ws.view(torch.float32)[: ws.numel() // 4].fill_(float("nan"))          # poison the scratch
        d = hardnet.forward_slots(x, count)
    assert not torch.isnan(d).any()
    for i, c in enumerate(count.tolist()):
        assert torch.equal(d[i, :c], full[i, :c])
        assert bool((d[i, c:] == 0).all())
