"""Conv stack of BetaVAE_H on the HIP library (cl_ica_amd/conv.py, clica_conv_*) against an fp64 evaluation of the same
nn.Sequential (/root/reference/kitti_masks/model.py:41-56: five Conv2d(k = 4) + ReLU stages): features, and the gradient of
every weight and bias, at small sizes (nc = 1 and 3, odd image counts) and at the full 2048-mask batch of BASELINE configs[4].
The reference's own numbers for this stack are the G14 goldens (tests/test_gpu_configs.py), which run through the same path."""
import numpy as np
import pytest
import torch

from conftest import PARITY, conv_formula

pytestmark = pytest.mark.gpu

_TAG = ""


@pytest.fixture(params=["f16x2", "f32"], autouse=True)
def conv_arith(request):
    """Every test of this file runs in both arithmetics of the 16C-deep stages (cl_ica_amd/conv.py: f16x2 = csrc/conv16.hip, the default;
    f32 = the fp32-MFMA kernels of csrc/linear.hip) against the same references at the same tolerances; f32 cases carry a `[conv_f32]` tag."""
    global _TAG
    from cl_ica_amd import conv
    prev = conv.set_arith(request.param)
    _TAG = "" if request.param == "f16x2" else " [conv_f32]"
    conv._POOL.clear()
    yield request.param
    conv.set_arith(prev)
    conv._POOL.clear()
    _TAG = ""

_STAGES = ((32, 4, 2, 1), (32, 4, 2, 1), (64, 4, 2, 1), (64, 4, 2, 1), (256, 4, 1, 0))


def _convs(nc, dtype=torch.float32):
    mods, width = [], nc
    for i, (out_ch, k, s, p) in enumerate(_STAGES):
        m = torch.nn.Conv2d(width, out_ch, k, s, p)
        # gain 1.6: Kaiming-uniform weights roughly halve the signal per ReLU stage; keep the last stages well away from zero
        m.weight.data = torch.tensor(conv_formula(tuple(m.weight.shape), 2 * i + 1) * np.float32(1.6))
        m.bias.data = torch.tensor(conv_formula(tuple(m.bias.shape), 2 * i + 2))
        mods.append(m.to("cuda", dtype))
        width = out_ch
    return mods


def _reference_fp64(x, convs32, dfeats, chunk=256):
    """Features and parameter gradients of the Conv2d + ReLU stack in fp64 (ATen's fp64 convolution on the GPU), in chunks of images."""
    convs = []
    for m in convs32:
        d = torch.nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding).to("cuda", torch.float64)
        d.weight.data = m.weight.data.double(); d.bias.data = m.bias.data.double()
        convs.append(d)
    feats = []
    for i0 in range(0, x.shape[0], chunk):
        h = x[i0:i0 + chunk].double()
        for m in convs:
            h = torch.relu(m(h))
        h = h.flatten(1)
        (h * dfeats[i0:i0 + chunk].double()).sum().backward()
        feats.append(h.detach())
    grads = []
    for m in convs:
        grads += [m.weight.grad, m.bias.grad]
    return torch.cat(feats), grads


def _run_hip(x, convs, dfeats):
    from cl_ica_amd.conv import conv_stack
    for m in convs:
        m.weight.grad = None; m.bias.grad = None
    feats = conv_stack(x, convs)
    feats.backward(dfeats)
    grads = []
    for m in convs:
        grads += [m.weight.grad, m.bias.grad]
    return feats.detach(), grads


def _compare(family, case, x, convs, dfeats, gate_ties=False):
    case = case + _TAG
    got_f, got_g = _run_hip(x, convs, dfeats)
    ref_f, ref_g = _reference_fp64(x, convs, dfeats)
    torch.cuda.synchronize()
    PARITY.check(family, case, "features", got_f.cpu().numpy(), ref_f.cpu().numpy())
    assert float(ref_f.abs().max()) > 1e-2 and float((ref_f > 0).float().mean()) > 0.05       # not a collapsed stack
    for i, (g, r) in enumerate(zip(got_g, ref_g)):
        name = f"stage{i // 2 + 1}." + ("weight" if i % 2 == 0 else "bias")
        assert g.shape == r.shape
        if gate_ties and i < 4:
            PARITY.check(family + "/grad", case, name, g.cpu().numpy(), r.cpu().numpy(), tol=5e-5,
                         note="ReLU gate ties at 16.8 M second-stage pre-activations: a pre-activation within fp32 rounding of zero is gated "
                              "differently by the fp32 and the fp64 forward, and one such pixel is worth ~1e-5 of max|dW| of the two stages "
                              "below it (measured 1.2e-5, unchanged by the accumulation order; the stages above it hold 1e-6)")
        else:
            PARITY.check(family + "/grad", case, name, g.cpu().numpy(), r.cpu().numpy())


@pytest.mark.parametrize("nc,images", [(1, 8), (3, 5), (1, 1)])
def test_conv_stack_small_vs_fp64(nc, images):
    g = torch.Generator().manual_seed(100 + nc + images)
    x = torch.rand(images, nc, 64, 64, generator=g).to("cuda")
    if nc == 1:
        x = (x > 0.6).float()          # binary masks, as the data set's
    dfeats = torch.randn(images, 256, generator=g).to("cuda")
    _compare("c5_conv_stack", f"nc={nc} images={images}", x, _convs(nc), dfeats)


def test_conv_stack_full_batch_vs_fp64():
    """BASELINE configs[4]'s batch: 2048 binary 64 x 64 masks."""
    g = torch.Generator().manual_seed(7)
    # blobs rather than white noise: threshold a smoothed field so that masks have the data set's large connected regions
    field = torch.nn.functional.avg_pool2d(torch.randn(2048, 1, 64, 64, generator=g), 9, 1, 4)
    x = (field > 0.05).float().to("cuda")
    # Upstream gradient of one sign pattern per feature, not white noise: with a zero-mean dfeats every weight gradient is a random-walk sum
    # over up to 592 k pixels, |sum| ~ sqrt(n) |term|, and ONE ReLU gate that the fp32 and the fp64 forward decide differently (a
    # pre-activation within rounding of zero) moves it by 1/sqrt(n) ~ 1e-3 -- measured: 3e-4 ... 1e-3 here and 2e-3 ... 5e-3 for
    # nn.Conv2d in fp32 on the same data (tools/conv_err_probe.py); that is the conditioning of the test, not of the kernels
    dfeats = ((torch.randn(2048, 256, generator=g).abs() + 0.1) / 2048).to("cuda")
    _compare("c5_conv_stack", "nc=1 images=2048", x, _convs(1), dfeats, gate_ties=True)


def test_conv_stack_buffers_are_reusable_and_switchable(monkeypatch):
    """(a) A second call on other data through the pooled buffers gives what a first call gives (nothing stale survives in the borders
    or the non-output rows); (b) a no-grad call hands its buffers back; (c) BetaVAE_H with CLICA_CONV=miopen (nn.Conv2d) agrees."""
    from cl_ica_amd import conv
    from cl_ica_amd.kitti_masks.model import BetaVAE_H
    g = torch.Generator().manual_seed(3)
    convs = _convs(1)
    xa = (torch.rand(16, 1, 64, 64, generator=g) > 0.5).float().to("cuda")
    xb = (torch.rand(16, 1, 64, 64, generator=g) > 0.3).float().to("cuda")
    d = torch.randn(16, 256, generator=g).to("cuda")
    conv._POOL.clear()
    fb0, gb0 = _run_hip(xb, convs, d)
    conv._POOL.clear()
    _run_hip(xa, convs, 3.0 * d)
    fb1, gb1 = _run_hip(xb, convs, d)
    assert torch.equal(fb0, fb1) and all(torch.equal(a, b) for a, b in zip(gb0, gb1))
    with torch.no_grad():
        conv.conv_stack(xa, convs)
    assert len(conv._POOL[(16, 1, xa.device.index)]) == 1
    net = BetaVAE_H(z_dim=5, nc=1, box_norm=True).to("cuda")
    mu_hip = net(xa)
    monkeypatch.setenv("CLICA_CONV", "miopen")
    mu_lib = net(xa)
    PARITY.check("c5_conv_stack", "BetaVAE_H hip vs nn.Conv2d" + _TAG, "mu", mu_hip.detach().cpu().numpy(), mu_lib.detach().cpu().numpy(), tol=1e-4,
                 note="fp32 MIOpen on the other side, not an fp64 reference")


@pytest.mark.parametrize("nc,images", [(1, 6), (3, 3)])
def test_conv_stack_input_gradient_vs_fp64(nc, images):
    """d loss / d image through the HIP stack (round 5: clica_conv_k4s2_dgrad_input; until then BetaVAE_H switched to nn.Conv2d / MIOpen for
    inputs that require a gradient): against the fp64 evaluation of the same nn.Sequential, together with the parameter gradients of the
    same backward pass; and BetaVAE_H has no second backend left behind a runtime condition."""
    from cl_ica_amd import conv
    from cl_ica_amd.kitti_masks.model import BetaVAE_H
    g = torch.Generator().manual_seed(40 + nc)
    convs = _convs(nc)
    x0 = torch.rand(images, nc, 64, 64, generator=g).to("cuda")
    dfeats = torch.randn(images, 256, generator=g).to("cuda")
    x = x0.clone().requires_grad_(True)
    for m in convs:
        m.weight.grad = None; m.bias.grad = None
    conv.conv_stack(x, convs).backward(dfeats)
    got_dx, got_w1 = x.grad.clone(), convs[0].weight.grad.clone()
    x64 = x0.double().requires_grad_(True)
    h = x64
    c64 = []
    for m in convs:
        d = torch.nn.Conv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding).to("cuda", torch.float64)
        d.weight.data = m.weight.data.double(); d.bias.data = m.bias.data.double()
        c64.append(d)
        h = torch.relu(d(h))
    (h.flatten(1) * dfeats.double()).sum().backward()
    PARITY.check("c5_conv_stack/grad", f"nc={nc} images={images}" + _TAG, "d loss / d image", got_dx.cpu().numpy(), x64.grad.cpu().numpy())
    PARITY.check("c5_conv_stack/grad", f"nc={nc} images={images}" + _TAG, "stage1.weight (same pass)", got_w1.cpu().numpy(), c64[0].weight.grad.cpu().numpy())
    assert float(x64.grad.abs().max()) > 0
    if nc == 1:
        import inspect
        assert "requires_grad" not in inspect.getsource(BetaVAE_H._encode), "BetaVAE_H must not pick its conv backend by the input's requires_grad"
        net = BetaVAE_H(z_dim=5, nc=1, box_norm=False).to("cuda")
        xi = x0.clone().requires_grad_(True)
        net(xi).sum().backward()
        assert xi.grad is not None and float(xi.grad.abs().max()) > 0


def test_conv_stack_full_batch_zero_mean_upstream_vs_fp64():
    """The full 2048-mask batch with a ZERO-MEAN upstream gradient (the all-positive dfeats of the test above cannot show
    cancellation).  Every weight gradient is then a random-walk sum over up to 592 k pixels, |sum| ~ sqrt(n) |term|, and ONE ReLU gate that
    the fp32 and the fp64 forward decide differently (a pre-activation within fp32 rounding of zero) moves it by ~1 / sqrt(n): that is the
    conditioning of the comparison, not of the kernels -- so the bound is stated against what fp32 nn.Conv2d / MIOpen shows ON THE SAME DATA
    against the same fp64 reference: every gradient of the HIP stack within 3 x the WORST of nn.Conv2d's ten gradients, and never beyond 3e-3
    (measured round 5: 1.3e-3 worst, nn.Conv2d 1.1e-3 worst; profiles/r5_parity_errors.json, family c5_conv_stack/zero_mean)."""
    g = torch.Generator().manual_seed(7)
    field = torch.nn.functional.avg_pool2d(torch.randn(2048, 1, 64, 64, generator=g), 9, 1, 4)
    x = (field > 0.05).float().to("cuda")
    dfeats = (torch.randn(2048, 256, generator=g) / 2048).to("cuda")
    convs = _convs(1)
    got_f, got_g = _run_hip(x, convs, dfeats)
    ref_f, ref_g = _reference_fp64(x, convs, dfeats)
    for m in convs:
        m.weight.grad = None; m.bias.grad = None
    h = x
    for m in convs:
        h = torch.relu(m(h))
    h.flatten(1).backward(dfeats)
    lib_g = []
    for m in convs:
        lib_g += [m.weight.grad, m.bias.grad]
    torch.cuda.synchronize()
    PARITY.check("c5_conv_stack/zero_mean", "nc=1 images=2048" + _TAG, "features", got_f.cpu().numpy(), ref_f.cpu().numpy())
    from conftest import rel_err
    # (per gradient the two fp32 evaluations land on different sides of the ties: compared one to one the bound flickers -- stage1.weight
    #  measured 3.4e-4 against 1.2e-4 for nn.Conv2d in one run, 1.0e-4 against 2.6e-4 in another; so the yardstick is nn.Conv2d's WORST
    #  gradient on this data)
    e_lib = max(rel_err(gl.cpu().numpy(), r.cpu().numpy()) for gl, r in zip(lib_g, ref_g))
    tol = min(3e-3, max(1e-5, 3.0 * e_lib))
    for i, (gh, r) in enumerate(zip(got_g, ref_g)):
        name = f"stage{i // 2 + 1}." + ("weight" if i % 2 == 0 else "bias")
        PARITY.check("c5_conv_stack/zero_mean", "nc=1 images=2048" + _TAG, name, gh.cpu().numpy(), r.cpu().numpy(), tol=tol,
                     note="zero-mean upstream gradient: bound = min(3e-3, 3 x the worst distance of fp32 nn.Conv2d's ten gradients from fp64 on the same data)")
