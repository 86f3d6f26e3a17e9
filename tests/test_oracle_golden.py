"""Pins oracle/ against golden vectors produced by the REAL reference (tools/make_golden.py)
and against the reference's own known-answer test.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import CFGS, checksums, det_image, det_normal
from oracle import pnpflow_oracle as O


@pytest.mark.parametrize("name,B", [("mnist", 3), ("tiny4", 2), ("celeba128", 1), ("afhq256", 1)])
def test_unet_forward_matches_reference(golden, name, B):
    g = golden("unet_" + name)
    cfg = O.unet_config(**CFGS[name])
    sd = O.synthetic_state_dict(cfg, 0)
    assert sum(v.numel() for v in sd.values()) == int(g["nparams"])
    shape = tuple(int(v) for v in g["shape"])
    x = det_normal(shape, 11)
    t = torch.from_numpy(g["t"])
    taps = {}
    with torch.no_grad():
        out = O.unet_forward(sd, cfg, x, t, taps)
    if "out" in g:
        np.testing.assert_allclose(out.numpy(), g["out"], rtol=0, atol=2e-5)
    else:
        H = shape[2]
        np.testing.assert_allclose(out[:, :, H // 2 - 16:H // 2 + 16, H // 2 - 16:H // 2 + 16].numpy(), g["out_crop"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(out[:, :, :8, :8].numpy(), g["out_corner"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(checksums(out), g["out_checksum"], rtol=1e-5)
    np.testing.assert_allclose(checksums(taps["temb"]), g["tap_temb"], rtol=1e-5)
    np.testing.assert_allclose(checksums(taps["mid"]), g["tap_mid2"], rtol=1e-5)


def test_param_counts_match_survey():
    # 128^2 net 34 473 667 params, 256^2 net 31 045 827, MNIST net 917 889
    n = lambda name: sum(int(np.prod(s)) for s in O.unet_param_shapes(O.unet_config(**CFGS[name])).values())
    assert n("celeba128") == 34473667
    assert n("afhq256") == 31045827
    assert n("mnist") == 917889


def test_reference_known_answer_box_mask():
    # pnpflow/tests/test_unit.py:14-20
    y = O.BoxInpainting(32).H(torch.ones(1, 3, 128, 128))
    torch.testing.assert_close(y[:, :, 32:64, 32:64], torch.zeros(1, 3, 32, 32))
    assert y[:, :, :32].min() == 1 and y[:, :, 96:].min() == 1


def test_degradations_match_reference(golden):
    g = golden("degradations")
    x64 = det_normal((2, 3, 64, 64), 21)
    for half in (10, 20):
        d = O.BoxInpainting(half)
        np.testing.assert_array_equal(d.H(x64).numpy(), g[f"box{half}_H"])
        np.testing.assert_array_equal(d.H_adj(x64).numpy(), g[f"box{half}_Hadj"])
    m128 = O.BoxInpainting(20).H(torch.ones(1, 1, 128, 128))[0, 0]
    z = (m128.sum(1) < 128).nonzero()
    assert [int(z.min()), int(z.max())] == list(g["box20_mask128_rows"]) == [44, 83]
    m256 = O.BoxInpainting(40).H(torch.ones(1, 1, 256, 256))[0, 0]
    z = (m256.sum(1) < 256).nonzero()
    assert [int(z.min()), int(z.max())] == list(g["box40_mask256_rows"])
    np.testing.assert_array_equal(O.RandomInpainting(0.7).H(x64).numpy(), g["rand_H"])
    m = O.random_mask_array(4, 128, 128, 0.7).astype(np.uint8)
    np.testing.assert_array_equal(np.packbits(m.reshape(-1)), g["randmask_B4_128_bits"])
    for (B, S) in ((4, 128), (32, 128), (16, 256)):
        m = O.random_mask_array(B, S, S, 0.7).astype(np.uint8)
        np.testing.assert_array_equal(m.reshape(B, -1).sum(1), g[f"randmask_B{B}_{S}_rowsum"])
        np.testing.assert_array_equal(m.reshape(B, -1)[:, :64], g[f"randmask_B{B}_{S}_first64"])
    # prefix consistency in B: shards slice one global mask
    np.testing.assert_array_equal(O.random_mask_array(32, 128, 128, 0.7)[:4], O.random_mask_array(4, 128, 128, 0.7))
    for sf in (2, 4):
        d = O.Superresolution(sf, 64)
        y = d.H(x64).contiguous()
        np.testing.assert_array_equal(y.numpy(), g[f"sr{sf}_H"])
        np.testing.assert_array_equal(d.H_adj(y).numpy(), g[f"sr{sf}_Hadj"])
        d = O.Superresolution(sf, 64, mode="bicubic")
        np.testing.assert_allclose(d.H(x64).numpy(), g[f"srbic{sf}_H"], atol=1e-6)
        np.testing.assert_allclose(d.H_adj(det_normal((2, 3, 64 // sf, 64 // sf), 24)).numpy(), g[f"srbic{sf}_Hadj"], atol=1e-6)
        # separable factorisation + roll offset used by the HIP kernel: tap K/2 of the 1-D factor sits at offset 0
        k = O.bicubic_filter(sf)[0, 0].numpy(); w = k.sum(0)
        np.testing.assert_allclose(np.outer(w, w), k, atol=1e-8)
        assert d.filter[0, 0, 0, 0] == k[2 * sf, 2 * sf]
    for sig in (1.0, 3.0):
        d = O.GaussianDeblurring(sig, 61, "fft", 3, 64)
        np.testing.assert_allclose(d.H(x64).numpy(), g[f"blur{sig}_H"], atol=1e-6)
        np.testing.assert_allclose(d.H_adj(x64).numpy(), g[f"blur{sig}_Hadj"], atol=1e-6)
    x128 = det_normal((1, 3, 128, 128), 22)
    d = O.GaussianDeblurring(1.0, 61, "fft", 3, 128)
    np.testing.assert_allclose(d.H(x128)[:, :, :16, :16].numpy(), g["blur1.0_128_H_crop"], atol=1e-6)
    x256 = det_normal((1, 3, 256, 256), 23)
    d = O.GaussianDeblurring(3.0, 61, "fft", 3, 256)
    np.testing.assert_allclose(d.H(x256)[:, :, :16, :16].numpy(), g["blur3.0_256_H_crop"], atol=1e-6)
    np.testing.assert_allclose(d.H_adj(x256)[:, :, 120:136, 120:136].numpy(), g["blur3.0_256_Hadj_crop"], atol=1e-6)
    np.testing.assert_allclose(O.gaussian_2d_kernel(1.0, 61)[25:36, 25:36].numpy(), g["gauss2d_1.0_61_center"], atol=1e-9)
    # separable factorisation used by the HIP blur kernel
    for sig in (1.0, 3.0):
        g1 = O.gaussian_1d_taps(sig, 61)
        np.testing.assert_allclose(np.outer(g1, g1), O.gaussian_2d_kernel(sig, 61).numpy(), atol=1e-7)


def test_adjoint_identity():
    x = det_normal((2, 3, 64, 64), 5)
    for d, yshape in ((O.BoxInpainting(10), (2, 3, 64, 64)), (O.RandomInpainting(0.7), (2, 3, 64, 64)),
                      (O.Superresolution(2, 64), (2, 3, 32, 32)), (O.Superresolution(4, 64, mode="bicubic"), (2, 3, 16, 16)), (O.GaussianDeblurring(3.0, 61, "fft", 3, 64), (2, 3, 64, 64))):
        y = det_normal(yshape, 6)
        a = (d.H(x).double() * y.double()).sum(); b = (x.double() * d.H_adj(y).double()).sum()
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a))


TRAJ = [("mnist_denoising", "mnist", lambda S: (O.Denoising(), 0.2)),
        ("tiny4_inpainting", "tiny4", lambda S: (O.BoxInpainting(10), 0.05)),
        ("tiny4_superresolution", "tiny4", lambda S: (O.Superresolution(2, S), 0.05)),
        ("tiny4_deblurring", "tiny4", lambda S: (O.GaussianDeblurring(1.0, 61, "fft", 3, S), 0.05)),
        ("tiny4_random_inpainting", "tiny4", lambda S: (O.RandomInpainting(0.7), 0.01)),
        ("tiny4_superresolution_bicubic", "tiny4", lambda S: (O.Superresolution(2, S, mode="bicubic"), 0.05))]


def det_laplace(shape, scale):
    """Same deterministic Laplace draw as tools/make_golden.py (inverse CDF of a Philox uniform)."""
    g = np.random.Generator(np.random.Philox(key=[41, 0]))
    u = torch.from_numpy(g.uniform(-0.5, 0.5, size=shape).astype(np.float32))
    return -scale * torch.sign(u) * torch.log1p(-2 * u.abs())


LAPLACE = [("laplace_tiny4_superresolution", lambda S: (O.Superresolution(2, S), 0.3)),
           ("laplace_tiny4_deblurring", lambda S: (O.GaussianDeblurring(1.0, 61, "fft", 3, S), 0.3)),
           ("laplace_tiny4_inpainting", lambda S: (O.BoxInpainting(10), 0.3))]


@pytest.mark.parametrize("tag,mk", LAPLACE)
def test_pnp_flow_laplace_trajectory_matches_reference(golden, tag, mk):
    g = golden("pnp_traj_" + tag)
    cfg = O.unet_config(**CFGS["tiny4"]); sd = O.synthetic_state_dict(cfg, 0)
    S, C = cfg["input_height"], cfg["input_channels"]
    degradation, sigma = mk(S)
    steps, ns = int(g["steps"]), int(g["num_samples"])
    clean = det_image((2, C, S, S), 31)
    hx = degradation.H(clean.clone())
    y = hx + det_laplace(tuple(hx.shape), torch.tensor(sigma)) * 1.0      # the reference adds Laplace(0, sigma) directly (pnp_flow.py:83-85)
    np.testing.assert_allclose(y.numpy(), g["noisy"], atol=1e-6)
    its = {}
    O.pnp_flow_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), degradation, y, sigma, steps=steps, num_samples=ns, alpha=float(g["alpha"]),
                       noise_fn=lambda it, s, like: det_normal(tuple(like.shape), 41, 1 + it * ns + s),
                       record=lambda it, xx: its.__setitem__(it, xx.clone()), noise_type="laplace")
    for it in (0, 1, 4, 9):
        np.testing.assert_allclose(its[it].numpy(), g[f"x_it{it}"], atol=2e-5, err_msg=f"iterate {it}")
    assert abs(float(g["lr_pnp_after"]) - sigma) < 1e-12      # lr_pnp scaled by sigma (not sigma^2) in place (pnp_flow.py:65)


@pytest.mark.parametrize("tag,net,mk", TRAJ)
def test_pnp_flow_trajectory_matches_reference(golden, tag, net, mk):
    g = golden("pnp_traj_" + tag)
    cfg = O.unet_config(**CFGS[net])
    sd = O.synthetic_state_dict(cfg, 0)
    S, C = cfg["input_height"], cfg["input_channels"]
    degradation, sigma = mk(S)
    assert sigma == float(g["sigma"])
    steps, ns = int(g["steps"]), int(g["num_samples"])
    clean = det_image((2, C, S, S), 31)
    y = O.make_measurement(clean, degradation, sigma, batch=0, noise=det_normal(tuple(degradation.H(clean).shape), 41, 0))
    np.testing.assert_allclose(y.numpy(), g["noisy"], atol=1e-6)
    its = {}
    x = O.pnp_flow_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), degradation, y, sigma, steps=steps, num_samples=ns,
                           alpha=float(g["alpha"]), noise_fn=lambda it, s, like: det_normal(tuple(like.shape), 41, 1 + it * ns + s),
                           record=lambda it, xx: its.__setitem__(it, xx.clone()))
    for it in (0, 1, 4, 9):
        np.testing.assert_allclose(its[it].numpy(), g[f"x_it{it}"], atol=2e-5, err_msg=f"iterate {it}")
    # the reference scales args.lr_pnp by sigma^2 in place (pnp_flow.py:61)
    assert abs(float(g["lr_pnp_after"]) - sigma ** 2) < 1e-12


def test_psnr_formula():
    a = torch.zeros(2, 3, 8, 8); b = torch.zeros(2, 3, 8, 8)
    b[0] += 0.2; b[1] += 0.02          # in [-1,1] units -> 0.1 / 0.01 after postprocess
    p = O.psnr_per_image(b, a)
    np.testing.assert_allclose(p.numpy(), [20.0, 40.0], atol=1e-4)


def test_engine_rng_restatement_statistics():
    z = O.engine_normal(1 << 16, seed=1234, stream=5)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    # known-answer of Philox4x32-10 (Random123 kat_vectors): ctr=0,key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    r = O.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    r = O.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full(2, 0xFFFFFFFF, np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


# ---- OT-ODE (pnpflow/methods/ot_ode.py) ------------------------------------------------------
@pytest.mark.parametrize("net", ["mnist", "tiny4"])
def test_unet_vjp_matches_reference(golden, net):
    g = golden("vjp_" + net)
    cfg = O.unet_config(**CFGS[net]); sd = O.synthetic_state_dict(cfg, 0)
    S, C = cfg["input_height"], cfg["input_channels"]
    x = det_normal((2, C, S, S), 51); vec = det_normal((2, C, S, S), 52)
    out = O.unet_vjp(sd, cfg, x, torch.from_numpy(g["t"]), vec)
    np.testing.assert_allclose(out.numpy(), g["g"], atol=2e-5)


OT_CASES = [("tiny4_random_inpainting", "tiny4", "random_inpainting", lambda S: (O.RandomInpainting(0.7), 0.01), 0.1, "constant"),
            ("tiny4_inpainting", "tiny4", "inpainting", lambda S: (O.BoxInpainting(10), 0.05), 0.1, "gamma_t"),
            ("tiny4_superresolution", "tiny4", "superresolution", lambda S: (O.Superresolution(2, S), 0.05), 0.1, "constant"),
            ("mnist_denoising", "mnist", "denoising", lambda S: (O.Denoising(), 0.2), 0.3, "gamma_t"),
            ("tiny4_gaussian_deblurring_FFT", "tiny4", "gaussian_deblurring_FFT", lambda S: (O.GaussianDeblurring(1.0, 61, "fft", 3, S), 0.05), 0.1,
             "constant")]


@pytest.mark.parametrize("tag,net,problem,mk,t0,gamma", OT_CASES)
def test_ot_ode_trajectory_matches_reference(golden, tag, net, problem, mk, t0, gamma):
    g = golden("ot_ode_traj_" + tag)
    cfg = O.unet_config(**CFGS[net]); sd = O.synthetic_state_dict(cfg, 0)
    S, C = cfg["input_height"], cfg["input_channels"]
    degradation, sigma = mk(S)
    steps = int(g["steps"])
    clean = det_image((2, C, S, S), 31)
    y = O.make_measurement(clean, degradation, sigma, 0, noise=det_normal(tuple(degradation.H(clean).shape), 61, 0))
    np.testing.assert_allclose(y.numpy(), g["noisy"], atol=1e-6)
    its = {}
    x = O.ot_ode_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), lambda a, t, v: O.unet_vjp(sd, cfg, a, t, v), degradation, problem,
                         y, sigma, steps=steps, start_time=t0, gamma=gamma,
                         init_noise=det_normal(tuple(degradation.H_adj(y).shape), 61, 1), record=lambda it, xx: its.__setitem__(it, xx.clone()))
    first = int(g["first"])
    for it in (first, first + 1, steps - 1):
        ref = g[f"x_it{it}"]
        np.testing.assert_allclose(its[it].numpy(), ref, atol=1e-4 * max(1.0, float(np.abs(ref).max())), err_msg=f"iterate {it}")


# ---- BASELINE-size nets: C4/C5 --------------------------------------------
def _check_crops(t, g, prefix, atol, rtol_sum=1e-5):
    H = t.shape[2]
    np.testing.assert_allclose(t[:, :, H // 2 - 16:H // 2 + 16, H // 2 - 16:H // 2 + 16].numpy(), g[prefix + "_crop"], atol=atol)
    np.testing.assert_allclose(t[:, :, :8, :8].numpy(), g[prefix + "_corner"], atol=atol)
    np.testing.assert_allclose(checksums(t)[1:], g[prefix + "_checksum"][1:], rtol=rtol_sum)     # sum|.|, sum .^2 (the plain sum cancels)


@pytest.mark.parametrize("net", ["celeba128", "afhq256"])
def test_unet_vjp_full_size_matches_reference(golden, net):
    """J^T vec on the 34.5 M / 31.0 M parameter nets vs the real reference's torch.autograd.functional.vjp (ot_ode.py:137-138)."""
    g = golden("vjp_" + net)
    cfg = O.unet_config(**CFGS[net]); sd = O.synthetic_state_dict(cfg, 0)
    S = cfg["input_height"]
    x = det_normal((1, 3, S, S), 51); vec = det_normal((1, 3, S, S), 52)
    out = O.unet_vjp(sd, cfg, x, torch.from_numpy(g["t"]), vec)
    _check_crops(out, g, "g", 2e-5 * float(g["g_absmax"]))


def test_ot_ode_c5_step_matches_reference(golden):
    """One real OT_ODE.solve_ip Euler step of config C5 (AFHQ-256 random inpainting, steps_ode=100, start_time=0.1, gamma constant)."""
    g = golden("ot_ode_step_afhq256_random_inpainting")
    cfg = O.unet_config(**CFGS["afhq256"]); sd = O.synthetic_state_dict(cfg, 0)
    B, S, sigma = int(g["B"]), 256, float(g["sigma"])
    degradation = O.RandomInpainting(0.7)
    clean = det_image((B, 3, S, S), 31)
    y = O.make_measurement(clean, degradation, sigma, 0, noise=det_normal((B, 3, S, S), 61, 0))
    _check_crops(y, g, "noisy", 1e-6)

    class _Stop(Exception):
        pass
    its = {}

    def rec(it, xx):
        its[it] = xx.clone()
        raise _Stop()
    try:
        O.ot_ode_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), lambda a, t, v: O.unet_vjp(sd, cfg, a, t, v), degradation,
                         "random_inpainting", y, sigma, steps=100, start_time=0.1, gamma="constant",
                         init_noise=det_normal((B, 3, S, S), 61, 1), record=rec)
    except _Stop:
        pass
    _check_crops(its[10], g, "x_it10", 2e-5 * float(np.abs(g["x_it10_crop"]).max()))


# ---- the reference's native ops (NCSN++ net): op/upfirdn2d.py, op/fused_act.py ------------------------------------------------
UPFIRDN_CASES = ["fir4_up2", "fir4_down2", "asym3x2_up3_down2_crop", "k1_identity", "k5_pad"]


@pytest.mark.parametrize("name", UPFIRDN_CASES)
def test_upfirdn2d_matches_reference(golden, name):
    g = golden("native_ops")
    x = det_normal((2, 3, 20, 24), 71)
    ux, uy, dx, dy, px0, px1, py0, py1 = [int(v) for v in g[name + "_p"]]
    out = O.upfirdn2d(x, torch.from_numpy(g[name + "_k"]), ux, uy, dx, dy, px0, px1, py0, py1)
    assert tuple(out.shape) == g[name + "_out"].shape
    np.testing.assert_allclose(out.numpy(), g[name + "_out"], atol=1e-6)


def test_fir_resampling_and_fused_act_match_reference(golden):
    g = golden("native_ops")
    x2 = det_normal((2, 4, 16, 16), 72)
    np.testing.assert_allclose(O.upsample_2d(x2, (1, 3, 3, 1), 2).numpy(), g["upsample_2d_1331"], atol=1e-6)
    np.testing.assert_allclose(O.downsample_2d(x2, (1, 3, 3, 1), 2).numpy(), g["downsample_2d_1331"], atol=1e-6)
    np.testing.assert_allclose(O.upsample_2d(x2, None, 2).numpy(), g["upsample_2d_default"], atol=1e-6)
    np.testing.assert_allclose(O.downsample_2d(x2, None, 2).numpy(), g["downsample_2d_default"], atol=1e-6)
    xb = det_normal((2, 5, 6, 7), 73); bias = det_normal((5,), 74)
    np.testing.assert_allclose(O.fused_leaky_relu(xb, bias).numpy(), g["fused_leaky_relu"], atol=1e-6)
    np.testing.assert_allclose(O.fused_leaky_relu(det_normal((3, 5), 75), bias).numpy(), g["fused_leaky_relu_2d"], atol=1e-6)


# ---- NCSN++ ("rectified") velocity net: oracle/ncsnpp_oracle.py vs the REAL reference module -----------------------------------
NCSNPP_CFGS = {
    "tiny": dict(image_size=32, nf=32, ch_mult=(1, 1, 2), num_res_blocks=2, attn_resolutions=(16,)),
    "wide": dict(image_size=32, nf=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,)),
    "afhq256": dict(image_size=256, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,)),
}


@pytest.mark.parametrize("name", ["tiny", "wide", "afhq256"])
def test_ncsnpp_oracle_matches_reference_module(golden, name):
    """tools/make_golden.py `gen_ncsnpp` ran pnpflow/image_generation/models/ncsnpp.py NCSNpp.forward on these weights / inputs."""
    from oracle import ncsnpp_oracle as NO
    c = NCSNPP_CFGS[name]
    cfg = NO.ncsnpp_config(**c)
    sd = NO.synthetic_state_dict(cfg, 0)
    g = golden(f"ncsnpp_{name}")
    B = 1 if name == "afhq256" else 2
    x = det_normal((B, 3, c["image_size"], c["image_size"]), 81)
    t = torch.from_numpy(g["t"])
    taps = {}
    y = NO.ncsnpp_forward(sd, cfg, x, t * 999, taps)
    tol = 1e-6 * float(g["y_absmax"])          # same ops in the same order: observed 0
    if "g" in g.files or "g_crop" in g.files:      # the VJP OT_ODE takes, by autograd through the real module
        gv = NO.ncsnpp_vjp(sd, cfg, x, t * 999, det_normal((B, 3, c["image_size"], c["image_size"]), 87))
        if name == "afhq256":
            _check_crops(gv, g, "g", 1e-6 * float(g["g_absmax"]))
        else:
            np.testing.assert_allclose(gv.numpy(), g["g"], atol=1e-6 * float(g["g_absmax"]))
    if name == "afhq256":
        assert sum(int(np.prod(s)) for s in NO.ncsnpp_param_shapes(cfg).values()) == 65574549      # the reference prints it (models/utils.py:97-100); + 2000 sigmas
        _check_crops(y, g, "y", tol)
    else:
        np.testing.assert_allclose(y.numpy(), g["y"], atol=tol)
        for k in g.files:
            if k.startswith("tap_"):
                np.testing.assert_allclose(taps[k[4:]].numpy(), g[k], atol=1e-6)


def test_oracle_ot_ode_generic_gmres_branch_matches_reference(golden):
    """ot_ode.py:118-128 + utils.py:972-1109 (GMRES): the real reference's iterates for a problem name outside the closed forms."""
    g = golden("ot_ode_traj_tiny4_deblurring_gmres")
    cfg = O.unet_config(**CFGS["tiny4"]); sd = O.synthetic_state_dict(cfg, 0)
    S = cfg["input_height"]; steps = int(g["steps"])
    its = {}
    O.ot_ode_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), lambda a, t, v: O.unet_vjp(sd, cfg, a, t, v),
                     O.GaussianDeblurring(1.0, 61, "fft", 3, S), "gaussian_deblurring", torch.from_numpy(g["noisy"]), float(g["sigma"]),
                     steps=steps, start_time=float(g["start_time"]), gamma="constant", init_noise=det_normal((2, 3, S, S), 61, 1),
                     record=lambda it, xx: its.__setitem__(it, xx.clone()))
    for it in (int(g["first"]), int(g["first"]) + 1, steps - 1):
        np.testing.assert_allclose(its[it].numpy(), g[f"x_it{it}"], atol=1e-5)


# ---- full-length recursions of the real solvers ------------------------------------
def _oracle_pnp(net, degradation, sigma, g, B, upto=None):
    cfg = O.unet_config(**CFGS[net]); sd = O.synthetic_state_dict(cfg, 0)
    S, C = cfg["input_height"], cfg["input_channels"]
    steps, ns = int(g["steps"]), int(g["num_samples"])
    clean = det_image((B, C, S, S), 31)
    y = O.make_measurement(clean, degradation, sigma, batch=0, noise=det_normal(tuple(degradation.H(clean).shape), 41, 0))
    its = {}

    class _Stop(Exception):
        pass

    def rec(it, xx):
        its[it] = xx.clone()
        if upto is not None and it >= upto:
            raise _Stop()
    try:
        O.pnp_flow_restore(lambda a, t: O.unet_forward(sd, cfg, a, t), degradation, y, sigma, steps=steps, num_samples=ns, alpha=float(g["alpha"]),
                           noise_fn=lambda it, s, like: det_normal(tuple(like.shape), 41, 1 + it * ns + s), record=rec)
    except _Stop:
        pass
    return its, y, clean


def test_pnp_flow_100x5_trajectory_matches_reference(golden):
    """The shipped loop length (pnp_flow.py:103-121: 100 outer iterations x 5 samples) on the 4-level test net: the oracle follows
    the real reference's iterates to the end, PSNR of the final iterate within the north_star's 0.05 dB."""
    g = golden("pnp_long_tiny4_inpainting")
    its, y, clean = _oracle_pnp("tiny4", O.BoxInpainting(10), 0.05, g, 2)
    np.testing.assert_allclose(y.numpy(), g["noisy"], atol=1e-6)
    for it in (0, 10, 50, 90, 99):
        np.testing.assert_allclose(its[it].numpy(), g[f"x_it{it}"], atol=2e-4, err_msg=f"iterate {it}")
    d = (O.psnr_per_image(its[99], clean) - O.psnr_per_image(torch.from_numpy(g["x_it99"]), clean)).abs().max()
    assert float(d) <= 0.05


def test_c1_mnist_denoising_50x5_matches_reference(golden):
    """BASELINE configs[0] at its own size: MNIST-shaped denoising, B = 8, 50 x 5, iterates 0, 5, ..., 45, 49."""
    g = golden("pnp_long_mnist_c1")
    its, y, clean = _oracle_pnp("mnist", O.Denoising(), 0.2, g, 8)
    for it in list(range(0, 50, 5)) + [49]:
        np.testing.assert_allclose(its[it].numpy(), g[f"x_it{it}"], atol=1e-4, err_msg=f"iterate {it}")


@pytest.mark.parametrize("tag,net,mk", [("c2", "celeba128", lambda S: (O.BoxInpainting(20), 0.05)),
                                        ("c3", "celeba128", lambda S: (O.GaussianDeblurring(1.0, 61, "fft", 3, S), 0.05))])
def test_first_outer_iterations_of_baseline_configs_match_reference(golden, tag, net, mk):
    """The first two outer iterations of BASELINE configs[1] / [2] (34.5 M-parameter net, their own operator parameters, the 100 x 5
    schedule) against the real reference.  (C4's 256^2 fixture is checked on the GPU only: 10 forwards of the 256^2
    net are minutes on these host cores.)"""
    g = golden("pnp_iter_" + tag)
    degradation, sigma = mk(128)
    its, y, clean = _oracle_pnp(net, degradation, sigma, g, 2, upto=1)
    _check_crops(y, g, "noisy", 1e-6)
    _check_crops(its[0], g, "x_it0", 5e-5)
    _check_crops(its[1], g, "x_it1", 5e-5)


# ---------------------------------------------------------------------------------------------
# third-party pins (tools/pin_thirdparty.py): picked up when present, "unpinned" in the skip reason otherwise
# ---------------------------------------------------------------------------------------------
def test_oracle_ssim_matches_thirdparty_ssim(thirdparty):
    """oracle SSIM vs ignite.metrics.SSIM(data_range=1.0) itself (pnpflow/utils.py:780-816)."""
    from conftest import thirdparty_pair
    g = thirdparty("ssim", "pytorch-ignite")
    for i, shape in enumerate(g["shapes"]):
        a, b = thirdparty_pair(tuple(int(v) for v in shape), int(g["seed"]))
        got = O.ssim_per_image(b, a).double().numpy()
        np.testing.assert_allclose(got, g[f"per_image_{i}"], atol=2e-5)
        np.testing.assert_allclose(got.mean(), float(g[f"batch_{i}"]), atol=2e-5)


def test_oracle_lpips_matches_thirdparty_lpips(thirdparty):
    """oracle LPIPS vs lpips.LPIPS(net='alex')(a, b, normalize=True) on the published weights (pnpflow/utils.py:677-724; north_star +-1e-3)."""
    from conftest import thirdparty_pair
    g = thirdparty("lpips", "lpips + torchvision")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    from pnpflow_amd.lpips import canonical_state_dict
    sd = canonical_state_dict(sd)
    for i, shape in enumerate(g["shapes"]):
        a, b = thirdparty_pair(tuple(int(v) for v in shape), int(g["seed"]))
        np.testing.assert_allclose(O.lpips_forward(sd, a, b, normalize=True).double().numpy(), g[f"d_{i}"], atol=1e-4)


def test_paintbrush_raster_matches_thirdparty_paintbrush(thirdparty):
    """product and oracle rasters vs cv2.line itself on the reference's seeded stroke sequence (pnpflow/utils.py:339-350, :904-969): bit-exact."""
    from pnpflow_amd.degradations import paintbrush_masks
    g = thirdparty("paintbrush", "opencv-python")
    for i, (B, H, W) in enumerate(g["shapes"]):
        np.testing.assert_array_equal(paintbrush_masks(int(B), int(H), int(W)), g[f"keep_{i}"])
        np.testing.assert_array_equal(O.paintbrush_mask_array(int(B), int(H), int(W)), g[f"keep_{i}"])


def test_biglong_fixtures_are_consistent_with_the_oracle_pieces(golden):
    """The B = 1 full-length fixtures of the BASELINE nets (tools/make_golden.py biglong): the stored measurement is the oracle's
    operator applied to the recipe image + the recipe noise, and the stored PSNR is the oracle's PSNR of the stored final iterate
    (the 500-evaluation recursion itself is followed on the GPU: tests/test_gpu_parity.py)."""
    from conftest import det_image, det_normal
    cases = {"c2": (128, O.BoxInpainting(20)), "c2_256": (256, O.BoxInpainting(40)), "c4": (256, O.Superresolution(4, 256))}
    for tag, (S, deg) in cases.items():
        g = golden("pnp_biglong_" + tag)
        clean = det_image((1, 3, S, S), 31)
        y = deg.H(clean)
        y = y + float(g["sigma"]) * det_normal(tuple(y.shape), 41, 0)
        np.testing.assert_allclose(y.numpy(), g["noisy"], atol=1e-6, err_msg=tag)
        np.testing.assert_allclose(O.psnr_per_image(torch.from_numpy(g["x_final"]), clean).numpy(), g["psnr_final"], atol=1e-4)
