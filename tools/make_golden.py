"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) here.

Run in the build container only:   python tools/make_golden.py
The fixtures hold inputs-by-recipe (seeded numpy Philox, see `det_normal`), the
reference's outputs (full tensors for small cases, crops + float64 checksums for the
128^2 / 256^2 nets) and nothing else - no reference source travels.

What each fixture pins:
  unet_*.npz        UNet.forward of pnpflow/models.py:442-495 on synthetic weights
  degradations.npz  H / H_adj of pnpflow/degradations.py + helper masks (utils.py:327-361)
  pnp_traj_*.npz    PNP_FLOW.solve_ip (pnpflow/methods/pnp_flow.py:54-172) iterates with
                    torch.randn_like replaced by a supplied noise sequence
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ref_import import import_reference  # noqa: E402
from oracle import pnpflow_oracle as O  # noqa: E402  (only for configs / weight recipe / det inputs)

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def det_normal(shape, seed, idx=0):
    """Deterministic N(0,1) float32 tensor from numpy Philox(key=[seed, idx])."""
    g = np.random.Generator(np.random.Philox(key=[seed, idx]))
    return torch.from_numpy(g.standard_normal(size=shape, dtype=np.float32))


def det_image(shape, seed):
    """Smooth-ish synthetic clean image in [-1,1]: randn -> 5x 3x3 box blur
    -> per-image min-max."""
    x = det_normal(shape, seed, 7)
    k = torch.ones(shape[1], 1, 3, 3) / 9.0
    for _ in range(5):
        x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), k, groups=shape[1])
    lo = x.amin(dim=(1, 2, 3), keepdim=True); hi = x.amax(dim=(1, 2, 3), keepdim=True)
    return ((x - lo) / (hi - lo) * 2 - 1).contiguous()


def checksums(t):
    d = t.double()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()], dtype=np.float64)


CFGS = {
    "mnist": dict(input_channels=1, input_height=28, ch=32, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(16,)),
    "tiny4": dict(input_channels=3, input_height=64, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=1, attn_resolutions=(16, 8)),
    "celeba128": dict(input_channels=3, input_height=128, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)),
    "afhq256": dict(input_channels=3, input_height=256, ch=32, ch_mult=(1, 2, 4, 8), num_res_blocks=6, attn_resolutions=(16, 8)),
}


def build_ref_unet(models, name):
    c = CFGS[name]
    cfg = O.unet_config(**c)
    sd = O.synthetic_state_dict(cfg, seed=0)
    m = models.UNet(c["input_channels"], c["input_height"], c["ch"], ch_mult=c["ch_mult"],
                    num_res_blocks=c["num_res_blocks"], attn_resolutions=c["attn_resolutions"])
    assert list(m.state_dict().keys()) == list(sd.keys()), "state_dict key order mismatch"
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m, cfg, sd


def gen_unet(models):
    for name, B in (("mnist", 3), ("tiny4", 2), ("celeba128", 1), ("afhq256", 1)):
        m, cfg, sd = build_ref_unet(models, name)
        c = CFGS[name]
        shape = (B, c["input_channels"], c["input_height"], c["input_height"])
        x = det_normal(shape, 11)
        t = torch.tensor([0.0, 0.37, 0.99][:B], dtype=torch.float32)
        taps = {}
        hooks = []
        # forward hooks on the top-level stages give per-stage checksums
        def mk(nm):
            def hook(mod, inp, out):
                taps[nm] = checksums(out)
            return hook
        hooks.append(m.begin_conv.register_forward_hook(mk("begin_conv")))
        hooks.append(m.temb_net.register_forward_hook(mk("temb")))
        for i, md in enumerate(m.mid_modules):
            hooks.append(md.register_forward_hook(mk(f"mid{i}")))
        with torch.no_grad():
            out = m(x, t)
        for h in hooks:
            h.remove()
        rec = dict(t=t.numpy(), shape=np.array(shape), out_checksum=checksums(out),
                   nparams=np.array(sum(v.numel() for v in sd.values())))
        for k, v in taps.items():
            rec["tap_" + k] = v
        if name in ("mnist", "tiny4"):
            rec["out"] = out.numpy()
        else:
            H = c["input_height"]
            rec["out_crop"] = out[:, :, H // 2 - 16:H // 2 + 16, H // 2 - 16:H // 2 + 16].numpy()
            rec["out_corner"] = out[:, :, :8, :8].numpy()
        np.savez_compressed(os.path.join(OUT, f"unet_{name}.npz"), **rec)
        print("unet", name, out.abs().mean().item(), out.std().item())


def gen_degradations(degr, utils):
    rec = {}
    x64 = det_normal((2, 3, 64, 64), 21)
    # the reference's only known-answer test (pnpflow/tests/test_unit.py:14-20)
    y = degr.BoxInpainting(32).H(torch.ones(1, 3, 128, 128))
    rec["box32_ones128_zero_rows"] = np.array([int((y[0, 0].sum(1) == 64).nonzero().min()), int((y[0, 0].sum(1) == 64).nonzero().max())])
    for half in (10, 20):
        d = degr.BoxInpainting(half)
        rec[f"box{half}_H"] = d.H(x64).numpy(); rec[f"box{half}_Hadj"] = d.H_adj(x64).numpy()
    d = degr.BoxInpainting(20)
    m128 = d.H(torch.ones(1, 1, 128, 128))[0, 0]
    rec["box20_mask128_rows"] = np.array([int((m128.sum(1) < 128).nonzero().min()), int((m128.sum(1) < 128).nonzero().max())])
    d = degr.BoxInpainting(40)
    m256 = d.H(torch.ones(1, 1, 256, 256))[0, 0]
    rec["box40_mask256_rows"] = np.array([int((m256.sum(1) < 256).nonzero().min()), int((m256.sum(1) < 256).nonzero().max())])
    # random mask: exact bit patterns
    d = degr.RandomInpainting(0.7)
    rec["rand_H"] = d.H(x64).numpy()
    for (B, S) in ((4, 128), (32, 128), (16, 256)):
        m = d.H(torch.ones(B, 1, S, S))[:, 0].numpy().astype(np.uint8)
        if B == 4:
            rec["randmask_B4_128_bits"] = np.packbits(m.reshape(-1))
        rec[f"randmask_B{B}_{S}_rowsum"] = m.reshape(B, -1).sum(1).astype(np.int64)
        rec[f"randmask_B{B}_{S}_first64"] = m.reshape(B, -1)[:, :64].copy()
    # superresolution
    for sf, S in ((2, 64), (4, 64)):
        d = degr.Superresolution(sf, S, device="cpu")
        ylow = d.H(x64)
        rec[f"sr{sf}_H"] = ylow.contiguous().numpy(); rec[f"sr{sf}_Hadj"] = d.H_adj(ylow.contiguous()).numpy()
    # bicubic superresolution (degradations.py:97-127; only reachable through the class API, main.py uses mode=None)
    for sf, S in ((2, 64), (4, 64)):
        d = degr.Superresolution(sf, S, mode="bicubic", device="cpu")
        ylow = d.H(x64)
        rec[f"srbic{sf}_H"] = ylow.contiguous().numpy(); rec[f"srbic{sf}_Hadj"] = d.H_adj(det_normal((2, 3, 64 // sf, 64 // sf), 24)).numpy()
    # gaussian deblurring (FFT, circular)
    for sig, S in ((1.0, 64), (3.0, 64)):
        d = degr.GaussianDeblurring(sig, 61, "fft", 3, S, "cpu")
        rec[f"blur{sig}_H"] = d.H(x64).numpy(); rec[f"blur{sig}_Hadj"] = d.H_adj(x64).numpy()
    x128 = det_normal((1, 3, 128, 128), 22)
    d = degr.GaussianDeblurring(1.0, 61, "fft", 3, 128, "cpu")
    rec["blur1.0_128_H_crop"] = d.H(x128)[:, :, :16, :16].numpy()
    rec["blur1.0_128_H_checksum"] = checksums(d.H(x128))
    x256 = det_normal((1, 3, 256, 256), 23)
    d = degr.GaussianDeblurring(3.0, 61, "fft", 3, 256, "cpu")
    rec["blur3.0_256_H_crop"] = d.H(x256)[:, :, :16, :16].numpy()
    rec["blur3.0_256_Hadj_crop"] = d.H_adj(x256)[:, :, 120:136, 120:136].numpy()
    rec["blur3.0_256_H_checksum"] = checksums(d.H(x256))
    rec["gauss2d_1.0_61_center"] = utils.gaussian_2d_kernel(1.0, 61)[25:36, 25:36].numpy()
    rec["gauss2d_3.0_61_row30"] = utils.gaussian_2d_kernel(3.0, 61)[30].numpy()
    np.savez_compressed(os.path.join(OUT, "degradations.npz"), **rec)
    print("degradations ok")


def gen_traj(models, degr, utils, pnp):
    """Run the real PNP_FLOW.solve_ip with torch.randn_like replaced by a supplied sequence."""
    cases = [
        ("mnist_denoising", "mnist", "denoising", lambda S: (degr.Denoising(), 0.2), 0.8, 2),
        ("tiny4_inpainting", "tiny4", "inpainting", lambda S: (degr.BoxInpainting(10), 0.05), 0.5, 2),
        ("tiny4_superresolution", "tiny4", "superresolution", lambda S: (degr.Superresolution(2, S, device="cpu"), 0.05), 0.3, 2),
        ("tiny4_deblurring", "tiny4", "gaussian_deblurring_FFT", lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, "cpu"), 0.05), 0.01, 2),
        ("tiny4_random_inpainting", "tiny4", "random_inpainting", lambda S: (degr.RandomInpainting(0.7), 0.01), 0.01, 2),
        ("tiny4_superresolution_bicubic", "tiny4", "superresolution", lambda S: (degr.Superresolution(2, S, mode="bicubic", device="cpu"), 0.05), 0.3, 2),
        # Laplace noise model (pnp_flow.py:42-43, 64-66, 81-85; sigma 0.3 from main.py:121-176)
        ("laplace_tiny4_superresolution", "tiny4", "superresolution", lambda S: (degr.Superresolution(2, S, device="cpu"), 0.3), 0.3, 2),
        ("laplace_tiny4_deblurring", "tiny4", "gaussian_deblurring_FFT", lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, "cpu"), 0.3), 0.01, 2),
        ("laplace_tiny4_inpainting", "tiny4", "inpainting", lambda S: (degr.BoxInpainting(10), 0.3), 0.5, 2),
    ]
    steps, num_samples = 10, 2
    for tag, net, problem, mk, alpha, B in cases:
        m, cfg, sd = build_ref_unet(models, net)
        c = CFGS[net]
        S = c["input_height"]
        degradation, sigma = mk(S)
        clean = det_image((B, c["input_channels"], S, S), 31)
        laplace = tag.startswith("laplace")
        args = utils.CfgNode(dict(method="pnp_flow", model="ot", dataset="celeba", problem=problem,
                                  noise_type="laplace" if laplace else "gaussian", num_samples=num_samples, steps_pnp=steps, lr_pnp=1.0,
                                  gamma_style="alpha_1_minus_t", alpha=alpha, max_batch=1, compute_time=False,
                                  compute_memory=False, save_results=True, batch=0, save_path_ip="/tmp"))
        iterates = {}
        seq = {"n": 0}

        def fake_randn_like(like, **kw):
            i = seq["n"]; seq["n"] += 1
            return det_normal(tuple(like.shape), 41, i)   # call 0 = measurement noise, then (it, sample) order

        def fake_laplace_sample(self_, sample_shape=torch.Size()):
            # deterministic Laplace(0, scale): call 0 of the noise sequence; inverse-CDF of a det uniform
            seq["n"] += 1
            g = np.random.Generator(np.random.Philox(key=[41, 0]))
            u = torch.from_numpy(g.uniform(-0.5, 0.5, size=tuple(self_.loc.shape)).astype(np.float32))
            return self_.loc - self_.scale * torch.sign(u) * torch.log1p(-2 * u.abs())
        saved_lap = torch.distributions.laplace.Laplace.sample
        torch.distributions.laplace.Laplace.sample = fake_laplace_sample

        def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
            iterates.setdefault(int(iter), rec_img.clone())
            iterates["noisy"] = noisy_img.clone()
        noop = lambda *a, **k: None
        saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
                 utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips)
        torch.randn_like = fake_randn_like
        utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
        utils.compute_average_psnr = utils.compute_average_ssim = utils.compute_average_lpips = noop
        try:
            solver = pnp.PNP_FLOW(m, torch.device("cpu"), args)
            solver.solve_ip([(clean, torch.zeros(B))], degradation, sigma)
        finally:
            (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
             utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips) = saved
            torch.distributions.laplace.Laplace.sample = saved_lap
        assert seq["n"] == 1 + steps * num_samples
        rec = dict(steps=np.array(steps), num_samples=np.array(num_samples), alpha=np.array(alpha), sigma=np.array(sigma),
                   noisy=iterates["noisy"].numpy(), lr_pnp_after=np.array(args.lr_pnp))
        for it in (0, 1, 4, 9):
            rec[f"x_it{it}"] = iterates[it].numpy()
        np.savez_compressed(os.path.join(OUT, f"pnp_traj_{tag}.npz"), **rec)
        print("traj", tag, iterates[9].abs().mean().item())


def gen_ot_ode(models, degr, utils, only=None):
    """Real OT_ODE.solve_ip (pnpflow/methods/ot_ode.py) iterates + a stand-alone VJP."""
    import pnpflow.methods.ot_ode as ot
    # stand-alone J^T vec on the MNIST net and the tiny4 net
    for net, B in ((("mnist", 2), ("tiny4", 2)) if only is None else ()):
        m, cfg, sd = build_ref_unet(models, net)
        c = CFGS[net]; S = c["input_height"]
        x = det_normal((B, c["input_channels"], S, S), 51); vec = det_normal((B, c["input_channels"], S, S), 52)
        t = torch.tensor([0.25, 0.8][:B])
        g = torch.autograd.functional.vjp(lambda z: m(z, t), inputs=x, v=vec)[1]
        np.savez_compressed(os.path.join(OUT, f"vjp_{net}.npz"), t=t.numpy(), g=g.numpy())
        print("vjp", net, g.abs().mean().item())
    cases = [("tiny4_random_inpainting", "tiny4", "random_inpainting", lambda S: (degr.RandomInpainting(0.7), 0.01), 0.1, "constant"),
             ("tiny4_inpainting", "tiny4", "inpainting", lambda S: (degr.BoxInpainting(10), 0.05), 0.1, "gamma_t"),
             ("tiny4_superresolution", "tiny4", "superresolution", lambda S: (degr.Superresolution(2, S, device="cpu"), 0.05), 0.1, "constant"),
             ("mnist_denoising", "mnist", "denoising", lambda S: (degr.Denoising(), 0.2), 0.3, "gamma_t"),
             ("tiny4_gaussian_deblurring_FFT", "tiny4", "gaussian_deblurring_FFT",
              lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, device="cpu"), 0.05), 0.1, "constant"),
             # any other problem name takes the reference's generic branch: per-image GMRES on r_t^2 H H^T + sigma^2 I (ot_ode.py:118-128)
             ("tiny4_deblurring_gmres", "tiny4", "gaussian_deblurring",
              lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, device="cpu"), 0.2), 0.3, "constant")]
    if only is not None:
        cases = [c for c in cases if c[0] in only]
    steps, B = 10, 2
    for tag, net, problem, mk, t0, gamma in cases:
        m, cfg, sd = build_ref_unet(models, net)
        c = CFGS[net]; S = c["input_height"]
        degradation, sigma = mk(S)
        clean = det_image((B, c["input_channels"], S, S), 31)
        args = utils.CfgNode(dict(method="ot_ode", model="ot", dataset="celeba", problem=problem, steps_ode=steps, start_time=t0,
                                  gamma=gamma, max_batch=1, compute_time=False, compute_memory=False, save_results=True, batch=0,
                                  save_path_ip="/tmp"))
        iterates = {}; seq = {"n": 0}

        def fake_randn_like(like, **kw):
            i = seq["n"]; seq["n"] += 1
            return det_normal(tuple(like.shape), 61, i)        # call 0: measurement noise, call 1: initialisation noise

        def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
            iterates.setdefault(int(iter), rec_img.clone()); iterates["noisy"] = noisy_img.clone()
        noop = lambda *a, **k: None
        saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
                 utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips)
        torch.randn_like = fake_randn_like
        utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
        utils.compute_average_psnr = utils.compute_average_ssim = utils.compute_average_lpips = noop
        try:
            ot.OT_ODE(m, torch.device("cpu"), args).solve_ip([(clean, torch.zeros(B))], degradation, sigma)
        finally:
            (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
             utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips) = saved
        assert seq["n"] == 2
        first = int(steps * t0)
        rec = dict(steps=np.array(steps), start_time=np.array(t0), sigma=np.array(sigma), noisy=iterates["noisy"].numpy(),
                   first=np.array(first))
        for it in (first, first + 1, steps - 1):
            rec[f"x_it{it}"] = iterates[it].numpy()
        np.savez_compressed(os.path.join(OUT, f"ot_ode_traj_{tag}.npz"), **rec)
        print("ot_ode", tag, iterates[steps - 1].abs().mean().item())


def _run_pnp(models, degr, utils, pnp, net, problem, mk, alpha, B, steps, num_samples, noise_seed=41, stop_after=None, capture_inputs=False):
    """The real PNP_FLOW.solve_ip with torch.randn_like replaced by det_normal(noise_seed, call index).  Returns
    (iterates {logging iteration: x}, inputs-of-iteration list when capture_inputs, noisy measurement, args)."""
    m, cfg, sd = build_ref_unet(models, net)
    c = CFGS[net]; S = c["input_height"]
    degradation, sigma = mk(S)
    clean = det_image((B, c["input_channels"], S, S), 31)
    args = utils.CfgNode(dict(method="pnp_flow", model="ot", dataset="celeba", problem=problem, noise_type="gaussian", num_samples=num_samples,
                              steps_pnp=steps, lr_pnp=1.0, gamma_style="alpha_1_minus_t", alpha=alpha, max_batch=1, compute_time=False,
                              compute_memory=False, save_results=True, batch=0, save_path_ip="/tmp"))
    iterates, inputs, seq = {}, [], {"n": 0}

    class _Stop(Exception):
        pass

    def fake_randn_like(like, **kw):
        i = seq["n"]; seq["n"] += 1
        return det_normal(tuple(like.shape), noise_seed, i)   # call 0 = measurement noise, then (iteration, sample) order

    def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
        iterates.setdefault(int(iter), rec_img.clone()); iterates["noisy"] = noisy_img.clone()
    noop = lambda *a, **k: None
    saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
             utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips)
    torch.randn_like = fake_randn_like
    utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
    utils.compute_average_psnr = utils.compute_average_ssim = utils.compute_average_lpips = noop
    try:
        solver = pnp.PNP_FLOW(m, torch.device("cpu"), args)
        if capture_inputs or stop_after is not None:
            orig = solver.grad_datafit

            def grad_hook(x, y, H, H_adj):       # called once at the top of every outer iteration with that iteration's input
                inputs.append(x.clone())
                if stop_after is not None and len(inputs) > stop_after:
                    raise _Stop()
                return orig(x, y, H, H_adj)
            solver.grad_datafit = grad_hook
        try:
            solver.solve_ip([(clean, torch.zeros(B))], degradation, sigma)
        except _Stop:
            pass
    finally:
        (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
         utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips) = saved
    return iterates, inputs, clean, sigma, args, seq["n"]


def gen_long(models, degr, utils, pnp):
    """Full-length recursions of the real solvers:
      pnp_long_tiny4_*      100 x 5 PnP-Flow on the 4-level test net, B = 2 (the shipped loop length: pnp_flow.py:103-121)
      pnp_long_mnist_c1     BASELINE configs[0]: MNIST-shaped denoising, B = 8, 50 x 5, iterates 0, 5, ..., 45, 49
      ot_ode_long_tiny4_*   90 Euler steps (steps_ode 100, start_time 0.1) of the real OT_ODE
      pnp_iter_{c2,c3,c4}   the first TWO outer iterations of BASELINE configs[1..3] on their own nets (34.5 M / 31.0 M parameters) and
                            operator parameters, B = 2, 100 x 5 schedule (crops + float64 checksums)"""
    import pnpflow.methods.ot_ode as ot
    cases = [("tiny4_inpainting", "tiny4", "inpainting", lambda S: (degr.BoxInpainting(10), 0.05), 0.5),
             ("tiny4_superresolution", "tiny4", "superresolution", lambda S: (degr.Superresolution(2, S, device="cpu"), 0.05), 0.3),
             ("tiny4_deblurring", "tiny4", "gaussian_deblurring_FFT", lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, "cpu"), 0.05), 0.01)]
    for tag, net, problem, mk, alpha in cases:
        its, _, clean, sigma, args, ncalls = _run_pnp(models, degr, utils, pnp, net, problem, mk, alpha, 2, 100, 5)
        assert ncalls == 1 + 500 and 99 in its
        rec = dict(steps=np.array(100), num_samples=np.array(5), alpha=np.array(alpha), sigma=np.array(sigma), noisy=its["noisy"].numpy(),
                   logged=np.array(sorted(k for k in its if k != "noisy")))
        for it in (0, 10, 50, 90, 99):
            rec[f"x_it{it}"] = its[it].numpy()
        np.savez_compressed(os.path.join(OUT, f"pnp_long_{tag}.npz"), **rec)
        print("long", tag, float(its[99].abs().mean()))
    # C1
    its, _, clean, sigma, args, ncalls = _run_pnp(models, degr, utils, pnp, "mnist", "denoising", lambda S: (degr.Denoising(), 0.2), 0.8, 8, 50, 5)
    assert ncalls == 1 + 250
    rec = dict(steps=np.array(50), num_samples=np.array(5), alpha=np.array(0.8), sigma=np.array(sigma), noisy=its["noisy"].numpy())
    for it in list(range(0, 50, 5)) + [49]:
        rec[f"x_it{it}"] = its[it].numpy()
    np.savez_compressed(os.path.join(OUT, "pnp_long_mnist_c1.npz"), **rec)
    print("long c1", float(its[49].abs().mean()))
    # first two outer iterations of C2 / C3 / C4 on the BASELINE nets
    big = [("c2", "celeba128", "inpainting", lambda S: (degr.BoxInpainting(20), 0.05), 0.5),
           ("c3", "celeba128", "gaussian_deblurring_FFT", lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, "cpu"), 0.05), 0.01),
           ("c4", "afhq256", "superresolution", lambda S: (degr.Superresolution(4, S, device="cpu"), 0.05), 0.3)]
    for tag, net, problem, mk, alpha in big:
        its, inputs, clean, sigma, args, ncalls = _run_pnp(models, degr, utils, pnp, net, problem, mk, alpha, 2, 100, 5, stop_after=2)
        assert len(inputs) == 3 and ncalls == 1 + 10        # inputs[k] = x entering iteration k = the result of iteration k - 1
        rec = dict(steps=np.array(100), num_samples=np.array(5), alpha=np.array(alpha), sigma=np.array(sigma), B=np.array(2))
        rec.update(crop_rec("noisy", its["noisy"])); rec.update(crop_rec("x_it0", inputs[1])); rec.update(crop_rec("x_it1", inputs[2]))
        np.savez_compressed(os.path.join(OUT, f"pnp_iter_{tag}.npz"), **rec)
        print("iter", tag, float(inputs[2].abs().mean()))
    # 90-step OT-ODE on the test net
    for tag, problem, mk, gamma in (("tiny4_random_inpainting", "random_inpainting", lambda S: (degr.RandomInpainting(0.7), 0.01), "constant"),
                                    ("tiny4_superresolution", "superresolution", lambda S: (degr.Superresolution(2, S, device="cpu"), 0.05), "constant")):
        m, cfg, sd = build_ref_unet(models, "tiny4")
        S, B, steps, t0 = 64, 2, 100, 0.1
        degradation, sigma = mk(S)
        clean = det_image((B, 3, S, S), 31)
        args = utils.CfgNode(dict(method="ot_ode", model="ot", dataset="celeba", problem=problem, steps_ode=steps, start_time=t0, gamma=gamma,
                                  max_batch=1, compute_time=False, compute_memory=False, save_results=True, batch=0, save_path_ip="/tmp"))
        iterates = {}; seq = {"n": 0}

        def fake_randn_like(like, **kw):
            i = seq["n"]; seq["n"] += 1
            return det_normal(tuple(like.shape), 61, i)        # call 0: measurement noise, call 1: initialisation noise

        def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
            iterates.setdefault(int(iter), rec_img.clone()); iterates["noisy"] = noisy_img.clone()
        noop = lambda *a, **k: None
        saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
                 utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips)
        torch.randn_like = fake_randn_like
        utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
        utils.compute_average_psnr = utils.compute_average_ssim = utils.compute_average_lpips = noop
        try:
            ot.OT_ODE(m, torch.device("cpu"), args).solve_ip([(clean, torch.zeros(B))], degradation, sigma)
        finally:
            (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
             utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips) = saved
        assert seq["n"] == 2 and 99 in iterates
        rec = dict(steps=np.array(steps), start_time=np.array(t0), sigma=np.array(sigma), noisy=iterates["noisy"].numpy(), first=np.array(10))
        for it in (10, 50, 99):
            rec[f"x_it{it}"] = iterates[it].numpy()
        np.savez_compressed(os.path.join(OUT, f"ot_ode_long_{tag}.npz"), **rec)
        print("ot_ode long", tag, float(iterates[99].abs().mean()))


def gen_biglong(models, degr, utils, pnp, only=None):
    """Full-length recursions of the real solvers on the BASELINE nets (pnp_flow.py:103-121, ot_ode.py:63-147), B = 1:
      pnp_biglong_c2      celeba128 net, BoxInpainting(20), sigma 0.05, alpha 0.5, 100 x 5          (BASELINE configs[1])
      pnp_biglong_c3      celeba128 net, Gaussian blur sigma 1 (61 taps), sigma 0.05, alpha 0.01    (configs[2])
      pnp_biglong_c4      afhq256 net, superresolution x4, sigma 0.05, alpha 0.3                     (configs[3])
      pnp_biglong_c2_256  afhq256 net, BoxInpainting(40) - the headline workload of bench.py (main.py:132-136 at 256^2)
      ot_ode_biglong_c5   afhq256 net, RandomInpainting(0.7), sigma 0.01, steps_ode 100, start_time 0.1: all 90 Euler steps (configs[4])
    Stored: crops + float64 checksums of the logged iterates, the final iterate in full, the measurement in full."""
    import pnpflow.methods.ot_ode as ot
    big = [("c2", "celeba128", "inpainting", lambda S: (degr.BoxInpainting(20), 0.05), 0.5),
           ("c3", "celeba128", "gaussian_deblurring_FFT", lambda S: (degr.GaussianDeblurring(1.0, 61, "fft", 3, S, "cpu"), 0.05), 0.01),
           ("c4", "afhq256", "superresolution", lambda S: (degr.Superresolution(4, S, device="cpu"), 0.05), 0.3),
           ("c2_256", "afhq256", "inpainting", lambda S: (degr.BoxInpainting(40), 0.05), 0.5)]
    for tag, net, problem, mk, alpha in big:
        if only is not None and tag not in only:
            continue
        its, _, clean, sigma, args, ncalls = _run_pnp(models, degr, utils, pnp, net, problem, mk, alpha, 1, 100, 5)
        assert ncalls == 1 + 500 and 99 in its
        rec = dict(steps=np.array(100), num_samples=np.array(5), alpha=np.array(alpha), sigma=np.array(sigma), B=np.array(1),
                   noisy=its["noisy"].numpy(), x_final=its[99].numpy(), psnr_final=O.psnr_per_image(its[99], clean).numpy())
        for it in (0, 10, 50, 99):
            rec.update(crop_rec(f"x_it{it}", its[it]))
        np.savez_compressed(os.path.join(OUT, f"pnp_biglong_{tag}.npz"), **rec)
        print("biglong", tag, float(its[99].abs().mean()), rec["psnr_final"], flush=True)
    if only is None or "c5" in only:
        m, cfg, sd = build_ref_unet(models, "afhq256")
        S, B, sigma = 256, 1, 0.01
        degradation = degr.RandomInpainting(0.7)
        clean = det_image((B, 3, S, S), 31)
        args = utils.CfgNode(dict(method="ot_ode", model="ot", dataset="afhq_cat", problem="random_inpainting", steps_ode=100, start_time=0.1,
                                  gamma="constant", max_batch=1, compute_time=False, compute_memory=False, save_results=True, batch=0,
                                  save_path_ip="/tmp"))
        iterates = {}; seq = {"n": 0}

        def fake_randn_like(like, **kw):
            i = seq["n"]; seq["n"] += 1
            return det_normal(tuple(like.shape), 61, i)        # call 0: measurement noise, call 1: initialisation noise

        def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
            iterates.setdefault(int(iter), rec_img.clone()); iterates["noisy"] = noisy_img.clone()
        noop = lambda *a, **k: None
        saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
                 utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips)
        torch.randn_like = fake_randn_like
        utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
        utils.compute_average_psnr = utils.compute_average_ssim = utils.compute_average_lpips = noop
        try:
            ot.OT_ODE(m, torch.device("cpu"), args).solve_ip([(clean, torch.zeros(B))], degradation, sigma)
        finally:
            (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images,
             utils.compute_average_psnr, utils.compute_average_ssim, utils.compute_average_lpips) = saved
        assert seq["n"] == 2 and 99 in iterates
        rec = dict(steps=np.array(100), start_time=np.array(0.1), sigma=np.array(sigma), B=np.array(B), first=np.array(10),
                   noisy=iterates["noisy"].numpy(), x_final=iterates[99].numpy(), psnr_final=O.psnr_per_image(iterates[99], clean).numpy())
        for it in (10, 50, 99):
            rec.update(crop_rec(f"x_it{it}", iterates[it]))
        np.savez_compressed(os.path.join(OUT, "ot_ode_biglong_c5.npz"), **rec)
        print("biglong c5", float(iterates[99].abs().mean()), rec["psnr_final"], flush=True)


def crop_rec(prefix, t):
    """crop + corner + float64 checksums of a (B,C,H,W) tensor (the big-net fixtures stay small)."""
    H = t.shape[2]
    return {prefix + "_crop": t[:, :, H // 2 - 16:H // 2 + 16, H // 2 - 16:H // 2 + 16].numpy().copy(),
            prefix + "_corner": t[:, :, :8, :8].numpy().copy(), prefix + "_checksum": checksums(t)}


def gen_big(models, degr, utils):
    """BASELINE-size nets (C4/C5): the reference's autograd VJP on the 34.5 M / 31.0 M parameter nets
    (ot_ode.py:137-138) and ONE real OT_ODE.solve_ip Euler step at 256^2 with random inpainting (config C5: steps_ode=100,
    start_time=0.1, gamma constant -> iteration 10, t = 0.1)."""
    import pnpflow.methods.ot_ode as ot
    for net in ("celeba128", "afhq256"):
        m, cfg, sd = build_ref_unet(models, net)
        c = CFGS[net]; S = c["input_height"]
        x = det_normal((1, 3, S, S), 51); vec = det_normal((1, 3, S, S), 52)
        t = torch.tensor([0.25])
        g = torch.autograd.functional.vjp(lambda z: m(z, t), inputs=x, v=vec)[1]
        rec = dict(t=t.numpy(), g_absmax=np.array(float(g.abs().max())))
        rec.update(crop_rec("g", g))
        np.savez_compressed(os.path.join(OUT, f"vjp_{net}.npz"), **rec)
        print("vjp", net, g.abs().mean().item())
    # one C5-shaped Euler step of the real solver
    m, cfg, sd = build_ref_unet(models, "afhq256")
    S, B, sigma = 256, 2, 0.01
    degradation = degr.RandomInpainting(0.7)
    clean = det_image((B, 3, S, S), 31)
    args = utils.CfgNode(dict(method="ot_ode", model="ot", dataset="afhq_cat", problem="random_inpainting", steps_ode=100, start_time=0.1,
                              gamma="constant", max_batch=1, compute_time=False, compute_memory=False, save_results=True, batch=0,
                              save_path_ip="/tmp"))
    iterates = {}; seq = {"n": 0}

    class _Stop(Exception):
        pass

    def fake_randn_like(like, **kw):
        i = seq["n"]; seq["n"] += 1
        return det_normal(tuple(like.shape), 61, i)        # call 0: measurement noise, call 1: initialisation noise

    def cap_psnr(clean_img, noisy_img, rec_img, a, H_adj, iter="final"):
        iterates[int(iter)] = rec_img.clone(); iterates["noisy"] = noisy_img.clone()
        raise _Stop()                                       # iteration 10 is the first one and a logging iteration: stop after it
    noop = lambda *a, **k: None
    saved = (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images)
    torch.randn_like = fake_randn_like
    utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images = cap_psnr, noop, noop, noop
    try:
        ot.OT_ODE(m, torch.device("cpu"), args).solve_ip([(clean, torch.zeros(B))], degradation, sigma)
    except _Stop:
        pass
    finally:
        (torch.randn_like, utils.compute_psnr, utils.compute_ssim, utils.compute_lpips, utils.save_images) = saved
    assert seq["n"] == 2 and 10 in iterates
    rec = dict(steps=np.array(100), start_time=np.array(0.1), sigma=np.array(sigma), first=np.array(10), B=np.array(B))
    rec.update(crop_rec("noisy", iterates["noisy"])); rec.update(crop_rec("x_it10", iterates[10]))
    np.savez_compressed(os.path.join(OUT, "ot_ode_step_afhq256_random_inpainting.npz"), **rec)
    print("ot_ode step afhq256", iterates[10].abs().mean().item())


def gen_ops():
    """The reference's native ops through their own pure-torch definitions: op/upfirdn2d.py `upfirdn2d_native` (:142-187) and the
    CPU branch of op/fused_act.py `fused_leaky_relu` (:84-96), imported with torch.utils.cpp_extension.load stubbed out (the
    modules JIT-compile their CUDA sources at import, which cannot happen here), plus upsample_2d / downsample_2d of
    models/up_or_down_sampling.py, which call them."""
    import importlib
    import types
    import torch.utils.cpp_extension as ce
    saved = ce.load
    ce.load = lambda *a, **k: types.SimpleNamespace()
    try:
        from ref_import import install_stubs, REF
        install_stubs()
        if REF not in sys.path:
            sys.path.insert(0, REF)
        up = importlib.import_module("pnpflow.image_generation.op.upfirdn2d")
        fa = importlib.import_module("pnpflow.image_generation.op.fused_act")
        uds = importlib.import_module("pnpflow.image_generation.models.up_or_down_sampling")
    finally:
        ce.load = saved
    rec = {}
    x = det_normal((2, 3, 20, 24), 71)
    cases = [("fir4_up2", np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 16.0, 2, 2, 1, 1, 2, 1, 2, 1),
             ("fir4_down2", np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, 1, 1, 2, 2, 1, 1, 1, 1),
             ("asym3x2_up3_down2_crop", np.array([[1.0, -2.0], [0.5, 3.0], [4.0, -1.5]]), 3, 2, 2, 3, -1, 2, 3, -2),
             ("k1_identity", np.array([[1.0]]), 1, 1, 1, 1, 0, 0, 0, 0),
             ("k5_pad", np.arange(25, dtype=np.float64).reshape(5, 5) / 25.0, 1, 1, 1, 1, 2, 2, 2, 2)]
    for name, k, ux, uy, dx, dy, px0, px1, py0, py1 in cases:
        kt = torch.from_numpy(np.asarray(k, dtype=np.float32))
        rec[name + "_k"] = kt.numpy(); rec[name + "_p"] = np.array([ux, uy, dx, dy, px0, px1, py0, py1])
        rec[name + "_out"] = up.upfirdn2d_native(x, kt, ux, uy, dx, dy, px0, px1, py0, py1).numpy()
    x2 = det_normal((2, 4, 16, 16), 72)
    rec["upsample_2d_1331"] = uds.upsample_2d(x2, (1, 3, 3, 1), factor=2).numpy()
    rec["downsample_2d_1331"] = uds.downsample_2d(x2, (1, 3, 3, 1), factor=2).numpy()
    rec["upsample_2d_default"] = uds.upsample_2d(x2, factor=2).numpy()
    rec["downsample_2d_default"] = uds.downsample_2d(x2, factor=2).numpy()
    xb = det_normal((2, 5, 6, 7), 73); bias = det_normal((5,), 74)
    rec["fused_leaky_relu"] = fa.fused_leaky_relu(xb, bias, 0.2, 2 ** 0.5).numpy()
    rec["fused_leaky_relu_2d"] = fa.fused_leaky_relu(det_normal((3, 5), 75), bias, 0.2, 2 ** 0.5).numpy()
    np.savez_compressed(os.path.join(OUT, "native_ops.npz"), **rec)
    print("native ops ok")


NCSNPP_CFGS = {
    # 3 levels 32/16/8: attention on the 16^2 level (down: per block, up: once), 1x1 shortcuts, down/up FIR blocks, both pyramids
    "tiny": dict(image_size=32, nf=32, ch_mult=(1, 1, 2), num_res_blocks=2, attn_resolutions=(16,)),
    # the widths of the real net (128/256 channels -> the fused attention kernel) on a 2-level 32^2 net
    "wide": dict(image_size=32, nf=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(16,)),
    # the reference's rectified-flow config (configs/rectified_flow/afhq_cat_pytorch_rf_gaussian.py): 65.6 M parameters
    "afhq256": dict(image_size=256, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,)),
}


def gen_ncsnpp():
    """NCSNpp.forward of the REAL reference module (pnpflow/image_generation/models/ncsnpp.py) on the oracle's synthetic weights.
    The reference's op/ modules JIT-compile CUDA sources at import: torch.utils.cpp_extension.load is stubbed for the import,
    and on CPU tensors the reference itself takes its pure-torch `upfirdn2d_native` branch (op/upfirdn2d.py:128-139).  The
    ml_collections config object is replaced by a plain attribute namespace with the fields NCSNpp.__init__ reads."""
    import importlib
    import types
    import torch.utils.cpp_extension as ce
    from oracle import ncsnpp_oracle as NO
    saved = ce.load
    ce.load = lambda *a, **k: types.SimpleNamespace()
    try:
        from ref_import import install_stubs, REF
        install_stubs()
        if REF not in sys.path:
            sys.path.insert(0, REF)
        nc = importlib.import_module("pnpflow.image_generation.models.ncsnpp")
    finally:
        ce.load = saved
    NS = types.SimpleNamespace

    def ref_config(c):      # default_lsun_configs.py:52-70 + celeba_hq_pytorch_rf_gaussian.py:43-64
        return NS(model=NS(nf=c["nf"], ch_mult=c["ch_mult"], num_res_blocks=c["num_res_blocks"], attn_resolutions=c["attn_resolutions"],
                           dropout=0., resamp_with_conv=True, conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True,
                           resblock_type='biggan', progressive='output_skip', progressive_input='input_skip', progressive_combine='sum',
                           attention_type='ddpm', embedding_type='fourier', init_scale=0., fourier_scale=16, conv_size=3,
                           nonlinearity='swish', scale_by_sigma=True, sigma_max=378, sigma_min=0.01, num_scales=2000, name='ncsnpp'),
                  data=NS(image_size=c["image_size"], num_channels=3, centered=True),
                  training=NS(continuous=False, sde='rectified_flow'))

    for name, c in NCSNPP_CFGS.items():
        cfg = NO.ncsnpp_config(**c)
        sd = NO.synthetic_state_dict(cfg, 0)
        m = nc.NCSNpp(ref_config(c)).eval()
        own = m.state_dict()
        assert set(own) - {"sigmas"} == set(sd), (set(own) ^ set(sd))
        for k, v in sd.items():
            assert tuple(own[k].shape) == tuple(v.shape), k
        m.load_state_dict(sd, strict=False)
        S = c["image_size"]
        B = 1 if name == "afhq256" else 2
        x = det_normal((B, 3, S, S), 81)
        t = torch.tensor([0.37, 0.81][:B])
        feats = {}
        hooks = []
        if name == "tiny":        # a few intermediate activations: first conv, first down block, the mid block's last module input
            hooks.append(m.all_modules[3].register_forward_hook(lambda mod, i, o: feats.__setitem__("conv_in", o.detach())))
            hooks.append(m.all_modules[4].register_forward_hook(lambda mod, i, o: feats.__setitem__("down0_0", o.detach())))
        with torch.no_grad():
            y = m(x, t * 999)
        for h in hooks:
            h.remove()
        rec = dict(t=t.numpy(), y_absmax=np.array(float(y.abs().max())))
        if name != "wide":       # the VJP OT_ODE takes (ot_ode.py:137-138), by autograd through the real module
            vec = det_normal((B, 3, S, S), 87)
            for prm in m.parameters():
                prm.requires_grad_(False)
            gv = torch.autograd.functional.vjp(lambda z: m(z, t * 999), x, vec)[1]
            rec["g_absmax"] = np.array(float(gv.abs().max()))
            if name == "afhq256":
                rec.update(crop_rec("g", gv))
            else:
                rec["g"] = gv.numpy()
            print("   vjp", float(gv.abs().mean()), float(gv.abs().max()))
        if name == "afhq256":
            rec.update(crop_rec("y", y))
        else:
            rec["y"] = y.numpy()
            for k, v in feats.items():
                rec["tap_" + k] = v.numpy()
        np.savez_compressed(os.path.join(OUT, f"ncsnpp_{name}.npz"), **rec)
        print("ncsnpp", name, float(y.abs().mean()), float(y.abs().max()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    models, degr, utils, pnp = import_reference()
    which = sys.argv[1:] or ["unet", "degr", "traj", "ot_ode", "big", "ops", "ncsnpp", "long"]
    if "unet" in which:
        gen_unet(models)
    if "degr" in which:
        gen_degradations(degr, utils)
    if "traj" in which:
        gen_traj(models, degr, utils, pnp)
    if "ot_ode" in which:
        gen_ot_ode(models, degr, utils)
    if "big" in which:
        gen_big(models, degr, utils)
    if "ops" in which:
        gen_ops()
    if "ncsnpp" in which:
        gen_ncsnpp()
    if "long" in which:
        gen_long(models, degr, utils, pnp)
    if "biglong" in which:
        gen_biglong(models, degr, utils, pnp, only=[w[8:] for w in which if w.startswith("biglong:")] or None)
    if "gmres" in which:
        gen_ot_ode(models, degr, utils, only=("tiny4_deblurring_gmres",))
