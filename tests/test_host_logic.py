"""CPU tests of the host side: the C-ABI library loads and exports every symbol declared in
include/pnpflow_hip.h (no compute calls), the config surface, the schedule scalars."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import pnpflow_amd._lib as L
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "pnpflow_hip.h")).read()
    declared = set(re.findall(r"\b(pf_[a-z_A-Z0-9]+)\s*\(", hdr))
    declared -= {"pf_iter_callback"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert set(L.SIGNATURES) <= declared
    assert lib.pf_abi_version() == L.PF_ABI_VERSION


def test_engine_create_fails_loudly_without_gpu_or_bad_cfg():
    import pnpflow_amd._lib as L
    from pnpflow_amd.models import UNet
    with pytest.raises(L.PnpFlowHipError):
        UNet(3, 100, 32, ch_mult=(1, 2, 4, 8), num_res_blocks=1, attn_resolutions=())   # invalid height on any box
    if not torch.cuda.is_available():
        with pytest.raises(L.PnpFlowHipError):
            UNet(3, 64, 32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=())       # no device: no silent fallback


def test_config_surface(tmp_path):
    from pnpflow_amd.utils import CfgNode, get_save_path_ip, load_cfg_from_cfg_file, merge_cfg_from_list
    cfg = load_cfg_from_cfg_file(os.path.join(ROOT, "config", "main_config.yaml"))
    assert cfg.method == "pnp_flow" and cfg.batch_size_ip == 4 and cfg.save_results is True
    cfg = merge_cfg_from_list(cfg, ["dataset", "celeba", "max_batch", "25", "new_key", "[1,2]", "noise_type", "gaussian"])
    assert cfg.dataset == "celeba" and cfg.max_batch == 25 and cfg.new_key == [1, 2]
    with pytest.raises(ValueError):
        merge_cfg_from_list(cfg, ["max_batch", "abc"])          # type mismatch is an error, as in the reference
    cfg.update(load_cfg_from_cfg_file(os.path.join(ROOT, "config", "dataset_config", "celeba.yaml")))
    m = load_cfg_from_cfg_file(os.path.join(ROOT, "config", "method_config", "pnp_flow.yaml"))
    assert cfg.dim_image == 128 and list(m.keys()) == ["steps_pnp", "lr_pnp", "gamma_style", "num_samples", "alpha"]
    assert get_save_path_ip(dict(a=1, b="x")) == "a=1/b=x"
    c = CfgNode(dict(a=1)); c.b = 2
    assert c["b"] == 2 and c.a == 1


def test_main_parse_args_and_problem_table(monkeypatch):
    import sys
    monkeypatch.chdir(ROOT)
    monkeypatch.setattr(sys, "argv", ["main.py", "--opts", "dataset", "celeba", "problem", "inpainting", "alpha", "0.5", "steps_pnp", "20"])
    import main as M
    cfg = M.parse_args()
    assert cfg.dim_image == 128 and cfg.alpha == 0.5 and cfg.steps_pnp == 20
    assert cfg.dict_cfg_method == dict(steps_pnp=20, lr_pnp=1.0, gamma_style="alpha_1_minus_t", num_samples=5, alpha=0.5)
    for problem, dim, kind, sig in (("denoising", 128, "Denoising", 0.2), ("inpainting", 128, "BoxInpainting", 0.05),
                                    ("inpainting", 256, "BoxInpainting", 0.05), ("random_inpainting", 256, "RandomInpainting", 0.01),
                                    ("superresolution", 256, "Superresolution", 0.05), ("gaussian_deblurring_FFT", 128, "GaussianDeblurring", 0.05)):
        d, s = M.make_degradation(problem, dim, 3, "gaussian", "cpu")
        assert type(d).__name__ == kind and s == sig
    d, _ = M.make_degradation("inpainting", 256, 3, "gaussian", "cpu"); assert d.half_size_mask == 40
    d, _ = M.make_degradation("superresolution", 128, 3, "gaussian", "cpu"); assert d.sf == 2
    d, _ = M.make_degradation("gaussian_deblurring_FFT", 256, 3, "gaussian", "cpu"); assert d.sigma == 3.0 and d.kernel_size == 61
    with pytest.raises(ValueError):
        M.make_degradation("nope", 128, 3, "gaussian", "cpu")


def test_schedule_scalars_match_reference_expressions():
    """t and lr_t are computed on the host with the reference's own fp32 expressions
    (pnp_flow.py:107-109, 29-37)."""
    from oracle import pnpflow_oracle as O
    from pnpflow_amd.methods.pnp_flow import PNP_FLOW
    from pnpflow_amd.utils import CfgNode

    class Dummy:
        def to(self, d):
            return self
    args = CfgNode(dict(method="pnp_flow", model="ot", gamma_style="alpha_1_minus_t", alpha=0.3, noise_type="gaussian"))
    s = PNP_FLOW.__new__(PNP_FLOW); s.args = args
    sigma, steps = 0.05, 100
    t_vals, coef = s._schedule(steps, sigma ** 2 * 1.0, sigma)
    for it in (0, 1, 37, 99):
        t1 = torch.ones(1) * (1 / steps) * it
        assert t_vals[it] == float(t1[0])
        lr_t = O.learning_rate_strat(sigma ** 2, t1, "alpha_1_minus_t", 0.3)
        assert abs(coef[it] - float(lr_t.reshape(-1)[0]) / sigma ** 2) < 1e-6
    assert coef[0] == pytest.approx(1.0, abs=1e-6)
    taps = None
    import pnpflow_amd.degradations as D
    g = D.GaussianDeblurring(3.0, 61, "fft", 3, 256, "cpu")
    np.testing.assert_allclose(g.taps_host, O.gaussian_1d_taps(3.0, 61).astype(np.float32), rtol=1e-6)
    m = D.RandomInpainting(0.7, global_batch=8, batch_offset=4).mask(4, 16, 16, "cpu").numpy()
    np.testing.assert_array_equal(m, O.random_mask_array(8, 16, 16, 0.7)[4:8].astype(np.uint8))


def test_reference_import_paths_resolve_to_the_engine():
    """Drop-in: the reference's import lines work unchanged and land on pnpflow_amd."""
    from pnpflow.methods.pnp_flow import PNP_FLOW
    from pnpflow.methods.ot_ode import OT_ODE
    from pnpflow.degradations import BoxInpainting, GaussianDeblurring, Superresolution, RandomInpainting, Denoising
    from pnpflow.utils import load_cfg_from_cfg_file, merge_cfg_from_list, define_model, load_model
    import pnpflow_amd.methods.pnp_flow as A
    assert PNP_FLOW is A.PNP_FLOW and OT_ODE.__module__ == "pnpflow_amd.methods.ot_ode"
    assert BoxInpainting(20).half_size_mask == 20 and Superresolution(4, 256).sf == 4


def test_bicubic_taps_are_the_separable_factor_of_the_reference_filter():
    """Superresolution(mode="bicubic") on the engine uses the 1-D factor of utils.py:365-396's normalised outer
    product; tap K/2 sits at offset 0 after the reference's roll by (-(K-1))//2."""
    from oracle import pnpflow_oracle as O
    import pnpflow_amd.degradations as D
    for sf in (2, 4):
        w = D.bicubic_taps(sf)
        k = O.bicubic_filter(sf)[0, 0].numpy()
        assert w.shape == (4 * sf,) and w.dtype == np.float32
        np.testing.assert_allclose(np.outer(w, w), k, atol=1e-7)
        assert abs(float(w.sum()) - 1.0) < 1e-6
        d = D.Superresolution(sf, 64, mode="bicubic", device="cpu")
        assert d.kind == 5 and d.taps_host.shape[0] == 4 * sf
        # roll offset: the filter tap that lands on pixel (0, 0) is k[K/2, K/2]
        assert O.Superresolution(sf, 64, mode="bicubic").filter[0, 0, 0, 0] == k[2 * sf, 2 * sf]
    with pytest.raises(NotImplementedError):
        D.Superresolution(2, 64, mode="nearest")


def test_ot_ode_scalars_match_reference_expressions():
    """OT_ODE._scalars reproduces ot_ode.py:69-73, 96 (the `delta * iteration**2` quirk of the superresolution branch)
    and :133-143 in fp32, per problem."""
    from pnpflow_amd.methods.ot_ode import OT_ODE
    from pnpflow_amd.utils import CfgNode
    for gamma in ("constant", "gamma_t"):
        s = OT_ODE.__new__(OT_ODE); s.args = CfgNode(dict(gamma=gamma))
        steps, delta, B = 100, 1 / 100, 3
        for it in (10, 37, 99):
            for problem in ("inpainting", "denoising", "gaussian_deblurring_FFT", "superresolution"):
                t1, omt, rt2, coef = s._scalars(it, delta, problem, B, "cpu")
                t_ref = torch.ones(B) * delta * it
                assert torch.equal(t1, t_ref) and torch.equal(omt, 1 - t_ref)
                if problem == "superresolution":
                    r = torch.tensor((1 - delta * it) ** 2 / ((1 - delta * it) ** 2 + delta * it ** 2)).float()
                    assert torch.allclose(rt2, r.expand(B))
                else:
                    assert torch.equal(rt2, ((1 - t_ref) ** 2 / ((1 - t_ref) ** 2 + t_ref ** 2)))
                g = torch.ones(B) if gamma == "constant" else torch.sqrt(t_ref / (t_ref ** 2 + (1 - t_ref) ** 2))
                assert torch.allclose(coef, (1 - t_ref) / t_ref * g)


def test_bench_helpers():
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    a = bench.det_image((2, 3, 16, 16), 5)
    assert a.shape == (2, 3, 16, 16) and float(a.min()) >= -1.0 and float(a.max()) <= 1.0
    assert torch.equal(a, bench.det_image((2, 3, 16, 16), 5))


# ---- round 2: data readers, shard draws, operator attributes (no GPU) ---------------------------------------------------
def _write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def test_celeba_reader_matches_the_reference_pipeline(tmp_path):
    """pnpflow/dataloaders.py:22-31,121-153: partition CSV (read with the reference's own pandas call, which drops the
    first listed image), CenterCrop(178) -> Resize(128) bilinear -> ToTensor -> Normalize(.5,.5); missing files filtered."""
    from PIL import Image
    from pnpflow_amd.dataloaders import DataLoaders
    rng = np.random.RandomState(0)
    d = tmp_path / "data" / "celeba" / "img_align_celeba"
    d.mkdir(parents=True)
    names = [f"{i:06d}.png" for i in range(1, 8)]
    imgs = {}
    for nme in names[:-1]:                      # the last listed file is missing on disk
        a = rng.randint(0, 256, size=(218, 178, 3)).astype(np.uint8)
        _write_png(str(d / nme), a); imgs[nme] = a
    with open(tmp_path / "data" / "celeba" / "list_eval_partition.csv", "w") as f:
        f.write("image_id,partition\n")
        for i, nme in enumerate(names):
            f.write(f"{nme},{2 if i >= 2 else 0}\n")
    loaders = DataLoaders("celeba", 3, 3, root=str(tmp_path) + "/").load_data()
    batches = list(loaders["test"])
    # partition 2 = names[2:], the last is missing -> 4 images in batches of 3 + 1
    assert [b[0].shape[0] for b in batches] == [3, 1]
    x0 = batches[0][0][0]
    assert x0.shape == (3, 128, 128) and x0.dtype == torch.float32 and -1.0 <= float(x0.min()) and float(x0.max()) <= 1.0
    ref = Image.fromarray(imgs[names[2]][20:198]).resize((128, 128), Image.BILINEAR)       # rows 20..197 = the centre crop of 218
    ref = (torch.from_numpy(np.asarray(ref).transpose(2, 0, 1).copy()).float() / 255 - 0.5) / 0.5
    assert torch.equal(x0, ref)
    # the reference's read_csv(header=0, names=..., skiprows=1) consumes the first data row as a header: partition 0 has names[1] only
    assert len(loaders["train"].dataset) == 1


def test_afhq_reader_and_empty_collate(tmp_path):
    from PIL import Image
    from pnpflow_amd.dataloaders import DataLoaders, custom_collate
    rng = np.random.RandomState(1)
    d = tmp_path / "data" / "afhq_cat" / "test" / "cat"
    d.mkdir(parents=True)
    arrs = []
    for i in (2, 0, 1):                                     # written out of order: the reader sorts
        a = rng.randint(0, 256, size=(300, 280, 3)).astype(np.uint8)
        _write_png(str(d / f"cat_{i}.png"), a); arrs.append((i, a))
    dl = DataLoaders("afhq_cat", 2, 2, root=str(tmp_path) + "/")
    assert dl.available("test") and not dl.available("val")
    batches = list(dl.load_data()["test"])
    assert [b[0].shape for b in batches] == [(2, 3, 256, 256), (1, 3, 256, 256)]
    a0 = dict(arrs)[0]
    ref = (torch.from_numpy(np.asarray(Image.fromarray(a0).resize((256, 256), Image.BILINEAR)).transpose(2, 0, 1).copy()).float() / 255 - 0.5) / 0.5
    assert torch.equal(batches[0][0][0], ref)
    e = custom_collate([(None, None)])
    assert e[0].numel() == 0 and e[1].numel() == 0


def test_engine_normal_offset_is_a_slice_of_the_global_stream():
    from oracle import pnpflow_oracle as O
    full = O.engine_normal(4001, 77, 5)
    for off, n in ((0, 13), (4, 100), (1001, 999), (3, 4), (3998, 3)):
        np.testing.assert_array_equal(O.engine_normal(n, 77, 5, offset=off), full[off:off + n])


def test_bench_shards_reassemble_the_global_batch():
    """bench.py's per-rank inputs are slices of ONE global draw: N ranks restore the images a single-device run restores."""
    import bench
    wl = dict(bench.WORKLOADS["tiny"]); wl["B"] = 2
    one = bench.shard_inputs(dict(wl, B=4), 0, 1)
    two = [bench.shard_inputs(wl, r, 2) for r in range(2)]
    assert (two[0][0], two[0][1], two[1][0], two[1][1]) == (0, 2, 2, 4)
    assert torch.equal(torch.cat([t[2] for t in two]), one[2]) and torch.equal(torch.cat([t[3] for t in two]), one[3])
    wo = dict(bench.WORKLOADS["c5"], dim=64, B=1)
    a = bench.shard_inputs(dict(wo, B=2), 0, 1); b = [bench.shard_inputs(wo, r, 2) for r in range(2)]
    assert torch.equal(torch.cat([t[4] for t in b]), a[4])


def test_operator_attributes_of_the_reference_api(golden=None):
    """GaussianDeblurring.filter (degradations.py:59-69) and Superresolution.downsampling_matrix (:110-111) exist for callers
    that keep the OT-ODE loop in Python (ot_ode.py:98, 110)."""
    import pnpflow_amd.degradations as D
    from oracle import pnpflow_oracle as O
    g = D.GaussianDeblurring(1.0, 61, "fft", 3, 128, device="cpu")
    f = g.filter
    assert f.shape == (1, 3, 128, 128)
    np.testing.assert_allclose(f.cpu().numpy(), O.GaussianDeblurring(1.0, 61, "fft", 3, 128).filter.numpy(), atol=1e-9)
    m = D.Superresolution(2, 8).downsampling_matrix
    x = torch.arange(64.0)
    assert torch.equal(m @ x, x.view(8, 8)[::2, ::2].reshape(-1)) and torch.equal(torch.diag(m @ m.T), torch.ones(16))


def test_paintbrush_masks_are_seeded_prefix_consistent_and_match_the_oracle():
    """PaintbrushInpainting (degradations.py:47-52): the stroke parameters come from Python's random.seed(42) sequence exactly as
    in the reference (utils.py:904-924); image i's mask does not depend on the batch size (shards slice the global batch).
    Strokes are rasterised by a restatement of cv2.line (OpenCV's ThickLine: fixed-point quadrilateral + outline + midpoint-circle
    caps; cv2 is absent, parity unpinned) - product (pnpflow_amd/cv_draw.py) and oracle are written separately."""
    import random
    from oracle import pnpflow_oracle as O
    from pnpflow_amd.degradations import paintbrush_masks
    m4 = paintbrush_masks(4, 64, 96)
    assert m4.shape == (4, 64, 96) and set(np.unique(m4)) <= {0, 1} and 0.02 < (m4 == 0).mean() < 0.6
    np.testing.assert_array_equal(paintbrush_masks(2, 64, 96), m4[:2])
    np.testing.assert_array_equal(O.paintbrush_mask_array(3, 64, 96), m4[:3])
    np.testing.assert_array_equal(O.paintbrush_mask_array(2, 128, 128), paintbrush_masks(2, 128, 128))
    # the first stroke of the reference's sequence: endpoints / thickness from random.seed(42)
    rng = random.Random(42)
    x1, x2 = rng.randint(48 - 30, 48 + 30), rng.randint(48 - 30, 48 + 30); y1, y2 = rng.randint(32 - 30, 32 + 30), rng.randint(32 - 30, 32 + 30)
    assert m4[0, min(max(y1, 0), 63), min(max(x1, 0), 95)] == 0 and m4[0, min(max(y2, 0), 63), min(max(x2, 0), 95)] == 0
    with pytest.raises(Exception):
        paintbrush_masks(1, 32, 32)


def test_rectified_model_surface():
    """The reference's import paths for the rectified (NCSN++) net resolve to the engine classes; configs carry the reference's
    hyper-parameters; unsupported switches and a missing device fail loudly (no fallback)."""
    import pnpflow_amd._lib as L
    from pnpflow.image_generation.models.ncsnpp import NCSNpp
    from pnpflow.image_generation.models import utils as mutils
    from pnpflow.image_generation.configs.rectified_flow.celeba_hq_pytorch_rf_gaussian import get_config as cfg_celeba
    from pnpflow.image_generation.configs.rectified_flow.afhq_cat_pytorch_rf_gaussian import get_config as cfg_afhq
    import pnpflow_amd.image_generation.models.ncsnpp as A
    assert NCSNpp is A.NCSNpp and mutils.get_model("ncsnpp") is NCSNpp
    for get, ds in ((cfg_celeba, "CelebA-HQ-Pytorch"), (cfg_afhq, "AFHQ-CAT-Pytorch")):
        c = get()
        assert c.data.dataset == ds and c.data.image_size == 256 and c.data.centered is True
        m = c.model
        assert (m.name, m.nf, tuple(m.ch_mult), m.num_res_blocks, tuple(m.attn_resolutions)) == ("ncsnpp", 128, (1, 1, 2, 2, 2, 2, 2), 2, (16,))
        assert m.fir and list(m.fir_kernel) == [1, 3, 3, 1] and m.skip_rescale and m.scale_by_sigma and m.resblock_type == "biggan"
        assert (m.progressive, m.progressive_input, m.progressive_combine, m.embedding_type) == ("output_skip", "input_skip", "sum", "fourier")
        assert c.training.sde == "rectified_flow" and c.training.continuous is False
    bad = cfg_afhq(); bad.model.progressive = "residual"
    with pytest.raises(NotImplementedError):
        NCSNpp(bad)
    bad = cfg_afhq(); bad.data.centered = False
    with pytest.raises(NotImplementedError):
        NCSNpp(bad)
    if not torch.cuda.is_available():
        with pytest.raises(L.PnpFlowHipError):
            NCSNpp(cfg_afhq())
    # the module order of the oracle's restatement = the engine's expected state_dict (checked on the GPU box against the engine)
    from oracle import ncsnpp_oracle as NO
    shapes = NO.ncsnpp_param_shapes(NO.ncsnpp_config())
    assert len(shapes) == 645 and shapes["all_modules.0.W"] == (128,) and shapes["all_modules.3.weight"] == (128, 3, 3, 3)


def test_ncsnpp_create_rejects_unsupported_configurations_before_touching_a_device():
    """pf_ncsnpp_create validates the hyper-parameters first (works without a GPU): widths that are not multiples of 32, too wide
    levels, odd FIR lengths, un-centred data, sizes the level count does not divide."""
    import ctypes as C
    import pnpflow_amd._lib as L
    lib = L.load()

    def cfg(**kw):
        c = L.PfNcsnppCfg()
        base = dict(image_size=32, num_channels=3, nf=32, num_levels=2, num_res_blocks=1, num_attn_resolutions=1, fir_taps=4,
                    skip_rescale=1, scale_by_sigma=1, centered=1)
        base.update(kw)
        for k, v in base.items():
            setattr(c, k, v)
        c.ch_mult[0], c.ch_mult[1] = 1, kw.get("mult1", 2)
        c.attn_resolutions[0] = 16
        for i, v in enumerate([1, 3, 3, 1]):
            c.fir_kernel[i] = float(v)
        return c

    for bad in (dict(nf=48), dict(nf=160), dict(mult1=16), dict(fir_taps=3), dict(centered=0), dict(image_size=33), dict(num_channels=4)):
        h = C.c_void_p()
        c = cfg(**{k: v for k, v in bad.items() if k != "mult1"}, **({"mult1": bad["mult1"]} if "mult1" in bad else {}))
        assert lib.pf_ncsnpp_create(0, C.byref(c), C.byref(h)) == -1, bad
        assert b"unsupported NCSN++ configuration" in lib.pf_last_error(None)
    assert lib.pf_engine_set_solver_time_scale(None, 999.0) == -1


# ---------------------------------------------------------------------------------------------
# bench.py: `--gpus N` is the job size
# ---------------------------------------------------------------------------------------------
def _bench(args, env_extra):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PNPFLOW_BENCH_DRY="1", PNPFLOW_DIST_BACKEND="gloo", **env_extra)
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=root, env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with NO torchrun wrapper must run a 2-rank job (it re-executes itself under
    torch.distributed.run) and report n_gpus = 2.  PNPFLOW_BENCH_DRY=1: rendezvous + reduction only, no device work (this
    container has no GPU); the rank / world logic is the same code the measured run goes through."""
    import json
    out = _bench(["--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0"], {})
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["ranks_in_job"] == 2 and rec["config"]["global_batch"] == 8 and rec["dry_run"] is True
    one = _bench(["--gpus", "1", "--workload", "tiny"], {})
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_is_not_gpus():
    out = _bench(["--gpus", "8", "--workload", "tiny"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_cv2_line_restatement_known_answers():
    """What OpenCV's thick lines are known for (drawing.cpp ThickLine): an EVEN thickness t covers t + 1 pixel rows (the body is
    p +- round(t/2 * n) in 16.16 fixed point, both borders inclusive), the end caps are filled midpoint circles of radius (t + 1) // 2
    around the integer endpoints, a degenerate line is just the cap, and drawing is symmetric under x <-> y transposition."""
    from oracle import pnpflow_oracle as O
    from pnpflow_amd.cv_draw import circle_filled, thick_line
    a = thick_line(np.zeros((40, 40), np.uint8), (10, 20), (30, 20), 8)
    rows = np.where(a.any(1))[0]; cols = np.where(a.any(0))[0]
    assert (rows.min(), rows.max()) == (16, 24) and (cols.min(), cols.max()) == (6, 34)
    assert (a[16, 10:31] == 255).all() and a[16, 9] == 0 and (a[20, 6:35] == 255).all()      # body rows span x1..x2, the caps add 4 px
    b = thick_line(np.zeros((40, 40), np.uint8), (10, 20), (30, 20), 9)
    rows = np.where(b.any(1))[0]
    assert (rows.min(), rows.max()) == (15, 25)                                              # odd thickness: +-5
    c = thick_line(np.zeros((40, 40), np.uint8), (20, 10), (20, 30), 8)
    np.testing.assert_array_equal(c, a.T)
    d = thick_line(np.zeros((40, 40), np.uint8), (20, 20), (20, 20), 8)
    e = np.zeros((40, 40), np.uint8); circle_filled(e, 20, 20, 4)
    np.testing.assert_array_equal(d, e)
    assert [int((r != 0).sum()) for r in e[16:25]] == [1, 5, 7, 7, 9, 7, 7, 5, 1]          # the r = 4 midpoint disc with its one-pixel tips
    assert e[16, 20] and e[24, 20] and e[20, 16] and e[20, 24] and not e[16, 19]
    rng = np.random.default_rng(1)
    for _ in range(60):          # product vs oracle restatement, incl. strokes that touch the border (clipLine path of the product)
        p0 = tuple(int(v) for v in rng.integers(8, 88, 2)); p1 = tuple(int(v) for v in rng.integers(8, 88, 2)); t = int(rng.integers(8, 16))
        np.testing.assert_array_equal(thick_line(np.zeros((96, 96), np.uint8), p0, p1, t), O.cv2_thick_line(np.zeros((96, 96), np.uint8), p0, p1, t))


def test_lpips_oracle_properties_and_key_mapping():
    """The oracle's LPIPS restatement (lpips.LPIPS(net='alex') forward): zero for identical inputs, symmetric, positive, and the
    `normalize` flag is exactly the package's 2x - 1; the engine-side key mapping accepts the published checkpoint layouts."""
    from oracle import pnpflow_oracle as O
    from pnpflow_amd.lpips import canonical_state_dict
    sd = O.synthetic_lpips_state_dict(0)
    g = torch.Generator().manual_seed(0)
    a = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1; b = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    d = O.lpips_forward(sd, a, b)
    assert d.shape == (2,) and float(d.min()) > 0
    assert float(O.lpips_forward(sd, a, a).abs().max()) == 0.0
    np.testing.assert_allclose(O.lpips_forward(sd, b, a).numpy(), d.numpy(), rtol=1e-6)
    np.testing.assert_allclose(O.lpips_forward(sd, (a + 1) / 2, (b + 1) / 2, normalize=True).numpy(), d.numpy(), rtol=1e-5)
    full = {f"net.slice{j + 1}.{idx}.weight": sd[f"features.{idx}.weight"] for j, idx in enumerate((0, 3, 6, 8, 10))}
    full.update({f"lins.{j}.model.1.weight": sd[f"lin{j}.model.1.weight"] for j in range(5)})
    full["scaling_layer.shift"] = torch.zeros(1, 3, 1, 1)
    can = canonical_state_dict(full)
    assert sorted(can) == sorted([f"features.{i}.weight" for i in (0, 3, 6, 8, 10)] + [f"lin{k}" for k in range(5)])
    assert can["lin2"].shape == (384,) and torch.equal(can["features.6.weight"], sd["features.6.weight"])


# ---- image grids (reference utils.py:433-538) -----------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C", [(1, 3), (2, 3), (6, 3), (4, 1), (3, 3)])
def test_save_images_writes_the_reference_file_set(tmp_path, B, C):
    pytest.importorskip("matplotlib")
    import types
    from pnpflow_amd import utils
    g = torch.Generator().manual_seed(B * 10 + C)
    clean = torch.rand(B, C, 12, 12, generator=g) * 2 - 1
    noisy = clean + 0.1 * torch.randn(B, C, 12, 12, generator=g)
    rec = clean + 0.01 * torch.randn(B, C, 12, 12, generator=g)
    args = types.SimpleNamespace(save_path_ip=str(tmp_path), problem="denoising", method="pnp_flow", batch=1, num_channels=C, eval_split="test")
    utils.save_images(clean, noisy, rec, args, lambda t: t, iter='final')
    names = sorted(os.listdir(tmp_path))
    for word in ("clean", "noisy", "pnp_flow"):
        assert f"denoising_{word}_batch1_final.png" in names
    # batch < 4 on the test split: one .eps per image and kind, the image's own PSNR (data range 1, after postprocess) in the name
    eps = [n for n in names if n.endswith(".eps")]
    assert len(eps) == 3 * B
    p0 = 10 * np.log10(1.0 / float((((rec[0] - clean[0]) / 2).double() ** 2).mean()))
    assert f"denoising_pnp_flow_batch1_im0_iterfinal_pnsr{p0:4.2f}.eps" in names
    assert "denoising_clean_batch1_im0.eps" in names
    # an intermediate iteration writes one grid of the restored images only; later batches write no per-image files
    args.batch = 7
    utils.save_images(clean, noisy, rec, args, lambda t: t, iter=50)
    names2 = set(os.listdir(tmp_path)) - set(names)
    assert names2 == {"denoising_pnp_flow_batch7_iter50.png"}


def test_measurement_noise_sources():
    """`solver.measurement_noise_source` / `--opts measurement_noise device`: "cpu" is the device-independent default,
    "device" is the reference's own draw (pnp_flow.py:79-80: torch.manual_seed(batch); torch.randn_like(noisy_img) on the tensor's device) -
    taken for the GLOBAL batch and sliced, so shards reproduce the single-device run.  On a CPU device both coincide with the reference."""
    from pnpflow_amd.utils import draw_measurement_noise
    gshape = (6, 3, 8, 8)
    torch.manual_seed(4)
    ref = torch.randn_like(torch.empty(gshape))               # what the reference draws for batch 4 on this device
    for src in ("cpu", "device"):
        whole = draw_measurement_noise(4, gshape, 0, 6, torch.device("cpu"), src)
        assert torch.equal(whole, ref)
        part = draw_measurement_noise(4, gshape, 2, 5, torch.device("cpu"), src)
        assert torch.equal(part, ref[2:5])
    with pytest.raises(ValueError):
        draw_measurement_noise(0, gshape, 0, 6, torch.device("cpu"), "host")
