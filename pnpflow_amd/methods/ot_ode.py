"""OT_ODE solver with the API of pnpflow/methods/ot_ode.py (reference :9-213).

Per iteration the reference runs a no-grad forward, a per-pixel closed-form solve, a second
forward with autograd graph and an input-gradient backward (ot_ode.py:71-147).  Here one
retained forward + the hand-written backward of the engine (pf_unet_forward_retain /
pf_unet_backward) replace the two forwards + autograd; the closed-form solve and the Euler
update are HIP pointwise kernels (pf_ot_ode_vec / pf_ot_ode_update).
"""
from __future__ import annotations

import ctypes as C
import os
from time import perf_counter

import numpy as np
import torch

from .. import _lib
from .. import parallel
from .. import utils


class OT_ODE(object):

    def __init__(self, model, device, args):
        self.device = device
        self.args = args
        self.model = model.to(device)
        self.method = args.method
        self.lib = _lib.load()
        self.use_graph = True           # one hipGraph per Euler step
        self.last_callback_seconds = 0.0
        self.init_noise = None          # optional override of the randn_like in `initialization` (parity runs)
        self.measurement_noise = None   # optional override of the torch.manual_seed(batch) draw
        self.measurement_noise_source = getattr(args, "measurement_noise", "cpu")      # "cpu" | "device" (the reference's: ot_ode.py:44-45), see PNP_FLOW
        self.last_restored = None

    def model_forward(self, x, t):
        if self.args.model == "ot":
            return self.model(x, t)
        if self.args.model == "rectified":        # ot_ode.py:21-25: model_fn(x, t * 999)
            return self.model(x.type(torch.float), t * 999)
        raise NotImplementedError("only the 'ot' U-Net and the 'rectified' NCSN++ net are implemented")

    def initialization(self, noisy_img, t0):
        noise = self.init_noise if self.init_noise is not None else torch.randn(noisy_img.shape).to(noisy_img.device)
        return t0 * noisy_img + (1 - t0) * noise

    # schedule scalars of iteration `it`, with the reference's own fp32 expressions (ot_ode.py:69-73, 96, 133-143)
    def _scalars(self, iteration, delta, problem, B, dev):
        t1 = torch.ones(B) * delta * iteration
        omt = 1 - t1
        if problem == "superresolution":    # reproduces `delta * iteration**2` (ot_ode.py:96)
            rt2 = torch.tensor((1 - delta * iteration) ** 2 / ((1 - delta * iteration) ** 2 + delta * iteration ** 2)).expand(B).clone()
        else:
            rt2 = (1 - t1) ** 2 / ((1 - t1) ** 2 + t1 ** 2)
        t = t1
        gamma = torch.ones(B) if self.args.gamma == "constant" else torch.sqrt(t / (t ** 2 + (1 - t) ** 2))
        coef = (1 - t) / t * gamma
        f = lambda a: a.to(torch.float32).contiguous().to(dev)
        return f(t1), f(omt), f(rt2), f(coef)

    def restore_batch(self, noisy_img, degradation, sigma_noise, iter_cb=None, cb_iterations=None):
        """The loop of solve_ip for one batch (ot_ode.py:49-52, 63-147) on the engine: pf_ot_ode_restore runs every Euler step on
        the device - schedule scalars from device tables, pre-allocated buffers, one hipGraph per step (retained forward ->
        closed-form / Fourier solve -> hand-written backward -> update).  iter_cb(iteration, x) is called on the host after the
        iterations in `cb_iterations` (None = every iteration)."""
        args = self.args
        problem = args.problem
        if hasattr(self.model, "set_solver_time_scale"):
            self.model.set_solver_time_scale(999.0 if args.model == "rectified" else 1.0)       # model_fn(x, t * 999), ot_ode.py:21-25
        if problem not in ("denoising", "inpainting", "random_inpainting", "paintbrush_inpainting", "superresolution", "gaussian_deblurring_FFT"):
            # OUT OF THE SURVEY 8 HOT-PATH SCOPE (unreachable for the five problems of main.py's table; kept from round 2,
            # host-orchestrated, golden-pinned, not part of any coverage claim): any other problem name takes the reference's generic
            # branch, a per-image GMRES on r_t^2 H H^T + sigma^2 I (ot_ode.py:118-128).
            # No entry of main.py's problem table reaches it; it runs as a host-orchestrated loop (not the device loop below).
            return self._restore_batch_generic(noisy_img, degradation, sigma_noise, iter_cb, cb_iterations)
        steps, delta = int(args.steps_ode), 1 / args.steps_ode
        first = int(steps * args.start_time)
        B = noisy_img.shape[0]
        Hh = self.model.input_height
        if B == 0:      # empty shard: join the logging collectives, restore nothing (see PNP_FLOW.restore_batch)
            return parallel.empty_shard_result(noisy_img, (0, self.model.input_channels, Hh, Hh), steps, iter_cb, cb_iterations, first=first)
        dev = noisy_img.device
        d = degradation.descriptor(B, Hh, Hh, dev)
        y = noisy_img.contiguous().float()
        x = self.initialization(degradation.H_adj(y.clone()), args.start_time).contiguous().float()     # ot_ode.py:50-52
        # per-iteration scalars with the reference's own fp32 expressions (one value per iteration: they are batch-constant)
        tab = np.zeros((4, steps), dtype=np.float32)
        for it in range(first, steps):
            t1, omt, rt2, coef = self._scalars(it, delta, problem, 1, "cpu")
            tab[0, it], tab[1, it], tab[2, it], tab[3, it] = float(t1[0]), float(omt[0]), float(rt2[0]), float(coef[0])
        prm = _lib.PfOtOdeParams()
        prm.steps, prm.first = steps, first
        fp = lambda r: tab[r].ctypes.data_as(C.POINTER(C.c_float))
        prm.host_t, prm.host_one_minus_t, prm.host_rt2, prm.host_coef = fp(0), fp(1), fp(2), fp(3)
        prm.sigma2 = float(np.float32(sigma_noise) ** 2) if problem == "superresolution" else float(sigma_noise ** 2)
        prm.delta = float(delta)
        prm.use_graph = 1 if self.use_graph else 0
        holder = {"err": None}
        self.last_callback_seconds = 0.0
        if iter_cb is not None:
            def _cb(it, user):
                t_cb = perf_counter()
                try:
                    if holder["err"] is None:
                        iter_cb(it, x)
                except BaseException as exc:      # must not unwind through the C frames: re-raised below
                    holder["err"] = exc
                self.last_callback_seconds += perf_counter() - t_cb
            cb = _lib.ITER_CB(_cb)
            if cb_iterations is not None:
                mask = np.zeros(steps, dtype=np.uint8)
                mask[[i for i in cb_iterations if 0 <= i < steps]] = 1
                holder["mask"] = mask
                prm.host_cb_mask = mask.ctypes.data
        else:
            cb = C.cast(None, _lib.ITER_CB)
        with _lib.solver_stream():       # engine launches and metric callbacks on ONE stream (a real one: graph capture)
            _lib.check(self.lib.pf_ot_ode_restore(self.model.handle, C.byref(d), C.byref(prm), y.data_ptr(), x.data_ptr(), B,
                                                  _lib.current_stream_ptr(), cb, None), self.model.handle, "pf_ot_ode_restore")
        if holder["err"] is not None:
            raise holder["err"]
        return x

    # ---- generic operator: Krylov solve of (r_t^2 H H^T + sigma^2 I) sol = d per image --------------------------------------
    @staticmethod
    def _gmres(apply_c, rhs, max_iter=100, tol=1e-6, atol=1e-6):
        """GMRES from a zero initial guess with the stopping rule of pnpflow/utils.py:972-1040 (|residual| < tol |rhs| or < atol,
        at most max_iter Krylov vectors).  Operator applications run on the engine's H / H_adj kernels; the Krylov bookkeeping
        (modified Gram-Schmidt, Givens-rotated Hessenberg least squares) is small host-driven tensor arithmetic."""
        nb = torch.linalg.vector_norm(rhs)
        if max_iter == 0 or float(nb) < 1e-8:
            return rhs                                           # utils.py:996-997 returns the right-hand side itself
        tiny = torch.finfo(rhs.dtype).eps
        basis = [rhs / nb if float(nb) > tiny else torch.zeros_like(rhs)]
        hess = np.zeros((max_iter + 1, max_iter), dtype=np.float64)
        cos_, sin_ = np.zeros(max_iter), np.zeros(max_iter)
        rhs_ls = np.zeros(max_iter + 1); rhs_ls[0] = float(nb)
        k = 0
        for k in range(max_iter):
            w = apply_c(basis[k])
            for i in range(k + 1):
                hik = torch.dot(w, basis[i]); w = w - hik * basis[i]; hess[i, k] = float(hik)
            wn = torch.linalg.vector_norm(w)
            hess[k + 1, k] = float(wn)
            basis.append(w / wn if float(wn) > tiny else torch.zeros_like(w))
            for i in range(k):                                   # earlier rotations on the new column
                a, b = hess[i, k], hess[i + 1, k]
                hess[i, k], hess[i + 1, k] = cos_[i] * a - sin_[i] * b, cos_[i] * b + sin_[i] * a
            r = float(np.hypot(hess[k, k], hess[k + 1, k]))
            cos_[k], sin_[k] = hess[k, k] / r, -hess[k + 1, k] / r
            hess[k, k] = cos_[k] * hess[k, k] - sin_[k] * hess[k + 1, k]; hess[k + 1, k] = 0.0
            rhs_ls[k + 1] = sin_[k] * rhs_ls[k]; rhs_ls[k] = cos_[k] * rhs_ls[k]
            if abs(rhs_ls[k + 1]) < tol * float(nb) or abs(rhs_ls[k + 1]) < atol:
                break
        import scipy.linalg
        yk = scipy.linalg.solve_triangular(hess[:k + 1, :k + 1], rhs_ls[:k + 1], lower=False)
        coeffs = torch.from_numpy(yk.astype(np.float32)).to(rhs.device)
        return torch.stack(basis[:k + 1], dim=0).T @ coeffs

    def _restore_batch_generic(self, noisy_img, degradation, sigma_noise, iter_cb=None, cb_iterations=None):
        args = self.args
        steps, delta = int(args.steps_ode), 1 / args.steps_ode
        first = int(steps * args.start_time)
        B = noisy_img.shape[0]
        dev = noisy_img.device
        Hop, Hadj = degradation.H, degradation.H_adj
        y = noisy_img.contiguous().float()
        x = self.initialization(Hadj(y.clone()), args.start_time).contiguous().float()
        label = 999.0 if args.model == "rectified" else 1.0
        self.last_callback_seconds = 0.0
        for it in range(first, steps):
            t1, omt, rt2, coef = self._scalars(it, delta, args.problem, B, dev)
            t1, omt, rt2, coef = (v.to(dev).float() for v in (t1, omt, rt2, coef))
            vt = self.model.forward_retain(x, t1 * label if label != 1.0 else t1)                 # v_theta(x, t), activations kept
            dres = y - Hop(x + omt.view(-1, 1, 1, 1) * vt)                                        # ot_ode.py:74-77
            sol = torch.zeros_like(dres)
            for i in range(B):
                def apply_c(z, i=i):
                    zz = z.reshape(dres.shape[1:]).unsqueeze(0)
                    return (rt2[i] * Hop(Hadj(zz)) + sigma_noise ** 2 * zz).reshape(-1)
                sol[i] = self._gmres(apply_c, dres[i].reshape(-1), 100).reshape(dres[i].shape)
            vec = Hadj(sol).contiguous()
            g = self.model.backward(vec)                                                          # J^T vec (ot_ode.py:137-138)
            x = x + delta * (vt + coef.view(-1, 1, 1, 1) * (vec + omt.view(-1, 1, 1, 1) * g))    # :141-147
            if iter_cb is not None and (cb_iterations is None or it in cb_iterations):
                t_cb = perf_counter(); iter_cb(it, x); self.last_callback_seconds += perf_counter() - t_cb
        return x

    def solve_ip(self, test_loader, degradation, sigma_noise, H_funcs=None):
        self.args.sigma_noise = sigma_noise
        H, H_adj = degradation.H, degradation.H_adj
        steps = self.args.steps_ode
        # multi-GPU: every rank restores its slice [lo, hi) of each batch; global draws sliced, metrics gathered (parallel.py)
        rank, world = parallel.rank_world()
        loader = iter(test_loader)
        for batch in range(self.args.max_batch):
            (clean_img, labels) = next(loader)
            self.args.batch = batch
            G = clean_img.shape[0]
            lo, hi = parallel.shard_range(G, rank, world)
            if world > 1:
                clean_img = clean_img[lo:hi]
                if hasattr(degradation, "set_shard"):
                    degradation.set_shard(G, lo)
            noisy_img = H(clean_img.clone().to(self.device))
            gshape = (G,) + tuple(noisy_img.shape[1:])
            if self.measurement_noise is not None:
                noise = self.measurement_noise(batch, noisy_img)
            else:
                noise = utils.draw_measurement_noise(batch, gshape, lo, hi, self.device, self.measurement_noise_source)      # ot_ode.py:44-45
            noisy_img = noisy_img + noise * sigma_noise
            clean_img = clean_img.to('cpu')
            if (world > 1 or self.measurement_noise_source == "device") and self.init_noise is None:
                # `initialization` draws randn_like(H_adj(y)) right after the measurement noise (ot_ode.py:27-28, 50-52): global draw, sliced
                # (on the generator the measurement noise came from)
                full = (G,) + tuple(clean_img.shape[1:])
                if self.measurement_noise_source == "device":
                    init = torch.randn(full, dtype=torch.float32, device=self.device)[lo:hi].contiguous()
                else:
                    init = torch.randn(full, dtype=torch.float32)[lo:hi].to(self.device)
            else:
                init = None
            if self.args.compute_time:
                torch.cuda.synchronize(); t0 = perf_counter()
            if self.args.compute_memory:
                torch.cuda.reset_peak_memory_stats(self.device)

            def on_iter(iteration, x):
                utils.compute_psnr(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)
                utils.compute_ssim(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)
                utils.compute_lpips(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)

            # the reference's logging iterations (ot_ode.py:149-150): the host is not involved on any other iteration
            log_its = [it for it in range(int(steps * self.args.start_time), int(steps))
                       if it % 10 == 0 or self.should_save_image(it, steps)] if self.args.save_results else []

            saved_init = self.init_noise
            if init is not None:
                self.init_noise = init
            try:
                x = self.restore_batch(noisy_img, degradation, sigma_noise, iter_cb=on_iter if self.args.save_results else None,
                                       cb_iterations=log_its)
            finally:
                self.init_noise = saved_init
            self.last_restored = x
            if self.args.compute_memory:
                utils.save_memory_use({"batch": batch, "max_allocated": torch.cuda.max_memory_allocated(self.device) + self.model.memory_bytes()},
                                      self.args)
            if self.args.compute_time:
                torch.cuda.synchronize()
                utils.save_time_use({"batch": batch, "time_per_batch": perf_counter() - t0 - self.last_callback_seconds}, self.args)
            if self.args.save_results:
                utils.save_images(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter='final')
                utils.compute_psnr(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=int(steps) - 1)
                utils.compute_ssim(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=int(steps) - 1)
                utils.compute_lpips(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=int(steps) - 1)
        if self.args.save_results:
            utils.compute_average_psnr(self.args)
            utils.compute_average_ssim(self.args)
            utils.compute_average_lpips(self.args)
        if self.args.compute_memory:
            utils.compute_average_memory(self.args)
        if self.args.compute_time:
            utils.compute_average_time(self.args)

    def should_save_image(self, iteration, steps):
        return iteration % (steps // 10) == 0

    def run_method(self, data_loaders, degradation, sigma_noise):
        folder = utils.get_save_path_ip(self.args.dict_cfg_method)
        self.args.save_path_ip = os.path.join(self.args.save_path, folder)
        os.makedirs(self.args.save_path_ip, exist_ok=True)
        self.solve_ip(data_loaders[self.args.eval_split], degradation, sigma_noise)
