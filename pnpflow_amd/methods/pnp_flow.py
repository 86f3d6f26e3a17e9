"""PNP_FLOW solver with the API of pnpflow/methods/pnp_flow.py (reference :10-188).

`solve_ip` keeps the reference's control flow (loader protocol, seeds, args mutation,
metric cadence) and hands each batch's inner loop (pnp_flow.py:93, 102-121) to the HIP
engine: pf_pnp_flow_restore runs gradient step -> interpolate -> U-Net -> average for
all iterations, one hipGraph replay per outer iteration.
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


class PNP_FLOW(object):

    def __init__(self, model, device, args):
        self.device = device
        self.args = args
        self.model = model.to(device)
        self.method = args.method
        self.coupling = self.args.model
        self.lib = _lib.load()
        # engine options (not in the reference): injected interpolation noise for parity runs,
        # Philox seed for throughput runs, hipGraph on/off
        self.noise = None          # optional (steps*num_samples, B, C, H, W) GPU tensor
        self.noise_seed = 0
        self.use_graph = True
        self.batch_samples = True  # the num_samples evaluations of an iteration run as one pass over num_samples*B images
        self.last_restored = None  # the final x of the last batch (the reference only writes it to disk)
        self.measurement_noise = None   # optional override of the torch.manual_seed(batch) draw (multi-GPU shards)
        # where the torch.manual_seed(batch) measurement noise is drawn: "cpu" (default: the same values on any device and on every
        # rank) or "device" - the reference's own behaviour (pnp_flow.py:79-80: torch.randn_like of a device tensor), drawn for the
        # GLOBAL batch on this rank's device generator and sliced, so that shards still reproduce the single-device run
        self.measurement_noise_source = getattr(args, "measurement_noise", "cpu")
        self.image_offset = 0      # multi-GPU shard: index of this shard's first image in the global batch (parallel.shard_range);
                                   # the shard then draws its slice of the global batch's interpolation noise (pf_pnp_params.elem_offset)
        self._interp_calls = 0

    # ---- the reference's small methods, kept for API parity ---------------------------------
    def model_forward(self, x, t):
        if self.coupling in {"ot", "indep"}:
            return self.model(x, t)
        if self.coupling == "rectified":          # pnp_flow.py:23-27: model_fn(x, t * 999)
            return self.model(x.type(torch.float), t * 999)
        raise NotImplementedError("only the 'ot'/'indep' U-Net and the 'rectified' NCSN++ net are implemented")

    def learning_rate_strat(self, lr, t):
        t = t.view(-1, 1, 1, 1)
        style = self.args.gamma_style
        if style == '1_minus_t':
            return lr * (1 - t)
        if style == 'sqrt_1_minus_t':
            return lr * torch.sqrt(1 - t)
        if style == 'alpha_1_minus_t':
            return lr * (1 - t) ** self.args.alpha
        return lr * torch.ones_like(t) if style == 'constant' else lr

    def grad_datafit(self, x, y, H, H_adj):
        if self.args.noise_type == 'gaussian':
            return H_adj(H(x) - y) / (self.args.sigma_noise ** 2)
        elif self.args.noise_type == 'laplace':
            r = H(x) - y
            return H_adj(2 * torch.heaviside(r, torch.zeros_like(r)) - 1) / self.args.sigma_noise
        raise ValueError('Noise type not supported')

    def interpolation_step(self, x, t):
        # a fresh eps per call, as torch.randn_like gives (pnp_flow.py:47-48): stream ids 2^63 + call number never collide with
        # restore_batch's streams (1 + (batch << 32) + iteration*num_samples + sample)
        eps = torch.empty_like(x)
        stream_id = (1 << 63) + self._interp_calls
        self._interp_calls += 1
        _lib.check(self.lib.pf_fill_normal_at(eps.data_ptr(), eps.numel(), self.noise_seed, stream_id,
                                              self.image_offset * x[0].numel(), _lib.current_stream_ptr()), None, "pf_fill_normal_at")
        return t * x + eps * (1 - t)

    def denoiser(self, x, t):
        v = self.model_forward(x, t)
        return x + (1 - t.view(-1, 1, 1, 1)) * v

    # ---- schedule scalars, computed with the reference's own fp32 expressions ------------------
    def _schedule(self, steps, lr, sigma_noise):
        delta = 1 / steps
        t_vals = np.empty(steps, dtype=np.float32)
        coef = np.empty(steps, dtype=np.float32)
        for it in range(int(steps)):
            t1 = torch.ones(1) * delta * it                          # pnp_flow.py:107-108
            lr_t = self.learning_rate_strat(lr, t1)                  # :109
            if not torch.is_tensor(lr_t):
                lr_t = torch.tensor([float(lr_t)])
            t_vals[it] = float(t1[0])
            # gaussian: grad/sigma^2 (pnp_flow.py:41); laplace: grad/sigma (:43)
            coef[it] = float(lr_t.reshape(-1)[0]) / (sigma_noise ** 2 if self.args.noise_type == 'gaussian' else sigma_noise)
        return t_vals, coef

    def restore_batch(self, noisy_img, degradation, sigma_noise, lr, iter_cb=None, cb_iterations=None):
        """Inner loop of solve_ip for one batch on the engine.  Returns x (B,C,H,W).
        iter_cb(iteration, x) is called on the host after the iterations in `cb_iterations` (None = every iteration);
        self.last_callback_seconds = host time spent inside those callbacks (excluded from `time_per_batch`)."""
        args = self.args
        steps, ns = int(args.steps_pnp), int(args.num_samples)
        B = noisy_img.shape[0]
        Cc, Hh = self.model.input_channels, self.model.input_height
        if B == 0:
            # empty shard (a batch with fewer images than ranks, e.g. the last partial batch): nothing to restore, but the
            # logging callbacks hold the job's collectives (per-image metric all_gather) and must be joined in the same order
            return parallel.empty_shard_result(noisy_img, (0, Cc, Hh, Hh), steps, iter_cb, cb_iterations)
        t_vals, coef = self._schedule(steps, lr, sigma_noise)
        if hasattr(self.model, "set_solver_time_scale"):
            # the engine's loop evaluates model(x, t * 999) for the rectified coupling (model_forward above).  NB the reference's
            # first iteration has t = 0, log(0 * 999) = -inf: its output is NaN from there on; the engine raises PF_ERR_NUMERIC.
            self.model.set_solver_time_scale(999.0 if self.coupling == "rectified" else 1.0)
        d = degradation.descriptor(B, Hh, Hh, noisy_img.device)
        prm = _lib.PfPnpParams()
        prm.steps, prm.num_samples = steps, ns
        prm.host_t = t_vals.ctypes.data_as(C.POINTER(C.c_float))
        prm.host_coef = coef.ctypes.data_as(C.POINTER(C.c_float))
        prm.seed = int(self.noise_seed)
        prm.stream_base = 1 + (int(getattr(args, "batch", 0)) << 32)
        prm.elem_offset = int(self.image_offset) * Cc * Hh * Hh
        if self.noise is not None:
            nz = self.noise.contiguous().float()
            assert nz.numel() == steps * ns * B * Cc * Hh * Hh
            prm.noise = nz.data_ptr()
        prm.use_graph = 1 if self.use_graph else 0
        prm.batch_samples = 1 if self.batch_samples else 0
        prm.noise_model = 1 if args.noise_type == 'laplace' else 0
        x = torch.empty((B, Cc, Hh, Hh), dtype=torch.float32, device=noisy_img.device)
        y = noisy_img.contiguous().float()
        holder = {"err": None}
        self.last_callback_seconds = 0.0
        if iter_cb is not None:
            def _cb(it, user):
                t_cb = perf_counter()
                try:
                    if holder["err"] is None:
                        iter_cb(it, x)
                except BaseException as exc:      # an exception must not unwind through the C frames: re-raised below
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
        holder["cb"] = cb
        with _lib.solver_stream():       # engine launches and metric callbacks on ONE stream (a real one: graph capture)
            _lib.check(self.lib.pf_pnp_flow_restore(self.model.handle, C.byref(d), C.byref(prm), y.data_ptr(), x.data_ptr(), B,
                                                    _lib.current_stream_ptr(), cb, None), self.model.handle, "pf_pnp_flow_restore")
        if holder["err"] is not None:
            raise holder["err"]
        return x

    def solve_ip(self, test_loader, degradation, sigma_noise, H_funcs=None):
        H = degradation.H
        H_adj = degradation.H_adj
        self.args.sigma_noise = sigma_noise
        steps = self.args.steps_pnp
        if self.args.noise_type == 'gaussian':
            self.args.lr_pnp = sigma_noise ** 2 * self.args.lr_pnp      # in place, as the reference (pnp_flow.py:61)
            lr = self.args.lr_pnp
        elif self.args.noise_type == 'laplace':
            self.args.lr_pnp = sigma_noise * self.args.lr_pnp           # pnp_flow.py:64-66
            lr = self.args.lr_pnp
        else:
            raise ValueError('Noise type not supported')

        # Multi-GPU (torchrun, one process per GPU): every rank walks the same loader and restores its contiguous slice
        # [lo, hi) of each batch; the batch-shaped random draws are taken for the whole batch and sliced, metrics are
        # all_gathered per image and written by rank 0 - so the result files equal a single-device run's (parallel.py).
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
            self.image_offset = lo
            noisy_img = H(clean_img.clone().to(self.device))
            gshape = (G,) + tuple(noisy_img.shape[1:])
            if self.measurement_noise is not None:
                noise = self.measurement_noise(batch, noisy_img)
            elif self.args.noise_type == 'laplace':
                # pnp_flow.py:81-85: unit-scale Laplace sample (scaled by sigma below), drawn on the CPU generator
                noise = torch.distributions.laplace.Laplace(torch.zeros(gshape), torch.ones(gshape)).sample()[lo:hi].to(self.device)
            else:
                # the reference draws on the device generator after torch.manual_seed(batch)
                # (pnp_flow.py:79-80); here the draw is made on the CPU generator so that it is
                # reproducible on any device (and identical on every rank), then sliced and moved.
                noise = utils.draw_measurement_noise(batch, gshape, lo, hi, self.device, self.measurement_noise_source)
            noisy_img = noisy_img + noise * sigma_noise
            clean_img = clean_img.to('cpu')

            if self.args.compute_time:
                torch.cuda.synchronize()
                t0 = perf_counter()
            if self.args.compute_memory:
                torch.cuda.reset_peak_memory_stats(self.device)

            def on_iter(iteration, x):
                utils.compute_psnr(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)
                utils.compute_ssim(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)
                utils.compute_lpips(clean_img, noisy_img, x.detach().clone(), self.args, H_adj, iter=iteration)

            # the reference's logging iterations (pnp_flow.py:128-139); the host is not involved on any other iteration
            log_its = [it for it in range(int(steps)) if it % 50 == 0 or self.should_save_image(it, steps)] if self.args.save_results else []
            x = self.restore_batch(noisy_img, degradation, sigma_noise, lr,
                                   iter_cb=on_iter if self.args.save_results else None, cb_iterations=log_its)
            self.last_restored = x

            if self.args.compute_memory:
                # torch's caching allocator (measurement, noise, output tensors) + the engine's own device memory
                utils.save_memory_use({"batch": batch, "max_allocated": torch.cuda.max_memory_allocated(self.device) + self.model.memory_bytes()},
                                      self.args)
            if self.args.compute_time:
                torch.cuda.synchronize()
                # the reference accumulates the iteration bodies only (pnp_flow.py:104-126): metric callbacks are excluded
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

    def run_method(self, data_loaders, degradation, sigma_noise, H_funcs=None):
        folder = utils.get_save_path_ip(self.args.dict_cfg_method)
        self.args.save_path_ip = os.path.join(self.args.save_path, folder)
        os.makedirs(self.args.save_path_ip, exist_ok=True)
        self.solve_ip(data_loaders[self.args.eval_split], degradation, sigma_noise, H_funcs)
