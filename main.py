"""CLI with the config surface of the reference's main.py (:29-56, :120-212):

    python main.py --opts dataset celeba problem inpainting method pnp_flow alpha 0.5 max_batch 25 batch_size_ip 4

Three YAML layers (main -> dataset -> method) then `--opts k v ...` (applied twice, as the
reference does), `args.dict_cfg_method` naming the results directory.  The restoration
runs on the HIP engine; datasets and checkpoints are not reachable offline, so when the
dataset folder / checkpoint is absent the run uses synthetic images / synthetic weights
and says so.
"""
import argparse
import os
import random

import numpy as np
import torch

from pnpflow_amd.utils import load_cfg_from_cfg_file, merge_cfg_from_list


def parse_args():
    parser = argparse.ArgumentParser(description='Main')
    cfg = load_cfg_from_cfg_file('./' + 'config/main_config.yaml')
    parser.add_argument('--opts', default=None, nargs=argparse.REMAINDER)
    a = parser.parse_args()
    if a.opts is not None:
        cfg = merge_cfg_from_list(cfg, a.opts)
    cfg.update(load_cfg_from_cfg_file(cfg.root + 'config/dataset_config/{}.yaml'.format(cfg.dataset)))
    method_file = cfg.root + 'config/method_config/{}.yaml'.format(cfg.method)
    cfg.update(load_cfg_from_cfg_file(method_file))
    if a.opts is not None:
        cfg = merge_cfg_from_list(cfg, a.opts)
    cfg.dict_cfg_method = {k: cfg[k] for k in load_cfg_from_cfg_file(method_file).keys()}
    return cfg


def make_degradation(problem, dim_image, num_channels, noise_type, device):
    """Problem table of the reference's main.py:120-179 -> (degradation, sigma_noise)."""
    from pnpflow_amd import degradations as D
    lap = noise_type == 'laplace'
    if problem == "denoising":
        return D.Denoising(), (0.3 if lap else 0.2)
    if problem == "inpainting":
        return D.BoxInpainting({128: 20, 256: 40}[dim_image]), (0.3 if lap else 0.05)
    if problem == "random_inpainting":
        return D.RandomInpainting(0.7), (0.3 if lap else 0.01)
    if problem == "superresolution":
        return D.Superresolution({128: 2, 256: 4}[dim_image], dim_image), (0.3 if lap else 0.05)
    if problem == "gaussian_deblurring_FFT":
        return D.GaussianDeblurring({128: 1.0, 256: 3.0}[dim_image], 61, "fft", num_channels, dim_image, device), (0.3 if lap else 0.05)
    raise ValueError("The problem you entered is not implemented by this engine: " + str(problem))


class SyntheticLoader:
    """Stands in for pnpflow/dataloaders.py when the dataset is not on disk: an iterable of
    (clean_img in [-1,1], labels) batches, the protocol solve_ip consumes (pnp_flow.py:70-73)."""

    def __init__(self, batch_size, channels, dim, n_batches, seed=1234):
        self.bs, self.c, self.dim, self.n, self.seed = batch_size, channels, dim, n_batches, seed

    def __iter__(self):
        for i in range(self.n):
            g = np.random.Generator(np.random.Philox(key=[self.seed, i]))
            x = torch.from_numpy(g.standard_normal(size=(self.bs, self.c, self.dim, self.dim), dtype=np.float32))
            k = torch.ones(self.c, 1, 3, 3) / 9.0
            for _ in range(5):
                x = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), k, groups=self.c)
            lo = x.amin(dim=(1, 2, 3), keepdim=True); hi = x.amax(dim=(1, 2, 3), keepdim=True)
            yield ((x - lo) / (hi - lo) * 2 - 1).contiguous(), torch.zeros(self.bs)


def main():
    args = parse_args()
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    print("device", device)
    if device.type != "cuda":
        raise SystemExit("pnpflow_amd needs an MI355X (there is no CPU path)")
    if args.seed is not None:
        random.seed(args.seed); torch.manual_seed(args.seed); np.random.seed(args.seed)
    from pnpflow_amd.methods.pnp_flow import PNP_FLOW
    from pnpflow_amd.utils import define_model, load_model

    (model, state) = define_model(args)
    if args.eval:
        model_path = args.output_root + 'model/{}/{}/model_final.pt'.format(args.dataset, args.model)
        if os.path.isfile(model_path):
            load_model(args.model, model, state, download=False, checkpoint_path=model_path, dataset=None, device=device)
        else:
            print(f"[pnpflow_amd] checkpoint {model_path} not found: using SYNTHETIC weights (PSNR is not meaningful)")
            from tools.synthetic_weights import synthetic_state_dict
            model.load_state_dict(synthetic_state_dict(model))
        model.eval()
        degradation, sigma_noise = make_degradation(args.problem, args.dim_image, args.num_channels, args.noise_type, device)
        print('Solving the {} inverse problem with the method {}...'.format(args.problem, args.method))
        print('sigma_noise', sigma_noise)
        print("[pnpflow_amd] dataset readers are out of scope offline: using synthetic clean images")
        loaders = {s: SyntheticLoader(args.batch_size_ip, args.num_channels, args.dim_image, args.max_batch) for s in ('train', 'val', 'test')}
        args.save_path = os.path.join(args.output_root, 'results_laplace' if args.noise_type == 'laplace' else 'results', args.dataset, args.model,
                                      args.problem, args.method, args.eval_split)
        os.makedirs(args.save_path, exist_ok=True)
        if args.method == 'pnp_flow':
            method = PNP_FLOW(model, device, args)
        elif args.method == 'ot_ode':
            from pnpflow_amd.methods.ot_ode import OT_ODE
            method = OT_ODE(model, device, args)
        else:
            raise ValueError("The method your entered does not exist")
        method.run_method(loaders, degradation, sigma_noise)


if __name__ == "__main__":
    main()
