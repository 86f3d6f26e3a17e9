"""CLI with the config surface of the reference's main.py (:29-56, :120-212):

    python main.py --opts dataset celeba problem inpainting method pnp_flow alpha 0.5 max_batch 25 batch_size_ip 4

Three YAML layers (main -> dataset -> method) then `--opts k v ...` (applied twice, as the
reference does), `args.dict_cfg_method` naming the results directory.  The restoration
runs on the HIP engine.  Like the reference, the run FAILS when the checkpoint
(`model/<dataset>/<model>/model_final.pt`) or the dataset folder (`data/<dataset>/...`,
pnpflow_amd/dataloaders.py) is missing; `--opts synthetic True` is the explicit opt-in to
seed-fixed synthetic weights / images (smoke runs without the downloads) and writes under
`results_synthetic*/` so that such numbers can never mix with real ones.

Multi-GPU: `torchrun --nproc-per-node N main.py --opts ...` - one process per GPU, every
rank restores its contiguous slice of each test batch, one all_gather of per-image metrics
per logging point, rank 0 writes the reference's result files (pnpflow_amd/parallel.py).
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
    if problem == "paintbrush_inpainting":
        return D.PaintbrushInpainting(), (0.3 if lap else 0.05)
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
    from pnpflow_amd import parallel
    rank, world, local = parallel.init_from_env()
    device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    if rank == 0:
        print("device", device, f"(rank {rank}/{world})" if world > 1 else "")
    if device.type != "cuda":
        raise SystemExit("pnpflow_amd needs an MI355X (there is no CPU path)")
    if args.seed is not None:
        random.seed(args.seed); torch.manual_seed(args.seed); np.random.seed(args.seed)
    from pnpflow_amd.dataloaders import DataLoaders
    from pnpflow_amd.methods.pnp_flow import PNP_FLOW
    from pnpflow_amd.utils import define_model, load_model

    synthetic = bool(getattr(args, "synthetic", False))
    args.device_index = local
    (model, state) = define_model(args)
    if args.eval:
        # main.py:91-102 of the reference: the OT checkpoint is model_final.pt, the rectified (NCSN++) training state model_final.pth
        model_path = args.output_root + 'model/{}/{}/model_final.{}'.format(args.dataset, args.model, 'pth' if args.model == 'rectified' else 'pt')
        if os.path.isfile(model_path):
            load_model(args.model, model, state, download=False, checkpoint_path=model_path, dataset=None, device=device)
        elif synthetic:
            print(f"[pnpflow_amd] synthetic=True: checkpoint {model_path} not found, using SYNTHETIC weights (metrics are not meaningful)")
            from tools.synthetic_weights import synthetic_state_dict
            model.load_state_dict(synthetic_state_dict(model))
        else:
            raise FileNotFoundError(f"{model_path} not found (the reference's checkpoint, download.sh). "
                                    "Pass `--opts synthetic True` to run on seed-fixed synthetic weights instead.")
        model.eval()
        degradation, sigma_noise = make_degradation(args.problem, args.dim_image, args.num_channels, args.noise_type, device)
        if rank == 0:
            print('Solving the {} inverse problem with the method {}...'.format(args.problem, args.method))
            print('sigma_noise', sigma_noise)
        dl = DataLoaders(args.dataset, args.batch_size_ip, args.batch_size_ip, root=args.root)
        if dl.available(args.eval_split):
            loaders = dl.load_data()
            data_synthetic = False
        elif synthetic:
            print(f"[pnpflow_amd] synthetic=True: dataset files {dl.paths()} not found, using SYNTHETIC clean images")
            loaders = {s: SyntheticLoader(args.batch_size_ip, args.num_channels, args.dim_image, args.max_batch) for s in ('train', 'val', 'test')}
            data_synthetic = True
        else:
            raise FileNotFoundError(f"dataset files {dl.paths()} not found. Pass `--opts synthetic True` to run on synthetic images instead.")
        results = 'results_laplace' if args.noise_type == 'laplace' else 'results'
        if data_synthetic or not os.path.isfile(model_path):
            results = results.replace('results', 'results_synthetic')        # never mixes with real-data / real-weight results
        args.save_path = os.path.join(args.output_root, results, args.dataset, args.model, args.problem, args.method, args.eval_split)
        os.makedirs(args.save_path, exist_ok=True)
        if args.method == 'pnp_flow':
            method = PNP_FLOW(model, device, args)
        elif args.method == 'ot_ode':
            from pnpflow_amd.methods.ot_ode import OT_ODE
            method = OT_ODE(model, device, args)
        else:
            raise ValueError("The method your entered does not exist")
        method.run_method(loaders, degradation, sigma_noise)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
