"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding helpers and the one
collective of the data path (PSNR all_gather).  No GPU, no engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pnpflow_oracle as O
from pnpflow_amd.parallel import empty_shard_result, gather_in_image_order, global_measurement_noise, mean_psnr, shard_range


def test_shard_range_partitions():
    for G in (1, 7, 32, 128, 255):
        for W in (1, 2, 3, 8):
            spans = [shard_range(G, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, G, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(G, rank, world)
    # per-image "PSNR" = global image index -> gathered vector must be 0..G-1 in order
    local = torch.arange(lo, hi, dtype=torch.float32) + 0.25
    allp = gather_in_image_order(local)
    m = mean_psnr(local)
    noise = global_measurement_noise(3, (G, 1, 4, 4), lo, hi)
    # shard of the global random-inpainting mask
    mask = np.random.RandomState(42).binomial(n=1, p=0.3, size=(G, 8, 8))[lo:hi]
    # the metric bookkeeping of a sharded solve_ip: gathered mean, written by rank 0 only
    from pnpflow_amd import utils
    from pnpflow_amd.parallel import rank_world
    assert rank_world() == (rank, world)
    args = utils.CfgNode(dict(save_path_ip=os.environ["PF_TEST_DIR"], batch=0))
    utils._append_metric(args, "psnr", "rec", 5, utils._global_mean(local))
    q.put((rank, allp.tolist(), m, noise.numpy(), mask))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [8, 7])
def test_two_rank_gather_and_global_draws(G, tmp_path):
    world, port = 2, _free_port()
    os.environ["PF_TEST_DIR"] = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, G, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [i + 0.25 for i in range(G)]
    for r in res:
        assert r[1] == expect and abs(r[2] - float(np.mean(expect))) < 1e-9
    g = torch.Generator().manual_seed(3)
    full = torch.randn((G, 1, 4, 4), generator=g).numpy()
    np.testing.assert_array_equal(np.concatenate([r[3] for r in res]), full)
    np.testing.assert_array_equal(np.concatenate([r[4] for r in res]), O.random_mask_array(G, 8, 8, 0.7))
    lines = open(os.path.join(str(tmp_path), "psnr_rec_batch0.txt")).read().strip().splitlines()
    assert len(lines) == 1 and lines[0].split()[0] == "5" and abs(float(lines[0].split()[1]) - float(np.mean(expect))) < 1e-9


def _worker_empty(rank, world, port, G, q):
    """A global batch smaller than the job (G < world): the ranks past G own an EMPTY shard.  They must contribute a 0-length
    vector to every metric gather and call the logging callbacks of the iterations the other ranks log (a rank that
    skips them leaves the others blocked in all_gather)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pnpflow_amd import utils
    lo, hi = shard_range(G, rank, world)
    args = utils.CfgNode(dict(save_path_ip=os.environ["PF_TEST_DIR"], batch=0))
    log_its = [0, 5, 9]
    seen = []

    def on_iter(it, x):        # what solve_ip's callback does: one gather per logging iteration
        seen.append(it)
        per_image = torch.arange(lo, lo + x.shape[0], dtype=torch.float32) + it
        utils._append_metric(args, "psnr", "rec", it, utils._global_mean(per_image))

    if hi == lo:
        x = empty_shard_result(torch.zeros(1), (0, 3, 4, 4), steps=10, iter_cb=on_iter, cb_iterations=log_its)
        assert x.shape == (0, 3, 4, 4)
        assert utils.psnr_per_image(x, x).numel() == 0 and utils.ssim_per_image(x, x).numel() == 0     # no engine call for 0 images
    else:
        x = torch.zeros(hi - lo, 3, 4, 4)
        for it in log_its:
            on_iter(it, x)
    q.put((rank, seen, hi - lo))
    dist.barrier()
    dist.destroy_process_group()


def test_empty_shard_joins_every_collective(tmp_path):
    world, port, G = 2, _free_port(), 1
    os.environ["PF_TEST_DIR"] = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_empty, args=(r, world, port, G, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][2] == 1 and res[1][2] == 0 and res[0][1] == res[1][1] == [0, 5, 9]
    lines = [l.split() for l in open(os.path.join(str(tmp_path), "psnr_rec_batch0.txt")).read().strip().splitlines()]
    assert [(int(a), float(b)) for a, b in lines] == [(0, 0.0), (5, 5.0), (9, 9.0)]      # the mean over the ONE image of the global batch


def _worker_images(rank, world, port, G, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from pnpflow_amd import utils
    lo, hi = shard_range(G, rank, world)
    g = torch.Generator().manual_seed(11)
    clean = (torch.rand(G, 3, 8, 8, generator=g) * 2 - 1)
    rec = clean + 0.02 * torch.randn(G, 3, 8, 8, generator=g)
    args = types.SimpleNamespace(save_path_ip=os.environ["PF_TEST_DIR"], problem="denoising", method="pnp_flow", batch=2, num_channels=3, eval_split="test")
    # every rank calls it with its shard (rank 1 of G = 1: an empty one); rank 0 draws the GLOBAL batch
    utils.save_images(clean[lo:hi], clean[lo:hi], rec[lo:hi], args, lambda t: t, iter='final')
    p = [10 * np.log10(1.0 / float((((rec[i] - clean[i]) / 2).double() ** 2).mean())) for i in range(G)]
    q.put((rank, p))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [5, 1])
def test_save_images_gathers_the_shards_in_image_order(G, tmp_path):
    """save_images under two ranks: the per-image .eps files rank 0 writes carry the PSNR of GLOBAL image i (shards of 3 + 2 images;
    with G = 1 rank 1 owns an empty shard and still has to join the gathers)."""
    pytest.importorskip("matplotlib")
    world, port = 2, _free_port()
    os.environ["PF_TEST_DIR"] = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_images, args=(r, world, port, G, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = os.listdir(tmp_path)
    assert "denoising_pnp_flow_batch2_final.png" in names
    for i, psnr in enumerate(res[0]):
        assert f"denoising_pnp_flow_batch2_im{i}_iterfinal_pnsr{psnr:4.2f}.eps" in names, (i, psnr, sorted(names))
    assert sum(n.endswith(".eps") for n in names) == 3 * G
