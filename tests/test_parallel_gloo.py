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
from pnpflow_amd.parallel import gather_in_image_order, global_measurement_noise, mean_psnr, shard_range


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
