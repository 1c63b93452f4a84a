"""N > 1 path rehearsed on the CPU: world_size-2 gloo, image sharding + one all-reduce of the packed normal equations.

The per-rank accumulation is done by the CPU oracle here (test infrastructure standing in for the HIP engine, which
needs a GPU); what is under test is the host logic that is shared with the GPU path: the image partition, the
"rank 0 contributes the shared groups" rule, the packed [N | n] buffer layout and the replicated solve."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as orc
    from bundle_adjustment_amd import distributed, scene
    fp = scene.config(name)
    o = orc.Oracle(fp)
    s2 = fp.sigma2apriori
    lo, hi = distributed.partition_images(fp, world)[rank]
    N, n = o.accumulate(fp.values, s2, lo, hi, shared=(rank == 0))
    buf = torch.from_numpy(np.concatenate([N, n]))          # the engine's reduce buffer layout: [packed N | n]
    dist.all_reduce(buf)
    tot = buf.numpy()
    N, n = tot[:fp.packed_length].copy(), tot[fp.packed_length:].copy()
    V = o.finalize(fp.values, N, n)
    o.precondition(V, N, n)
    assert o.solve(N, n, False) == 0
    o.precondition(V, None, n)
    np.save(os.path.join(out_dir, f"dx_{rank}.npy"), n)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["tiny_block", "tiny_free"])
def test_two_rank_sharding_matches_single_rank(tmp_path, oracle_mod, name):
    from bundle_adjustment_amd import distributed, scene
    fp = scene.config(name)
    parts = distributed.partition_images(fp, 2)
    assert parts[0][0] == 0 and parts[-1][1] == fp.n_images and parts[0][1] == parts[1][0]
    assert 0 < parts[0][1] < fp.n_images
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    dx0 = np.load(tmp_path / "dx_0.npy"); dx1 = np.load(tmp_path / "dx_1.npy")
    np.testing.assert_array_equal(dx0, dx1)                 # replicated solve: bitwise identical on both ranks
    o = oracle_mod.Oracle(fp)
    ref, _, _, _ = o.step(fp.values, fp.sigma2apriori)
    np.testing.assert_allclose(dx0, ref, rtol=0, atol=1e-10 * np.abs(ref).max())


def test_partition_is_balanced_and_contiguous():
    from bundle_adjustment_amd import distributed, scene
    fp = scene.config("cfg3")
    for world in (1, 2, 4, 8):
        parts = distributed.partition_images(fp, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == fp.n_images
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        cost = distributed.image_costs(fp)
        loads = [cost[a:b].sum() for a, b in parts]
        assert max(loads) <= 1.25 * (sum(loads) / world) + cost.max()
