"""The N > 1 path end to end with two ranks on ONE GPU: two processes, each with its own engine over its image range, the
reduce buffers summed by torch.distributed (backend gloo: device tensors staged through the host; RCCL refuses two ranks
on one device).  Everything except the transport is what runs on 8 GPUs: partition, accumulate, reduce buffer (incl. the
LM diagonal corrections), finalize, replicated solve, EO slice exchange, summed Omega, and the FULL final pass expanded from the
reduced inverse with the expansion's inputs exchanged between the ranks."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene():
    from bundle_adjustment_amd import scene
    return scene.make_scene(8, 60, 40, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)


def _worker(rank, world, port, out_dir):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    from bundle_adjustment_amd import distributed, engine
    fp = _scene()
    lo, hi = distributed.partition_images(fp, world)[rank]
    eng = engine.Engine(fp, image_range=(lo, hi), apply_shared=(rank == 0), expansion_exchange=True)
    eng.set_parameters(fp.values)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    res = []
    for lam in (0.0, 0.5):
        dx = distributed.sharded_step(eng, dist, dev, fp.sigma2apriori, lam)
        res += [dx, [distributed.sharded_omega(eng, dist, dev, fp.sigma2apriori, dx)]]
    # the final pass of MatrixInversion.FULL on shards: all of Qxx expanded from the reduced inverse, the F bands and L_E^-1 of the
    # other rank's images arriving through the second all-reduce (jaicov_neq_expansion_buffer)
    dxf = distributed.sharded_step(eng, dist, dev, fp.sigma2apriori, 0.0, invert=engine.INVERT_FULL_EXPANDED)
    assert eng.cofactor_order() == fp.n_unknowns and eng.reduced_order() == fp.n_unknowns - 6 * fp.n_images
    res += [dxf, eng.get_cofactor()]
    # ... and a SECOND expanded pass on the same engines after the parameters have moved (the first pass used to leave the
    # other rank's L_E^-1 records in this engine's array, and the next expansion buffer carried them into the sum: EO cofactors R times too large)
    eng.update(dxf)
    dxg = distributed.sharded_step(eng, dist, dev, fp.sigma2apriori, 0.0, invert=engine.INVERT_FULL_EXPANDED)
    res += [dxg, eng.get_cofactor(), eng.get_parameters()]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate(res))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_oracle(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    fp = _scene()
    U = fp.n_unknowns
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(r0, r1)                      # replicated solve + exchanged EO slice: identical on both ranks
    o = oracle_mod.Oracle(fp)
    for i, lam in enumerate((0.0, 0.5)):
        dxo, _, _, _ = o.step(fp.values, fp.sigma2apriori, lam, False)
        base = i * (U + 1)
        np.testing.assert_allclose(r0[base:base + U], dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
        omo = o.omega(fp.values, fp.sigma2apriori, dxo)
        assert abs(r0[base + U] - omo) <= 1e-9 * omo
    dxo, Qo, _, _ = o.step(fp.values, fp.sigma2apriori, 0.0, True)          # dspsv + dsptri at full order (MX:338-366)
    base = 2 * (U + 1)
    np.testing.assert_allclose(r0[base:base + U], dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    nq = Qo.size
    Q = r0[base + U:base + U + nq]
    np.testing.assert_allclose(Q, Qo, rtol=0, atol=1e-9 * np.abs(Qo).max())
    # the second expanded pass, at the parameters the engines moved to
    base += U + nq
    vals = r0[base + U + nq:]
    assert vals.size == fp.values.size
    dxo2, Qo2, _, _ = o.step(vals, fp.sigma2apriori, 0.0, True)
    np.testing.assert_allclose(r0[base:base + U], dxo2, rtol=0, atol=1e-9 * max(np.abs(dxo2).max(), np.abs(dxo).max()))
    np.testing.assert_allclose(r0[base + U:base + U + nq], Qo2, rtol=0, atol=1e-9 * np.abs(Qo2).max())


def test_shards_without_the_exchange_promise_take_the_literal_full_route(oracle_mod):
    """a sharded engine whose caller has NOT promised to sum the expansion's inputs (engine option expansion_exchange
    off) cannot expand the reduced inverse -- it holds its own images' F bands and L_E^-1 only --, so a final pass announced as
    FULL_EXPANDED falls back to the literal FULL route (engine.hip effective_invert): the UNREDUCED system is assembled per shard,
    summed (here: two engines in one process, the buffers added with torch as the all-reduce would), factored and inverted at full order.
    Held to the oracle's dspsv + dsptri (MX:338-366)."""
    import torch
    from bundle_adjustment_amd import distributed, engine
    from bundle_adjustment_amd.distributed import DeviceArray
    fp = _scene()
    s2, U = fp.sigma2apriori, fp.n_unknowns
    dev = torch.device("cuda", 0)
    parts = distributed.partition_images(fp, 2)
    engs = [engine.Engine(fp, image_range=parts[r], apply_shared=(r == 0)) for r in range(2)]
    for e in engs:
        e.set_parameters(fp.values)
        e.prepare_inverse(engine.INVERT_FULL_EXPANDED)
        e.accumulate(s2, 0.0)
    assert all(e.reduced_order() == U for e in engs)                        # nothing was pre-eliminated: the literal route
    ts = [torch.as_tensor(DeviceArray(*e.reduce_buffer()), device=dev) for e in engs]
    assert ts[0].numel() == ts[1].numel() == U * (U + 1) // 2 + U
    tot = ts[0] + ts[1]
    for t in ts:
        t.copy_(tot)
    torch.cuda.synchronize(dev)
    out = []
    for e in engs:
        e.finalize(s2, 0.0)
        dx = e.solve(engine.INVERT_FULL_EXPANDED)
        assert e.cofactor_order() == U
        out.append((dx, e.get_cofactor()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    o = oracle_mod.Oracle(fp)
    dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, True)
    np.testing.assert_allclose(out[0][0], dxo, rtol=0, atol=1e-9 * np.abs(dxo).max())
    np.testing.assert_allclose(out[0][1], Qo, rtol=0, atol=1e-9 * np.abs(Qo).max())
    for e in engs:
        e.close()
