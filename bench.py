#!/usr/bin/env python
"""bench.py -- LM iterations/s of the MI355X normal-equation engine on the BASELINE.json headline scene.

    python bench.py --gpus N --steps K --warmup W
        N > 1: one rank per GPU.  Either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
        (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or started plainly: bench.py then starts the N ranks itself
        (torch.distributed.run as a child process, before anything touches a GPU) and exits with their status.

A "step" is one intermediate Gauss-Newton/LM pass of BundleAdjustment.estimateModel (BundleAdjustment.java:228-355):
residual + Jacobian rows, N = A'PA / n = A'Pw, datum/damping/preconditioner, factorisation + substitution, parameter
update -- inputs resident in HBM when the timed region starts.  Workload at N = 1: config 4 of BASELINE.json
(500 images x 5000 points, 500 image points per image, one dense 1000 x 1000 dispersion per image + a dense 45 x 45
control-point block, U = 18 014).  For N > 1 the images are sharded contiguously over the ranks (strong scaling), the
packed normal equations are summed with one all-reduce (RCCL) and every rank solves the identical system.

One JSON line on stdout (rank 0): metric/value (+ roofline of the dominant kernel + cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD's public MI355X fp64 matrix (= fp64 vector) peak; the local microarch guide lists
                               # no fp64 row.  bench also reports the measured issue-rate ceiling (roofline.peak_measured).


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--iterations-only", action="store_true",
                    help="only the timed LM passes (no final passes, no dense-mode measurement, no CPU baseline): profiling runs")
    ap.add_argument("--no-dense-mode", action="store_true", help="skip the densified J'WJ MFMA measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(fp, eng, sigma2):
    """CPU restatement (oracle, single thread) on a bounded sample of the same workload, extrapolated to one pass.

    sample: (a) the fair dense-block assembly of `nb` of the image blocks (T = P A, A'T on the compact columns),
    (b) the packed Bunch-Kaufman factorisation dsptrf (what MX.solve runs, MathExtension.java:348) on the leading
    n_s x n_s block of the preconditioned normal matrix taken from the engine, extrapolated with the U^3 law."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    o = orc.Oracle(fp)
    U = fp.n_unknowns
    t_asm = 0.0
    sample = []
    if fp.n_image_blocks > 0:
        nb = 2
        N = np.zeros(fp.packed_length); n = np.zeros(U)
        for b in range(nb):
            Pm = o.block_weight(sigma2, b)       # cached by the reference after the first pass: not timed
            t = time.perf_counter()
            o.block_fair(fp.values, sigma2, b, Pm, N, n)
            t_asm += time.perf_counter() - t
        t_asm *= fp.n_image_blocks / nb
        sample.append(f"fair assembly of {nb}/{fp.n_image_blocks} image blocks")
        del N
    else:
        N = np.zeros(fp.packed_length); n = np.zeros(U)
        cnt = min(fp.n_image_points, 4000)
        t = time.perf_counter()
        o.faithful_image_points(fp.values, sigma2, 0, cnt, N, n)
        t_asm = (time.perf_counter() - t) * fp.n_image_points / cnt
        sample.append(f"faithful rows+stacking of {cnt}/{fp.n_image_points} image points")
    # factorisation sample
    Np, _ = eng.get_normal()
    ns = min(U, 2200)
    ap = Np[: ns * (ns + 1) // 2].copy()
    del Np
    dg = ap[np.arange(ns) * (np.arange(ns) + 3) // 2].copy()
    V = np.where(dg > 2.0 ** -53, 1.0 / np.sqrt(np.abs(dg) + 1e-300), 1.0)
    L = orc.lib()
    L.oracle_precondition(ns, V.ctypes.data_as(orc._pd), ap.ctypes.data_as(orc._pd), None)
    ipiv = np.zeros(ns, np.int32)
    t = time.perf_counter()
    L.oracle_dsptrf(ns, ap.ctypes.data_as(orc._pd), ipiv.ctypes.data_as(orc._pi))
    t_fac_s = time.perf_counter() - t
    t_fac = t_fac_s * (U / ns) ** 3
    sample.append(f"packed dsptrf of the leading {ns}x{ns} block ({t_fac_s:.1f} s) extrapolated by (U/{ns})^3")
    total = t_asm + t_fac
    res = {"value": 1.0 / total, "unit": "iterations/s", "cores": 1, "kind": "port", "host_nproc": os.cpu_count(), "host_cpu": _cpu_model(),
           "sample": "; ".join(sample) + f"; estimated pass = {t_asm:.0f} s assembly + {t_fac:.0f} s factorisation"}
    # the same single-threaded port run ONCE in full at this config (tests/golden/make_cfg4_golden.py, in the build container):
    # measured, not extrapolated -- quoted beside the live sample so that the extrapolation can be judged
    meas = os.path.join(ROOT, "tests", "golden", "cfg4", "cfg4_oracle.json")
    if fp.n_unknowns == 18014 and os.path.exists(meas):
        try:
            m = json.load(open(meas))
            sec = m["seconds"]
            res["measured_full_pass"] = {"kind": "port, measured", "cores": 1, "intermediate_pass_s": sec["pass1_total"],
                                         "final_pass_with_inverse_s": sec["pass2_total"], "stages_s": sec["pass1"],
                                         "final_pass_stages_s": sec["pass2"], "host": m["host"],
                                         "iterations_per_s": 1.0 / sec["pass1_total"]}
        except Exception:
            pass
    # the packed Bunch-Kaufman factorisation at the FULL order, measured on a host of this pool (EPYC 9575F, the model every GPU box
    # of the pool has shown) by scripts/cfg4_exact.py on the device's own matrix: no extrapolation for the dominant CPU stage
    log = os.path.join(ROOT, "profiles", "r03_cfg4_accuracy.log")
    if fp.n_unknowns == 18014 and os.path.exists(log):
        import re
        found = dict((int(o), float(t)) for o, t in re.findall(r"dsptrf of order (\d+): (\d+) s", open(log).read()))
        if found:
            res["measured_dsptrf_on_pool_host_s"] = {"by_order": found, "threads_running": 2, "cores_each": 1,
                                                     "source": "profiles/r03_cfg4_accuracy.log (scripts/cfg4_exact.py, two factorisations side by side)"}
    return res


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def spawn_ranks(a):
    """--gpus N > 1 without a launcher: start N ranks (one per GPU) as a child job and return its exit code.  Nothing in
    this process has touched a GPU yet (torch.cuda.device_count() does not initialise one on this stack)."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs, this node shows {have}: refusing to print an N = 1 line "
              f"for an N = {a.gpus} request", file=sys.stderr)
        return 2
    port = int(os.environ.get("MASTER_PORT", "0")) or 29500 + os.getpid() % 400
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: the two must agree")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("JAICOV_BENCH_FORCE_DIST"))   # the env var rehearses the collective path on 1 GPU
    from bundle_adjustment_amd import distributed, engine, scene

    fp = scene.config(a.config)
    lo, hi = distributed.partition_images(fp, world)[rank]
    t_create = time.perf_counter()
    eng = engine.Engine(fp, device=local, image_range=(lo, hi) if use_dist else None, apply_shared=(rank == 0), expansion_exchange=use_dist)
    create_wall_ms = 1e3 * (time.perf_counter() - t_create)
    create_timings = eng.create_timings()
    eng.set_parameters(fp.values)
    s2 = fp.sigma2apriori
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints its version banner on stdout when the first communicator is created: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            # The communicator is created AFTER the engine and without device_id (lazy).  Through round 4 the other order slowed the
            # factorisation by 25 % (30.1 -> 38.4 ms, world size 1): hardware-queue oversubscription; with one
            # CU-masked queue per engine either order gives the same time (profiles/r05_c_dist_ab.log).  The order is kept.
            dist.init_process_group("nccl")
            warm = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", local))
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def step():
        if use_dist:
            dx = distributed.sharded_step(eng, dist, torch.device("cuda", local), s2)
        else:
            eng.build(s2, 0.0)
            dx = eng.solve(False)
        eng.update(dx)
        return dx

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local])
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    eng.set_profiling(True)
    eng.kernel_stats(reset=True)
    stage = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        for k, v in eng.timings().items():
            stage[k] = stage.get(k, 0.0) + v
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ks = eng.kernel_stats()
    eng.set_profiling(False)
    # health of the dataflow factorisation on EVERY rank (a repeated factorisation costs >= 0.25 s: a line measured with one is flagged)
    flow = [ks["flow_retries"], ks["flow_stale_events"], ks["flow_stale_confirmed"], ks["flow_rescued"]]
    if use_dist:
        ft = torch.tensor(flow, dtype=torch.float64, device="cuda")
        dist.all_reduce(ft)
        flow = [int(v) for v in ft.tolist()]

    out = None
    if rank == 0:
        U = fp.n_unknowns
        achieved = ks["flops"] / (ks["ms"] * 1e-3) / 1e12 if ks["ms"] > 0 else 0.0
        out = {
            "metric": "LM iterations/sec", "value": a.steps / elapsed, "unit": "iterations/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.config}: {fp.n_images} images x {fp.n_points} points, {fp.n_image_points} image points, "
                                   f"{fp.n_image_blocks} dense per-image dispersion blocks, U={U}, d={fp.rank_defect}",
                       "parallelism": (f"images sharded over {world} rank(s), packed reduced N all-reduced (RCCL), solve REPLICATED on every rank: "
                                       "expected_scaling predicts N > 1 slower than N = 1" if world > 1 or use_dist else
                                       "1 rank: no collective; with N ranks the images shard, the packed reduced N is all-reduced and the solve is replicated"),
                       "rccl_ranks": (dist.get_world_size() if use_dist else 0)},
            "stage_ms_per_step": {k: v / a.steps for k, v in stage.items()},
            "flow": {"retries": flow[0], "stale_events": flow[1], "stale_confirmed": flow[2], "rescued": flow[3],
                     "ok": flow[0] == 0, "note": "summed over ranks since engine creation; retries > 0 = a factorisation was abandoned and repeated inside the timed region or the warm-up"},
            # engine creation (one-time; jaicov_neq_create_timings): the FIRST engine of the process, which also pays the first launch of
            # every kernel and the HIP context; scripts/create_time.py times the 2nd and 3rd (0.20 s at config 4)
            "create_ms": {"wall_first_engine_of_the_process": create_wall_ms, **create_timings,
                          "note": "dispersions_to_weights_ms = upload of the dense dispersions (dispersion_upload_host_ms of host time, pageable memory) + batched Cholesky / inverse / refinement (batchinv.hip)"},
            "refinement": {"steps_per_solve": ks["refine_steps"], "last_correction_rel": ks["last_refinement_correction"],
                           "note": "iterative refinement of dx (two-fold-precision residual + forward/backward substitution), inside the timed step (stage 'solve')"},
            "roofline": {"kernel": ("gemm_f64_kernel<0, 0, 128, 128, 1> (Cholesky trailing update of the stream-scheduled factorisation, fp64 MFMA 16x16x4)"
                                    if os.environ.get("JAICOV_FACTOR_FORM") == "streams" else
                                    "chol_tile_kernel<2, false> (dataflow Cholesky: the whole factorisation of the EO-reduced normal matrix in one "
                                    "persistent launch, fp64 MFMA 16x16x4, with potrf_chain_kernel's two workgroups beside it for the diagonal "
                                    "blocks; algorithmic flops = order^3 / 3; symbols as listed by rocprofv3)"),
                         "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "launches": ks["launches"], "avg_launch_ms": ks["ms"] / max(ks["launches"], 1),
                         "algorithmic_flops_per_launch": ks["flops"] / max(ks["launches"], 1)},
        }
        # second roofline, for the HBM-bound half of the pass: algorithmic bytes of the structure-aware assembly (DESIGN.md
        # section 4) over its measured stage time.  Per dense image group of m rows: D^-1 read in full for T = D^-1 [A_c | w]
        # (8 m^2) and its lower triangle once more by the point x point gather, which forms the EO-eliminated weights
        # P' = sigma2 D^-1 - U U' on the fly (4 m^2; nothing of P' is written or read back any more); the lower triangle of the
        # reduced N written once by the gather (4 e0^2; the factorisation reads N itself: no scaled copy); Jacobian rows written
        # and read once (2 x 0.42 kB per image point).
        if fp.n_image_blocks and world == 1:
            m2 = float(np.sum((2.0 * np.diff(fp.blk_ip_begin)) ** 2))
            e0 = eng.reduced_order()
            abytes = 8.0 * m2 + 4.0 * m2 + 4.0 * float(e0) ** 2 + 2.0 * 420.0 * fp.n_image_points
            asm_ms = stage.get("assembly", 0.0) / a.steps
            out["assembly_roofline"] = {"bound": "hbm", "achieved": abytes / (asm_ms * 1e-3) / 1e9 if asm_ms > 0 else 0.0,
                                        "peak": 8000.0, "unit": "GB/s", "frac": abytes / (asm_ms * 1e-3) / 1e9 / 8000.0 if asm_ms > 0 else 0.0,
                                        "algorithmic_bytes_per_pass": abytes, "stage_ms": asm_ms}
        # measured issue ceiling of v_mfma_f64_16x16x4_f64 on this device (back-to-back independent MFMAs in registers):
        # constant operands / random operands (the chip is power-limited on fp64 MFMA, the two differ)
        try:
            import ctypes as C
            L = engine.load_library()
            L.jaicov_debug_mfma_peak.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            meas = []
            for iters in (20000, -20000):
                ms_, tf_ = C.c_double(), C.c_double()
                L.jaicov_debug_mfma_peak(2048, iters, C.byref(ms_), C.byref(tf_))
                meas.append(tf_.value)
            out["roofline"]["peak_measured"] = {"constant_operands": meas[0], "random_operands": meas[1], "unit": "TFLOP/s"}
            if meas[1] > 0:
                out["roofline"]["frac_of_measured_peak"] = achieved / meas[1]     # against what the chip sustains on random operands (power-limited)
        except Exception:
            pass
        # registers / spills / scratch of the kernels of the LM pass, from the resource remarks of the build that produced the loaded
        # library (csrc/kernel_resources.json, written by the Makefile)
        try:
            kr = json.load(open(os.path.join(ROOT, "bundle-adjustment_amd", "csrc", "kernel_resources.json")))
            # (blk_pp_gather_kernel<true, true, false, true>: the deterministic assembly (the default; pass-major since round 5), <true, false, false, false> the arrival-order form; chol_tile_kernel<1, true>: the one-kernel form the PMC counters
            # are collected on -- not kernels of the default LM pass, listed because figures of this line's family quote them)
            lm = ("chol_tile_kernel<2, false>", "chol_tile_kernel<1, true>", "potrf_chain_kernel", "blk_pp_gather_kernel<true, false, false", "blk_pp_gather_kernel<true, true, false", "blk_T_mfma_kernel", "blk_elim_kernel",
                  "blk_tfix_kernel", "blk_cc", "blk_pc_gather", "rows_kernel", "backsolve_chain8_kernel", "forwardsolve_chain8_kernel", "forwardsolve_chain_kernel", "gemm_f64_kernel<0, 1, 128, 128, 0>",
                  "symv_dd_tile_kernel", "symv_dd_reduce_kernel", "blk_backsub_kernel", "damp_and_precond_kernel")
            tab = {}
            for name, v in kr.items():
                short = name.replace("jaicov::", "").replace("void ", "")
                if any(short.startswith(k) for k in lm):
                    tab[short.split("(")[0]] = {k: v.get(k, 0) for k in ("vgprs", "agprs", "vgpr_spills", "scratch_bytes_per_lane", "lds_bytes", "occupancy")}
            out["kernel_resources"] = {"source": "csrc/kernel_resources.json (hipcc -Rpass-analysis=kernel-resource-usage of the shipped build)",
                                       "kernels_with_vgpr_spills": sorted(k for k, v in tab.items() if v["vgpr_spills"]),
                                       "lm_pass": tab,
                                       "kernels_with_scratch": sorted(k for k, v in tab.items() if v["scratch_bytes_per_lane"])}
        except Exception:
            pass
        if world >= 1:
            # what the replicated solve lets N GPUs do: T(N) = assembly / N + R(N) + the rest; R = pack + ring
            # all-reduce of the packed reduced system + unpack.  A model, printed so that a scaling run can be read against it.
            st = out["stage_ms_per_step"]
            asm = st.get("rows", 0.0) + st.get("assembly", 0.0)
            rest = out["ms_per_step"] - asm if world == 1 else None
            gb = 8e-9 * (eng.reduced_order() * (eng.reduced_order() + 1) / 2 + 2 * eng.reduced_order())
            busbw = {2: 120.0, 4: 250.0, 8: 320.0}      # GB/s, assumed: one xGMI link (153 peak) at N = 2, rings over 3 / 7 links beyond
            if rest is not None:
                out["expected_scaling"] = {"model": "T(N) = (rows + assembly) / N + 0.72 ms pack/unpack + 2 (N-1)/N x bytes / busbw(N) + everything else (replicated)",
                                           "reduce_buffer_GB": gb, "assumed_busbw_GBps": busbw,
                                           "ms_per_step": {"1": out["ms_per_step"], **{str(n): asm / n + 0.72 + 1e3 * 2 * (n - 1) / n * gb / busbw[n] + rest for n in (2, 4, 8)}},
                                           "note": "N > 1 is expected to be SLOWER than N = 1 while the solve is replicated (the shardable assembly is ~12 % of the pass)"}
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need a pass each and cannot be collected
        # inside this run): scripts/round_profile.sh regenerates profiles/pmc_traffic.json; the line says which file it quotes.
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                import hashlib
                raw = open(pmc, "rb").read()
                rec = json.loads(raw)
                stale = rec.get("kernel", "").split(" ")[0].split("<")[0] not in out["roofline"]["kernel"]
                out["roofline"]["traffic"] = None if stale else rec.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = {"file": "profiles/pmc_traffic.json", "sha1": hashlib.sha1(raw).hexdigest()[:12],
                                                     "collected": rec.get("collected"), "kernel": rec.get("kernel"),
                                                     "stale_for_this_kernel": stale}
            except Exception:
                pass
    if not a.iterations_only:
        # final pass (BA:252-280): same build, then solve with the full inverse and omega -- reported, not `value`.
        # Run twice: the first call allocates the inverse's buffers (7.8 GB) and builds the tile maps of the full order.
        for rep in range(2):
            sync()
            t1 = time.perf_counter()
            # MatrixInversion.FULL as estimate() / estimateModel() run it: all of Qxx expanded from the inverse of the EO-reduced
            # system (JAICOV_INVERT_FULL_EXPANDED); a sharded engine holds only its images' F bands and L_E^-1: they are summed over
            # the ranks by one more all-reduce (distributed.sharded_step, jaicov_neq_expansion_buffer)
            if use_dist:
                dx = distributed.sharded_step(eng, dist, torch.device("cuda", local), s2, invert=engine.INVERT_FULL_EXPANDED)
            else:
                eng.prepare_inverse(engine.INVERT_FULL_EXPANDED)
                eng.build(s2, 0.0)
                dx = eng.solve(engine.INVERT_FULL_EXPANDED)
            om = distributed.sharded_omega(eng, dist, torch.device("cuda", local), s2, dx) if use_dist else eng.omega(s2, dx)
            sync()
            if rank == 0:
                out["final_pass_ms" if rep else "final_pass_first_call_ms"] = 1e3 * (time.perf_counter() - t1)
                out["final_pass_stage_ms"] = eng.timings()
                out["sigma0_ratio"] = om / fp.degree_of_freedom / s2
        # final pass of MatrixInversion.REDUCED / PRE_ELIMINATION (BA:261-267): cofactor matrix of the border, points,
        # interior orientation and distortion only = inverse of the EO-reduced system (the second run is reported,
        # the first one allocates the inverse's buffers)
        for rep in range(0 if use_dist else 2):     # the literal route: factorisation + inverse of the unreduced system (order U)
            sync()
            t4 = time.perf_counter()
            eng.prepare_inverse(engine.INVERT_FULL)
            eng.build(s2, 0.0)
            eng.solve(engine.INVERT_FULL)
            sync()
            if rank == 0:
                out["final_pass_literal_full_ms"] = 1e3 * (time.perf_counter() - t4)
                out["final_pass_literal_full_stage_ms"] = eng.timings()
        for rep in range(0 if a.iterations_only else 2):
            sync()
            t2 = time.perf_counter()
            if use_dist:
                distributed.sharded_step(eng, dist, torch.device("cuda", local), s2, invert=engine.INVERT_REDUCED)
            else:
                eng.prepare_inverse(engine.INVERT_REDUCED)
                eng.build(s2, 0.0)
                eng.solve(engine.INVERT_REDUCED)
            sync()
            if rank == 0:
                out["final_pass_reduced_ms"] = 1e3 * (time.perf_counter() - t2)
                out["final_pass_reduced_stage_ms"] = eng.timings()
                out["final_pass_reduced_order"] = eng.cofactor_order()
    if rank == 0 and world == 1 and not a.no_dense_mode and not a.iterations_only:
        # BASELINE.json's second figure, "J'WJ MFMA-util%": the jointly dispersed image groups contracted as dense
        # A'(PA) on the fp64 matrix cores (engine option assembly_mode = 1, csrc/densemode.hip).  The default
        # structure-aware assembly computes the same N with ~2 % of the arithmetic; this is the literal form.
        de = engine.Engine(fp, device=local, assembly_mode=1)
        de.set_parameters(fp.values)
        de.accumulate(s2)
        de.set_profiling(True); de.kernel_stats(reset=True)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        de.accumulate(s2)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t3)
        ks2 = de.kernel_stats()
        de.close()
        ach = ks2["dense_flops"] / (ks2["dense_gemm_ms"] * 1e-3) / 1e12 if ks2["dense_gemm_ms"] > 0 else 0.0
        out["jtwj_dense_mode"] = {
            "kernel": "gemm_f64_kernel<KC,XC> (B = P A) + gemm_f64_kernel<XC,XC> (S = A'B, lower), batched over all images of the pass",
            "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
            "gemm_ms_per_pass": ks2["dense_gemm_ms"], "assembly_ms_per_pass": wall,
            "algorithmic_flops_per_pass": ks2["dense_flops"],
            "structure_aware_assembly_ms_per_pass": out["stage_ms_per_step"].get("assembly")}
    if rank == 0 and world == 1 and not use_dist and not a.iterations_only:
        # engine creation once the process is warm (the first engine above also paid the HIP context, the code objects and the first
        # launch of every kernel): a second engine of the same problem, created and destroyed
        t_c = time.perf_counter()
        e2 = engine.Engine(fp, device=local)
        out["create_ms"]["wall_second_engine"] = 1e3 * (time.perf_counter() - t_c)
        out["create_ms"]["second_engine"] = e2.create_timings()
        e2.close()
    if rank == 0 and world == 1 and not use_dist and not a.iterations_only and a.config == "cfg4" and not os.environ.get("JAICOV_BENCH_NO_LOCAL_SCENE"):
        # The same stage on a scene WITH spatial locality (scene.config("cfg4_local"): config 4's size and stochastic model on a block flown in
        # strips, first-seen numbering -- the shape of real photogrammetric blocks, and of the bundled example): the random-visibility
        # BASELINE scene is the kind case for the point x point gather.  Reported, not `value`.
        try:
            fpl = scene.config("cfg4_local")
            el = engine.Engine(fpl, device=local)
            el.set_parameters(fpl.values)
            acc = {}
            for it in range(6):
                el.build(fpl.sigma2apriori, 0.0)
                dxl = el.solve(False)
                if it >= 2:
                    for k, v in el.timings().items():
                        acc[k] = acc.get(k, 0.0) + v / 4.0
            out["assembly_ms_local_scene"] = acc.get("assembly")
            out["local_scene"] = {"workload": f"cfg4_local: {fpl.n_images} images x {fpl.n_points} points in strips, {fpl.n_image_points} image points, U={fpl.n_unknowns}",
                                  "stage_ms_per_step": acc, "gather_strip_columns": el.kernel_stats().get("gather_strip_columns")}
            el.close()
            del fpl
        except Exception as ex:      # the line must not die on the extra scene
            out["assembly_ms_local_scene"] = None
            out["local_scene"] = {"error": str(ex)}
    if rank == 0:
        if not a.no_cpu_baseline and not a.iterations_only:
            out["cpu_baseline"] = cpu_baseline(fp, eng, s2)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
