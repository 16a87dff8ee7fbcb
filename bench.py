#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json on MI355X.

  metric : graph-optimize LM iterations/sec on the synthetic 5000-pose / 1000-landmark graph with
           loop closures (BASELINE.json configs[2], the configuration the metric is quoted on).
  step   : one Levenberg-Marquardt iteration (Jacobian build + linear solve(s) + update + chi2 +
           accept/reject) over one device-resident batch of `--batch` independent graphs.
  value  : graphs-per-GPU x n_gpus x K / seconds  (graph-iterations per second, whole job),
           inputs resident in HBM before the timed region.

One process per GPU; for N>1 launch with torch.distributed.run (RCCL): independent graphs shard
across ranks with no data-path collective ("scaling": "weak"); the barrier + max-over-ranks timing
uses torch.distributed.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SSLAM_BENCH_BATCH", "512")),
                    help="independent graphs resident per GPU (512 x ~10 MB of H + 7.5 MB of L each: far beyond the 256 MiB MALL)")
    ap.add_argument("--poses", type=int, default=5000)
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--distinct", type=int, default=4, help="distinct seeds generated per rank (tiled to --batch)")
    ap.add_argument("--solver", type=int, default=-1, help="-1 library default, 0 PCG, 1 sparse Cholesky")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frontend", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-graph latency section (used under rocprofv3 --pmc)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch  # loaded first so that libamdhip64 is shared with the product library
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    import numpy as np
    from semantic_slam_amd import GraphSLAM, GraphBatch, load_library
    from semantic_slam_amd import distributed as D
    from semantic_slam_amd.synth import make_graph
    from oracle.oracle import GraphProblem  # only for problem packing + the cpu_baseline leg

    lib = load_library()
    if lib.sslam_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    dev = local_rank if world > 1 else 0

    # ---- synthetic workload, resident in HBM before timing -------------------------------------
    t_setup = time.time()
    tmpdir = tempfile.mkdtemp(prefix="sslam_bench_")
    paths, problems = [], []
    for d in range(max(1, min(args.distinct, args.batch))):
        g = make_graph(args.poses, args.landmarks, seed=1000 * rank + d)
        gp = GraphProblem.from_synth(g)
        problems.append(gp)
        G0 = GraphSLAM.from_problem(gp, device=dev)
        p = os.path.join(tmpdir, f"g{d}.g2o")
        G0.save(p)
        paths.append(p)
        del G0
    graphs = []
    for k in range(args.batch):
        G = GraphSLAM(False, dev)
        G.load(paths[k % len(paths)])
        if args.solver >= 0:
            G.set_option("solver", args.solver)
        graphs.append(G)
    batch = GraphBatch(graphs)
    batch.upload()
    setup_s = time.time() - t_setup

    def sync_all():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # ---- warmup (untimed), then reset to the initial estimates -----------------------------------
    if args.warmup > 0:
        batch.optimize(args.warmup)
    batch.upload()
    batch.set_profiling(True)
    sync_all()
    t0 = time.perf_counter()
    stats = batch.optimize(args.steps)       # blocks until the stream is idle (hipStreamSynchronize)
    sync_all()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, device="cuda" if dist is not None else None)
    iters_done = [s.iterations for s in stats]
    assert min(iters_done) == args.steps, f"LM terminated early: {min(iters_done)} < {args.steps}"
    # whole-job value: graph-iterations of ALL ranks / max-over-ranks time (replicas: no data-path collective)
    value = D.aggregate_throughput(float(args.batch * args.steps), dt, device="cuda" if dist is not None else None)

    # ---- kernel times (hipEvents on the batch's stream, inside the timed region) -------------------
    names = ["linearize", "chi2", "spmv", "pcg_update", "precond", "oplus", "factor", "solve"]
    ktimes = {n: batch.kernel_time(n) for n in names}
    batch.set_profiling(False)
    dominant = max(ktimes, key=lambda n: ktimes[n][0])
    Eo = int((problems[0].etype == 0).sum()); El = problems[0].ne - Eo
    jac_bytes = batch.linearize_bytes()
    lin_ms, lin_n = ktimes["linearize"]
    jac_ms = lin_ms / max(lin_n, 1)
    jac_gbs = jac_bytes / (jac_ms * 1e-3) / 1e9 if jac_ms > 0 else 0.0
    roof_jac = {"bound": "hbm", "kernel": "jacobian_build", "achieved": round(jac_gbs, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(jac_gbs / 8000.0, 4), "traffic": None, "bytes_per_launch": jac_bytes,
                "ms_per_launch": round(jac_ms, 5), "launches": lin_n}
    # measured HBM traffic of the Jacobian build (PMC FETCH_SIZE / WRITE_SIZE, collected with rocprofv3 in separate
    # passes, FETCH x2 per MI355X_MICROARCH.md; committed under profiles/).  Only valid for the profiled batch size.
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_jacobian_build.json")))
        if pmc.get("algorithmic_bytes") == jac_bytes:
            roof_jac["traffic"] = int(pmc["hbm_bytes_fetch_x2"])
            roof_jac["traffic_raw_counters"] = int(pmc["hbm_bytes_raw"])
    except (OSError, ValueError):
        pass
    roofline = roof_jac
    if dominant == "factor" and ktimes["factor"][1] > 0:
        fbytes = batch.info("factor_bytes")
        ms = ktimes["factor"][0] / ktimes["factor"][1]
        gbs = fbytes / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "block_cholesky_factor (k_chol_level launches per elimination-tree level + k_chol_tail "
                                              "of one numeric factorisation)",
                    "achieved": round(gbs, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                    "bytes_per_launch": int(fbytes), "ms_per_launch": round(ms, 4), "launches": ktimes["factor"][1],
                    "levels": int(batch.info("factor_levels")), "factor_doubles": int(batch.info("factor_lnz")),
                    "note": "one 'launch' = the dependent chain of kernels of one factorisation; latency / instruction-issue bound "
                            "(6x6 blocks), algorithmic bytes = read H,b + write L,y once"}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_factor.json")))
            if pmc.get("algorithmic_bytes") == int(fbytes):
                roofline["traffic"] = int(pmc["hbm_bytes_fetch_x2"])
                roofline["traffic_raw_counters"] = int(pmc["hbm_bytes_raw"])
        except (OSError, ValueError):
            pass
    if dominant == "spmv" and ktimes["spmv"][1] > 0:
        # H bytes + 3 vectors x 8*dim per block-SpMV
        Np, Nl = args.poses - 1, args.landmarks
        h_bytes = 8 * (36 * Np + 9 * Nl + 36 * Eo + 18 * El)
        spmv_bytes = args.batch * (h_bytes + 3 * 8 * (6 * Np + 3 * Nl))
        ms = ktimes["spmv"][0] / ktimes["spmv"][1]
        gbs = spmv_bytes / (ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "pcg_block_spmv", "achieved": round(gbs, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(gbs / 8000.0, 4), "traffic": None, "bytes_per_launch": spmv_bytes,
                    "ms_per_launch": round(ms, 5), "launches": ktimes["spmv"][1],
                    "note": "bytes assume every still-active graph; converged graphs early-exit"}

    out = {
        "metric": "graph-optimize LM iters/sec (5k poses, 1k landmarks)",
        "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": f"synthetic ({len(paths)} distinct seeds per GPU tiled to the batch)",
        "config": {"workload": f"synthetic {args.poses}-pose / {args.landmarks}-landmark graph with loop closures "
                               f"(BASELINE.json configs[2]), batch of {args.batch} independent graphs per GPU",
                   "graphs_per_gpu": args.batch, "se3_edges": Eo, "landmark_edges": El,
                   "solver": int(args.solver), "parallelism": f"replicas x{world}"},
        "keyframes_landmarks_per_sec": round(value * (args.poses + args.landmarks), 1),
        "chi2_after": stats[0].chi2_after, "lm_trials": stats[0].trials, "solver_iterations": stats[0].solver_iterations,
        "roofline": roofline, "roofline_jacobian_build": roof_jac,
        "kernel_ms": {n: [round(v[0], 3), v[1]] for n, v in ktimes.items() if v[1]},
        "setup_seconds": round(setup_s, 1),
    }

    if rank == 0:
        if not args.no_single:
            # ---- single-graph latency (same graph, batch of one) -----------------------------------
            one = GraphSLAM(False, dev); one.load(paths[0])
            if args.solver >= 0:
                one.set_option("solver", args.solver)
            b1 = GraphBatch([one]); b1.upload(); b1.optimize(1); b1.upload()
            t1 = time.perf_counter(); s1 = b1.optimize(args.steps); d1 = time.perf_counter() - t1
            out["single_graph"] = {"iters_per_sec": round(args.steps / d1, 2), "ms_per_iter": round(1e3 * d1 / args.steps, 3),
                                   "regime": "latency-bound (working set < L2/MALL)", "chi2_after": s1[0].chi2_after}
            del b1, one
        if not args.no_cpu_baseline and world == 1:
            # ---- CPU baseline: the oracle (restated reference algorithm), 1 core, bounded sample ----
            reps, cpu_t, budget = 0, 0.0, 15.0
            while cpu_t < budget:
                gp = problems[reps % len(problems)].copy()
                st = gp.optimize(args.steps)
                cpu_t += st.seconds
                reps += 1
            out["cpu_baseline"] = {"value": round(reps * args.steps / cpu_t, 3), "unit": "iters/s", "cores": 1, "kind": "port",
                                   "sample": f"{reps} runs of {args.steps} LM iterations on the same {args.poses}/{args.landmarks} graphs "
                                             f"(oracle/oracle_graph.c: LM + min-degree sparse Cholesky, gcc -O3 -march=native)",
                                   "host_cores_available": os.cpu_count()}
            # the same port on many host cores at once (independent graphs, one per thread; the C call releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            ncore = min(len(os.sched_getaffinity(0)), 64)
            work = [problems[k % len(problems)].copy() for k in range(ncore)]
            deadline = time.perf_counter() + 6.0
            def run(gp):
                n = 0
                while n == 0 or time.perf_counter() < deadline:
                    gp.copy().optimize(args.steps)
                    n += 1
                return n
            tA = time.perf_counter()
            with ThreadPoolExecutor(max_workers=ncore) as ex:
                runs = sum(ex.map(run, work))
            dA = time.perf_counter() - tA
            out["cpu_baseline_multicore"] = {"value": round(runs * args.steps / dA, 3), "unit": "iters/s", "cores": ncore, "kind": "port",
                                             "sample": f"{runs} runs of {args.steps} LM iterations over {ncore} threads in {dA:.1f} s, "
                                                       f"independent graphs (same oracle)"}
        if not args.no_frontend:
            try:
                from semantic_slam_amd import segmentation
                out["frontend"] = segmentation.bench_frontend(dev, cpu_baseline=(not args.no_cpu_baseline and world == 1))
            except ImportError:
                pass
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
