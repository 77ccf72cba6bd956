#!/usr/bin/env python3
"""bench.py -- headline benchmark of the bvector/aggregator hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  aggregator::pipeline<agg_opt_only_counts> + combine_and_sub(pipe)
  (src/bmaggregator.h:1292-1399) = fused 256-way AND + COUNT over 256 bit-vectors of
  1e9 bits each, data set A of SURVEY.md section 8(d): v = common OR noise_v, both
  Bernoulli 10 % (mirrors GenerateTestCollection, tests/perf/perf.cpp:234-267), so no
  early exit is possible and every operand block must be read.

A "step" = one pass of the hot path over the resident vectors (one kernel launch,
plus for N > 1 one RCCL all-reduce of the 8-byte popcount).  Inputs are generated
on the device and are resident in HBM before the timed region starts.

N > 1: document-sharded index ("weak" scaling): every rank owns 256 x 1e9-bit
shards of a N x 1e9-bit collection; block columns are independent
(src/bmaggregator.h:1184-1218) so no bit data crosses xGMI.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0xB17A61C
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)


def cpu_baseline(nvec: int, dq: int, sample_blocks: int, gpu_count_on_sample: int | None):
    """Time the reference (oracle/_ref, kind "reference") or the C port on a bounded sample of
    the same workload: all nvec vectors restricted to their first `sample_blocks` blocks."""
    import numpy as np
    import oracle
    P = oracle.port()
    kind, orc = "port", P
    try:
        flags = open("/proc/cpuinfo").read()
        if oracle.have_reference("avx2") and " avx2 " in flags and " bmi2 " in flags:
            orc, kind = oracle.reference("avx2"), "reference"
        elif oracle.have_reference("scalar"):
            orc, kind = oracle.reference("scalar"), "reference"
    except Exception:
        pass
    nbits = sample_blocks * 65536
    t0 = time.perf_counter()
    vecs = []
    for v in range(nvec):
        w = P.gen_words(SEED, v, dq, 1_000_000_000, with_common=True, word_off=0, nwords=sample_blocks * 2048)
        vecs.append(orc.import_words(w, True, nbits))
    t_gen = time.perf_counter() - t0
    groups = [(vecs, [])]
    best, reps, cnt = None, 0, None
    t_start = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_start < 6.0 and reps < 50):
        t0 = time.perf_counter()
        cnt = orc.pipeline_counts(groups)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    bits = nvec * nbits
    out = {"value": round(bits / best / 1e9, 3), "unit": "Gbit/s", "cores": 1, "kind": kind,
           "impl": orc.name,
           "sample": f"{nvec} vectors x first {sample_blocks} blocks ({bits / 8e9:.2f} GB operand bytes), "
                     f"counts-only pipeline, best of {reps}; input build {t_gen:.1f}s not timed",
           "count": int(cnt[0]), "ms": round(best * 1e3, 3),
           "host_cores_available": os.cpu_count()}
    if gpu_count_on_sample is not None:
        out["matches_gpu"] = bool(int(cnt[0]) == int(gpu_count_on_sample))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nvec", type=int, default=256)
    ap.add_argument("--nbits", type=int, default=1_000_000_000)
    ap.add_argument("--density-q16", type=int, default=6554)      # 10 %
    ap.add_argument("--independent", action="store_true", help="data set B (no common part; early exit)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=512)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set): always go through RCCL, also for a 1-rank job, so the
    # collective path is exercised wherever the launcher is used
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"

    import bitmagic_amd as bm
    # one explicit (non-null) stream shared by the HIP kernels, torch and RCCL, so that the
    # all-reduce is stream-ordered after the count kernel and HIP events see everything
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    ctx = bm.context(local_rank, tstream.cuda_stream)

    # ---- build the resident collection (not timed) ----------------------------------
    t0 = time.perf_counter()
    base_id = rank * args.nvec * 4                       # distinct shard content per rank
    vecs = [bm.bvector.generate(ctx, SEED, base_id + v if world > 1 else v, args.density_q16, args.nbits,
                                with_common=not args.independent) for v in range(args.nvec)]
    agg = bm.aggregator(ctx)
    pipe = bm.aggregator.pipeline(ctx)
    ag = pipe.add()
    for v in vecs:
        ag.add(v, 0)
    pipe.complete()
    ctx.synchronize()
    t_build = time.perf_counter() - t0
    nblocks = vecs[0].info()["nblocks"]
    op_bytes = pipe.operand_bytes()                      # algorithmic bytes per launch
    stat = vecs[0].calc_stat()
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")

    def step():
        agg.run_counts_dev(pipe, counts.data_ptr())
        if use_dist:
            dist.all_reduce(counts)                      # RCCL: 8 bytes per arg-group, same stream as the kernel

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.timer_start()                                    # HIP events on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop_ms()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total_count = int(counts.item())

    if rank == 0:
        bits_per_step = world * args.nvec * args.nbits
        value = bits_per_step * args.steps / dt / 1e9
        k_ms = ev_ms / args.steps
        achieved = op_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"agg_and_count_{args.nvec}x{args.nbits}":
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                pass
        res = {
            "metric": "Gbits/s + % HBM roofline, 256-way fused AND+COUNT on 1B-bit vectors",
            "value": round(value, 2), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"aggregator pipeline combine_and_sub counts-only: {args.nvec}-way AND+COUNT, "
                                   f"{args.nvec} x {args.nbits}-bit vectors per GPU, "
                                   + ("data set B (independent 10%)" if args.independent else
                                      "data set A (common 10% OR noise 10%, no early exit)"),
                       "baseline_config": "configs[2]", "vectors": args.nvec, "bits_per_vector": args.nbits,
                       "density_q16": args.density_q16, "blocks_per_vector": nblocks,
                       "block_types_vec0": stat, "sharding": f"block/document range x{world}",
                       "result_count": total_count, "build_seconds": round(t_build, 2),
                       "hbm_resident_bytes": ctx.mem_used()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_pipe_counts", "algorithmic_bytes_per_launch": op_bytes,
                         "avg_launch_ms": round(k_ms, 4),
                         "timing": "hipEvent pair on the launch stream around the timed region / steps"},
        }
        if world == 1 and not args.no_cpu:
            sb = min(args.cpu_sample_blocks, nblocks)
            gpu_sample = int(agg._run_pipeline(pipe, 0, sb)[0])
            try:
                res["cpu_baseline"] = cpu_baseline(args.nvec, args.density_q16, sb,
                                                   gpu_sample if not args.independent and args.nbits == 1_000_000_000 else None)
            except Exception as e:  # the baseline is a reported number, never the product path
                res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
