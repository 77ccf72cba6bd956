#!/usr/bin/env python3
"""bench.py -- benchmarks of the bvector/aggregator hot path on MI355X.  Prints ONE JSON line on rank 0.

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  aggregator::pipeline<agg_opt_only_counts> + combine_and_sub(pipe)
  (src/bmaggregator.h:1292-1399) = fused 256-way AND + COUNT over 256 bit-vectors of
  1e9 bits each, data set A of SURVEY.md section 8(d): v = common OR noise_v, both
  Bernoulli 10 % (mirrors GenerateTestCollection, tests/perf/perf.cpp:234-267), so no
  early exit is possible and every operand block must be read.

A "step" = one pass of the hot path over the resident vectors (one kernel launch, plus for N > 1 one RCCL
all-reduce of the 8-byte popcount).  Inputs are generated on the device and are resident in HBM before the
timed region starts.

N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), --scaling:
  strong (default for N > 1): the collection is FIXED at 256 x 1e9 bits; rank r holds only the block range
      shard_range(15259, r, N) of every vector (bmx_vec_generate_shard), runs the same fused kernel over its
      shard and the only exchange is one RCCL all-reduce of the popcount -- the block-range sharding north_star
      names (column independence: src/bmaggregator.h:1184-1218).  value = 256e9 bits / step time.
  weak: every rank owns its own 256 x 1e9-bit collection (a document-sharded index).  Reported as the second
      figure "weak_scaling" of the strong line unless --no-weak.

--config 1|3|4 run the other BASELINE configs through the same JSON schema (roofline + cpu_baseline):
  1 pairwise count_and/or/xor/sub + materialised ops on 1e9-bit vectors, rotating over distinct vector pairs so
    that nothing is served from the 256 MB Infinity Cache;  3 rank/select, 10 M queries on a 4e9-bit vector;
  4 combine_or over 4096 x 4e9-bit sparse vectors, block-range sharded over the ranks.
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
NBITS_1G = 1_000_000_000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
METRIC = "Gbits/s + % HBM roofline, 256-way fused AND+COUNT on 1B-bit vectors"


# ----------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself (oracle/_ref, kind "reference") or the C port, timed on the GPU box's
# host cores.  Only this leg of bench.py touches oracle/ -- as the thing measured NEXT to the GPU, never
# as the product path.
# ----------------------------------------------------------------------------------------------------
def _pick_oracle():
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
    return P, orc, kind


def cpu_baseline_1core(nvec: int, dq: int, nbits: int, sample_blocks: int, gpu_count_on_sample):
    """all nvec vectors restricted to their first `sample_blocks` blocks, one core (the reference aggregator is
    single-threaded by construction: this is the primary CPU number)"""
    P, orc, kind = _pick_oracle()
    sbits = min(nbits, sample_blocks * 65536)
    t0 = time.perf_counter()
    vecs = []
    for v in range(nvec):
        w = P.gen_words(SEED, v, dq, nbits, with_common=True, word_off=0, nwords=sample_blocks * 2048)
        vecs.append(orc.import_words(w, True, sbits))
    t_gen = time.perf_counter() - t0
    groups = [(vecs, [])]
    best, reps, cnt = None, 0, None
    t_start = time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_start < 5.0 and reps < 40):
        t0 = time.perf_counter()
        cnt = orc.pipeline_counts(groups)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
    bits = nvec * sample_blocks * 65536
    out = {"value": round(bits / best / 1e9, 3), "unit": "Gbit/s", "cores": 1, "kind": kind, "impl": orc.name,
           "sample": f"{nvec} vectors x first {sample_blocks} blocks ({bits / 8e9:.2f} GB operand bytes), "
                     f"counts-only pipeline, best of {reps}; input build {t_gen:.1f}s not timed",
           "count": int(cnt[0]), "ms": round(best * 1e3, 3), "host_cores_available": os.cpu_count()}
    if gpu_count_on_sample is not None:
        out["matches_gpu"] = bool(int(cnt[0]) == int(gpu_count_on_sample))
    return out


def _cpu_worker_main(argv):
    """`bench.py --cpu-worker lo hi nvec dq nbits reps`: one independent replica (SURVEY section 8d "secondary = one
    replica per core"): block columns [lo, hi) of every vector -- block ranges are independent, so the per-shard
    counts add up exactly.  Builds its inputs, prints READY, waits for a line on stdin, runs `reps` passes."""
    lo, hi, nvec, dq, nbits, reps = (int(x) for x in argv)
    P, orc, kind = _pick_oracle()
    sbits = max(min(nbits, hi * 65536) - lo * 65536, 0)
    vecs = []
    for v in range(nvec):
        w = P.gen_words(SEED, v, dq, nbits, with_common=True, word_off=lo * 2048, nwords=(hi - lo) * 2048)
        vecs.append(orc.import_words(w, True, sbits))
    groups = [(vecs, [])]
    sys.stdout.write("READY\n"); sys.stdout.flush()
    sys.stdin.readline()
    spans, cnt = [], 0
    for _ in range(reps):
        t0 = time.perf_counter()                         # CLOCK_MONOTONIC: comparable across processes
        cnt = int(orc.pipeline_counts(groups)[0]) if hi > lo else 0
        spans.append((t0, time.perf_counter()))
    sys.stdout.write(json.dumps({"count": cnt, "spans": spans, "kind": kind, "impl": orc.name}) + "\n")
    sys.stdout.flush()


def cpu_baseline_allcores(nvec: int, dq: int, nbits: int, cores: int, reps: int = 3):
    """the FULL workload on all host cores: every block column of all nvec vectors, block ranges fanned over
    `cores` worker processes (plain interpreters that never load HIP).  Returns the exact full-size count (pins
    the GPU's headline result against the reference) and the aggregate rate = all operand bits x reps /
    (last end - first start) of the passes, all replicas running at once."""
    import subprocess
    from bitmagic_amd.sharding import shard_range
    nblocks = (nbits + 65535) // 65536
    cores = max(1, min(cores, nblocks))
    t0 = time.perf_counter()
    procs = []
    for w in range(cores):
        lo, hi = shard_range(nblocks, w, cores)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(lo), str(hi), str(nvec),
                                       str(dq), str(nbits), str(reps)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      text=True, cwd=ROOT))
    try:
        for p in procs:
            line = p.stdout.readline()
            if line.strip() != "READY":
                raise RuntimeError(f"cpu worker failed to start: {line!r}")
        t_ready = time.perf_counter() - t0
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
    full = sum(r["count"] for r in res)
    first = min(r["spans"][0][0] for r in res); last = max(r["spans"][-1][1] for r in res)
    bits = nvec * nblocks * 65536
    return {"full_count": int(full), "allcores_gbit_s": round(bits * reps / (last - first) / 1e9, 1), "cores_used": cores,
            "allcores_ms_per_pass": round((last - first) / reps * 1e3, 2), "allcores_kind": res[0]["kind"],
            "allcores_impl": res[0]["impl"],
            "allcores_sample": f"the whole workload: {nvec} vectors x {nblocks} blocks, one block-range replica per core, "
                               f"{reps} back-to-back passes with all replicas running (last end - first start); "
                               f"input build {t_ready:.1f}s not timed"}


# ----------------------------------------------------------------------------------------------------
def setup_dist(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # launched by torch.distributed.run (RANK set): always go through RCCL, also for a 1-rank job, so the
    # collective path is exercised wherever the launcher is used
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    # test hook (tools/gpu_runs: exercising the N > 1 code path on a ONE-GPU box): every rank on device 0, gloo instead of RCCL
    # (RCCL refuses two ranks on one device).  Timings of such a run mean nothing; it checks sharding, gathers and the JSON line.
    one_dev = os.environ.get("BMX_BENCH_TEST_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if use_dist and one_dev:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up: keep stdout for the ONE JSON line
        # (the banner goes to stderr) by creating the communicator -- init + a first all-reduce -- under a redirect
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            t = torch.zeros(1, dtype=torch.int64, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    return world, rank, local_rank, use_dist


def timed_region(step, steps, warmup, ctx, use_dist):
    """W warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; wall time = max over ranks.
    Also returns the HIP-event time of the same region on the launch stream."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.timer_start()                                    # HIP events on the launch stream
    t0 = time.perf_counter()
    for _ in range(steps):
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
    return float(tmax.item()), ev_ms


def event_avg_ms(fn, reps, ctx):
    """average device time of fn() over reps back-to-back calls (HIP events on the launch stream)"""
    fn()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


def gather_floats(x: float, use_dist, world):
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if os.environ.get("BMX_BENCH_TEST_ONE_DEVICE") == "1" else "cuda")
    if not use_dist:
        return [float(x)]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def traffic_note(workload: str):
    """HBM bytes per launch from the PMC passes of the last profiling run (rocprofv3 --pmc cannot run inside
    this process): a constant read from profiles/traffic_latest.json, labelled with its source"""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(tpath))
        if tj.get("workload") == workload:
            return tj.get("hbm_bytes_per_launch"), f"{tj.get('source', 'profiles/traffic_latest.json')} (PMC passes of a separate rocprofv3 run, not measured in this run)"
    except Exception:
        pass
    return None, None


# ----------------------------------------------------------------------------------------------------
# configs[2]: the headline
# ----------------------------------------------------------------------------------------------------
def run_headline(args):
    # CPU legs first (rank 0 of a 1-GPU run only), before torch / HIP are loaded into this process
    cpu = None
    nblocks_full = (args.nbits + 65535) // 65536
    sb = min(args.cpu_sample_blocks, nblocks_full)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu:
        try:
            cpu = cpu_baseline_1core(args.nvec, args.density_q16, args.nbits, sb, None)
            if not args.no_allcores and not args.independent:
                ncores = args.cpu_cores or len(os.sched_getaffinity(0))
                cpu.update(cpu_baseline_allcores(args.nvec, args.density_q16, args.nbits, ncores))
        except Exception as e:  # the baseline is a reported number, never the product path
            cpu = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    import torch
    import torch.distributed as dist
    import bitmagic_amd as bm
    world, rank, local_rank, use_dist = setup_dist(args)
    scaling = args.scaling if args.scaling != "auto" else "strong"       # N = 1: both modes are the same run
    tstream = torch.cuda.Stream()                        # one explicit stream shared by the HIP kernels, torch and RCCL
    torch.cuda.set_stream(tstream)
    ctx = bm.context(local_rank, tstream.cuda_stream)
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")
    agg = bm.aggregator(ctx)

    def run_mode(mode):
        """build the rank's resident collection for `mode`, time it, free it"""
        t0 = time.perf_counter()
        if mode == "strong":
            lo, hi = bm.shard_range(nblocks_full, rank, world)
            vecs = [bm.bvector.generate(ctx, SEED, v, args.density_q16, args.nbits, with_common=not args.independent,
                                        block_range=(lo, hi) if world > 1 else None) for v in range(args.nvec)]
        else:
            base_id = rank * args.nvec * 4               # distinct content per rank
            lo, hi = 0, nblocks_full
            vecs = [bm.bvector.generate(ctx, SEED, base_id + v if world > 1 else v, args.density_q16, args.nbits,
                                        with_common=not args.independent) for v in range(args.nvec)]
        pipe = bm.aggregator.pipeline(ctx)
        ag = pipe.add()
        for v in vecs:
            ag.add(v, 0)
        pipe.complete()
        ctx.synchronize()
        t_build = time.perf_counter() - t0
        op_bytes = pipe.operand_bytes()                  # algorithmic bytes of this rank's launch

        def kernel():
            agg.run_counts_dev(pipe, counts.data_ptr())

        def step():
            kernel()
            if use_dist:
                dist.all_reduce(counts)                  # RCCL: 8 bytes per arg-group, same stream as the kernel
        dt, ev_ms = timed_region(step, args.steps, args.warmup, ctx, use_dist)
        total = int(counts.item())
        k_ms = event_avg_ms(kernel, max(5, min(args.steps, 20)), ctx)      # the kernel alone
        ar_us = None
        if use_dist:
            ar_us = event_avg_ms(lambda: dist.all_reduce(counts), 20, ctx) * 1e3
        r = {"dt": dt, "ev_ms": ev_ms, "count": total, "op_bytes": op_bytes, "k_ms": k_ms, "ar_us": ar_us,
             "build_s": t_build, "blocks": hi - lo, "plan": pipe.describe(), "nlaunch": pipe.launches(), "stat0": vecs[0].calc_stat(), "mem": ctx.mem_used(),
             "pipe": pipe, "vecs": vecs}
        return r

    main = run_mode(scaling)
    k_all = gather_floats(main["k_ms"], use_dist, world)
    bytes_all = gather_floats(float(main["op_bytes"]), use_dist, world)
    # 1-GPU shard efficiency: the 1/8 block range of the same collection, time x 8 vs the full time
    shard_eff = None
    if world == 1 and scaling == "strong" and not args.no_shard_probe and not args.independent:
        lo, hi = bm.shard_range(nblocks_full, 0, 8)
        pipe = main["pipe"]
        t_sh = event_avg_ms(lambda: agg.run_counts_dev(pipe, counts.data_ptr(), lo, hi), 20, ctx)
        b_sh = pipe.operand_bytes(lo, hi)
        shard_eff = {"blocks": hi - lo, "kernel": pipe.describe(lo, hi), "ms": round(t_sh, 4), "GBps": round(b_sh / t_sh / 1e6, 1),
                     "rate_vs_full": round((b_sh / t_sh) / (main["op_bytes"] / main["k_ms"]), 4),
                     "note": "block columns [0, 1907) of the resident collection = what one of 8 GPUs runs under --scaling strong"}
    gpu_sample = None
    if world == 1 and not args.no_cpu:
        gpu_sample = int(agg._run_pipeline(main["pipe"], 0, sb)[0])
    weak = None
    del main["pipe"], main["vecs"]
    if world > 1 and scaling == "strong" and not args.no_weak:
        ctx.trim()
        w = run_mode("weak")
        weak = {"value": round(world * args.nvec * args.nbits * args.steps / w["dt"] / 1e9, 2), "unit": "Gbit/s",
                "ms_per_step": round(w["dt"] / args.steps * 1e3, 4), "result_count": w["count"],
                "note": "every rank owns its own 256 x 1e9-bit collection (document-sharded index)"}
        del w["pipe"], w["vecs"]
    if rank == 0:
        bits_per_step = args.nvec * args.nbits * (world if scaling == "weak" else 1)
        value = bits_per_step * args.steps / main["dt"] / 1e9
        achieved = main["op_bytes"] / (main["k_ms"] * 1e-3) / 1e9
        # the PMC figure belongs to the headline data set on one GPU (density 10 %, data set A, 6 launch windows): any other
        # density / data set / shard runs another kernel or another launch plan and carries no traffic figure
        headline = args.density_q16 == 6554 and not args.independent and world == 1
        traffic, tsrc = traffic_note(f"agg_and_count_{args.nvec}x{args.nbits}") if headline else (None, None)
        if traffic is not None and main["nlaunch"] != 6: traffic, tsrc = None, None
        res = {
            "metric": METRIC, "value": round(value, 2), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(main["dt"] / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"aggregator pipeline combine_and_sub counts-only: {args.nvec}-way AND+COUNT, "
                                   f"{args.nvec} x {args.nbits}-bit vectors "
                                   + ("per GPU, " if scaling == "weak" and world > 1 else "in total, ")
                                   + ("data set B (independent 10%)" if args.independent else
                                      "data set A (common 10% OR noise 10%, no early exit)"),
                       "baseline_config": "configs[2]", "vectors": args.nvec, "bits_per_vector": args.nbits,
                       "density_q16": args.density_q16, "blocks_per_vector": nblocks_full,
                       "block_types_vec0": main["stat0"],
                       "sharding": (f"block-range shards: rank r holds blocks shard_range({nblocks_full}, r, {world}) of every vector"
                                    if scaling == "strong" else f"document shards x{world}"),
                       "blocks_per_rank": main["blocks"], "result_count": main["count"],
                       "build_seconds": round(main["build_s"], 2), "hbm_resident_bytes": main["mem"]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                         "kernel": main["plan"], "launches_per_step": main["nlaunch"],
                         "algorithmic_bytes_per_launch": main["op_bytes"] // main["nlaunch"],
                         "avg_launch_ms": round(main["k_ms"] / main["nlaunch"], 4),
                         "algorithmic_bytes_per_step": main["op_bytes"], "kernel_ms_per_step": round(main["k_ms"], 4),
                         "scope": "rank 0's GPU",
                         "timing": "hipEvent pair on the launch stream around back-to-back passes of the kernel alone (a pass = "
                                   "launches_per_step launches over equal column windows); avg_launch_ms = pass time / launches, "
                                   "inter-launch gaps included"},
            "per_rank": {"kernel_ms": [round(x, 4) for x in k_all],
                         "GBps": [round(b / (k * 1e-3) / 1e9, 1) for b, k in zip(bytes_all, k_all)],
                         "allreduce_us": None if main["ar_us"] is None else round(main["ar_us"], 1),
                         "step_event_ms": round(main["ev_ms"] / args.steps, 4)},
        }
        if args.independent:
            # data set B: the AND dies after a few operands and the kernel stops reading a column there (the reference's digest
            # exit); full-read bytes over an early-exit time is not a bandwidth: report time and rate only
            res["roofline"].update({"achieved": None, "frac": None, "per_rank_note": "early exit: bytes actually read are not the full operand bytes",
                                    "note": "early-exit data set: time and Gbit/s of LOGICAL operand bits only; no bandwidth figure"})
            res["per_rank"]["GBps"] = None
        if shard_eff:
            res["shard_1of8_on_one_gpu"] = shard_eff
        if weak:
            res["weak_scaling"] = weak
        if cpu is not None:
            if cpu.get("value") is not None and not args.independent:
                cpu["matches_gpu"] = bool(cpu["count"] == gpu_sample)
                if "full_count" in cpu:
                    cpu["matches_gpu_full"] = bool(cpu["full_count"] == main["count"])
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------
# configs[1]: pairwise ops on 1e9-bit vectors, HBM-cold
# ----------------------------------------------------------------------------------------------------
def run_pairwise(args):
    import ctypes as C
    import torch
    import bitmagic_amd as bm
    from bitmagic_amd import _ffi
    world, rank, local_rank, use_dist = setup_dist(args)
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = bm.context(local_rank, s.cuda_stream)
    L = _ffi.lib()
    dq = args.density_q16
    npairs = args.pairs                                  # distinct pairs: npairs x 250 MB >> 256 MB Infinity Cache
    va = [bm.bvector.generate(ctx, SEED, 2 * i + 1, dq, args.nbits) for i in range(npairs)]
    vb = [bm.bvector.generate(ctx, SEED, 2 * i + 2, dq, args.nbits) for i in range(npairs)]
    pair_bytes = []
    for a, b in zip(va, vb):
        ia, ib = a.info(), b.info()
        pair_bytes.append((ia["counts"][2] + ib["counts"][2]) * 8192 + 2 * (ia["gap_words"] + ib["gap_words"]))
    dcnt = torch.zeros(npairs, dtype=torch.int64, device="cuda")
    per_op = {}
    for op, name in enumerate(["and", "or", "xor", "sub"]):
        def sweep(op=op):
            for i in range(npairs):
                L.bmx_count_op2_dev(ctx._h, op, va[i]._h, vb[i]._h, C.c_void_p(dcnt.data_ptr() + 8 * i))
        ms = event_avg_ms(sweep, 5, ctx) / npairs
        per_op[name] = {"kernel_ms": round(ms, 4), "GBps": round(sum(pair_bytes) / npairs / ms / 1e6, 1)}
        # materialised result (opt_none), whole host call incl. result creation
        keep = []
        def mat(op=op):
            for i in range(npairs):
                keep.append(bm.bvector._op2(op, va[i], vb[i], bm.opt_none))
                if len(keep) > 2: keep.pop(0)
        mat(); ctx.synchronize()
        t0 = time.perf_counter(); mat(); ctx.synchronize(); host_ms = (time.perf_counter() - t0) * 1e3 / npairs
        out_blocks = keep[-1].info()["counts"][2]
        per_op[name]["materialised_host_call_ms"] = round(host_ms, 4)
        per_op[name]["materialised_GBps"] = round((pair_bytes[-1] + out_blocks * 8192) / host_ms / 1e6, 1)
        keep.clear()
    # timed region per the contract: a "step" = count_and over every pair (npairs launches)
    def step():
        for i in range(npairs):
            L.bmx_count_op2_dev(ctx._h, 0, va[i]._h, vb[i]._h, C.c_void_p(dcnt.data_ptr() + 8 * i))
    dt, ev_ms = timed_region(step, args.steps, args.warmup, ctx, use_dist)
    torch.cuda.synchronize()
    counts = dcnt.cpu().tolist()
    k_ms = ev_ms / args.steps / npairs
    bytes_launch = sum(pair_bytes) / npairs
    achieved = bytes_launch / k_ms / 1e6
    c1_traffic, c1_tsrc = None, None
    if args.nbits == 1_000_000_000 and dq == 6554 and os.environ.get("BMX_PAIR_STREAM", "-1") == "-1":
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_config1.json")))
            c1_traffic, c1_tsrc = tj["hbm_bytes_per_launch"], tj["source"] + " (PMC pass of a separate rocprofv3 run, not measured in this run)"
        except Exception:
            pass
    res = {"metric": "Gbit/s of operand bits, pairwise count_and on 1e9-bit vectors (HBM-cold rotation)",
           "value": round(2 * args.nbits * npairs * args.steps / dt / 1e9, 2), "unit": "Gbit/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"bm::count_and/or/xor/sub + bit_and/or/xor/sub on 2 x {args.nbits}-bit vectors, density q16 {dq}, "
                                  f"rotating over {npairs} distinct pairs ({sum(pair_bytes) / 1e9:.2f} GB: not Infinity-Cache resident)",
                      "baseline_config": "configs[1]", "block_types_vec0": va[0].calc_stat(), "per_op": per_op,
                      "count_and": counts[:4]},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": c1_traffic, "traffic_source": c1_tsrc,
                        "kernel": ("k_count_op2_stream<4, true>" if all(v.calc_stat()["bit_blocks"] == v.info()["nblocks"] for v in (va[0], vb[0]))
                                   and va[0].info()["nblocks"] >= 2048 and os.environ.get("BMX_PAIR_STREAM", "-1") == "-1" else "k_count_op2"),
                        "algorithmic_bytes_per_launch": int(bytes_launch), "avg_launch_ms": round(k_ms, 4),
                        "timing": "hipEvent pair on the launch stream around the timed region / (steps x pairs)"}}
    if not args.no_cpu:
        try:
            P, orc, kind = _pick_oracle()
            sb = min(2048, (args.nbits + 65535) // 65536)
            wa = P.gen_words(SEED, 1, dq, args.nbits, word_off=0, nwords=sb * 2048)
            wb = P.gen_words(SEED, 2, dq, args.nbits, word_off=0, nwords=sb * 2048)
            ha, hb = orc.import_words(wa, True, sb * 65536), orc.import_words(wb, True, sb * 65536)
            best = None
            for _ in range(20):
                t0 = time.perf_counter(); c = orc.count_op2(0, ha, hb); d = time.perf_counter() - t0
                best = d if best is None else min(best, d)
            res["cpu_baseline"] = {"value": round(2 * sb * 65536 / best / 1e9, 2), "unit": "Gbit/s", "cores": 1, "kind": kind,
                                   "impl": orc.name, "sample": f"bm::count_and on the first {sb} blocks of pair 0, best of 20", "count": int(c)}
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(res))


# ----------------------------------------------------------------------------------------------------
# configs[3]: rank / select, 10 M random queries on one 4e9-bit vector
# ----------------------------------------------------------------------------------------------------
def run_rank_select(args):
    import numpy as np
    import torch
    import bitmagic_amd as bm
    from bitmagic_amd import _ffi
    world, rank, local_rank, use_dist = setup_dist(args)
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = bm.context(local_rank, s.cuda_stream)
    L = _ffi.lib()
    nbits, nq = 4_000_000_000, args.queries
    v = bm.bvector.generate(ctx, SEED, 7, args.density_q16, nbits)
    rs = v.build_rs_index()
    build_ms = event_avg_ms(lambda: v.build_rs_index(), 3, ctx)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    qn = torch.randint(0, nbits, (nq,), device="cuda", dtype=torch.int64, generator=g)
    cnt = rs.count()
    qr = torch.randint(1, cnt + 1, (nq,), device="cuda", dtype=torch.int64, generator=g)
    out = torch.zeros(nq, dtype=torch.int64, device="cuda"); pos = torch.zeros(nq, dtype=torch.int64, device="cuda")
    found = torch.zeros(nq, dtype=torch.uint8, device="cuda")
    do_rank = lambda: _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, qn.data_ptr(), nq, out.data_ptr()))
    do_sel = lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qr.data_ptr(), nq, pos.data_ptr(), found.data_ptr()))
    rank_ms = event_avg_ms(do_rank, 10, ctx); sel_ms = event_avg_ms(do_sel, 10, ctx)
    def step():
        do_rank(); do_sel()
    dt, ev_ms = timed_region(step, args.steps, args.warmup, ctx, use_dist)
    chk = torch.zeros(nq, dtype=torch.int64, device="cuda")
    _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, pos.data_ptr(), nq, chk.data_ptr())); torch.cuda.synchronize()
    ok = bool((chk == qr).all().item()) and bool(found.all().item())
    # rank touches 8 (rcount) + 2 (cum) + 8 (desc) + 128 B (bit line) per query; HBM moves 128 B lines
    line_bytes = nq * (4 * 128)
    achieved = line_bytes / rank_ms / 1e6
    res = {"metric": "M queries/s, rank + select (bmrs.h RS-index) on one 4e9-bit vector",
           "value": round(2 * nq * args.steps / dt / 1e6, 1), "unit": "Mqueries/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"{nq} random rank(n) + {nq} random select(r) per step on a {nbits}-bit vector, density q16 {args.density_q16}",
                      "baseline_config": "configs[3]", "block_types": v.calc_stat(), "count": cnt,
                      "rs_build_ms": round(build_ms, 4), "rank_ms": round(rank_ms, 4), "select_ms": round(sel_ms, 4),
                      "rank_Mq_s": round(nq / rank_ms / 1e3, 1), "select_Mq_s": round(nq / sel_ms / 1e3, 1),
                      "rank_select_roundtrip_ok": ok},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_rank",
                        "algorithmic_bytes_per_launch": line_bytes, "avg_launch_ms": round(rank_ms, 4),
                        "note": "random access: 4 distinct 128-B lines per rank query (running count, cumulative row, descriptor, bit line); "
                                "the bound is the HBM transaction rate, not streaming bandwidth"}}
    if not args.no_cpu:
        try:
            P, orc, kind = _pick_oracle()
            sbits = 512 * 65536
            w = P.gen_words(SEED, 7, args.density_q16, nbits, word_off=0, nwords=512 * 2048)
            hv = orc.import_words(w, True, sbits); hrs = orc.rs_build(hv)
            rng = np.random.default_rng(1)
            q = rng.integers(0, sbits, 1_000_000, dtype=np.uint64)
            t0 = time.perf_counter(); r = hrs.rank(q); d1 = time.perf_counter() - t0
            rr = rng.integers(1, hrs.count() + 1, 1_000_000, dtype=np.uint64)
            t0 = time.perf_counter(); hrs.select(rr); d2 = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": round(2e6 / (d1 + d2) / 1e6, 2), "unit": "Mqueries/s", "cores": 1, "kind": kind, "impl": orc.name,
                                   "sample": "1 M rank + 1 M select on the first 512 blocks (cache-friendlier than the 4e9-bit vector)",
                                   "rank_Mq_s": round(1.0 / d1, 2), "select_Mq_s": round(1.0 / d2, 2)}
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Mqueries/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(res))


# ----------------------------------------------------------------------------------------------------
# configs[4]: combine_or over 4096 x 4e9-bit sparse vectors, block-range sharded over the ranks (strong)
# ----------------------------------------------------------------------------------------------------
def run_or_sharded(args):
    import torch
    import torch.distributed as dist
    import bitmagic_amd as bm
    world, rank, local_rank, use_dist = setup_dist(args)
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    ctx = bm.context(local_rank, s.cuda_stream)
    nbits, nvec, dq = 4_000_000_000, args.or_vecs, 13                  # 13/65536 = 0.02 %
    nblocks = (nbits + 65535) // 65536
    lo, hi = bm.shard_range(nblocks, rank, world)
    t0 = time.perf_counter()
    vecs = [bm.bvector.generate(ctx, SEED, 10000 + i, dq, nbits, block_range=(lo, hi) if world > 1 else None) for i in range(nvec)]
    ctx.synchronize(); t_build = time.perf_counter() - t0
    gap_bytes = sum(v.info()["gap_words"] for v in vecs) * 2
    agg = bm.aggregator(ctx)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    last = []
    def step():
        t = agg.combine_or(vecs)
        cnt.fill_(t.count())
        if use_dist:
            dist.all_reduce(cnt)
        last[:] = [t]
    dt, ev_ms = timed_region(step, args.steps, args.warmup, ctx, use_dist)
    gb = torch.tensor([gap_bytes], dtype=torch.int64, device="cuda")
    if use_dist:
        dist.all_reduce(gb)
    tot_bytes = int(gb.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        achieved = gap_bytes / (ev_ms / args.steps) / 1e6
        res = {"metric": "Gbit/s of logical operand bits, aggregator combine_or over 4096 x 4e9-bit sparse vectors",
               "value": round(nvec * nbits * args.steps / dt / 1e9, 1), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u16", "data": "synthetic",
               "config": {"workload": f"aggregator::combine_or over {nvec} x {nbits}-bit vectors at 0.02 % (all GAP blocks), result materialised + counted",
                          "baseline_config": "configs[4]", "block_types_vec0": vecs[0].calc_stat(), "blocks_per_rank": hi - lo,
                          "gap_operand_bytes_total": tot_bytes, "result_count": int(cnt.item()),
                          "result_types_rank0": last[0].calc_stat(), "build_seconds": round(t_build, 1)},
               "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_agg_or_gap_tiled",
                            "algorithmic_bytes_per_launch": gap_bytes, "avg_launch_ms": round(ev_ms / args.steps, 4),
                            "note": "host call incl. result creation, layout scan and count; algorithmic bytes = 2 x (len + 1) per GAP operand"}}
        if not args.no_cpu and world == 1:
            try:
                P, orc, kind = _pick_oracle()
                sb, nv = 64, min(nvec, 1024)
                hv = [orc.import_words(P.gen_words(SEED, 10000 + i, dq, nbits, word_off=0, nwords=sb * 2048), True, sb * 65536) for i in range(nv)]
                t0 = time.perf_counter(); r = orc.agg_or(hv); d = time.perf_counter() - t0
                res["cpu_baseline"] = {"value": round(nv * sb * 65536 / d / 1e9, 1), "unit": "Gbit/s", "cores": 1, "kind": kind, "impl": orc.name,
                                       "sample": f"combine_or over {nv} vectors x first {sb} blocks, one pass", "count": r.count()}
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4], help="BASELINE.json configs[] index (2 = the headline)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"])
    ap.add_argument("--nvec", type=int, default=256)
    ap.add_argument("--nbits", type=int, default=NBITS_1G)
    ap.add_argument("--density-q16", type=int, default=6554)      # 10 %
    ap.add_argument("--independent", action="store_true", help="data set B (no common part; early exit)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=512)
    ap.add_argument("--cpu-cores", type=int, default=0, help="all-cores baseline: processes to use (0 = every core of the affinity mask)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-allcores", action="store_true")
    ap.add_argument("--no-weak", action="store_true")
    ap.add_argument("--no-shard-probe", action="store_true")
    ap.add_argument("--pairs", type=int, default=6)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--or-vecs", type=int, default=4096)
    args = ap.parse_args()
    if args.config == 1:
        run_pairwise(args)
    elif args.config == 3:
        run_rank_select(args)
    elif args.config == 4:
        run_or_sharded(args)
    else:
        run_headline(args)


if __name__ == "__main__":
    main()
