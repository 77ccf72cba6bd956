#!/usr/bin/env python3
"""bench.py -- benchmarks of the bvector/aggregator hot path on MI355X.  Prints ONE JSON line on rank 0.

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  aggregator::pipeline<agg_opt_only_counts> + combine_and_sub(pipe)
  (src/bmaggregator.h:1292-1399) = fused 256-way AND + COUNT over 256 bit-vectors of
  1e9 bits each, data set A of SURVEY.md section 8(d): v = common OR noise_v, both
  Bernoulli 10 % (mirrors GenerateTestCollection, tests/perf/perf.cpp:234-267), so no
  early exit is possible and every operand block must be read.

A "step" = one pass of the hot path over the resident vectors (the launches of one run, plus for N > 1 the
exchange of the 8-byte popcount).  Inputs are generated on the device and are resident in HBM before the timed
region starts.

--gpus N, N > 1 -- three ways in, all of them use N GPUs or exit non-zero (never an n_gpus: 1 line):
  * launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK in the environment):
    one process per GPU, torch.distributed over RCCL, one all-reduce of the count per step ("mode": "torchrun");
  * plain `python bench.py --gpus N` (no RANK): ONE process drives N devices through the product's own multi-GPU
    layer -- bmx_group_create(devices, N, BMX_GROUP_RCCL) + bmx_gpipeline_run_counts: kernels enqueued on every
    member, then an in-library ncclAllReduce over xGMI ("mode": "group"; --group-exchange host sums on the host);
  * `python bench.py --gpus N --launcher torchrun`: re-executes itself under torch.distributed.run.
  --scaling strong (default): the collection is FIXED at 256 x 1e9 bits; rank / member r holds only its block range
      of every vector (column independence: src/bmaggregator.h:1184-1218).  value = 256e9 bits / step time.
  weak: every rank owns its own 256 x 1e9-bit share.  Reported as the second figure "weak_scaling" of the strong line
      unless --no-weak.

--config 1|3|4 run the other BASELINE configs through the same JSON schema (roofline + cpu_baseline):
  1 pairwise count_and/or/xor/sub + materialised ops on 1e9-bit vectors, rotating over distinct vector pairs so
    that nothing is served from the 256 MB Infinity Cache;  3 rank/select, 10 M queries on a 4e9-bit vector;
  4 combine_or over 4096 x 4e9-bit sparse vectors, block-range sharded over the ranks / members.
The default run (config 2, one GPU) appends their one-line summaries as "other_configs" (--no-others skips them).
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
NBITS_4G = 4_000_000_000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
METRIC = "Gbits/s + % HBM roofline, 256-way fused AND+COUNT on 1B-bit vectors"
ONE_DEV_HOOK = "BMX_BENCH_TEST_ONE_DEVICE"     # test hook: every rank / member on device 0 (see setup_dist)


# ----------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself (oracle/_ref, kind "reference") or the C port, timed on the GPU box's
# host cores.  Only these legs of bench.py touch oracle/ -- as the thing measured NEXT to the GPU and as the
# checker of its results, never as the product path.
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
    """`bench.py --cpu-worker <kind> ...`: one independent replica (SURVEY section 8d "secondary = one replica per
    core") over block columns [lo, hi) of the workload -- block ranges are independent, so per-range results add up
    exactly.  Builds its inputs, prints READY, waits for a line on stdin, works, prints one JSON line.
      and  lo hi nvec dq nbits reps [c]  the headline: counts-only 256-way AND (c = 0: data set B, no common part)
      pair lo hi ida idb dq nbits reps   configs[1]: count_and/or/xor/sub of one pair
      rank lo hi id dq nbits             configs[3]: count of the range, then (second stdin line = JSON {"n": [...],
                                         "r": [...]} in range-local coordinates) rank / select answers
      or   nvec dq nbits b0,b1,...       configs[4]: combine_or over nvec vectors restricted to each listed block
    """
    kind, rest = argv[0], argv[1:]
    P, orc, okind = _pick_oracle()

    def build(vec_id, dq, nbits, lo, hi, with_common):
        sbits = max(min(nbits, hi * 65536) - lo * 65536, 0)
        w = P.gen_words(SEED, vec_id, dq, nbits, with_common=with_common, word_off=lo * 2048, nwords=(hi - lo) * 2048)
        return orc.import_words(w, True, sbits)

    def ready():
        sys.stdout.write("READY\n"); sys.stdout.flush()
        sys.stdin.readline()

    def reply(obj):
        obj.update({"kind": okind, "impl": orc.name})
        sys.stdout.write(json.dumps(obj) + "\n"); sys.stdout.flush()

    if kind == "and":
        lo, hi, nvec, dq, nbits, reps = (int(x) for x in rest[:6])
        with_common = (int(rest[6]) != 0) if len(rest) > 6 else True         # 0: data set B (independent operands)
        vecs = [build(v, dq, nbits, lo, hi, with_common) for v in range(nvec)]
        groups = [(vecs, [])]
        ready()
        spans, cnt = [], 0
        for _ in range(reps):
            t0 = time.perf_counter()                     # CLOCK_MONOTONIC: comparable across processes
            cnt = int(orc.pipeline_counts(groups)[0]) if hi > lo else 0
            spans.append((t0, time.perf_counter()))
        reply({"count": cnt, "spans": spans})
    elif kind == "pair":
        lo, hi, ida, idb, dq, nbits, reps = (int(x) for x in rest)
        a, b = build(ida, dq, nbits, lo, hi, False), build(idb, dq, nbits, lo, hi, False)
        ready()
        spans, counts = [], [0, 0, 0, 0]
        for _ in range(reps):
            t0 = time.perf_counter()
            c = int(orc.count_op2(0, a, b)) if hi > lo else 0
            spans.append((t0, time.perf_counter()))
            counts[0] = c
        for op in (1, 2, 3):
            counts[op] = int(orc.count_op2(op, a, b)) if hi > lo else 0
        reply({"counts": counts, "spans": spans})
    elif kind == "rank":
        import numpy as np
        lo, hi, vid, dq, nbits = (int(x) for x in rest)
        v = build(vid, dq, nbits, lo, hi, False)
        rs = orc.rs_build(v) if hi > lo else None
        ready()
        reply({"count": int(rs.count()) if rs is not None else 0})
        q = json.loads(sys.stdin.readline())
        n = np.asarray(q.get("n", []), dtype=np.uint64); r = np.asarray(q.get("r", []), dtype=np.uint64)
        ans_n = [int(x) for x in rs.rank(n)] if n.size else []
        ans_r = []
        if r.size:
            pos, _found = rs.select(r)                   # oracle RS.select -> (pos, found)
            ans_r = [int(x) for x in np.asarray(pos)]
        reply({"rank": ans_n, "select": ans_r})
    elif kind == "or":
        nvec, dq, nbits = (int(x) for x in rest[:3])
        blocks = [int(x) for x in rest[3].split(",") if x != ""]
        per = {b: [build(10000 + i, dq, nbits, b, b + 1, False) for i in range(nvec)] for b in blocks}
        ready()
        t0 = time.perf_counter()
        counts = {str(b): int(orc.agg_or(per[b]).count()) for b in blocks}
        reply({"counts": counts, "seconds": time.perf_counter() - t0})
    else:
        raise SystemExit(f"unknown cpu worker kind {kind}")


def _spawn_workers(arglists):
    import subprocess
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker"] + [str(a) for a in al],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT) for al in arglists]
    return procs


def _finish_workers(procs):
    for p in procs:
        try:
            p.stdin.close()
        except Exception:
            pass
        try:
            p.wait(timeout=30)
        except Exception:
            p.kill()


def _go(procs):
    for p in procs:
        line = p.stdout.readline()
        if line.strip() != "READY":
            raise RuntimeError(f"cpu worker failed to start: {line!r}")
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()


def cpu_baseline_allcores(nvec: int, dq: int, nbits: int, cores: int, reps: int = 3, with_common: bool = True):
    """the FULL workload on all host cores: every block column of all nvec vectors, block ranges fanned over
    `cores` worker processes (plain interpreters that never load HIP).  Returns the exact full-size count (pins
    the GPU's headline result against the reference) and the aggregate rate = all operand bits x reps /
    (last end - first start) of the passes, all replicas running at once."""
    from bitmagic_amd.sharding import shard_range
    nblocks = (nbits + 65535) // 65536
    cores = max(1, min(cores, nblocks))
    t0 = time.perf_counter()
    procs = _spawn_workers([["and", *shard_range(nblocks, w, cores), nvec, dq, nbits, reps, int(with_common)] for w in range(cores)])
    try:
        _go(procs)
        t_ready = time.perf_counter() - t0
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        _finish_workers(procs)
    full = sum(r["count"] for r in res)
    first = min(r["spans"][0][0] for r in res); last = max(r["spans"][-1][1] for r in res)
    bits = nvec * nblocks * 65536
    return {"full_count": int(full), "allcores_gbit_s": round(bits * reps / (last - first) / 1e9, 1), "cores_used": cores,
            "allcores_ms_per_pass": round((last - first) / reps * 1e3, 2), "allcores_kind": res[0]["kind"],
            "allcores_impl": res[0]["impl"],
            "allcores_sample": f"the whole workload: {nvec} vectors x {nblocks} blocks, one block-range replica per core, "
                               f"{reps} back-to-back passes with all replicas running (last end - first start); "
                               f"input build {t_ready:.1f}s not timed"}


def cpu_pair_allcores(ida: int, idb: int, dq: int, nbits: int, cores: int, reps: int = 3):
    """configs[1] on all host cores: count_and/or/xor/sub of ONE full-size pair, block ranges fanned out"""
    from bitmagic_amd.sharding import shard_range
    nblocks = (nbits + 65535) // 65536
    cores = max(1, min(cores, nblocks))
    procs = _spawn_workers([["pair", *shard_range(nblocks, w, cores), ida, idb, dq, nbits, reps] for w in range(cores)])
    try:
        _go(procs)
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        _finish_workers(procs)
    counts = [sum(r["counts"][op] for r in res) for op in range(4)]
    # a worker's range of a 1e9-bit pair is ~60 blocks = microseconds of work: the span from the first start to the last end across
    # 256 processes measures how fast this process can write 256 "go" lines, not the work (round 5 reported 12 Gbit/s for 256 cores
    # that way).  Reported: the operand bits / the BUSIEST worker's own time per pass -- what the cores deliver when they run together.
    busiest = max(sum(t1 - t0 for t0, t1 in r["spans"]) for r in res) / reps
    return {"full_counts": counts, "allcores_gbit_s": round(2 * nblocks * 65536 / busiest / 1e9, 1), "cores_used": cores,
            "allcores_timing": "operand bits / the busiest worker's own time per pass (input build and the release of the workers not timed)"}


def cpu_rank_allcores(vid: int, dq: int, nbits: int, cores: int, sample_n, sample_r_fn):
    """configs[3] on all host cores: every worker indexes its block range of the 4e9-bit vector; the range totals are
    prefix-summed here (the one exchange SURVEY section 8(e) names), sampled rank / select queries are routed to the
    range that owns them.  sample_r_fn(total) -> the sampled 1-based ranks.  -> (total, rank answers, select answers)"""
    import numpy as np
    from bitmagic_amd.sharding import shard_range
    nblocks = (nbits + 65535) // 65536
    cores = max(1, min(cores, nblocks))
    ranges = [shard_range(nblocks, w, cores) for w in range(cores)]
    procs = _spawn_workers([["rank", lo, hi, vid, dq, nbits] for lo, hi in ranges])
    try:
        _go(procs)
        cnt = [json.loads(p.stdout.readline())["count"] for p in procs]
        before = np.concatenate([[0], np.cumsum(np.asarray(cnt, dtype=np.uint64))]).astype(np.uint64)
        total = int(before[-1])
        sample_n = np.asarray(sample_n, dtype=np.uint64)
        sample_r = np.asarray(sample_r_fn(total), dtype=np.uint64)
        los = np.asarray([lo for lo, _ in ranges], dtype=np.uint64)
        own_n = np.searchsorted(los, sample_n >> np.uint64(16), side="right") - 1
        own_r = np.searchsorted(before, sample_r, side="left") - 1          # before[m] < r <= before[m + 1]
        for w, p in enumerate(procs):
            qn = sample_n[own_n == w] - los[w] * np.uint64(65536)
            qr = sample_r[own_r == w] - before[w]
            p.stdin.write(json.dumps({"n": [int(x) for x in qn], "r": [int(x) for x in qr]}) + "\n"); p.stdin.flush()
        ans_n = np.zeros(sample_n.size, np.uint64); ans_r = np.zeros(sample_r.size, np.uint64)
        for w, p in enumerate(procs):
            a = json.loads(p.stdout.readline())
            ans_n[own_n == w] = np.asarray(a["rank"], dtype=np.uint64) + before[w]
            ans_r[own_r == w] = np.asarray(a["select"], dtype=np.uint64) + los[w] * np.uint64(65536)
    finally:
        _finish_workers(procs)
    return total, sample_n, ans_n, sample_r, ans_r


def cpu_or_sample(nvec: int, dq: int, nbits: int, blocks, cores: int):
    """configs[4] on all host cores, bounded: combine_or over ALL nvec vectors restricted to each sampled block column
    (a column is independent, src/bmaggregator.h:1113-1121).  -> {block: count}, busiest worker's seconds"""
    blocks = list(blocks)
    cores = max(1, min(cores, len(blocks)))
    lists = [blocks[w::cores] for w in range(cores)]
    procs = _spawn_workers([["or", nvec, dq, nbits, ",".join(str(b) for b in bl)] for bl in lists])
    try:
        _go(procs)
        res = [json.loads(p.stdout.readline()) for p in procs]
    finally:
        _finish_workers(procs)
    counts = {}
    for r in res:
        counts.update({int(k): v for k, v in r["counts"].items()})
    return counts, max(r["seconds"] for r in res), res[0]["kind"], res[0]["impl"], cores


# ----------------------------------------------------------------------------------------------------
# process set-up
# ----------------------------------------------------------------------------------------------------
class Env:
    """how this process takes part in the run: mode single | torchrun | group"""
    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.one_dev = os.environ.get(ONE_DEV_HOOK) == "1"
        launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
        self.mode = "torchrun" if launched else ("group" if args.gpus > 1 else "single")
        self.use_dist = launched
        self.dist = None
        self.stream = None
        self.ctx = None

    def need_devices(self, n):
        have = self.torch.cuda.device_count()
        if not self.one_dev and have < n:
            sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible devices, found {have}\n")
            sys.exit(2)

    def setup(self):
        """single / torchrun: one device, one explicit stream shared by the HIP kernels, torch and RCCL"""
        import bitmagic_amd as bm
        torch, args = self.torch, self.args
        if self.mode == "torchrun":
            if self.world != args.gpus:
                sys.stderr.write(f"bench.py: --gpus {args.gpus} but torch.distributed.run started {self.world} ranks\n")
                sys.exit(2)
            self.need_devices(self.world)
        else:
            self.need_devices(1)
        if self.one_dev:
            # test hook (exercising the N > 1 code path on a ONE-GPU box): every rank on device 0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device).  Timings of such a run mean nothing; it checks sharding,
            # gathers and the JSON line.
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if self.use_dist:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.one_dev:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                # RCCL prints a version banner on STDOUT when its first communicator comes up: keep stdout for the ONE
                # JSON line (the banner goes to stderr) by creating the communicator under a redirect
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                            device_id=torch.device("cuda", self.local_rank))
                    t = torch.zeros(1, dtype=torch.int64, device="cuda")
                    dist.all_reduce(t)
                    torch.cuda.synchronize()
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        self.ctx = bm.context(self.local_rank, self.stream.cuda_stream)
        return self

    def rccl_ranks(self):
        if not self.use_dist:
            return None
        return self.dist.get_world_size()

    def finish(self):
        if self.use_dist:
            self.dist.destroy_process_group()


def reexec_torchrun(args):
    """`bench.py --gpus N --launcher torchrun` without RANK: become the launcher"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv = [a for a in sys.argv[1:] if a != "--launcher" and a != "torchrun" and not a.startswith("--launcher=")]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    os.execv(sys.executable, cmd)


def timed_region(step, steps, warmup, env):
    """W warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; wall time = max over ranks.
    Also returns the HIP-event time of the same region on the launch stream."""
    torch, dist = env.torch, env.dist
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if env.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    env.ctx.timer_start()                                # HIP events on the launch stream
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ev_ms = env.ctx.timer_stop_ms()
    torch.cuda.synchronize()
    if env.use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if env.one_dev else "cuda")
    if env.use_dist:
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


def gather_floats(x: float, env):
    torch = env.torch
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if env.one_dev else "cuda")
    if not env.use_dist:
        return [float(x)]
    out = [torch.zeros_like(t) for _ in range(env.world)]
    env.dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def traffic_file(name: str, kernel: str = None, workload: str = None):
    """HBM bytes per launch from the PMC passes of a separate rocprofv3 run (rocprofv3 --pmc cannot run inside this process):
    a constant read from profiles/<name>, which is stamped with the kernel and the commit it was measured on
    (tools/make_traffic_json.py).  The figure is DROPPED -- traffic null, the reason in traffic_source -- when the file
    carries no stamp, or when the kernel this run reports is not the one the file was measured on."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None, None, {}
    if workload is not None and tj.get("workload") != workload:
        return None, None, tj
    if not tj.get("kernel") or not tj.get("commit"):
        return None, f"dropped: profiles/{name} is not stamped with the kernel / commit it was measured on", tj
    if kernel is not None and tj["kernel"] not in kernel:
        return None, f"dropped: profiles/{name} was measured on {tj['kernel']} (commit {tj['commit']}), this run used {kernel[:60]}", tj
    return (tj["hbm_bytes_per_launch"],
            f"{tj['source']} [kernel {tj['kernel']}, commit {tj['commit']}] (PMC passes of a separate rocprofv3 run, not measured in this run)", tj)


def dataset_label(args):
    """the data set as it was actually generated (density from --density-q16, not a fixed string)"""
    pct = args.density_q16 / 65536 * 100
    d = f"{pct:.3g}%"
    if args.independent:
        return f"data set B (independent Bernoulli {d}: the AND dies early)"
    return f"data set A (common {d} OR noise {d}: no early exit)"


def headline_workload(args, scaling, world):
    return (f"aggregator pipeline combine_and_sub counts-only: {args.nvec}-way AND+COUNT, "
            f"{args.nvec} x {args.nbits}-bit vectors " + ("per GPU, " if scaling == "weak" and world > 1 else "in total, ")
            + dataset_label(args))


# ----------------------------------------------------------------------------------------------------
# configs[2]: the headline
# ----------------------------------------------------------------------------------------------------
def headline_cpu(args):
    """CPU legs (rank 0 of a 1-GPU run only), before torch / HIP are loaded into this process"""
    nblocks_full = (args.nbits + 65535) // 65536
    sb = min(args.cpu_sample_blocks, nblocks_full)
    try:
        cpu = cpu_baseline_1core(args.nvec, args.density_q16, args.nbits, sb, None)
        if not args.no_allcores and not args.independent:
            ncores = args.cpu_cores or len(os.sched_getaffinity(0))
            cpu.update(cpu_baseline_allcores(args.nvec, args.density_q16, args.nbits, ncores))
    except Exception as e:  # the baseline is a reported number, never the product path
        cpu = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    return cpu


def run_headline(args, env, cpu):
    import bitmagic_amd as bm
    torch, dist = env.torch, env.dist
    world, rank, use_dist, ctx = env.world, env.rank, env.use_dist, env.ctx
    nblocks_full = (args.nbits + 65535) // 65536
    sb = min(args.cpu_sample_blocks, nblocks_full)
    scaling = args.scaling if args.scaling != "auto" else "strong"       # N = 1: both modes are the same run
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
        # a collection of vectors WITHOUT bit-blocks (densities below ~0.4 %) is an index the application prepares once at load
        # time (bmx_collection_prepare, the AND role: the vectors' 0-runs in column-major order); collections are never built
        # behind the caller's back, so the bench does what such a caller does and reports the one-time cost
        prep_ms = None
        if all(v.info()["counts"][bm.BIT] == 0 for v in vecs) and os.environ.get("BMX_GAP_PACK", "-1") != "0" and not args.no_prepare:
            ctx.synchronize(); tp = time.perf_counter()
            ctx.collection_prepare(vecs, bm.ROLE_AND); ctx.synchronize()
            prep_ms = (time.perf_counter() - tp) * 1e3
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
        dt, ev_ms = timed_region(step, args.steps, args.warmup, env)
        total = int(counts.item())
        k_ms = event_avg_ms(kernel, max(5, min(args.steps, 20)), ctx)      # the kernel alone
        ar_us = None
        if use_dist:
            ar_us = event_avg_ms(lambda: dist.all_reduce(counts), 20, ctx) * 1e3
        return {"dt": dt, "ev_ms": ev_ms, "count": total, "op_bytes": op_bytes, "k_ms": k_ms, "ar_us": ar_us,
                "build_s": t_build, "blocks": hi - lo, "plan": pipe.describe(), "nlaunch": pipe.launches(),
                "stat0": vecs[0].calc_stat(), "mem": ctx.mem_used(), "pipe": pipe, "vecs": vecs, "prep_ms": prep_ms}

    main = run_mode(scaling)
    k_all = gather_floats(main["k_ms"], env)
    bytes_all = gather_floats(float(main["op_bytes"]), env)
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
    if world == 1 and cpu is not None:
        gpu_sample = int(agg._run_pipeline(main["pipe"], 0, sb)[0])
    # the materialised form of the same aggregation: aggregator::combine_and (BASELINE configs[2] names "combine_and + count"):
    # the WHOLE host call (row build, kernel, result layout), result stored with opt_compress (src/bmaggregator.h:1210)
    mat = None
    if world == 1 and scaling == "strong" and not args.no_others and not args.independent:
        ts = []
        for _ in range(4):
            ctx.synchronize(); t0 = time.perf_counter()
            t, _any = agg.combine_and_sub(main["vecs"], [])
            ts.append((time.perf_counter() - t0) * 1e3)
            rc, rstat = t.count(), t.calc_stat()
            del t
        mms = min(ts[1:])
        mbytes = main["op_bytes"] + rstat["bit_blocks"] * 8192
        mat = {"host_call_ms": round(mms, 4), "result_count": rc, "result_block_types": rstat, "count_equal": bool(rc == main["count"]),
               "algorithmic_bytes": mbytes, "frac": round(mbytes / mms / 1e6 / HBM_PEAK_GBS, 4),
               "note": "bmx_agg_and_sub over the 256 resident vectors: best of 3 whole host calls (pipeline rows built per call, "
                       "k_agg_and_sub, result kinds + layout); bytes = operand blocks + 8,192 B per stored result block"}
    # host -> device: what the boundary costs when the operands start on the host (SURVEY section 8(d) "separately report H2D upload time")
    h2d = None
    if world == 1 and scaling == "strong" and not args.no_others and not args.independent:
        try:
            uv = main["vecs"][1]
            kinds, offs, bits, gaps = uv.block_table()
            words = uv.to_words()
            def best_of(fn, n=4):
                tt = []
                for _ in range(n):
                    ctx.synchronize(); t0 = time.perf_counter(); u = fn(); ctx.synchronize(); tt.append(time.perf_counter() - t0); del u
                return min(tt[1:])
            tb = best_of(lambda: bm.bvector.from_block_table(ctx, args.nbits, kinds, offs, bits, gaps))
            ti = best_of(lambda: bm.bit_import_u32(ctx, words, True))
            nbytes = bits.nbytes + gaps.nbytes + kinds.nbytes + offs.nbytes
            h2d = {"vector": f"one {args.nbits}-bit vector of the collection ({uv.calc_stat()['bit_blocks']} bit-blocks)", "bytes": int(nbytes),
                   "bmx_vec_upload_ms": round(tb * 1e3, 3), "bmx_vec_upload_GBps": round(nbytes / tb / 1e9, 2),
                   "bmx_vec_import_bits_ms": round(ti * 1e3, 3), "bmx_vec_import_bits_GBps": round(words.nbytes / ti / 1e9, 2),
                   "whole_collection_upload_s_estimate": round(tb * args.nvec, 2),
                   "note": "host block table (contiguous slabs = what a freeze()d bm::bvector<> hands over, include/bmx/bm_adapter.hpp flatten_view) "
                           "-> bmx_vec_upload, pageable host memory, best of 3 calls incl. the final synchronise; import_bits = raw words, "
                           "classified and compressed on the device.  Never part of `value`: the timed region starts with the operands resident in HBM"}
            del kinds, offs, bits, gaps, words
        except Exception as e:
            h2d = {"error": str(e)}
    weak = None
    del main["pipe"], main["vecs"]
    if world > 1 and scaling == "strong" and not args.no_weak:
        ctx.trim()
        w = run_mode("weak")
        weak = {"value": round(world * args.nvec * args.nbits * args.steps / w["dt"] / 1e9, 2), "unit": "Gbit/s",
                "ms_per_step": round(w["dt"] / args.steps * 1e3, 4), "result_count": w["count"],
                "note": f"every rank owns its own {args.nvec} x {args.nbits}-bit collection (document-sharded index)"}
        del w["pipe"], w["vecs"]
    ctx.trim()
    if rank != 0:
        return None
    bits_per_step = args.nvec * args.nbits * (world if scaling == "weak" else 1)
    value = bits_per_step * args.steps / main["dt"] / 1e9
    achieved = main["op_bytes"] / (main["k_ms"] * 1e-3) / 1e9
    # the PMC figure belongs to the headline data set on one GPU (density 10 %, data set A, 6 launch windows): any other
    # density / data set / shard runs another kernel or another launch plan and carries no traffic figure
    headline = args.density_q16 == 6554 and not args.independent and world == 1
    traffic, tsrc, _ = traffic_file("traffic_latest.json", kernel=main["plan"], workload=f"agg_and_count_{args.nvec}x{args.nbits}") if headline else (None, None, {})
    if traffic is not None and main["nlaunch"] != 6: traffic, tsrc = None, None
    res = {
        "metric": METRIC, "value": round(value, 2), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(main["dt"] / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "mode": env.mode,
        "config": {"workload": headline_workload(args, scaling, world),
                   "baseline_config": "configs[2]", "vectors": args.nvec, "bits_per_vector": args.nbits,
                   "density_q16": args.density_q16, "blocks_per_vector": nblocks_full,
                   "block_types_vec0": main["stat0"],
                   "sharding": (f"block-range shards: rank r holds blocks shard_range({nblocks_full}, r, {world}) of every vector"
                                if scaling == "strong" else f"document shards x{world}"),
                   "blocks_per_rank": main["blocks"], "result_count": main["count"],
                   "build_seconds": round(main["build_s"], 2), "hbm_resident_bytes": main["mem"],
                   "prepared_collection_ms": None if main["prep_ms"] is None else round(main["prep_ms"], 2)},
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
                     "rccl_ranks": env.rccl_ranks(),
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
    if mat:
        res["materialised_combine_and"] = mat
    if h2d:
        res["h2d_upload"] = h2d
    if weak:
        res["weak_scaling"] = weak
    if cpu is not None:
        if cpu.get("value") is not None and not args.independent:
            cpu["matches_gpu"] = bool(cpu["count"] == gpu_sample)
            if "full_count" in cpu:
                cpu["matches_gpu_full"] = bool(cpu["full_count"] == main["count"])
        res["cpu_baseline"] = cpu
    return res


def run_headline_group(args):
    """plain `bench.py --gpus N` (no launcher): ONE process, N devices, through the product's multi-GPU layer
    (bmx_group + bmx_gpipeline_run_counts; include/bmx.h "device groups")"""
    import torch
    import bitmagic_amd as bm
    n = args.gpus
    one_dev = os.environ.get(ONE_DEV_HOOK) == "1"
    have = torch.cuda.device_count()
    if not one_dev and have < n:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible devices, found {have}\n")
        sys.exit(2)
    devices = [0] * n if one_dev else list(range(n))
    want_rccl = args.group_exchange == "rccl" and not one_dev
    rccl_error = None
    grp = None
    if want_rccl:
        sys.stdout.flush()
        saved = os.dup(1); os.dup2(2, 1)                # RCCL's banner goes to stderr, stdout carries the ONE JSON line
        try:
            grp = bm.group(devices, bm.GROUP_RCCL)
        except Exception as e:
            rccl_error = str(e)
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
    if grp is None:
        grp = bm.group(devices, bm.GROUP_HOST_SUM)
    exchange = "rccl_allreduce" if (want_rccl and rccl_error is None) else "host_sum"
    rccl_ranks = grp.rccl_ranks()
    nblocks_full = (args.nbits + 65535) // 65536
    scaling = args.scaling if args.scaling != "auto" else "strong"
    gagg = bm.gaggregator(grp)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def run_mode(mode):
        nbits = args.nbits if mode == "strong" else args.nbits * n      # weak: every member holds a 1e9-bit share
        t0 = time.perf_counter()
        vecs = [bm.gbvector.generate(grp, SEED, v, args.density_q16, nbits, with_common=not args.independent)
                for v in range(args.nvec)]
        pipe = bm.gpipeline(grp)
        ag = pipe.add()
        for v in vecs:
            ag.add(v, 0)
        pipe.complete()
        sync_all()
        t_build = time.perf_counter() - t0
        step = lambda: gagg.combine_and_sub(pipe)
        for _ in range(args.warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cnt = step()
        sync_all()
        dt = time.perf_counter() - t0
        k_ms = [0.0] * n; x_ms = [0.0] * n
        reps = max(5, min(args.steps, 20))
        for _ in range(reps):                            # per-member device times (HIP events on the member streams)
            step()
            k_ms = [a + b for a, b in zip(k_ms, pipe.last_ms())]
            x_ms = [a + b for a, b in zip(x_ms, pipe.last_exchange_ms())]
        k_ms = [a / reps for a in k_ms]; x_ms = [a / reps for a in x_ms]
        plan, nlaunch = pipe.describe(0)
        r = {"dt": dt, "count": int(cnt[0]), "k_ms": k_ms, "x_ms": x_ms, "bytes": pipe.operand_bytes(), "plan": plan,
             "nlaunch": nlaunch, "build_s": t_build, "stat0": vecs[0].info(),
             "ranges": [grp.shard_range((nbits + 65535) // 65536, m) for m in range(n)]}
        del pipe, vecs
        return r

    main = run_mode(scaling)
    weak = None
    if scaling == "strong" and not args.no_weak:
        w = run_mode("weak")
        weak = {"value": round(n * args.nvec * args.nbits * args.steps / w["dt"] / 1e9, 2), "unit": "Gbit/s",
                "ms_per_step": round(w["dt"] / args.steps * 1e3, 4), "result_count": w["count"],
                "note": f"{args.nvec} vectors of {n} x {args.nbits} bits: every member holds a {args.nbits}-bit share of each"}
    bits_per_step = args.nvec * args.nbits * (n if scaling == "weak" else 1)
    value = bits_per_step * args.steps / main["dt"] / 1e9
    achieved = main["bytes"][0] / (main["k_ms"][0] * 1e-3) / 1e9
    c = main["stat0"]["counts"]
    res = {
        "metric": METRIC, "value": round(value, 2), "unit": "Gbit/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(main["dt"] / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "mode": "group", "exchange": exchange,
        "config": {"workload": headline_workload(args, scaling, n), "baseline_config": "configs[2]", "vectors": args.nvec,
                   "bits_per_vector": args.nbits, "density_q16": args.density_q16, "blocks_per_vector": nblocks_full,
                   "block_types_vec0": {"null_blocks": c[0], "full_blocks": c[1], "bit_blocks": c[2], "gap_blocks": c[3]},
                   "sharding": "bmx_group: member m holds blocks bmx_group_shard_range(nblocks, m) of every vector",
                   "member_block_ranges": main["ranges"], "devices": devices, "result_count": main["count"],
                   "build_seconds": round(main["build_s"], 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": main["plan"],
                     "launches_per_step": main["nlaunch"], "algorithmic_bytes_per_launch": main["bytes"][0] // max(main["nlaunch"], 1),
                     "avg_launch_ms": round(main["k_ms"][0] / max(main["nlaunch"], 1), 4),
                     "algorithmic_bytes_per_step": main["bytes"][0], "kernel_ms_per_step": round(main["k_ms"][0], 4),
                     "scope": "member 0's GPU",
                     "timing": "hipEvent pair on member 0's stream around its launches of one bmx_gpipeline_run_counts call, "
                               "averaged over the calls after the timed region"},
        "per_rank": {"kernel_ms": [round(x, 4) for x in main["k_ms"]],
                     "GBps": [round(b / (k * 1e-3) / 1e9, 1) if k > 0 else None for b, k in zip(main["bytes"], main["k_ms"])],
                     "allreduce_us": round(max(main["x_ms"]) * 1e3, 1), "rccl_ranks": rccl_ranks,
                     "exchange_us_per_member": [round(x * 1e3, 1) for x in main["x_ms"]],
                     "note": "allreduce_us = device time from the end of a member's kernel to the end of the exchange (it "
                             "includes waiting for the slowest member)"},
    }
    if one_dev:
        res["test_hook"] = f"{ONE_DEV_HOOK}=1: all members on device 0; timings mean nothing"
    res["rccl_ranks"] = rccl_ranks                              # (also under per_rank) printed whatever the exchange turned out to be
    res["rccl_error"] = rccl_error
    res["rccl_requested"] = bool(want_rccl)
    if weak:
        res["weak_scaling"] = weak
    grp.close()
    return res


# ----------------------------------------------------------------------------------------------------
# configs[1]: pairwise ops on 1e9-bit vectors, HBM-cold
# ----------------------------------------------------------------------------------------------------
def run_pairwise(args, env, dq=None, quick=False):
    import ctypes as C
    import bitmagic_amd as bm
    from bitmagic_amd import _ffi
    torch, ctx = env.torch, env.ctx
    L = _ffi.lib()
    dq = args.density_q16 if dq is None else dq
    nbits = args.nbits
    npairs = args.pairs                                  # distinct pairs: npairs x 250 MB >> 256 MB Infinity Cache
    va = [bm.bvector.generate(ctx, SEED, 2 * i + 1, dq, nbits) for i in range(npairs)]
    vb = [bm.bvector.generate(ctx, SEED, 2 * i + 2, dq, nbits) for i in range(npairs)]
    pair_bytes = []
    for a, b in zip(va, vb):
        ia, ib = a.info(), b.info()
        pair_bytes.append(a.operand_bytes() + b.operand_bytes())            # 8,192 B per bit-block + 2 x (len + 1) per GAP block
    dcnt = torch.zeros(4 * npairs, dtype=torch.int64, device="cuda")
    per_op = {}
    mat0 = []
    for op, name in enumerate(["and", "or", "xor", "sub"]):
        def sweep(op=op):
            for i in range(npairs):
                L.bmx_count_op2_dev(ctx._h, op, va[i]._h, vb[i]._h, C.c_void_p(dcnt.data_ptr() + 8 * (op * npairs + i)))
        ms = event_avg_ms(sweep, 3 if quick else 5, ctx) / npairs
        per_op[name] = {"kernel_ms": round(ms, 4), "GBps": round(sum(pair_bytes) / npairs / ms / 1e6, 1)}
        # materialised result (opt_none), whole host call incl. result creation
        keep = []
        def mat(op=op):
            for i in range(npairs):
                keep.append(bm.bvector._op2(op, va[i], vb[i], bm.opt_none))
                if len(keep) > 2: keep.pop(0)
        mat(); ctx.synchronize()
        mat0.append(int(bm.bvector._op2(op, va[0], vb[0], bm.opt_none).count()))     # the materialised result of pair 0, counted (checked against the reference below)
        t0 = time.perf_counter(); mat(); ctx.synchronize(); host_ms = (time.perf_counter() - t0) * 1e3 / npairs
        out_blocks = keep[-1].info()["counts"][2]
        per_op[name]["materialised_host_call_ms"] = round(host_ms, 4)
        per_op[name]["materialised_GBps"] = round((pair_bytes[-1] + out_blocks * 8192) / host_ms / 1e6, 1)
        keep.clear()
        # the same operations through the asynchronous entry (bmx_op2_dev): every pair enqueued, ONE wait per sweep -- what a caller
        # that chains / batches operations pays per operation
        if True:
            def mat_async(op=op):
                ps = [bm.bvector.op2_async(op, va[i], vb[i]) for i in range(npairs)]
                return [p.wait() for p in ps]
            rs = mat_async(); ctx.synchronize()
            assert int(rs[0].count()) == mat0[-1]
            del rs
            t0 = time.perf_counter(); rs = mat_async(); ctx.synchronize(); a_ms = (time.perf_counter() - t0) * 1e3 / npairs
            per_op[name]["materialised_async_ms_per_op"] = round(a_ms, 4)
            del rs
    torch.cuda.synchronize()
    # what the box gives a plain 2-read : 1-write elementwise kernel of the same sizes (torch.bitwise_and(out=) over rotating
    # 125 MB tensors): the yardstick for the materialised ops above, whose kernel moves the same bytes
    rw_probe = None
    own_probe = None
    if True:                                             # (cheap: also in the quick legs of the driver line)
        try:
            # the library's own yardstick: c = a & b in the launch shape of k_op2_stream (non-temporal 16-byte loads / stores),
            # no descriptors, no classification -- at 1, 2, 4, 8 workgroups per CU, rotating over 3 buffer triples
            best = None
            for wgs in (1, 2, 4, 8):
                pm = C.c_float()
                _ffi.check(L.bmx_probe_stream_rw(ctx._h, ((nbits + 65535) // 65536) * 8192, 3, wgs, 12, C.byref(pm)))
                if best is None or pm.value < best[1]: best = (wgs, pm.value)
            nb_bytes = ((nbits + 65535) // 65536) * 8192
            own_probe = {"kernel": "k_probe_rw<4> (bmx_probe_stream_rw): c = a & b, a wave per stretch of 8-KiB blocks, nt loads + nt stores",
                         "best_wgs_per_cu": best[0], "ms": round(best[1], 4), "GBps_read_plus_written": round(3 * nb_bytes / best[1] / 1e6, 1)}
        except Exception as e:
            own_probe = {"error": str(e)}
    if not quick:
        try:
            nw = (nbits + 63) // 64
            xs = [torch.randint(0, 1 << 62, (nw,), dtype=torch.int64, device="cuda") for _ in range(3 * 3)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(3): torch.bitwise_and(xs[3 * i], xs[3 * i + 1], out=xs[3 * i + 2])
            torch.cuda.synchronize(); e0.record()
            reps = 12
            for r in range(reps):
                i = r % 3
                torch.bitwise_and(xs[3 * i], xs[3 * i + 1], out=xs[3 * i + 2])
            e1.record(); torch.cuda.synchronize()
            pms = e0.elapsed_time(e1) / reps
            rw_probe = {"kernel": "torch.bitwise_and(a, b, out=c) on int64 tensors of the same size, rotating over 3 triples", "ms": round(pms, 4),
                        "GBps_read_plus_written": round(3 * nw * 8 / pms / 1e6, 1)}
            del xs
        except Exception as e:
            rw_probe = {"error": str(e)}
    all_counts = dcnt.cpu().tolist()
    pair0 = [all_counts[op * npairs] for op in range(4)]
    # timed region per the contract: a "step" = count_and over every pair (npairs launches)
    def step():
        for i in range(npairs):
            L.bmx_count_op2_dev(ctx._h, 0, va[i]._h, vb[i]._h, C.c_void_p(dcnt.data_ptr() + 8 * i))
    steps, warmup = (20, 3) if quick else (args.steps, args.warmup)      # (steps of well under a millisecond: enough of them for a stable wall time)
    dt, ev_ms = timed_region(step, steps, warmup, env)
    torch.cuda.synchronize()
    counts = dcnt.cpu().tolist()
    k_ms = ev_ms / steps / npairs
    bytes_launch = sum(pair_bytes) / npairs
    achieved = bytes_launch / k_ms / 1e6
    all_bit = all(v.calc_stat()["bit_blocks"] == v.info()["nblocks"] for v in (va[0], vb[0]))
    c1_traffic, c1_tsrc = None, None
    if nbits == NBITS_1G and dq == 6554 and os.environ.get("BMX_PAIR_STREAM", "-1") == "-1":
        c1_traffic, c1_tsrc, _ = traffic_file("traffic_config1.json", kernel=L_pair_kernel_name(all_bit, va[0].info()["nblocks"]))
    elif nbits == NBITS_1G and dq == 655 and os.environ.get("BMX_PAIR_LOOP", "-1") == "-1":
        c1_traffic, c1_tsrc, _ = traffic_file("traffic_config1_1pct.json", kernel=L_pair_kernel_name(all_bit, va[0].info()["nblocks"]))
    elif nbits == NBITS_1G and dq == 32768 and os.environ.get("BMX_PAIR_STREAM", "-1") == "-1":
        c1_traffic, c1_tsrc, _ = traffic_file("traffic_config1_50pct.json", kernel=L_pair_kernel_name(all_bit, va[0].info()["nblocks"]))
    pct = dq / 65536 * 100
    res = {"metric": "Gbit/s of operand bits, pairwise count_and on 1e9-bit vectors (HBM-cold rotation)",
           "value": round(2 * nbits * npairs * steps / dt / 1e9, 2), "unit": "Gbit/s", "n_gpus": 1,
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"bm::count_and/or/xor/sub + bit_and/or/xor/sub on 2 x {nbits}-bit vectors, Bernoulli {pct:.3g}% (density q16 {dq}), "
                                  f"rotating over {npairs} distinct pairs ({sum(pair_bytes) / 1e9:.2f} GB: not Infinity-Cache resident)",
                      "baseline_config": "configs[1]", "block_types_vec0": va[0].calc_stat(), "per_op": per_op, "read_write_probe": rw_probe, "own_read_write_probe": own_probe,
                      "count_and": counts[:4], "pair0_counts_and_or_xor_sub": pair0, "pair0_materialised_counts": mat0},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": c1_traffic, "traffic_source": c1_tsrc,
                        "kernel": L_pair_kernel_name(all_bit, va[0].info()["nblocks"]),
                        "algorithmic_bytes_per_launch": int(bytes_launch), "avg_launch_ms": round(k_ms, 4),
                        "timing": "hipEvent pair on the launch stream around the timed region / (steps x pairs)"}}
    del va, vb
    ctx.trim()
    if not args.no_cpu:
        try:
            P, orc, kind = _pick_oracle()
            sb = min(2048, (nbits + 65535) // 65536)
            wa = P.gen_words(SEED, 1, dq, nbits, word_off=0, nwords=sb * 2048)
            wb = P.gen_words(SEED, 2, dq, nbits, word_off=0, nwords=sb * 2048)
            ha, hb = orc.import_words(wa, True, sb * 65536), orc.import_words(wb, True, sb * 65536)
            best = None
            for _ in range(20):
                t0 = time.perf_counter(); c = orc.count_op2(0, ha, hb); d = time.perf_counter() - t0
                best = d if best is None else min(best, d)
            cpu = {"value": round(2 * sb * 65536 / best / 1e9, 2), "unit": "Gbit/s", "cores": 1, "kind": kind,
                   "impl": orc.name, "sample": f"bm::count_and on the first {sb} blocks of pair 0, best of 20", "count": int(c)}
            if not args.no_allcores:
                ncores = args.cpu_cores or len(os.sched_getaffinity(0))
                full = cpu_pair_allcores(1, 2, dq, nbits, ncores)
                # (no all-cores RATE for a pairwise operation: a worker's share of a 1e9-bit pair is ~60 blocks = microseconds, and neither
                # the span over 256 processes (round 5: it timed the release of the workers) nor the busiest worker's own time (one
                # descheduled process sets it) measures the cores; the fan-out is the whole-pair CHECK of the four counts)
                cpu.update({"full_counts_and_or_xor_sub": full["full_counts"], "cores_used": full["cores_used"],
                            "allcores_sample": "count_and/or/xor/sub of the WHOLE pair 0, block ranges fanned over the host cores",
                            "matches_gpu_full": bool(full["full_counts"] == pair0),
                            "matches_gpu_full_materialised": bool(full["full_counts"] == mat0)})
            res["cpu_baseline"] = cpu
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    return res


def _select_kernel_name(rs_info=None):
    """the kernel bmx_select_batch_dev takes for a big batch, by the library's own rule (bmx.hip: select lines if the index holds them;
    else the rank lines' directory with its summary in LDS -- two lanes per query unless BMX_RS_LANES says otherwise; else the tables)"""
    if rs_info and rs_info.get("select_offset_bits") and os.environ.get("BMX_RS_SELECT_SEL", "-1") != "0":
        return ("k_select_sel<u%d> (select lines: the ones' positions laid out %d per 128-byte line by build_rs_index -- one lane, one line per "
                "query, no search)" % (rs_info["select_offset_bits"], 60 if rs_info["select_offset_bits"] == 16 else 30))
    sl = os.environ.get("BMX_RS_SELECT_LINES", "2")
    lanes = os.environ.get("BMX_RS_LANES", "0")
    if os.environ.get("BMX_RS_LINES", "1") == "0" or sl == "0" or (rs_info is not None and not rs_info.get("has_lines")):
        return "k_select_l<%s> (block index: running counts, cumulative row, bit line)" % (lanes if lanes in ("2", "4") else "4")
    if sl == "1":
        return "k_select_lines<%s> (block index + octant directory, then the rank line guessed by interpolation and verified by its header)" % (lanes if lanes in ("2", "4") else "4")
    if os.environ.get("BMX_RS_SELECT_TOP", "-1") != "0":
        return ("k_select_top<%s> (the select directory's 65,536-entry summary in LDS -- one 1024-thread workgroup per CU -- then the rank line, "
                "interpolated guess verified by the line's header)" % (lanes if lanes in ("2", "4") else "2"))
    return "k_select_sdir<%s> (select directory over the rank lines: the line of every 2^k-th one, interpolated guess verified by the line's header)" % (lanes if lanes in ("2", "4") else "4")


def L_pair_kernel_name(all_bit, nblocks):
    env_ps = os.environ.get("BMX_PAIR_STREAM", "-1")
    if all_bit and nblocks >= 2048 and env_ps == "-1":
        return "k_count_op2_stream<4, true>"
    env_pl = os.environ.get("BMX_PAIR_LOOP", "-1")
    if nblocks >= 2048 and env_pl != "0":
        return "k_count_op2_loop<4,%s> (persistent: a wave walks every 4096th column, one memory round trip per column)" % (
            "false" if os.environ.get("BMX_PAIR_NT", "1") == "0" else "true")
    return "k_count_op2 (a wave per block column)"


# ----------------------------------------------------------------------------------------------------
# configs[3]: rank / select, 10 M random queries on one 4e9-bit vector
# ----------------------------------------------------------------------------------------------------
def run_rank_select(args, env, quick=False, dq=None):
    import ctypes as C
    import numpy as np
    import bitmagic_amd as bm
    from bitmagic_amd import _ffi
    torch, ctx = env.torch, env.ctx
    L = _ffi.lib()
    nbits, nq, dq = NBITS_4G, args.queries, (args.density_q16 if dq is None else dq)
    v = bm.bvector.generate(ctx, SEED, 7, dq, nbits)
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
    steps, warmup = (20, 3) if quick else (args.steps, args.warmup)      # (steps of well under a millisecond: enough of them for a stable wall time)
    dt, ev_ms = timed_region(step, steps, warmup, env)
    chk = torch.zeros(nq, dtype=torch.int64, device="cuda")
    _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, pos.data_ptr(), nq, chk.data_ptr())); torch.cuda.synchronize()
    ok = bool((chk == qr).all().item()) and bool(found.all().item())
    # the same rank batch with the queries sorted by position (every bit line is then touched by neighbours in time):
    # what bucketing the batch by block index could gain at most, next to what sorting costs
    qs, _ = torch.sort(qn)
    sorted_ms = event_avg_ms(lambda: _ffi.check(L.bmx_rank_batch_dev(ctx._h, v._h, rs._h, qs.data_ptr(), nq, out.data_ptr())), 5, ctx)
    sort_ms = event_avg_ms(lambda: torch.sort(qn), 3, ctx)
    # select: the same batch with the ranks in ascending order ("every k-th element", cursor-style enumeration: neighbours in the
    # batch share lines), and the round-4 kernel (global directory, two dependent reads) next to the LDS-directory one
    qrs, _ = torch.sort(qr)
    env_int = lambda k, d: int(os.environ.get(k, d))
    keep_tuning = {"rs_sorted_hint": 0, "rs_select_top": env_int("BMX_RS_SELECT_TOP", -1), "rs_select_sel": env_int("BMX_RS_SELECT_SEL", -1)}
    try:
        ctx.set_tuning("rs_sorted_hint", 1)
        sel_sorted_ms = event_avg_ms(lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qrs.data_ptr(), nq, pos.data_ptr(), found.data_ptr())), 5, ctx)
        ctx.set_tuning("rs_sorted_hint", 0)
        # the kernels of the earlier rounds in the same run, through the same index: select lines off -> the rank lines' directory
        # with its summary in LDS (round 5, k_select_top<2>); that off too -> the directory in global memory (round 4, k_select_sdir<4>)
        ctx.set_tuning("rs_select_sel", 0)
        sel_top_ms = event_avg_ms(do_sel, 5, ctx)
        ctx.set_tuning("rs_select_top", 0)
        sel_sdir_ms = event_avg_ms(do_sel, 5, ctx)
    finally:
        for k_, v_ in keep_tuning.items(): ctx.set_tuning(k_, v_)
    sel_by_batch = {}
    for nqq in (100_000, 1_000_000, nq):
        if nqq <= nq:
            sel_by_batch[str(nqq)] = round(event_avg_ms(lambda: _ffi.check(L.bmx_select_batch_dev(ctx._h, v._h, rs._h, qr.data_ptr(), nqq, pos.data_ptr(), found.data_ptr())), 5, ctx), 4)
    # the bound: random 128-byte lines per second this box gathers from a buffer as large as the vector's bit slab
    info = v.info()
    slab_bytes = max(info["counts"][2] * 8192, 1 << 20)
    pm = C.c_float()
    _ffi.check(L.bmx_probe_random_lines(ctx._h, slab_bytes, nq, 5, C.byref(pm)))
    ceil_lines_s = nq / (pm.value * 1e-3)
    rank_lines_s = nq / (rank_ms * 1e-3)
    sel_lines_s = nq / (sel_ms * 1e-3)
    rank_kernel = ("k_rank_lines<2> (the vector laid out as rank lines by build_rs_index: count before the line + 960 bits per 128-B line)"
                   if rs.info()["has_lines"] and os.environ.get("BMX_RS_LANES", "0") != "8" else "k_rank_l / k_rank (descriptor + running count + cumulative row + bit line)")
    traffic, tsrc, tj = (traffic_file("traffic_config3.json" if dq == 6554 else "traffic_config3_1pct.json", kernel=rank_kernel)
                         if (nbits == NBITS_4G and dq in (6554, 655) and nq == 10_000_000) else (None, None, {}))
    rsi = rs.info()
    sel_kernel = _select_kernel_name(rsi)
    straffic, stsrc, stj = (traffic_file("traffic_config3_select.json" if dq == 6554 else "traffic_config3_1pct_select.json", kernel=sel_kernel)
                            if (nbits == NBITS_4G and dq in (6554, 655) and nq == 10_000_000) else (None, None, {}))
    pct = dq / 65536 * 100
    res = {"metric": "M queries/s, rank + select (bmrs.h RS-index) on one 4e9-bit vector",
           "value": round(2 * nq * steps / dt / 1e6, 1), "unit": "Mqueries/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"{nq} random rank(n) + {nq} random select(r) per step on a {nbits}-bit vector, Bernoulli {pct:.3g}% (density q16 {dq})",
                      "baseline_config": "configs[3]", "block_types": v.calc_stat(), "count": cnt,
                      "rs_build_ms": round(build_ms, 4), "rank_ms": round(rank_ms, 4), "select_ms": round(sel_ms, 4),
                      "hbm_resident_bytes": ctx.mem_used(),
                      "rank_Mq_s": round(nq / rank_ms / 1e3, 1), "select_Mq_s": round(nq / sel_ms / 1e3, 1),
                      "rank_select_roundtrip_ok": ok,
                      "rank_ms_sorted_queries": round(sorted_ms, 4), "sort_ms_torch": round(sort_ms, 4),
                      "select_ms_sorted_ranks": round(sel_sorted_ms, 4), "select_ms_lds_directory_kernel": round(sel_top_ms, 4),
                      "select_ms_global_directory_kernel": round(sel_sdir_ms, 4),
                      "rs_index": {"bytes": rsi["bytes"], "rank_lines": rsi["has_lines"], "select_offset_bits": rsi["select_offset_bits"],
                                   "select_lines_bytes": rsi["select_lines_bytes"]},
                      "select_ms_by_batch": sel_by_batch,
                      "bucketing_note": "rank over the same queries pre-sorted by position vs the cost of sorting them (torch.sort): "
                                        "bucketing a batch by block pays only if sort + sorted run < the unsorted run"},
           "roofline": {"bound": "hbm", "achieved": round(rank_lines_s / 1e9, 3), "peak": round(ceil_lines_s / 1e9, 3),
                        "unit": "G lines/s (random 128-byte lines)", "frac": round(rank_lines_s / ceil_lines_s, 4),
                        "traffic": traffic, "traffic_source": tsrc,
                        "kernel": rank_kernel,
                        "algorithmic_bytes_per_launch": nq * 128, "avg_launch_ms": round(rank_ms, 4),
                        "peak_source": f"bmx_probe_random_lines in this run: {nq} random 128-B lines (8 lanes x 16 B, the access shape of "
                                       f"a rank query's bit line) over a {slab_bytes / 1e6:.0f} MB buffer in {pm.value:.4f} ms",
                        "select": {"kernel": sel_kernel, "achieved": round(sel_lines_s / 1e9, 3), "frac": round(sel_lines_s / ceil_lines_s, 4),
                                   "avg_launch_ms": round(sel_ms, 4),
                                   "note": "queries per second against the same random-line rate.  With select lines (round 6) a select reads ONE "
                                           "128-byte line that holds the answer -- no guess, no retry; the round-5 kernel (rank lines + directory "
                                           "summary in LDS: a second line whenever the interpolated guess is off, 1.8 missed lines per query by PMC) "
                                           "and the round-4 kernel (directory in global memory) are timed beside it through the same index "
                                           "(select_ms_lds_directory_kernel, select_ms_global_directory_kernel)"},
                        "select_traffic": straffic, "select_traffic_source": stsrc,
                        "select_lines_per_query": (round(stj["tcc_miss_per_launch"] / nq, 3) if straffic and stj.get("tcc_miss_per_launch") else None),
                        "rank_lines_per_query": (round(tj["tcc_miss_per_launch"] / nq, 3) if traffic and tj.get("tcc_miss_per_launch") else None),
                        "as_bandwidth_GBps": round(nq * 128 / rank_ms / 1e6, 1),
                        "frac_bytes": round(nq * 128 / rank_ms / 1e6 / HBM_PEAK_GBS, 4),
                        "select_frac_bytes": round(nq * 136 / sel_ms / 1e6 / HBM_PEAK_GBS, 4),
                        "frac_bytes_note": "the same times against the 8 TB/s streaming peak: 128 B per rank query, 128 + 8 B per select query",
                        "note": "random access: ONE 128-B line per rank query is what the algorithm needs, and with the rank-line layout "
                                "(running count interleaved with the bits) it is also all the kernel reads; the bound is the transaction "
                                "rate of random lines (SURVEY section 8(d)), measured by the probe, not the 8 TB/s streaming peak"}}
    if not args.no_cpu:
        try:
            P, orc, kind = _pick_oracle()
            sbits = 512 * 65536
            w = P.gen_words(SEED, 7, dq, nbits, word_off=0, nwords=512 * 2048)
            hv = orc.import_words(w, True, sbits); hrs = orc.rs_build(hv)
            rng = np.random.default_rng(1)
            q = rng.integers(0, sbits, 1_000_000, dtype=np.uint64)
            t0 = time.perf_counter(); r = hrs.rank(q); d1 = time.perf_counter() - t0
            rr = rng.integers(1, hrs.count() + 1, 1_000_000, dtype=np.uint64)
            t0 = time.perf_counter(); hrs.select(rr); d2 = time.perf_counter() - t0
            cpu = {"value": round(2e6 / (d1 + d2) / 1e6, 2), "unit": "Mqueries/s", "cores": 1, "kind": kind, "impl": orc.name,
                   "sample": "1 M rank + 1 M select on the first 512 blocks (cache-friendlier than the 4e9-bit vector)",
                   "rank_Mq_s": round(1.0 / d1, 2), "select_Mq_s": round(1.0 / d2, 2)}
            if not args.no_allcores:
                # full-size check: total count + sampled rank / select answers of the reference over the WHOLE vector
                ncores = args.cpu_cores or len(os.sched_getaffinity(0))
                ns = 20000
                sn = rng.integers(0, nbits, ns, dtype=np.uint64)
                total, sn, an, sr, ar = cpu_rank_allcores(7, dq, nbits, ncores, sn,
                                                          lambda tot: rng.integers(1, tot + 1, ns, dtype=np.uint64))
                g_rank = v.count_to(sn, rs)
                g_found, g_pos = v.select(sr, rs)
                cpu.update({"full_count": total, "sampled_queries": 2 * ns, "cores_used": min(ncores, (nbits + 65535) // 65536),
                            "allcores_sample": f"the whole {nbits}-bit vector indexed by block range over the host cores; total count + "
                                               f"{ns} rank + {ns} select answers compared with the GPU's",
                            "matches_gpu_full": bool(total == cnt and (np.asarray(g_rank) == an).all()
                                                     and np.asarray(g_found).all() and (np.asarray(g_pos) == ar).all())})
            res["cpu_baseline"] = cpu
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Mqueries/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    del rs, v
    ctx.trim()
    return res


# ----------------------------------------------------------------------------------------------------
# configs[4]: combine_or over 4096 x 4e9-bit sparse vectors, block-range sharded over the ranks (strong)
# ----------------------------------------------------------------------------------------------------
def run_or_sharded(args, env, quick=False):
    """configs[4].  Three numbers, kept apart:
      cold_ms  -- the timed region: aggregator::combine_or over the operand list as the reference holds it (4096 separate
                  GAP-block vectors), no packed collection in force; `value`, `ms_per_step` and `roofline` are THIS call, on
                  the algorithmic bytes of SURVEY section 8(d): 2 x (len + 1) B per GAP operand block + 8,192 B per stored block
      build_ms -- bmx_collection_prepare(vecs, ROLE_OR): the column-major packed collection of the set (+ member directory)
      warm_ms  -- the same call once the collection is in force; its own roofline on the bytes the collection holds
    plus what a subset of the collection's vectors costs through the member directory and through the cold path."""
    import numpy as np
    import bitmagic_amd as bm
    torch, dist, ctx = env.torch, env.dist, env.ctx
    world, rank, use_dist = env.world, env.rank, env.use_dist
    nbits, nvec, dq = NBITS_4G, args.or_vecs, 13                       # 13/65536 = 0.02 %
    nblocks = (nbits + 65535) // 65536
    lo, hi = bm.shard_range(nblocks, rank, world)
    t0 = time.perf_counter()
    vecs = [bm.bvector.generate(ctx, SEED, 10000 + i, dq, nbits, block_range=(lo, hi) if world > 1 else None) for i in range(nvec)]
    ctx.synchronize(); t_build = time.perf_counter() - t0
    gap_bytes = sum(v.operand_bytes() for v in vecs)                      # exact: 2 x (len + 1) per GAP block (no slab padding)
    cnt = torch.zeros(1, dtype=torch.int64, device="cpu" if env.one_dev else "cuda")
    last = []
    # the operand pointer array is built ONCE, as a C caller holding `const bmx_vec*[]` would: re-marshalling 4096 Python
    # objects into a ctypes array per call costs 0.2-0.3 ms of interpreter time that is not the library's
    import ctypes as C
    from bitmagic_amd import _ffi
    L = _ffi.lib()
    arr = (C.c_void_p * max(len(vecs), 1))(*[v._h for v in vecs])
    def call(a, n):
        h = C.c_void_p()
        _ffi.check(L.bmx_agg_or_opt(ctx._h, a, n, 0, C.byref(h)))          # aggregator::combine_or (opt_none)
        return bm.bvector(ctx, h)
    def step():
        t = call(arr, len(vecs))
        cnt.fill_(t.count())                                               # (the kernel folds the count of its result: no second pass)
        if use_dist:
            dist.all_reduce(cnt)
        last[:] = [t]
    assert ctx.pack_stats()["collections"] == 0
    steps, warmup = (10, 2) if quick else (args.steps, max(args.warmup, 1))
    dt, ev_ms = timed_region(step, steps, warmup, env)                     # ---- cold: no collection exists
    cold_ms = ev_ms / steps
    cold_count = int(cnt.item())
    result_bytes = last[0].calc_stat()["bit_blocks"] * 8192
    gb = torch.tensor([gap_bytes], dtype=torch.int64, device="cpu" if env.one_dev else "cuda")
    if use_dist:
        dist.all_reduce(gb)
    tot_bytes = int(gb.item())
    # ---- build: the packed collection of the set, then the same call again
    have_coll = os.environ.get("BMX_GAP_PACK", "-1") != "0"               # (gap_pack 0: collections are switched off altogether)
    prep_wall_ms = 0.0
    if have_coll:
        t0 = time.perf_counter(); ctx.collection_prepare(vecs, bm.ROLE_OR); ctx.synchronize(); prep_wall_ms = (time.perf_counter() - t0) * 1e3
    pack = ctx.pack_stats()
    step(); step()
    warm_ms = event_avg_ms(step, 6 if quick else 10, ctx)
    warm_count = int(cnt.item())
    # ---- a subset of the collection's vectors (half of them, shuffled): member directory vs the cold path on the same list
    sub = None
    if world == 1 and nvec >= 128 and have_coll and not args.no_subset:
        rng = np.random.default_rng(5)
        pick = rng.permutation(nvec)[: nvec // 2]
        sarr = (C.c_void_p * len(pick))(*[vecs[int(i)]._h for i in pick])
        keep = []
        def sub_call():
            keep[:] = [call(sarr, len(pick))]
        sub_call(); sub_call()
        sub_coll_ms = event_avg_ms(sub_call, 5, ctx); c1 = keep[0].count()
        ctx.set_tuning("coll_members", 1)                                 # forced: the members' pieces of the column regions
        sub_call(); sub_call()
        sub_dir_ms = event_avg_ms(sub_call, 3, ctx); c3 = keep[0].count()
        ctx.set_tuning("coll_members", -1)
        ctx.set_tuning("gap_pack", 0)
        sub_call(); sub_call()
        sub_cold_ms = event_avg_ms(sub_call, 5, ctx); c2 = keep[0].count()
        ctx.set_tuning("gap_pack", -1)
        sub = {"vectors": int(len(pick)), "default_dispatch_ms": round(sub_coll_ms, 4), "member_directory_forced_ms": round(sub_dir_ms, 4),
               "no_collection_ms": round(sub_cold_ms, 4), "same_count": bool(c1 == c2 == c3),
               "note": "a sparse OR list of >= 64 vectors takes the row kernel over the vectors' own slabs even when a collection covers it "
                       "(a member's piece of a column is ~26 B there); the member directory serves AND / SUB lists and pipelines of long lists"}
        keep.clear()
    res = None
    if rank == 0:
        ms = dt / steps * 1e3
        needed = gap_bytes + result_bytes                                  # 2 x (len + 1) per GAP operand block + 8,192 per stored block
        achieved = needed / cold_ms / 1e6
        rows = os.environ.get("BMX_OR_ROWS", "-1") != "0"
        kname = ("k_agg_or_rows<4>: tiles of 14 block columns, one coalesced row per (operand, tile) through the vectors' tile directories"
                 if rows else "k_agg_or_gap_tiled<1,1> (descriptor-table kernel)")
        traffic, tsrc, tj = traffic_file("traffic_config4.json", kernel=kname) if (world == 1 and nvec == 4096) else (None, None, {})
        wneed = pack["run_bytes"] + result_bytes
        wtraffic, wtsrc, wtj = traffic_file("traffic_config4_warm.json", kernel="k_coll_apply<OR,512>") if (world == 1 and nvec == 4096) else (None, None, {})
        res = {"metric": "Gbit/s of logical operand bits, aggregator combine_or over 4096 x 4e9-bit sparse vectors (first call, operands in the reference's format)",
               "value": round(nvec * nbits * steps / dt / 1e9, 1), "unit": "Gbit/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "u16", "data": "synthetic", "mode": env.mode,
               "config": {"workload": f"aggregator::combine_or over {nvec} x {nbits}-bit vectors at 0.02 % (all GAP blocks), result materialised + counted",
                          "baseline_config": "configs[4]", "block_types_vec0": vecs[0].calc_stat(), "blocks_per_rank": hi - lo,
                          "gap_operand_bytes_total": tot_bytes, "result_count": cold_count,
                          "result_types_rank0": last[0].calc_stat(), "build_seconds": round(t_build, 1),
                          "cold_ms": round(cold_ms, 4), "build_ms": round(pack["last_build_ms"], 2), "prepare_call_wall_ms": round(prep_wall_ms, 2),
                          "warm_ms": round(warm_ms, 4), "warm_count_equal": bool(warm_count == cold_count),
                          "break_even_calls": (round(pack["last_build_ms"] / (cold_ms - warm_ms), 1) if cold_ms > warm_ms else None),
                          "subset_of_the_collection": sub,
                          "packed_collection": {"bytes": pack["bytes"], "run_bytes": pack["run_bytes"],
                                                "note": "bytes = run entries (4 B per multi-bit run, 2 B per single-bit run) + column tables; the member "
                                                        "directory (8 B per member and column) that serves subsets and pipelines is added by the first call "
                                                        "that names only some of the members (tile build, bmx_kernels10.h)"}},
               "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                            "kernel": kname, "algorithmic_bytes_per_launch": needed, "avg_launch_ms": round(cold_ms, 4),
                            "note": "the FIRST call over the operand list (no packed collection): whole host call incl. result creation and count, "
                                    "hipEvent-timed; algorithmic bytes = 2 x (len + 1) B per GAP operand block + 8,192 B per stored result block "
                                    "(SURVEY section 8(d))",
                            "warm": {"kernel": "k_coll_apply<OR,512>: one workgroup per block column over the prepared packed collection",
                                     "avg_launch_ms": round(warm_ms, 4), "algorithmic_bytes_per_launch": wneed,
                                     "achieved": round(wneed / warm_ms / 1e6, 1), "frac": round(wneed / warm_ms / 1e6 / HBM_PEAK_GBS, 4),
                                     "traffic": wtraffic, "traffic_source": wtsrc,
                                     "reference_format_GBps": round(gap_bytes / warm_ms / 1e6, 1),
                                     "note": "after bmx_collection_prepare: the bytes are the run lists as the collection keeps them (a single-bit "
                                             "run is 2 B where the reference's GAP block spends 4 B); reference_format_GBps = the same time against "
                                             "the 2 x (len + 1) B of section 8(d) -- above the HBM peak, which is what re-coding buys, not a roofline"}}}
        if not args.no_cpu and world == 1:
            try:
                P, orc, kind = _pick_oracle()
                sb, nv = 64, min(nvec, 1024)
                hv = [orc.import_words(P.gen_words(SEED, 10000 + i, dq, nbits, word_off=0, nwords=sb * 2048), True, sb * 65536) for i in range(nv)]
                t0 = time.perf_counter(); r = orc.agg_or(hv); d = time.perf_counter() - t0
                cpu = {"value": round(nv * sb * 65536 / d / 1e9, 1), "unit": "Gbit/s", "cores": 1, "kind": kind, "impl": orc.name,
                       "sample": f"combine_or over {nv} vectors x first {sb} blocks, one pass", "count": r.count()}
                del hv
                if not args.no_allcores:
                    # bounded full-width check: ALL nvec vectors, a spread sample of block columns, against the GPU result's
                    # count over exactly those columns
                    ncores = args.cpu_cores or len(os.sched_getaffinity(0))
                    nsample = min(nblocks, max(8, min(8 * ncores, 2048)))     # (0.25 s per column and core: 8 columns per worker)
                    blocks = sorted(set(int(x) for x in np.linspace(0, nblocks - 1, nsample)))
                    ref, secs, _k, _i, used = cpu_or_sample(nvec, dq, nbits, blocks, ncores)
                    t = last[0]
                    trs = t.build_rs_index()
                    l = np.asarray(blocks, dtype=np.uint64) * np.uint64(65536)
                    r_ = np.minimum(l + np.uint64(65535), np.uint64(nbits - 1))
                    got = t.count_range(l, r_, trs)
                    cpu.update({"sampled_block_columns": len(blocks), "cores_used": used,
                                "allcores_sample": f"combine_or over ALL {nvec} vectors on {len(blocks)} block columns spread over the range "
                                                   f"({(len(blocks) + used - 1) // used} column(s) per worker process, {secs:.1f} s), compared with count_range of the GPU result",
                                "matches_gpu_sample": bool([int(x) for x in got] == [ref[b] for b in blocks])})
                res["cpu_baseline"] = cpu
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    last.clear()
    del vecs
    ctx.trim()
    return res


def run_sparse_and(args, env, dq=197, quick=True):
    """configs[2] at inverted-index densities (every block of every vector a GAP block): the 256-way fused AND+COUNT, three
    numbers kept apart as for configs[4]:
      cold_ms  -- the counts pipeline over the operands as the reference holds them, NO packed collection in force
                  (k_agg_and_rows, bmx_kernels9.h); `value`, `ms_per_step` and `roofline` are this run, on SURVEY 8(d) bytes
      build_ms -- bmx_collection_prepare(vecs, ROLE_AND)
      warm_ms  -- the same pipeline once the collection is in force (k_coll_apply<AND_COUNT>), its own roofline"""
    import bitmagic_amd as bm
    torch, ctx = env.torch, env.ctx
    nvec, nbits = args.nvec, args.nbits
    nblocks = (nbits + 65535) // 65536
    t0 = time.perf_counter()
    vecs = [bm.bvector.generate(ctx, SEED, v, dq, nbits, with_common=True) for v in range(nvec)]
    ctx.synchronize(); t_build = time.perf_counter() - t0
    if any(v.info()["counts"][bm.BIT] for v in vecs):
        raise RuntimeError(f"density {dq}/65536 leaves bit-blocks: not the GAP-only case")
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")
    agg = bm.aggregator(ctx)
    pipe = bm.aggregator.pipeline(ctx)
    ag = pipe.add()
    for v in vecs:
        ag.add(v, 0)
    pipe.complete()
    alg = pipe.operand_bytes()
    assert ctx.pack_stats()["collections"] == 0
    def kernel():
        agg.run_counts_dev(pipe, counts.data_ptr())
    steps, warmup = (10, 2) if quick else (args.steps, max(args.warmup, 1))
    dt, ev_ms = timed_region(kernel, steps, warmup, env)                   # ---- cold: no collection exists
    cold_count = int(counts.item())
    cold_plan = pipe.describe()
    cold_ms = event_avg_ms(kernel, steps, ctx)
    assert ctx.pack_stats()["collections"] == 0
    have_coll = os.environ.get("BMX_GAP_PACK", "-1") != "0"
    build_ms = warm_ms = warm_plan = warm_count = None
    if have_coll:
        ctx.collection_prepare(vecs, bm.ROLE_AND); ctx.synchronize()
        pack = ctx.pack_stats(); build_ms = pack["last_build_ms"]
        kernel(); kernel()
        warm_ms = event_avg_ms(kernel, steps, ctx)
        warm_count = int(counts.item()); warm_plan = pipe.describe()
    achieved = alg / cold_ms / 1e6
    pct = dq / 65536 * 100
    sp_traffic, sp_tsrc, _ = (traffic_file("traffic_config2_dq%d.json" % dq, kernel=cold_plan) if (nvec == 256 and nbits == NBITS_1G and dq in (197, 66)) else (None, None, {}))
    res = {"metric": METRIC + f" -- at {pct:.1f} % (GAP-only operands), first call", "value": round(nvec * nbits * steps / dt / 1e9, 1), "unit": "Gbit/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u16", "data": "synthetic", "mode": env.mode,
           "config": {"workload": f"{nvec}-way fused AND+COUNT over {nvec} x {nbits}-bit vectors, common {pct:.2f} % + own {pct:.2f} % (data set A), every block a GAP block; "
                                  "counts pipeline over the operands in the reference's format, no packed collection",
                      "baseline_config": "configs[2] (sparse)", "density_q16": dq, "block_types_vec0": vecs[0].calc_stat(), "result_count": cold_count,
                      "build_seconds": round(t_build, 1), "cold_ms": round(cold_ms, 4),
                      "build_ms": None if build_ms is None else round(build_ms, 2), "warm_ms": None if warm_ms is None else round(warm_ms, 4),
                      "warm_count_equal": None if warm_count is None else bool(warm_count == cold_count),
                      "break_even_calls": (round(build_ms / (cold_ms - warm_ms), 1) if (warm_ms is not None and cold_ms > warm_ms) else None)},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": sp_traffic, "traffic_source": sp_tsrc, "kernel": cold_plan, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(cold_ms, 4),
                        "note": "algorithmic bytes = 2 x (len + 1) B per GAP operand block (SURVEY section 8(d)); hipEvent pair around back-to-back runs"}}
    if warm_ms is not None:
        res["roofline"]["warm"] = {"kernel": warm_plan, "avg_launch_ms": round(warm_ms, 4), "achieved": round(alg / warm_ms / 1e6, 1),
                                   "frac": round(alg / warm_ms / 1e6 / HBM_PEAK_GBS, 4), "reference_format_GBps": round(alg / warm_ms / 1e6, 1)}
    if not args.no_cpu:
        try:
            sb = min(args.cpu_sample_blocks, nblocks)
            procs = _spawn_workers([["and", 0, sb, nvec, dq, nbits, 3]])
            try:
                _go(procs); one = json.loads(procs[0].stdout.readline())
            finally:
                _finish_workers(procs)
            best = min(b - a for a, b in one["spans"])
            cpu = {"value": round(nvec * sb * 65536 / best / 1e9, 1), "unit": "Gbit/s", "cores": 1, "kind": one["kind"], "impl": one["impl"],
                   "sample": f"{nvec} vectors x first {sb} blocks, counts-only pipeline, best of {len(one['spans'])} passes",
                   "matches_gpu": bool(one["count"] == int(agg._run_pipeline(pipe, 0, sb)[0]))}
            if not args.no_allcores:
                ncores = args.cpu_cores or len(os.sched_getaffinity(0))
                cpu.update(cpu_baseline_allcores(nvec, dq, nbits, ncores, reps=2))
                cpu["matches_gpu_full"] = bool(cpu["full_count"] == cold_count)
            res["cpu_baseline"] = cpu
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 1, "kind": "reference", "sample": f"failed: {e}"}
    del pipe, vecs
    ctx.trim()
    return res


def run_dataset_b(args, env):
    """SURVEY section 8(d) config 3, data set B: 256 INDEPENDENT 10 % vectors -- the AND of every block column is empty after ~8 operands
    and the reference leaves the column there (digest -> 0, src/bmaggregator.h:1994-2120, :2052); the kernel tests its running
    result after every batch of operands and stops the column the same way.  Reported as time and Gbit/s of LOGICAL operand bits
    (what the call covers); there is no bandwidth figure -- almost none of the bytes are read, that is the point."""
    import bitmagic_amd as bm
    torch, ctx = env.torch, env.ctx
    nvec, nbits, dq = 256, NBITS_1G, 6554
    vecs = [bm.bvector.generate(ctx, SEED, v, dq, nbits, with_common=False) for v in range(nvec)]
    pipe = bm.aggregator.pipeline(ctx)
    ag = pipe.add()
    for v in vecs: ag.add(v, 0)
    pipe.complete(); ctx.synchronize()
    agg = bm.aggregator(ctx)
    counts = torch.zeros(1, dtype=torch.int64, device="cuda")
    kernel = lambda: agg.run_counts_dev(pipe, counts.data_ptr())
    dt, ev_ms = timed_region(kernel, 20, 3, env)
    total = int(counts.item())
    ms = ev_ms / 20
    res = {"metric": "Gbit/s of logical operand bits, 256-way fused AND+COUNT on 1e9-bit vectors, data set B (independent operands: early exit)",
           "value": round(nvec * nbits / ms / 1e6, 1), "unit": "Gbit/s", "n_gpus": 1, "steps": 20, "warmup": 3, "ms_per_step": round(dt / 20 * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"aggregator::combine_and + count over {nvec} x {nbits}-bit INDEPENDENT vectors at 10 % (data set B): every column's AND is empty after a few operands",
                      "baseline_config": "configs[2], data set B", "result_count": total, "kernel_plan": pipe.describe(),
                      "operand_bytes_if_nothing_exited": pipe.operand_bytes()},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": pipe.describe()[:80],
                        "avg_launch_ms": round(ms, 4),
                        "note": "no bandwidth figure: the call reads a few operands per column and leaves; `value` counts the logical bits the call covers"}}
    if not args.no_cpu:
        try:
            ncores = args.cpu_cores or len(os.sched_getaffinity(0))
            full = cpu_baseline_allcores(nvec, dq, nbits, ncores, reps=3, with_common=False)
            res["cpu_baseline"] = {"value": full["allcores_gbit_s"], "unit": "Gbit/s", "cores": full["cores_used"], "kind": full["allcores_kind"], "impl": full["allcores_impl"],
                                   "sample": full["allcores_sample"], "full_count": full["full_count"], "matches_gpu_full": bool(full["full_count"] == total),
                                   "allcores_gbit_s": full["allcores_gbit_s"], "cores_used": full["cores_used"]}
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "Gbit/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    del pipe, vecs
    ctx.trim()
    return res


def run_plumbing(args, env):
    """BASELINE configs[0]: two 1 M-bit bm::bvector<> at 10 %: bit_and + count on the CPU reference with AVX2 OFF (the scalar
    build of the unmodified BitMagic, oracle/_ref/libbmref_scalar.so) -- the reference's own tests/perf shape
    (AndCountTest, tests/perf/perf.cpp:2281) -- next to the same two calls through the GPU engine (16 blocks: launch-bound,
    reported for completeness; the numbers that matter at this size are equal results)."""
    import bitmagic_amd as bm
    import oracle
    ctx = env.ctx
    nbits, dq = 1_000_000, 6554
    P = oracle.port()
    have_scalar = oracle.have_reference("scalar")
    orc = oracle.reference("scalar") if have_scalar else P
    wa, wb = P.gen_words(SEED, 1, dq, nbits), P.gen_words(SEED, 2, dq, nbits)
    ha, hb = orc.import_words(wa, True, nbits), orc.import_words(wb, True, nbits)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        r = orc.op2(0, ha, hb, 0); c_cpu = r.count()
    d_and = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): c2 = orc.count_op2(0, ha, hb)
    d_cnt = (time.perf_counter() - t0) / reps
    ga, gb = bm.bit_import_u32(ctx, wa, True), bm.bit_import_u32(ctx, wb, True)
    def step():
        t = bm.bvector.bit_and(ga, gb); step.c = t.count()
    steps, warmup = max(args.steps, 20), max(args.warmup, 3)
    dt, ev_ms = timed_region(step, steps, warmup, env)
    ob = ga.operand_bytes() + gb.operand_bytes()
    ms = dt / steps * 1e3
    # the same as ONE host call (bmx_op2_count: result vector + its count), and the count alone (bm::count_and: no result)
    def step1():
        t, step1.c = bm.bvector.op2_count(bm.AND, ga, gb)
    def step0():
        step0.c = bm.count_and(ga, gb)
    dt1, _ = timed_region(step1, steps, warmup, env)
    dt0, _ = timed_region(step0, steps, warmup, env)
    res = {"metric": "Gbit/s of operand bits, bit_and + count on two 1M-bit vectors (plumbing case)", "value": round(2 * nbits * steps / dt / 1e9, 3),
           "unit": "Gbit/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
           "config": {"workload": "two 1,000,000-bit vectors at 10 % (16 blocks each): bvector::bit_and (3-operand) + count()", "baseline_config": "configs[0]",
                      "block_types_vec0": ga.calc_stat(), "gpu_count": int(step.c), "cpu_count": int(c_cpu), "counts_equal": bool(step.c == c_cpu == c2 == step1.c == step0.c),
                      "one_call_op2_count_ms": round(dt1 / steps * 1e3, 4), "count_and_only_ms": round(dt0 / steps * 1e3, 4)},
           "roofline": {"bound": "hbm", "achieved": round(ob / (ev_ms / steps) / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ob / (ev_ms / steps) / 1e6 / HBM_PEAK_GBS, 6), "traffic": None, "kernel": "k_op2 (16 waves; folds the result's block kinds and its popcount: one launch, one synchronise; count() finds the count with the vector)",
                        "algorithmic_bytes_per_launch": int(ob), "avg_launch_ms": round(ev_ms / steps, 4),
                        "note": "two host calls (Python -> C-ABI) over 16 blocks: launch + synchronise latency, not bandwidth; configs[0] is the CPU-runnable plumbing case"},
           "cpu_baseline": {"value": round(2 * nbits / d_and / 1e9, 2), "unit": "Gbit/s", "cores": 1, "kind": "reference" if have_scalar else "port",
                            "impl": orc.name, "sample": f"bit_and (3-operand, new result vector) + count(), {reps} repetitions; AVX2 off (scalar build)",
                            "count_and_only_Gbit_s": round(2 * nbits / d_cnt / 1e9, 2), "us_per_bit_and_plus_count": round(d_and * 1e6, 2)}}
    return res


def run_or_group(args):
    """configs[4] through the product's device group: plain `bench.py --config 4 --gpus N`"""
    import torch
    import bitmagic_amd as bm
    n = args.gpus
    one_dev = os.environ.get(ONE_DEV_HOOK) == "1"
    have = torch.cuda.device_count()
    if not one_dev and have < n:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible devices, found {have}\n")
        sys.exit(2)
    devices = [0] * n if one_dev else list(range(n))
    grp = bm.group(devices, bm.GROUP_HOST_SUM)
    nbits, nvec, dq = (args.nbits if args.nbits != NBITS_1G else NBITS_4G), args.or_vecs, 13
    t0 = time.perf_counter()
    vecs = [bm.gbvector.generate(grp, SEED, 10000 + i, dq, nbits) for i in range(nvec)]
    for d in sorted(set(devices)): torch.cuda.synchronize(d)
    t_build = time.perf_counter() - t0
    gap_bytes = sum(v.info()["gap_words"] for v in vecs) * 2
    member_bytes = [0] * n                                        # what every member holds of the operands: the balance of the cut
    for v in vecs[:: max(1, nvec // 64)]:
        for m in range(n):
            member_bytes[m] += v.shard_info(m)["gap_words"] * 2
    import ctypes as C
    from bitmagic_amd import _ffi
    L = _ffi.lib()
    arr = (C.c_void_p * max(len(vecs), 1))(*[v._h for v in vecs])       # built once, as a C caller's pointer array would be
    keep = []
    def step():
        h = C.c_void_p()
        _ffi.check(L.bmx_gagg_or(grp._h, arr, len(vecs), 0, C.byref(h)))
        t = bm.gbvector(grp, h)
        keep[:] = [t, t.count()]
    for _ in range(args.warmup): step()
    for d in sorted(set(devices)): torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps): step()
    for d in sorted(set(devices)): torch.cuda.synchronize(d)
    dt = time.perf_counter() - t0
    ms = dt / args.steps * 1e3
    res = {"metric": "Gbit/s of logical operand bits, aggregator combine_or over 4096 x 4e9-bit sparse vectors",
           "value": round(nvec * nbits * args.steps / dt / 1e9, 1), "unit": "Gbit/s", "n_gpus": n, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u16", "data": "synthetic", "mode": "group", "exchange": "host_sum", "rccl_ranks": 0, "rccl_error": None,
           "rccl_requested": False,
           "config": {"workload": f"aggregator::combine_or over {nvec} x {nbits}-bit vectors at 0.02 % (all GAP blocks), result materialised (sharded) + counted",
                      "baseline_config": "configs[4]", "devices": devices, "gap_operand_bytes_total": gap_bytes,
                      "result_count": int(keep[1]), "build_seconds": round(t_build, 1),
                      "member_block_ranges": [grp.shard_range((nbits + 65535) // 65536, m) for m in range(n)],
                      "member_gap_bytes": member_bytes},
           "roofline": {"bound": "hbm", "achieved": round(gap_bytes / ms / 1e6, 1), "peak": HBM_PEAK_GBS * n, "unit": "GB/s",
                        "frac": round(gap_bytes / ms / 1e6 / (HBM_PEAK_GBS * n), 4), "traffic": None,
                        "kernel": "bmx_gagg_or: bmx_agg_or per member on persistent workers",
                        "algorithmic_bytes_per_launch": gap_bytes, "avg_launch_ms": round(ms, 4),
                        "note": "whole host call over all members (result creation, layout scan, count); peak = n x 8 TB/s"}}
    keep.clear(); del vecs
    grp.close()
    return res


def summary_of(res):
    """one-line summary of another config's result for the headline line's "other_configs" """
    if res is None:
        return None
    out = {"metric": res["metric"], "value": res["value"], "unit": res["unit"], "ms_per_step": res["ms_per_step"],
           "steps": res["steps"], "workload": res["config"]["workload"],
           "roofline": {k: res["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "traffic", "frac_bytes", "select_frac_bytes", "select_traffic", "select_lines_per_query", "rank_lines_per_query") if k in res["roofline"] or k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "traffic")}}
    cpu = res.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "matches_gpu", "allcores_gbit_s", "cores_used", "matches_gpu_full", "matches_gpu_full_materialised", "matches_gpu_sample") if k in cpu}
    for k in ("per_op", "rank_ms", "select_ms", "select_ms_sorted_ranks", "select_ms_lds_directory_kernel", "select_ms_global_directory_kernel", "rs_build_ms", "rs_index", "rank_Mq_s", "select_Mq_s", "rank_select_roundtrip_ok", "result_count", "cold_ms", "build_ms", "warm_ms",
              "break_even_calls", "subset_of_the_collection", "own_read_write_probe"):
        if k in res["config"]:
            out[k] = res["config"][k]
    if "select" in res["roofline"]:
        out["roofline"]["select_frac"] = res["roofline"]["select"]["frac"]
        out["roofline"]["select_kernel"] = res["roofline"]["select"]["kernel"][:24]
    if "warm" in res["roofline"]:
        out["roofline"]["warm"] = {k: res["roofline"]["warm"].get(k) for k in ("kernel", "avg_launch_ms", "achieved", "frac", "reference_format_GBps")}
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[0, 1, 2, 3, 4], help="BASELINE.json configs[] index (2 = the headline)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"])
    ap.add_argument("--launcher", default="auto", choices=["auto", "group", "torchrun"],
                    help="--gpus N > 1 without RANK in the environment: 'group' (default) = one process over the product's "
                         "device group (bmx_group + RCCL); 'torchrun' = re-execute under torch.distributed.run")
    ap.add_argument("--group-exchange", default="rccl", choices=["rccl", "host"], help="group mode: in-library ncclAllReduce or host sum")
    ap.add_argument("--nvec", type=int, default=256)
    ap.add_argument("--nbits", type=int, default=NBITS_1G)
    ap.add_argument("--density-q16", type=int, default=6554)      # 10 %
    ap.add_argument("--independent", action="store_true", help="data set B (no common part; early exit)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=512)
    ap.add_argument("--cpu-cores", type=int, default=0, help="all-cores baseline: processes to use (0 = every core of the affinity mask)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-allcores", action="store_true")
    ap.add_argument("--no-weak", action="store_true")
    ap.add_argument("--no-subset", action="store_true", help="config 4: skip the subset-of-the-collection line (PMC passes: full-size launches only)")
    ap.add_argument("--no-prepare", action="store_true", help="GAP-only collections: do not prepare the packed collection (descriptor-table kernels)")
    ap.add_argument("--no-shard-probe", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="headline run: skip the one-line summaries of configs 1, 3, 4")
    ap.add_argument("--pairs", type=int, default=6)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--or-vecs", type=int, default=4096)
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if args.gpus > 1 and not launched:
        if args.launcher == "torchrun":
            return reexec_torchrun(args)
        if args.config == 2:
            res = run_headline_group(args)
            print(json.dumps(res))
            # the in-library all-reduce was asked for and did not come up over all N ranks: say so with the exit status too
            # (the line above still carries the host-sum measurement, its `exchange`, `rccl_ranks` and `rccl_error`)
            if res.get("rccl_requested") and res.get("rccl_ranks") != args.gpus:
                sys.stderr.write(f"bench.py: --group-exchange rccl: {res.get('rccl_ranks')} RCCL ranks for --gpus {args.gpus} ({res.get('rccl_error')})\n")
                sys.exit(3)
            return
        if args.config == 4:
            print(json.dumps(run_or_group(args))); return
        sys.stderr.write(f"bench.py: --config {args.config} is a one-GPU configuration; --gpus {args.gpus} is not supported for it\n")
        sys.exit(2)
    if launched and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}\n")
        sys.exit(2)
    if launched and args.gpus > 1 and args.config in (0, 1, 3):
        sys.stderr.write(f"bench.py: --config {args.config} is a one-GPU configuration\n")
        sys.exit(2)
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1
    cpu = None
    if args.config == 2 and single and not args.no_cpu:
        cpu = headline_cpu(args)                         # before torch / HIP are loaded into this process
    env = Env(args).setup()
    if args.config == 0:
        res = run_plumbing(args, env)
    elif args.config == 1:
        res = run_pairwise(args, env)
    elif args.config == 3:
        res = run_rank_select(args, env)
    elif args.config == 4:
        res = run_or_sharded(args, env)
    else:
        res = run_headline(args, env, cpu)
        standard = (args.nvec == 256 and args.nbits == NBITS_1G and args.density_q16 == 6554 and not args.independent)
        if res is not None and single and standard and not args.no_others:
            others = {}
            for name, fn in (("configs[0]", lambda: run_plumbing(args, env)),
                             ("configs[1]", lambda: run_pairwise(args, env, quick=True)),
                             ("configs[1] at 1 %", lambda: run_pairwise(args, env, dq=655, quick=True)),
                             ("configs[1] at 50 %", lambda: run_pairwise(args, env, dq=32768, quick=True)),
                             ("configs[3]", lambda: run_rank_select(args, env, quick=True)),
                             ("configs[3] at 1 %", lambda: run_rank_select(args, env, quick=True, dq=655)),
                             ("configs[4]", lambda: run_or_sharded(args, env, quick=True)),
                             ("configs[2] data set B", lambda: run_dataset_b(args, env)),
                             ("configs[2] at 0.3 %", lambda: run_sparse_and(args, env, dq=197)),
                             ("configs[2] at 0.1 %", lambda: run_sparse_and(args, env, dq=66))):
                try:
                    t0 = time.perf_counter()
                    s = summary_of(fn())
                    s["wall_s"] = round(time.perf_counter() - t0, 1)
                    others[name] = s
                except Exception as e:                   # a failing side leg must not take the headline line with it
                    others[name] = {"error": str(e)}
            res["other_configs"] = others
    if res is not None:
        print(json.dumps(res))
    env.finish()


if __name__ == "__main__":
    main()
