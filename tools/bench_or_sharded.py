#!/usr/bin/env python3
"""BASELINE configs[4]: aggregator combine_or over 4096 x 4e9-bit sparse vectors, block-range sharded over the
GPUs of one node: rank r holds blocks shard_range(nblocks, r, world) of EVERY vector, ORs
them locally, and the only exchange is an RCCL all-reduce of the popcount.

  1 GPU :  python tools/bench_or_sharded.py
  N GPUs:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_or_sharded.py
Strong scaling: the collection is fixed, per-GPU work shrinks with N."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bitmagic_amd as bm

SEED = 0xB17A61C
ap = argparse.ArgumentParser()
ap.add_argument("--nvec", type=int, default=4096)
ap.add_argument("--nbits", type=int, default=4_000_000_000)
ap.add_argument("--density-q16", type=int, default=13)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
if use_dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = bm.context(lr, s.cuda_stream)
nblocks = (a.nbits + 65535) // 65536
lo, hi = bm.shard_range(nblocks, rank, world)
vecs = [bm.bvector.generate(ctx, SEED, 10000 + i, a.density_q16, a.nbits, block_range=(lo, hi)) for i in range(a.nvec)]
gap_bytes = sum(v.info()["gap_words"] for v in vecs) * 2
agg = bm.aggregator(ctx)
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")

def step():
    t = agg.combine_or(vecs)
    cnt.fill_(t.count())                       # local popcount of the shard result
    if use_dist: dist.all_reduce(cnt)          # the only exchange: 8 bytes
    return t

step(); torch.cuda.synchronize()
if use_dist: dist.barrier()
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
if use_dist: dist.barrier()
dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
gb = torch.tensor([gap_bytes], dtype=torch.int64, device="cuda")
if use_dist:
    dist.all_reduce(dt, op=dist.ReduceOp.MAX); dist.all_reduce(gb)
if rank == 0:
    ms = float(dt.item()) / a.steps * 1e3
    print(json.dumps({"config": 4, "workload": f"combine_or_{a.nvec}x{a.nbits}", "n_gpus": world, "scaling": "strong",
                      "blocks_per_rank": hi - lo, "ms_per_or": round(ms, 3), "operand_GB": round(int(gb.item()) / 1e9, 2),
                      "TBps_algorithmic": round(int(gb.item()) / ms / 1e9, 2), "result_count": int(cnt.item())}))
if use_dist: dist.destroy_process_group()
