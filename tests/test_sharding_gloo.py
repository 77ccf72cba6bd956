"""CPU, world_size 2 over gloo: the N>1 path = block-range shards + one all-reduce of
the per-rank popcounts.  The per-shard counts come from the oracle here (no GPU);
the sharding / reduction code is the product's (bitmagic_amd.sharding)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bitmagic_amd.sharding import allreduce_counts, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 15259, 61036):
        for w in (1, 2, 3, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port_no, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import oracle
    from cases import make_inputs
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port_no)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = oracle.port()
    words, nbits = make_inputs(P, "mixed_1pct")
    vecs = [P.import_words(w, True, nbits) for w in words]
    groups = [(vecs[:3], []), (vecs[:2], vecs[3:5]), (vecs[:1], [])]
    lo, hi = shard_range(vecs[0].nblocks, rank, world)
    local = P.pipeline_counts(groups, lo, hi)
    t = torch.from_numpy(local.astype(np.int64))
    allreduce_counts(t)
    full = P.pipeline_counts(groups)
    # strong-scaling bench layout: every rank BUILDS only its block range of every vector (bmx_vec_generate_shard /
    # gen_words with a word offset) -- the shard of a generated vector equals the generated shard
    gbits, nv = 21 * 65536 + 777, 5
    glo, ghi = shard_range(22, rank, world)
    sh = [P.import_words(P.gen_words(0xB17A61C, v, 6554, gbits, with_common=True, word_off=glo * 2048, nwords=(ghi - glo) * 2048),
                         True, max(min(gbits, ghi * 65536) - glo * 65536, 0)) for v in range(nv)]
    t2 = torch.from_numpy(P.pipeline_counts([(sh, [])]).astype(np.int64))
    allreduce_counts(t2)
    whole = [P.import_words(P.gen_words(0xB17A61C, v, 6554, gbits, with_common=True), True, gbits) for v in range(nv)]
    full2 = P.pipeline_counts([(whole, [])])
    q.put((rank, t.tolist() + t2.tolist(), [int(x) for x in full] + [int(x) for x in full2]))
    dist.destroy_process_group()


def test_two_rank_sharded_counts_match_unsharded():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port_no = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port_no, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(60)
    for rank, got, full in res:
        assert got == full, (rank, got, full)


def test_bench_allcores_cpu_baseline_protocol():
    """bench.py's all-cores CPU leg (one block-range replica per worker process, READY/go protocol): the fanned-out
    full count equals the single-process count of the same workload"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import oracle
    nvec, dq, nbits = 6, 6554, 11 * 65536 + 5
    out = bench.cpu_baseline_allcores(nvec, dq, nbits, cores=3, reps=2)
    P = oracle.port()
    vecs = [P.import_words(P.gen_words(bench.SEED, v, dq, nbits, with_common=True), True, nbits) for v in range(nvec)]
    assert out["full_count"] == int(P.pipeline_counts([(vecs, [])])[0])
    assert out["cores_used"] == 3 and out["allcores_gbit_s"] > 0
    one = bench.cpu_baseline_1core(nvec, dq, nbits, 4, None)
    sub = [P.import_words(P.gen_words(bench.SEED, v, dq, nbits, with_common=True, word_off=0, nwords=4 * 2048), True, 4 * 65536) for v in range(nvec)]
    assert one["count"] == int(P.pipeline_counts([(sub, [])])[0]) and one["cores"] == 1
