mkdir -p gpurun_out
cat > /tmp/soak.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle, bitmagic_amd as bm
import test_gpu_stress as S, test_gpu_parity as P
ctx = bm.context(0); port = oracle.port()
bad = 0
for seed in range(60, 1500):
    try: S.test_random_block_tables(ctx, port, seed)
    except AssertionError as e: bad += 1; print("FAIL seed", seed, str(e)[:200])
for args in [(2, 50, 45), (17, 80, 64), (90, 60, 33), (333, 150, 70), (600, 20, 40), (250, 300, 37)]:
    try: P.test_sparse_state_of_gap_lists(ctx, port, *args)
    except AssertionError as e: bad += 1; print("FAIL sparse", args, str(e)[:200])
for dq, nv in [(7, 130), (100, 64), (65400, 40), (2000, 35)]:
    try: P.test_many_gap_operands(ctx, port, dq, nv)
    except AssertionError as e: bad += 1; print("FAIL many", dq, nv, str(e)[:200])
print("soak done, failures:", bad)
PY
timeout 1200 python /tmp/soak.py > gpurun_out/soak.log 2>&1; tail -6 gpurun_out/soak.log
