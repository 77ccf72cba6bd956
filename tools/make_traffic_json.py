"""profiles/traffic_*.json from the PMC passes of an end-of-round run (tools/gpu_runs/r04_all.sh writes pmc_*.txt: one line per
(kernel, counter): "<kernel> <counter> per launch avg <x> launches <n>").  Every file is stamped with the kernel the counters
were collected on and the commit of the library that ran; bench.py drops a figure whose stamp does not match the kernel it runs.
Units as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE / WRITE_SIZE are in KB, FETCH_SIZE counts a
128-byte request as 64 B (x 2); TCC_MISS x 128 B is the cross-check.
Usage: python tools/make_traffic_json.py profiles/r04final [commit]"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run = sys.argv[1].rstrip("/")
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()

def counters(fname, kernel):
    out = {}
    path = os.path.join(ROOT, run, fname)
    if not os.path.exists(path): return out
    for line in open(path):
        m = re.match(r"(.*) (\w+) per launch avg ([0-9.eE+-]+) launches (\d+)", line.strip())
        if m and kernel in m.group(1): out[m.group(2)] = (float(m.group(3)), int(m.group(4)), m.group(1).strip())
    return out

TARGETS = [  # (json file, pmc file, kernel substring, workload label, kernel stamp)
    ("traffic_latest.json", "pmc_headline.txt", "k_pipe_counts_bits2", "agg_and_count_256x1000000000", "k_pipe_counts_bits2"),
    ("traffic_config1.json", "pmc_config1.txt", "k_count_op2_stream", "pairwise_count_2x1000000000_dq6554", "k_count_op2_stream"),
    ("traffic_config1_1pct.json", "pmc_config1_1pct.txt", "k_count_op2_loop", "pairwise_count_2x1000000000_dq655", "k_count_op2_loop"),
    ("traffic_config3.json", "pmc_config3.txt", "k_rank_lines", "rank_10M_on_4e9_bits_dq6554", "k_rank_lines"),
    ("traffic_config4.json", "pmc_config4.txt", "k_agg_or_rows", "combine_or_4096x4000000000_dq13_first_call", "k_agg_or_rows"),
    ("traffic_config4_warm.json", "pmc_config4.txt", "k_coll_apply", "combine_or_4096x4000000000_dq13_prepared_collection", "k_coll_apply<OR,512>"),
    ("traffic_config1_50pct.json", "pmc_config1_50pct.txt", "k_count_op2_stream", "pairwise_count_2x1000000000_dq32768", "k_count_op2_stream"),
    ("traffic_config3_1pct.json", "pmc_config3_1pct.txt", "k_rank_lines", "rank_10M_on_4e9_bits_dq655", "k_rank_lines"),
    ("traffic_config3_select.json", "pmc_config3.txt", "k_select_sel", "select_10M_on_4e9_bits_dq6554", "k_select_sel"),
    ("traffic_config3_1pct_select.json", "pmc_config3_1pct.txt", "k_select_sel", "select_10M_on_4e9_bits_dq655", "k_select_sel"),
    ("traffic_config2_dq197.json", "pmc_dq197.txt", "k_agg_and_rows", "agg_and_count_256x1000000000_dq197_first_call", "k_agg_and_rows"),
    ("traffic_config2_dq66.json", "pmc_dq66.txt", "k_agg_and_rows", "agg_and_count_256x1000000000_dq66_first_call", "k_agg_and_rows"),
]
for jname, pmc, ksub, workload, stamp in TARGETS:
    c = counters(pmc, ksub)
    if "FETCH_SIZE" not in c:
        print(jname, "-- no FETCH_SIZE for", ksub, "in", pmc, ": left as it is"); continue
    fetch_kb, nl, kfull = c["FETCH_SIZE"]
    write_kb = c.get("WRITE_SIZE", (0.0, 0, ""))[0]
    hbm = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    path = os.path.join(ROOT, "profiles", jname)
    old = {}
    try: old = json.load(open(path))
    except Exception: pass
    src = (f"{run}/{pmc}: rocprofv3 --kernel-trace --pmc FETCH_SIZE ({fetch_kb:,.1f} KB avg of {nl} launches of {kfull}) x 1024 x 2 "
           f"(gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE ({write_kb:,.1f} KB) x 1024, separate passes")
    if "TCC_MISS_sum" in c:
        src += f"; cross-check TCC_MISS_sum {c['TCC_MISS_sum'][0]:,.0f} x 128 B = {c['TCC_MISS_sum'][0] * 128 / 1e9:.3f} GB"
    j = {"workload": workload, "kernel": stamp, "commit": commit, "hbm_bytes_per_launch": hbm}
    if "algorithmic_bytes_per_launch" in old: j["algorithmic_bytes_per_launch"] = old["algorithmic_bytes_per_launch"]
    if "TCC_MISS_sum" in c: j["tcc_miss_per_launch"] = c["TCC_MISS_sum"][0]              # 128-byte lines that left the L2
    if "TCP_TCC_READ_REQ_sum" in c: j["tcp_tcc_read_req_per_launch"] = c["TCP_TCC_READ_REQ_sum"][0]
    j["source"] = src
    json.dump(j, open(path, "w"), indent=1); open(path, "a").write("\n")
    print(jname, stamp, commit, hbm, ("= %.3f x algorithmic" % (hbm / j["algorithmic_bytes_per_launch"])) if "algorithmic_bytes_per_launch" in j else "")
