#!/usr/bin/env python3
"""Per-kernel resource figures of the gfx950 code object inside bitmagic_amd/lib/libbmx.so (no GPU needed): the
.hip_fatbin section is a clang offload bundle; its amdgcn entry is an ELF whose AMDGPU notes list, per kernel,
.private_segment_fixed_size (scratch bytes per lane: register spills), .vgpr_count, .sgpr_count, .group_segment_fixed_size.
`python tools/scratch_check.py [lib]` prints the kernels with scratch; kernels() is used by tests/test_abi_host.py."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """-> [(triple, bytes)] of every amdgcn entry of every bundle in the file"""
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        b = data.find(MAGIC, pos)
        if b < 0:
            break
        n = struct.unpack_from("<Q", data, b + len(MAGIC))[0]
        p = b + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p); p += 24
            triple = data[p:p + tlen].decode(); p += tlen
            if "amdgcn" in triple and size:
                out.append((triple, data[b + off:b + off + size]))
        pos = b + len(MAGIC)
    return out


def kernels(lib_path=None):
    """-> {kernel symbol: {"scratch": bytes per lane, "vgpr": n, "sgpr": n, "lds": bytes}}"""
    lib_path = lib_path or os.path.join(ROOT, "bitmagic_amd", "lib", "libbmx.so")
    res = {}
    for triple, blob in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # one record per kernel, keys in alphabetical order: "  - .agpr_count:" opens a record
        for rec in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            f = {}
            for k in ("private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                m = re.search(r"\n\s*\." + k + r":\s*(\d+)", rec)
                f[k] = int(m.group(1)) if m else 0
            m = re.search(r"\n\s*\.symbol:\s*'?([^\s']+?)\.kd'?\s*\n", rec)
            if not m:
                continue
            res[m.group(1)] = {"scratch": f["private_segment_fixed_size"], "vgpr": f["vgpr_count"], "sgpr": f["sgpr_count"],
                               "lds": f["group_segment_fixed_size"], "vgpr_spill": f["vgpr_spill_count"], "sgpr_spill": f["sgpr_spill_count"]}
    return res


if __name__ == "__main__":
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else None)
    bad = {k: v for k, v in ks.items() if v["scratch"]}
    print(f"{len(ks)} kernels, {len(bad)} with scratch")
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1]["scratch"]):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"  scratch {v['scratch']:4d} B/lane  vgpr {v['vgpr']:3d}  {name[:150]}")
