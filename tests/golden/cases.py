"""Deterministic input construction shared by make_golden.py (runs against the
reference in the build container) and the tests (run anywhere).  Inputs are
rebuilt from the normative counter-based generator (oracle/bmx_oracle.c
bmo_gen_word64); the fixtures pin them with a SHA-256 so a generator drift is caught."""
from __future__ import annotations

import hashlib

import numpy as np

SEED = 0xB17A61C
BLOCK_WORDS = 2048

# name -> (nbits, n_vectors, density_q16, common_density_q16 or None, structured)
CASES = {
    "gap_sparse":   (3 * 65536 + 4321, 6, 66, None, True),      # 0.1 %  -> all GAP
    "gap_half":     (3 * 65536 + 4321, 6, 328, None, True),     # 0.5 %  -> all GAP, long run lists
    "mixed_1pct":   (4 * 65536 + 77, 6, 655, 655, True),        # 1 %    -> bit/GAP mix
    "bit_10pct":    (3 * 65536 + 4321, 6, 6554, 6554, True),    # 10 %   -> bit-blocks, correlated (config 3A)
    "bit_10pct_ind": (3 * 65536 + 1, 6, 6554, None, False),     # 10 %   independent (config 3B: AND empties)
    "bit_50pct":    (2 * 65536 + 40000, 6, 32768, None, True),  # 50 %
    "dense_gap":    (3 * 65536 + 4321, 6, 65470, None, True),   # 99.9 % -> GAP with long 1-runs
    "one_block":    (1000, 6, 6554, None, False),               # config 1 scale: a single partial block
}


def structure(words: np.ndarray, case: str, v: int) -> np.ndarray:
    """Plant NULL / FULL blocks and long runs so every block kind and every
    pairwise kind combination occurs (mirrors FillSetsRandomMethod's mixing,
    tests/stress/t.cpp:917)."""
    nblocks = (words.size + BLOCK_WORDS - 1) // BLOCK_WORDS
    h = int(hashlib.sha256(f"{case}:{v}".encode()).hexdigest(), 16)
    for nb in range(nblocks):
        r = (h >> (4 * nb)) & 15
        lo, hi = nb * BLOCK_WORDS, min((nb + 1) * BLOCK_WORDS, words.size)
        if r == 0:
            words[lo:hi] = 0
        elif r == 1:
            words[lo:hi] = 0xFFFFFFFF
        elif r == 2:                       # a long run of ones inside the block
            words[lo + (hi - lo) // 4: lo + (hi - lo) // 2] = 0xFFFFFFFF
        elif r == 3:                       # a long hole
            words[lo + (hi - lo) // 3: lo + 2 * (hi - lo) // 3] = 0
    return words


def make_inputs(port, case: str):
    """-> list of uint32 word arrays (one per vector), nbits"""
    nbits, n, dq, cdq, structured = CASES[case]
    out = []
    for v in range(n):
        w = port.gen_words(SEED, v, dq, nbits)
        if cdq:
            w |= port.gen_words(SEED, 0xFFFFFFFF, cdq, nbits)
        if structured:
            w = structure(w, case, v)
            # keep bits >= nbits zero
            tail = nbits % 32
            nw = (nbits + 31) // 32
            w[nw:] = 0
            if tail:
                w[nw - 1] &= (1 << tail) - 1
        out.append(w)
    return out, nbits


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# operand selections exercised per case: (AND indices, SUB indices)
AGG_GROUPS = [([0, 1], []), ([0, 1, 2], []), ([0, 1, 2, 3, 4, 5], []), ([0, 1], [2]), ([0], [1, 2, 3]),
              ([0, 1, 2], [3, 4, 5]), ([3], []), ([1, 0, 5, 2], [4])]
OR_SETS = [[0, 1], [0, 1, 2, 3, 4, 5], [2], [5, 3, 1]]
PAIRS = [(0, 1), (1, 0), (2, 3), (4, 5), (0, 0)]
# combine_shift_right_and operand sequences (src[0] is shifted n-1 times): short patterns (register path,
# shifts < 32), 40 and 70 operands (shifts across words), one operand (plain copy)
SHIFT_SETS = [[0, 1], [1, 0], [0, 1, 2], [5, 4, 3, 2, 1, 0], [2], [0, 1] * 20, [3, 4, 5, 0, 1, 2, 2] * 10]


def rank_queries(nbits: int) -> np.ndarray:
    rng = np.random.default_rng(12345)
    q = rng.integers(0, nbits, size=512).astype(np.uint64)
    fixed = [0, 1, 31, 32, 1023, 1024, 21823, 21824, 21825, 32736, 32737, 43647, 43648, 43649, 54560, 54561,
             65534, 65535, 65536, 65537, nbits - 1]
    fixed = [f for f in fixed if f < nbits]
    return np.concatenate([q, np.array(fixed, np.uint64)])


def select_queries(count: int) -> np.ndarray:
    rng = np.random.default_rng(54321)
    base = [0, 1, 2, count, count + 1, count + 1000]
    if count > 0:
        q = rng.integers(1, count + 1, size=512).astype(np.uint64)
        return np.concatenate([q, np.array(base, np.uint64)])
    return np.array(base, np.uint64)


# ---- vectors longer than 2^32 bits (the reference's BM64ADDR build) ----------
BM64_NBITS = 6_500_000_000            # 99,183 blocks; block 65,536 starts at bit 2^32
BM64_NVEC = 4


# pipeline::set_search_count_limit values of the golden cases (1 = stop after the first block with a hit)
SEARCH_LIMITS = [1, 40, 3000, 0xFFFFFFFF]                   # (the last one = bm::id_max, the default: no limit)


def bm64_build(o, seed: int):
    """one test vector, built bit by bit through the oracle / reference API (same calls on both)"""
    rng = np.random.default_rng(1000 + seed)
    v = o.new(BM64_NBITS)
    pos = np.concatenate([rng.integers(0, BM64_NBITS, size=3000), rng.integers(1 << 32, (1 << 32) + 200000, size=20000),
                          np.array([0, (1 << 32) - 1, 1 << 32, BM64_NBITS - 1])])
    for p in pos:
        v.set_bit(int(p))
    v.set_range((1 << 32) - 70000, (1 << 32) + 70000)                  # FULL block right below bit 2^32, runs across it
    v.set_range(6_000_000_000 + seed * 1000, 6_000_200_000)
    v.optimize()
    return v


def bm64_queries(count: int):
    rank_q = np.array([0, (1 << 32) - 1, 1 << 32, (1 << 32) + 5, (1 << 32) + 65536, 5_000_000_000, 6_000_100_000,
                       BM64_NBITS - 1], np.uint64)
    sel_q = np.array([1, 2, 100, max(count // 2, 1), max(count - 1, 1), count, count + 1], np.uint64)
    return rank_q, sel_q


# ---- round 3: set_range_hint (find_first_and_sub + pipelines with search masks) and the sparse_vector_scanner ----
def range_hints(nbits: int):
    """(from, to) search ranges: inside one block (bit-masked by the reference, bmaggregator.h:980-988), across blocks
    (block-granular), the first / last block, the whole vector"""
    last = nbits - 1
    h = [(0, 0), (5, 700), (1000, 65535), (65536, 65536 + 4000), (40000, 70000), (60000, 2 * 65536 + 100), (65536 * 2, last),
         (last - min(last, 300), last), (0, last), (100000, 100050)]
    return [(a, min(b, last)) for a, b in h if a <= last]


HINT_GROUPS = [([0, 1], []), ([0, 1], [2]), ([3], []), ([0], [1, 2, 3]), ([1, 0, 5, 2], [4])]

# scanner fixtures: (name, rows, value generator) -- the planes are rebuilt from these values by numpy in the tests,
# the expected result vectors come from bm::sparse_vector_scanner<> (make_golden.py)
SCANNER_ROWS = 3 * 65536 + 1234
SCANNER_VALUES = [0, 1, 2, 50, 95, 96, 97, 128, 69999, 70000, 70003, 70006, 70007, 131071, 131072, 4000000]
SCANNER_RANGES = [(0, 0), (0, 5), (3, 3), (90, 10), (10, 69999), (96, 70003), (70001, 70005), (1, 4000000)]
SCANNER_EQ_BATCH = [1, 2, 17, 64, 96, 97, 120, 70000, 70003, 70006, 131072, 5000000, 0, 33, 34, 35]


def scanner_values(with_null: bool):
    """-> (values uint32[rows], is_null uint8[rows] or None): small values (7 planes), rare wide ones (sparse high planes),
    a stretch of zeros, a constant stretch"""
    rng = np.random.default_rng(20260925)
    v = rng.integers(0, 97, SCANNER_ROWS).astype(np.uint32)
    idx = np.arange(SCANNER_ROWS)
    v[idx % 1000 == 0] = (70000 + (idx[idx % 1000 == 0] % 7)).astype(np.uint32)
    v[70000:72000] = 0
    v[140000:150000] = 33
    if not with_null:
        return v, None
    isn = (idx % 11 == 3).astype(np.uint8)
    isn[-1] = 0                                    # the last row is assigned: size() == rows either way
    v = v.copy(); v[isn != 0] = 0                  # NULL rows are stored as 0
    return v, isn


# signed containers (bm::sparse_vector<int, ..>), IN-list find_eq and invert
SCANNER_S_VALUES = [-2147483648, -70001, -70000, -97, -96, -50, -2, -1, 0, 1, 2, 50, 96, 97, 70000, 70003, 2147483647]
SCANNER_S_RANGES = [(-5, 5), (-50, -10), (10, 50), (0, 0), (-1, -1), (-1, 0), (-70003, 70003), (-2147483648, -1), (0, 2147483647), (40, -40)]
SCANNER_IN_LISTS = [[1, 2, 3], [0, 33], [96, 70003, 5000000], [17], [0], [50, 50, 51]]


def scanner_values_signed(with_null: bool):
    """-> (values int32[rows], is_null or None): the unsigned fixture's magnitudes with alternating signs, the extremes
    of the type, a stretch of -1 (sign plane only) and of zeros"""
    u, isn = scanner_values(with_null)
    v = u.astype(np.int64)
    idx = np.arange(SCANNER_ROWS)
    v[idx % 3 == 1] = -v[idx % 3 == 1]
    v[5] = -2147483648; v[6] = 2147483647
    v[100000:101000] = -1
    if isn is not None:
        v[isn != 0] = 0
    return v.astype(np.int32), isn


def s2u(v: np.ndarray) -> np.ndarray:
    """base_sparse_vector::s2u (src/bmbmatrix.h:2536-2548): sign in bit 0, magnitude above it"""
    v = v.astype(np.int64)
    return np.where(v >= 0, v << 1, ((-(v + 1)) << 1) | 1).astype(np.uint64)
