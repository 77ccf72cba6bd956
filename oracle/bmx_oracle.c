/*
 * bmx_oracle.c -- CPU restatement of the BitMagic bvector/aggregator hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see bmx_oracle.h).  Parity status: PINNED against
 * the reference (oracle/ref_shim.cpp) and tests/golden/.
 *
 * Citations are file:line relative to /root/reference/.
 */
#define _POSIX_C_SOURCE 200112L
#include "bmx_oracle.h"

#include <stdlib.h>
#include <string.h>

#define POP32(x) ((uint32_t)__builtin_popcount((uint32_t)(x)))
#define POP64(x) ((uint32_t)__builtin_popcountll((uint64_t)(x)))

struct bmo_vec {
    uint64_t nbits;     /* logical size_ */
    uint32_t nblocks;   /* length of the block table */
    uint8_t* kind;      /* BMO_NULL / FULL / BIT / GAP */
    void**   blk;       /* uint32_t[2048] (BIT) or uint16_t[len+1] (GAP) */
};

/* ======================================================================
 * Synthetic generator (normative; the HIP kernel bmx_gen implements the
 * same arithmetic).  Counter-based: word = f(seed, vec_id, word index).
 * Exact per-bit Bernoulli(d / 65536): fold 16 hashed words, LSB of d first,
 * OR where the density bit is 1, AND where it is 0.
 * ====================================================================== */
static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

uint64_t bmo_gen_word64(uint64_t seed, uint32_t vec_id, uint64_t w64, uint32_t d)
{
    if (d >= 65536u) return ~0ull;
    uint64_t base = seed ^ ((uint64_t)vec_id * 0x9E3779B97F4A7C15ull);
    uint64_t acc = 0;
    for (unsigned k = 0; k < 16; ++k) {
        uint64_t r = mix64(base + (w64 * 16u + k) * 0xD6E8FEB86659FD93ull);
        acc = ((d >> k) & 1u) ? (acc | r) : (acc & r);
    }
    return acc;
}

#define BMO_COMMON_ID 0xFFFFFFFFu

void bmo_gen_words(uint64_t seed, uint32_t vec_id, int with_common, uint32_t d,
                   uint64_t nbits, uint64_t word_off, uint64_t nwords32, uint32_t* out)
{
    for (uint64_t i = 0; i < nwords32; i += 2) {
        uint64_t w64 = (word_off + i) >> 1;
        uint64_t v = bmo_gen_word64(seed, vec_id, w64, d);
        if (with_common) v |= bmo_gen_word64(seed, BMO_COMMON_ID, w64, d);
        uint64_t bit0 = w64 * 64u;
        if (bit0 >= nbits) v = 0;
        else if (nbits - bit0 < 64u) v &= (~0ull) >> (64u - (nbits - bit0));
        out[i] = (uint32_t)v;
        if (i + 1 < nwords32) out[i + 1] = (uint32_t)(v >> 32);
    }
}

/* ======================================================================
 * Bit-block primitives
 * ====================================================================== */

/* src/bmfunc.h:5808 bit_block_count */
uint32_t bmo_bit_block_count(const uint32_t* blk)
{
    const uint64_t* b = (const uint64_t*)blk;
    uint32_t c = 0;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS / 2; ++i) c += POP64(b[i]);
    return c;
}

/* src/bmfunc.h:5827 bit_block_count(block, digest): only waves named by the digest */
uint32_t bmo_bit_block_count_digest(const uint32_t* blk, uint64_t digest)
{
    uint32_t c = 0;
    while (digest) {
        unsigned wave = (unsigned)__builtin_ctzll(digest);
        digest &= digest - 1;
        const uint64_t* b = (const uint64_t*)(blk + wave * BMO_WAVE_WORDS);
        for (unsigned i = 0; i < BMO_WAVE_WORDS / 2; ++i) c += POP64(b[i]);
    }
    return c;
}

static inline int wave_nonzero(const uint32_t* w)
{
    const uint64_t* b = (const uint64_t*)w;
    uint64_t acc = 0;
    for (unsigned i = 0; i < BMO_WAVE_WORDS / 2; ++i) acc |= b[i];
    return acc != 0;
}

/* src/bmfunc.h:1239 calc_block_digest0 */
uint64_t bmo_calc_block_digest0(const uint32_t* blk)
{
    uint64_t d = 0;
    for (unsigned w = 0; w < 64; ++w)
        d |= (uint64_t)wave_nonzero(blk + w * BMO_WAVE_WORDS) << w;
    return d;
}

/* src/bmfunc.h:1281 update_block_digest0 */
static uint64_t update_block_digest0(const uint32_t* blk, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d);
        d &= d - 1;
        if (!wave_nonzero(blk + wave * BMO_WAVE_WORDS)) digest &= ~(1ull << wave);
    }
    return digest;
}

/* src/bmfunc.h:1211 block_init_digest0 */
static void block_init_digest0(uint32_t* blk, uint64_t digest)
{
    for (unsigned w = 0; w < 64; ++w)
        memset(blk + w * BMO_WAVE_WORDS, ((digest >> w) & 1) ? 0xFF : 0, BMO_WAVE_WORDS * 4);
}

/* src/bmfunc.h:1144 digest_mask */
static inline uint64_t digest_mask(unsigned from, unsigned to)
{
    unsigned df = from >> 10, dt = to >> 10;
    return ((~0ull) >> (63 - (dt - df))) << df;
}

/* src/bmfunc.h:6040 bit_block_calc_change (number of runs = 1 + transitions) */
uint32_t bmo_bit_block_calc_change(const uint32_t* blk)
{
    uint32_t runs = 1;
    uint32_t prev = blk[0] & 1u;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) {
        uint32_t w = blk[i];
        /* transitions inside the word: bit k differs from bit k-1 (k=1..31) */
        runs += POP32((w ^ (w >> 1)) & 0x7FFFFFFFu);
        /* transition across the word border */
        runs += ((w & 1u) != prev);
        prev = w >> 31;
    }
    return runs;
}

/* src/bmfunc.h:6147 bit_block_calc_count_range: popcount of bits [left..right] */
uint32_t bmo_bit_block_count_range(const uint32_t* blk, uint32_t left, uint32_t right)
{
    uint32_t wl = left >> 5, wr = right >> 5;
    uint32_t ml = ~0u << (left & 31u);
    uint32_t mr = ~0u >> (31u - (right & 31u));
    if (wl == wr) return POP32(blk[wl] & ml & mr);
    uint32_t c = POP32(blk[wl] & ml);
    for (uint32_t i = wl + 1; i < wr; ++i) c += POP32(blk[i]);
    return c + POP32(blk[wr] & mr);
}

/* ---- digest-driven AND / SUB kernels (src/bmfunc.h:7652,7833,7768,7706 and
 *      8898,8958,9073,9016).  Only waves whose digest bit is set are touched;
 *      a wave that becomes all-zero clears its digest bit. ---- */
uint64_t bmo_and_2way(uint32_t* dst, const uint32_t* s1, const uint32_t* s2, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i)
            acc |= (dst[off + i] = s1[off + i] & s2[off + i]);
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

static uint64_t and_1(uint32_t* dst, const uint32_t* s, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i) acc |= (dst[off + i] &= s[off + i]);
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

static uint64_t and_3way(uint32_t* dst, const uint32_t* s1, const uint32_t* s2, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i)
            acc |= (dst[off + i] &= s1[off + i] & s2[off + i]);
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

uint64_t bmo_and_5way(uint32_t* dst, const uint32_t* s0, const uint32_t* s1,
                      const uint32_t* s2, const uint32_t* s3, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i)
            acc |= (dst[off + i] &= s0[off + i] & s1[off + i] & s2[off + i] & s3[off + i]);
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

static uint64_t sub_1(uint32_t* dst, const uint32_t* s, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i) acc |= (dst[off + i] &= ~s[off + i]);
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

static uint64_t sub_3way(uint32_t* dst, const uint32_t* s1, const uint32_t* s2, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i)
            acc |= (dst[off + i] &= ~(s1[off + i] | s2[off + i]));
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

static uint64_t sub_5way(uint32_t* dst, const uint32_t* s0, const uint32_t* s1,
                         const uint32_t* s2, const uint32_t* s3, uint64_t digest)
{
    uint64_t d = digest;
    while (d) {
        unsigned wave = (unsigned)__builtin_ctzll(d); d &= d - 1;
        unsigned off = wave * BMO_WAVE_WORDS; uint32_t acc = 0;
        for (unsigned i = 0; i < BMO_WAVE_WORDS; ++i)
            acc |= (dst[off + i] &= ~(s0[off + i] | s1[off + i] | s2[off + i] | s3[off + i]));
        if (!acc) digest &= ~(1ull << wave);
    }
    return digest;
}

/* whole-block OR family (src/bmfunc.h:8592,8709,8753): returns 1 iff result is all-ones */
static int or_1(uint32_t* dst, const uint32_t* s)
{
    uint32_t acc = ~0u;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) acc &= (dst[i] |= s[i]);
    return acc == ~0u;
}
static int or_3way(uint32_t* dst, const uint32_t* s1, const uint32_t* s2)
{
    uint32_t acc = ~0u;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) acc &= (dst[i] |= s1[i] | s2[i]);
    return acc == ~0u;
}
static int or_5way(uint32_t* dst, const uint32_t* s0, const uint32_t* s1,
                   const uint32_t* s2, const uint32_t* s3)
{
    uint32_t acc = ~0u;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i)
        acc &= (dst[i] |= s0[i] | s1[i] | s2[i] | s3[i]);
    return acc == ~0u;
}

static int block_is_all_zero(const uint32_t* b)   /* src/bmfunc.h:1669 */
{
    const uint64_t* p = (const uint64_t*)b; uint64_t acc = 0;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS / 2; ++i) acc |= p[i];
    return acc == 0;
}
static int block_is_all_one(const uint32_t* b)    /* src/bmfunc.h:6838 */
{
    const uint64_t* p = (const uint64_t*)b; uint64_t acc = ~0ull;
    for (unsigned i = 0; i < BMO_BLOCK_WORDS / 2; ++i) acc &= p[i];
    return acc == ~0ull;
}

/* ======================================================================
 * GAP primitives.  GAP block (src/bmfunc.h:1722):
 *   buf[0] = (len << 3) | (level << 1) | start_bit ; buf[1..len] = inclusive
 *   run ends, ascending, buf[len] == 65535.
 * ====================================================================== */
static inline unsigned gap_len(const uint16_t* g) { return (unsigned)(g[0] >> 3); }

/* set bits [pos, pos+cnt) -- src/bmfunc.h:4520 or_bit_block */
static void or_bit_range(uint32_t* dst, uint32_t pos, uint32_t cnt)
{
    uint32_t last = pos + cnt - 1;
    uint32_t wl = pos >> 5, wr = last >> 5;
    uint32_t ml = ~0u << (pos & 31u), mr = ~0u >> (31u - (last & 31u));
    if (wl == wr) { dst[wl] |= ml & mr; return; }
    dst[wl] |= ml;
    for (uint32_t i = wl + 1; i < wr; ++i) dst[i] = ~0u;
    dst[wr] |= mr;
}
/* clear bits -- src/bmfunc.h:4568 sub_bit_block */
static void sub_bit_range(uint32_t* dst, uint32_t pos, uint32_t cnt)
{
    uint32_t last = pos + cnt - 1;
    uint32_t wl = pos >> 5, wr = last >> 5;
    uint32_t ml = ~0u << (pos & 31u), mr = ~0u >> (31u - (last & 31u));
    if (wl == wr) { dst[wl] &= ~(ml & mr); return; }
    dst[wl] &= ~ml;
    for (uint32_t i = wl + 1; i < wr; ++i) dst[i] = 0u;
    dst[wr] &= ~mr;
}
/* flip bits -- src/bmfunc.h:4611 xor_bit_block */
static void xor_bit_range(uint32_t* dst, uint32_t pos, uint32_t cnt)
{
    uint32_t last = pos + cnt - 1;
    uint32_t wl = pos >> 5, wr = last >> 5;
    uint32_t ml = ~0u << (pos & 31u), mr = ~0u >> (31u - (last & 31u));
    if (wl == wr) { dst[wl] ^= ml & mr; return; }
    dst[wl] ^= ml;
    for (uint32_t i = wl + 1; i < wr; ++i) dst[i] ^= ~0u;
    dst[wr] ^= mr;
}

/* iterate runs: calls fn(dst, start, count) for every run whose value == want */
typedef void (*range_fn)(uint32_t*, uint32_t, uint32_t);
static void gap_for_runs(uint32_t* dst, const uint16_t* g, unsigned want, range_fn fn)
{
    unsigned len = gap_len(g);
    unsigned val = g[0] & 1u;
    uint32_t start = 0;
    for (unsigned k = 1; k <= len; ++k) {
        uint32_t end = g[k];
        if (val == want) fn(dst, start, end - start + 1);
        start = end + 1; val ^= 1u;
    }
}

/* src/bmfunc.h:4796 gap_add_to_bitset */
static void gap_add_to_bitset(uint32_t* dst, const uint16_t* g) { gap_for_runs(dst, g, 1, or_bit_range); }
/* src/bmfunc.h:4847 gap_and_to_bitset: clear the 0-runs */
static void gap_and_to_bitset(uint32_t* dst, const uint16_t* g) { gap_for_runs(dst, g, 0, sub_bit_range); }
/* src/bmfunc.h:4669 gap_sub_to_bitset: clear the 1-runs */
static void gap_sub_to_bitset(uint32_t* dst, const uint16_t* g) { gap_for_runs(dst, g, 1, sub_bit_range); }
/* src/bmfunc.h:4768 gap_xor_to_bitset */
static void gap_xor_to_bitset(uint32_t* dst, const uint16_t* g) { gap_for_runs(dst, g, 1, xor_bit_range); }

/* src/bmfunc.h:5232 gap_convert_to_bitset */
void bmo_gap_convert_to_bitset(uint32_t* dest, const uint16_t* gap)
{
    memset(dest, 0, BMO_BLOCK_WORDS * 4);
    gap_add_to_bitset(dest, gap);
}

/* digest-assisted forms (src/bmfunc.h:4884 / 4700): runs that only cover waves
 * already zero in the digest are skipped, then the digest is re-validated. */
static uint64_t gap_clear_runs_digest(uint32_t* dst, const uint16_t* g, unsigned want, uint64_t digest)
{
    if (!digest) return digest;
    unsigned len = gap_len(g);
    unsigned val = g[0] & 1u;
    uint32_t start = 0;
    uint32_t stop_pos = (64u - (unsigned)__builtin_clzll(digest)) << 10; /* first bit past the digest tail */
    for (unsigned k = 1; k <= len; ++k) {
        uint32_t end = g[k];
        if (val == want && (digest & digest_mask(start, end)))
            sub_bit_range(dst, start, end - start + 1);
        start = end + 1; val ^= 1u;
        if (start >= stop_pos) break;
    }
    return update_block_digest0(dst, digest);
}
uint64_t bmo_gap_and_to_bitset_digest(uint32_t* dst, const uint16_t* gap, uint64_t digest)
{ return gap_clear_runs_digest(dst, gap, 0, digest); }
uint64_t bmo_gap_sub_to_bitset_digest(uint32_t* dst, const uint16_t* gap, uint64_t digest)
{ return gap_clear_runs_digest(dst, gap, 1, digest); }

/* src/bmfunc.h:3079 gap_bit_count */
uint32_t bmo_gap_bit_count(const uint16_t* g)
{
    unsigned len = gap_len(g);
    unsigned val = g[0] & 1u;
    uint32_t start = 0, c = 0;
    for (unsigned k = 1; k <= len; ++k) {
        uint32_t end = g[k];
        if (val) c += end - start + 1;
        start = end + 1; val ^= 1u;
    }
    return c;
}

/* src/bmfunc.h:1844 gap_bfind: smallest k with buf[k] >= pos; *is_set = value of that run */
unsigned bmo_gap_bfind(const uint16_t* g, uint32_t pos, unsigned* is_set)
{
    unsigned lo = 1, hi = gap_len(g);
    while (lo < hi) {
        unsigned mid = (lo + hi) >> 1;
        if (g[mid] < pos) lo = mid + 1; else hi = mid;
    }
    *is_set = (g[0] & 1u) ^ ((lo - 1) & 1u);
    return lo;
}

/* src/bmfunc.h:1909 gap_test */
unsigned bmo_gap_test(const uint16_t* g, uint32_t pos)
{
    unsigned is_set; (void)bmo_gap_bfind(g, pos, &is_set); return is_set;
}

/* src/bmfunc.h:3499 gap_bit_count_to: ones in [0..right] */
uint32_t bmo_gap_bit_count_to(const uint16_t* g, uint32_t right)
{
    unsigned len = gap_len(g);
    unsigned val = g[0] & 1u;
    uint32_t start = 0, c = 0;
    for (unsigned k = 1; k <= len; ++k) {
        uint32_t end = g[k];
        if (end >= right) { if (val) c += right - start + 1; return c; }
        if (val) c += end - start + 1;
        start = end + 1; val ^= 1u;
    }
    return c;
}

/* ones in [left..right] -- src/bmfunc.h:3173 gap_bit_count_range */
static uint32_t gap_bit_count_range(const uint16_t* g, uint32_t left, uint32_t right)
{
    uint32_t c = bmo_gap_bit_count_to(g, right);
    if (left) c -= bmo_gap_bit_count_to(g, left - 1);
    return c;
}

/* src/bmfunc.h:5542 bit_block_to_gap (unbounded destination: the caller has
 * checked calc_change < BMO_GAP_THRESHOLD or passes a 65537-entry buffer).
 * A run ends just before every position whose bit differs from its predecessor. */
unsigned bmo_bit_to_gap(uint16_t* dest, const uint32_t* blk)
{
    unsigned len = 0;
    uint32_t prev = blk[0] & 1u;            /* bit "-1" := bit 0, so no transition at 0 */
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) {
        uint32_t w = blk[i];
        uint32_t t = w ^ ((w << 1) | prev); /* bit k set iff bit k != bit k-1 */
        prev = w >> 31;
        while (t) {
            unsigned k = (unsigned)__builtin_ctz(t); t &= t - 1;
            dest[++len] = (uint16_t)(i * 32u + k - 1u);
        }
    }
    dest[++len] = 65535u;
    dest[0] = (uint16_t)((len << 3) | (blk[0] & 1u));
    return len;
}

/* two-pointer merge of two run lists (src/bmfunc.h:3747 gap_buff_op and its
 * wrappers :7275,7423,7342,7469).  Returns len, or 0 if dest_cap would be exceeded. */
static inline unsigned apply_op(int op, unsigned a, unsigned b)
{
    switch (op) {
    case BMO_AND: return a & b;
    case BMO_OR:  return a | b;
    case BMO_XOR: return a ^ b;
    default:      return a & (b ^ 1u);
    }
}

unsigned bmo_gap_op(int op, const uint16_t* a, const uint16_t* b, uint16_t* dest, unsigned dest_cap)
{
    unsigned ia = 1, ib = 1;
    unsigned va = a[0] & 1u, vb = b[0] & 1u;
    unsigned cur = apply_op(op, va, vb);
    unsigned start_val = cur, len = 0;
    for (;;) {
        uint32_t ea = a[ia], eb = b[ib];
        uint32_t e = ea < eb ? ea : eb;      /* both operands constant on [.., e] */
        if (ea == e) { ++ia; va ^= 1u; }
        if (eb == e) { ++ib; vb ^= 1u; }
        if (e == 65535u) {
            if (len + 1 >= dest_cap) return 0;
            dest[++len] = 65535u;
            break;
        }
        unsigned nxt = apply_op(op, va, vb);
        if (nxt != cur) {
            if (len + 1 >= dest_cap) return 0;
            dest[++len] = (uint16_t)e;
            cur = nxt;
        }
    }
    dest[0] = (uint16_t)((len << 3) | start_val);
    return len;
}

/* popcount of (a OP b) for two GAP blocks (src/bmfunc.h:7317,7443,7398,7514) */
static uint32_t gap_count_op(int op, const uint16_t* a, const uint16_t* b)
{
    unsigned ia = 1, ib = 1;
    unsigned va = a[0] & 1u, vb = b[0] & 1u;
    uint32_t start = 0, c = 0;
    for (;;) {
        uint32_t ea = a[ia], eb = b[ib];
        uint32_t e = ea < eb ? ea : eb;
        if (apply_op(op, va, vb)) c += e - start + 1;
        if (ea == e) { ++ia; va ^= 1u; }
        if (eb == e) { ++ib; vb ^= 1u; }
        if (e == 65535u) break;
        start = e + 1;
    }
    return c;
}

/* popcount of (bit-block OP gap) without materialising
 * (src/bmfunc.h:4955,5013,5086,5162).  'bit_is_a' selects the SUB direction. */
static uint32_t gap_bitset_count_op(int op, const uint32_t* blk, const uint16_t* g, int bit_is_a)
{
    unsigned len = gap_len(g);
    unsigned val = g[0] & 1u;
    uint32_t start = 0, c = 0;
    for (unsigned k = 1; k <= len; ++k) {
        uint32_t end = g[k];
        uint32_t n = end - start + 1;
        uint32_t ones = bmo_bit_block_count_range(blk, start, end);
        switch (op) {
        case BMO_AND: if (val) c += ones; break;
        case BMO_OR:  c += val ? n : ones; break;
        case BMO_XOR: c += val ? (n - ones) : ones; break;
        default:      /* SUB */
            if (bit_is_a) { if (!val) c += ones; }       /* bit & ~gap */
            else          { if (val) c += n - ones; }    /* gap & ~bit */
            break;
        }
        start = end + 1; val ^= 1u;
    }
    return c;
}

/* ======================================================================
 * Vector: flat block table
 * ====================================================================== */
static uint32_t* alloc_bit_block(void)
{
    void* p = NULL;
    if (posix_memalign(&p, 64, BMO_BLOCK_WORDS * 4)) return NULL;
    return (uint32_t*)p;
}

bmo_vec* bmo_vec_new(uint64_t nbits)
{
    bmo_vec* v = (bmo_vec*)calloc(1, sizeof(*v));
    v->nbits = nbits;
    v->nblocks = (uint32_t)((nbits + BMO_BLOCK_BITS - 1) / BMO_BLOCK_BITS);
    v->kind = (uint8_t*)calloc(v->nblocks ? v->nblocks : 1, 1);
    v->blk = (void**)calloc(v->nblocks ? v->nblocks : 1, sizeof(void*));
    return v;
}

static void block_release(bmo_vec* v, uint32_t nb)
{
    free(v->blk[nb]); v->blk[nb] = NULL; v->kind[nb] = BMO_NULL;
}

void bmo_vec_free(bmo_vec* v)
{
    if (!v) return;
    for (uint32_t i = 0; i < v->nblocks; ++i) free(v->blk[i]);
    free(v->blk); free(v->kind); free(v);
}

uint64_t bmo_vec_nbits(const bmo_vec* v) { return v->nbits; }
uint32_t bmo_vec_nblocks(const bmo_vec* v) { return v->nblocks; }

/* GAP capacity level, src/bmfunc.h:5418 gap_calc_level with the default
 * length table {128,256,512,1280} (src/bmconst.h:396-403) */
static int gap_calc_level(unsigned len)
{
    if (len <= 124) return 0;
    if (len <= 252) return 1;
    if (len <= 508) return 2;
    if (len <= 1276) return 3;
    return -1;
}

static uint16_t* gap_clone(const uint16_t* g, unsigned len)
{
    uint16_t* p = (uint16_t*)malloc((len + 1) * 2);
    memcpy(p, g, (len + 1) * 2);
    int lvl = gap_calc_level(len);
    p[0] = (uint16_t)((p[0] & ~6u) | ((unsigned)(lvl < 0 ? 3 : lvl) << 1));
    return p;
}

/* Store a computed bit-block into table slot nb following
 * blocks_manager::opt_copy_bit_block (src/bmblocks.h:1355-1409):
 *   runs == 1 -> NULL / FULL ; runs < 1276 -> GAP ; else bit-block copy.
 * opt == 0 (opt_none): plain copy (copy_bit_block, :1340). */
static void store_bit_block(bmo_vec* v, uint32_t nb, const uint32_t* src, int opt)
{
    if (opt) {
        uint32_t runs = bmo_bit_block_calc_change(src);
        if (runs == 1) { v->kind[nb] = src[0] ? BMO_FULL : BMO_NULL; v->blk[nb] = NULL; return; }
        if (runs < BMO_GAP_THRESHOLD) {
            uint16_t tmp[BMO_GAP_MAX_LEN + 8];
            unsigned len = bmo_bit_to_gap(tmp, src);
            v->blk[nb] = gap_clone(tmp, len); v->kind[nb] = BMO_GAP; return;
        }
    }
    uint32_t* p = alloc_bit_block();
    memcpy(p, src, BMO_BLOCK_WORDS * 4);
    v->blk[nb] = p; v->kind[nb] = BMO_BIT;
}

/* src/bmbvimport.h:46 bit_import_u32 (+ optimize_bit_block, src/bmblocks.h:1412) */
bmo_vec* bmo_vec_import(const uint32_t* words, uint64_t nwords, int optimize)
{
    bmo_vec* v = bmo_vec_new(nwords * 32u);
    uint32_t tmp[BMO_BLOCK_WORDS];
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
        uint64_t off = (uint64_t)nb * BMO_BLOCK_WORDS;
        uint64_t n = nwords - off < BMO_BLOCK_WORDS ? nwords - off : BMO_BLOCK_WORDS;
        memcpy(tmp, words + off, n * 4);
        if (n < BMO_BLOCK_WORDS) memset(tmp + n, 0, (BMO_BLOCK_WORDS - n) * 4);
        store_bit_block(v, nb, tmp, optimize);
    }
    return v;
}

bmo_vec* bmo_vec_from_table(uint64_t nbits, uint32_t nblocks, const uint8_t* kinds,
                            const uint32_t* offs, const uint32_t* bit_slab,
                            const uint16_t* gap_slab)
{
    bmo_vec* v = bmo_vec_new((uint64_t)nblocks * BMO_BLOCK_BITS);
    v->nbits = nbits;
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        v->kind[nb] = kinds[nb];
        if (kinds[nb] == BMO_BIT) {
            uint32_t* p = alloc_bit_block();
            memcpy(p, bit_slab + (uint64_t)offs[nb] * BMO_BLOCK_WORDS, BMO_BLOCK_WORDS * 4);
            v->blk[nb] = p;
        } else if (kinds[nb] == BMO_GAP) {
            const uint16_t* g = gap_slab + offs[nb];
            v->blk[nb] = gap_clone(g, gap_len(g));
        }
    }
    return v;
}

void bmo_vec_stat(const bmo_vec* v, uint32_t counts[4], uint64_t* gap_words)
{
    counts[0] = counts[1] = counts[2] = counts[3] = 0; *gap_words = 0;
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
        counts[v->kind[nb]]++;
        if (v->kind[nb] == BMO_GAP) *gap_words += gap_len((const uint16_t*)v->blk[nb]) + 1;
    }
}

void bmo_vec_flatten(const bmo_vec* v, uint8_t* kinds, uint32_t* offs,
                     uint32_t* bit_slab, uint16_t* gap_slab)
{
    uint32_t nbit = 0; uint64_t ngap = 0;
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
        kinds[nb] = v->kind[nb]; offs[nb] = 0;
        if (v->kind[nb] == BMO_BIT) {
            memcpy(bit_slab + (uint64_t)nbit * BMO_BLOCK_WORDS, v->blk[nb], BMO_BLOCK_WORDS * 4);
            offs[nb] = nbit++;
        } else if (v->kind[nb] == BMO_GAP) {
            const uint16_t* g = (const uint16_t*)v->blk[nb];
            unsigned n = gap_len(g) + 1;
            memcpy(gap_slab + ngap, g, n * 2);
            offs[nb] = (uint32_t)ngap; ngap += n;
        }
    }
}

/* expand block nb into a caller buffer; returns pointer to block data (either
 * the stored bit-block or tmp) */
static const uint32_t* block_as_bits(const bmo_vec* v, uint32_t nb, uint32_t* tmp)
{
    if (nb >= v->nblocks || v->kind[nb] == BMO_NULL) { memset(tmp, 0, BMO_BLOCK_WORDS * 4); return tmp; }
    switch (v->kind[nb]) {
    case BMO_FULL: memset(tmp, 0xFF, BMO_BLOCK_WORDS * 4); return tmp;
    case BMO_BIT:  return (const uint32_t*)v->blk[nb];
    default:       bmo_gap_convert_to_bitset(tmp, (const uint16_t*)v->blk[nb]); return tmp;
    }
}

void bmo_vec_to_words(const bmo_vec* v, uint32_t* out, uint64_t nwords)
{
    uint32_t tmp[BMO_BLOCK_WORDS];
    uint64_t nb_total = (nwords + BMO_BLOCK_WORDS - 1) / BMO_BLOCK_WORDS;
    for (uint64_t nb = 0; nb < nb_total; ++nb) {
        const uint32_t* b = block_as_bits(v, (uint32_t)nb, tmp);
        uint64_t off = nb * BMO_BLOCK_WORDS;
        uint64_t n = nwords - off < BMO_BLOCK_WORDS ? nwords - off : BMO_BLOCK_WORDS;
        memcpy(out + off, b, n * 4);
    }
}

/* bvector::optimize(opt_compress) block rule: src/bmblocks.h:1412-1436 for
 * bit-blocks; GAP blocks that are all-0 / all-1 collapse (src/bmfunc.h:1696,1709) */
void bmo_vec_optimize(bmo_vec* v)
{
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
        if (v->kind[nb] == BMO_BIT) {
            uint32_t* b = (uint32_t*)v->blk[nb];
            v->blk[nb] = NULL; v->kind[nb] = BMO_NULL;
            store_bit_block(v, nb, b, 1);
            free(b);
        } else if (v->kind[nb] == BMO_GAP) {
            const uint16_t* g = (const uint16_t*)v->blk[nb];
            if (gap_len(g) == 1) { int one = g[0] & 1; block_release(v, nb); v->kind[nb] = one ? BMO_FULL : BMO_NULL; }
        }
    }
}

int bmo_vec_equal(const bmo_vec* a, const bmo_vec* b)
{
    uint32_t ta[BMO_BLOCK_WORDS], tb[BMO_BLOCK_WORDS];
    uint32_t n = a->nblocks > b->nblocks ? a->nblocks : b->nblocks;
    for (uint32_t nb = 0; nb < n; ++nb) {
        const uint32_t* pa = block_as_bits(a, nb, ta);
        const uint32_t* pb = block_as_bits(b, nb, tb);
        if (memcmp(pa, pb, BMO_BLOCK_WORDS * 4)) return 0;
    }
    return 1;
}

/* src/bm.h:2431 count() -> src/bmblocks.h:1710 block_bitcount */
static uint32_t block_count(const bmo_vec* v, uint32_t nb)
{
    switch (v->kind[nb]) {
    case BMO_NULL: return 0;
    case BMO_FULL: return BMO_BLOCK_BITS;
    case BMO_BIT:  return bmo_bit_block_count((const uint32_t*)v->blk[nb]);
    default:       return bmo_gap_bit_count((const uint16_t*)v->blk[nb]);
    }
}

uint64_t bmo_vec_count(const bmo_vec* v)
{
    uint64_t c = 0;
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) c += block_count(v, nb);
    return c;
}

int bmo_vec_get_bit(const bmo_vec* v, uint64_t n)
{
    uint32_t nb = (uint32_t)(n >> 16), nbit = (uint32_t)(n & 0xFFFFu);
    if (nb >= v->nblocks) return 0;
    switch (v->kind[nb]) {
    case BMO_NULL: return 0;
    case BMO_FULL: return 1;
    case BMO_BIT:  return (((const uint32_t*)v->blk[nb])[nbit >> 5] >> (nbit & 31u)) & 1u;
    default:       return (int)bmo_gap_test((const uint16_t*)v->blk[nb], nbit);
    }
}

static uint32_t* block_make_bits(bmo_vec* v, uint32_t nb)
{
    if (v->kind[nb] == BMO_BIT) return (uint32_t*)v->blk[nb];
    uint32_t* p = alloc_bit_block();
    uint32_t tmp[BMO_BLOCK_WORDS];
    memcpy(p, block_as_bits(v, nb, tmp), BMO_BLOCK_WORDS * 4);
    free(v->blk[nb]);
    v->blk[nb] = p; v->kind[nb] = BMO_BIT;
    return p;
}

void bmo_vec_set_bit(bmo_vec* v, uint64_t n)
{
    uint32_t nb = (uint32_t)(n >> 16), nbit = (uint32_t)(n & 0xFFFFu);
    if (nb >= v->nblocks) return;
    uint32_t* b = block_make_bits(v, nb);
    b[nbit >> 5] |= 1u << (nbit & 31u);
}

void bmo_vec_set_range(bmo_vec* v, uint64_t l, uint64_t r)
{
    for (uint64_t nb = l >> 16; nb <= (r >> 16) && nb < v->nblocks; ++nb) {
        uint64_t b0 = nb << 16;
        uint32_t from = (uint32_t)(l > b0 ? l - b0 : 0);
        uint32_t to = (uint32_t)(r < b0 + 65535u ? r - b0 : 65535u);
        if (from == 0 && to == 65535u) { block_release(v, (uint32_t)nb); v->kind[nb] = BMO_FULL; continue; }
        uint32_t* b = block_make_bits(v, (uint32_t)nb);
        or_bit_range(b, from, to - from + 1);
    }
}

/* ======================================================================
 * Pairwise operations: SURVEY Appendix A.1 truth table
 * (src/bm.h:7100 AND, 6945 OR, 7018 XOR, 7285 SUB; outer loops :6185,5973,6072,6403)
 * ====================================================================== */
static void store_gap_result(bmo_vec* t, uint32_t nb, const uint16_t* g, unsigned len)
{
    /* blocks_manager::clone_gap_block, src/bmblocks.h:865-889 */
    if (len == 1 && !(g[0] & 1u)) return;                 /* all-zero: leave NULL */
    if (gap_calc_level(len) < 0) {                        /* too long: convert to bit-block */
        uint32_t* p = alloc_bit_block();
        bmo_gap_convert_to_bitset(p, g);
        t->blk[nb] = p; t->kind[nb] = BMO_BIT; return;
    }
    t->blk[nb] = gap_clone(g, len); t->kind[nb] = BMO_GAP;
}

static void clone_block(bmo_vec* t, uint32_t nb, const bmo_vec* s, int invert)
{
    /* blocks_manager::clone_assign_block, src/bmblocks.h:894-937 */
    uint8_t k = s->kind[nb];
    if (k == BMO_NULL) { if (invert) t->kind[nb] = BMO_FULL; return; }
    if (k == BMO_FULL) { if (!invert) t->kind[nb] = BMO_FULL; return; }
    if (k == BMO_BIT) {
        uint32_t* p = alloc_bit_block();
        const uint32_t* q = (const uint32_t*)s->blk[nb];
        for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) p[i] = invert ? ~q[i] : q[i];
        t->blk[nb] = p; t->kind[nb] = BMO_BIT; return;
    }
    const uint16_t* g = (const uint16_t*)s->blk[nb];
    if (gap_len(g) == 1) {                      /* gap_is_all_zero / gap_is_all_one */
        unsigned one = (g[0] & 1u) ^ (unsigned)invert;
        if (one) t->kind[nb] = BMO_FULL;
        return;
    }
    uint16_t* p = gap_clone(g, gap_len(g));
    if (invert) p[0] ^= 1u;                      /* gap_invert */
    t->blk[nb] = p; t->kind[nb] = BMO_GAP;
}

static uint8_t kind_at(const bmo_vec* v, uint32_t nb) { return nb < v->nblocks ? v->kind[nb] : BMO_NULL; }

static void bit_op_block(int op, uint32_t* p, const uint32_t* q)
{
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i) {
        switch (op) {
        case BMO_AND: p[i] &= q[i]; break;
        case BMO_OR:  p[i] |= q[i]; break;
        case BMO_XOR: p[i] ^= q[i]; break;
        default:      p[i] &= ~q[i]; break;
        }
    }
}

/* One block of a 3-operand op; mirrors combine_operation_block_{and,or,xor,sub}
 * (src/bm.h:7100,6945,7018,7285) including which cases test for an empty /
 * saturated result.  Returns 1 when the reference reports "optimization may be
 * needed" (the caller then applies optimize_bit_block under opt_compress). */
static int op2_block(int op, bmo_vec* t, uint32_t nb, const bmo_vec* a, const bmo_vec* b)
{
    static uint32_t ones[BMO_BLOCK_WORDS];
    static int ones_init = 0;
    if (!ones_init) { memset(ones, 0xFF, sizeof(ones)); ones_init = 1; }
    uint16_t gtmp[BMO_BLOCK_BITS / 2 + 16];
    uint8_t ka = kind_at(a, nb), kb = kind_at(b, nb);
    int a_real_full = 0;

    switch (op) {
    case BMO_AND:
        if (ka == BMO_NULL || kb == BMO_NULL) return 0;
        if (ka == BMO_FULL && kb == BMO_FULL) { t->kind[nb] = BMO_FULL; return 0; }
        if (ka == BMO_FULL) { clone_block(t, nb, b, 0); return 0; }
        if (kb == BMO_FULL) { clone_block(t, nb, a, 0); return 0; }
        break;
    case BMO_OR:
        if (ka == BMO_NULL) { if (kb != BMO_NULL) clone_block(t, nb, b, 0); return 0; }
        if (kb == BMO_NULL) { clone_block(t, nb, a, 0); return 0; }
        if (ka == BMO_FULL || kb == BMO_FULL) { t->kind[nb] = BMO_FULL; return 0; }
        break;
    case BMO_XOR:
        if (ka == BMO_NULL) { if (kb != BMO_NULL) clone_block(t, nb, b, 0); return 0; }
        if (kb == BMO_NULL) { clone_block(t, nb, a, 0); return 0; }
        if (ka == BMO_FULL) { clone_block(t, nb, b, 1); return 0; }
        if (kb == BMO_FULL) { clone_block(t, nb, a, 1); return 0; }
        break;
    default: /* SUB */
        if (ka == BMO_NULL) return 0;
        if (kb == BMO_NULL) { clone_block(t, nb, a, 0); return 0; }
        if (kb == BMO_FULL) return 0;
        if (ka == BMO_FULL) a_real_full = 1;          /* FULL_BLOCK_REAL_ADDR: bit-block rules */
        break;
    }

    if (ka == BMO_GAP && kb == BMO_GAP) {
        unsigned len = bmo_gap_op(op, (const uint16_t*)a->blk[nb], (const uint16_t*)b->blk[nb],
                                  gtmp, sizeof(gtmp) / 2);
        store_gap_result(t, nb, gtmp, len);
        return 0;
    }
    uint32_t* p = alloc_bit_block();
    if (ka == BMO_GAP) {                              /* G op B */
        const uint16_t* g = (const uint16_t*)a->blk[nb];
        const uint32_t* q = (const uint32_t*)b->blk[nb];
        if (op == BMO_SUB) {                          /* expand G, bit_block_sub; acc==0 => empty */
            bmo_gap_convert_to_bitset(p, g);
            bit_op_block(BMO_SUB, p, q);
            if (block_is_all_zero(p)) { free(p); return 0; }
        } else {                                      /* clone B, apply G */
            memcpy(p, q, BMO_BLOCK_WORDS * 4);
            if (op == BMO_AND) { gap_and_to_bitset(p, g); if (block_is_all_zero(p)) { free(p); return 0; } }
            else if (op == BMO_OR) gap_add_to_bitset(p, g);
            else gap_xor_to_bitset(p, g);
        }
    } else if (kb == BMO_GAP) {                       /* B op G : clone B, apply G */
        const uint16_t* g = (const uint16_t*)b->blk[nb];
        memcpy(p, a_real_full ? ones : (const uint32_t*)a->blk[nb], BMO_BLOCK_WORDS * 4);
        switch (op) {
        case BMO_AND: gap_and_to_bitset(p, g); if (block_is_all_zero(p)) { free(p); return 0; } break;
        case BMO_OR:  gap_add_to_bitset(p, g); break;
        case BMO_XOR: gap_xor_to_bitset(p, g); break;
        default:      gap_sub_to_bitset(p, g); break;   /* no emptiness test (:7336-7339) */
        }
    } else {                                          /* B op B */
        memcpy(p, a_real_full ? ones : (const uint32_t*)a->blk[nb], BMO_BLOCK_WORDS * 4);
        bit_op_block(op, p, (const uint32_t*)b->blk[nb]);
        if (op == BMO_OR) { if (block_is_all_one(p)) { free(p); t->kind[nb] = BMO_FULL; return 0; } }
        else if (block_is_all_zero(p)) { free(p); return 0; }
    }
    t->blk[nb] = p; t->kind[nb] = BMO_BIT;
    return 1;
}

bmo_vec* bmo_op2(int op, const bmo_vec* a, const bmo_vec* b, int opt_compress)
{
    uint64_t nbits = a->nbits > b->nbits ? a->nbits : b->nbits;   /* src/bm.h:6219-6221 */
    bmo_vec* t = bmo_vec_new(nbits);
    if (a == b) {
        /* aliasing, handled up front by the reference: bit_and(bv, bv) is a plain copy (src/bm.h:6191-6195),
         * bit_or(bv, bv) ORs bv into the emptied target = copies every block as it is (:5984-5988 with the
         * "0,x -> copy x" rule), bit_xor / bit_sub of a vector with itself are empty (:6081-6082, 6412-6413).
         * No block is re-optimised on these paths whatever opt_mode says. */
        if (op == BMO_AND || op == BMO_OR)
            for (uint32_t nb = 0; nb < t->nblocks; ++nb) clone_block(t, nb, a, 0);
        return t;
    }
    for (uint32_t nb = 0; nb < t->nblocks; ++nb) {
        int need_opt = op2_block(op, t, nb, a, b);
        if (need_opt && opt_compress && t->kind[nb] == BMO_BIT) {   /* optimize_bit_block, src/bmblocks.h:1412 */
            uint32_t* p = (uint32_t*)t->blk[nb];
            t->blk[nb] = NULL; t->kind[nb] = BMO_NULL;
            store_bit_block(t, nb, p, 1);
            free(p);
        }
    }
    return t;
}

/* bm::count_and/or/xor/sub without materialising
 * (src/bmalgo_impl.h:189-434) */
uint64_t bmo_count_op2(int op, const bmo_vec* a, const bmo_vec* b)
{
    static uint32_t ones[BMO_BLOCK_WORDS];
    static int ones_init = 0;
    if (!ones_init) { memset(ones, 0xFF, sizeof(ones)); ones_init = 1; }
    uint32_t n = a->nblocks > b->nblocks ? a->nblocks : b->nblocks;
    uint64_t total = 0;
    for (uint32_t nb = 0; nb < n; ++nb) {
        uint8_t ka = kind_at(a, nb), kb = kind_at(b, nb);
        if (ka == BMO_NULL && kb == BMO_NULL) continue;
        if (op == BMO_AND && (ka == BMO_NULL || kb == BMO_NULL)) continue;
        if (op == BMO_SUB && ka == BMO_NULL) continue;
        if (ka == BMO_NULL) { total += block_count(b, nb); continue; }      /* OR / XOR */
        if (kb == BMO_NULL) { total += block_count(a, nb); continue; }      /* OR / XOR / SUB */
        /* FULL is replaced by the real all-ones block (BLOCK_ADDR_SAN, src/bmdef.h:168) */
        const uint32_t* pa = ka == BMO_FULL ? ones : (ka == BMO_BIT ? (const uint32_t*)a->blk[nb] : NULL);
        const uint32_t* pb = kb == BMO_FULL ? ones : (kb == BMO_BIT ? (const uint32_t*)b->blk[nb] : NULL);
        if (!pa && !pb) { total += gap_count_op(op, (const uint16_t*)a->blk[nb], (const uint16_t*)b->blk[nb]); continue; }
        if (pa && !pb) { total += gap_bitset_count_op(op, pa, (const uint16_t*)b->blk[nb], 1); continue; }
        if (!pa && pb) { total += gap_bitset_count_op(op, pb, (const uint16_t*)a->blk[nb], 0); continue; }
        /* bit x bit: src/bmfunc.h:8031,8107,8181,8255 */
        uint32_t c = 0;
        const uint64_t* xa = (const uint64_t*)pa; const uint64_t* xb = (const uint64_t*)pb;
        for (unsigned i = 0; i < BMO_BLOCK_WORDS / 2; ++i) {
            switch (op) {
            case BMO_AND: c += POP64(xa[i] & xb[i]); break;
            case BMO_OR:  c += POP64(xa[i] | xb[i]); break;
            case BMO_XOR: c += POP64(xa[i] ^ xb[i]); break;
            default:      c += POP64(xa[i] & ~xb[i]); break;
            }
        }
        total += c;
    }
    return total;
}

/* ======================================================================
 * Aggregator: SURVEY Appendix A.3
 * ====================================================================== */
typedef struct {
    const uint32_t** bit; size_t nbit;
    const uint16_t** gap; size_t ngap;
} arg_list;

/* one block column of combine_and_sub (src/bmaggregator.h:1720): returns the
 * digest of the result held in tb1, or sets *is_full.  Evaluation order and
 * early exits follow :1764-1798 and process_bit_blocks_and/sub (:1994,2125);
 * the single-bit probe mode (:2052-2086) is a pure optimisation and omitted. */
static uint64_t and_sub_column(uint32_t nb,
                               const bmo_vec* const* src_and, size_t n_and,
                               const bmo_vec* const* src_sub, size_t n_sub,
                               arg_list* A, arg_list* S, uint32_t* tb1, int* is_full)
{
    *is_full = 0;
    /* sort_input_blocks_and (:2315): any NULL operand => empty column */
    A->nbit = A->ngap = 0;
    int has_full = 0;
    for (size_t k = 0; k < n_and; ++k)
        if (kind_at(src_and[k], nb) == BMO_NULL) return 0;
    for (size_t k = 0; k < n_and; ++k) {
        const bmo_vec* v = src_and[k];
        switch (v->kind[nb]) {
        case BMO_GAP:  A->gap[A->ngap++] = (const uint16_t*)v->blk[nb]; break;
        case BMO_FULL: has_full = 1; break;
        default:       A->bit[A->nbit++] = (const uint32_t*)v->blk[nb]; break;
        }
    }
    int all_full = has_full && !A->nbit && !A->ngap;
    if (!all_full && !(A->nbit | A->ngap)) return 0;
    /* sort_input_blocks_or for the SUB group (:2278): any FULL => empty column */
    S->nbit = S->ngap = 0;
    for (size_t k = 0; k < n_sub; ++k) {
        const bmo_vec* v = src_sub[k];
        switch (kind_at(v, nb)) {
        case BMO_NULL: break;
        case BMO_FULL: return 0;
        case BMO_GAP:  S->gap[S->ngap++] = (const uint16_t*)v->blk[nb]; break;
        default:       S->bit[S->nbit++] = (const uint32_t*)v->blk[nb]; break;
        }
    }
    if (all_full && !n_sub) { *is_full = 1; return ~0ull; }           /* :1751-1757 */

    /* process_bit_blocks_and (:1994) */
    uint64_t digest = ~0ull;
    size_t k = 0;
    if (all_full || A->nbit == 0) { block_init_digest0(tb1, digest); }
    else if (A->nbit == 1) { memcpy(tb1, A->bit[0], BMO_BLOCK_WORDS * 4); digest = bmo_calc_block_digest0(tb1); k = 1; }
    else {
        digest = bmo_and_2way(tb1, A->bit[0], A->bit[1], ~0ull); k = 2;
        for (; k + 4 < A->nbit && digest; k += 4)
            digest = bmo_and_5way(tb1, A->bit[k], A->bit[k + 1], A->bit[k + 2], A->bit[k + 3], digest);
        for (; k + 2 < A->nbit && digest; k += 2)
            digest = and_3way(tb1, A->bit[k], A->bit[k + 1], digest);
        for (; k < A->nbit && digest; ++k)
            digest = and_1(tb1, A->bit[k], digest);
    }
    if (!digest) return 0;
    /* process_bit_blocks_sub (:2125) */
    k = 0;
    for (; k + 4 < S->nbit && digest; k += 4)
        digest = sub_5way(tb1, S->bit[k], S->bit[k + 1], S->bit[k + 2], S->bit[k + 3], digest);
    for (; k + 2 < S->nbit && digest; k += 2)
        digest = sub_3way(tb1, S->bit[k], S->bit[k + 1], digest);
    for (; k < S->nbit && digest; ++k)
        digest = sub_1(tb1, S->bit[k], digest);
    /* process_gap_blocks_and / _sub (:1820,1854) */
    for (k = 0; k < A->ngap && digest; ++k)
        digest = bmo_gap_and_to_bitset_digest(tb1, A->gap[k], digest);
    for (k = 0; k < S->ngap && digest; ++k)
        digest = bmo_gap_sub_to_bitset_digest(tb1, S->gap[k], digest);
    return digest;
}

static void arg_list_init(arg_list* L, size_t n)
{
    L->bit = (const uint32_t**)malloc((n + 1) * sizeof(void*));
    L->gap = (const uint16_t**)malloc((n + 1) * sizeof(void*));
    L->nbit = L->ngap = 0;
}
static void arg_list_free(arg_list* L) { free(L->bit); free(L->gap); }

/* clear the waves of tb1 that are not in the digest: the reference leaves them
 * stale because every consumer is digest-driven; a stored block must be clean
 * (bit_block_and_2way "does not touch non-digest waves") */
static void zero_non_digest(uint32_t* tb1, uint64_t digest)
{
    for (unsigned w = 0; w < 64; ++w)
        if (!((digest >> w) & 1)) memset(tb1 + w * BMO_WAVE_WORDS, 0, BMO_WAVE_WORDS * 4);
}

static uint32_t max_blocks(const bmo_vec* const* v, size_t n, uint64_t* nbits)
{
    uint32_t m = 0; *nbits = 0;
    for (size_t i = 0; i < n; ++i) {
        if (v[i]->nblocks > m) m = v[i]->nblocks;
        if (v[i]->nbits > *nbits) *nbits = v[i]->nbits;
    }
    return m;
}

/* aggregator::combine_and_sub(target, and[], n, sub[], n, false)  src/bmaggregator.h:1162.
 * Result blocks are always stored with opt_compress (:1210-1211). */
bmo_vec* bmo_agg_and_sub(const bmo_vec* const* src_and, size_t n_and,
                         const bmo_vec* const* src_sub, size_t n_sub)
{
    uint64_t nbits, nbits2;
    uint32_t nblocks = max_blocks(src_and, n_and, &nbits);
    (void)max_blocks(src_sub, n_sub, &nbits2);
    if (nbits2 > nbits) nbits = nbits2;
    bmo_vec* t = bmo_vec_new(nbits);
    if (!n_and) return t;                                   /* :1170-1174 */
    arg_list A, S; arg_list_init(&A, n_and); arg_list_init(&S, n_sub);
    uint32_t* tb1 = alloc_bit_block();
    for (uint32_t nb = 0; nb < nblocks && nb < t->nblocks; ++nb) {
        int is_full;
        uint64_t digest = and_sub_column(nb, src_and, n_and, src_sub, n_sub, &A, &S, tb1, &is_full);
        if (is_full) { t->kind[nb] = BMO_FULL; continue; }
        if (!digest) continue;
        zero_non_digest(tb1, digest);
        store_bit_block(t, nb, tb1, 1);
    }
    free(tb1); arg_list_free(&A); arg_list_free(&S);
    return t;
}

void bmo_agg_pipeline_results(const bmo_vec* const* and_list, const uint32_t* and_n,
                              const bmo_vec* const* sub_list, const uint32_t* sub_n, size_t ngroups,
                              bmo_vec** results_out, uint64_t* counts_out, bmo_vec** or_target_out)
{
    size_t ao = 0, so = 0;
    uint64_t nbits = 0;
    bmo_vec** tmp = (bmo_vec**)calloc(ngroups ? ngroups : 1, sizeof(*tmp));
    size_t nres = 0;
    for (size_t g = 0; g < ngroups; ao += and_n[g], so += sub_n[g], ++g) {
        results_out[g] = NULL; counts_out[g] = 0;
        if (!and_n[g]) continue;                                   /* :1352-1354 */
        bmo_vec* t = bmo_agg_and_sub(and_list + ao, and_n[g], sub_list + so, sub_n[g]);
        if (t->nbits > nbits) nbits = t->nbits;
        uint64_t c = bmo_vec_count(t);
        counts_out[g] = c;
        if (c) { results_out[g] = t; tmp[nres++] = t; } else bmo_vec_free(t);   /* lazily created: NULL if nothing found */
    }
    bmo_vec* ort = bmo_agg_or((const bmo_vec* const*)tmp, nres);
    if (ort->nbits < nbits) ort->nbits = nbits;
    bmo_vec_optimize(ort);
    *or_target_out = ort;
    free(tmp);
}

static int block_find_first(const uint32_t* b, uint32_t* bit)      /* src/bmfunc.h:9499 bit_find_first */
{
    for (unsigned i = 0; i < BMO_BLOCK_WORDS; ++i)
        if (b[i]) { *bit = i * 32u + (uint32_t)__builtin_ctz(b[i]); return 1; }
    return 0;
}

int bmo_vec_find_first(const bmo_vec* v, uint64_t* pos)
{
    uint32_t tmp[BMO_BLOCK_WORDS];
    for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
        if (v->kind[nb] == BMO_NULL) continue;
        uint32_t bit;
        if (block_find_first(block_as_bits(v, nb, tmp), &bit)) { *pos = ((uint64_t)nb << 16) + bit; return 1; }
    }
    return 0;
}

/* aggregator::find_first_and_sub  src/bmaggregator.h:1458 (per column :1534-1547) */
int bmo_find_first_and_sub(const bmo_vec* const* src_and, size_t n_and,
                           const bmo_vec* const* src_sub, size_t n_sub, uint64_t* idx)
{
    uint64_t nbits;
    uint32_t nblocks = max_blocks(src_and, n_and, &nbits);
    if (!n_and) return 0;
    arg_list A, S; arg_list_init(&A, n_and); arg_list_init(&S, n_sub);
    uint32_t* tb1 = alloc_bit_block();
    int found = 0;
    for (uint32_t nb = 0; nb < nblocks && !found; ++nb) {
        int is_full;
        uint64_t digest = and_sub_column(nb, src_and, n_and, src_sub, n_sub, &A, &S, tb1, &is_full);
        if (is_full) { *idx = (uint64_t)nb << 16; found = 1; break; }
        if (!digest) continue;
        zero_non_digest(tb1, digest);
        uint32_t bit;
        if (block_find_first(tb1, &bit)) { *idx = ((uint64_t)nb << 16) + bit; found = 1; }
    }
    free(tb1); arg_list_free(&A); arg_list_free(&S);
    return found;
}

/* aggregator::combine_or  src/bmaggregator.h:1101, per column :1626;
 * result stored with opt_mode_ = opt_none (:917,1658) */
bmo_vec* bmo_agg_or(const bmo_vec* const* src, size_t n) { return bmo_agg_or_opt(src, n, 0); }

/* opt_compress != 0: aggregator::set_optimization() (:359) before combine_or -- blocks go through
 * opt_copy_bit_block(.., opt_mode_, ..) (:1658, src/bmblocks.h:1355) */
bmo_vec* bmo_agg_or_opt(const bmo_vec* const* src, size_t n, int opt_compress)
{
    uint64_t nbits;
    uint32_t nblocks = max_blocks(src, n, &nbits);
    bmo_vec* t = bmo_vec_new(nbits);
    if (!n) return t;                                       /* :1105-1109 */
    arg_list L; arg_list_init(&L, n);
    uint32_t* tb1 = alloc_bit_block();
    for (uint32_t nb = 0; nb < nblocks && nb < t->nblocks; ++nb) {
        /* sort_input_blocks_or (:2278) */
        L.nbit = L.ngap = 0; int full = 0;
        for (size_t k = 0; k < n && !full; ++k) {
            switch (kind_at(src[k], nb)) {
            case BMO_NULL: break;
            case BMO_FULL: full = 1; break;
            case BMO_GAP:  L.gap[L.ngap++] = (const uint16_t*)src[k]->blk[nb]; break;
            default:       L.bit[L.nbit++] = (const uint32_t*)src[k]->blk[nb]; break;
            }
        }
        if (full) { t->kind[nb] = BMO_FULL; continue; }
        if (!(L.nbit | L.ngap)) continue;
        /* process_bit_blocks_or (:1924) */
        size_t k = 0; int all_one = 0;
        if (L.nbit) memcpy(tb1, L.bit[k++], BMO_BLOCK_WORDS * 4);
        else memset(tb1, 0, BMO_BLOCK_WORDS * 4);
        for (; k + 4 <= L.nbit && !all_one; k += 4)
            all_one = or_5way(tb1, L.bit[k], L.bit[k + 1], L.bit[k + 2], L.bit[k + 3]);
        for (; k + 2 <= L.nbit && !all_one; k += 2)
            all_one = or_3way(tb1, L.bit[k], L.bit[k + 1]);
        for (; k < L.nbit && !all_one; ++k)
            all_one = or_1(tb1, L.bit[k]);
        if (all_one) { t->kind[nb] = BMO_FULL; continue; }
        /* process_gap_blocks_or (:1808) */
        for (k = 0; k < L.ngap; ++k) gap_add_to_bitset(tb1, L.gap[k]);
        store_bit_block(t, nb, tb1, opt_compress);   /* opt_none: plain copy, even if empty */
    }
    free(tb1); arg_list_free(&L);
    return t;
}

/* aggregator::combine_shift_right_and  src/bmaggregator.h:2494-2529 (outer loop with the per-operand
 * carry_overs[] array, :2479-2489), one block column :2534-2606, one stage process_shift_right_and
 * :2611-2669 (fused shift+AND bit_block_shift_r1_and_unr; NULL argument: carry out, block zeroed :2655-2663;
 * GAP argument: shift against the all-ones block then gap_and_to_bitset :2622-2640).
 * Restated without the digest bookkeeping (the digest only skips work on waves that are already zero;
 * the ":2579 0 into 00000 block" skip is kept because it also leaves carry_overs[k] untouched).
 * count != NULL mirrors set_compute_count(true) (:363,2593-2597): nothing is stored, *count = count().
 * any: return at the first column that produced a block (:2519). */
static int shift_right_and_run(const bmo_vec* const* src, size_t n, int opt_compress, int any,
                               bmo_vec* t, uint64_t* count)
{
    uint64_t nbits;
    uint32_t nblocks = max_blocks(src, n, &nbits);
    unsigned char* carry_overs = (unsigned char*)calloc(n ? n : 1, 1);
    uint32_t* blk = alloc_bit_block();
    uint32_t* arg = alloc_bit_block();
    int found_any = 0;
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        /* first operand: plain copy (:2546-2570) */
        int blk_zero = 0;
        switch (kind_at(src[0], nb)) {
        case BMO_NULL: memset(blk, 0, BMO_BLOCK_WORDS * 4); blk_zero = 1; break;
        case BMO_FULL: memset(blk, 0xFF, BMO_BLOCK_WORDS * 4); break;
        case BMO_GAP:  bmo_gap_convert_to_bitset(blk, (const uint16_t*)src[0]->blk[nb]); break;
        default:       memcpy(blk, src[0]->blk[nb], BMO_BLOCK_WORDS * 4); break;
        }
        carry_overs[0] = 0;
        for (size_t k = 1; k < n; ++k) {
            unsigned co = carry_overs[k];
            if (blk_zero && !co) continue;                                  /* :2579 */
            /* shift right by one (bit p -> p+1), carry in at bit 0, carry out from bit 65535 */
            unsigned co_out = blk[BMO_BLOCK_WORDS - 1] >> 31;
            for (uint32_t w = BMO_BLOCK_WORDS - 1; w > 0; --w) blk[w] = (blk[w] << 1) | (blk[w - 1] >> 31);
            blk[0] = (blk[0] << 1) | co;
            switch (kind_at(src[k], nb)) {
            case BMO_NULL: memset(blk, 0, BMO_BLOCK_WORDS * 4); break;     /* :2655-2663 */
            case BMO_FULL: break;                                          /* get_block() hands out the real all-ones block */
            case BMO_GAP:
                bmo_gap_convert_to_bitset(arg, (const uint16_t*)src[k]->blk[nb]);
                for (uint32_t w = 0; w < BMO_BLOCK_WORDS; ++w) blk[w] &= arg[w];
                break;
            default: {
                const uint32_t* a = (const uint32_t*)src[k]->blk[nb];
                for (uint32_t w = 0; w < BMO_BLOCK_WORDS; ++w) blk[w] &= a[w];
                break; }
            }
            carry_overs[k] = (unsigned char)co_out;
            blk_zero = 1;
            for (uint32_t w = 0; w < BMO_BLOCK_WORDS; ++w) if (blk[w]) { blk_zero = 0; break; }
        }
        if (!blk_zero) {                                                    /* :2590-2604 */
            if (count) *count += bmo_bit_block_count(blk);
            else store_bit_block(t, nb, blk, opt_compress);
            found_any = 1;
            if (any) break;
        }
    }
    free(blk); free(arg); free(carry_overs);
    return found_any;
}

bmo_vec* bmo_agg_shift_right_and(const bmo_vec* const* src, size_t n, int opt_compress, int any, int* found)
{
    uint64_t nbits;
    (void)max_blocks(src, n, &nbits);
    bmo_vec* t = bmo_vec_new(nbits);
    int f = n ? shift_right_and_run(src, n, opt_compress, any, t, NULL) : 0;   /* empty list: cleared target (:2499) */
    if (found) *found = f;
    return t;
}

uint64_t bmo_agg_shift_right_and_count(const bmo_vec* const* src, size_t n)
{
    uint64_t c = 0;
    if (n) (void)shift_right_and_run(src, n, 0, 0, NULL, &c);
    return c;
}

/* counts-only pipeline: src/bmaggregator.h:1292-1399
 *   count[p] += is_full ? 65536 : bit_block_count(tb1, digest) */
void bmo_agg_pipeline_counts(const bmo_vec* const* and_list, const uint32_t* and_n,
                             const bmo_vec* const* sub_list, const uint32_t* sub_n,
                             size_t ngroups, uint32_t nb_from, uint32_t nb_to,
                             uint64_t* counts_out)
{
    size_t max_n = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        if (and_n[g] > max_n) max_n = and_n[g];
        if (sub_n[g] > max_n) max_n = sub_n[g];
        counts_out[g] = 0;
    }
    arg_list A, S; arg_list_init(&A, max_n); arg_list_init(&S, max_n);
    uint32_t* tb1 = alloc_bit_block();
    /* block column outermost, arg-group innermost (:1326-1351) */
    for (uint32_t nb = nb_from; nb < nb_to; ++nb) {
        size_t ao = 0, so = 0;
        for (size_t g = 0; g < ngroups; ao += and_n[g], so += sub_n[g], ++g) {
            if (!and_n[g]) continue;
            int is_full;
            uint64_t digest = and_sub_column(nb, and_list + ao, and_n[g], sub_list + so, sub_n[g],
                                             &A, &S, tb1, &is_full);
            if (is_full) counts_out[g] += BMO_BLOCK_BITS;
            else if (digest) counts_out[g] += bmo_bit_block_count_digest(tb1, digest);
        }
    }
    free(tb1); arg_list_free(&A); arg_list_free(&S);
}

/* ======================================================================
 * Rank / select: SURVEY Appendix A.5.  The index keeps, per block, the
 * population count and the reference's packed sub-count word
 * (src/bm.h:2646-2656), plus running totals (rs_index::rcount, src/bmrs.h:361).
 * ====================================================================== */
struct bmo_rs {
    uint32_t total_blocks;
    uint32_t* bcount;      /* per-block popcount */
    uint64_t* sub;         /* first | second<<16 | aux0<<32 | aux1<<48 */
    uint64_t* rcount;      /* inclusive running count */
    uint64_t  count;
};

bmo_rs* bmo_rs_build(const bmo_vec* v)
{
    bmo_rs* rs = (bmo_rs*)calloc(1, sizeof(*rs));
    uint32_t n = v->nblocks;
    rs->total_blocks = n;
    rs->bcount = (uint32_t*)calloc(n ? n : 1, 4);
    rs->sub = (uint64_t*)calloc(n ? n : 1, 8);
    rs->rcount = (uint64_t*)calloc(n ? n : 1, 8);
    static uint32_t ones[BMO_BLOCK_WORDS];
    memset(ones, 0xFF, sizeof(ones));
    uint64_t run = 0;
    for (uint32_t nb = 0; nb < n; ++nb) {
        uint32_t first = 0, second = 0, third = 0; uint64_t aux0 = 0, aux1 = 0;
        if (v->kind[nb] == BMO_GAP) {
            const uint16_t* g = (const uint16_t*)v->blk[nb];
            first = bmo_gap_bit_count_to(g, BMO_RS3_BORDER0);
            second = gap_bit_count_range(g, BMO_RS3_BORDER0 + 1, BMO_RS3_BORDER1);
            third = gap_bit_count_range(g, BMO_RS3_BORDER1 + 1, 65535u);
            unsigned is_set;
            aux0 = (uint64_t)bmo_gap_bfind(g, BMO_RS3_BORDER0 + 1, &is_set) << 1; aux0 |= is_set;
            aux1 = (uint64_t)bmo_gap_bfind(g, BMO_RS3_BORDER1 + 1, &is_set) << 1; aux1 |= is_set;
        } else if (v->kind[nb] != BMO_NULL) {
            const uint32_t* b = v->kind[nb] == BMO_FULL ? ones : (const uint32_t*)v->blk[nb];
            first = bmo_bit_block_count_range(b, 0, BMO_RS3_BORDER0);
            second = bmo_bit_block_count_range(b, BMO_RS3_BORDER0 + 1, BMO_RS3_BORDER1);
            third = bmo_bit_block_count_range(b, BMO_RS3_BORDER1 + 1, 65535u);
            aux0 = bmo_bit_block_count_range(b, 0, BMO_RS3_BORDER0 + BMO_RS3_HALF_SPAN);
            aux1 = bmo_bit_block_count_range(b, 0, BMO_RS3_BORDER1 + BMO_RS3_HALF_SPAN);
        }
        rs->bcount[nb] = first + second + third;
        rs->sub[nb] = (uint64_t)(first | (second << 16)) | ((aux0 & 0xFFFFu) << 32) | ((aux1 & 0xFFFFu) << 48);
        run += rs->bcount[nb];
        rs->rcount[nb] = run;
    }
    rs->count = run;
    return rs;
}

void bmo_rs_free(bmo_rs* rs) { if (rs) { free(rs->bcount); free(rs->sub); free(rs->rcount); free(rs); } }
uint64_t bmo_rs_count(const bmo_rs* rs) { return rs->count; }
uint32_t bmo_rs_total_blocks(const bmo_rs* rs) { return rs->total_blocks; }
void bmo_rs_export(const bmo_rs* rs, uint32_t* bcount, uint64_t* sub_count)
{
    memcpy(bcount, rs->bcount, (size_t)rs->total_blocks * 4);
    memcpy(sub_count, rs->sub, (size_t)rs->total_blocks * 8);
}

/* bvector::count_to / rank: ones in [0..n] inclusive (src/bm.h:3120-3167) */
uint64_t bmo_rank(const bmo_vec* v, const bmo_rs* rs, uint64_t n)
{
    uint32_t nb = (uint32_t)(n >> 16), nbit = (uint32_t)(n & 0xFFFFu);
    if (nb >= rs->total_blocks) return rs->count;
    uint64_t c = nb ? rs->rcount[nb - 1] : 0;
    switch (v->kind[nb]) {
    case BMO_NULL: return c;
    case BMO_FULL: return c + nbit + 1;
    case BMO_BIT:  return c + bmo_bit_block_count_range((const uint32_t*)v->blk[nb], 0, nbit);
    default:       return c + bmo_gap_bit_count_to((const uint16_t*)v->blk[nb], nbit);
    }
}

/* bvector::select: position of the rank-th (1-based) set bit (src/bm.h:5350)
 * -> rs_index::find (src/bmrs.h:492) -> block_find_rank (src/bmfunc.h:9754) */
int bmo_select(const bmo_vec* v, const bmo_rs* rs, uint64_t rank, uint64_t* pos)
{
    if (!rank || rank > rs->count) return 0;
    /* lower_bound over the running counts */
    uint32_t lo = 0, hi = rs->total_blocks - 1;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (rs->rcount[mid] < rank) lo = mid + 1; else hi = mid;
    }
    uint32_t nb = lo;
    uint32_t r = (uint32_t)(rank - (nb ? rs->rcount[nb - 1] : 0));   /* 1..65536 */
    uint32_t bit = 0;
    switch (v->kind[nb]) {
    case BMO_FULL: bit = r - 1; break;
    case BMO_BIT: {
        const uint64_t* b = (const uint64_t*)v->blk[nb];
        unsigned i = 0;
        for (;; ++i) { uint32_t pc = POP64(b[i]); if (r <= pc) break; r -= pc; }
        uint64_t w = b[i];
        for (uint32_t s = 1; s < r; ++s) w &= w - 1;                /* word_select64, src/bmfunc.h:1057 */
        bit = i * 64u + (uint32_t)__builtin_ctzll(w);
        break; }
    default: {   /* gap_find_rank, src/bmfunc.h:3457 */
        const uint16_t* g = (const uint16_t*)v->blk[nb];
        unsigned len = gap_len(g), val = g[0] & 1u; uint32_t start = 0;
        for (unsigned k = 1; k <= len; ++k) {
            uint32_t end = g[k];
            if (val) { uint32_t n = end - start + 1; if (r <= n) { bit = start + r - 1; break; } r -= n; }
            start = end + 1; val ^= 1u;
        }
        break; }
    }
    *pos = ((uint64_t)nb << 16) + bit;
    return 1;
}

/* src/bm.h:3548: count_range(l, r) = rank(r) - rank(l-1); arguments are swapped when l > r (:3554) */
uint64_t bmo_count_range(const bmo_vec* v, const bmo_rs* rs, uint64_t left, uint64_t right)
{
    if (left > right) { uint64_t t = left; left = right; right = t; }
    if (left == right) return (uint64_t)bmo_vec_get_bit(v, left);
    return bmo_rank(v, rs, right) - (left ? bmo_rank(v, rs, left - 1) : 0);
}
/* src/bm.h:3229: rank(n) - bit(n) */
uint64_t bmo_rank_corrected(const bmo_vec* v, const bmo_rs* rs, uint64_t n)
{ return bmo_rank(v, rs, n) - (uint64_t)bmo_vec_get_bit(v, n); }
/* src/bm.h:3173: bit(n) ? rank(n) : 0 */
uint64_t bmo_count_to_test(const bmo_vec* v, const bmo_rs* rs, uint64_t n)
{ return bmo_vec_get_bit(v, n) ? bmo_rank(v, rs, n) : 0; }
/* src/bm.h:5279 find_rank(rank, from, pos, rs): position of the rank-th set bit at or after `from` */
int bmo_find_rank(const bmo_vec* v, const bmo_rs* rs, uint64_t rank, uint64_t from, uint64_t* pos)
{
    if (!rank) return 0;
    uint64_t before = from ? bmo_rank(v, rs, from - 1) : 0;
    return bmo_select(v, rs, rank + before, pos);
}

void bmo_rank_batch(const bmo_vec* v, const bmo_rs* rs, const uint64_t* n, size_t q, uint64_t* out)
{
    for (size_t i = 0; i < q; ++i) out[i] = bmo_rank(v, rs, n[i]);
}

void bmo_select_batch(const bmo_vec* v, const bmo_rs* rs, const uint64_t* r, size_t q,
                      uint64_t* pos, uint8_t* found)
{
    for (size_t i = 0; i < q; ++i) { pos[i] = 0; found[i] = (uint8_t)bmo_select(v, rs, r[i], &pos[i]); }
}
