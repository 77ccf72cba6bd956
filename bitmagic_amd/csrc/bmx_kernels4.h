// bmx_kernels4.h -- bit-sliced comparison search over resident slices: the range-search half of the
// sparse_vector_scanner call pattern.
#pragma once
#include "bmx_kernels3.h"

// ---------------------------------------------------------------------------
// Reference (unsigned value types, src/bmsparsevec_algo.h):
//   find_gt  :2690 -> find_gt_horizontal_u :2914   OR of the planes above the value's top bit ("surely greater"),
//                                                  then, walking the set bits of the value from the top: eq &= plane[bit];
//                                                  for every ZERO bit below it: out |= eq & plane[j]   (bit_or_and)
//   find_ge  :2717 -> find_gt(val - 1), val == 0 -> every row [0, size)                            (:2730-2786)
//   find_lt  :2790 = [0, size) - find_ge(val)      find_le :2824 = [0, size) - find_gt(val)
//   find_range :2862 = find_ge(from) - find_gt(to)
//   find_zero :2290 = [0, size) - OR(planes)       find_nonzero :4464 = OR(planes)
//   NULL elements are stored as 0: results that can contain value 0 are AND-ed with the not-NULL vector
//   (correct_nulls :2376, needs_null_correct_* :1703-1735).
// That is a chain of whole-vector OR / AND / SUB passes (one read + one write of a temporary per step).
// Here: ONE pass.  A wave owns a block column and walks the planes from the top bit down keeping two
// register blocks per bound -- gt (rows already known to be greater) and eq (rows equal so far):
//     bit of the bound set   : eq &= P            bit clear : gt |= eq & P;  eq &= ~P
// Every plane block is read exactly once, nothing intermediate touches HBM, and the walk stops as soon as
// no row is "equal so far" for any bound (the remaining planes cannot change the answer).
//   GT = gt    GE = gt | eq    LT = ~(gt | eq)    LE = ~gt    RANGE = (gt0 | eq0) & ~gt1    EQ = eq
//   NONZERO = ~eq(0)  ZERO = eq(0)          -- all restricted to rows [0, size) and, where flagged, to not-NULL rows.
// Results are identical sets: both compute {i : sv[i] OP value} over the same planes.
// ---------------------------------------------------------------------------
enum { CMP_GT = 0, CMP_GE = 1, CMP_LT = 2, CMP_LE = 3, CMP_RANGE = 4, CMP_EQ = 5, CMP_ZERO = 6, CMP_NONZERO = 7,
       CMP_SRANGE = 8 /* signed range across zero: magnitude <= v0 where the sign plane is clear, <= v1 where it is set */ };

// Signed containers (bm::sparse_vector<int, ..>): plane 0 is the SIGN, planes 1.. hold the magnitude -- v >= 0 is stored
// as v << 1, v < 0 as ((-(v + 1)) << 1) | 1 (base_sparse_vector::s2u, src/bmbmatrix.h:2536-2548).  A signed comparison is
// an unsigned comparison of the magnitude planes on ONE side of the sign and all / none of the rows on the other (the
// reference builds it from whole-vector passes: find_gt_horizontal_s, src/bmsparsevec_algo.h:3033-3160): the same one-pass
// walk over planes 1.., then one of these combinations with the sign block S.
enum { SIGN_NONE = 0,
       SIGN_NONNEG_ONLY = 1,      // r & ~S          rows >= 0 that satisfy the magnitude predicate
       SIGN_NONNEG_ALL = 2,       // (r & S) | ~S    every row >= 0, plus the negative ones that satisfy it
       SIGN_NEG_ONLY = 3,         // r & S
       SIGN_NEG_ALL = 4 };        // (r & ~S) | S

__device__ __forceinline__ u32x4 sign_combine(u32x4 r, u32x4 S, int mode)
{
    switch (mode) {
    case SIGN_NONNEG_ONLY: return r & ~S;
    case SIGN_NONNEG_ALL:  return (r & S) | ~S;
    case SIGN_NEG_ONLY:    return r & S;
    case SIGN_NEG_ALL:     return (r & ~S) | S;
    default:               return r;
    }
}

// rows [0, size) of block nb as a register image
__device__ __forceinline__ void blk_size_mask(Blk& m, u32 nb, u64 size, u32 lane)
{
    u64 base = (u64)nb << 16;
    if (size >= base + 65536ull) { blk_fill(m, ~0u); return; }
    if (size <= base) { blk_fill(m, 0u); return; }
    u32 r = (u32)(size - base);                                // 1..65535 rows of this block are inside
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 w0 = ((u32)i * 256u + lane * 4u) << 5;             // first bit of word .x
        u32 ws[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 lo = w0 + (u32)j * 32u;
            ws[j] = r >= lo + 32u ? ~0u : (r <= lo ? 0u : ((1u << (r - lo)) - 1u));
        }
        m.r[i].x = ws[0]; m.r[i].y = ws[1]; m.r[i].z = ws[2]; m.r[i].w = ws[3];
    }
}

// TWO: two bounds at once (find_range); the one-bound form carries half the accumulators (fewer registers, more waves per CU)
template <bool TWO, bool SGN = false>
__global__ __launch_bounds__(256)
void k_slice_compare(const u64* const* __restrict__ descs /* per plane: descriptor table or null */,
                     const u32* __restrict__ nblk, u32 nplanes, u32 ncols, int pred, u64 v0, u64 v1, u64 size,
                     const u64* __restrict__ nn_desc /* not-NULL vector or null */, u32 nn_blocks, int null_correct,
                     int count_only, int xcd_swz,
                     uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, u64* __restrict__ slots,
                     const u64* __restrict__ sign_desc, u32 sign_blocks, int sign_mode, u64* __restrict__ stat)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 nb = uniform32(bid * 4u + wave);
    u32 cnt = 0;
    if (nb < ncols) {
        constexpr bool two = TWO;
        u32 read_bytes = 0;
        // a bound with a set bit above every plane: nothing is equal to it and nothing is greater
        bool dead0 = nplanes < 64u && (v0 >> nplanes) != 0ull;
        bool dead1 = two && nplanes < 64u && (v1 >> nplanes) != 0ull;
        Blk gt0, eq0, gt1, eq1;
        blk_fill(gt0, 0u); blk_fill(eq0, dead0 ? 0u : ~0u);
        blk_fill(gt1, 0u); blk_fill(eq1, (two && !dead1) ? ~0u : 0u);
        bool live = !dead0 || (two && !dead1);
        u32* l = lds + wave * 2048u;
        for (u32 b = nplanes; b-- > 0u && live; ) {
            const u64* dt = (const u64*)uniform64((u64)(uintptr_t)descs[b]);
            u64 d = (dt && nb < uniform32(nblk[b])) ? uniform64(dt[nb]) : 0ull;
            u32 bit0 = (u32)(v0 >> b) & 1u, bit1 = (u32)(v1 >> b) & 1u;
            if (DESC_K(d) == K_NULL) {                                   // absent plane / NULL block: P = 0
                if (bit0) blk_fill(eq0, 0u);
                if (two && bit1) blk_fill(eq1, 0u);
            } else {
                Blk P;
                blk_from_desc(d, P, l, lane);
                read_bytes += DESC_K(d) == K_BIT ? 8192u : (DESC_K(d) == K_GAP ? 2u * ((GMETA(d) >> 1) + 1u) : 0u);
                if (bit0) blk_and(eq0, P);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { gt0.r[i] |= eq0.r[i] & P.r[i]; eq0.r[i] &= ~P.r[i]; }
                }
                if (two) {
                    if (bit1) blk_and(eq1, P);
                    else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { gt1.r[i] |= eq1.r[i] & P.r[i]; eq1.r[i] &= ~P.r[i]; }
                    }
                }
            }
            u32 any = blk_lane_or(eq0) | (two ? blk_lane_or(eq1) : 0u);
            live = __ballot(any != 0u) != 0ull;
        }
        Blk res;
        switch (pred) {
        case CMP_GT: res = gt0; break;
        case CMP_GE: res = gt0; blk_or(res, eq0); break;
        case CMP_LT: res = gt0; blk_or(res, eq0);
#pragma unroll
            for (int i = 0; i < 8; ++i) res.r[i] = ~res.r[i];
            break;
        case CMP_LE:
#pragma unroll
            for (int i = 0; i < 8; ++i) res.r[i] = ~gt0.r[i];
            break;
        case CMP_RANGE: res = gt0; blk_or(res, eq0); blk_andn(res, gt1); break;
        case CMP_EQ: case CMP_ZERO: res = eq0; break;
        case CMP_SRANGE: break;                                          // (combined with the sign block below)
        default:                                                         // NONZERO
#pragma unroll
            for (int i = 0; i < 8; ++i) res.r[i] = ~eq0.r[i];
            break;
        }
        Blk m;
        if (SGN && (sign_mode != SIGN_NONE || pred == CMP_SRANGE)) {
            u64 sd = (sign_desc && nb < sign_blocks) ? uniform64(sign_desc[nb]) : 0ull;
            blk_from_desc(sd, m, l, lane);                                // the sign block S
            read_bytes += DESC_K(sd) == K_BIT ? 8192u : (DESC_K(sd) == K_GAP ? 2u * ((GMETA(sd) >> 1) + 1u) : 0u);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                res.r[i] = pred == CMP_SRANGE ? ((~gt0.r[i] & ~m.r[i]) | (~gt1.r[i] & m.r[i])) : sign_combine(res.r[i], m.r[i], sign_mode);
        }
        if (stat && lane == 0 && read_bytes) atomicAdd(reinterpret_cast<unsigned long long*>(stat), (unsigned long long)read_bytes);
        blk_size_mask(m, nb, size, lane);
        blk_and(res, m);
        if (null_correct && nn_desc) {
            u64 d = nb < nn_blocks ? uniform64(nn_desc[nb]) : 0ull;
            blk_from_desc(d, m, l, lane);
            blk_and(res, m);
        }
        bool zero = blk_is_zero(res);
        if (count_only) { if (!zero) cnt = wave_sum(blk_lane_popcount(res)); }
        else if (zero) store_trivial(K_NULL, nb, desc, st, lane);
        else store_result(res, nb, 1, slab, desc, st, lane);
    }
    if (count_only) count_fanin(cnt, slots, lane, wave);
}

// Comparison search in HALF-BLOCK passes (find_range first, then every predicate): the four accumulators of k_slice_compare<true> cost 282 VGPRs = one wave per
// SIMD, and occupancy is what this kernel lives on (the one-bound form went from 0.61 to 0.39 ms with two waves).  A wave
// walks the planes of its column twice, over register rows 0..3 and then 4..7 of every block: half the accumulators, the
// same bytes read (two 4-KiB pieces per plane block instead of one 8-KiB piece), the "nobody is equal any more" exit per half.
// Same result: the predicate is evaluated independently per row.
template <int RP>
__device__ __forceinline__ void part_from_desc(u64 d, Part<RP>& P, u32 h, u32* lds, u32 lane)
{
    u32 k = DESC_K(d);
    if (k == K_BIT) part_load<RP, false>(P, as_gc4(DESC_P(d)) + h * (u32)RP * 64u, lane);
    else if (k == K_GAP) {
        Blk t;
        gap_decode(as_gc16(DESC_P(d)), lds, t, lane, GMETA(d));
#pragma unroll
        for (int i = 0; i < RP; ++i) {
#pragma unroll
            for (int q = 0; q < 8 / RP; ++q) if ((u32)q == h) P.r[i] = t.r[q * RP + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < RP; ++i) P.r[i] = (u32x4)(k == K_FULL ? ~0u : 0u);
    }
}

template <bool TWO, int RP, bool SGN = false>
__global__ __launch_bounds__(256)
void k_slice_compare_halves(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 nplanes, u32 ncols, int pred, u64 v0, u64 v1, u64 size,
                          const u64* __restrict__ nn_desc, u32 nn_blocks, int null_correct, int count_only, int xcd_swz,
                          uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, u64* __restrict__ slots,
                          const u64* __restrict__ sign_desc, u32 sign_blocks, int sign_mode, u64* __restrict__ stat)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 nb = uniform32(bid * 4u + wave);
    u32 cnt = 0;
    if (nb < ncols) {
        const bool dead0 = nplanes < 64u && (v0 >> nplanes) != 0ull;       // a bound with a bit above every plane: nothing equals or exceeds it
        const bool dead1 = TWO && nplanes < 64u && (v1 >> nplanes) != 0ull;
        u32* l = lds + wave * 2048u;
        const u64 base = (u64)nb << 16;
        const u32 lim = size <= base ? 0u : (size - base >= 65536ull ? 65536u : (u32)(size - base));
        Blk out;
        u32 read_bytes = 0;
        const bool use_sign = SGN && (sign_mode != SIGN_NONE || pred == CMP_SRANGE);
        const u64 sd = (use_sign && sign_desc && nb < sign_blocks) ? uniform64(sign_desc[nb]) : 0ull;
#pragma unroll 1
        for (u32 h = 0; h < (u32)(8 / RP); ++h) {
            Part<RP> gt0, eq0, gt1, eq1;
#pragma unroll
            for (int i = 0; i < RP; ++i) { gt0.r[i] = (u32x4)(0u); gt1.r[i] = (u32x4)(0u); eq0.r[i] = (u32x4)(dead0 ? 0u : ~0u); eq1.r[i] = (u32x4)((TWO && !dead1) ? ~0u : 0u); }
            bool live = !dead0 || (TWO && !dead1);
            for (u32 b = nplanes; b-- > 0u && live; ) {
                const u64* dt = (const u64*)uniform64((u64)(uintptr_t)descs[b]);
                u64 d = (dt && nb < uniform32(nblk[b])) ? uniform64(dt[nb]) : 0ull;
                u32 bit0 = (u32)(v0 >> b) & 1u, bit1 = (u32)(v1 >> b) & 1u;
                if (DESC_K(d) == K_NULL) {
#pragma unroll
                    for (int i = 0; i < RP; ++i) { if (bit0) eq0.r[i] = (u32x4)(0u); if (TWO && bit1) eq1.r[i] = (u32x4)(0u); }
                } else {
                    Part<RP> P;
                    part_from_desc<RP>(d, P, h, l, lane);
                    // (algorithmic bytes: a bit-block piece is RP KiB; a GAP block -- decoded whole by every pass that needs it --
                    // is charged its share of the passes, so that a column never counts more than its planes hold)
                    read_bytes += DESC_K(d) == K_BIT ? (u32)RP * 1024u : (DESC_K(d) == K_GAP ? (2u * ((GMETA(d) >> 1) + 1u) * (u32)RP) / 8u : 0u);
#pragma unroll
                    for (int i = 0; i < RP; ++i) {
                        if (bit0) eq0.r[i] &= P.r[i]; else { gt0.r[i] |= eq0.r[i] & P.r[i]; eq0.r[i] &= ~P.r[i]; }
                        if (TWO) { if (bit1) eq1.r[i] &= P.r[i]; else { gt1.r[i] |= eq1.r[i] & P.r[i]; eq1.r[i] &= ~P.r[i]; } }
                    }
                }
                u32 any = 0u;
#pragma unroll
                for (int i = 0; i < RP; ++i) { u32x4 t = TWO ? (eq0.r[i] | eq1.r[i]) : eq0.r[i]; any |= t.x | t.y | t.z | t.w; }
                live = __ballot(any != 0u) != 0ull;
            }
            Part<RP> res;                                                   // the predicate, inside [0, size), not NULL where 0 is admitted
            Part<RP> nn;
            bool use_nn = null_correct && nn_desc;
            if (use_nn) part_from_desc<RP>(nb < nn_blocks ? uniform64(nn_desc[nb]) : 0ull, nn, h, l, lane);
            Part<SGN ? RP : 1> S;
            if constexpr (SGN) if (use_sign) {
                part_from_desc<RP>(sd, S, h, l, lane);
                read_bytes += DESC_K(sd) == K_BIT ? (u32)RP * 1024u : (DESC_K(sd) == K_GAP ? (2u * ((GMETA(sd) >> 1) + 1u) * (u32)RP) / 8u : 0u);
            }
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                u32x4 r;
                switch (pred) {
                case CMP_GT: r = gt0.r[i]; break;
                case CMP_GE: r = gt0.r[i] | eq0.r[i]; break;
                case CMP_LT: r = ~(gt0.r[i] | eq0.r[i]); break;
                case CMP_LE: r = ~gt0.r[i]; break;
                case CMP_RANGE: r = (gt0.r[i] | eq0.r[i]) & ~gt1.r[i]; break;
                case CMP_EQ: case CMP_ZERO: r = eq0.r[i]; break;
                case CMP_SRANGE: if constexpr (SGN) r = (~gt0.r[i] & ~S.r[i]) | (~gt1.r[i] & S.r[i]); else r = (u32x4)(0u); break;
                default: r = ~eq0.r[i]; break;                              // NONZERO
                }
                if constexpr (SGN) if (use_sign && pred != CMP_SRANGE) r = sign_combine(r, S.r[i], sign_mode);
                u32 w0 = (((h * (u32)RP + (u32)i) * 256u + lane * 4u) << 5);    // first row of word .x
                u32 ws[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { u32 lo = w0 + (u32)j * 32u; ws[j] = lim >= lo + 32u ? ~0u : (lim <= lo ? 0u : ((1u << (lim - lo)) - 1u)); }
                r.x &= ws[0]; r.y &= ws[1]; r.z &= ws[2]; r.w &= ws[3];
                if (use_nn) r &= nn.r[i];
                res.r[i] = r;
            }
            if (count_only) {
                u32 c = 0;
#pragma unroll
                for (int i = 0; i < RP; ++i) { c += __popcll(((u64)res.r[i].y << 32) | res.r[i].x); c += __popcll(((u64)res.r[i].w << 32) | res.r[i].z); }
                cnt += c;
            } else {
#pragma unroll
                for (int i = 0; i < RP; ++i) {
#pragma unroll
                    for (int q = 0; q < 8 / RP; ++q) if ((u32)q == h) out.r[q * RP + i] = res.r[i];
                }
            }
        }
        if (stat && lane == 0 && read_bytes) atomicAdd(reinterpret_cast<unsigned long long*>(stat), (unsigned long long)read_bytes);
        if (count_only) cnt = wave_sum(cnt);
        else if (blk_is_zero(out)) store_trivial(K_NULL, nb, desc, st, lane);
        else store_result(out, nb, 1, slab, desc, st, lane);
    }
    if (count_only) count_fanin(cnt, slots, lane, wave);
}

// ---------------------------------------------------------------------------
// Batched equality counts over bit-planes: counts[q] = number of rows whose value equals values[q].
// The reference answers every query with its own AND-SUB group over the planes (prepare_and_sub_aggregator,
// src/bmsparsevec_algo.h:2593-2640, run as a pipeline :3236,3408): work and plane traffic grow with the number of queries
// (k_pipe_counts_staged: 25 ms for 512 queries over 32 planes x 1e9 rows).  The planes are a bit-matrix: TRANSPOSE it and
// every row's value is in a register -- 32 plane words of the same 32 rows are one 32 x 32 bit tile, five butterfly stages
// turn it into 32 values -- and look each value up in a hash table of the queried values (LDS, open addressing).  Every
// plane block is read ONCE whatever the number of queries; the result is the same multiset count the groups compute
// (value 0 is not handled here: NULL elements are stored as 0, the host routes it through k_slice_compare).
// A wave owns a block column at a time (persistent grid); GAP planes are expanded to raw bits beforehand (raw[p]).
// Register allocation decides the speed here (245 VGPRs = two waves per SIMD; PMC: VALU busy 55 %, 43 % of the wave
// cycles waiting, LDS ~20 %): 8-byte loads with 512 B per wave load made the compiler keep a second set of plane words in
// flight (304 VGPRs, or 196 B of spills when capped) and measured 2.5-2.7 ms against 1.65 ms; capping at three waves per
// SIMD (amdgpu_waves_per_eu: 284 B of spills) measured 5.0 ms.  4-byte loads (256 B per wave load, this form) and 8-byte
// loads at a 16-byte stride run at the same 1.65 ms.
struct EqPlanes { const u64* desc[32]; const uint4* raw[32]; u32 nblk[32]; };

__device__ __forceinline__ void bit_transpose32(u32 (&a)[32])
{
    // out[r] bit p = in[p] bit r (LSB numbering); Hacker's Delight 7-3 with the shifts mirrored.  The two coarse stages move
    // whole bytes: one v_perm_b32 per output word instead of shift / xor / and / xor / shift / xor
#pragma unroll
    for (int k = 0; k < 16; ++k) {                                  // j = 16: halves
        u32 x = a[k], y = a[k + 16];
        a[k] = __builtin_amdgcn_perm(y, x, 0x05040100u);
        a[k + 16] = __builtin_amdgcn_perm(y, x, 0x07060302u);
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {                                  // j = 8: bytes
        if (k & 8) continue;
        u32 x = a[k], y = a[k + 8];
        a[k] = __builtin_amdgcn_perm(y, x, 0x06020400u);
        a[k + 8] = __builtin_amdgcn_perm(y, x, 0x07030501u);
    }
#pragma unroll
    for (int s = 2; s < 5; ++s) {
        const int j = 16 >> s;
        const u32 m = s == 2 ? 0x0F0F0F0Fu : s == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k & j) continue;
            u32 t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k + j] ^= t;
            a[k] ^= t << j;
        }
    }
}

// Lookup: a 64 Kbit presence filter (one LDS read per row, no loop: 2,048 queried values leave ~3 % of the rows), the
// survivors of a 2,048-row step are compacted into a per-wave LDS queue with ballots (no atomics, no round trips) and then
// probed against the exact table 64 at a time with every lane busy.  (The first form probed the open-addressing table
// straight from the row loop: 32 divergent probe loops per step, each an LDS round trip with a few lanes alive -- 15.9 ms
// per pass over 32 planes x 1e9 rows instead of ~1 ms.)
#define EQ_FILTER_WORDS 2048u
#define EQ_QUEUE 2048u                  // one slot per row of a step: the queue cannot overflow, whatever share of the rows is looked for
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x2* gcptr2;

// NP = 16: containers of up to 16 planes -- planes 16..31 are compile-time zeros and the butterfly folds away over them
template <int NP>
__global__ __launch_bounds__(256)
void k_slice_eq_counts(EqPlanes pl, u32 nplanes, u32 ncols, u64 size, const u32* __restrict__ g_keys, const u16* __restrict__ g_idx,
                       u32 tab_size /* power of two */, u32 shift, u32 nvals, u64* __restrict__ counts)
{
    extern __shared__ u32 lds_dyn[];
    u32* keys = lds_dyn;                                            // tab_size keys (0 = empty)
    u32* cnt = keys + tab_size;                                     // nvals counters
    u32* filt = cnt + nvals;                                        // presence filter, 64 Kbit
    u32* queue = filt + EQ_FILTER_WORDS;                            // 4 waves x EQ_QUEUE survivors
    u16* idx = reinterpret_cast<u16*>(queue + 4u * EQ_QUEUE);       // tab_size slots -> query ordinal
    for (u32 i = threadIdx.x; i < EQ_FILTER_WORDS; i += blockDim.x) filt[i] = 0u;
    for (u32 i = threadIdx.x; i < nvals; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    for (u32 i = threadIdx.x; i < tab_size; i += blockDim.x) {
        u32 k = g_keys[i];
        keys[i] = k; idx[i] = g_idx[i];
        if (k) { u32 hb = (k * 0x85EBCA6Bu) >> 16; atomicOr(&filt[hb >> 5], 1u << (hb & 31u)); }
    }
    __syncthreads();
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    const u32 tmask = tab_size - 1u;
    u32* myq = queue + wave * EQ_QUEUE;
    for (u32 c = uniform32(blockIdx.x * 4u + wave); c < ncols; c += gridDim.x * 4u) {
        u64 base[32];                                               // per plane: 0 = all zero, 1 = all ones, else the block
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            u64 b = 0ull;
            if (p < NP && (u32)p < nplanes) {
                if (pl.raw[p]) b = (u64)(uintptr_t)(pl.raw[p] + (size_t)c * 512u);
                else if (pl.desc[p] && c < pl.nblk[p]) {
                    u64 d = uniform64(pl.desc[p][c]);
                    b = DESC_K(d) == K_BIT ? DESC_P(d) : (DESC_K(d) == K_FULL ? 1ull : 0ull);
                }
            }
            base[p] = uniform64(b);                                // (wave-uniform: keep it in scalar registers)
        }
        const u64 row0 = (u64)c << 16;
        const u32 lim = size <= row0 ? 0u : (size - row0 >= 65536ull ? 65536u : (u32)(size - row0));
#pragma unroll 1
        for (u32 k = 0; k < 32u; ++k) {                             // step k: word k * 64 + lane of every plane block (256 B per wave load)
            {
                u32 a[32];
                u32 any = 0u;
#pragma unroll
                for (int p = 0; p < 32; ++p) {
                    if (p >= NP) { a[p] = 0u; continue; }
                    if (base[p] > 1ull) a[p] = __builtin_nontemporal_load((const __attribute__((address_space(1))) u32*)(uintptr_t)base[p] + k * 64u + lane);
                    else a[p] = base[p] ? ~0u : 0u;
                    any |= a[p];
                }
                const u32 wb = ((k * 64u + lane) << 5);                               // first row of this word inside the block
                const u32 vm = lim >= wb + 32u ? ~0u : (lim <= wb ? 0u : ((1u << (lim - wb)) - 1u));
                if (__ballot((any & vm) != 0u) == 0ull) continue;          // 2,048 rows of zeros: nothing to look up
                bit_transpose32(a);
                u32 f[32];
#pragma unroll
                for (int r = 0; r < 32; ++r) { u32 hb = (a[r] * 0x85EBCA6Bu) >> 16; f[r] = filt[hb >> 5] >> (hb & 31u); }
                u32 nq = 0;                                                 // survivors of this step (wave-uniform)
#pragma unroll
                for (int r = 0; r < 32; ++r) {
                    bool hit = (f[r] & 1u) && a[r] != 0u && ((vm >> r) & 1u);
                    u64 m = __ballot(hit);
                    if (m) {
                        u32 pos = nq + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                        if (hit) myq[pos] = a[r];
                        nq += (u32)__popcll(m);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (u32 b = 0; b < nq; b += 64u) {
                    if (b + lane < nq) {
                        u32 v = myq[b + lane];
                        u32 h = (v * 0x9E3779B1u) >> shift;
                        for (;;) { u32 kk = keys[h]; if (kk == v) { atomicAdd(&cnt[idx[h]], 1u); break; } if (kk == 0u) break; h = (h + 1u) & tmask; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nvals; i += blockDim.x)
        if (cnt[i]) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[i]), (unsigned long long)cnt[i]);
}

// The same pass for MORE queried values than the 2,048-value table above takes (round 3; the host used to re-read the planes
// once per 2,048 values: 8,192 queries = 4 passes).  What bounds the batch is LDS, so the table is made leaner instead of
// the planes being re-read: a slot is {key, count} -- the count sits WITH the key (no ordinal array; the host, which built
// the table, maps slots back to queries), any table size (multiplicative range reduction instead of a power-of-two mask)
// at 1.5 slots per value, 512 threads sharing one table, per-wave queues of 1,024 entries flushed inside a step when more
// than half full (a step can produce 2,048 survivors), and a 128 Kbit filter so that ~9,000 values still leave few false
// probes.  LDS = 12 B x slots + 16 KiB filter + 32 KiB queues: up to 9,216 values in ONE pass over the planes.
#define EQB_MAX_VALUES 9216u
#define EQB_SLOTS(nv) ((((nv) * 3u / 2u) + 63u) & ~63u)

// FBITS = log2 of the filter bits (17: 16 KiB, 18: 32 KiB), QSZ = queue entries per wave (flushed when more than half full;
// checked every QSZ / 128 rows of a step, which add at most QSZ / 2 survivors)
// WG = threads per workgroup (all waves share the table), WPE = waves per SIMD the registers are held to, FG = filter reads
// in flight per lane (32: all rows of a step at once, 243 VGPRs = 2 waves per SIMD; 8: 168 VGPRs = 3 waves per SIMD)
template <int NP, int FBITS, int QSZ, int WG = 512, int WPE = 2, int FG = 32>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WPE)))
void k_slice_eq_counts_big(EqPlanes pl, u32 nplanes, u32 ncols, u64 size, const u32* __restrict__ g_keys, u32 tab, u64* __restrict__ counts /* per slot */)
{
    extern __shared__ u32 lds_dyn[];
    u32* keys = lds_dyn;                                            // tab keys (0 = empty)
    u32* cnt = keys + tab;                                          // tab counters
    u32* filt = cnt + tab;                                          // presence filter, 2^FBITS bits
    constexpr u32 FW = 1u << (FBITS - 5);
    u32* queue = filt + FW;                                         // 8 waves x QSZ survivors
    for (u32 i = threadIdx.x; i < FW; i += blockDim.x) filt[i] = 0u;
    for (u32 i = threadIdx.x; i < tab; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    for (u32 i = threadIdx.x; i < tab; i += blockDim.x) {
        u32 k = g_keys[i];
        keys[i] = k;
        if (k) { u32 hb = (k * 0x85EBCA6Bu) >> (32 - FBITS); atomicOr(&filt[hb >> 5], 1u << (hb & 31u)); }
    }
    __syncthreads();
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32* myq = queue + wave * (u32)QSZ;
    constexpr u32 NW = (u32)WG / 64u;
    for (u32 c = uniform32(blockIdx.x * NW + wave); c < ncols; c += gridDim.x * NW) {
        u64 base[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            u64 b = 0ull;
            if (p < NP && (u32)p < nplanes) {
                if (pl.raw[p]) b = (u64)(uintptr_t)(pl.raw[p] + (size_t)c * 512u);
                else if (pl.desc[p] && c < pl.nblk[p]) {
                    u64 d = uniform64(pl.desc[p][c]);
                    b = DESC_K(d) == K_BIT ? DESC_P(d) : (DESC_K(d) == K_FULL ? 1ull : 0ull);
                }
            }
            base[p] = uniform64(b);
        }
        const u64 row0 = (u64)c << 16;
        const u32 lim = size <= row0 ? 0u : (size - row0 >= 65536ull ? 65536u : (u32)(size - row0));
#pragma unroll 1
        for (u32 k = 0; k < 32u; ++k) {
            u32 a[32];
            u32 any = 0u;
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                if (p >= NP) { a[p] = 0u; continue; }
                if (base[p] > 1ull) a[p] = __builtin_nontemporal_load((const __attribute__((address_space(1))) u32*)(uintptr_t)base[p] + k * 64u + lane);
                else a[p] = base[p] ? ~0u : 0u;
                any |= a[p];
            }
            const u32 wb = ((k * 64u + lane) << 5);
            const u32 vm = lim >= wb + 32u ? ~0u : (lim <= wb ? 0u : ((1u << (lim - wb)) - 1u));
            if (__ballot((any & vm) != 0u) == 0ull) continue;
            bit_transpose32(a);
            u32 nq = 0;
            auto flush = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (u32 b = 0; b < nq; b += 64u) {
                    if (b + lane < nq) {
                        u32 v = myq[b + lane];
                        u32 h = __umulhi(v * 0x9E3779B1u, tab);
                        for (;;) { u32 kk = keys[h]; if (kk == v) { atomicAdd(&cnt[h], 1u); break; } if (kk == 0u) break; h = h + 1u == tab ? 0u : h + 1u; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                nq = 0;
            };
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += FG) {
                u32 f[FG];
#pragma unroll
                for (int r = 0; r < FG; ++r) { u32 hb = (a[r0 + r] * 0x85EBCA6Bu) >> (32 - FBITS); f[r] = filt[hb >> 5] >> (hb & 31u); }
#pragma unroll
                for (int rr = 0; rr < FG; ++rr) {
                    const int r = r0 + rr;
                    bool hit = (f[rr] & 1u) && a[r] != 0u && ((vm >> r) & 1u);
                    u64 m = __ballot(hit);
                    if (m) {
                        u32 pos = nq + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                        if (hit) myq[pos] = a[r];
                        nq += (u32)__popcll(m);
                    }
                    if ((r & (QSZ / 128 - 1)) == QSZ / 128 - 1 && r != 31 && nq > (u32)QSZ / 2u) flush();
                }
            }
            if (nq) flush();
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < tab; i += blockDim.x)
        if (cnt[i]) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[i]), (unsigned long long)cnt[i]);
}
