// bmx_kernels6.h -- column-major packed GAP collections (round 3).
#pragma once
#include "bmx_kernels5.h"

// ---------------------------------------------------------------------------
// Why.  combine_or / combine_and[_sub] over MANY GAP-only operands (BASELINE configs[4]: 4096 sparse vectors; the all-GAP
// 256-way AND) read every operand's block of a column from that operand's own slab: thousands of streams visited in
// pieces of ~60 B .. 1.5 KiB.  Round 2 measured the floor of that access pattern with the run application compiled out
// (3.78 of 4.14 ms): the layout, not the arithmetic, is the bound.  Vectors are immutable and
// device-resident, so the library owns the layout: the GAP blocks of an operand SET are transposed ONCE into
// column-major order and every later aggregation over that set streams one contiguous region per block column.
//
// What is stored.  A block column of the collection is a flat bag of INTERVALS, one 32-bit entry per run:
// start | end << 16 (both inclusive, 0..65535).  Which runs depends on the role of the set:
//   polarity 1 (OR list, SUB list): the 1-runs of every GAP operand.  OR = union of the bag.
//   polarity 0 (AND list):          the 0-runs.  AND_i x_i = NOT OR_i NOT x_i: the AND of the operands is the complement of
//                                   the union of their 0-runs (the reference clears the 0-runs one operand at a time:
//                                   gap_and_to_bitset, src/bmfunc.h:4847; process_gap_blocks_and, src/bmaggregator.h:1820).
// Operand identity is gone -- neither union needs it -- so a column is one sequential stream whatever the operand count,
// and a NULL / FULL operand of a column is a per-column flag (any NULL ends an AND column, src/bmaggregator.h:2327;
// any FULL saturates an OR column, :2300; FULL operands of an AND list are ignored, :2346).
// The same information as the GAP blocks themselves (Appendix B: run k covers e[k-1]+1 .. e[k]): 4 B per run of the wanted
// polarity ~= 2 x len bytes per block, i.e. the algorithmic bytes of SURVEY section 8(d) minus headers.
//
// How a bag is applied (coll_apply_run): every entry is independent.  A run inside one 32-bit word is ONE ds_or.  A longer
// run ORs its two edge words and covers the words in between through a word-granular counting trick: +1 at the first
// interior word, -1 behind the last one in a 2048-entry array D; one prefix sum over D at the end of the column marks the
// covered words (coll_fold).  At most four LDS atomics per run whatever its length, no per-operand loop, no decode.
// LDS per workgroup: bitmap 8 KiB + D 8 KiB (+ 8 KiB for the SUB bag): several workgroups per CU overlap their phases.
// ---------------------------------------------------------------------------

#define COLL_FLAG_FULL 1u       // some operand of the column is a FULL block
#define COLL_FLAG_NULL 2u       // some operand of the column is NULL (or shorter than the column index)
#define COLL_FLAG_BIT  4u       // some operand holds a bit-block here: the packed path cannot be used (host falls back)
#define COLL_NGAP(f) ((f) >> 8) // GAP operands of the column

typedef const __attribute__((address_space(1))) u64* coll_gc64;
typedef const __attribute__((address_space(1))) u32* coll_gc32;

// runs of the wanted polarity in a GAP block with GMETA m = len << 1 | start bit
__device__ __forceinline__ u32 coll_runs_of(u32 meta, u32 polarity)
{
    u32 len = meta >> 1, s = meta & 1u;
    u32 ones = s ? (len + 1u) >> 1 : len >> 1;
    return polarity ? ones : len - ones;
}

// pass 1: one thread per block column walks the operand list (lanes = consecutive columns: every read of an operand's
// descriptor table is coalesced): pre[i][c] = entries of column c that precede operand i (< 2^30) | operand i's block kind << 30, cnt[c] = entries of the column,
// flags[c] = FULL / NULL / bit-block marks | GAP operand count << 8
__global__ __launch_bounds__(256)
void k_coll_count(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 n, u32 ncols, u32 polarity,
                  u32* __restrict__ pre, u32* __restrict__ cnt, u32* __restrict__ flags)
{
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    u32 run = 0, fl = 0, ngap = 0;
    for (u32 i = 0; i < n; ++i) {
        u32 nb = nblk[i];
        u64 d = c < nb ? descs[i][c] : 0ull;                    // beyond the operand's end: NULL
        u32 k = DESC_K(d);
        pre[(size_t)i * ncols + c] = run | (k << 30);             // (the operand's block kind rides in the two top bits: member directory, bmx_kernels8.h)
        if (k == K_GAP) { run += coll_runs_of(GMETA(d), polarity); ++ngap; }
        else if (k == K_FULL) fl |= COLL_FLAG_FULL;
        else if (k == K_NULL) fl |= COLL_FLAG_NULL;
        else fl |= COLL_FLAG_BIT;
    }
    cnt[c] = run;
    flags[c] = fl | (ngap << 8);
}

// pass 2: column offsets (in entries; every column starts on a 16-byte boundary): exclusive scan of cnt rounded up to 4.
// Single workgroup; off[ncols] = total.
__global__ __launch_bounds__(1024)
void k_coll_offsets(const u32* __restrict__ cnt, u32 ncols, u64* __restrict__ off)
{
    __shared__ u64 sm[16];
    u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u64 carry = 0;
    for (u32 base = 0; base < ncols; base += 1024u) {
        u32 c = base + tid;
        u32 v = c < ncols ? (cnt[c] + 3u) & ~3u : 0u;
        u32 incl = wave_scan_incl(v, lane);
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        u64 offw = 0, tot = 0;
#pragma unroll
        for (u32 i = 0; i < 16; ++i) { u64 a = sm[i]; if (i < w) offw += a; tot += a; }
        __syncthreads();
        if (c < ncols) off[c] = carry + offw + (u64)(incl - v);
        carry += tot;
    }
    if (tid == 0) off[ncols] = carry;
}

// COLL_YT column tiles per workgroup in the passes that walk (operand, column) blocks: a workgroup per tile was 7.8 M workgroups of
// 32 blocks each for configs[4] -- bound by the dispatch rate, not by memory
#define COLL_YT 8u

// The wanted runs of one 16-byte chunk of a GAP block, from registers (round 4: the build read every run end with a 2-byte load).
// Chunk q holds words 8q .. 8q + 7 of the block (word 0 = header, word k = end of run k); nx = the first dword of chunk q + 1.
// (s = 1: the block starts with a run of the wanted polarity.)  A block that starts with a 0-run (s = 0) has its 1-runs at even k: run m = (words 2m + 1, 2m + 2) -- the chunk's pairs
// (1,2) (3,4) (5,6) (7, next 0); a block that starts with a 1-run (s = 1) at odd k: run m = (words 2m, 2m + 1) -- pairs
// (0,1) (2,3) (4,5) (6,7), run 0 starting at bit 0.  Slot i of chunk q is run m = 4q + i, valid while m < m_cnt.
struct CollPairs { u32 start[4], end[4]; };
__device__ __forceinline__ void coll_chunk_pairs(const u32x4& x, u32 nx, u32 s, u32 q, CollPairs& p)
{
    const u32 w[9] = {x.x & 0xFFFFu, x.x >> 16, x.y & 0xFFFFu, x.y >> 16, x.z & 0xFFFFu, x.z >> 16, x.w & 0xFFFFu, x.w >> 16, nx & 0xFFFFu};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 lo = s ? w[2 * i] : w[2 * i + 1], hi = s ? w[2 * i + 1] : w[2 * i + 2];
        p.start[i] = (s && q == 0u && i == 0) ? 0u : lo + 1u;
        p.end[i] = hi;
    }
}

// pass 3: L lanes per (operand, column) write the block's runs of the wanted polarity behind the entries of the operands
// before it.  grid.x = operand (adjacent workgroups fill adjacent pieces of the same columns), grid.y = tile of 256 / L columns
template <int L>
__global__ __launch_bounds__(256)
void k_coll_scatter(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 ncols, u32 polarity,
                    const u32* __restrict__ pre, const u64* __restrict__ off, u32* __restrict__ runs)
{
    const u32 i = blockIdx.x;
    const u32 t = threadIdx.x % L;
    const u32 nb_i = nblk[i];
    const u64* __restrict__ di = descs[i];
    auto dload = [&](u32 by) -> u64 { const u32 c = by * (256u / L) + threadIdx.x / L; return (c < ncols && c < nb_i) ? di[c] : 0ull; };
    u64 dn = dload(blockIdx.y * COLL_YT);                 // the next tile's descriptor is requested a tile ahead
    for (u32 by = blockIdx.y * COLL_YT; by < (blockIdx.y + 1u) * COLL_YT; ++by) {
        const u32 c = by * (256u / L) + threadIdx.x / L;
        if (by * (256u / L) >= ncols) break;
        const u64 d = dn;
        dn = dload(by + 1u);
        if (DESC_K(d) != K_GAP) continue;
        u32 meta = GMETA(d), len = meta >> 1, s = meta & 1u;
        u32 m_cnt = coll_runs_of(meta, polarity);
        u32* out = runs + off[c] + (pre[(size_t)i * ncols + c] & 0x3FFFFFFFu);
        // run k (1-based) has the value s ^ ((k - 1) & 1) and covers e[k-1]+1 .. e[k] (e[0] = -1; word k of the block = e[k]): the
        // wanted runs are every second one from k = 1 (the block starts with a wanted run) or k = 2; lane t takes chunks t, t + L, ...
        // as 16-byte loads (coll_chunk_pairs), run m lands in out[m]
        const u32 s_eff = s == polarity ? 1u : 0u;
        const u32 nchunks = (len + 1u + 7u) >> 3;
        gcptr4 g4 = as_gc4(DESC_P(d));
        for (u32 q = t; q < nchunks; q += (u32)L) {
            const u32x4 x = g4[q];
            const u32 nx = q + 1u < nchunks ? g4[q + 1u].x : 0xFFFFFFFFu;
            CollPairs p;
            coll_chunk_pairs(x, nx, s_eff, q, p);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (4u * q + (u32)k < m_cnt) out[4u * q + (u32)k] = p.start[k] | (p.end[k] << 16);
        }
    }
}

// ---- split bags (round 3): single-bit runs as 16-bit positions ----
// A bag of 1-runs of SPARSE operands is almost all single bits, and a single bit does not need a start and an end: the
// polarity-1 collection keeps, per column, the runs longer than one bit as before (32-bit start | end << 16, padded to 4)
// and behind them the single-bit runs as 16-bit positions (padded to 8): half the bytes per run of the reference's own GAP
// encoding (two 16-bit run ends per isolated bit).  The bag is a set, so nothing else changes: same union, same result.
//   cnt[c]   = 1-runs of the column (as before), cnt_s[c] = the single-bit ones among them
//   off[c]   = first 32-bit word of the column; multis at [off, off + round4(cnt - cnt_s)), singles behind them

// count pass: L lanes per (operand, column) count the single-bit 1-runs of the block -> sgl[i][c] (raw count); lane t takes
// chunks t, t + L, ... of the block as 16-byte loads
template <int L>
__global__ __launch_bounds__(256)
void k_coll_count_singles(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 ncols, u32* __restrict__ sgl)
{
    const u32 i = blockIdx.x;
    const u32 t = threadIdx.x % L;
    const u32 nb_i = nblk[i];
    const u64* __restrict__ di = descs[i];
    auto dload = [&](u32 by) -> u64 { const u32 c = by * (256u / L) + threadIdx.x / L; return (c < ncols && c < nb_i) ? di[c] : 0ull; };
    u64 dn = dload(blockIdx.y * COLL_YT);                 // the next tile's descriptor is requested a tile ahead
    for (u32 by = blockIdx.y * COLL_YT; by < (blockIdx.y + 1u) * COLL_YT; ++by) {
        const u32 c = by * (256u / L) + threadIdx.x / L;
        if (by * (256u / L) >= ncols) break;
        const bool in = c < ncols && c < nb_i;
        const u64 d = dn;
        dn = dload(by + 1u);
        u32 ns = 0;
        if (DESC_K(d) == K_GAP) {
            const u32 meta = GMETA(d), len = meta >> 1, s = meta & 1u;
            const u32 m_cnt = coll_runs_of(meta, 1u);
            const u32 nchunks = (len + 1u + 7u) >> 3;
            gcptr4 g4 = as_gc4(DESC_P(d));
            for (u32 q = t; q < nchunks; q += (u32)L) {
                const u32x4 x = g4[q];
                const u32 nx = q + 1u < nchunks ? g4[q + 1u].x : 0xFFFFFFFFu;
                CollPairs p;
                coll_chunk_pairs(x, nx, s, q, p);
#pragma unroll
                for (int k = 0; k < 4; ++k) ns += (4u * q + (u32)k < m_cnt && p.start[k] == p.end[k]) ? 1u : 0u;
            }
        }
#pragma unroll
        for (u32 o = 1; o < (u32)L; o <<= 1) ns += __shfl_xor(ns, o, 64);
        if (in && t == 0) sgl[(size_t)i * ncols + c] = ns;
        else if (!in && t == 0 && c < ncols) sgl[(size_t)i * ncols + c] = 0u;
    }
}

// per column: exclusive prefix of the singles over the operand list (in place), cnt_s[c], and the column's size in 32-bit words
__global__ __launch_bounds__(256)
void k_coll_prefix_singles(u32* __restrict__ sgl, u32 n, u32 ncols, const u32* __restrict__ cnt, u32* __restrict__ cnt_s, u32* __restrict__ words)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    u32 run = 0;
    for (u32 i = 0; i < n; ++i) { const u32 t = sgl[(size_t)i * ncols + c]; sgl[(size_t)i * ncols + c] = run; run += t; }
    cnt_s[c] = run;
    const u32 nm = cnt[c] - run;
    words[c] = ((nm + 3u) & ~3u) + (((run + 7u) & ~7u) >> 1);
}

// scatter pass: L lanes per (operand, column); lane t takes chunks t, t + L, ... of the block as 16-byte loads, the group
// agrees on the write positions (exclusive prefix over the L lanes, a running base over the rounds), every lane writes its
// runs.  (The member's piece of the column is a SET: the order of its entries is free.)
template <int L>
__global__ __launch_bounds__(256)
void k_coll_scatter_split(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 ncols,
                          const u32* __restrict__ pre /* 1-runs before operand i */, const u32* __restrict__ pre_s /* singles before operand i */,
                          const u32* __restrict__ cnt, const u32* __restrict__ cnt_s, const u64* __restrict__ off, u32* __restrict__ runs)
{
    const u32 i = blockIdx.x;
    const u32 t = threadIdx.x % L;
    const u32 nb_i = nblk[i];
    const u32 lane = threadIdx.x & 63u, last = (lane & ~((u32)L - 1u)) + (u32)L - 1u;
    const u64* __restrict__ di = descs[i];
    auto dload = [&](u32 by) -> u64 { const u32 c = by * (256u / L) + threadIdx.x / L; return (c < ncols && c < nb_i) ? di[c] : 0ull; };
    u64 dn = dload(blockIdx.y * COLL_YT);                 // the next tile's descriptor is requested a tile ahead
    for (u32 by = blockIdx.y * COLL_YT; by < (blockIdx.y + 1u) * COLL_YT; ++by) {
        if (by * (256u / L) >= ncols) break;
        const u32 c = by * (256u / L) + threadIdx.x / L;
        const u64 d = dn;
        dn = dload(by + 1u);
        if (DESC_K(d) != K_GAP) continue;                     // (a whole group of L lanes skips together)
        const u32 meta = GMETA(d), len = meta >> 1, s = meta & 1u;
        const u32 m_cnt = coll_runs_of(meta, 1u);
        if (!m_cnt) continue;
        const u32 nchunks = (len + 1u + 7u) >> 3;
        gcptr4 g4 = as_gc4(DESC_P(d));
        const u64 base = off[c];
        const u32 ps = pre_s[(size_t)i * ncols + c], pt = pre[(size_t)i * ncols + c] & 0x3FFFFFFFu;
        const u32 nm_col = cnt[c] - cnt_s[c];
        u32* om = runs + base + (pt - ps);
        u16* os = reinterpret_cast<u16*>(runs + base + ((nm_col + 3u) & ~3u)) + ps;
        for (u32 q0 = 0; q0 < nchunks; q0 += (u32)L) {
            const u32 q = q0 + t;
            const bool have = q < nchunks;
            const u32x4 x = have ? g4[q] : (u32x4)(0u);
            const u32 nx = (have && q + 1u < nchunks) ? g4[q + 1u].x : 0xFFFFFFFFu;
            CollPairs p;
            coll_chunk_pairs(x, nx, s, q, p);
            bool valid[4], single[4];
            u32 ns = 0, nm = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                valid[k] = have && 4u * q + (u32)k < m_cnt;
                single[k] = valid[k] && p.start[k] == p.end[k];
                ns += single[k] ? 1u : 0u; nm += (valid[k] && !single[k]) ? 1u : 0u;
            }
            u32 is = ns, im = nm;                             // inclusive prefixes inside the group
#pragma unroll
            for (u32 o = 1; o < (u32)L; o <<= 1) {
                const u32 a = __shfl_up(is, o, 64), b = __shfl_up(im, o, 64);
                if (t >= o) { is += a; im += b; }
            }
            u16* ws = os + (is - ns); u32* wm = om + (im - nm);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (single[k]) *ws++ = (u16)p.start[k];
                else if (valid[k]) *wm++ = p.start[k] | (p.end[k] << 16);
            }
            os += __shfl(is, (int)last, 64); om += __shfl(im, (int)last, 64);
        }
    }
}

// ---- applying a bag ----
__device__ __forceinline__ void coll_apply_run(u32 r, bool valid, u32* U, int* D, u32& any_long)
{
    u32 s = r & 0xFFFFu, e = r >> 16;
    u32 ws = s >> 5, we = e >> 5;
    u32 lo = ~0u << (s & 31u), hi = ~0u >> (31u - (e & 31u));
    bool same = ws == we;
    if (valid) atomicOr(&U[ws], same ? (lo & hi) : lo);
    if (valid && !same) {
        atomicOr(&U[we], hi);
        if (we - ws > 1u) { atomicAdd(&D[ws + 1u], 1); atomicSub(&D[we], 1); any_long = 1u; }
    }
}

// all entries [0, cnt) of one column, 16 bytes per lane per load, four loads per batch; PF: the next batch is requested
// before the current one is applied (two batches in flight per lane; indices past the end are clamped to the column's last
// chunk so that every iteration issues the same loads).  Returns (per thread) whether it produced an interior-word span
// CW: the four loads of a lane are 1 KiB apart, i.e. a WAVE reads 4 KiB in one piece per batch (instead of four 1-KiB pieces
// 8 KiB apart with the workgroup covering 32 KiB between them)
template <int WG, bool PF, bool CW = false>
__device__ __forceinline__ u32 coll_apply_bag(const u32* __restrict__ runs, u64 off, u32 cnt, u32* U, int* D, u32 tid, bool diag_loads_only = false)
{
    gcptr4 p = as_gc4(runs + off);
    const u32 nq = (cnt + 3u) >> 2;
    u32 any_long = 0u;
    if (!nq) return 0u;
    constexpr u32 JS = CW ? 64u : (u32)WG;                    // distance between a lane's loads of one batch (16-byte units)
    const u32 first = CW ? (tid >> 6) * 256u + (tid & 63u) : tid;
    auto fetch = [&](u32x4 (&v)[4], u32 q0) {
#pragma unroll
        for (u32 j = 0; j < 4; ++j) { u32 q = q0 + j * JS; v[j] = __builtin_nontemporal_load(&p[q < nq ? q : nq - 1u]); }
    };
    auto apply = [&](const u32x4 (&v)[4], u32 q0) {
#ifdef BMX_DIAG
        if (diag_loads_only) {                                // timing probe: the loads alone (one OR per value keeps them alive)
            u32 t = 0;
            for (u32 j = 0; j < 4; ++j) t |= v[j].x | v[j].y | v[j].z | v[j].w;
            if (t == 0x12345u) U[tid] = t;
            return;
        }
#endif
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            u32 q = q0 + j * JS;
            bool in = q < nq;
            u32 e0 = q * 4u;
            coll_apply_run(v[j].x, in && e0 < cnt, U, D, any_long);
            coll_apply_run(v[j].y, in && e0 + 1u < cnt, U, D, any_long);
            coll_apply_run(v[j].z, in && e0 + 2u < cnt, U, D, any_long);
            coll_apply_run(v[j].w, in && e0 + 3u < cnt, U, D, any_long);
        }
    };
    if constexpr (PF) {
        // every thread of the workgroup runs the same number of rounds (the bound does not depend on tid)
        u32x4 a[4], b[4];
        const u32 rounds = (nq + 4u * WG - 1u) / (4u * WG);
        fetch(a, first);
        for (u32 r = 0; r < rounds; r += 2u) {
            fetch(b, first + (r + 1u) * 4u * WG);
            apply(a, first + r * 4u * WG);
            fetch(a, first + (r + 2u) * 4u * WG);
            apply(b, first + (r + 1u) * 4u * WG);
        }
    } else {
        const u32 rounds = (nq + 4u * WG - 1u) / (4u * WG);
        for (u32 r = 0; r < rounds; ++r) { u32x4 v[4]; fetch(v, first + r * 4u * WG); apply(v, first + r * 4u * WG); }
    }
    return any_long;
}

// the single-bit runs of a split bag: ns 16-bit positions behind the multi-bit runs; 16 bytes = 8 positions per lane and load,
// the same batch shape as coll_apply_bag
template <int WG, bool CW>
__device__ __forceinline__ void coll_apply_singles(const u32* __restrict__ words, u64 off_words, u32 ns, u32* U, u32 tid)
{
    gcptr4 p = as_gc4(words + off_words);
    const u32 nq = (ns + 7u) >> 3;
    if (!nq) return;
    constexpr u32 JS = CW ? 64u : (u32)WG;
    const u32 first = CW ? (tid >> 6) * 256u + (tid & 63u) : tid;
    const u32 rounds = (nq + 4u * WG - 1u) / (4u * WG);
    for (u32 r = 0; r < rounds; ++r) {
        u32x4 v[4];
        const u32 q0 = first + r * 4u * WG;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) { u32 q = q0 + j * JS; v[j] = __builtin_nontemporal_load(&p[q < nq ? q : nq - 1u]); }
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            const u32 q = q0 + j * JS;
            const u32 e0 = q * 8u;
            const u32 w4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (u32 t = 0; t < 4; ++t) {
                const u32 lo = w4[t] & 0xFFFFu, hi = w4[t] >> 16;
                if (q < nq && e0 + 2u * t < ns) atomicOr(&U[lo >> 5], 1u << (lo & 31u));
                if (q < nq && e0 + 2u * t + 1u < ns) atomicOr(&U[hi >> 5], 1u << (hi & 31u));
            }
        }
    }
}

// interior words covered by a long run: prefix sum of D over the 2048 words; covered words become all ones.  D is left
// zeroed.  Called by the whole workgroup between barriers; sm: WG / 64 ints.
template <int WG>
__device__ __forceinline__ void coll_fold(u32* U, int* D, int* sm, u32 tid)
{
    constexpr u32 W = 2048u / WG;                     // words per thread (8 for 256 threads)
    const u32 lane = tid & 63u, wave = tid >> 6;
    int d[W]; int sum = 0;
#pragma unroll
    for (u32 k = 0; k < W; k += 4u) {
        u32x4 t = *reinterpret_cast<u32x4*>(&D[tid * W + k]);
        d[k] = (int)t.x; d[k + 1] = (int)t.y; d[k + 2] = (int)t.z; d[k + 3] = (int)t.w;
        *reinterpret_cast<u32x4*>(&D[tid * W + k]) = (u32x4)(0u);
    }
#pragma unroll
    for (u32 k = 0; k < W; ++k) sum += d[k];
    int incl = (int)wave_scan_incl((u32)sum, lane);
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    int running = incl - sum;
#pragma unroll
    for (u32 i = 0; i < WG / 64u; ++i) if (i < wave) running += sm[i];
#pragma unroll
    for (u32 k = 0; k < W; ++k) { running += d[k]; if (running > 0) U[tid * W + k] = ~0u; }
    __syncthreads();
}

// algorithmic operand bytes of a vector: 8,192 B per bit-block, 2 x (len + 1) B per GAP block
__global__ __launch_bounds__(256)
void k_vec_alg_bytes(const u64* __restrict__ desc, u32 nblocks, u64* __restrict__ out)
{
    u32 nb = blockIdx.x * blockDim.x + threadIdx.x;
    u32 b = 0;
    if (nb < nblocks) {
        u64 d = desc[nb];
        u32 k = DESC_K(d);
        b = k == K_BIT ? 8192u : (k == K_GAP ? 2u * ((GMETA(d) >> 1) + 1u) : 0u);
    }
    b = wave_sum(b);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)b);
}

enum { COLL_OR = 0, COLL_AND_STORE = 1, COLL_AND_COUNT = 2 };

// One workgroup per block column.
//   COLL_OR         result = union of the bag (polarity 1), stored with the aggregator's optimisation mode
//                   (combine_or, src/bmaggregator.h:1101,1658)
//   COLL_AND_STORE  result = NOT union(AND bag, polarity 0) AND NOT union(SUB bag, polarity 1), stored with opt_compress
//                   (combine_and_sub, :1162,1210)
//   COLL_AND_COUNT  the same, counted (counts-only pipeline of one arg-group, :1292-1399)
template <int MODE, int WG, bool PF = false, bool CW = false>
__global__ __launch_bounds__(WG)
void k_coll_apply(const u32* __restrict__ runs, const u64* __restrict__ off, const u32* __restrict__ cnt,
                  const u32* __restrict__ flags, u32 ncols_a,
                  const u32* __restrict__ s_runs, const u64* __restrict__ s_off, const u32* __restrict__ s_cnt,
                  const u32* __restrict__ s_flags, u32 ncols_s,
                  u32 col_base, u32 ncols, int opt_compress, u64* __restrict__ counts,
                  uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, u32 hint_from, u32 hint_to,
                  FoldOut kinds /* COLL_OR without opt_compress: the result's kind counts folded in-kernel (no layout scan) */,
                  const u32* __restrict__ a_cnt_s /* OR bag in the split format: its singles per column (else null) */,
                  const u32* __restrict__ s_cnt_s /* the same for the SUB bag */)
{
    __shared__ __attribute__((aligned(16))) u32 U[2048];
    __shared__ __attribute__((aligned(16))) int D[2048];
    __shared__ int sm[WG / 64];
    __shared__ u32 s_long;
    __shared__ u32 part[WG / 64];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#ifdef BMX_DIAG
    const int diag = opt_compress & 1536;                     // (tuning build: timing probes, see coll_diag_bits in bmx.hip)
    opt_compress &= 1;
#endif
    const u32 c = col_base + blockIdx.x;
    if (c >= ncols) return;
    if (MODE != COLL_OR && (c < hint_from || c >= hint_to)) {
        if (MODE == COLL_AND_STORE && wave == 0) store_trivial(K_NULL, c, desc, st, lane);
        return;
    }
    const u32 fl = c < ncols_a ? uniform32(flags[c]) : (MODE == COLL_OR ? 0u : COLL_FLAG_NULL);
    const u32 n_ent = c < ncols_a ? uniform32(cnt[c]) : 0u;
    if (MODE == COLL_OR) {
        // trivial columns (every thread of the workgroup takes the same path: the kind fold below has a barrier)
        if ((fl & COLL_FLAG_FULL) || COLL_NGAP(fl) == 0u) {
            const u32 k = (fl & COLL_FLAG_FULL) ? (u32)K_FULL : (u32)K_NULL;
            if (wave == 0) store_trivial(k, c, desc, st, lane);
            if (kinds.slots) kind_fanin_fold(wave == 0 ? k : 4u, kinds, lane, wave);
            return;
        }
    } else {
        const u32 sfl = (s_flags && c < ncols_s) ? uniform32(s_flags[c]) : 0u;
        // any NULL operand in the AND list, or a FULL one in the SUB list: the column is empty (:2327, :1746)
        if ((fl & COLL_FLAG_NULL) || (sfl & COLL_FLAG_FULL)) {
            if (MODE == COLL_AND_STORE && wave == 0) store_trivial(K_NULL, c, desc, st, lane);
            return;
        }
    }
    constexpr u32 W = 2048u / WG;
#pragma unroll
    for (u32 k = 0; k < W; k += 4u) {
        *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
        *reinterpret_cast<u32x4*>(&D[tid * W + k]) = (u32x4)(0u);
    }
    if (tid == 0) s_long = 0u;
    __syncthreads();
    const u64 a_off = uniform64(off[c < ncols_a ? c : 0u]);
    const u32 a_ns = (a_cnt_s && c < ncols_a) ? uniform32(a_cnt_s[c]) : 0u;     // (split bag: the singles among the n_ent runs)
    const u32 a_nm = n_ent - a_ns;
#ifdef BMX_DIAG
    u32 al = coll_apply_bag<WG, PF, CW>(runs, a_off, a_nm, U, D, tid, (diag & 512) != 0);
    if (diag & 1024) return;                                  // timing probe: no fold, no store
#else
    u32 al = coll_apply_bag<WG, PF, CW>(runs, a_off, a_nm, U, D, tid);
#endif
    if (a_ns) coll_apply_singles<WG, CW>(runs, a_off + ((a_nm + 3u) & ~3u), a_ns, U, tid);
    if (al) s_long = 1u;
    __syncthreads();
    if (s_long) coll_fold<WG>(U, D, sm, tid);                 // (block-uniform; the barriers inside are reached by every thread)
    if (MODE == COLL_OR) {
        u32 kind = 4u;
        if (wave == 0) {
            Blk b; blk_from_lds(b, U, lane);
            kind = store_result_mode(b, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);
        }
        if (kinds.slots) kind_fanin_fold(kind, kinds, lane, wave);
        return;
    }
    // AND: the accumulator is the complement of the union of the 0-runs ...
    u32 acc[W];
#pragma unroll
    for (u32 k = 0; k < W; ++k) acc[k] = ~U[tid * W + k];
    // ... minus the union of the SUB bag's 1-runs
    const u32 s_ent = (s_cnt && c < ncols_s) ? uniform32(s_cnt[c]) : 0u;
    if (s_ent) {
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < W; k += 4u) *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
        if (tid == 0) s_long = 0u;
        __syncthreads();
        const u64 so = uniform64(s_off[c]);
        const u32 s_ns = s_cnt_s ? uniform32(s_cnt_s[c]) : 0u, s_nm = s_ent - s_ns;
        u32 sl = coll_apply_bag<WG, PF, CW>(s_runs, so, s_nm, U, D, tid);
        if (s_ns) coll_apply_singles<WG, CW>(s_runs, so + ((s_nm + 3u) & ~3u), s_ns, U, tid);
        if (sl) s_long = 1u;
        __syncthreads();
        if (s_long) coll_fold<WG>(U, D, sm, tid);
#pragma unroll
        for (u32 k = 0; k < W; ++k) acc[k] &= ~U[tid * W + k];
    }
    if (MODE == COLL_AND_COUNT) {
        u32 pc = 0;
#pragma unroll
        for (u32 k = 0; k < W; ++k) pc += (u32)__popc(acc[k]);
        pc = wave_sum(pc);
        if (lane == 0) part[wave] = pc;
        __syncthreads();
        if (tid == 0) {
            u32 t = 0;
#pragma unroll
            for (u32 i = 0; i < WG / 64u; ++i) t += part[i];
            if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[0]), (unsigned long long)t);
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < W; ++k) U[tid * W + k] = acc[k];
    __syncthreads();
    if (wave == 0) { Blk b; blk_from_lds(b, U, lane); store_result(b, c, 1, slab, desc, st, lane); }
}

// ---------------------------------------------------------------------------
// Index-list output: a vector as SORTED positions of its set bits -- what an inverted-index consumer reads back
// (aggregator::combine_and_sub(BII bi, ...) src/bmaggregator.h:1226-1284 walks the result blocks through
// for_each_bit_blk / bit_visitor_back_inserter_adaptor; sparse_vector_scanner::find_eq(sv, value, BII)
// src/bmsparsevec_algo.h:1096).  Two passes: per-block popcounts (k_block_counts) + running sum (k_rs_scan), then one
// wave per block writes its positions behind the blocks before it (k_expand_indices).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_block_counts(const u64* __restrict__ desc, u32 nblocks, u32* __restrict__ bcount)
{
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks) return;
    u64 d = uniform64(desc[nb]);
    u32 k = DESC_K(d), c = 0;
    if (k == K_FULL) c = 65536u;
    else if (k == K_BIT) { Blk b; blk_load(b, as_gc4(DESC_P(d)), lane); c = wave_sum(blk_lane_popcount(b)); }
    else if (k == K_GAP) c = wave_sum(gap_lane_popcount(as_gc16(DESC_P(d)), lane, GMETA(d)));
    if (lane == 0) bcount[nb] = c;
}

template <typename T>
__device__ __forceinline__ void emit_word_bits(u32 w, u64 base, T* __restrict__ out, u64& off)
{
    while (w) { u32 b = (u32)__builtin_ctz(w); out[off++] = (T)(base + b); w &= w - 1u; }
}

// T = u32 / u64 positions.  rcount = inclusive running count per block; positions at or beyond `cap` are not written
// (the host checks the total against the capacity before it reads anything)
template <typename T>
__global__ __launch_bounds__(256)
void k_expand_indices(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, T* __restrict__ out, u64 cap, u64 pos_base)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks) return;
    u64 d = uniform64(desc[nb]);
    u32 k = DESC_K(d);
    if (k == K_NULL) return;
    u64 first = nb ? rcount[nb - 1u] : 0ull;
    u64 last = rcount[nb];
    if (last > cap) return;                                   // (never read: the call fails with the needed size)
    u64 bit0 = pos_base + ((u64)nb << 16);
    if (k == K_FULL) {
        for (u32 i = lane; i < 65536u; i += 64u) out[first + i] = (T)(bit0 + i);
        return;
    }
    Blk b;
    blk_from_desc(d, b, lds + wave * 2048u, lane);
    u64 row_off = first;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // row i = words i*256 .. i*256+255; lane l holds the four consecutive words i*256 + 4l .. + 3
        u32 c = (u32)__popc(b.r[i].x) + (u32)__popc(b.r[i].y) + (u32)__popc(b.r[i].z) + (u32)__popc(b.r[i].w);
        u32 incl = wave_scan_incl(c, lane);
        u32 tot = uniform32(__shfl(incl, 63, 64));
        u64 off = row_off + (incl - c);
        u64 wb = bit0 + ((u64)((u32)i * 256u + lane * 4u) << 5);
        emit_word_bits<T>(b.r[i].x, wb, out, off);
        emit_word_bits<T>(b.r[i].y, wb + 32u, out, off);
        emit_word_bits<T>(b.r[i].z, wb + 64u, out, off);
        emit_word_bits<T>(b.r[i].w, wb + 96u, out, off);
        row_off += tot;
    }
}

// position of the k-th set bit (k = 1..popcount) of a word, without a loop: five halving steps on popcounts (word_select,
// src/bmfunc.h:1084, does `w &= w - 1` k - 1 times: up to 31 dependent iterations that a wave runs for its slowest lane)
__device__ __forceinline__ u32 select_in_word(u32 w, u32 k)
{
    u32 pos = 0u, c;
    c = (u32)__popc(w & 0xFFFFu); { const bool m = k > c; k -= m ? c : 0u; pos += m ? 16u : 0u; w = m ? w >> 16 : w; }
    c = (u32)__popc(w & 0xFFu);   { const bool m = k > c; k -= m ? c : 0u; pos += m ? 8u : 0u;  w = m ? w >> 8 : w; }
    c = (u32)__popc(w & 0xFu);    { const bool m = k > c; k -= m ? c : 0u; pos += m ? 4u : 0u;  w = m ? w >> 4 : w; }
    c = (u32)__popc(w & 0x3u);    { const bool m = k > c; k -= m ? c : 0u; pos += m ? 2u : 0u;  w = m ? w >> 2 : w; }
    c = w & 1u;                   { const bool m = k > c; pos += m ? 1u : 0u; }
    return pos;
}

// ---------------------------------------------------------------------------
// Rank lines (round 3): count_to / rank with ONE random 128-byte line per query.
// k_rank reads four things per query -- descriptor, running count, cumulative count of the 1024-bit wave, the bit line --
// and the PMC shows 1.6 lines per query leaving the L2 (the 7.8 MB cumulative table does not stay in a 4 MB L2): 0.8 of the
// box's random-line rate by traffic, 0.49 by what the query needs.  The classic fix of succinct rank dictionaries
// (rank9 / poppy: counts interleaved with the bits they describe) maps directly onto 128-byte lines: a line holds a 64-bit
// count of the ones BEFORE it in the whole vector and the next 960 bits, so rank(n) = header + popcount of the line's bits
// up to n -- one line, no index table, no block kinds (GAP / FULL / NULL blocks are stored expanded).  A 64 Kbit block is 69
// lines (the last one partly used): 8,832 B per block = +7.8 % over raw bits, on top of the vector itself: HBM capacity is
// not the constraint here (a 4e9-bit vector: 539 MB next to its 500 MB).  Same results by construction: the header is the
// running count the index already has, the rest is a popcount.
// ---------------------------------------------------------------------------
#define RL_LINES 69u            // ceil(65536 / 960)
#define RL_BITS 960u            // data bits per line (30 words behind the 2-word header)

__global__ __launch_bounds__(256)
void k_rs_lines(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, u32* __restrict__ lines,
                u16* __restrict__ dir8 /* [nblocks][8]: ones of the block before bit 8192 k */)
{
    __shared__ u32 lds_all[4 * 2048];
    __shared__ u64 hdr_all[4 * 72];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks) return;
    u32* lds = lds_all + wave * 2048u;
    u64* hdr = hdr_all + wave * 72u;
    Blk b;
    blk_from_desc(uniform64(desc[nb]), b, lds, lane);
    blk_to_lds(b, lds, lane);                                 // linear word order
    const u64 before = nb ? rcount[nb - 1u] : 0ull;
    // ones of every line, then their exclusive prefix: lanes 0..63 own lines 0..63, lanes 0..4 also lines 64..68
    u32 c0 = 0, c1 = 0;
    for (u32 t = 0; t < 30u; ++t) {
        u32 w0 = lane * 30u + t, w1 = (lane + 64u) * 30u + t;
        c0 += w0 < 2048u ? (u32)__popc(lds[w0]) : 0u;
        c1 += (lane + 64u < RL_LINES && w1 < 2048u) ? (u32)__popc(lds[w1]) : 0u;
    }
    u32 i0 = wave_scan_incl(c0, lane);
    u32 tot0 = uniform32(__shfl(i0, 63, 64));
    u32 i1 = wave_scan_incl(c1, lane);
    hdr[lane] = before + (u64)(i0 - c0);
    if (lane + 64u < RL_LINES) hdr[lane + 64u] = before + (u64)tot0 + (u64)(i1 - c1);
    // the block's coarse directory for select: ones before every 8,192-bit octant (256 words each)
    {
        u32 pc = 0;
        for (u32 t = 0; t < 32u; ++t) pc += (u32)__popc(lds[lane * 32u + t]);      // lane = 32 words = 1/64 of the block
        u32 incl = wave_scan_incl(pc, lane);
        u32 excl = incl - pc;
        if ((lane & 7u) == 0u) dir8[(size_t)nb * 8u + (lane >> 3)] = (u16)excl;    // lanes 0, 8, 16, ..: 256 words apiece
    }
    u32* out = lines + (size_t)nb * (RL_LINES * 32u);
    for (u32 o = lane; o < RL_LINES * 32u; o += 64u) {        // coalesced 256-byte stores
        u32 line = o >> 5, t = o & 31u;
        u32 v;
        if (t < 2u) { u64 h = hdr[line]; v = t ? (u32)(h >> 32) : (u32)h; }
        else { u32 w = line * 30u + (t - 2u); v = w < 2048u ? lds[w] : 0u; }
        out[o] = v;
    }
}

// Groups of 2 or 4 lanes sit inside a quad: their exchanges are DPP quad permutes (one VALU instruction each) instead of
// ds_bpermute round trips through the LDS crossbar (__shfl*).  quad_perm control = sel0 | sel1 << 2 | sel2 << 4 | sel3 << 6.
template <int CTRL>
__device__ __forceinline__ u32 quad_perm(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// the value of the group's lane 0 in every lane of the group
template <u32 LPQ>
__device__ __forceinline__ u32 group_first(u32 v, u32 lane)
{
    if constexpr (LPQ == 4u) return quad_perm<0x00>(v);                       // [0,0,0,0]
    else if constexpr (LPQ == 2u) return quad_perm<0xA0>(v);                  // [0,0,2,2]
    else return __shfl(v, lane & ~(LPQ - 1u), 64);
}
template <u32 LPQ>
__device__ __forceinline__ u32 group_sum(u32 v)
{
    if constexpr (LPQ == 2u || LPQ == 4u) {
        v += quad_perm<0xB1>(v);                                              // [1,0,3,2]: lane ^ 1
        if constexpr (LPQ == 4u) v += quad_perm<0x4E>(v);                     // [2,3,0,1]: lane ^ 2
        return v;
    } else {
#pragma unroll
        for (u32 o = 1; o < LPQ; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
}
template <u32 LPQ>
__device__ __forceinline__ u32 group_max(u32 v)
{
    if constexpr (LPQ == 2u || LPQ == 4u) {
        u32 t = quad_perm<0xB1>(v); v = t > v ? t : v;
        if constexpr (LPQ == 4u) { t = quad_perm<0x4E>(v); v = t > v ? t : v; }
        return v;
    } else {
#pragma unroll
        for (u32 o = 1; o < LPQ; o <<= 1) { u32 t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
        return v;
    }
}
// exclusive prefix of `mine` inside the group of LPQ lanes; total in `tot`
template <u32 LPQ>
__device__ __forceinline__ u32 group_excl(u32 mine, u32 sub, u32 lane, u32& tot)
{
    u32 incl = mine;
    if constexpr (LPQ == 2u) {
        const u32 t = quad_perm<0xA0>(incl);                                  // [0,0,2,2]: the lower lane of the pair
        if (sub >= 1u) incl += t;
        tot = quad_perm<0xF5>(incl);                                          // [1,1,3,3]: the pair's upper lane
    } else if constexpr (LPQ == 4u) {
        u32 t = quad_perm<0x90>(incl);                                        // [0,0,1,2]: lane - 1
        if (sub >= 1u) incl += t;
        t = quad_perm<0x44>(incl);                                            // [0,1,0,1]: lane - 2
        if (sub >= 2u) incl += t;
        tot = quad_perm<0xFF>(incl);                                          // [3,3,3,3]
    } else {
#pragma unroll
        for (u32 o = 1; o < LPQ; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (sub >= o) incl += t; }
        tot = __shfl(incl, (lane & ~(LPQ - 1u)) + (LPQ - 1u), 64);
    }
    return incl - mine;
}

template <u32 LPQ>
__global__ __launch_bounds__(256)
void k_rank_lines(const u32* __restrict__ lines, u32 nblocks, u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ out)
{
    constexpr u32 NV = 8u / LPQ;
    const u32 sub = threadIdx.x & (LPQ - 1u);
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + (64u / LPQ) - 1ull) / (64u / LPQ) * (64u / LPQ);
    for (; qi < nq_round; qi += stride) {
        const bool live = qi < nq;
        const u64 n = live ? q[qi] : 0ull;
        const u64 nb64 = n >> 16;
        const bool in = live && nb64 < nblocks;
        const u32 off = in ? (u32)(n & 0xFFFFu) : 0u;
        const u32 j = off / RL_BITS, pos = off - j * RL_BITS;            // line of the block, bit inside its data words
        gcptr4 p = as_gc4(lines + ((size_t)(in ? (u32)nb64 : 0u) * RL_LINES + j) * 32u) + sub * NV;
        u32x4 v[NV];
#pragma unroll
        for (u32 i = 0; i < NV; ++i) v[i] = p[i];
        u32 part = 0;
#pragma unroll
        for (u32 i = 0; i < NV; ++i) {
            const u32 w = (sub * NV + i) * 4u;                            // word of the line held in v[i].x
            // data word d = w - 2 covers bits [32 d, 32 d + 31] of the line; words 0, 1 are the header
            if (w >= 2u) part += word_count_to(v[i].x, w - 2u, pos);
            if (w + 1u >= 2u) part += word_count_to(v[i].y, w + 1u - 2u, pos);
            part += word_count_to(v[i].z, w, pos) + word_count_to(v[i].w, w + 1u, pos);
        }
        part = group_sum<LPQ>(part);
        const u64 head = ((u64)v[0].y << 32) | v[0].x;                   // valid in the group's lane 0 (sub == 0)
        if (live && sub == 0) out[qi] = in ? head + part : total;        // past the end: the total (src/bm.h:3133)
    }
}

// ---------------------------------------------------------------------------
// k_select with LPQ = 2 / 4 lanes per query (k_select gives a query 8 lanes).  select is three dependent stages -- the last
// <= 32 running counts, then previous count + descriptor + cumulative row, then the bit line -- and the PMC shows it at
// half the random-line rate its 2 lines per query would allow: it is bound by how many queries a CU keeps in flight, so
// fewer lanes per query (16 / 32 queries per wave step instead of 8) is the lever.  Same arithmetic, regrouped.
// ---------------------------------------------------------------------------
template <u32 LPQ>
__global__ __launch_bounds__(256)
void k_select_l(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, const u16* __restrict__ cum,
                const u16* __restrict__ gidx, const u64* __restrict__ sample, u32 nsamples, u32 shift,
                u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    constexpr u32 NV = 8u / LPQ;                  // 16-byte pieces of a 128-byte row / line per lane
    constexpr u32 NR = 32u / LPQ;                 // running counts of the last level per lane
    __shared__ u64 s_sample[2048];
    for (u32 i = threadIdx.x; i < nsamples; i += blockDim.x) s_sample[i] = sample[i];
    __syncthreads();
    const u32 lane = lane_id();
    const u32 sub = lane & (LPQ - 1u);
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + (64u / LPQ) - 1ull) / (64u / LPQ) * (64u / LPQ);
    for (; qi < nq_round; qi += stride) {
        const bool live = qi < nq;
        const u64 r = live ? q[qi] : 0ull;
        const bool ok = live && r != 0ull && r <= total && nblocks != 0u;
        u32 sr_lo = 0, sr_hi = 0;
        if (ok) {
            // rs_index::find (src/bmrs.h:492): top level in LDS
            u32 glo = 0, ghi = nsamples - 1u;
            while (glo < ghi) { u32 mid = glo + ((ghi - glo) >> 1); if (s_sample[mid] < r) glo = mid + 1u; else ghi = mid; }
            u32 lo = glo << shift, hi = ((glo + 1u) << shift) - 1u;
            if (hi > nblocks - 1u) hi = nblocks - 1u;
            while (hi - lo >= 32u) { u32 mid = lo + ((hi - lo) >> 1); if (rcount[mid] < r) lo = mid + 1u; else hi = mid; }
            sr_lo = lo; sr_hi = hi;
        }
        // stage 1: the last <= 32 running counts, NR per lane (unconditional reads, index clamped)
        u32 below = 0;
        {
            u64 rc[NR];
#pragma unroll
            for (u32 t = 0; t < NR; ++t) { u32 idx = sr_lo + sub * NR + t; rc[t] = rcount[idx <= sr_hi ? idx : sr_hi]; }
#pragma unroll
            for (u32 t = 0; t < NR; ++t) { u32 idx = sr_lo + sub * NR + t; below += (ok && idx <= sr_hi && rc[t] < r) ? 1u : 0u; }
        }
        below = group_sum<LPQ>(below);
        const u32 nb = ok ? sr_lo + below : 0u;
        // stage 2: previous running count, descriptor, my share of the 64-entry cumulative row
        const u64 prev = rcount[nb ? nb - 1u : 0u];
        const u64 d = desc[nb];
        u32x4 cv[NV];
#pragma unroll
        for (u32 i = 0; i < NV; ++i) cv[i] = as_gc4(cum + (size_t)nb * 64u)[sub * NV + i];
        const u32 kd = ok ? DESC_K(d) : K_NULL;
        u32 rr = 0, w = 0; u64 result = 0;
        if (ok) {
            rr = (u32)(r - (nb ? prev : 0ull));                       // 1..65536 inside the block
            if (kd == K_FULL) result = ((u64)nb << 16) + rr - 1u;
        }
        {
            // digest wave: the last w with cum[w] < rr (cum[0] = 0 < rr always; the row is non-decreasing)
            u32 nlt = 0, best = 0;
#pragma unroll
            for (u32 i = 0; i < NV; ++i) {
                const u32 c16[8] = {cv[i].x & 0xFFFFu, cv[i].x >> 16, cv[i].y & 0xFFFFu, cv[i].y >> 16, cv[i].z & 0xFFFFu, cv[i].z >> 16, cv[i].w & 0xFFFFu, cv[i].w >> 16};
#pragma unroll
                for (int j = 0; j < 8; ++j) { bool lt = c16[j] < rr; nlt += lt ? 1u : 0u; best = (lt && c16[j] > best) ? c16[j] : best; }
            }
            nlt = group_sum<LPQ>(nlt);
            best = group_max<LPQ>(best);
            if (ok && kd != K_FULL) { w = nlt - 1u; rr -= best; }        // 1..1024 inside the wave
        }
        // stage 3: bit-blocks: my share of the wave's 128-byte line (others read their valid cumulative row and ignore it)
        u32x4 v[NV];
        {
            gcptr4 p = kd == K_BIT ? as_gc4(DESC_P(d)) + w * 8u + sub * NV : as_gc4(cum + (size_t)nb * 64u) + sub * NV;
#pragma unroll
            for (u32 i = 0; i < NV; ++i) v[i] = p[i];
        }
        {
            const bool isb = kd == K_BIT;
            u32 wd[4 * NV]; u32 mine = 0;
#pragma unroll
            for (u32 i = 0; i < NV; ++i) {
                wd[4 * i] = isb ? v[i].x : 0u; wd[4 * i + 1] = isb ? v[i].y : 0u; wd[4 * i + 2] = isb ? v[i].z : 0u; wd[4 * i + 3] = isb ? v[i].w : 0u;
            }
#pragma unroll
            for (u32 k = 0; k < 4 * NV; ++k) mine += (u32)__popc(wd[k]);
            u32 tot;
            const u32 excl = group_excl<LPQ>(mine, sub, lane, tot);
            if (isb && rr > excl && rr <= excl + mine) {
                u32 need = rr - excl;                                // 1..mine within my words
                u32 word = 0, wi = 0; bool got = false;
#pragma unroll
                for (u32 k = 0; k < 4 * NV; ++k) {
                    const u32 pc = (u32)__popc(wd[k]);
                    if (!got) { if (need <= pc) { word = wd[k]; wi = k; got = true; } else need -= pc; }
                }
                const u32 bit = (w * 32u + sub * 4u * NV + wi) * 32u + select_in_word(word, need);   // word_select (src/bmfunc.h:1084)
                pos[qi] = ((u64)nb << 16) + bit;
            }
        }
        {
            // GAP blocks: gap_find_rank (src/bmfunc.h:3457) restricted to the digest wave, LPQ lanes x 4 runs per round
            const bool gq = ok && kd == K_GAP;
            if (__ballot(gq) != 0ull) {
                gcptr16 g = gq ? as_gc16(DESC_P(d)) : (gcptr16)(uintptr_t)cum;       // idle lanes read a valid dummy
                u32 len = 0, s0 = 0, lo = 1, need = rr;
                const u32 from = w << 10;
                if (gq) { len = GMETA(d) >> 1; s0 = GMETA(d) & 1u; lo = gidx[(size_t)nb * 64u + w]; }
                bool searching = gq;
                for (u32 k0 = lo; __ballot(searching && k0 <= len) != 0ull; k0 += 4u * LPQ) {
                    const u32 kf = k0 + sub * 4u;
                    u32 ev[5];
#pragma unroll
                    for (u32 t = 0; t < 5; ++t) { u32 kk = kf - 1u + t; ev[t] = (u32)g[(searching && kk <= len) ? kk : 1u]; }
                    u32 cnt[4], st[4], mine = 0;
#pragma unroll
                    for (u32 t = 0; t < 4; ++t) {
                        const u32 k = kf + t;
                        const bool one = searching && k <= len && (s0 ^ ((k - 1u) & 1u)) != 0u;
                        u32 start = (k == 1u) ? 0u : ev[t] + 1u;
                        if (start < from) start = from;
                        st[t] = start;
                        cnt[t] = (one && ev[t + 1u] >= start) ? ev[t + 1u] - start + 1u : 0u;
                        mine += cnt[t];
                    }
                    u32 gtot;
                    const u32 excl = group_excl<LPQ>(mine, sub, lane, gtot);
                    if (searching && mine != 0u && need > excl && need <= excl + mine) {
                        u32 rem = need - excl;
#pragma unroll
                        for (u32 t = 0; t < 4; ++t) {
                            if (rem != 0u && rem <= cnt[t]) { pos[qi] = ((u64)nb << 16) + st[t] + rem - 1u; rem = 0u; }
                            else if (rem != 0u) rem -= cnt[t];
                        }
                    }
                    if (searching && need <= gtot) searching = false;        // some lane of the group had the hit
                    else need -= gtot;
                }
            }
        }
        if (live && sub == 0) {
            found[qi] = ok ? 1 : 0;
            if (!ok) pos[qi] = 0;
            else if (kd != K_BIT && kd != K_GAP) pos[qi] = result;
        }
    }
}

// ---------------------------------------------------------------------------
// select over rank lines (round 3).  k_select[_l] reads, per query, <= 32 running counts (L2), the block's 128-byte
// cumulative row (misses the L2 six times out of ten) and the bit line: 2 lines per query past the L2 and a long
// instruction path.  With the vector laid out as rank lines the block search stays (LDS samples + <= 32 running counts),
// then a 16-byte directory entry (ones before every 8,192-bit octant of the block: 1 MB for 4e9 bits, L2-resident) gives
// the octant, the line inside it is GUESSED by interpolation (ones are spread evenly enough in most data) and verified
// against the line's own header -- the exact count of ones before it -- stepping to the neighbour line when the guess is
// off.  One line per query when the guess holds, and never a wrong answer: the headers decide, not the guess.
// ---------------------------------------------------------------------------
template <u32 LPQ>
__global__ __launch_bounds__(256)
void k_select_lines(const u32* __restrict__ lines, const u16* __restrict__ dir8, u32 nblocks, const u64* __restrict__ rcount,
                    const u64* __restrict__ sample, u32 nsamples, u32 shift, u64 total,
                    const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    constexpr u32 NV = 8u / LPQ;
    constexpr u32 NR = 32u / LPQ;
    __shared__ u64 s_sample[2048];
    for (u32 i = threadIdx.x; i < nsamples; i += blockDim.x) s_sample[i] = sample[i];
    __syncthreads();
    const u32 lane = lane_id();
    const u32 sub = lane & (LPQ - 1u);
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + (64u / LPQ) - 1ull) / (64u / LPQ) * (64u / LPQ);
    for (; qi < nq_round; qi += stride) {
        const bool live = qi < nq;
        const u64 r = live ? q[qi] : 0ull;
        const bool ok = live && r != 0ull && r <= total && nblocks != 0u;
        u32 sr_lo = 0, sr_hi = 0;
        if (ok) {
            u32 glo = 0, ghi = nsamples - 1u;
            while (glo < ghi) { u32 mid = glo + ((ghi - glo) >> 1); if (s_sample[mid] < r) glo = mid + 1u; else ghi = mid; }
            u32 lo = glo << shift, hi = ((glo + 1u) << shift) - 1u;
            if (hi > nblocks - 1u) hi = nblocks - 1u;
            while (hi - lo >= 32u) { u32 mid = lo + ((hi - lo) >> 1); if (rcount[mid] < r) lo = mid + 1u; else hi = mid; }
            sr_lo = lo; sr_hi = hi;
        }
        u32 below = 0;
        {
            u64 rc[NR];
#pragma unroll
            for (u32 t = 0; t < NR; ++t) { u32 idx = sr_lo + sub * NR + t; rc[t] = rcount[idx <= sr_hi ? idx : sr_hi]; }
#pragma unroll
            for (u32 t = 0; t < NR; ++t) { u32 idx = sr_lo + sub * NR + t; below += (ok && idx <= sr_hi && rc[t] < r) ? 1u : 0u; }
        }
        below = group_sum<LPQ>(below);
        const u32 nb = ok ? sr_lo + below : 0u;
        // ones before the block / in the block, and the octant directory of the block (three independent L2-resident reads)
        const u64 prev = rcount[nb ? nb - 1u : 0u];
        const u64 cur = rcount[nb];
        const u32x4 dv = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((uintptr_t)(dir8 + (size_t)nb * 8u));
        const u64 before = nb ? prev : 0ull;
        const u32 rr = ok ? (u32)(r - before) : 1u;                 // 1..65536 inside the block
        const u32 btot = ok ? (u32)(cur - before) : 1u;
        const u32 d8[8] = {dv.x & 0xFFFFu, dv.x >> 16, dv.y & 0xFFFFu, dv.y >> 16, dv.z & 0xFFFFu, dv.z >> 16, dv.w & 0xFFFFu, dv.w >> 16};
        u32 k = 0;
#pragma unroll
        for (u32 t = 1; t < 8u; ++t) k += (d8[t] < rr) ? 1u : 0u;        // the last octant with fewer than rr ones before it (non-decreasing)
        u32 ob = 0, oe = 0;
#pragma unroll
        for (u32 t = 0; t < 8u; ++t) { if (t == k) ob = d8[t]; if (t == k + 1u) oe = d8[t]; }
        if (k == 7u) oe = btot;
        const u32 cw = oe > ob ? oe - ob : 1u;                        // ones inside the octant (>= 1: the rr-th one is there)
        // interpolated bit position of the rr-th one inside the octant -> the line to look at first
        u32 guess = (k << 13) + (u32)(((u64)(rr - ob - 1u) * 8192ull) / cw);
        if (guess > 65535u) guess = 65535u;
        u32 j = guess / RL_BITS;
        bool searching = ok;
        u64 result = 0;
        for (u32 it = 0; __ballot(searching) != 0ull && it < 80u; ++it) {
            gcptr4 p = as_gc4(lines + ((size_t)nb * RL_LINES + j) * 32u) + sub * NV;
            u32x4 v[NV];
#pragma unroll
            for (u32 i = 0; i < NV; ++i) v[i] = p[i];
            u32 wd[4 * NV];
#pragma unroll
            for (u32 i = 0; i < NV; ++i) { wd[4 * i] = v[i].x; wd[4 * i + 1] = v[i].y; wd[4 * i + 2] = v[i].z; wd[4 * i + 3] = v[i].w; }
            // the header (count before the line, whole vector) sits in the first two words of the group's lane 0
            u32 hlo = sub == 0 ? wd[0] : 0u, hhi = sub == 0 ? wd[1] : 0u;
            hlo = group_first<LPQ>(hlo, lane); hhi = group_first<LPQ>(hhi, lane);
            const u64 hdr = ((u64)hhi << 32) | hlo;
            if (sub == 0) { wd[0] = 0u; wd[1] = 0u; }
            u32 mine = 0;
#pragma unroll
            for (u32 t = 0; t < 4 * NV; ++t) mine += (u32)__popc(wd[t]);
            u32 ltot;
            const u32 excl = group_excl<LPQ>(mine, sub, lane, ltot);
            const u64 hb = hdr - before;                              // ones of the block before this line
            const bool left = searching && (u64)rr <= hb;             // the rr-th one lies in an earlier line
            const bool right = searching && (u64)rr > hb + ltot;      // ... in a later one
            if (searching && !left && !right) {
                const u32 need0 = rr - (u32)hb;                       // 1..ltot inside this line
                if (need0 > excl && need0 <= excl + mine) {
                    u32 need = need0 - excl, word = 0, wi = 0; bool got = false;
#pragma unroll
                    for (u32 t = 0; t < 4 * NV; ++t) {
                        const u32 pc = (u32)__popc(wd[t]);
                        if (!got) { if (need <= pc) { word = wd[t]; wi = t; got = true; } else need -= pc; }
                    }
                    // line word index (sub * 4 NV + wi), data word = that - 2, bit of the line = 32 * data word + bit of the word
                    const u32 bit = j * RL_BITS + ((sub * 4u * NV + wi) - 2u) * 32u + select_in_word(word, need);
                    pos[qi] = ((u64)nb << 16) + bit;
                }
                searching = false;
            }
            if (left) j = j ? j - 1u : 0u;
            if (right) j = j + 1u < RL_LINES ? j + 1u : RL_LINES - 1u;
        }
        (void)result;
        if (live && sub == 0) {
            found[qi] = ok ? 1 : 0;
            if (!ok) pos[qi] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// select through a SELECT directory over the rank lines (round 3).  k_select_lines above still walks the block index first:
// sample search, <= 32 running counts, then two more table reads (running counts of the block, its octant directory) before
// the line -- three dependent round trips and a long instruction path per query (0.72 ms for 10 M queries where rank, one
// line per query, takes 0.21 ms).  The headers of the rank lines already ARE a sorted global sequence (ones before every
// line), so select is a search over lines, and the classic sampled-select directory starts it next to the answer:
//     sdir[m] = the line holding one number m S (0-based), m = 0 .. ceil(total / S) - 1;  sdir[last + 1] = the last line
// (S = the power of two next to 10 x the vector's average ones per line, so an entry spans ~10 lines whatever the density
// and the interpolation lands within a fraction of a line: 4 bytes per ~10 lines = 0.4 % of the lines, ~2 MB for a 4e9-bit
// vector: L2-resident).  A query reads TWO adjacent entries (the target line lies between them), guesses the line by
// interpolation, reads it, and lets the header decide: found / earlier / later.  A miss takes up to three secant steps
// (the header says how many ones away the answer is), then bisects between the bounds the headers have established, so
// skewed data costs O(log) lines instead of a walk.  Two dependent round trips (directory, line) when
// the guess holds -- and never a wrong answer: the headers decide, not the guess.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_rs_sdir(const u32* __restrict__ lines, u64 nlines, u64 total, u32 shift, u32* __restrict__ sdir, u64 nent /* incl. the sentinel */)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nlines) return;
    const u64 h = *reinterpret_cast<const u64*>(lines + j * 32u);
    const u64 hn = j + 1u < nlines ? *reinterpret_cast<const u64*>(lines + (j + 1u) * 32u) : total;
    const u64 S = 1ull << shift;
    for (u64 m = (h + S - 1u) >> shift; (m << shift) < hn; ++m) sdir[m] = (u32)j;      // ones h .. hn - 1 (0-based) live in line j
    if (j == nlines - 1u) sdir[nent - 1u] = (u32)j;
}

// the search the select kernels share: the target line lies in [lo, hi] (two adjacent directory entries 2^shift ones apart);
// line j is read and its header decides -- found / earlier / later; a miss takes up to three secant steps (the header says how
// many ones away the answer is; span = lines the entry's 2^shift ones are spread over), then bisects.  Iterations it0 ..
// max_it - 1 are run; `searching` says whether the query is still open afterwards (lo / hi / j then hold its state).
template <u32 LPQ>
__device__ __forceinline__ void select_load(const u32* __restrict__ lines, u32 j, u32 sub, u32x4 (&v)[8u / LPQ])
{
    constexpr u32 NV = 8u / LPQ;
    gcptr4 p = as_gc4(lines + (size_t)j * 32u) + sub * NV;
#pragma unroll
    for (u32 i = 0; i < NV; ++i) v[i] = p[i];
}

// one step of the search on the loaded line j: found (pos written, searching cleared) or the bounds and the next line updated
template <u32 LPQ>
__device__ __forceinline__ void select_eval(const u32x4 (&v)[8u / LPQ], u32& lo, u32& hi, u32& j, u64 span, u32 shift, u64 r,
                                            bool& searching, bool& failed, u32 it, u64 qi, u32 lane, u32 sub, u64* __restrict__ pos)
{
    constexpr u32 NV = 8u / LPQ;
    u32 wd[4 * NV];
#pragma unroll
    for (u32 i = 0; i < NV; ++i) { wd[4 * i] = v[i].x; wd[4 * i + 1] = v[i].y; wd[4 * i + 2] = v[i].z; wd[4 * i + 3] = v[i].w; }
    u32 hlo = sub == 0 ? wd[0] : 0u, hhi = sub == 0 ? wd[1] : 0u;
    hlo = group_first<LPQ>(hlo, lane); hhi = group_first<LPQ>(hhi, lane);
    const u64 hdr = ((u64)hhi << 32) | hlo;                   // ones of the vector before this line
    if (sub == 0) { wd[0] = 0u; wd[1] = 0u; }
    u32 mine = 0;
#pragma unroll
    for (u32 t = 0; t < 4 * NV; ++t) mine += (u32)__popc(wd[t]);
    u32 ltot;
    const u32 excl = group_excl<LPQ>(mine, sub, lane, ltot);
    const bool left = searching && r <= hdr;                  // the r-th one lies in an earlier line
    const bool right = searching && r > hdr + ltot;           // ... in a later one
    if (searching && !left && !right) {
        const u32 need0 = (u32)(r - hdr);                     // 1..ltot inside this line
        if (need0 > excl && need0 <= excl + mine) {
            u32 need = need0 - excl, word = 0, wi = 0; bool got = false;
#pragma unroll
            for (u32 t = 0; t < 4 * NV; ++t) {
                const u32 pc = (u32)__popc(wd[t]);
                if (!got) { if (need <= pc) { word = wd[t]; wi = t; got = true; } else need -= pc; }
            }
            const u32 nb = j / RL_LINES, lj = j - nb * RL_LINES;
            const u32 bit = lj * RL_BITS + ((sub * 4u * NV + wi) - 2u) * 32u + select_in_word(word, need);
            pos[qi] = ((u64)nb << 16) + bit;
        }
        searching = false;
    }
    if (left) hi = j - 1u;
    if (right) lo = j + 1u;
    if (left || right) {
        if (lo > hi) { searching = false; failed = true; }    // (cannot happen with consistent headers: reported as not found)
        else if (it < 3u) {
            const u64 away = left ? hdr - r : r - hdr - ltot - 1u;
            u64 step = 1u + ((away * span) >> shift);                 // away / (S / span) lines, at least the neighbour
            if (step > (u64)(hi - lo) + 1u) step = (u64)(hi - lo) + 1u;
            u32 nj = left ? (j >= step ? j - (u32)step : 0u) : j + (u32)step;
            j = nj < lo ? lo : (nj > hi ? hi : nj);
        }
        else j = lo + ((hi - lo) >> 1);
    }
}

template <u32 LPQ>
__device__ __forceinline__ void select_steps(const u32* __restrict__ lines, u32& lo, u32& hi, u32& j, u64 span, u32 shift, u64 r,
                                             bool& searching, bool& failed, u32 it0, u32 max_it, u64 qi, u32 lane, u32 sub, u64* __restrict__ pos)
{
    for (u32 it = it0; __ballot(searching) != 0ull && it < max_it; ++it) {
        u32x4 v[8u / LPQ];
        select_load<LPQ>(lines, j, sub, v);
        select_eval<LPQ>(v, lo, hi, j, span, shift, r, searching, failed, it, qi, lane, sub, pos);
    }
}

template <u32 LPQ>
__device__ __forceinline__ void select_walk(const u32* __restrict__ lines, u32 lo, u32 hi, u32 fr, u32 shift, u64 r, bool ok, bool live,
                                            u64 qi, u32 lane, u32 sub, u64* __restrict__ pos, u8* __restrict__ found)
{
    u32 j = lo + (u32)(((u64)(hi - lo) * fr) >> shift);
    const u64 span = (u64)(hi - lo) + 1u;
    bool searching = ok, failed = false;
    select_steps<LPQ>(lines, lo, hi, j, span, shift, r, searching, failed, 0u, 64u, qi, lane, sub, pos);
    failed = failed || searching;                                    // (iteration cap reached: same)
    if (live && sub == 0) {
        found[qi] = (ok && !failed) ? 1 : 0;
        if (!ok || failed) pos[qi] = 0;
    }
}

template <u32 LPQ>
__global__ __launch_bounds__(256)
void k_select_sdir(const u32* __restrict__ lines, const u32* __restrict__ sdir, u32 shift, u64 total,
                   const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    const u32 lane = lane_id();
    const u32 sub = lane & (LPQ - 1u);
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + (64u / LPQ) - 1ull) / (64u / LPQ) * (64u / LPQ);
    for (; qi < nq_round; qi += stride) {
        const bool live = qi < nq;
        const u64 r = live ? q[qi] : 0ull;
        const bool ok = live && r != 0ull && r <= total;
        const u64 idx0 = ok ? r - 1u : 0ull;
        const u64 m = idx0 >> shift;
        const u32 lo = sdir[m], hi = sdir[m + 1u];
        select_walk<LPQ>(lines, lo, hi, (u32)(idx0 & ((1ull << shift) - 1u)), shift, r, ok, live, qi, lane, sub, pos, found);
    }
}

// ---------------------------------------------------------------------------
// Round 5: select with the directory IN LDS.  k_select_sdir pays two dependent global reads per query -- the directory pair
// (1.7 MB for configs[3]: it competes with the lines for the L2 and misses about once per query) and the line: 2.0 lines per
// query, 0.45 ms for 10 M queries where rank takes 0.21.  A directory of 65,536 entries fits the 160 KiB of a CU: entry m = the
// line of one number m 2^shift (shift chosen so that the vector's ones give <= 65,535 entries: ~64 lines per entry for
// configs[3]) as a 16-bit offset from a 32-bit base per 256 entries -- 129 KiB, copied into LDS once per workgroup (one
// workgroup per CU, 33 MB for the whole launch).  How often the guess is a line off -- corrected in round 6 against the PMC of
// profiles/r05final (TCC_MISS 18.0 M per 10 M queries = 1.8 lines per query where rank misses 0.96) and a replay of the search on
// Bernoulli data (profiles/r06_select/README.md: 1.55 line reads per query at 10 %, 1.86 at 1 %): about every SECOND query, not one
// in eight.  An entry knows the LINE of its sampled one, not where in the line it sits (half a line of bias between the two ends of
// the interpolation), and between two samples ~85 lines apart the ones wander by ~0.35 lines at 10 % density, ~1.2 lines at 1 %.
// The kernel is bound by those second lines (it costs per missed line what rank costs), not by its instructions.  Vectors whose
// index holds select lines (bmx_kernels11.h: <= ~12 % ones) do not come here any more; this kernel serves the denser ones, where the
// wander is small (0.15 lines at 50 %).
// ---------------------------------------------------------------------------
#define STOP_ENTRIES 65536u
#define STOP_GROUP 64u                  // entries per 32-bit base
#define STOP_BASES (STOP_ENTRIES / STOP_GROUP)
#define STOP_BYTES (STOP_BASES * 4u + STOP_ENTRIES * 2u)
// Round 6: an entry is the POSITION of its sampled one to 1 / 2^fb of a line (fb <= 3), not just its line: with line-granular
// entries the two ends of the interpolation are half a line off on average and every second guess lands a line away
// (tools/sim_select_guess.py: 1.55 line reads per query at 10 % density, 1.50 at 50 %; with the position: 1.28 / 1.10).
// k_rs_stop_pos: P8[m] = 8 x line + (bit of the one inside the line's 960) / 120 for one number m << d_ones (0-based), found in the
// line the directory names; the sentinel entry: the end of the directory's last line.  k_rs_stop_range: the largest spread of a
// group of 64 entries (the host picks fb so that it fits 16 bits).  k_rs_stop_pack: base[g] + 16-bit offsets at that fb.
__global__ __launch_bounds__(256)
void k_rs_stop_pos(const u32* __restrict__ lines, const u32* __restrict__ sdir, u64 nent, u32 sh, u32 ssh, u32 n_top, u64 total, u32* __restrict__ P8)
{
    const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_top) return;
    const u64 k = (u64)m << ssh;                                   // the one (0-based) this entry samples
    if (k >= total) { P8[m] = sdir[nent - 1u] * 8u + 7u; return; } // the sentinel
    const u32 j = sdir[k >> sh];
    const u32* L = lines + (size_t)j * 32u;
    const u64 h = *reinterpret_cast<const u64*>(L);
    u32 need = (u32)(k - h) + 1u, bit = 0u;                        // the need-th one of the line's 30 data words
    for (u32 w = 2u; w < 32u; ++w) {
        const u32 x = L[w], pc = (u32)__popc(x);
        if (need <= pc) { bit = (w - 2u) * 32u + select_in_word(x, need); break; }
        need -= pc;
    }
    P8[m] = j * 8u + bit / 120u;
}
__global__ __launch_bounds__(256)
void k_rs_stop_range(const u32* __restrict__ P8, u32 n_top, u32* __restrict__ max_range)
{
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g * STOP_GROUP >= n_top) return;
    // (entry (g + 1) x 64 reads its offset against group g + 1's base, but hi of entry g x 64 + 63 needs nothing of group g)
    const u32 last = g * STOP_GROUP + STOP_GROUP - 1u < n_top ? g * STOP_GROUP + STOP_GROUP - 1u : n_top - 1u;
    atomicMax(max_range, P8[last] - P8[g * STOP_GROUP]);
}
__global__ __launch_bounds__(256)
void k_rs_stop_pack(const u32* __restrict__ P8, u32 n_top, u32 down /* 3 - fb */, u32* __restrict__ base, u16* __restrict__ t16)
{
    const u32 m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_top) return;
    const u32 b = P8[m & ~(STOP_GROUP - 1u)] >> down, v = P8[m] >> down;
    if ((m & (STOP_GROUP - 1u)) == 0u) base[m / STOP_GROUP] = b;
    t16[m] = (u16)(v - b);
}

template <u32 LPQ>
__global__ __launch_bounds__(1024)
void k_select_top(const u32* __restrict__ lines, const u32* __restrict__ stop /* base[1024] then t16[65536] */, u32 shift, u32 fb, u64 total,
                  const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    extern __shared__ u32 lds_dyn[];
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(stop);
        u32x4* dst = reinterpret_cast<u32x4*>(lds_dyn);
        for (u32 i = threadIdx.x; i < STOP_BYTES / 16u; i += 1024u) dst[i] = src[i];
    }
    __syncthreads();
    const u32* base = lds_dyn;
    const u16* t16 = reinterpret_cast<const u16*>(lds_dyn + STOP_BASES);
    const u32 lane = lane_id(), wave = uniform32(threadIdx.x >> 6);
    const u32 sub = lane & (LPQ - 1u), grp = lane / LPQ;
    constexpr u32 GPW = 64u / LPQ;                                    // queries a wave reads lines for at a time
    // A query whose first line was not the right one does not hold its wave for a second round trip (with 16 .. 32 queries
    // per wave nearly every wave would wait for somebody: the wave's rounds, not the queries' reads, would set the time): it
    // is parked -- {query, lo, hi, next line} -- in the wave's queue and the wave reads on; whenever the queue holds a
    // wave-load of them they take a round of their own.
    u32x4* Q = reinterpret_cast<u32x4*>(lds_dyn + STOP_BASES + STOP_ENTRIES / 2u) + wave * 64u;
    u32 qcount = 0u;
    // the two entries around one number r: their lines (lo, hi: the bounds of the search) and the interpolated first guess g
    auto entry_of = [&](u64 r, u32& lo, u32& hi, u32& g) {
        const u64 idx0 = r - 1u;
        const u32 m = (u32)(idx0 >> shift);
        const u32 plo = base[m / STOP_GROUP] + t16[m], phi = base[(m + 1u) / STOP_GROUP] + t16[m + 1u];
        lo = plo >> fb; hi = phi >> fb;
        const u64 fr = idx0 & ((1ull << shift) - 1u);
        // positions in half units: 2 plo + 1 (the middle of the entry's own unit) + the share of the distance
        const u64 p2 = 2ull * plo + 1ull + ((2ull * (u64)(phi - plo) * fr) >> shift);
        const u32 j = (u32)(p2 >> (fb + 1u));
        g = j < lo ? lo : (j > hi ? hi : j);
    };
    auto retry_round = [&](u32 first, u32 n) {                       // entries first .. first + n - 1 of the queue, to the end
        const bool have = grp < n;
        const u32x4 e = Q[first + (have ? grp : 0u)];
        const u64 qi = e.x;
        const u64 r = have ? q[qi] : 1ull;
        u32 lo0, hi0, g0;
        entry_of(r, lo0, hi0, g0);
        u32 lo = e.y, hi = e.z, j = e.w;
        bool searching = have, failed = false;
        select_steps<LPQ>(lines, lo, hi, j, (u64)(hi0 - lo0) + 1u, shift, r, searching, failed, 1u, 64u, qi, lane, sub, pos);
        failed = failed || searching;
        if (have && sub == 0) { found[qi] = failed ? 0 : 1; if (failed) pos[qi] = 0; }
    };
    u64 qi0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + GPW - 1ull) / GPW * GPW;
    constexpr u32 UN = 2u;                                            // queries per lane group whose lines are in flight together
    // the ranks are requested one iteration ahead: rank -> directory -> line is then ONE global round trip per iteration
    u64 rn[UN];
#pragma unroll
    for (u32 u = 0; u < UN; ++u) { const u64 qq = qi0 + u * stride; rn[u] = qq < nq ? __builtin_nontemporal_load(&q[qq]) : 0ull; }
    for (; qi0 < nq_round; qi0 += UN * stride) {
        u64 qi[UN], r[UN]; bool live[UN], ok[UN], searching[UN], failed[UN];
        u32 lo[UN], hi[UN], j[UN]; u64 span[UN];
        u32x4 v[UN][8u / LPQ];
#pragma unroll
        for (u32 u = 0; u < UN; ++u) {
            qi[u] = qi0 + u * stride;
            live[u] = qi[u] < nq;
            r[u] = rn[u];
            ok[u] = live[u] && r[u] != 0ull && r[u] <= total;
            entry_of(ok[u] ? r[u] : 1ull, lo[u], hi[u], j[u]);
            span[u] = (u64)(hi[u] - lo[u]) + 1u;
            searching[u] = ok[u]; failed[u] = false;
        }
#pragma unroll
        for (u32 u = 0; u < UN; ++u) { const u64 qq = qi0 + (UN + u) * stride; rn[u] = qq < nq ? __builtin_nontemporal_load(&q[qq]) : 0ull; }
#pragma unroll
        for (u32 u = 0; u < UN; ++u) select_load<LPQ>(lines, j[u], sub, v[u]);
#pragma unroll
        for (u32 u = 0; u < UN; ++u) {
            if (qi[u] >= nq_round) continue;                          // (wave-uniform)
            select_eval<LPQ>(v[u], lo[u], hi[u], j[u], span[u], shift, r[u], searching[u], failed[u], 0u, qi[u], lane, sub, pos);
            if (live[u] && sub == 0 && !searching[u]) {
                found[qi[u]] = (ok[u] && !failed[u]) ? 1 : 0;
                if (!ok[u] || failed[u]) pos[qi[u]] = 0;
            }
            // park the open ones
            const u64 open = __ballot(searching[u] && sub == 0);
            if (searching[u] && sub == 0) {
                const u32 at = qcount + (u32)__builtin_amdgcn_mbcnt_hi((u32)(open >> 32), __builtin_amdgcn_mbcnt_lo((u32)open, 0u));
                u32x4 e; e.x = (u32)qi[u]; e.y = lo[u]; e.z = hi[u]; e.w = j[u];
                Q[at] = e;
            }
            qcount += (u32)__popcll(open);
            if (qcount >= GPW) { qcount -= GPW; retry_round(qcount, GPW); }
        }
    }
    if (qcount) retry_round(0u, qcount);
}
