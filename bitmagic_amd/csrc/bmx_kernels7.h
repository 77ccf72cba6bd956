// bmx_kernels7.h -- round 4: combine_or over thousands of SPARSE GAP-only operands straight from the vectors' own slabs
// (aggregator::combine_or, src/bmaggregator.h:1101-1121; per column sort_input_blocks_or :2278 + process_gap_blocks_or
// :1907 -> gap_add_to_bitset src/bmfunc.h:4684), i.e. the FIRST aggregation over an operand list -- no packed copy, any list.
#pragma once
#include "bmx_kernels6.h"

// ---------------------------------------------------------------------------
// Why a second descriptor-table kernel.  Round 2 concluded that k_agg_or_gap_tiled (bmx_kernels2.h) sat on the floor of its
// access pattern -- 4096 slabs visited in 1-KiB pieces -- because the kernel with the run application compiled out still
// took 3.78 of 4.1 ms.  Round 4 measured the pattern by itself (tools/probes/pieces_probe.hip, tiled_probe.hip,
// profiles/r04a, r04b): 4096 separately allocated streams read in 1-KiB pieces by one 1024-thread workgroup per CU stream
// at 6.5-6.7 TB/s when every piece is ONE coalesced 16-B-per-lane wave load and four of them are in flight per wave; the
// lane-per-block shape of the tiled kernel (4 x 16 B per lane at a 64-B stride) reaches 6.1 TB/s, and what the real kernel
// adds on top is 2 GB of 8-byte descriptors (one per operand and column) read through a three-deep dependent chain.
// So the pattern was never the floor; the descriptor traffic, the request shape and the per-lane chain were.
//
// This kernel works on tiles of ORR_TILE = 14 block columns (14 accumulators = 112 KiB of LDS, one 1024-thread workgroup
// per CU) and changes how an operand's piece arrives:
//  * TILE DIRECTORY (k_build_tdir, built once when a vector is created: 16 B per 14 blocks next to the 8 B per block of
//    the descriptor table): where the GAP data of the tile's blocks starts in the vector's slab (the device slab keeps GAP
//    blocks in block order, back to back on 16-byte boundaries), how many 16-byte chunks it spans (n <= 64), which of the
//    columns hold GAP blocks and a 64-bit mask M of the chunks that START a block.  (14 columns, not 16: configs[4]
//    averages 3.9 chunks per block, so 16 blocks pass 64 chunks a third of the time and 14 blocks 0.3 % of the time.)
//  * ROWS: the piece is read as ONE wave load (lane L takes chunk L, 16 B), four or eight rows (= operands) in flight per
//    wave.  Because blocks are 16-byte aligned a chunk belongs to exactly one block: lane L's block is the popcount of M
//    at or below L -- no header, no length: the slab's padding words are 0xFFFF (the creation paths see to it), a value
//    no run end but a block's last one has, so a word pair whose first half is 0xFFFF is no run; blocks that start with a
//    1-run (bit 0 of the block set: one block in 5,000 at configs[4]'s density) hand their tile to the descriptor path.
//    The directory also says whether ANY 1-run of the tile is longer than one bit (TREC_LONG; the builder reads the run
//    lists once): a tile without one -- 96 % of configs[4]'s -- is applied with six vector instructions per run and no
//    test but "is this pair a run".  Every lane applies the four 1-run slots of its chunk: a single-bit run (nearly all of them in a sparse
//    operand) is one ds_or without control flow, anything else takes ONE shared branch per chunk -- all 64 lanes busy
//    whatever the block lengths, ~50 vector instructions per KiB of run lists.
//  * RECORDS: a wave owns every 16th operand; the directory records of 64 of its operands are fetched by ONE gather (lane j
//    = operand j of the batch), one batch ahead, and handed to the row loop through v_readlane: no per-row dependent
//    chain, 16 B instead of 128 B of descriptors per (operand, tile).
// Tiles the directory cannot describe (more than 64 chunks, a FULL block, a block starting with a 1-run, GAP data not contiguous) carry a flag and are
// applied from the descriptor table by 14 lanes after the batch (the old way, out of the hot loop): correct for any input, fast for the sparse ones this
// kernel exists for.  The host picks this kernel when the operands average <= 4.1 chunks per GAP block (bmx.hip).
// ---------------------------------------------------------------------------

#define ORR_TILE 14u
#define TREC_SLOW 0x80u
#define TREC_NCH(i) ((i) & 0x7Fu)
#define TREC_GAPMASK(i) (((i) >> 8) & 0x3FFFu)
#define TREC_ALLGAP 0x400000u                 // every column of the tile holds a GAP block: a block's ordinal is its column
#define TREC_LONG   0x800000u                 // some 1-run of the tile is longer than one bit

// tile directory of one vector: one thread per tile of ORR_TILE block columns; entry (16 B) =
//   {first chunk of the tile in the slab, n chunks | TREC_SLOW | GAP columns << 8 | TREC_ALLGAP, M lo, M hi}
// M = chunks that start a block
// bcnt (round 5, may be null): per block, the 1-runs of the block as multi-bit | single-bit << 16 (what a split collection keeps
// apart, bmx_kernels6.h) -- 0 for NULL / FULL blocks, 0xFFFFFFFF for the GAP blocks of a tile the directory does not describe.
// The builder reads the run lists anyway (TREC_LONG): the counts let bmx_collection_prepare size the column regions of a
// collection without a pass over the operands' run lists (bmx_kernels10.h).
__global__ __launch_bounds__(256)
void k_build_tdir(const u64* __restrict__ desc, u32 nblocks, u64 gaps_base, u32x4* __restrict__ tdir, u32 ntiles, u32* __restrict__ bcnt)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    u64 next = 0, first = 0, M = 0;
    bool have = false, slow = false;
    u32 nch = 0, gapmask = 0;
    u32 ch_of[ORR_TILE];
    for (u32 k = 0; k < ORR_TILE; ++k) {
        const u32 c = t * ORR_TILE + k;
        const u64 d = c < nblocks ? desc[c] : 0ull;
        const u32 kind = DESC_K(d);
        ch_of[k] = 0u;
        if (kind == K_GAP) {
            const u64 a = DESC_P(d);
            const u32 meta = GMETA(d);
            const u32 ch = ((meta >> 1) + 1u + 7u) >> 3;          // header + len run ends, in 16-byte chunks
            if (!have) { first = a; have = true; }
            else if (a != next) slow = true;                      // not back to back: the row load would read something else
            if (meta & 1u) slow = true;                           // starts with a 1-run: the row code pairs words as (0-run end, 1-run end)
            if (nch < 64u) M |= 1ull << nch;
            gapmask |= 1u << k;
            ch_of[k] = ch;
            nch += ch; next = a + (u64)ch * 16u;
        } else if (kind != K_NULL) slow = true;                   // FULL (or a bit-block): the descriptor path takes the tile
    }
    if (nch > 64u || ((first - gaps_base) & 15ull) || ((first - gaps_base) >> 36)) slow = true;
    // any 1-run longer than one bit?  The same word pairs the row kernel forms: (word 2i + 1, word 2i + 2) of every chunk, the
    // last one reaching into the next chunk (padding words are 0xFFFF by now: the writers of the slab ran before this kernel)
    bool any_long = false;
    if (have && !slow) {
        gcptr4 g = (gcptr4)(uintptr_t)first;
        u32x4 cur = g[0];
        u32 q = 0u;
        for (u32 k = 0; k < ORR_TILE; ++k) {
            u32 m = 0u, sg = 0u;
            for (u32 j = 0; j < ch_of[k]; ++j, ++q) {
                const u32x4 nxt = q + 1u < nch ? g[q + 1u] : (u32x4)(0xFFFFFFFFu);
                const u32 x[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 p = x[i] >> 16, e = x[i + 1] & 0xFFFFu;
                    const bool run = p != 0xFFFFu, one = e - p == 1u;
                    m += (run && !one) ? 1u : 0u; sg += (run && one) ? 1u : 0u;
                }
                cur = nxt;
            }
            any_long = any_long || m != 0u;
            if (bcnt && t * ORR_TILE + k < nblocks) bcnt[t * ORR_TILE + k] = m | (sg << 16);
        }
    } else if (bcnt) {
        for (u32 k = 0; k < ORR_TILE; ++k)
            if (t * ORR_TILE + k < nblocks) bcnt[t * ORR_TILE + k] = ((gapmask >> k) & 1u) ? 0xFFFFFFFFu : 0u;
    }
    u32x4 r;
    r.x = (have && !slow) ? (u32)((first - gaps_base) >> 4) : 0u;
    r.y = (slow ? TREC_SLOW : nch) | (gapmask << 8) | (gapmask == (1u << ORR_TILE) - 1u ? TREC_ALLGAP : 0u) | (any_long ? TREC_LONG : 0u);
    r.z = slow ? 0u : (u32)M; r.w = slow ? 0u : (u32)(M >> 32);
    tdir[t] = r;
}

// position of the r-th (0-based) set bit of a 16-bit mask
__device__ __forceinline__ u32 nth_set_bit16(u32 m, u32 r)
{
    u32 pos = 0u, c;
    c = (u32)__popc(m & 0xFFu); { const bool g = r >= c; r -= g ? c : 0u; pos += g ? 8u : 0u; m = g ? m >> 8 : m; }
    c = (u32)__popc(m & 0xFu);  { const bool g = r >= c; r -= g ? c : 0u; pos += g ? 4u : 0u; m = g ? m >> 4 : m; }
    c = (u32)__popc(m & 0x3u);  { const bool g = r >= c; r -= g ? c : 0u; pos += g ? 2u : 0u; m = g ? m >> 2 : m; }
    c = m & 1u;                 { const bool g = r >= c; pos += g ? 1u : 0u; }
    return pos;
}

// records of one batch of 64 operands of a wave: lane j holds operand j's.  A device address has 48 bits: the upper half of
// ahi carries the row's chunk count (bits 16..22), the all-GAP flag (bit 23) and the long-run flag (bit 24), so that the row loop reads four words
// (alo, ahi, mlo, mhi) per row and `info` only for a tile with NULL columns
struct OrRec { u32 alo, ahi, mlo, mhi, info, dlo, dhi, nblk; };
#define OREC_NCH(ahi) (((ahi) >> 16) & 0x7Fu)
#define OREC_ALLGAP(ahi) ((ahi) & 0x800000u)
#define OREC_LONG(ahi) ((ahi) & 0x1000000u)

// operand table entry (host-built, 32 B): tile directory, GAP slab, descriptor table, blocks
//   e0 = {tdir lo, tdir hi, gaps lo, gaps hi}, e1 = {desc lo, desc hi, nblocks, 0}
__device__ __forceinline__ void or_rec_fetch_a(u32x4& e0, u32x4& e1, gcptr4 optab, u32 op, u32 n)
{
    const u32 o = op < n ? op : n - 1u;
    e0 = optab[2u * o]; e1 = optab[2u * o + 1u];
}
__device__ __forceinline__ bool or_rec_ok(const u32x4& e0, const u32x4& e1, u32 op, u32 n, u32 tile)
{
    return op < n && (e0.x | e0.y) != 0u && tile < (e1.z + ORR_TILE - 1u) / ORR_TILE;
}
__device__ __forceinline__ u32x4 or_rec_fetch_b(const u32x4& e0, const u32x4& e1, u32 op, u32 n, u32 tile, gcptr4 dummy)
{
    const u64 td = (u64)e0.x | ((u64)e0.y << 32);
    gcptr4 p = or_rec_ok(e0, e1, op, n, tile) ? (gcptr4)(uintptr_t)td + tile : dummy;
    return *p;
}
// an empty (or slow) row still loads something: its address is the operand table's
__device__ __forceinline__ void or_rec_make(OrRec& r, const u32x4& e0, const u32x4& e1, const u32x4& t, u32 op, u32 n, u32 tile, u64 dummy)
{
    const bool ok = or_rec_ok(e0, e1, op, n, tile);
    const u32 info = ok ? t.y : 0u;
    const u32 nch = TREC_NCH(info);
    const u64 a = nch ? ((u64)e0.z | ((u64)e0.w << 32)) + ((u64)t.x << 4) : dummy;
    r.alo = (u32)a; r.ahi = ((u32)(a >> 32) & 0xFFFFu) | (nch << 16) | ((info & TREC_ALLGAP) ? 0x800000u : 0u) | ((info & TREC_LONG) ? 0x1000000u : 0u);
    r.info = info; r.mlo = ok ? t.z : 0u; r.mhi = ok ? t.w : 0u;
    r.dlo = e1.x; r.dhi = e1.y; r.nblk = ok ? e1.z : 0u;
}

// bits s..e (inclusive, s <= e) of one block into its accumulator; more than one word: edge words by atomics, interior by stores
// (an OR accumulator only gains bits, so a plain store of all-ones cannot lose anything another lane adds)
__device__ __forceinline__ void or_set_range(u32* acc, u32 s, u32 e)
{
    const u32 wl = s >> 5, wr = e >> 5;
    const u32 lo = ~0u << (s & 31u), hi = ~0u >> (31u - (e & 31u));
    if (wl == wr) atomicOr(&acc[wl], lo & hi);
    else {
        atomicOr(&acc[wl], lo);
        atomicOr(&acc[wr], hi);
        for (u32 w = wl + 1u; w < wr; ++w) acc[w] = ~0u;
    }
}

// one row = the GAP data of one operand's tile, lane L holding its L-th 16-byte chunk (lanes past the last chunk hold a
// copy of it and stay out).  Words of the chunk: block words k = 8q .. 8q + 7 (k = 0: the header; run k ends at word k; the
// blocks here start with a 0-run, so run k is a 1-run for even k, src/bmfunc.h:4684).  The four 1-run slots of the chunk
// are the word pairs (end of the 0-run before, end of the 1-run) = words (2i + 1, 2i + 2): lo16 = previous end, hi16 =
// end.  A pair is a run iff its lo16 is not 0xFFFF (the last word of a block and every padding word).
__device__ __forceinline__ void or_row_apply(const u32x4& c, u32 nx, u32 nch, bool has_long, u32* acc, u32 lane)
{
    if (lane < nch) {
        const u32 x[5] = {c.x, c.y, c.z, c.w, nx};
        if (!has_long) {                                                   // (wave-uniform) every run of the tile is one bit
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 y = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);
                const u32 e = y >> 16;
                atomicOr(&acc[e >> 5], (y & 0xFFFFu) != 0xFFFFu ? 1u << (e & 31u) : 0u);
            }
        } else {
            bool rare = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 y = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);
                const u32 e = y >> 16, p = y & 0xFFFFu;
                const bool single = e - p == 1u;                           // the run is one bit
                atomicOr(&acc[e >> 5], single ? 1u << (e & 31u) : 0u);
                rare = rare || (!single && p != 0xFFFFu);
            }
            if (rare) {                                                    // runs longer than one bit
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 y = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);
                    const u32 e = y >> 16, p = y & 0xFFFFu;
                    if (p != 0xFFFFu && e - p != 1u) or_set_range(acc, p + 1u, e);
                }
            }
        }
    }
}

// a tile the directory cannot describe: ORR_TILE lanes, a block each, through the descriptor table (as k_agg_or_gap_tiled does)
__device__ __noinline__ void or_row_slow(u64 desc_tab, u32 nblk, u32 c0, u32* accs, u32* full, u32 lane)
{
    if (lane < ORR_TILE) {
        const u32 col = c0 + lane;
        const u64 d = col < nblk ? ((const __attribute__((address_space(1))) u64*)(uintptr_t)desc_tab)[col] : 0ull;
        const u32 k = DESC_K(d);
        if (k == K_FULL) full[lane] = 1u;
        if (k == K_GAP) {
            GapHead h;
            gap_head_fetch(h, DESC_P(d), true);
            gap_or_lane_fast(h, DESC_P(d), accs + lane * 2048u);
        }
    }
}

template <int DEPTH>
__global__ __launch_bounds__(1024)
void k_agg_or_rows(const u32x4* __restrict__ optab_, u32 n, u32 ncols, int opt_compress, int xcd_swz,
                   uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, FoldOut kinds, FoldOut total)
{
    extern __shared__ u32 lds_dyn[];                 // ORR_TILE x 2048 u32 accumulators + 16 flags
    u32* full = lds_dyn + ORR_TILE * 2048u;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 tile = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 c0 = tile * ORR_TILE;
    gcptr4 optab = as_gc4((const void*)optab_);
    const u64 dummy = (u64)(uintptr_t)optab_;
    {
        u32x4* l4 = reinterpret_cast<u32x4*>(lds_dyn);
#pragma unroll
        for (int i = 0; i < 7; ++i) l4[i * 1024 + tid] = (u32x4)(0u);      // 14 x 8 KiB = 7 x 1024 x 16 B
        if (tid < 16u) full[tid] = 0u;
    }
    __syncthreads();
    const u32 le_lo = lane >= 31u ? ~0u : (2u << lane) - 1u;
    const u32 le_hi = lane < 32u ? 0u : (lane == 63u ? ~0u : (2u << (lane - 32u)) - 1u);
    // operands of this wave: wave + 16 * (64 * batch + j), j = the lane that keeps the record
    const u32 nrows = wave < n ? (n - wave + 15u) / 16u : 0u;
    const u32 nbatch = (nrows + 63u) / 64u;
    auto op_of = [&](u32 b) { return wave + 16u * (64u * b + lane); };
    u32x4 a0, a1, b0, b1, t;
    OrRec cur, nxt;
    or_rec_fetch_a(a0, a1, optab, op_of(0u), n);
    or_rec_fetch_a(b0, b1, optab, op_of(1u), n);
    t = or_rec_fetch_b(a0, a1, op_of(0u), n, tile, optab);
    or_rec_make(cur, a0, a1, t, op_of(0u), n, tile, dummy);
    u32x4 c[DEPTH];
    const u32 lane16 = lane << 4;
    // a row's load: lanes beyond the row's chunks repeat its last chunk (same address: no extra traffic), an empty row reads
    // the operand table -- every load is issued unconditionally, so the compiler's vmcnt bookkeeping stays exact
    auto issue = [&](u32x4& dst, const OrRec& rec, u32 j) {
        const u32 ahi = (u32)__builtin_amdgcn_readlane((int)rec.ahi, (int)j);
        const u32 nch = OREC_NCH(ahi);
        const u64 a = (u64)(u32)__builtin_amdgcn_readlane((int)rec.alo, (int)j) | ((u64)(ahi & 0xFFFFu) << 32);
        const u32 last16 = nch ? (nch - 1u) << 4 : 0u;
        dst = *(gcptr4)(uintptr_t)(a + (lane16 < last16 ? lane16 : last16));
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) issue(c[k], cur, (u32)k);
    for (u32 b = 0; b < nbatch; ++b) {
        // the next batch's records (their table entries were requested a batch ago), the table entries of the one after
        t = or_rec_fetch_b(b0, b1, op_of(b + 1u), n, tile, optab);
        u32x4 n0, n1;
        or_rec_fetch_a(n0, n1, optab, op_of(b + 2u), n);
        for (u32 j = 0; j < 64u; j += DEPTH) {
            if (j + DEPTH == 64u) or_rec_make(nxt, b0, b1, t, op_of(b + 1u), n, tile, dummy);   // (uniform) first needed by the look-ahead below
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                const u32 jj = j + (u32)k;
                const u32 ahi = (u32)__builtin_amdgcn_readlane((int)cur.ahi, (int)jj);
                const u32 mlo = (u32)__builtin_amdgcn_readlane((int)cur.mlo, (int)jj), mhi = (u32)__builtin_amdgcn_readlane((int)cur.mhi, (int)jj);
                // the next chunk's first dword (lane 63: all ones, i.e. "no run")
                const u32 nx = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)c[k].x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                u32 col = (u32)__popc(mlo & le_lo) + (u32)__popc(mhi & le_hi) - 1u;      // ordinal of this lane's block among the tile's GAP blocks
                if (!OREC_ALLGAP(ahi))                                                   // (wave-uniform) NULL columns in the tile
                    col = nth_set_bit16(TREC_GAPMASK((u32)__builtin_amdgcn_readlane((int)cur.info, (int)jj)), col) & 15u;
#ifdef BMX_DIAG
                if (opt_compress & 512) { if ((c[k].x ^ c[k].y ^ c[k].z ^ c[k].w ^ nx ^ col) == 0x12345679u) lds_dyn[lane] = 1u; }   // timing probe: the loads alone
                else
#endif
                or_row_apply(c[k], nx, OREC_NCH(ahi), OREC_LONG(ahi) != 0u, lds_dyn + col * 2048u, lane);
                if (j + DEPTH < 64u) issue(c[k], cur, jj + DEPTH);
                else issue(c[k], nxt, (u32)k);
            }
        }
        // tiles the directory handed back (rare): from the descriptor table, one at a time
        u64 slow = __ballot((cur.info & TREC_SLOW) != 0u);
        while (slow) {
            const u32 jj = (u32)__builtin_ctzll(slow);
            slow &= slow - 1ull;
            const u64 dt = (u64)(u32)__builtin_amdgcn_readlane((int)cur.dlo, (int)jj) | ((u64)(u32)__builtin_amdgcn_readlane((int)cur.dhi, (int)jj) << 32);
            or_row_slow(dt, (u32)__builtin_amdgcn_readlane((int)cur.nblk, (int)jj), c0, lds_dyn, full, lane);
        }
        cur = nxt; b0 = n0; b1 = n1;
    }
    __syncthreads();
    // one wave per column of the tile: classify + store (opt_copy_bit_block rule), kinds and the result's count folded here
    u32 kind = 4u, pop = 0u;
    {
        const u32 tc = wave, cc = c0 + tc;
        if (tc < ORR_TILE && cc < ncols) {
            if (full[tc]) { store_trivial(K_FULL, cc, desc, st, lane); kind = K_FULL; pop = 65536u; }
#ifdef BMX_DIAG
            else if (opt_compress & 2048) { store_trivial(K_NULL, cc, desc, st, lane); kind = K_NULL; }   // timing probe: nothing classified or stored
#endif
            else {
                Blk bk;
                blk_from_lds(bk, lds_dyn + tc * 2048u, lane);
                kind = store_result_mode(bk, cc, (opt_compress & 1) ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);
                pop = uniform32(wave_sum(blk_lane_popcount(bk)));
            }
        }
    }
#ifdef BMX_DIAG
    if (opt_compress & 1024) return;                                         // timing probe: no folds
#endif
    kind_fanin_fold(kind, kinds, lane, wave);
    count_fanin_fold(pop, total, lane, wave);
}
