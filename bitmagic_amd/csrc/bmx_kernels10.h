// bmx_kernels10.h -- round 5: building a packed collection of SPARSE operands (polarity 1, split bags: the OR / SUB role of
// BASELINE configs[4]) through the vectors' tile directories -- the transposition of aggregator::combine_or's operand set
// (src/bmaggregator.h:1101-1121; per column sort_input_blocks_or :2278) into the column-major run lists of bmx_kernels6.h.
#pragma once
#include "bmx_kernels9.h"

// ---------------------------------------------------------------------------
// Why a second build.  The round-3 / 4 passes (k_coll_count, k_coll_count_singles, k_coll_prefix_singles, k_coll_scatter_split,
// k_coll_dir: bmx_kernels6.h / 8.h) are launched over (operand, column tile): 29.4 ms for configs[4] (14.0 GB read twice +
// 6.5 GB written = 0.09 of the roofline).  Three things held them there:
//   * 8 lanes walk one ~56-byte block after a dependent descriptor read, one block per lane group in flight: latency, not
//     bandwidth (k_coll_count_singles: 14 GB in 7.7 ms);
//   * the pieces of a column written by operand i and operand i + 1 come from different workgroups -- neighbours in the grid,
//     i.e. on different XCDs with different L2s -- so every 128-byte line of the output is written in ~30-byte pieces that no
//     cache can merge (k_coll_scatter_split: 16.5 ms);
//   * 2 GB of per-(operand, column) prefixes go to memory and come back (pre, sgl), and are transposed again into the member
//     directory.
// Here a WORKGROUP owns a tile of 14 block columns for ALL operands, reads an operand's piece of the tile as one coalesced
// row through its tile directory (exactly as k_agg_or_rows does, bmx_kernels7.h) and writes every column's region front to
// back itself: the prefixes never leave the chip, and the lines of the output are completed by one workgroup within
// microseconds (the L2 merges them).  Two passes over the run lists remain -- the sizes of the column regions must be known
// before the first run can be placed:
//   k_coll2_count    per (tile, group of 64 operands, column): multi-bit and single-bit 1-runs -> bt; per column: cnt, cnt_s,
//                    flags, size in words
//   k_coll_offsets   (bmx_kernels6.h) column offsets
//   k_coll2_scatter  the runs into their places + the member directory dir[c][i], dir_s[c][i]
// Tiles a directory cannot describe (more than 64 chunks, a FULL block, a block that starts with a 1-run) are walked from the
// descriptor table by 14 lanes, a block each -- correct for any operand, fast for the sparse ones this path is chosen for.
// The layout produced is the one bmx_kernels6.h documents (member order inside a column, multi-bit runs then single-bit
// positions); inside a member's piece the runs are in block order.
// ---------------------------------------------------------------------------

#define C2_GROUP 64u                 // operands per group: a wave's batch of records (lane = operand)
#define C2_MAX_N 16384u              // operands per collection through this path (the group prefixes of a tile live in LDS)

// 1-runs of one row chunk: bit i of `multi` / `single` = pair i of the chunk is a run longer than one bit / of one bit;
// y[i] = lo16: the position BEFORE the run, hi16: its last position (the word pairs or_row_apply forms, bmx_kernels7.h)
__device__ __forceinline__ void c2_classify(const u32x4& c, u32 nx, u32& multi, u32& single, u32 (&y)[4])
{
    const u32 x[5] = {c.x, c.y, c.z, c.w, nx};
    multi = 0u; single = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        y[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);
        const u32 p = y[i] & 0xFFFFu, e = y[i] >> 16;
        const bool run = p != 0xFFFFu;
        const bool one = e - p == 1u;
        multi |= (run && !one) ? 1u << i : 0u;
        single |= (run && one) ? 1u << i : 0u;
    }
}

// column (0 .. 13) of this lane's chunk: ordinal of its block among the tile's GAP blocks -> column through the GAP mask
__device__ __forceinline__ u32 c2_col_of_lane(u32 mlo, u32 mhi, u32 le_lo, u32 le_hi, u32 ahi, u32 info)
{
    u32 col = (u32)__popc(mlo & le_lo) + (u32)__popc(mhi & le_hi) - 1u;
    if (!OREC_ALLGAP(ahi)) col = nth_set_bit16(TREC_GAPMASK(info), col);
    return col & 15u;
}

// a block walked by ONE lane (tiles the directory hands back): f(start, end) for every 1-run
template <class F>
__device__ __forceinline__ void c2_walk_block(u64 d, F&& f)
{
    const u32 meta = GMETA(d), len = meta >> 1, s = meta & 1u;
    const u32 m_cnt = coll_runs_of(meta, 1u);
    const u32 nchunks = (len + 1u + 7u) >> 3;
    gcptr4 g4 = as_gc4(DESC_P(d));
    for (u32 q = 0; q < nchunks; ++q) {
        const u32x4 x = g4[q];
        const u32 nx = q + 1u < nchunks ? g4[q + 1u].x : 0xFFFFFFFFu;
        CollPairs p;
        coll_chunk_pairs(x, nx, s, q, p);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (4u * q + (u32)k < m_cnt) f(p.start[k], p.end[k]);
    }
}

// the descriptor of column c0 + lane of a row the directory handed back (lanes >= 14 / columns beyond the operand: NULL)
__device__ __forceinline__ u64 c2_slow_desc(u64 desc_tab, u32 nblk, u32 c0, u32 lane)
{
    const u32 col = c0 + lane;
    return (lane < ORR_TILE && col < nblk) ? ((const __attribute__((address_space(1))) u64*)(uintptr_t)desc_tab)[col] : 0ull;
}

struct C2CountOut { u32* cnt; u32* cnt_s; u32* flags; u32* words; u32* bt; };

// a row's load: lanes beyond the row's chunks repeat its last chunk, an empty / handed-back row reads the operand table --
// every load is issued unconditionally (exact vmcnt bookkeeping, as in k_agg_or_rows)
__device__ __forceinline__ u32x4 c2_row_load(const OrRec& rec, u32 j, u32 lane16, u64 dummy)
{
    const bool in = j < C2_GROUP;                                       // (the look-ahead past a group's last row reads the operand table)
    const u32 ahi = (u32)__builtin_amdgcn_readlane((int)rec.ahi, (int)(j & 63u));
    const u32 nch = in ? OREC_NCH(ahi) : 0u;
    const u64 a = in ? ((u64)(u32)__builtin_amdgcn_readlane((int)rec.alo, (int)(j & 63u)) | ((u64)(ahi & 0xFFFFu) << 32)) : dummy;
    const u32 last16 = nch ? (nch - 1u) << 4 : 0u;
    return *(gcptr4)(uintptr_t)(a + (lane16 < last16 ? lane16 : last16));
}

// pass 1.  grid = tiles of 14 columns; bt[(tile * ngroups + G) * 16 + col] = multis | singles << 16 of group G's operands.
// The counts come from the per-block figures behind every vector's tile directory (k_build_tdir, bmx_kernels7.h): 4 B per
// (operand, column) instead of the run lists themselves (the first version of this pass read all of them: 14 GB, 3.1 ms for
// configs[4]).  Blocks of tiles the directory does not describe carry no figure and are walked from the descriptor table.
__global__ __launch_bounds__(1024)
void k_coll2_count(const u32x4* __restrict__ optab_, u32 n, u32 ncols, u32 ngroups, int xcd_swz, C2CountOut o)
{
    __shared__ u32 T[16][16];
    __shared__ u32 WM[16], WS[16], WF[16], WN[16];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = uniform32(tid >> 6);
    const u32 tile = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 c0 = tile * ORR_TILE;
    gcptr4 optab = as_gc4((const void*)optab_);
    const u64 dummy = (u64)(uintptr_t)optab_;
    if (tid < 16u) { WM[tid] = 0u; WS[tid] = 0u; WF[tid] = 0u; WN[tid] = 0u; }
    if (lane < 16u) T[wave][lane] = 0u;
    __syncthreads();
    u32 fl = 0u, ngap = 0u;                                              // lanes 0 .. 13: flags / GAP operands of column c0 + lane
    for (u32 G = wave; G < ngroups; G += 16u) {
        const u32 op = G * C2_GROUP + lane;
        u32x4 e0, e1;
        or_rec_fetch_a(e0, e1, optab, op, n);
        const u32x4 t = or_rec_fetch_b(e0, e1, op, n, tile, optab);
        OrRec rec;
        or_rec_make(rec, e0, e1, t, op, n, tile, dummy);
        const bool member = op < n;
        const bool fastrow = member && (rec.info & TREC_SLOW) == 0u;
        u64 slow_m = __ballot(member && (rec.info & TREC_SLOW) != 0u);
        // the blocks' counts: behind the operand's directory of (nblocks + 13) / 14 tiles.  Lane k (< 14) reads column k of one
        // operand after the other -- 56 contiguous bytes, one cache line per operand -- eight operands in flight
        const u64 td = (u64)e0.x | ((u64)e0.y << 32);
        const u32 nblk_l = e1.z;
        const bool have = fastrow && td != 0ull && tile < (nblk_l + ORR_TILE - 1u) / ORR_TILE;
        const u64 bc_l = have ? td + (u64)((nblk_l + ORR_TILE - 1u) / ORR_TILE) * 16u + (u64)c0 * 4u : 0ull;      // (0: nothing to read for this operand)
        const u32 bclo = (u32)bc_l, bchi = (u32)(bc_l >> 32);
        u32 acc_m = 0u, acc_s = 0u;
        for (u32 j = 0; j < C2_GROUP; j += 8u) {
            u32 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u32 jj = j + (u32)q;
                const u64 a = (u64)(u32)__builtin_amdgcn_readlane((int)bclo, (int)jj) | ((u64)(u32)__builtin_amdgcn_readlane((int)bchi, (int)jj) << 32);
                const u32 gm = TREC_GAPMASK((u32)__builtin_amdgcn_readlane((int)rec.info, (int)jj));
                const bool rd = a != 0ull && lane < ORR_TILE && ((gm >> lane) & 1u) != 0u;
                v[q] = *(const __attribute__((address_space(1))) u32*)(uintptr_t)(rd ? a + (u64)lane * 4u : dummy);
                if (!rd) v[q] = 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc_m += v[q] & 0xFFFFu; acc_s += v[q] >> 16; }
        }
        if (lane < ORR_TILE) T[wave][lane] += acc_m | (acc_s << 16);
        // block kinds of the rows the directory describes: GAP where its mask says so, NULL elsewhere (a FULL block makes a
        // tile slow): a ballot per column over the 64 operands of the batch, lane k keeps column k
#pragma unroll
        for (u32 k = 0; k < ORR_TILE; ++k) {
            const bool gbit = ((rec.info >> (8u + k)) & 1u) != 0u;
            const u64 g = __ballot(fastrow && gbit), z = __ballot(fastrow && !gbit);
            if (lane == k) { ngap += (u32)__popcll(g); fl |= z ? COLL_FLAG_NULL : 0u; }
        }
        // rows the directory handed back: a lane per column through the descriptor table
        while (slow_m) {
            const u32 jj = (u32)__builtin_ctzll(slow_m);
            slow_m &= slow_m - 1ull;
            const u64 dt = (u64)(u32)__builtin_amdgcn_readlane((int)rec.dlo, (int)jj) | ((u64)(u32)__builtin_amdgcn_readlane((int)rec.dhi, (int)jj) << 32);
            const u32 nblk = (u32)__builtin_amdgcn_readlane((int)rec.nblk, (int)jj);
            if (lane < ORR_TILE) {
                const u64 d = c2_slow_desc(dt, nblk, c0, lane);
                const u32 kd = DESC_K(d);
                if (kd == K_GAP) {
                    u32 m = 0u, s = 0u;
                    c2_walk_block(d, [&](u32 st, u32 en) { if (st == en) ++s; else ++m; });
                    T[wave][lane] += m | (s << 16);
                    ++ngap;
                } else fl |= kd == K_FULL ? COLL_FLAG_FULL : (kd == K_NULL ? COLL_FLAG_NULL : COLL_FLAG_BIT);
            }
        }
        if (lane < 16u) {
            const u32 tot = T[wave][lane];
            o.bt[((size_t)tile * ngroups + G) * 16u + lane] = tot;
            if (tot) { atomicAdd(&WM[lane], tot & 0xFFFFu); atomicAdd(&WS[lane], tot >> 16); }
            T[wave][lane] = 0u;
        }
    }
    if (lane < ORR_TILE) { if (fl) atomicOr(&WF[lane], fl); if (ngap) atomicAdd(&WN[lane], ngap); }
    __syncthreads();
    if (tid < ORR_TILE && c0 + tid < ncols) {
        const u32 nm = WM[tid], ns = WS[tid];
        o.cnt[c0 + tid] = nm + ns; o.cnt_s[c0 + tid] = ns;
        o.flags[c0 + tid] = WF[tid] | (WN[tid] << 8);
        o.words[c0 + tid] = ((nm + 3u) & ~3u) + (((ns + 7u) & ~7u) >> 1);
    }
}

// pass 3.  grid = tiles; dynamic LDS = 2 x ngroups x 16 words (the groups' bases inside the tile's columns).
// A wave takes a group of 64 operands in sub-batches of R rows held in registers: (a) runs per (row, column) into a small LDS
// table, (b) their prefix over the rows = where every (operand, column) piece starts inside its column (+ the member
// directory), (c) the runs into a per-wave LDS staging area laid out as the column regions are, (d) every column's piece of
// the sub-batch written out as ONE contiguous run of 2- / 4-byte elements.  (The first version wrote the runs straight from
// the row lanes: every store instruction touched ~14 cache lines with a few bytes each -- 2 G partial-line writes for 51 M
// lines of output, 14.1 ms.  Staged: lines are written whole.)  A sub-batch whose pieces do not fit the staging area -- dense
// operands -- takes the direct stores.
#define C2_SKIP_DIR 1
#define C2_SKIP_RUNS 6
#define C2_CAPM 32u                  // staged multi-bit runs per column and sub-batch
#define C2_CAPS 160u                 // staged single-bit positions per column and sub-batch
template <int R, int NW>
__global__ __launch_bounds__(NW * 64)
void k_coll2_scatter(const u32x4* __restrict__ optab_, u32 n, u32 ncols, u32 ngroups, int xcd_swz, const u32* __restrict__ bt,
                     const u64* __restrict__ off, const u32* __restrict__ cnt, const u32* __restrict__ cnt_s,
                     u32* __restrict__ runs, u32* __restrict__ dirm, u32* __restrict__ dirs,
                     int skip /* C2_SKIP_DIR: the member directory is not written (bmx_collection_prepare: it is built when a call first needs it);
                                 C2_SKIP_RUNS: only the directory is (that later pass) */)
{
    extern __shared__ u32 lds_dyn[];
    u32* Pm = lds_dyn;                          // [ngroups][16]: multi-bit runs of the column before group G
    u32* Ps = lds_dyn + (size_t)ngroups * 16u;  // ... single-bit runs
    __shared__ u32 TT[NW][R][16], TMm[NW][R + 1][16], TSs[NW][R + 1][16];
    __shared__ u32 SM[NW][ORR_TILE][C2_CAPM];
    __shared__ u16 SS[NW][ORR_TILE][C2_CAPS];
    __shared__ u64 OC[16];
    __shared__ u32 NM4[16];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = uniform32(tid >> 6);
    const u32 tile = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const u32 c0 = tile * ORR_TILE;
    gcptr4 optab = as_gc4((const void*)optab_);
    const u64 dummy = (u64)(uintptr_t)optab_;
    if (tid < 16u) {
        const u32 c = c0 + tid;
        const bool in = tid < ORR_TILE && c < ncols;
        const u32 nall = in ? cnt[c] : 0u, ns = in ? cnt_s[c] : 0u;
        OC[tid] = in ? off[c] : 0ull;
        NM4[tid] = (nall - ns + 3u) & ~3u;
        if (in && !(skip & C2_SKIP_DIR)) { dirm[(size_t)c * (n + 1u) + n] = nall - ns; dirs[(size_t)c * (n + 1u) + n] = ns; }
        u32 rm = 0u, rs = 0u;
        for (u32 G = 0; G < ngroups; ++G) {
            const u32 v = bt[((size_t)tile * ngroups + G) * 16u + tid];
            Pm[G * 16u + tid] = rm; Ps[G * 16u + tid] = rs;
            rm += v & 0xFFFFu; rs += v >> 16;
        }
    }
    __syncthreads();
    const u32 le_lo = lane >= 31u ? ~0u : (2u << lane) - 1u;
    const u32 le_hi = lane < 32u ? 0u : (lane == 63u ? ~0u : (2u << (lane - 32u)) - 1u);
    const u32 lane16 = lane << 4;
    u32* tt = &TT[wave][0][0]; u32* tm = &TMm[wave][0][0]; u32* ts = &TSs[wave][0][0];
    auto wfence = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (u32 G = wave; G < ngroups; G += (u32)NW) {
        const u32 op = G * C2_GROUP + lane;
        u32x4 e0, e1;
        or_rec_fetch_a(e0, e1, optab, op, n);
        const u32x4 t = or_rec_fetch_b(e0, e1, op, n, tile, optab);
        OrRec rec;
        or_rec_make(rec, e0, e1, t, op, n, tile, dummy);
        const u64 slow_all = __ballot(op < n && (rec.info & TREC_SLOW) != 0u);
        u32 rm = Pm[G * 16u + (lane & 15u)], rs = Ps[G * 16u + (lane & 15u)];      // lanes 0 .. 13: running bases of column c0 + lane
        for (u32 j0 = 0; j0 < C2_GROUP; j0 += (u32)R) {
            u32x4 c[R];
#pragma unroll
            for (int r = 0; r < R; ++r) c[r] = c2_row_load(rec, j0 + (u32)r, lane16, dummy);
            wfence();                                                       // (the sub-batch before has read tm / ts / the staging area)
            for (u32 z = lane; z < (u32)R * 16u; z += 64u) tt[z] = 0u;
            // (a) runs per (row, column)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u32 jj = j0 + (u32)r;
                const u32 ahi = (u32)__builtin_amdgcn_readlane((int)rec.ahi, (int)jj);
                const u32 nch = OREC_NCH(ahi);
                if (nch) {
                    const u32 mlo = (u32)__builtin_amdgcn_readlane((int)rec.mlo, (int)jj), mhi = (u32)__builtin_amdgcn_readlane((int)rec.mhi, (int)jj);
                    const u32 info = (u32)__builtin_amdgcn_readlane((int)rec.info, (int)jj);
                    const u32 nx = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)c[r].x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                    const u32 col = c2_col_of_lane(mlo, mhi, le_lo, le_hi, ahi, info);
                    u32 multi, single, y[4];
                    c2_classify(c[r], nx, multi, single, y);
                    const u32 v = lane < nch ? (u32)__popc(multi) | ((u32)__popc(single) << 16) : 0u;
                    if (v) atomicAdd(&tt[r * 16 + (int)col], v);
                } else if ((slow_all >> jj) & 1ull) {
                    const u64 dt = (u64)(u32)__builtin_amdgcn_readlane((int)rec.dlo, (int)jj) | ((u64)(u32)__builtin_amdgcn_readlane((int)rec.dhi, (int)jj) << 32);
                    const u32 nblk = (u32)__builtin_amdgcn_readlane((int)rec.nblk, (int)jj);
                    if (lane < ORR_TILE) {
                        const u64 d = c2_slow_desc(dt, nblk, c0, lane);
                        if (DESC_K(d) == K_GAP) {
                            u32 m = 0u, sg = 0u;
                            c2_walk_block(d, [&](u32 st, u32 en) { if (st == en) ++sg; else ++m; });
                            tt[r * 16 + (int)lane] += m | (sg << 16);
                        }
                    }
                }
            }
            wfence();
            // (b) positions of every (row, column) piece inside its column; row R = the end of the sub-batch's piece
            const u32 rm0 = rm, rs0 = rs;
            if (lane < 16u) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const u32 v = tt[r * 16 + (int)lane];
                    tm[r * 16 + (int)lane] = rm; ts[r * 16 + (int)lane] = rs;
                    rm += v & 0xFFFFu; rs += v >> 16;
                }
                tm[R * 16 + (int)lane] = rm; ts[R * 16 + (int)lane] = rs;
            }
            // does every column's piece fit the staging area?  (wave-uniform)
            const bool staged = __ballot(lane < ORR_TILE && (rm - rm0 > C2_CAPM || rs - rs0 > C2_CAPS)) == 0ull;
            wfence();
            // the member directory: lane = (column, row), rows fastest -- a column's R entries are consecutive words.  (Every
            // lane takes part in the ds_bpermutes: a disabled source lane would read as zero.)
#pragma unroll
            for (u32 it = 0; it < (ORR_TILE * (u32)R + 63u) / 64u; ++it) {
                const u32 z = it * 64u + lane;
                const bool zin = z < ORR_TILE * (u32)R;
                const u32 k = zin ? z / (u32)R : 0u, r = zin ? z % (u32)R : 0u, cc = c0 + k, jj = j0 + r, i = G * C2_GROUP + jj;
                const u32 info = (u32)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)rec.info);
                const u64 dt = (u64)(u32)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)rec.dlo) | ((u64)(u32)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)rec.dhi) << 32);
                const u32 nblk = (u32)__builtin_amdgcn_ds_bpermute((int)(jj << 2), (int)rec.nblk);
                if (zin && i < n && cc < ncols && !(skip & C2_SKIP_DIR)) {
                    u32 kind = ((info >> (8u + k)) & 1u) ? (u32)K_GAP : (u32)K_NULL;
                    if (info & TREC_SLOW) kind = DESC_K(c2_slow_desc(dt, nblk, c0, k));
                    dirm[(size_t)cc * (n + 1u) + i] = tm[r * 16u + k] | (kind << 30);
                    dirs[(size_t)cc * (n + 1u) + i] = ts[r * 16u + k];
                }
            }
            // (c) the runs into their places: the staging area (positions relative to the sub-batch's piece) or, when a piece does
            // not fit, global memory directly
            if (!(skip & C2_SKIP_RUNS))
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const u32 jj = j0 + (u32)r;
                const u32 ahi = (u32)__builtin_amdgcn_readlane((int)rec.ahi, (int)jj);
                const u32 nch = OREC_NCH(ahi);
                if (nch) {
                    const u32 mlo = (u32)__builtin_amdgcn_readlane((int)rec.mlo, (int)jj), mhi = (u32)__builtin_amdgcn_readlane((int)rec.mhi, (int)jj);
                    const u32 info = (u32)__builtin_amdgcn_readlane((int)rec.info, (int)jj);
                    const u32 nx = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)c[r].x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                    const u32 col = c2_col_of_lane(mlo, mhi, le_lo, le_hi, ahi, info);
                    u32 multi, single, y[4];
                    c2_classify(c[r], nx, multi, single, y);
                    const bool act = lane < nch;
                    if (!act) { multi = 0u; single = 0u; }
                    const u32 v = (u32)__popc(multi) | ((u32)__popc(single) << 16);
                    const u32 excl = wave_scan_incl(v, lane) - v;
                    // rank inside the block: runs of the lanes of this block before this one = excl - excl(first lane of the block)
                    const u32 klo = mlo & le_lo, khi = mhi & le_hi;
                    const u32 first = khi ? 63u - (u32)__builtin_clz(khi) : 31u - (u32)__builtin_clz(klo | 1u);
                    const u32 rank = excl - (u32)__builtin_amdgcn_ds_bpermute((int)(first << 2), (int)excl);
                    if (act) {
                        const u32 pm = tm[r * 16 + (int)col] + (rank & 0xFFFFu), ps = ts[r * 16 + (int)col] + (rank >> 16);
                        if (staged) {
                            u32* om = &SM[wave][col][pm - tm[col]];
                            u16* os = &SS[wave][col][ps - ts[col]];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if ((multi >> i) & 1u) *om++ = ((y[i] & 0xFFFFu) + 1u) | (y[i] & 0xFFFF0000u);
                                else if ((single >> i) & 1u) *os++ = (u16)(y[i] >> 16);
                            }
                        } else {
                            u32* om = runs + OC[col] + pm;
                            u16* os = reinterpret_cast<u16*>(runs + OC[col] + NM4[col]) + ps;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if ((multi >> i) & 1u) *om++ = ((y[i] & 0xFFFFu) + 1u) | (y[i] & 0xFFFF0000u);
                                else if ((single >> i) & 1u) *os++ = (u16)(y[i] >> 16);
                            }
                        }
                    }
                } else if ((slow_all >> jj) & 1ull) {
                    const u64 dt = (u64)(u32)__builtin_amdgcn_readlane((int)rec.dlo, (int)jj) | ((u64)(u32)__builtin_amdgcn_readlane((int)rec.dhi, (int)jj) << 32);
                    const u32 nblk = (u32)__builtin_amdgcn_readlane((int)rec.nblk, (int)jj);
                    if (lane < ORR_TILE) {
                        const u64 d = c2_slow_desc(dt, nblk, c0, lane);
                        if (DESC_K(d) == K_GAP) {
                            const u32 pm = tm[r * 16 + (int)lane], ps = ts[r * 16 + (int)lane];
                            if (staged) {
                                u32* om = &SM[wave][lane][pm - tm[lane]];
                                u16* os = &SS[wave][lane][ps - ts[lane]];
                                c2_walk_block(d, [&](u32 st, u32 en) { if (st == en) *os++ = (u16)st; else *om++ = st | (en << 16); });
                            } else {
                                u32* om = runs + OC[lane] + pm;
                                u16* os = reinterpret_cast<u16*>(runs + OC[lane] + NM4[lane]) + ps;
                                c2_walk_block(d, [&](u32 st, u32 en) { if (st == en) *os++ = (u16)st; else *om++ = st | (en << 16); });
                            }
                        }
                    }
                }
            }
            // (d) every column's piece of the sub-batch, contiguous in the staging area, to its place in the column region
            if (staged && !(skip & C2_SKIP_RUNS)) {
                wfence();
                // (all the staged values of a lane are read first, then stored: fourteen dependent LDS round trips otherwise)
                u32 vm[ORR_TILE]; u16 vs[ORR_TILE][(C2_CAPS + 63u) / 64u];
#pragma unroll
                for (u32 k = 0; k < ORR_TILE; ++k) {
                    vm[k] = SM[wave][k][lane < C2_CAPM ? lane : 0u];
#pragma unroll
                    for (u32 q = 0; q < (C2_CAPS + 63u) / 64u; ++q) vs[k][q] = SS[wave][k][q * 64u + lane < C2_CAPS ? q * 64u + lane : 0u];
                }
#pragma unroll
                for (u32 k = 0; k < ORR_TILE; ++k) {
                    const u32 m0 = tm[k], m1 = tm[R * 16 + (int)k], s0 = ts[k], s1 = ts[R * 16 + (int)k];       // (uniform: every lane reads the same words)
                    u32* om = runs + OC[k] + m0;
                    u16* os = reinterpret_cast<u16*>(runs + OC[k] + NM4[k]) + s0;
                    if (lane < m1 - m0) om[lane] = vm[k];
#pragma unroll
                    for (u32 q = 0; q < (C2_CAPS + 63u) / 64u; ++q) if (q * 64u + lane < s1 - s0) os[q * 64u + lane] = vs[k][q];
                }
            }
        }
    }
}
