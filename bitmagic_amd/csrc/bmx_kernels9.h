// bmx_kernels9.h -- round 5: the FIRST aggregation over GAP-only operands in the AND roles (combine_and / combine_and_sub
// and every arg-group of a counts or results pipeline: aggregator::combine_and_sub, src/bmaggregator.h:1162,1292-1399; per
// column sort_input_blocks_and :2315 + process_gap_blocks_and :1820 -> gap_and_to_bitset src/bmfunc.h:4847 and
// process_gap_blocks_sub :1854 -> gap_sub_to_bitset :4669) straight from the operands' own slabs: no packed collection.
#pragma once
#include "bmx_kernels8.h"

// ---------------------------------------------------------------------------
// What rounds 3 / 4 had for this case.  k_pipe_counts_gapcount (bmx_kernels5.h) counts the operands covering every position
// in two 64-KiB byte-counter arrays: 136 KiB of LDS, so ONE 1024-thread workgroup per CU whose phases (zero 128 KiB, fetch
// pointers -> headers -> run lists, scan 128 KiB) never overlap with anybody else's: 0.37 / 0.17 of the roofline at 0.3 % /
// 0.1 %.  The packed collection (bmx_kernels6.h) showed what the arithmetic allows once the run lists arrive as a stream:
// AND_i x_i = NOT OR_i NOT x_i -- the union of the operands' 0-runs, an edge-word bitmap U plus a word-granular difference
// array D for the interior of long runs (coll_apply_run / coll_fold: at most four LDS atomics per run whatever its length) --
// in 16 KiB of LDS, several workgroups per CU: 0.66 / 0.57.  But it needs the collection (5.6 ms to build, once).
//
// This kernel is that arithmetic on the reference-format operands themselves:
//  * one workgroup per (block column, arg-group); its waves share the operand list of the pipeline row (k_pipe_sort: GAP
//    entries packed from the back of a region, NULL / FULL short-circuits already folded into the row flags);
//  * a row entry carries the block's len and start bit next to the pointer (GMETA), so a block's run list is requested
//    without waiting for its header: entry -> data is the whole dependent chain, and the entries of up to 64 operands of a
//    wave arrive in ONE gather;
//  * a run list is read in PIECES of 64 chunks of 16 B -- one coalesced wave load of 1 KiB, lane L = chunk L -- DEPTH
//    pieces in flight per wave whatever the operands' lengths (a 780-word block of the 0.3 % case is two pieces, a 265-word
//    block of the 0.1 % case is half a piece);
//  * a chunk (block words 8c .. 8c + 7; word k = last position of run k) holds four runs of the wanted value: for blocks
//    whose wanted runs have odd k they are the chunk's four dwords as they stand (lo16 = end of the run before, hi16 = end
//    of the run), for the others the dwords shifted by one word (v_alignbit with the dword before; lane 0 takes the last
//    word of the lane before it by DPP, of the piece before it from an SGPR) -- no decode, no per-run loop.
// The SUB list is the same union with the other polarity (the 1-runs), complemented into the accumulator.
// ---------------------------------------------------------------------------

enum { AR_COUNT = 0, AR_STORE = 1 };

// one run [s, e] of a union: its edge words into U; the words between them through the difference array D (+1 behind the first,
// -1 at the last: an adjacent pair of words nets to nothing, so no test for "longer than two words").  r = lo16: the position
// BEFORE the run (0xFFFF for a run that starts the block: the 16-bit add wraps to 0) | hi16: its last position
#ifdef BMX_DIAG
// timing probes: 4 = the vector work without the LDS atomics (folded into a register), 8 = the atomics at conflict-free addresses
__device__ __forceinline__ void ar_apply_run_diag(u32 r, bool valid, u32* U, int* D, int diag, u32 lane, u32& sink)
{
    const u32 s = (u32)(u16)((u16)r + (u16)1u), e = r >> 16;
    u32 ws = s >> 5, we = r >> 21;
    const u32 ml = ~0u << (s & 31u), mh = ~(~1u << (e & 31u));
    const bool same = ws == we;
    if (diag & 4) { if (valid) { sink ^= (same ? (ml & mh) : ml) + ws; if (!same) sink ^= mh + we; } return; }
    if (valid) {
        atomicOr(&U[lane], same ? (ml & mh) : ml);
        if (!same) { atomicOr(&U[lane + 64u], mh); atomicAdd(&D[lane + 128u], 1); atomicSub(&D[lane + 192u], 1); }
    }
}
#endif
__device__ __forceinline__ void ar_apply_run(u32 r, bool valid, u32* U, int* D)
{
    const u32 s = (u32)(u16)((u16)r + (u16)1u), e = r >> 16;
    const u32 ws = s >> 5, we = r >> 21;
    const u32 ml = ~0u << (s & 31u), mh = ~(~1u << (e & 31u));
    const bool same = ws == we;
    if (valid) {
        atomicOr(&U[ws], same ? (ml & mh) : ml);
        if (!same) { atomicOr(&U[we], mh); atomicAdd(&D[ws + 1u], 1); atomicSub(&D[we], 1); }
    }
}

// union of the runs of value `want` of the n GAP operands whose row entries sit at list_back[0], list_back[-1], ...
// (pointer | GMETA << 48) into U / D.  The run lists of a wave's operands are read as ONE stream of 16-byte chunks, 64
// chunks (one per lane) per load whatever the blocks' lengths: a piece holds the tail of one block, whole blocks and the
// head of another, so every lane of every load carries four runs (the first version gave a block its own loads: 52 % of the
// lanes busy at 0.1 %, 77 % at 0.3 % -- and the kernel is bound by its vector instructions, not by memory or LDS:
// profiles/r05_and_rows).  Which block a lane's chunk belongs to: the blocks that START inside the piece as a 64-bit mask
// (a scalar loop over those few operands), the lane's ordinal by v_mbcnt, its entry by ds_bpermute.
template <int WG, int DEPTH, bool NT>
__device__ __forceinline__ void ar_union_list(const u64* __restrict__ list_back, u32 n, u32 want, u32* U, int* D, u32 lane, u32 wave, u64 dummy, int diag = 0)
{
    constexpr u32 NW = (u32)WG / 64u;
    const u32 mine = wave < n ? (n - wave + NW - 1u) / NW : 0u;          // operands wave, wave + NW, ...
    for (u32 b0 = 0; b0 < mine; b0 += 64u) {
        const u32 nb = mine - b0 < 64u ? mine - b0 : 64u;
        u64 ent = 0ull;
        if (lane < nb) ent = *(list_back - (size_t)(wave + NW * (b0 + lane)));
        const u32 elo = (u32)ent, ehi = (u32)(ent >> 32);
        const u32 nch_l = lane < nb ? (((ehi >> 17) & 0xFFFu) + 8u) >> 3 : 0u;     // header + len run ends, in 16-byte chunks
        const u32 incl = wave_scan_incl(nch_l, lane);
        const u32 cum = incl - nch_l;                                     // first chunk of operand `lane` in the wave's stream
        const u32 T = (u32)__builtin_amdgcn_readlane((int)incl, 63);      // chunks of the batch (>= 1)
        const u32 npieces = (T + 63u) >> 6;
        u32x4 c[DEPTH];
        u32 cs[DEPTH], ms[DEPTH];                                         // per lane: chunk index inside its block (0x0FFFFFFF: lane idle), GMETA word of its block
        u32 iq = 0u, jn = 0u;                                             // (scalar) next piece to request, next operand to start
        auto issue = [&](int k) {
            // every slot always issues one load (past the last piece: the row itself), so the compiler's vmcnt bookkeeping stays exact
            const bool have = iq < npieces;
            const u32 base = iq << 6;
            const u32 jprev = jn;
            u64 M = 0ull;                                                 // blocks that start inside this piece, by position
            if (have) {
                while (jn < nb) {
                    const u32 cj = (u32)__builtin_amdgcn_readlane((int)cum, (int)jn);
                    if (cj >= base + 64u) break;
                    M |= 1ull << (cj - base);
                    ++jn;
                }
            }
            // operand of this lane's chunk: (operands started before the piece) - 1 + (starts at positions <= lane)
            const u64 Ms = M >> 1;
            const u32 o = (u32)__builtin_amdgcn_mbcnt_hi((u32)(Ms >> 32), __builtin_amdgcn_mbcnt_lo((u32)Ms, jprev - 1u + (u32)(M & 1ull)));
            const u32 g = base + lane;
            const u32 gc = g < T ? g : T - 1u;                             // (idle lanes repeat the stream's last chunk: same address, no traffic)
            const int oa = (int)(o << 2);
            const u32 plo = (u32)__builtin_amdgcn_ds_bpermute(oa, (int)elo);
            const u32 phi = (u32)__builtin_amdgcn_ds_bpermute(oa, (int)ehi);
            const u32 pcu = (u32)__builtin_amdgcn_ds_bpermute(oa, (int)cum);
            const u32 ci = gc - pcu;
            u64 a = ((u64)plo | ((u64)(phi & 0xFFFFu) << 32)) + ((u64)ci << 4);
            if (!have) a = dummy;
            c[k] = NT ? __builtin_nontemporal_load((gcptr4)(uintptr_t)a) : *(gcptr4)(uintptr_t)a;
            cs[k] = (have && g < T) ? ci : 0x0FFFFFFFu;               // (idle: 8 ci stays huge, so no run of the lane is valid)
            ms[k] = phi;
            ++iq;
        };
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) issue(k);
        u32 carry = 0u;
        for (u32 q0 = 0; q0 < npieces; q0 += (u32)DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                if (q0 + (u32)k < npieces) {                              // (scalar)
                    const u32 ci = cs[k];
                    const u32 len = (ms[k] >> 17) & 0xFFFu;
                    const bool odd = ((ms[k] >> 16) & 1u) == want;        // wanted runs have odd k: the dwords as they stand; else shifted by one word
                    // the dword before this chunk: the lane before's .w (lane 0: the piece before's last one)
                    const u32 pw = (u32)__builtin_amdgcn_update_dpp((int)carry, (int)c[k].w, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    carry = (u32)__builtin_amdgcn_readlane((int)c[k].w, 63);
                    const u32 sel = odd ? 0x07060504u : 0x05040302u;      // v_perm: (d[t + 1]) or (d[t + 1] : d[t]) >> 16
                    const u32 x0 = ci == 0u ? (c[k].x | 0xFFFFu) : c[k].x; // the header word counts as "position -1": run 1 starts at 0
                    const u32 d[5] = {pw, x0, c[k].y, c[k].z, c[k].w};
                    // run t of the chunk is run kk = 8 ci + 2 t + odd of the block: valid iff 1 <= kk <= len, i.e. 2 t <= lim (lanes
                    // without a chunk: lim < 0); kk = 0 (even blocks' first pair) is the header, not a run
                    const int lim = (int)len - (int)(8u * ci + (odd ? 1u : 0u));       // (idle lanes: negative)
                    const int lim0 = (ci == 0u && !odd) ? -1 : lim;
#ifdef BMX_DIAG
                    if (diag & 1) { if ((d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ sel ^ (u32)lim0) == 0x12345679u) U[lane] = 1u; }    // timing probe: the loads alone
                    else if (diag & 12) {
                        u32 sink = 0u;
                        ar_apply_run_diag(__builtin_amdgcn_perm(d[1], d[0], sel), lim0 >= 0, U, D, diag, lane, sink);
                        ar_apply_run_diag(__builtin_amdgcn_perm(d[2], d[1], sel), lim >= 2, U, D, diag, lane, sink);
                        ar_apply_run_diag(__builtin_amdgcn_perm(d[3], d[2], sel), lim >= 4, U, D, diag, lane, sink);
                        ar_apply_run_diag(__builtin_amdgcn_perm(d[4], d[3], sel), lim >= 6, U, D, diag, lane, sink);
                        if (sink == 0x12345679u) U[lane] = 1u;
                    } else
#endif
                    {
                        ar_apply_run(__builtin_amdgcn_perm(d[1], d[0], sel), lim0 >= 0, U, D);
                        ar_apply_run(__builtin_amdgcn_perm(d[2], d[1], sel), lim >= 2, U, D);
                        ar_apply_run(__builtin_amdgcn_perm(d[3], d[2], sel), lim >= 4, U, D);
                        ar_apply_run(__builtin_amdgcn_perm(d[4], d[3], sel), lim >= 6, U, D);
                    }
                }
                issue(k);
            }
        }
    }
}

// One workgroup per (block column, arg-group) of a pipeline whose operands hold no bit-blocks.
//   AR_COUNT  counts[g] += popcount(AND of the AND list, minus the union of the SUB list)  (:1292-1399, counts only)
//   AR_STORE  the same block stored as column `col` of a result vector with opt_compress (:1210); ngroups = 1
template <int MODE, int WG, int DEPTH, bool NT>
__device__ __forceinline__ void ar_item(u32 item, u32* U, int* D, int* sm, u32* part, const u64* __restrict__ dmat, const u32* __restrict__ row_off, const u32* __restrict__ and_n,
                    const u32* __restrict__ sub_n, u32 col_stride, u32 ngroups, u32 col_from, u32 nitems, int xcd_swz,
                    u64* __restrict__ counts, uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                    u32 hint_from, u32 hint_to, int diag)
{
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = uniform32(tid >> 6);       // (uniform: the piece bookkeeping of ar_union_list stays in SGPRs)
    const u32 ci = item / ngroups, g = item - ci * ngroups;
    const u32 col = col_from + ci;
    if (MODE == AR_STORE && (col < hint_from || col >= hint_to)) { if (wave == 0) store_trivial(K_NULL, col, desc, st, lane); return; }
    const u64* row = dmat + (size_t)col * col_stride + uniform32(row_off[g]);
    const u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) { if (MODE == AR_STORE && wave == 0) store_trivial(K_NULL, col, desc, st, lane); return; }
    if (flags & ROW_FULL) {
        if (MODE == AR_STORE) { if (wave == 0) store_trivial(K_FULL, col, desc, st, lane); }
        else if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), 65536ull);
        return;
    }
    const u32 nga = (u32)((hdr >> 16) & 0xFFFFu), ngs = (u32)(hdr >> 48);       // (no bit-block operands in these pipelines)
    const u32 na = uniform32(and_n[g]), ns = uniform32(sub_n[g]);
    constexpr u32 W = 2048u / (u32)WG;
#pragma unroll
    for (u32 k = 0; k < W; k += 4u) {
        *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
        *reinterpret_cast<u32x4*>(&D[tid * W + k]) = (u32x4)(0u);
    }
    __syncthreads();
    const u64 dummy = (u64)(uintptr_t)dmat;
    // the union of the AND operands' 0-runs (FULL operands were dropped by the sort, :2346) ...
    if (nga) ar_union_list<WG, DEPTH, NT>(row + 2 + na - 1u, nga, 0u, U, D, lane, wave, dummy, diag);
#ifdef BMX_DIAG
    if (diag & 2) return;                                      // timing probe: no fold, no count
#endif
    __syncthreads();
    if (nga) coll_fold<WG>(U, D, sm, tid);                     // (block-uniform; the barriers inside are reached by every thread)
    // ... complemented: the accumulator
    u32 acc[W];
#pragma unroll
    for (u32 k = 0; k < W; ++k) acc[k] = ~U[tid * W + k];
    // ... minus the union of the SUB operands' 1-runs (:1854)
    if (ngs) {
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < W; k += 4u) *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
        __syncthreads();
        ar_union_list<WG, DEPTH, NT>(row + 2 + na + ns - 1u, ngs, 1u, U, D, lane, wave, dummy, diag);
        __syncthreads();
        coll_fold<WG>(U, D, sm, tid);
#pragma unroll
        for (u32 k = 0; k < W; ++k) acc[k] &= ~U[tid * W + k];
    }
    if (MODE == AR_COUNT) {
        u32 pc = 0;
#pragma unroll
        for (u32 k = 0; k < W; ++k) pc += (u32)__popc(acc[k]);
        pc = wave_sum(pc);
        if (lane == 0) part[wave] = pc;
        __syncthreads();
        if (tid == 0) {
            u32 t = 0;
#pragma unroll
            for (u32 i = 0; i < (u32)WG / 64u; ++i) t += part[i];
            if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)t);
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < W; ++k) U[tid * W + k] = acc[k];
    __syncthreads();
    if (wave == 0) { Blk b; blk_from_lds(b, U, lane); store_result(b, col, 1, slab, desc, st, lane); }
}

// ipw (>= 1) consecutive items per workgroup (a knob: a counts pipeline of many arg-groups over short lists is hundreds of
// thousands of items; measured, it is bound by the dependent round trips inside an item, not by the dispatch rate: default 1)
template <int MODE, int WG, int DEPTH, bool NT>
__global__ __launch_bounds__(WG)
void k_agg_and_rows(const u64* __restrict__ dmat, const u32* __restrict__ row_off, const u32* __restrict__ and_n,
                    const u32* __restrict__ sub_n, u32 col_stride, u32 ngroups, u32 col_from, u32 nitems, int xcd_swz,
                    u64* __restrict__ counts, uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                    u32 hint_from, u32 hint_to, int diag, u32 ipw)
{
    __shared__ __attribute__((aligned(16))) u32 U[2048];
    __shared__ __attribute__((aligned(16))) int D[2048];
    __shared__ int sm[WG / 64];
    __shared__ u32 part[WG / 64];
    const u32 b = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    for (u32 k = 0; k < ipw; ++k) {
        const u32 item = b * ipw + k;
        if (item >= nitems) return;
        if (k) __syncthreads();                                   // (the item before has read U / D / part)
        ar_item<MODE, WG, DEPTH, NT>(item, U, D, sm, part, dmat, row_off, and_n, sub_n, col_stride, ngroups, col_from, nitems, xcd_swz, counts, slab, desc, st,
                                     hint_from, hint_to, diag);
    }
}

// ---------------------------------------------------------------------------
// pipeline::set_search_count_limit (src/bmaggregator.h:255, honoured per arg-group at :1362-1367: a group whose count has
// reached the limit is not evaluated on the following blocks).  The counts run walks ascending windows of block columns; after
// each window this one-workgroup kernel folds the window's compact counts into the per-group totals, notes the window at
// whose end a group reached the limit (stop[]: a results run truncates the group's vector there) and COMPACTS the groups
// that still need hits -- their ids and their rows of every per-group table the counts kernels read (row offsets, operand
// counts, collection member ranges, plane masks) -- so that the next window is launched over those groups only.  The number
// of groups left goes to pinned host memory: the host decides the next launch from that word, the counts stay on the device.
// ---------------------------------------------------------------------------
struct LimitTables {
    const u32* row_off; const u32* and_n; const u32* sub_n;           // the pipeline's full tables (indexed by group)
    u32* ro; u32* an; u32* sn;                                        // the compacted ones (indexed by position in `active`)
    const CollGroup* cg_in; CollGroup* cg_out;                        // member ranges of groups served by packed collections (or null)
    const u32* gmask_in; const u32* gskip_in; u32* gmask_out; u32* gskip_out; u32 nchunks;   // LDS-staged kernel: plane masks (or null)
};

__global__ __launch_bounds__(1024)
void k_limit_step(u64* __restrict__ totals, const u64* __restrict__ cc, const u32* __restrict__ active_in /* null: every group */, u32 n_in,
                  u64 limit, u32 win_end, u32* __restrict__ stop, u32* __restrict__ active_out, LimitTables t,
                  u32* __restrict__ n_active_dev, u64* __restrict__ n_active_host)
{
    __shared__ u32 wsum[16];
    __shared__ u32 base;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) base = 0u;
    __syncthreads();
    for (u32 i0 = 0; i0 < n_in; i0 += 1024u) {
        const u32 i = i0 + tid;
        bool keep = false; u32 g = 0u;
        if (i < n_in) {
            g = active_in ? active_in[i] : i;
            const u64 c = totals[g] + cc[i];
            totals[g] = c;
            keep = c < limit;
            if (!keep && stop[g] == 0xFFFFFFFFu) stop[g] = win_end;
        }
        const u64 m = __ballot(keep);
        if (lane == 0) wsum[wave] = (u32)__popcll(m);
        __syncthreads();
        u32 off = base;
        for (u32 w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            const u32 pos = off + (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            active_out[pos] = g;
            t.ro[pos] = t.row_off[g]; t.an[pos] = t.and_n[g]; t.sn[pos] = t.sub_n[g];
            if (t.cg_in) t.cg_out[pos] = t.cg_in[g];
            if (t.gmask_in) {
                for (u32 k = 0; k < t.nchunks; ++k) t.gmask_out[(size_t)pos * t.nchunks + k] = t.gmask_in[(size_t)g * t.nchunks + k];
                t.gskip_out[pos] = t.gskip_in[g];
            }
        }
        __syncthreads();
        if (tid == 0) { u32 s = 0; for (u32 w = 0; w < 16u; ++w) s += wsum[w]; base += s; }
        __syncthreads();
    }
    if (tid == 0) { *n_active_dev = base; *n_active_host = (u64)base; }
}

// ---------------------------------------------------------------------------
// The same limit on the ASYNCHRONOUS counts entry (round 6, bmx_pipeline_run_counts_dev): no host decision between windows, so
// nothing is compacted -- every window is enqueued over ALL arg-groups, and after a window this kernel (a thread per group) folds
// the window's counts into the totals and points a group that has enough at a NULL entry of every table the counts kernels
// read: the always-ROW_EMPTY row that ends each column record (row kernels return at its header), an empty member range
// (k_coll_members: an empty AND list ends the column), gskip = 1 (the LDS-staged kernel).  A finished group then costs one
// header read per (column, group) item instead of its operands -- "can find more, cannot find less" (src/bmaggregator.h:1362).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_pipe_null_rows(u64* __restrict__ dmat, u32 ncols, u32 col_stride, u32 null_off)
{
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    u64* row = dmat + (size_t)c * col_stride + null_off;
    row[0] = 0ull; row[1] = ROW_EMPTY;
}

__global__ __launch_bounds__(256)
void k_limit_null(u64* __restrict__ totals, const u64* __restrict__ cc, u32 ng, u64 limit, u32 null_off,
                  u32* __restrict__ ro, CollGroup* __restrict__ cg, u32* __restrict__ gskip, u64* __restrict__ out /* non-null on the last window: the totals */)
{
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const u64 c = totals[g] + cc[g];
    totals[g] = c;
    if (c >= limit) {
        ro[g] = null_off;
        if (cg) cg[g] = CollGroup{0u, 0u, 0u, 0u};
        if (gskip) gskip[g] = 1u;
    }
    if (out) out[g] = c;
}

