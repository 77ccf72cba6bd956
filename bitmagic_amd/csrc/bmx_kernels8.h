// bmx_kernels8.h -- round 4: packed collections as first-class objects: a member directory per block column, so that an
// aggregation over ANY subset (in any order) of a prepared collection's vectors, and every arg-group of a pipeline over
// them, works from the collection's column regions (aggregator::combine_or / combine_and_sub take an arbitrary operand
// list per call, src/bmaggregator.h:1101-1121,1162; pipelines batch many arg-groups over shared operands, :1292-1399).
#pragma once
#include "bmx_kernels7.h"

// ---------------------------------------------------------------------------
// Member directory.  A column of a collection is the concatenation of its members' runs in member order, so member i owns
// the entries [dir[c][i], dir[c][i + 1]) of column c -- the prefix k_coll_count computes anyway (`pre`).  It is kept,
// transposed to column-major (dir[c][0 .. n], n + 1 words per column: one contiguous read per workgroup) with the block
// kind of member i in column c in the two top bits of dir[c][i]; split bags (polarity 1) have two of them: multi-bit runs
// (dir) and single-bit positions (dir_s).  4 B (8 B split) per (member, column) next to the ~2 x len B of run data.
// ---------------------------------------------------------------------------
#define CDIR_MASK 0x3FFFFFFFu
#define CDIR_KIND(x) ((x) >> 30)

// pre[i][c] (operand-major, kind in bits 30..31; sgl[i][c] = singles before operand i, split bags only) -> dir[c][i], dir_s[c][i]
// 32 x 32 tiles through LDS: both sides coalesced.  Entry n of a column = its total.
__global__ __launch_bounds__(1024)
void k_coll_dir(const u32* __restrict__ pre, const u32* __restrict__ sgl, const u32* __restrict__ cnt, const u32* __restrict__ cnt_s,
                u32 n, u32 ncols, u32* __restrict__ dirm, u32* __restrict__ dirs)
{
    __shared__ u32 tm[32][33], ts[32][33];
    const u32 tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;
    const u32 cx = blockIdx.x * 32u, ix = blockIdx.y * 32u;
    {
        const u32 i = ix + ty, c = cx + tx;
        u32 vm = 0u, vs = 0u;
        if (c < ncols) {
            if (i < n) {
                const u32 p = pre[(size_t)i * ncols + c];
                vs = sgl ? sgl[(size_t)i * ncols + c] : 0u;
                vm = ((p & CDIR_MASK) - vs) | (p & ~CDIR_MASK);
            } else if (i == n) { vs = cnt_s ? cnt_s[c] : 0u; vm = cnt[c] - vs; }
        }
        tm[ty][tx] = vm; ts[ty][tx] = vs;
    }
    __syncthreads();
    const u32 c = cx + ty, i = ix + tx;
    if (c < ncols && i <= n) {
        dirm[(size_t)c * (n + 1u) + i] = tm[tx][ty];
        if (dirs) dirs[(size_t)c * (n + 1u) + i] = ts[tx][ty];
    }
}

struct CollView { const u32* runs; const u64* off; const u32* dir; const u32* dir_s; u32 nvec, ncols; };
struct CollGroup { u32 a_off, a_n, s_off, s_n; };       // member-index ranges of an arg-group's AND / OR list and SUB list

#define CMF_NULL 1u
#define CMF_FULL 2u
#define CMF_GAP  4u

__device__ __forceinline__ u32 wave_max_u32(u32 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)v, o, 64); v = t > v ? t : v; }
    return v;
}

// The listed members' runs of column c into the wave's bitmap U (2048 words of LDS).  Four members per step, 16 lanes each:
// the lanes of a quarter-wave read consecutive entries of their member's piece (one request however short the piece).  A run
// inside one or two words is the owning lane's one or two ds_or; a run with interior words -- nearly every 0-run of a sparse
// AND-list member -- leaves its edges to the lane and has the WAVE store the interior, 64 words per instruction.
// Returns the kinds met (CMF_*), the same in every lane.
__device__ __forceinline__ void coll_wave_run(u32 r, bool valid, u32* U, u32 lane)
{
    const u32 s = r & 0xFFFFu, e = r >> 16;
    const u32 ws = s >> 5, we = e >> 5;
    const u32 lo = ~0u << (s & 31u), hi = ~0u >> (31u - (e & 31u));
    const bool same = ws == we;
    if (valid) atomicOr(&U[ws], same ? (lo & hi) : lo);
    if (valid && !same) atomicOr(&U[we], hi);
    u64 lm = __ballot(valid && we > ws + 1u);
    while (lm) {                                                       // (wave-uniform loop over the lanes that hold a long run)
        const u32 l = (u32)__builtin_ctzll(lm);
        lm &= lm - 1ull;
        const u32 w0 = (u32)__builtin_amdgcn_readlane((int)ws, (int)l) + 1u, w1 = (u32)__builtin_amdgcn_readlane((int)we, (int)l);
        for (u32 w = w0 + lane; w < w1; w += 64u) U[w] = ~0u;          // an OR accumulator only gains bits: a plain store of all-ones loses nothing
    }
}

// 64 members at a time, so that the dependent reads of a member (its index -> its directory words -> its entries) are three
// round trips for the whole batch instead of three per member: lane j fetches member j's index and directory words, then,
// round after round (a round = up to 16 entries of every member), the entries of all 16 quads are requested before the
// first one is applied.
__device__ __forceinline__ u32 coll_wave_members(const CollView& C, u32 c, const u32* __restrict__ midx, u32 m, u32* U, u32 lane)
{
    if (!m) return 0u;
    if (c >= C.ncols) return CMF_NULL;                                 // past the collection's last column: every member is NULL there
    const u32* dm = C.dir + (size_t)c * (C.nvec + 1u);
    const u32* ds = C.dir_s ? C.dir_s + (size_t)c * (C.nvec + 1u) : nullptr;
    const u64 off = C.off[c];
    const u32 nm_col = dm[C.nvec] & CDIR_MASK;
    const u32* multis = C.runs + off;
    const u16* singles = reinterpret_cast<const u16*>(C.runs + off + ((nm_col + 3u) & ~3u));
    const u32 sub = lane & 15u, grp = lane >> 4;
    u32 flags = 0u;
    for (u32 j0 = 0; j0 < m; j0 += 64u) {
        const bool in = j0 + lane < m;
        const u32 i = in ? midx[j0 + lane] : 0u;
        const u32 a0 = in ? dm[i] : 0u, b = in ? dm[i + 1u] & CDIR_MASK : 0u;
        const u32 sa = (in && ds) ? ds[i] : 0u, sb = (in && ds) ? ds[i + 1u] : 0u;
        if (in) { const u32 kind = CDIR_KIND(a0); flags |= kind == K_NULL ? CMF_NULL : kind == K_FULL ? CMF_FULL : CMF_GAP; }
        const u32 a = a0 & CDIR_MASK;
        const u32 mc = m - j0 < 64u ? m - j0 : 64u, nsteps = (mc + 3u) >> 2;
        for (u32 k16 = 0; ; k16 += 16u) {
            if (__ballot(in && (b - a > k16 || sb - sa > k16)) == 0ull) break;      // nobody has entries left in this round
            u32 em[16], es[16], vm = 0u, vs = 0u;
#pragma unroll
            for (u32 q = 0; q < 16u; ++q) {
                em[q] = 0u; es[q] = 0u;
                if (q < nsteps) {                                      // (wave-uniform)
                    const int src = (int)((4u * q + grp) << 2);
                    const u32 aq = (u32)__builtin_amdgcn_ds_bpermute(src, (int)a), bq = (u32)__builtin_amdgcn_ds_bpermute(src, (int)b);
                    const u32 saq = (u32)__builtin_amdgcn_ds_bpermute(src, (int)sa), sbq = (u32)__builtin_amdgcn_ds_bpermute(src, (int)sb);
                    const u32 e = aq + k16 + sub, se = saq + k16 + sub;
                    if (e < bq) { em[q] = multis[e]; vm |= 1u << q; }
                    if (se < sbq) { es[q] = singles[se]; vs |= 1u << q; }
                }
            }
#pragma unroll
            for (u32 q = 0; q < 16u; ++q) {
                if (q < nsteps) {
                    coll_wave_run(em[q], (vm >> q) & 1u, U, lane);
                    if ((vs >> q) & 1u) atomicOr(&U[es[q] >> 5], 1u << (es[q] & 31u));
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) flags |= __shfl_xor(flags, o, 64);
    return flags;
}

enum { CM_OR_STORE = 0, CM_AND_STORE = 1, CM_AND_COUNT = 2 };

// One WAVE per (block column, arg-group) item, items of a column next to each other (the column's region stays in the caches
// across its groups), four waves per workgroup, 8 KiB of LDS per wave.
//   CM_OR_STORE   result = union of the listed members of A (polarity 1), stored with the aggregator's optimisation mode
//   CM_AND_STORE  result = AND of the listed members of A (polarity 0: complement of the union of their 0-runs) minus the
//                 union of the listed members of S (polarity 1), stored with opt_compress (combine_and_sub, :1162,1210)
//   CM_AND_COUNT  the same per arg-group, counted (:1392-1399)
#define CM_WAVES 4
template <int MODE>
__global__ __launch_bounds__(CM_WAVES * 64)
void k_coll_members(CollView A, CollView S, const u32* __restrict__ midx, const CollGroup* __restrict__ groups, u32 ngroups,
                    u32 col_base, u32 ncols, int opt_compress, u64* __restrict__ counts,
                    uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st)
{
    __shared__ __attribute__((aligned(16))) u32 lds_all[CM_WAVES * 2048];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 item = (u64)blockIdx.x * CM_WAVES + wave;
    const u32 c = col_base + (u32)(item / ngroups), g = (u32)(item % ngroups);
    if (c >= ncols) return;                                            // (no workgroup barrier below: a wave may leave)
    u32* U = lds_all + wave * 2048u;
    const u32 a_off = uniform32(groups[g].a_off), a_n = uniform32(groups[g].a_n);
    const u32 s_off = uniform32(groups[g].s_off), s_n = uniform32(groups[g].s_n);
    auto zero = [&]() {
        u32x4* u4 = reinterpret_cast<u32x4*>(U);
#pragma unroll
        for (int i = 0; i < 8; ++i) u4[i * 64 + lane] = (u32x4)(0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto settle = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    zero();
    const u32 fa = coll_wave_members(A, c, midx + a_off, a_n, U, lane);
    settle();
    if (MODE == CM_OR_STORE) {
        // any FULL member saturates the column (:2300); no GAP member: nothing to store (:2294)
        if ((fa & CMF_FULL) || !(fa & CMF_GAP)) { store_trivial((fa & CMF_FULL) ? (u32)K_FULL : (u32)K_NULL, c, desc, st, lane); return; }
        Blk b; blk_from_lds(b, U, lane);
        (void)store_result_mode(b, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);
        return;
    }
    // AND: an empty AND list or a NULL member ends the column (:1170, :2327); FULL members are ignored (:2346)
    bool empty = a_n == 0u || (fa & CMF_NULL);
    Blk acc;
    blk_from_lds(acc, U, lane);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc.r[i] = ~acc.r[i];
    if (!empty && s_n) {
        settle();
        zero();
        const u32 fs = coll_wave_members(S, c, midx + s_off, s_n, U, lane);
        settle();
        if (fs & CMF_FULL) empty = true;                               // a FULL block in the SUB list empties the column (:1746)
        Blk sb; blk_from_lds(sb, U, lane);
        blk_andn(acc, sb);
    }
    if (MODE == CM_AND_COUNT) {
        if (empty) return;
        const u32 pc = wave_sum(blk_lane_popcount(acc));
        if (lane == 0 && pc) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)pc);
        return;
    }
    if (empty) { store_trivial(K_NULL, c, desc, st, lane); return; }
    store_result(acc, c, 1, slab, desc, st, lane);
}
