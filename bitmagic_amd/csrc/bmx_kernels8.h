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

// the runs of the listed members of column c into U / D (coll_apply_run, bmx_kernels6.h); flags: kinds met
template <int WG>
__device__ __forceinline__ u32 coll_members_apply(const CollView& C, u32 c, const u32* __restrict__ midx, u32 m, u32* U, int* D, u32 tid, u32& flags)
{
    u32 any_long = 0u;
    if (c >= C.ncols) { if (m && tid == 0) flags |= CMF_NULL; return 0u; }       // past the collection's last column: every member is NULL there
    const u32* dm = C.dir + (size_t)c * (C.nvec + 1u);
    const u32* ds = C.dir_s ? C.dir_s + (size_t)c * (C.nvec + 1u) : nullptr;
    const u64 off = C.off[c];
    const u32 nm_col = dm[C.nvec] & CDIR_MASK;
    const u32* multis = C.runs + off;
    const u16* singles = reinterpret_cast<const u16*>(C.runs + off + ((nm_col + 3u) & ~3u));
    for (u32 j = tid; j < m; j += (u32)WG) {
        const u32 i = midx[j];
        const u32 a0 = dm[i], b = dm[i + 1u] & CDIR_MASK;
        const u32 kind = CDIR_KIND(a0);
        flags |= kind == K_NULL ? CMF_NULL : kind == K_FULL ? CMF_FULL : CMF_GAP;
        for (u32 e = a0 & CDIR_MASK; e < b; ++e) coll_apply_run(multis[e], true, U, D, any_long);
        if (ds) {
            const u32 sb = ds[i + 1u];
            for (u32 e = ds[i]; e < sb; ++e) { const u32 p = singles[e]; atomicOr(&U[p >> 5], 1u << (p & 31u)); }
        }
    }
    return any_long;
}

enum { CM_OR_STORE = 0, CM_AND_STORE = 1, CM_AND_COUNT = 2 };

// One workgroup per block column; the arg-groups of a counts pipeline are walked inside (the column's region stays in the
// caches across them).
//   CM_OR_STORE   result = union of the listed members of A (polarity 1), stored with the aggregator's optimisation mode
//   CM_AND_STORE  result = AND of the listed members of A (polarity 0: complement of the union of their 0-runs) minus the
//                 union of the listed members of S (polarity 1), stored with opt_compress (combine_and_sub, :1162,1210)
//   CM_AND_COUNT  the same per arg-group, counted (:1392-1399)
template <int MODE, int WG>
__global__ __launch_bounds__(WG)
void k_coll_members(CollView A, CollView S, const u32* __restrict__ midx, const CollGroup* __restrict__ groups, u32 ngroups,
                    u32 col_base, u32 ncols, int opt_compress, u64* __restrict__ counts,
                    uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st)
{
    __shared__ __attribute__((aligned(16))) u32 U[2048];
    __shared__ __attribute__((aligned(16))) int D[2048];
    __shared__ int sm[WG / 64];
    __shared__ u32 s_long, s_flags;
    __shared__ u32 part[WG / 64];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 c = col_base + blockIdx.x;
    if (c >= ncols) return;
    constexpr u32 W = 2048u / WG;
    for (u32 g = 0; g < ngroups; ++g) {
        const u32 a_off = uniform32(groups[g].a_off), a_n = uniform32(groups[g].a_n);
        const u32 s_off = uniform32(groups[g].s_off), s_n = uniform32(groups[g].s_n);
        __syncthreads();                                               // (the previous group's readers of U / part are done)
#pragma unroll
        for (u32 k = 0; k < W; k += 4u) {
            *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
            *reinterpret_cast<u32x4*>(&D[tid * W + k]) = (u32x4)(0u);
        }
        if (tid == 0) { s_long = 0u; s_flags = 0u; }
        __syncthreads();
        u32 fl = 0u;
        u32 al = coll_members_apply<WG>(A, c, midx + a_off, a_n, U, D, tid, fl);
        if (fl) atomicOr(&s_flags, fl);
        if (al) s_long = 1u;
        __syncthreads();
        const u32 fa = s_flags;
        if (s_long) coll_fold<WG>(U, D, sm, tid);
        if (MODE == CM_OR_STORE) {
            // any FULL member saturates the column (:2300); no GAP member: nothing to store (:2294)
            if ((fa & CMF_FULL) || !(fa & CMF_GAP)) { if (wave == 0) store_trivial((fa & CMF_FULL) ? (u32)K_FULL : (u32)K_NULL, c, desc, st, lane); }
            else if (wave == 0) { Blk b; blk_from_lds(b, U, lane); (void)store_result_mode(b, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane); }
            continue;
        }
        // AND: an empty AND list or a NULL member ends the column (:1170, :2327); FULL members are ignored (:2346)
        bool empty = a_n == 0u || (fa & CMF_NULL);
        u32 acc[W];
#pragma unroll
        for (u32 k = 0; k < W; ++k) acc[k] = ~U[tid * W + k];
        if (!empty && s_n) {
            __syncthreads();
#pragma unroll
            for (u32 k = 0; k < W; k += 4u) *reinterpret_cast<u32x4*>(&U[tid * W + k]) = (u32x4)(0u);
            if (tid == 0) { s_long = 0u; s_flags = 0u; }
            __syncthreads();
            u32 fs = 0u;
            u32 sl = coll_members_apply<WG>(S, c, midx + s_off, s_n, U, D, tid, fs);
            if (fs) atomicOr(&s_flags, fs);
            if (sl) s_long = 1u;
            __syncthreads();
            if (s_flags & CMF_FULL) empty = true;                      // a FULL block in the SUB list empties the column (:1746)
            if (s_long) coll_fold<WG>(U, D, sm, tid);
#pragma unroll
            for (u32 k = 0; k < W; ++k) acc[k] &= ~U[tid * W + k];
        }
        if (MODE == CM_AND_COUNT) {
            u32 pc = 0;
#pragma unroll
            for (u32 k = 0; k < W; ++k) pc += (u32)__popc(acc[k]);
            pc = empty ? 0u : wave_sum(pc);
            if (lane == 0) part[wave] = pc;
            __syncthreads();
            if (tid == 0) {
                u32 t = 0;
#pragma unroll
                for (u32 i = 0; i < WG / 64u; ++i) t += part[i];
                if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)t);
            }
            continue;
        }
        if (empty) { if (wave == 0) store_trivial(K_NULL, c, desc, st, lane); continue; }
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < W; ++k) U[tid * W + k] = acc[k];
        __syncthreads();
        if (wave == 0) { Blk b; blk_from_lds(b, U, lane); store_result(b, c, 1, slab, desc, st, lane); }
    }
}
