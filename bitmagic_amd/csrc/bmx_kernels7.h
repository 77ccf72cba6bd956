// bmx_kernels7.h -- bm::count_* over two vectors of ANY block kinds as a stream (round 3).
#pragma once
#include "bmx_kernels6.h"

// ---------------------------------------------------------------------------
// k_count_op2 gives every block column its own short-lived wave: descriptor -> blocks -> count -> exit, three
// workgroup rounds per CU; it takes ~45 us for two 1e9-bit vectors whatever they hold (250 MB of bit-blocks or the 180 MB
// of the 1 % mixed case: 53 % of the HBM peak).  k_count_op2_stream fixed that for bit-block-only operands: a wave owns a
// contiguous stretch of columns and has the next column's loads in flight while it counts the current one.  This is the
// same stream for mixed operands.  The difficulty is that a pipelined loop must issue the SAME number of loads in every
// iteration (hipcc joins wait counters to vmcnt(0) where paths with different numbers of outstanding loads meet), and a
// column needs 8 loads per bit-block, 2 per GAP block, none for NULL / FULL.  So a wave first reads the <= 64 descriptor
// pairs of its stretch (one coalesced load), sorts its columns by LOAD SHAPE with ballots, and walks each of the nine
// shapes (bit / GAP / none on either side) with its own two-deep pipelined loop.  GAP blocks are prefetched into
// registers (two 16-byte chunks per lane = 1,023 words; longer blocks fetch their tail when they are decoded) and
// decoded from there into the wave's 8 KiB of LDS.  Same result: a sum of per-column popcounts.
// ---------------------------------------------------------------------------

// decode a GAP block whose first 1,024 words are in registers (c0 = words 8*lane.., c1 = words 8*(lane+64)..) into `out`
__device__ __forceinline__ void gap_decode_regs(u32x4 c0, u32x4 c1, u64 d, u32* lds, Blk& out, u32 lane)
{
    const u32 meta = GMETA(d), len = meta >> 1, sbit = meta & 1u;
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 64 + lane] = (u32x4)(0u);
    const u32 nch = (len + 8u) >> 3;
    u32 carry = 0u;
    u64 longm = 0ull; u32 lws = 0, lwe = 0;                      // (at most one queued long interior per lane and chunk round)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (64u * (u32)j >= nch) break;                           // wave-uniform
        const u32 c = lane + 64u * (u32)j;
        u32x4 q = j == 0 ? c0 : c1;
        if (j == 2) q = c < nch ? as_gc4(DESC_P(d))[c] : (u32x4)(0u);       // a block of more than 1,023 words: its tail, fetched now
        const u32 x[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
        u32 prev = __shfl_up(x[7], 1, 64);
        if (lane == 0) prev = carry;
        carry = __builtin_amdgcn_readlane(x[7], 63);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // word k of the block = end of run k; run k has the value sbit ^ ((k - 1) & 1): the 1-runs are the odd k when
            // sbit = 1 and the even k when sbit = 0 (8c is even)
            u32 k, sraw, e;
            if (sbit) { k = 8u * c + 2u * (u32)t + 1u; sraw = x[2 * t]; e = x[2 * t + 1]; }
            else      { k = 8u * c + 2u * (u32)t;      sraw = t == 0 ? prev : x[2 * t - 1]; e = x[2 * t]; }
            const bool ok = c < nch && k >= 1u && k <= len;
            const u32 s = k == 1u ? 0u : sraw + 1u;
            const u32 ws = s >> 5, we = e >> 5;
            const u32 lo = ~0u << (s & 31u), hi = ~0u >> (31u - (e & 31u));
            if (ok) {
                if (ws == we) atomicOr(&lds[ws], lo & hi);
                else {
                    atomicOr(&lds[ws], lo); atomicOr(&lds[we], hi);
                    if (we - ws > 1u) {
                        if (we - ws <= 9u) { for (u32 w = ws + 1u; w < we; ++w) lds[w] = ~0u; }   // all-ones is absorbing: plain stores
                        else { lws = ws; lwe = we; }
                    }
                }
            }
            // long interiors: the whole wave fills them, one run at a time
            u64 m = __ballot(ok && ws != we && we - ws > 9u);
            while (m) {
                const u32 src = (u32)__builtin_ctzll(m);
                m &= m - 1ull;
                const u32 a = __builtin_amdgcn_readlane(lws, src), b = __builtin_amdgcn_readlane(lwe, src);
                for (u32 w = a + 1u + lane; w < b; w += 64u) lds[w] = ~0u;
            }
        }
    }
    (void)longm;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.r[i] = l4[i * 64 + lane];
}

enum { SH_BIT = 0, SH_GAP = 1, SH_NONE = 2 };

template <int SH>
__device__ __forceinline__ void mixed_load(Blk& x, u64 d, u32 lane)
{
    if constexpr (SH == SH_BIT) part_load<8, true>(x, as_gc4(DESC_P(d)), lane);
    else if constexpr (SH == SH_GAP) {
        const u32 len = GMETA(d) >> 1, nch = (len + 8u) >> 3;     // >= 1
        gcptr4 g4 = as_gc4(DESC_P(d));
        const u32 i0 = lane < nch ? lane : nch - 1u, i1 = lane + 64u < nch ? lane + 64u : nch - 1u;     // unconditional, clamped
        x.r[0] = g4[i0]; x.r[1] = g4[i1];
    }
}

template <int SH>
__device__ __forceinline__ void mixed_finish(Blk& x, u64 d, u32* lds, u32 lane)
{
    if constexpr (SH == SH_GAP) { u32x4 c0 = x.r[0], c1 = x.r[1]; gap_decode_regs(c0, c1, d, lds, x, lane); }
    else if constexpr (SH == SH_NONE) blk_fill(x, DESC_K(d) == K_FULL ? ~0u : 0u);
}

// all columns of one load shape: bits of `mask` = lanes (columns of the stretch) to visit; da / db = the lane-held descriptors
template <int SA, int SB>
__device__ __forceinline__ u32 mixed_walk(u64 mask, u64 dav, u64 dbv, int op, u32* lds, u32 lane)
{
    u32 cnt = 0;
    if (!mask) return 0u;
    Blk x0, y0, x1, y1;
    u32 i0 = (u32)__builtin_ctzll(mask); mask &= mask - 1ull;
    u64 a0 = readlane64(dav, i0), b0 = readlane64(dbv, i0);
    mixed_load<SA>(x0, a0, lane); mixed_load<SB>(y0, b0, lane);
    for (;;) {
        // the next column of this shape (or, at the end, the current one again: its lines are in the L2, the loads keep
        // the iteration's load count uniform)
        const bool more1 = mask != 0ull;
        const u32 i1 = more1 ? (u32)__builtin_ctzll(mask) : i0;
        mask &= mask - 1ull;
        const u64 a1 = readlane64(dav, i1), b1 = readlane64(dbv, i1);
        mixed_load<SA>(x1, a1, lane); mixed_load<SB>(y1, b1, lane);
        mixed_finish<SA>(x0, a0, lds, lane); mixed_finish<SB>(y0, b0, lds, lane);
        blk_op(op, x0, y0);
        cnt += blk_lane_popcount(x0);
        if (!more1) break;
        const bool more0 = mask != 0ull;
        i0 = more0 ? (u32)__builtin_ctzll(mask) : i1;
        mask &= mask - 1ull;
        a0 = readlane64(dav, i0); b0 = readlane64(dbv, i0);
        mixed_load<SA>(x0, a0, lane); mixed_load<SB>(y0, b0, lane);
        mixed_finish<SA>(x1, a1, lds, lane); mixed_finish<SB>(y1, b1, lds, lane);
        blk_op(op, x1, y1);
        cnt += blk_lane_popcount(x1);
        if (!more0) break;
    }
    return cnt;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void k_count_op2_mixed(int op, const u64* __restrict__ da, u32 na, const u64* __restrict__ db, u32 nbk, u32 nblocks, u32 per_wave /* <= 64 */,
                       FoldOut fold)
{
    __shared__ u32 lds_all[WAVES * 2048];
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32* lds = lds_all + wave * 2048u;
    const u32 w = uniform32(blockIdx.x * (u32)WAVES + wave);
    const u32 c0 = w * per_wave;
    u32 cnt = 0;
    if (c0 < nblocks) {
        const u32 n = c0 + per_wave < nblocks ? per_wave : nblocks - c0;
        const u32 c = c0 + lane;
        const bool in = lane < n;
        u64 a = (in && c < na) ? da[c] : 0ull, b = (in && c < nbk) ? db[c] : 0ull;
        const u32 ka = DESC_K(a), kb = DESC_K(b);
        // columns that cannot contribute (combine_count_operation_with_block rules, src/bmalgo_impl.h:189-434, :873-894)
        const bool skip = !in || (ka == K_NULL && kb == K_NULL) || (op == BMX_AND && (ka == K_NULL || kb == K_NULL)) ||
                          (op == BMX_SUB && (ka == K_NULL || kb == K_FULL));
        const u32 sa = ka == K_BIT ? SH_BIT : (ka == K_GAP ? SH_GAP : SH_NONE), sb = kb == K_BIT ? SH_BIT : (kb == K_GAP ? SH_GAP : SH_NONE);
        const u32 shape = skip ? 9u : sa * 3u + sb;
#define WALK(SA, SB) cnt += mixed_walk<SA, SB>(__ballot(shape == (u32)(SA * 3 + SB)), a, b, op, lds, lane)
        WALK(SH_BIT, SH_BIT); WALK(SH_BIT, SH_GAP); WALK(SH_GAP, SH_BIT); WALK(SH_GAP, SH_GAP);
        WALK(SH_BIT, SH_NONE); WALK(SH_NONE, SH_BIT); WALK(SH_GAP, SH_NONE); WALK(SH_NONE, SH_GAP); WALK(SH_NONE, SH_NONE);
#undef WALK
        cnt = wave_sum(cnt);
    }
    count_fanin_fold(cnt, fold, lane, wave);
}
