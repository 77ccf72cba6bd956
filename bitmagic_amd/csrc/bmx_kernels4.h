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
enum { CMP_GT = 0, CMP_GE = 1, CMP_LT = 2, CMP_LE = 3, CMP_RANGE = 4, CMP_EQ = 5, CMP_ZERO = 6, CMP_NONZERO = 7 };

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
template <bool TWO>
__global__ __launch_bounds__(256)
void k_slice_compare(const u64* const* __restrict__ descs /* per plane: descriptor table or null */,
                     const u32* __restrict__ nblk, u32 nplanes, u32 ncols, int pred, u64 v0, u64 v1, u64 size,
                     const u64* __restrict__ nn_desc /* not-NULL vector or null */, u32 nn_blocks, int null_correct,
                     int count_only, int xcd_swz,
                     uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, u64* __restrict__ slots)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 nb = uniform32(bid * 4u + wave);
    u32 cnt = 0;
    if (nb < ncols) {
        constexpr bool two = TWO;
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
        default:                                                         // NONZERO
#pragma unroll
            for (int i = 0; i < 8; ++i) res.r[i] = ~eq0.r[i];
            break;
        }
        Blk m;
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
