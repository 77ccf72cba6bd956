// bmx_kernels3.h -- aggregator::combine_shift_right_and: the sequence-search
// primitive, the only aggregator operation with a cross-block dependency.
#pragma once
#include "bmx_kernels2.h"

// ---------------------------------------------------------------------------
// Reference (src/bmaggregator.h:2494-2669): block columns are visited in order; per column
//   T_0 = src[0];   T_k = shift_right_1(T_{k-1}, carry_in = carry_overs[k]) & src[k],  k = 1..n-1
// where carry_overs[k] is the bit stage k shifted out of the PREVIOUS column (process_shift_right_and
// :2618, bit_block_shift_r1_and_unr).  "Right" moves bit p to p+1.  Unrolled over the whole vector:
//   R[p] = AND_k src[k][p - (n-1-k)]      (bits from below position 0 are 0)
// so result block nb is the AND over k of a 65,536-bit WINDOW of src[k] that starts s_k = n-1-k bits
// below the block: the top s_k bits of block nb-1 followed by the low part of block nb (s_k < 65536;
// in general blocks nb-q-1 and nb-q with s_k = 65536 q + r).  The serial carry chain disappears: every
// column is independent again, one wave per column, early exit when the running AND is zero (the
// reference's digest test, :2562).
//
// Window construction: r == 0 -> the block itself.  Otherwise the needed tail of the lower block and
// the upper block are staged in a 16 KiB LDS window (any block kind: GAP decoded, FULL ones, NULL
// zeros) and each lane reads its 32 words back at a -r bit offset (two LDS words + a 64-bit shift).
// Fast path for r < 32 with a bit-block on top (patterns up to 32 symbols, the common case): the
// funnel shift is done in registers, the word below comes from the neighbouring lane.
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) u32* gcptr32;

// last word (bits 65504..65535) of any block kind
__device__ __forceinline__ u32 blk_last_word(u64 d)
{
    u32 k = DESC_K(d);
    if (k == K_BIT) return ((gcptr32)(uintptr_t)DESC_P(d))[2047];
    if (k == K_FULL) return ~0u;
    return 0u;                                               // NULL (GAP is routed to the LDS path)
}

// out = window of (d, nd) for result block nb and shift s; false when the window is all zero (uniform)
__device__ __forceinline__ bool shifted_window(const u64* __restrict__ d, u32 nd, u32 nb, u32 s,
                                               Blk& out, u32* W, u32 lane)
{
    u32 q = s >> 16, r = s & 65535u;
    if (nb < q) return false;
    u32 cb = nb - q;
    u64 dc = desc_at(d, nd, cb);
    u64 dp = (r && cb) ? desc_at(d, nd, cb - 1u) : 0ull;
    u32 kc = DESC_K(dc), kp = DESC_K(dp);
    if (kc == K_NULL && kp == K_NULL) return false;
    if (r == 0u) { blk_from_desc(dc, out, W + 2048, lane); return true; }
    u32 wq = r >> 5, wr = r & 31u;
    if (wq == 0u && kc == K_BIT && kp != K_GAP) {            // registers only
        Blk c;
        blk_load(c, as_gc4(DESC_P(dc)), lane);
        u32 below = uniform32(blk_last_word(dp));            // word under row 0 / lane 0
        u32 sh = 32u - wr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32 up = __shfl_up(c.r[i].w, 1, 64);             // previous lane's top word
            u32 row_last = __shfl(c.r[i].w, 63, 64);
            if (lane == 0) up = below;
            out.r[i].x = (c.r[i].x << wr) | (up >> sh);
            out.r[i].y = (c.r[i].y << wr) | (c.r[i].x >> sh);
            out.r[i].z = (c.r[i].z << wr) | (c.r[i].y >> sh);
            out.r[i].w = (c.r[i].w << wr) | (c.r[i].z >> sh);
            below = row_last;
        }
        return true;
    }
    // general: stage [tail of the lower block | upper block] in LDS
    if (kp == K_GAP) { Blk t; gap_decode(as_gc16(DESC_P(dp)), W, t, lane, GMETA(dp)); blk_to_lds(t, W, lane); }
    else if (kp == K_BIT) {
        gcptr32 p = (gcptr32)(uintptr_t)DESC_P(dp);
        for (u32 t = lane; t <= wq; t += 64u) W[2047u - wq + t] = p[2047u - wq + t];
    } else {
        u32 fill = kp == K_FULL ? ~0u : 0u;
        for (u32 t = lane; t <= wq; t += 64u) W[2047u - wq + t] = fill;
    }
    {
        Blk c;
        blk_from_desc(dc, c, W + 2048, lane);
        blk_to_lds(c, W + 2048, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 sh = 32u - wr;                                       // 1..32: 64-bit shift keeps wr == 0 exact
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 idx = 2048u + (u32)i * 256u + lane * 4u - wq;
        u32 w[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) w[c] = W[idx - 1u + (u32)c];
        out.r[i].x = (u32)((((u64)w[1] << 32) | w[0]) >> sh);
        out.r[i].y = (u32)((((u64)w[2] << 32) | w[1]) >> sh);
        out.r[i].z = (u32)((((u64)w[3] << 32) | w[2]) >> sh);
        out.r[i].w = (u32)((((u64)w[4] << 32) | w[3]) >> sh);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// One wave per result block column.  count_only mirrors set_compute_count(true) (:2595): no target,
// the popcount goes through the striped fan-in.  Operands are visited from the last one (shift 0,
// the plain block: result is a subset of it) towards the first.
__global__ __launch_bounds__(256)
void k_shift_right_and(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 n, u32 ncols,
                       int opt_compress, int count_only, int xcd_swz,
                       uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                       u64* __restrict__ slots)
{
    extern __shared__ u32 lds_dyn[];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 nb = uniform32(bid * 4u + wave);
    u32* W = lds_dyn + wave * 4096u;
    u32 cnt = 0;
    if (nb < ncols) {
        Blk acc;
        blk_fill(acc, ~0u);
        bool zero = false;
        for (u32 k = n; k-- > 0u && !zero; ) {
            Blk wdw;
            const u64* d = (const u64*)uniform64((u64)(uintptr_t)descs[k]);
            u32 nd = uniform32(nblk[k]);
            if (!shifted_window(d, nd, nb, n - 1u - k, wdw, W, lane)) { zero = true; break; }
            blk_and(acc, wdw);
            zero = blk_is_zero(acc);
        }
        if (count_only) { if (!zero) cnt = wave_sum(blk_lane_popcount(acc)); }
        else if (zero) store_trivial(K_NULL, nb, desc, st, lane);
        else store_result(acc, nb, opt_compress, slab, desc, st, lane, !opt_compress);   // opt_none: copy_bit_block (:2600)
    }
    if (count_only) count_fanin(cnt, slots, lane, wave);
}

// `any` form (:2519): the reference returns at the first column that produced a block, so the target
// holds exactly that block.  All columns were computed in parallel here; keep the first, drop the rest.
__global__ __launch_bounds__(1024)
void k_keep_first_block(BlockStat* __restrict__ st, u64* __restrict__ desc, u32 nblocks)
{
    __shared__ u32 first;
    if (threadIdx.x == 0) first = 0xFFFFFFFFu;
    __syncthreads();
    for (u32 nb = threadIdx.x; nb < nblocks; nb += 1024u)
        if (st[nb].kind != K_NULL) atomicMin(&first, nb);
    __syncthreads();
    u32 f = first;
    for (u32 nb = threadIdx.x; nb < nblocks; nb += 1024u)
        if (nb != f && st[nb].kind != K_NULL) { st[nb] = BlockStat{0u, 1u, 0u, K_NULL}; desc[nb] = DESC_MAKE(0, K_NULL); }
}
