// bmx_kernels5.h -- counts-only AND / AND-SUB over columns whose operands are ALL GAP blocks: a counting formulation.
#pragma once
#include "bmx_kernels4.h"

// ---------------------------------------------------------------------------
// The reference applies GAP operands one after the other to the accumulator (process_gap_blocks_and / _sub,
// src/bmaggregator.h:1820,1854 -> gap_and_to_bitset / gap_sub_to_bitset, src/bmfunc.h:4893,4800); the wave64 counterpart of
// that (bmx_device.h gap_apply_list: run application, then membership tests of the surviving bits) costs ~300 wave
// instructions per operand and is VALU-bound.  Same result, different arithmetic: COUNT how many
// operands cover each position.  Two byte-counter arrays in LDS -- S[p] = 1-runs starting at p, E[p] = 1-runs ending at p --
// take two LDS atomics per 1-run whatever its length; cover(p) = sum_{q<=p} S[q] - sum_{q<p} E[q] is one prefix scan over
// the block; the AND of n operands is cover == n, the SUB group is cover != 0.
// A 1024-thread workgroup owns a (column, group); the first AND operand is decoded into the bitmap B, the others are
// counted in chunks of <= 255 (byte counters); run ends are fetched with 16-byte loads, 8 ends per lane, GC_AHEAD (6)
// operands ahead.  LDS: S 64 KiB + E 64 KiB + B 8 KiB.
// ---------------------------------------------------------------------------
#define GC_WAVES 16u
#define GC_AHEAD 6

// the 1-runs held by one 16-byte chunk (words 8c .. 8c+7 of the block; word 0 is the header, word k = end of run k)
__device__ __forceinline__ void gc_count_chunk(u32* S, u32* E, u32x4 d, u32 c, u32 prev, u32 len, u32 sbit, bool act)
{
    u32 x[8] = {d.x & 0xFFFFu, d.x >> 16, d.y & 0xFFFFu, d.y >> 16, d.z & 0xFFFFu, d.z >> 16, d.w & 0xFFFFu, d.w >> 16};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        // run k has the value sbit ^ ((k - 1) & 1); 8c is even, so the 1-runs sit at the odd words when sbit = 1, at the even ones when sbit = 0
        u32 k, sraw, e;
        if (sbit) { k = 8u * c + 2u * (u32)t + 1u; sraw = x[2 * t]; e = x[2 * t + 1]; }
        else      { k = 8u * c + 2u * (u32)t;      sraw = t == 0 ? prev : x[2 * t - 1]; e = x[2 * t]; }
        bool ok = act && k >= 1u && k <= len;
        u32 s = k == 1u ? 0u : sraw + 1u;
        if (ok) {
            atomicAdd(&S[s >> 2], 1u << ((s & 3u) << 3));
            atomicAdd(&E[e >> 2], 1u << ((e & 3u) << 3));
        }
    }
}

// two 16-byte chunks per lane cover blocks of up to 1,023 words (every block of the sparse regime); the third chunk of a
// longer block is fetched when the operand is counted
__device__ __forceinline__ void gc_stage_load(u32x4 (&d)[2], u64 gaddr, u32 len, u32 lane)
{
    gcptr4 g4 = as_gc4(gaddr & 0x0000FFFFFFFFFFFFull);
    u32 nch = len ? (len + 8u) >> 3 : 0u;                       // (len 0 = an empty slot of the prefetch ring: nothing is read)
#pragma unroll
    for (int j = 0; j < 2; ++j) { u32 c = lane + 64u * (u32)j; d[j] = c < nch ? g4[c] : (u32x4)(0u); }
}

__device__ __forceinline__ void gc_count_operand(u32* S, u32* E, const u32x4 (&d)[2], u64 gaddr, u32 len, u32 sbit, u32 lane)
{
    u32 nch = (len + 8u) >> 3;
    u32 carry = 0u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (64u * (u32)j >= nch) break;                           // wave-uniform
        u32 c = lane + 64u * (u32)j;
        u32x4 q;
        if (j < 2) q = d[j];
        else q = c < nch ? as_gc4(gaddr & 0x0000FFFFFFFFFFFFull)[c] : (u32x4)(0u);         // long block: its third chunk, fetched now
        u32 last = q.w >> 16;
        u32 prev = __shfl_up(last, 1, 64);
        if (lane == 0) prev = carry;
        carry = __builtin_amdgcn_readlane(last, 63);
        gc_count_chunk(S, E, q, c, prev, len, sbit, c < nch);
    }
}

// counts operands [i0, i0 + n) of a GAP pointer list (walked backwards from list_back) into S / E
__device__ __forceinline__ void gc_accumulate(u32* S, u32* E, const u64* __restrict__ list_back, u32 i0, u32 n, u32 lane, u32 wave)
{
    // my operands: i0 + wave, i0 + wave + 16, ...  (<= 16 of them for n <= 255); their headers in one round trip
    u32 mine = wave < n ? (n - wave + GC_WAVES - 1u) / GC_WAVES : 0u;
    u64 ptr = 0ull; u32 hdr = 0u;
    if (lane < mine) { ptr = *(list_back - (i0 + wave + GC_WAVES * lane)); hdr = (u32)(*as_gc16(ptr)); }
    u32x4 st[GC_AHEAD][2];
    u32 ln[GC_AHEAD], sb[GC_AHEAD]; u64 pp[GC_AHEAD];
#pragma unroll
    for (int a = 0; a < GC_AHEAD; ++a) {
        u32 j = (u32)a < mine ? (u32)a : 0u;
        u64 p = readlane64(ptr, j); u32 h = __builtin_amdgcn_readlane(hdr, j);
        ln[a] = (u32)a < mine ? h >> 3 : 0u; sb[a] = h & 1u; pp[a] = p;
        gc_stage_load(st[a], p, ln[a], lane);
    }
    for (u32 j0 = 0; j0 < mine; j0 += GC_AHEAD) {
#pragma unroll
        for (int a = 0; a < GC_AHEAD; ++a) {
            if (j0 + (u32)a < mine) gc_count_operand(S, E, st[a], pp[a], ln[a], sb[a], lane);
            u32 jn = j0 + (u32)a + GC_AHEAD;                        // refill the slot with the operand GC_AHEAD further on
            bool more = jn < mine;
            u32 jj = more ? jn : 0u;
            u64 p = readlane64(ptr, jj); u32 h = __builtin_amdgcn_readlane(hdr, jj);
            ln[a] = more ? h >> 3 : 0u; sb[a] = h & 1u; pp[a] = p;
            gc_stage_load(st[a], p, ln[a], lane);
        }
    }
}

// cover(p) from S / E, combined into the bitmap B: SUBTRACT = false: B &= (cover == n); true: B &= ~(cover != 0).
// returns (block-wide) whether B still holds a bit
template <bool SUBTRACT>
__device__ __forceinline__ bool gc_scan_combine(const u32* S, const u32* E, u32* B, u32 n, int* scan_sm, u32* flag, u32 tid)
{
    const u32 lane = tid & 63u, wave = tid >> 6;
    u32 sw[16], ew[16];
    const u32x4* S4 = reinterpret_cast<const u32x4*>(S) + tid * 4u;
    const u32x4* E4 = reinterpret_cast<const u32x4*>(E) + tid * 4u;
    int delta = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32x4 a = S4[q], b = E4[q];
        sw[4 * q] = a.x; sw[4 * q + 1] = a.y; sw[4 * q + 2] = a.z; sw[4 * q + 3] = a.w;
        ew[4 * q] = b.x; ew[4 * q + 1] = b.y; ew[4 * q + 2] = b.z; ew[4 * q + 3] = b.w;
    }
#pragma unroll
    for (int w = 0; w < 16; ++w) delta += (int)__builtin_amdgcn_sad_u8(sw[w], 0u, 0u) - (int)__builtin_amdgcn_sad_u8(ew[w], 0u, 0u);
    int incl = (int)wave_scan_incl((u32)delta, lane);
    if (lane == 63) scan_sm[wave] = incl;
    if (tid == 0) *flag = 0u;
    __syncthreads();
    int running = incl - delta;
#pragma unroll
    for (u32 i = 0; i < GC_WAVES; ++i) if (i < wave) running += scan_sm[i];
    u32 bits[2] = {0u, 0u};
#pragma unroll
    for (int w = 0; w < 16; ++w) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            running += (int)((sw[w] >> (8 * b)) & 255u);
            bool m = SUBTRACT ? (running != 0) : (running == (int)n);
            bits[w >> 3] |= (m ? 1u : 0u) << (((w & 7) << 2) + b);
            running -= (int)((ew[w] >> (8 * b)) & 255u);
        }
    }
    u32 b0 = B[2u * tid], b1 = B[2u * tid + 1u];
    b0 = SUBTRACT ? (b0 & ~bits[0]) : (b0 & bits[0]);
    b1 = SUBTRACT ? (b1 & ~bits[1]) : (b1 & bits[1]);
    B[2u * tid] = b0; B[2u * tid + 1u] = b1;
    if (__ballot((b0 | b1) != 0u) != 0ull && lane == 0) *flag = 1u;
    __syncthreads();
    return *flag != 0u;
}

// STORE: the single-group materialising form (combine_and_sub over GAP-only operands): the column's bitmap is stored as a
// result block (opt_compress, src/bmaggregator.h:1210) instead of being counted; columns outside [hint_from, hint_to) are NULL
template <bool STORE>
__global__ __launch_bounds__(1024)
void k_pipe_counts_gapcount(const u64* __restrict__ dmat, const u32* __restrict__ row_off, const u32* __restrict__ and_n,
                            const u32* __restrict__ sub_n, u32 col_stride, u32 ngroups, u32 col_from, u32 nitems,
                            u64* __restrict__ counts, uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                            u32 hint_from, u32 hint_to)
{
    extern __shared__ u32 lds_dyn[];
    u32* S = lds_dyn;                       // 65,536 byte counters
    u32* E = S + 16384u;
    u32* B = E + 16384u;                    // the accumulator: 2,048 words
    __shared__ int scan_sm[GC_WAVES];
    __shared__ u32 flag;
    __shared__ u32 part[GC_WAVES];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 item = blockIdx.x;
    if (item >= nitems) return;
    const u32 c = item / ngroups, g = item - c * ngroups;
    const u32 col = col_from + c;
    if (STORE && (col < hint_from || col >= hint_to)) { if (wave == 0) store_trivial(K_NULL, col, desc, st, lane); return; }
    const u64* row = dmat + (size_t)col * col_stride + row_off[g];
    const u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) { if (STORE && wave == 0) store_trivial(K_NULL, col, desc, st, lane); return; }
    if (flags & ROW_FULL) {
        if (STORE) { if (wave == 0) store_trivial(K_FULL, col, desc, st, lane); }
        else if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), 65536ull);
        return;
    }
    const u32 nga = (u32)((hdr >> 16) & 0xFFFFu), ngs = (u32)(hdr >> 48);       // (no bit-block operands in these pipelines)
    const u32 na = uniform32(and_n[g]), ns = uniform32(sub_n[g]);
    const u64* pa_back = row + 2 + na - 1u;                                     // GAP pointers are packed from the back of a region
    const u64* ps_back = row + 2 + na + ns - 1u;
    // accumulator = the first AND operand (or all ones: only FULL operands in the AND group): wave 0 sets its 1-runs in B
    // while the workgroup zeroes the counters of the first chunk
    u32x4* Z = reinterpret_cast<u32x4*>(S);                                      // S and E are contiguous: 8,192 x 16 B
    B[2u * tid] = nga ? 0u : ~0u; B[2u * tid + 1u] = nga ? 0u : ~0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) Z[(u32)q * 1024u + tid] = (u32x4)(0u);
    __syncthreads();
    if (nga && wave == 0) gap_apply_lds_wave<GAP_OR>(as_gc16(uniform64(*pa_back)), B, lane);
    bool alive = true;
    bool zeroed = true;
    for (u32 i0 = 1; i0 < nga && alive; i0 += 255u) {
        const u32 n = nga - i0 < 255u ? nga - i0 : 255u;
        if (!zeroed) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Z[(u32)q * 1024u + tid] = (u32x4)(0u);
            __syncthreads();
        }
        zeroed = false;
        gc_accumulate(S, E, pa_back, i0, n, lane, wave);
        __syncthreads();
        alive = gc_scan_combine<false>(S, E, B, n, scan_sm, &flag, tid);
    }
    if (zeroed) __syncthreads();                                                 // (a single AND operand: B is complete after this)
    for (u32 i0 = 0; i0 < ngs && alive; i0 += 255u) {
        const u32 n = ngs - i0 < 255u ? ngs - i0 : 255u;
        if (!zeroed) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Z[(u32)q * 1024u + tid] = (u32x4)(0u);
            __syncthreads();
        }
        zeroed = false;
        gc_accumulate(S, E, ps_back, i0, n, lane, wave);
        __syncthreads();
        alive = gc_scan_combine<true>(S, E, B, n, scan_sm, &flag, tid);
    }
    if (STORE) {
        if (wave == 0) {
            if (!alive) store_trivial(K_NULL, col, desc, st, lane);
            else { Blk t; blk_from_lds(t, B, lane); store_result(t, col, 1, slab, desc, st, lane); }
        }
        return;
    }
    if (!alive) return;
    u32 cnt = wave_sum((u32)__popc(B[2u * tid]) + (u32)__popc(B[2u * tid + 1u]));
    if (lane == 0) part[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        u32 t = 0;
        for (u32 i = 0; i < GC_WAVES; ++i) t += part[i];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)t);
    }
}
