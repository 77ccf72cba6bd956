// bmx_kernels11.h -- round 6: select lines.  bvector::select / find_rank (src/bm.h:5350, rs_index::find src/bmrs.h:492,
// bit_find_rank src/bmfunc.h:8500) with ONE 128-byte line per query and no search.
#pragma once
#include "bmx_kernels10.h"

// ---------------------------------------------------------------------------
// Why.  k_select_top (bmx_kernels6.h) guesses the rank line of one number r by interpolating between two directory entries
// and lets the line's header decide.  The round-5 counters (profiles/r05final/pmc_config3.txt) show 1.8 missed lines per query:
// the guess is a line off for about every second query -- the directory knows the LINE of its sampled ones, not where in the
// line they sit (half a line of bias), and between two samples the ones of a Bernoulli vector wander by ~0.35 lines at 10 %
// density, ~1.2 lines at 1 % (a sample every ~85 lines is all that 128 KiB of LDS hold).  No directory of that size removes the
// second line; the lever is the same one rank pulled in round 3 -- lay the vector out ONCE MORE in the form the query wants:
//
//   select line k = 128 bytes = [ u64: position of one number K k ][ K offsets: the low 16 / 32 bits of the positions of ones
//                   K k .. K k + K - 1 ]          K = 60 with 16-bit offsets, 30 with 32-bit offsets
//
//   select(r):  g = r - 1;  k = g / K;  i = g - K k;   pos = base_k + ((low_i - low(base_k)) mod 2^16|32)
//
// One line, two loads by ONE lane, no header test, no retry: the ones are addressed, not searched.  A stored offset is the low
// bits of the ABSOLUTE position (for 16 bits: the offset inside its 64 Kbit block), so the build needs no second pass to learn a
// line's base; the difference is exact whenever the K ones of a line span fewer than 2^16 (2^32) bits.  k_rs_sel_check verifies
// that for every line against the running block counts; one line that fails makes build_rs_index take the 32-bit form for
// the whole vector (a 32-bit form fails only past 2^32 bits, BM64ADDR: then the vector keeps the directory kernels).
// Cost: 128 / K bytes per one = 2.13 B (4.27 B) -- 853 MB for the 4e8 ones of configs[3] at 10 %, 85 MB at 1 %.  Memory
// policy (rs_select_sel -1): built where that is <= 2 x what the vector and its rank lines hold on the device; dense vectors
// (> ~12 % ones) keep k_select_top, whose interpolation error shrinks with the density anyway.
// Results are the reference's by construction: the stored positions ARE the set bits in order (the enumeration
// k_expand_indices / bvector::enumerator gives), select(r) reads the r-th.
// ---------------------------------------------------------------------------
#define SL_BYTES 128u
template <typename T> struct SelFmt;
template <> struct SelFmt<u16> { static constexpr u32 K = 60u, UNIT = 16u; };
template <> struct SelFmt<u32> { static constexpr u32 K = 30u, UNIT = 32u; };

// Build: one wave per block (any kind).  The ones of a row of the block (256 words) get their global numbers from the wave's
// popcount prefix; their low bits are staged in LDS in that order (the 8 KiB the GAP decoder used), and the wave writes the
// staged run out front to back -- 16-bit offsets in aligned pairs: 256 contiguous bytes per store instruction instead of 64
// two-byte stores strewn over a dozen lines (the first form of this kernel: 1.3 ms for configs[3], now the write of 853 MB
// and the read of 500 MB at stream speed).  A line's base is written by whoever holds its slot 0.
template <typename T>
__device__ __forceinline__ void sel_stage_word(u32 w, u32 lowbase, u32& idx, u32 cb, T* __restrict__ stage, u32 cap)
{
    while (w) {
        const u32 b = (u32)__builtin_ctz(w);
        const u32 at = idx - cb;
        if (at < cap) stage[at] = (T)(lowbase + b);
        ++idx; w &= w - 1u;
    }
}

// rcount = inclusive running count per block
template <typename T>
__global__ __launch_bounds__(256)
void k_rs_sel_build(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, u8* __restrict__ sel)
{
    constexpr u32 K = SelFmt<T>::K;
    constexpr u32 CAP = 8192u / (u32)sizeof(T);                           // staged ones per pass
    __shared__ u32 lds[4 * 2048];
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    const u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks) return;
    const u64 d = uniform64(desc[nb]);
    if (DESC_K(d) == K_NULL) return;
    const u64 first = nb ? rcount[nb - 1u] : 0ull;
    const u64 bit0 = (u64)nb << 16;
    const u64 hi = sizeof(T) == 2 ? bit0 : (bit0 & 0xFFFFFFFF00000000ull);     // pos = hi + the stored low bits
    Blk b;
    blk_from_desc(d, b, lds + wave * 2048u, lane);
    T* stage = reinterpret_cast<T*>(lds + wave * 2048u);
    // (one 64-bit division per block: the ones of the block are numbered from the block's first line in 32 bits)
    const u64 line0 = first / K;
    const u32 rem0 = (u32)(first - line0 * K);
    u8* const sel0 = sel + line0 * SL_BYTES;
    u32 row_off = rem0;                                                   // number of the row's first one, counted from line0's slot 0
#pragma unroll                                                           // (a row index known at compile time: the block image stays in registers)
    for (int i = 0; i < 8; ++i) {
        // row i = words i*256 .. i*256+255; lane l holds the four consecutive words i*256 + 4l .. + 3
        const u32 c = (u32)__popc(b.r[i].x) + (u32)__popc(b.r[i].y) + (u32)__popc(b.r[i].z) + (u32)__popc(b.r[i].w);
        const u32 incl = wave_scan_incl(c, lane);
        const u32 tot = uniform32(__shfl(incl, 63, 64));
        const u32 lowbase = (u32)(bit0 + ((u64)((u32)i * 256u + lane * 4u) << 5));   // low 32 bits of the position of the lane's first bit
        for (u32 cb = 0; cb < tot; cb += CAP) {
            u32 idx = incl - c;
            if (c && idx < cb + CAP && idx + c > cb) {
                sel_stage_word<T>(b.r[i].x, lowbase, idx, cb, stage, CAP);
                sel_stage_word<T>(b.r[i].y, lowbase + 32u, idx, cb, stage, CAP);
                sel_stage_word<T>(b.r[i].z, lowbase + 64u, idx, cb, stage, CAP);
                sel_stage_word<T>(b.r[i].w, lowbase + 96u, idx, cb, stage, CAP);
            }
            const u32 n = tot - cb < CAP ? tot - cb : CAP;
            const u32 g0 = row_off + cb;
            if constexpr (sizeof(T) == 2) {
                const u32 e0 = g0 & 1u;                                   // slot parity = one-number parity (K is even, and so is line0 K)
                auto single = [&](u32 e) {
                    const u32 g = g0 + e; const u32 line = g / K; const u32 slot = g - line * K;
                    u8* L = sel0 + (size_t)line * SL_BYTES;
                    reinterpret_cast<u16*>(L + 8)[slot] = stage[e];
                    if (slot == 0u) *reinterpret_cast<u64*>(L) = hi + stage[e];
                };
                if (e0 && lane == 0u) single(0u);
                const u32 np = (n - e0) >> 1;
                for (u32 p = lane; p < np; p += 64u) {
                    const u32 e = e0 + 2u * p;
                    const u32 g = g0 + e; const u32 line = g / K; const u32 slot = g - line * K;
                    u8* L = sel0 + (size_t)line * SL_BYTES;
                    const u32 lo = stage[e], hi16 = stage[e + 1u];
                    *reinterpret_cast<u32*>(L + 8 + slot * 2u) = lo | (hi16 << 16);
                    if (slot == 0u) *reinterpret_cast<u64*>(L) = hi + lo;
                }
                if (((n - e0) & 1u) && lane == 63u) single(n - 1u);
            } else {
                for (u32 e = lane; e < n; e += 64u) {
                    const u32 g = g0 + e; const u32 line = g / K; const u32 slot = g - line * K;
                    u8* L = sel0 + (size_t)line * SL_BYTES;
                    reinterpret_cast<u32*>(L + 8)[slot] = stage[e];
                    if (slot == 0u) *reinterpret_cast<u64*>(L) = hi + stage[e];
                }
            }
        }
        row_off += tot;
    }
}

// every line's ones must span fewer than 2^UNIT bits.  The stored low bits locate a one inside its 2^UNIT-bit unit (a 64 Kbit
// block for 16 bits, 65,536 blocks for 32); the unit the line's LAST one lies in follows from the running block counts.
template <typename T>
__global__ __launch_bounds__(256)
void k_rs_sel_check(const u8* __restrict__ sel, u64 nsel, u64 count, const u64* __restrict__ rcount, u32 nblocks, u32* __restrict__ bad)
{
    constexpr u32 K = SelFmt<T>::K, BPU = SelFmt<T>::UNIT - 16u;         // log2(blocks per unit)
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nsel) return;
    const u8* L = sel + k * SL_BYTES;
    const u64 base = *reinterpret_cast<const u64*>(L);
    u64 g_last = k * K + (K - 1u);
    if (g_last >= count) g_last = count - 1u;
    const T low = reinterpret_cast<const T*>(L + 8)[(u32)(g_last - k * K)];
    const u64 unit0 = base >> SelFmt<T>::UNIT;
    const u64 last_blk = (u64)nblocks - 1u;
    u64 e0 = ((unit0 + 1u) << BPU) - 1u; if (e0 > last_blk) e0 = last_blk;
    bool ok;
    if (g_last < rcount[e0]) ok = true;                                  // in the base's own unit
    else if (e0 == last_blk) ok = false;                                 // (cannot happen: every one lies in some block)
    else {
        u64 e1 = ((unit0 + 2u) << BPU) - 1u; if (e1 > last_blk) e1 = last_blk;
        ok = g_last < rcount[e1] && low < (T)base;                       // in the next unit, and before the base's offset
    }
    if (!ok) atomicOr(bad, 1u);
}

// one LANE per query, UN queries of a lane in flight: 64 x UN lines per wave and round trip
template <typename T>
__global__ __launch_bounds__(256)
void k_select_sel(const u8* __restrict__ sel, u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    constexpr u32 K = SelFmt<T>::K;
    constexpr u32 UN = 2u;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 qi0 = (u64)blockIdx.x * blockDim.x + threadIdx.x; qi0 < nq; qi0 += UN * stride) {
        u64 base[UN]; T low[UN]; bool live[UN], ok[UN];
#pragma unroll
        for (u32 u = 0; u < UN; ++u) {
            const u64 qq = qi0 + u * stride;
            live[u] = qq < nq;
            const u64 r = live[u] ? __builtin_nontemporal_load(&q[qq]) : 0ull;
            ok[u] = live[u] && r != 0ull && r <= total;
            const u64 g = ok[u] ? r - 1u : 0ull;
            const u64 k = g / K;
            const u32 i = (u32)(g - k * K);
            const u8* L = sel + k * SL_BYTES;
            base[u] = *reinterpret_cast<const u64*>(L);
            low[u] = reinterpret_cast<const T*>(L + 8)[i];
        }
#pragma unroll
        for (u32 u = 0; u < UN; ++u) {
            const u64 qq = qi0 + u * stride;
            if (live[u]) {
                pos[qq] = ok[u] ? base[u] + (T)(low[u] - (T)base[u]) : 0ull;
                found[qq] = ok[u] ? 1 : 0;
            }
        }
    }
}
