// bmx_device.h -- wave64 device primitives for 64 Kbit bit-blocks and GAP blocks (gfx950).
//
// Register image of a bit-block ("Blk"): one wavefront holds one whole block,
// lane l keeps r[i] = the 16 bytes at byte offset i*1024 + l*16 (i = 0..7).
// Every HBM access of a block is therefore 8 fully coalesced 1 KiB wave loads
// (global_load_dwordx4 with a wave-uniform SGPR base).  32-bit word w of the
// block (src/bmconst.h:55: 2048 words) lives at (i, l, j) with w = i*256 + l*4 + j.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// (len << 1 | start bit) of a GAP block, carried in bits 48..60 of its descriptor
#define GMETA(e) ((u32)((e) >> 48) & 0x1FFFu)
#define GMETA_NONE 0xFFFFFFFFu
#define BMX_DESC_KIND(d) ((u32)((d) & 3ull))
#define BMX_DESC_PTR(d)  ((d) & ~3ull)

// plain clang vector (not HIP's uint4 class) so it can live behind an
// address_space(1) pointer: loads become global_load_dwordx4 with an SGPR base
// instead of flat_load (flat also ties up lgkmcnt and blocks load/ALU overlap).
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gcptr4;
typedef __attribute__((address_space(1))) u32x4* gptr4;
typedef const __attribute__((address_space(1))) u16* gcptr16;

// ROWS register rows of a block (ROWS KiB); a whole block is Part<8>
template <int ROWS> struct Part { u32x4 r[ROWS]; };
typedef Part<8> Blk;

__device__ __forceinline__ gcptr4 as_gc4(u64 addr) { return (gcptr4)(uintptr_t)addr; }
__device__ __forceinline__ gcptr4 as_gc4(const void* p) { return (gcptr4)(uintptr_t)p; }
__device__ __forceinline__ gptr4 as_g4(void* p) { return (gptr4)(uintptr_t)p; }
// (pipeline rows keep a GAP block's GMETA in bits 48..60 next to its pointer -- k_pipe_sort -- so the address is masked here)
__device__ __forceinline__ gcptr16 as_gc16(u64 addr) { return (gcptr16)(uintptr_t)(addr & 0x0000FFFFFFFFFFFFull); }

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ u64 uniform64(u64 v)
{
    u32 lo = __builtin_amdgcn_readfirstlane((u32)v);
    u32 hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 uniform32(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ u32 wave_sum(u32 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum across the 64 lanes: six v_add with a DPP operand (row_shr 1/2/4/8 inside the
// rows of 16 lanes, then row_bcast:15 / row_bcast:31 to carry the row totals), no LDS crossbar traffic.
// update_dpp(old = 0, ...) yields 0 in lanes without a source, i.e. the identity of the addition.
__device__ __forceinline__ u32 wave_scan_incl(u32 v, u32 /*lane*/)
{
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

template <int ROWS, bool NT>
__device__ __forceinline__ void part_load(Part<ROWS>& b, gcptr4 p, u32 lane)
{
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        if constexpr (NT) b.r[i] = __builtin_nontemporal_load(p + i * 64 + lane);
        else b.r[i] = p[i * 64 + lane];
    }
}
template <int ROWS>
__device__ __forceinline__ bool part_is_zero(const Part<ROWS>& b)
{
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) v |= b.r[i].x | b.r[i].y | b.r[i].z | b.r[i].w;
    return __ballot(v != 0u) == 0ull;
}
template <int ROWS>
__device__ __forceinline__ bool part_is_ones(const Part<ROWS>& b)
{
    u32 v = ~0u;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) v &= b.r[i].x & b.r[i].y & b.r[i].z & b.r[i].w;
    return __ballot(v != ~0u) == 0ull;
}

__device__ __forceinline__ void blk_load(Blk& b, gcptr4 p, u32 lane)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) b.r[i] = p[i * 64 + lane];
}
__device__ __forceinline__ void blk_store(const Blk& b, gptr4 p, u32 lane)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i * 64 + lane] = b.r[i];
}
__device__ __forceinline__ void blk_fill(Blk& b, u32 v)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) b.r[i] = (u32x4)(v);
}
__device__ __forceinline__ void blk_and(Blk& a, const Blk& b)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) a.r[i] &= b.r[i];
}
__device__ __forceinline__ void blk_andn(Blk& a, const Blk& b)   // a &= ~b
{
#pragma unroll
    for (int i = 0; i < 8; ++i) a.r[i] &= ~b.r[i];
}
__device__ __forceinline__ void blk_or(Blk& a, const Blk& b)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) a.r[i] |= b.r[i];
}
__device__ __forceinline__ void blk_xor(Blk& a, const Blk& b)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) a.r[i] ^= b.r[i];
}
__device__ __forceinline__ void blk_op(int op, Blk& a, const Blk& b)
{
    switch (op) { case 0: blk_and(a, b); break; case 1: blk_or(a, b); break;
                  case 2: blk_xor(a, b); break; default: blk_andn(a, b); break; }
}
// lane-local OR of all words (non-zero iff this lane holds any set bit)
__device__ __forceinline__ u32 blk_lane_or(const Blk& b)
{
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v |= b.r[i].x | b.r[i].y | b.r[i].z | b.r[i].w;
    return v;
}
__device__ __forceinline__ u32 blk_lane_and(const Blk& b)
{
    u32 v = ~0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) v &= b.r[i].x & b.r[i].y & b.r[i].z & b.r[i].w;
    return v;
}
// wave-uniform tests (bit_is_all_zero src/bmfunc.h:1669, is_bits_one :6838)
__device__ __forceinline__ bool blk_is_zero(const Blk& b) { return __ballot(blk_lane_or(b) != 0u) == 0ull; }
__device__ __forceinline__ bool blk_is_ones(const Blk& b) { return __ballot(blk_lane_and(b) != ~0u) == 0ull; }

// lane-local popcount (bit_block_count src/bmfunc.h:5808); caller wave_sum()s it
__device__ __forceinline__ u32 blk_lane_popcount(const Blk& b)
{
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += __popcll(((u64)b.r[i].y << 32) | b.r[i].x);
        c += __popcll(((u64)b.r[i].w << 32) | b.r[i].z);
    }
    return c;
}

// ---------------------------------------------------------------------------
// Transition masks: bit k of t(word) is set iff bit k differs from its
// predecessor in linear bit order (bit_block_calc_change src/bmfunc.h:6040).
// Bit 0 of word 0 has no predecessor.  Returns the lane-local transition count
// and leaves the masks in t.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 blk_transitions(const Blk& b, Blk& t, u32 lane)
{
    u32 cnt = 0;
    u32 prev_row_last = 0;          // MSB of word (i-1, 63, 3), wave-uniform
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 w0 = b.r[i].x, w1 = b.r[i].y, w2 = b.r[i].z, w3 = b.r[i].w;
        u32 up = __shfl_up(w3, 1, 64) >> 31;                  // MSB of lane-1's last word
        u32 p0 = lane ? up : (i ? prev_row_last : (w0 & 1u)); // block start: no transition
        t.r[i].x = w0 ^ ((w0 << 1) | p0);
        t.r[i].y = w1 ^ ((w1 << 1) | (w0 >> 31));
        t.r[i].z = w2 ^ ((w2 << 1) | (w1 >> 31));
        t.r[i].w = w3 ^ ((w3 << 1) | (w2 >> 31));
        cnt += __popc(t.r[i].x) + __popc(t.r[i].y) + __popc(t.r[i].z) + __popc(t.r[i].w);
        prev_row_last = __shfl(w3, 63, 64) >> 31;
    }
    return cnt;
}

// ---------------------------------------------------------------------------
// GAP decode (normative rule: SURVEY.md Appendix B; reference
// gap_convert_to_bitset src/bmfunc.h:5232 / gap_add_to_bitset :4796).
// Data-parallel form: scatter a toggle bit at e[k]+1 for k = 1..len-1 (plus
// position 0 when the block starts with a 1-run) into a per-wave 8 KiB LDS
// bitmap, then take the inclusive prefix-XOR over the 65,536 positions:
// in-word by shift/xor doubling, across words by ballot + mbcnt parity.
// lds: 2048 u32 private to this wave.  All 64 lanes must call.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 prefix_xor32(u32 x)
{
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
    return x;
}

// meta = GMETA of the block's descriptor when the caller has it: the run ends are then requested at once,
// without a round trip for the header word.
__device__ __forceinline__ void gap_decode_xor(gcptr16 g, u32* lds, Blk& out, u32 lane, u32 meta = GMETA_NONE)
{
    u32 len, sbit;
    if (meta != GMETA_NONE) { len = meta >> 1; sbit = meta & 1u; }
    else { u32 hdr = g[0]; len = hdr >> 3; sbit = hdr & 1u; }
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 64 + lane] = (u32x4)(0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (sbit && lane == 0) atomicXor(&lds[0], 1u);
    // run ends are fetched 8 wave-loads at a time (independent, issued together) before the LDS atomics
    // that consume them: one memory round trip per 512 runs instead of one per 64
    for (u32 kb = 1; kb < len; kb += 512u) {
        u32 e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { u32 k = kb + (u32)j * 64u + lane; e[j] = k < len ? (u32)g[k] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (e[j] != 0xFFFFFFFFu) { u32 p = e[j] + 1u; atomicXor(&lds[p >> 5], 1u << (p & 31u)); }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    u32 carry = 0;                      // value of the bit preceding the current row (wave-uniform)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32x4 t = l4[i * 64 + lane];
        u32 x0 = prefix_xor32(t.x), x1 = prefix_xor32(t.y), x2 = prefix_xor32(t.z), x3 = prefix_xor32(t.w);
        // MSB of the in-word prefix = parity of the word's toggles
        u64 b0 = __ballot((int)(x0 >> 31)), b1 = __ballot((int)(x1 >> 31));
        u64 b2 = __ballot((int)(x2 >> 31)), b3 = __ballot((int)(x3 >> 31));
        // toggles in lower lanes of this row (all four words of each lower lane)
        u32 below = __builtin_amdgcn_mbcnt_hi((u32)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)b0, 0));
        below = __builtin_amdgcn_mbcnt_hi((u32)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)b1, below));
        below = __builtin_amdgcn_mbcnt_hi((u32)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((u32)b2, below));
        below = __builtin_amdgcn_mbcnt_hi((u32)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((u32)b3, below));
        u32 c0 = (carry ^ below) & 1u;
        u32 c1 = c0 ^ (x0 >> 31), c2 = c1 ^ (x1 >> 31), c3 = c2 ^ (x2 >> 31);
        out.r[i].x = x0 ^ (0u - c0);
        out.r[i].y = x1 ^ (0u - c1);
        out.r[i].z = x2 ^ (0u - c2);
        out.r[i].w = x3 ^ (0u - c3);
        carry ^= (u32)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3)) & 1u;
    }
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Applying a GAP operand to an accumulator block held in LDS (2048 u32 per wave):
// the run-parallel twin of gap_add_to_bitset / gap_and_to_bitset / gap_sub_to_bitset
// (src/bmfunc.h:4796,4847,4669 with or_bit_block / sub_bit_block :4520,4568).
//   GAP_OR : set the 1-runs        GAP_AND : clear the 0-runs      GAP_SUB : clear the 1-runs
// First / last word of a run go through ds atomics (several lanes may touch one
// word), whole words in between are plain stores (all writers store the same value).
// Cost follows the run list (2 B per run end read, ~2 LDS ops per run) instead of the
// fixed ~600 instructions of gap_decode -- what makes thousands of sparse GAP
// operands per column (BASELINE configs[4]) affordable.
// ---------------------------------------------------------------------------
enum { GAP_OR = 0, GAP_AND = 1, GAP_SUB = 2 };

template <int MODE>
__device__ __forceinline__ void lds_apply_edge(u32* lds, u32 w, u32 mask)
{
    if constexpr (MODE == GAP_OR) atomicOr(&lds[w], mask); else atomicAnd(&lds[w], ~mask);
}

// one run [s, e] (inclusive bit positions); words strictly inside are returned for the caller to fill
template <int MODE>
__device__ __forceinline__ void lds_apply_run_edges(u32* lds, u32 s, u32 e)
{
    u32 wl = s >> 5, wr = e >> 5;
    u32 ml = ~0u << (s & 31u), mr = ~0u >> (31u - (e & 31u));
    if (wl == wr) lds_apply_edge<MODE>(lds, wl, ml & mr);
    else { lds_apply_edge<MODE>(lds, wl, ml); lds_apply_edge<MODE>(lds, wr, mr); }
}

// wave-parallel over the runs of ONE operand: lane j takes every 64th run of the wanted polarity
template <int MODE>
__device__ __forceinline__ void gap_apply_lds_wave(gcptr16 g, u32* lds, u32 lane, u32 meta = GMETA_NONE)
{
    const u32 fill = (MODE == GAP_OR) ? ~0u : 0u;
    u32 len, sbit;
    if (meta != GMETA_NONE) { len = meta >> 1; sbit = meta & 1u; }
    else { u32 hdr = g[0]; len = hdr >> 3; sbit = hdr & 1u; }
    const u32 want = (MODE == GAP_AND) ? 0u : 1u;
    u32 k0 = (sbit == want) ? 1u : 2u;                 // first run (1-based) with the wanted value
    // 4 steps of 64 runs per batch: the 8 run-end loads of a batch are issued together, then applied
    for (u32 kb = k0; kb <= len; kb += 512u) {
        u32 ee[4], ss[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                      // unconditional reads (index clamped), selected afterwards:
            u32 k = kb + (u32)j * 128u + 2u * lane;        // a predicated read whose value is used inside the predicate
            u32 kk = k <= len ? k : len;                   // makes hipcc wait for every single one
            ee[j] = (u32)g[kk]; ss[j] = (u32)g[kk - 1u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 k = kb + (u32)j * 128u + 2u * lane;
            bool act = k <= len;
            ee[j] = act ? ee[j] : 0xFFFFFFFFu;
            ss[j] = act ? ((k == 1u) ? 0u : ss[j] + 1u) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kb + (u32)j * 128u > len) break;           // wave-uniform
            bool act = ee[j] != 0xFFFFFFFFu;
            u32 s = ss[j], e = act ? ee[j] : 0u;
            if (act) lds_apply_run_edges<MODE>(lds, s, e);
            u32 wl = s >> 5, wr = e >> 5;
            u32 inner = (act && wr > wl + 1u) ? wr - wl - 1u : 0u;
            // short interiors: each lane fills its own; long ones (> 16 words): the whole wave helps
            if (inner && inner <= 16u) for (u32 w = wl + 1u; w < wr; ++w) lds[w] = fill;
            u64 longm = __ballot(inner > 16u);
            while (longm) {
                u32 l = (u32)__builtin_ctzll(longm); longm &= longm - 1ull;
                u32 a = __builtin_amdgcn_readlane(wl, l) + 1u, b = __builtin_amdgcn_readlane(wr, l);
                for (u32 w = a + lane; w < b; w += 64u) lds[w] = fill;
            }
        }
    }
}

// GAP block -> register image (gap_convert_to_bitset src/bmfunc.h:5232): zero the wave's LDS block, set the 1-runs
// run-parallel (gap_apply_lds_wave<GAP_OR>: cost follows the run count -- ~60 instructions for a sparse block, ~300 for
// a 1,200-run one), read it back.  The fixed-cost twin gap_decode_xor (toggle scatter + prefix-XOR, ~600 instructions
// whatever the block holds) is what round 1 used everywhere; it stays as the cross-check of the parity tests.
__device__ __forceinline__ void gap_decode(gcptr16 g, u32* lds, Blk& out, u32 lane, u32 meta = GMETA_NONE)
{
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 64 + lane] = (u32x4)(0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    gap_apply_lds_wave<GAP_OR>(g, lds, lane, meta);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) out.r[i] = l4[i * 64 + lane];
    __builtin_amdgcn_wave_barrier();
}

// one lane walks the wanted runs of its own operand (64 operands per wave step): for many short operands.
// The block is fetched 16 B (8 run ends) at a time -- GAP blocks are 16-B aligned in the slab -- with the
// next chunk requested before the current one is processed: one memory round trip per 8 run ends.
// Only runs of the wanted polarity are visited: with x_i the i-th dword of the block (lo16 = word 2i,
// hi16 = word 2i+1), run 2i+1 is [lo(x_i)+1, hi(x_i)] (run 1 starts at 0) and run 2i+2 is
// [hi(x_i)+1, lo(x_i+1)].
template <int MODE>
__device__ __forceinline__ void gap_apply_chunk(u32* lds, const u32 x[5], u32 c, u32 len, bool odd_runs)
{
    const u32 fill = (MODE == GAP_OR) ? ~0u : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 k, s, e;
        if (odd_runs) { k = 8u * c + 2u * i + 1u; s = (k == 1u) ? 0u : (x[i] & 0xFFFFu) + 1u; e = x[i] >> 16; }
        else          { k = 8u * c + 2u * i + 2u; s = (x[i] >> 16) + 1u; e = x[i + 1] & 0xFFFFu; }
        if (k <= len) {
            lds_apply_run_edges<MODE>(lds, s, e);
            for (u32 w = (s >> 5) + 1u; w < (e >> 5); ++w) lds[w] = fill;
        }
    }
}

// The first 64 bytes (31 run ends: every block of a sparse vector) are requested in ONE go, before the
// header is known -- GAP slabs carry a 64-byte guard at their end for this -- so a short block costs a
// single memory round trip; longer blocks continue 16 B at a time with one chunk of run-ahead.
template <int MODE>
__device__ __forceinline__ void gap_apply_lds_lane(gcptr16 g, u32* lds)
{
    const u32 want = (MODE == GAP_AND) ? 0u : 1u;
    gcptr4 g4 = (gcptr4)(uintptr_t)g;
    u32x4 c0 = g4[0], c1 = g4[1], c2 = g4[2], c3 = g4[3];
    u32 hdr = c0.x & 0xFFFFu;
    u32 len = hdr >> 3;
    bool odd_runs = (hdr & 1u) == want;     // wanted runs are 1,3,5,.. (else 2,4,6,..)
    u32 nchunks = (len + 8u) >> 3;          // ceil((len + 1) / 8)
    { u32 x[5] = {c0.x, c0.y, c0.z, c0.w, c1.x}; gap_apply_chunk<MODE>(lds, x, 0, len, odd_runs); }
    if (nchunks > 1u) { u32 x[5] = {c1.x, c1.y, c1.z, c1.w, c2.x}; gap_apply_chunk<MODE>(lds, x, 1, len, odd_runs); }
    if (nchunks > 2u) { u32 x[5] = {c2.x, c2.y, c2.z, c2.w, c3.x}; gap_apply_chunk<MODE>(lds, x, 2, len, odd_runs); }
    if (nchunks > 3u) {
        u32x4 cur = c3;
        for (u32 c = 3; c < nchunks; ++c) {
            u32x4 nxt = (c + 1u < nchunks) ? g4[c + 1u] : cur;
            u32 x[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
            gap_apply_chunk<MODE>(lds, x, c, len, odd_runs);
            cur = nxt;
        }
    }
}

// ---- OR of MANY sparse GAP operands (column-tile kernel): branch-light lane mode -----------------
// The 64-byte head of the next operand's block is fetched while the current one is applied (GapHead),
// and a run is applied without control flow in the common case: runs of a sparse vector sit inside one
// word, so the first-word mask is OR-ed unconditionally (an inactive slot ORs 0) and only a run that
// crosses words takes the branch.  ~20 VALU per run instead of ~45 (the kernel is VALU-bound: 16 waves
// per CU each issuing ~16 runs per operand).
struct GapHead { u32x4 c[4]; };

__device__ __forceinline__ void gap_head_fetch(GapHead& h, u64 gaddr, bool is_gap)
{
    gcptr4 g4 = (gcptr4)(uintptr_t)gaddr;
    if (is_gap) {
#pragma unroll
        for (int j = 0; j < 4; ++j) h.c[j] = g4[j];
    } else h.c[0].x = 0u;                                              // header 0 = length 0: nothing is applied (the rest is don't-care)
}

// x[0..4] = dwords 4c..4c+4 of the block; lim = len - (odd_runs ? 1 : 2): wanted run i of the chunk is
// active iff 8c + 2i <= lim (signed: lim is -1 / -2 for an empty header).  Even start: the u16 stream is
// re-paired with one v_alignbit per run so both polarities share the code below.
__device__ __forceinline__ void gap_or_chunk_fast(u32* lds, const u32 x[5], u32 c, int lim, bool odd_runs)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 y = odd_runs ? x[i] : __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);   // lo16 = end of the 0-run, hi16 = end of the 1-run
        u32 s = (y & 0xFFFFu) + 1u;
        if (c == 0u && i == 0) s = odd_runs ? 0u : s;                  // run 1 starts at bit 0 (lo16 is the header)
        u32 e = y >> 16;
        bool act = (int)(8u * c + 2u * (u32)i) <= lim;
        u32 wl = s >> 5, wr = e >> 5;
        u32 lo = 1u << (s & 31u), hi2 = 2u << (e & 31u);
        bool same = wl == wr;
        u32 m = (same ? hi2 : 0u) - lo;                                // bits s..e of the word, or s..31
        atomicOr(&lds[wl], act ? m : 0u);                              // inactive slot: ORs 0 (wl <= 2048: inside the tile)
        if (act && !same) {                                            // rare for sparse operands
            atomicOr(&lds[wr], hi2 - 1u);
            for (u32 w = wl + 1u; w < wr; ++w) lds[w] = ~0u;
        }
    }
}

__device__ __forceinline__ void gap_or_lane_fast(const GapHead& h, u64 gaddr, u32* lds)
{
    u32 hdr = h.c[0].x & 0xFFFFu;
    u32 len = hdr >> 3;
    bool odd_runs = (hdr & 1u) != 0u;                                  // 1-runs are runs 1,3,5,.. (else 2,4,..)
    int lim = (int)len - (odd_runs ? 1 : 2);
    u32 nchunks = (len + 8u) >> 3;
    { u32 x[5] = {h.c[0].x, h.c[0].y, h.c[0].z, h.c[0].w, h.c[1].x}; gap_or_chunk_fast(lds, x, 0, lim, odd_runs); }
    if (nchunks > 1u) { u32 x[5] = {h.c[1].x, h.c[1].y, h.c[1].z, h.c[1].w, h.c[2].x}; gap_or_chunk_fast(lds, x, 1, lim, odd_runs); }
    if (nchunks > 2u) { u32 x[5] = {h.c[2].x, h.c[2].y, h.c[2].z, h.c[2].w, h.c[3].x}; gap_or_chunk_fast(lds, x, 2, lim, odd_runs); }
    if (nchunks > 3u) {
        gcptr4 g4 = (gcptr4)(uintptr_t)gaddr;
        u32x4 cur = h.c[3];
        for (u32 c = 3; c < nchunks; ++c) {
            u32x4 nxt = (c + 1u < nchunks) ? g4[c + 1u] : cur;
            u32 x[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
            gap_or_chunk_fast(lds, x, c, lim, odd_runs);
            cur = nxt;
        }
    }
}

// Variant 2 of the run application (VALU-bound kernel: fewer instructions per set bit).  In a sparse vector nearly
// every block starts with a 0-run and nearly every 1-run is ONE bit: then the u16 stream is a list of bit positions
// (every second word), and a run costs alignbit + sub + shift + shift + address + select + ds_or.  Runs that span more
// than one bit (or blocks that start with a 1-run) take the general path of gap_or_chunk_fast.
__device__ __forceinline__ void gap_or_chunk_v2(u32* lds, const u32 x[5], u32 c, int lim)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 y = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);       // lo16 = end of the 0-run, hi16 = end of the 1-run
        u32 e = y >> 16;
        u32 d = e - (y & 0xFFFFu);                                   // length of the 1-run
        bool act = (int)(8u * c + 2u * (u32)i) <= lim;
        bool single = d == 1u;
        atomicOr(&lds[e >> 5], (act && single) ? (1u << (e & 31u)) : 0u);
        if (act && !single) {                                        // rare for sparse operands
            u32 s = e - d + 1u;
            u32 wl = s >> 5, wr = e >> 5;
            u32 lo = 1u << (s & 31u), hi2 = 2u << (e & 31u);
            if (wl == wr) atomicOr(&lds[wl], hi2 - lo);
            else {
                atomicOr(&lds[wl], 0u - lo);
                atomicOr(&lds[wr], hi2 - 1u);
                for (u32 w = wl + 1u; w < wr; ++w) lds[w] = ~0u;
            }
        }
    }
}

__device__ __forceinline__ void gap_or_lane_v2(const GapHead& h, u64 gaddr, u32* lds)
{
    u32 hdr = h.c[0].x & 0xFFFFu;
    u32 len = hdr >> 3;
    bool odd_runs = (hdr & 1u) != 0u;
    if (__ballot(odd_runs) != 0ull) { gap_or_lane_fast(h, gaddr, lds); return; }     // some block starts with a 1-run: general code for the wave
    int lim = (int)len - 2;
    u32 nchunks = (len + 8u) >> 3;
    { u32 x[5] = {h.c[0].x, h.c[0].y, h.c[0].z, h.c[0].w, h.c[1].x}; gap_or_chunk_v2(lds, x, 0, lim); }
    if (nchunks > 1u) { u32 x[5] = {h.c[1].x, h.c[1].y, h.c[1].z, h.c[1].w, h.c[2].x}; gap_or_chunk_v2(lds, x, 1, lim); }
    if (nchunks > 2u) { u32 x[5] = {h.c[2].x, h.c[2].y, h.c[2].z, h.c[2].w, h.c[3].x}; gap_or_chunk_v2(lds, x, 2, lim); }
    if (nchunks > 3u) {
        gcptr4 g4 = (gcptr4)(uintptr_t)gaddr;
        u32x4 cur = h.c[3];
        for (u32 c = 3; c < nchunks; ++c) {
            u32x4 nxt = (c + 1u < nchunks) ? g4[c + 1u] : cur;
            u32 x[5] = {cur.x, cur.y, cur.z, cur.w, nxt.x};
            gap_or_chunk_v2(lds, x, c, lim);
            cur = nxt;
        }
    }
}

__device__ __forceinline__ bool lds_blk_is_zero(const u32* lds, u32 lane)
{
    const u32x4* l4 = reinterpret_cast<const u32x4*>(lds);
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { u32x4 t = l4[i * 64 + lane]; v |= t.x | t.y | t.z | t.w; }
    return __ballot(v != 0u) == 0ull;
}

__device__ __forceinline__ void blk_to_lds(const Blk& b, u32* lds, u32 lane)
{
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 64 + lane] = b.r[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void blk_from_lds(Blk& b, const u32* lds, u32 lane)
{
    const u32x4* l4 = reinterpret_cast<const u32x4*>(lds);
#pragma unroll
    for (int i = 0; i < 8; ++i) b.r[i] = l4[i * 64 + lane];
}

__device__ __forceinline__ u32 lds_blk_popcount(const u32* lds, u32 lane)
{
    const u32x4* l4 = reinterpret_cast<const u32x4*>(lds);
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32x4 t = l4[i * 64 + lane];
        c += (u32)__popc(t.x) + (u32)__popc(t.y) + (u32)__popc(t.z) + (u32)__popc(t.w);
    }
    return uniform32(wave_sum(c));
}

// ---------------------------------------------------------------------------
// Sparse state of an AND / SUB accumulator.  Once the running result holds at most
// SPARSE_CAP set bits it is kept as a list of bit positions (<= 16 per lane, in
// registers) and every further GAP operand is a membership test of those positions:
// the operand's run ends are staged in LDS with coalesced 16-byte reads (one request
// per 512 runs, requested one operand ahead, header two ahead) and each candidate is
// located by a branch-free binary search (gap_bfind / gap_test, src/bmfunc.h:1721,1803).
// This is the wave64 counterpart of the reference's digest narrowing
// (gap_and_to_bitset(dest, gap, digest) src/bmfunc.h:4893): work follows the number
// of surviving bits, not the number of runs of the operand.  Results are identical:
// acc & gap (or acc & ~gap) restricted to the set bits of acc.
// LDS use inside the wave's 8 KiB block: run ends in words [0, 640), candidate list in
// [640, 1152); the block itself is rebuilt from the survivors at the end.
// Measured alternative (round 2, dropped): RUN-side filtering -- accumulator kept as an LDS bitmap A, per operand
// B |= A & run over the kept runs (or A &= ~run over the dropped ones, whichever covers fewer words), swap, zero.
// It needs no "few survivors" precondition, but decoding every run slot (~35 instructions per run: 390 runs per operand
// at 0.3 %) costs more than 196 candidates x 10 probes: 4.45 ms against 2.75 ms on the all-GAP 256-way case, 3.25 against
// 1.57 ms at 0.1 %.  Either way a (column, operand) pair costs ~300 wave instructions; the kernel is VALU-bound there.
// ---------------------------------------------------------------------------
#define SPARSE_CAP 1024u
#define SPARSE_NONE 0xFFFFFFFFu

struct GapStage { u32x4 d[3]; };

__device__ __forceinline__ void gap_stage_load(GapStage& s, gcptr16 g, u32 hdr, u32 lane)
{
    gcptr4 g4 = (gcptr4)(uintptr_t)g;
    u32 nch = ((hdr >> 3) + 8u) >> 3;                  // 16-byte chunks holding words 0..len
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 3; ++j) { u32 c = lane + 64u * (u32)j; s.d[j] = c < nch ? g4[c] : z; }
}
__device__ __forceinline__ void gap_stage_store(const GapStage& s, u32* lds, u32 hdr, u32 lane)
{
    u32x4* l4 = reinterpret_cast<u32x4*>(lds);
    u32 nch = ((hdr >> 3) + 8u) >> 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) { u32 c = lane + 64u * (u32)j; if (c < nch) l4[c] = s.d[j]; }
}

template <int MODE, int NJ>
__device__ __forceinline__ u32 sparse_test(u32 (&cand)[16], const u16* G, u32 hdr)
{
    u32 len = hdr >> 3, sbit = hdr & 1u;
    u32 k[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) k[j] = 0u;
    // k = number of run ends below the position  =>  the position lies in run k+1, value sbit ^ (k & 1)
    for (u32 step = 1u << (31 - __builtin_clz(len)); step; step >>= 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32 t = k[j] + step; t = t < len ? t : len;       // G[len] = 65535 never compares below
            u32 v = G[t];
            k[j] = v < cand[j] ? t : k[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        u32 bitv = (sbit ^ k[j]) & 1u;
        bool keep = (MODE == GAP_AND) ? (bitv != 0u) : (bitv == 0u);
        if (!keep) cand[j] = SPARSE_NONE;
    }
    u32 alive = 0;                                                    // scalar: ballots, no cross-lane traffic
#pragma unroll
    for (int j = 0; j < NJ; ++j) alive += (u32)__popcll(__ballot(cand[j] != SPARSE_NONE));
    return alive;
}

// acc (in LDS, <= SPARSE_CAP bits) op= operands i0..n-1 of the list; true when nothing survives,
// else the block is rebuilt in LDS.
template <int MODE>
__device__ __forceinline__ bool gap_apply_sparse(const u64* __restrict__ plist_back, u32 i0, u32 n, u32* lds, u32 lane)
{
    u16* list = reinterpret_cast<u16*>(lds + 640);
    u32 cand[16];
    u32 total, nj;
    {   // set bits -> position list (lane-contiguous via a wave scan of the lane popcounts)
        Blk b; blk_from_lds(b, lds, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 cnt = blk_lane_popcount(b);
        u32 incl = wave_scan_incl(cnt, lane);
        total = uniform32(__shfl(incl, 63, 64));
        u32 off = incl - cnt;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32 ws[4] = {b.r[i].x, b.r[i].y, b.r[i].z, b.r[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                u32 w = ws[c];
                u32 base = ((u32)i * 256u + lane * 4u + (u32)c) << 5;
                while (w) { u32 bit = (u32)__builtin_ctz(w); w &= w - 1u; list[off++] = (u16)(base + bit); }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nj = (total + 63u) >> 6;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            u32 idx = (u32)j * 64u + lane;
            cand[j] = idx < total ? (u32)list[idx] : SPARSE_NONE;
        }
    }
    const u16* G = reinterpret_cast<const u16*>(lds);
    u32 i = i0;
    gcptr16 g0 = as_gc16(uniform64(*(plist_back - i)));
    u32 h0 = uniform32((u32)g0[0]);
    GapStage cur; gap_stage_load(cur, g0, h0, lane);
    gcptr16 g1 = as_gc16(uniform64(*(plist_back - (i + 1u < n ? i + 1u : n - 1u))));
    u32 h1v = (u32)g1[0];
    for (; i < n; ++i) {
        u32 h1 = uniform32(h1v);
        GapStage nxt; gap_stage_load(nxt, g1, h1, lane);             // operand i+1 (re-read of the last at the end)
        u32 i2 = i + 2u < n ? i + 2u : n - 1u;
        gcptr16 g2 = as_gc16(uniform64(*(plist_back - i2)));
        u32 h2v = (u32)g2[0];
        gap_stage_store(cur, lds, h0, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 alive;
        if (nj <= 1u) alive = sparse_test<MODE, 1>(cand, G, h0);
        else if (nj <= 2u) alive = sparse_test<MODE, 2>(cand, G, h0);
        else if (nj <= 3u) alive = sparse_test<MODE, 3>(cand, G, h0);
        else if (nj <= 4u) alive = sparse_test<MODE, 4>(cand, G, h0);
        else if (nj <= 6u) alive = sparse_test<MODE, 6>(cand, G, h0);
        else if (nj <= 8u) alive = sparse_test<MODE, 8>(cand, G, h0);
        else if (nj <= 12u) alive = sparse_test<MODE, 12>(cand, G, h0);
        else alive = sparse_test<MODE, 16>(cand, G, h0);
        if (alive == 0u) return true;
        if (((alive + 63u) >> 6) < nj) {                              // fewer slots suffice: compact through the list
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) mine += cand[j] != SPARSE_NONE ? 1u : 0u;
            u32 off = wave_scan_incl(mine, lane) - mine;
#pragma unroll
            for (int j = 0; j < 16; ++j) if (cand[j] != SPARSE_NONE) list[off++] = (u16)cand[j];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            nj = (alive + 63u) >> 6;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                u32 idx = (u32)j * 64u + lane;
                cand[j] = idx < alive ? (u32)list[idx] : SPARSE_NONE;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cur = nxt; g0 = g1; h0 = h1; g1 = g2; h1v = h2v;
    }
    {   // survivors -> block image
        u32x4* l4 = reinterpret_cast<u32x4*>(lds);
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < 8; ++r) l4[r * 64 + lane] = z;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (cand[j] != SPARSE_NONE) atomicOr(&lds[cand[j] >> 5], 1u << (cand[j] & 31u));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return false;
}

// Apply n GAP operands (pointer list walked BACKWARDS from plist_back: row regions pack GAP pointers
// from their end) to the wave's LDS accumulator.  Returns true when an AND / SUB accumulator became
// all-zero (the caller stops: the reference's digest == 0 exit in process_gap_blocks_and/sub,
// src/bmaggregator.h:1820,1854).  Few operands: one at a time, runs spread over the lanes.  Many
// (>= 32): the first 8 that way, the rest one operand per lane, 64 at a time.  AND / SUB test the
// accumulator's population after every step: zero ends the column, <= SPARSE_CAP bits with at least
// two operands left switches to the sparse state above.
template <int MODE>
__device__ __forceinline__ bool gap_apply_list(const u64* __restrict__ plist_back, u32 n, u32* lds, u32 lane)
{
    const bool CHECK = MODE != GAP_OR;
    u32 head = n >= 32u ? 8u : n;
    u32 i = 0;
    if constexpr (CHECK) {
        if (n >= 2u) {
            u32 pop = lds_blk_popcount(lds, lane);
            if (pop == 0u) return true;
            if (pop <= SPARSE_CAP) return gap_apply_sparse<MODE>(plist_back, 0u, n, lds, lane);
        }
    }
    for (; i < head; ++i) {
        gap_apply_lds_wave<MODE>(as_gc16(uniform64(*(plist_back - i))), lds, lane);
        if constexpr (CHECK) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            u32 pop = lds_blk_popcount(lds, lane);
            if (pop == 0u) return true;
            if (pop <= SPARSE_CAP && i + 2u < n) return gap_apply_sparse<MODE>(plist_back, i + 1u, n, lds, lane);
        }
    }
    if (i < n) {                                       // many operands: lane-per-operand, pointers one step ahead
        u64 p = i + lane < n ? *(plist_back - (i + lane)) : 0ull;
        for (; i < n; i += 64u) {
            u64 pn = i + 64u + lane < n ? *(plist_back - (i + 64u + lane)) : 0ull;
            if (i + lane < n) gap_apply_lds_lane<MODE>(as_gc16(p), lds);
            p = pn;
            if constexpr (CHECK) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                u32 pop = lds_blk_popcount(lds, lane);
                if (pop == 0u) return true;
                if (pop <= SPARSE_CAP && i + 64u + 2u <= n)
                    return gap_apply_sparse<MODE>(plist_back, i + 64u, n, lds, lane);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return false;
}

// popcount of a GAP block without expanding it (gap_bit_count_unr src/bmfunc.h:3107).
// Returns the lane-local partial; caller wave_sum()s.
__device__ __forceinline__ u32 gap_lane_popcount(gcptr16 g, u32 lane, u32 meta = GMETA_NONE)
{
    u32 len, s;
    if (meta != GMETA_NONE) { len = meta >> 1; s = meta & 1u; }
    else { u32 hdr = g[0]; len = hdr >> 3; s = hdr & 1u; }
    u32 c = 0;
    // run k (1-based) covers (e[k-1], e[k]] and has value s ^ ((k-1)&1); both ends are read unconditionally
    for (u32 k = 1 + lane; k <= len; k += 64) {
        u32 e = g[k], pe = g[k - 1];                   // k == 1: pe is the header word, unused
        bool one = (s ^ ((k - 1u) & 1u)) != 0u;
        c += one ? ((k == 1u) ? e + 1u : e - pe) : 0u;
    }
    return c;
}

// ---------------------------------------------------------------------------
// synthetic generator: normative arithmetic in oracle/bmx_oracle.c bmo_gen_word64
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ u64 gen_word64(u64 seed, u32 vec_id, u64 w64, u32 d)
{
    if (d >= 65536u) return ~0ull;
    u64 base = seed ^ ((u64)vec_id * 0x9E3779B97F4A7C15ull);
    u64 acc = 0;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) {
        u64 r = mix64(base + (w64 * 16u + k) * 0xD6E8FEB86659FD93ull);
        acc = ((d >> k) & 1u) ? (acc | r) : (acc & r);
    }
    return acc;
}

// ---------------------------------------------------------------------------
// software-pipelined fold of a list of bit-block operands (see k_pipe_counts_bits2)
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 readlane64(u64 v, u32 l)
{
    u32 lo = __builtin_amdgcn_readlane((u32)v, l);
    u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

template <int U, bool NT, int OPK, int ROWS = 8>
__device__ __forceinline__ bool pipe_chain(Part<ROWS>& acc, const u64* __restrict__ plist, u32 n, u32 lane, u32 poff = 0u)
{
    // Folds operands 0 .. n-1 of plist into acc: OPK 0 = AND, 1 = AND-NOT, 2 = OR.  Returns true when
    // acc saturated (all-zero for AND / AND-NOT, all-ones for OR) so the caller can stop early.
    // ROWS < 8: acc is the ROWS KiB slice of the block that starts poff 16-byte units into it (thin
    // shards: more, smaller work items per block column).
    // plist is wave-uniform, so plist[i] is a scalar (SMEM) load: it counts on lgkmcnt, not vmcnt,
    // and is issued one batch ahead -- the vector-memory pipeline never waits for a pointer.
    if (n == 0) return false;
    u64 pn[U];                                               // pointers of the batch to issue next
    auto fetch = [&](u32 k) {
#pragma unroll
        for (int u = 0; u < U; ++u) pn[u] = plist[k + u < n ? k + u : n - 1u];   // tail: repeat the last operand
    };
    auto issue = [&](Part<ROWS>* buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) part_load<ROWS, NT>(buf[u], as_gc4(uniform64(pn[u])) + poff, lane);
    };
    auto consume = [&](Part<ROWS>* buf) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) {
                if constexpr (OPK == 0) acc.r[i] &= buf[u].r[i];
                else if constexpr (OPK == 1) acc.r[i] &= ~buf[u].r[i];
                else acc.r[i] |= buf[u].r[i];
            }
        if constexpr (OPK == 2) return part_is_ones<ROWS>(acc); else return part_is_zero<ROWS>(acc);
    };
    // The loop body is ONE basic block that issues batch j+1 before consuming batch j and batch
    // j+2 before consuming batch j+1; the only branch is the back-edge.  (With an early-exit branch
    // between issue and consume, LLVM sinks the loads below the branch -- or, for a guarded issue,
    // merges the wait counters to vmcnt(0) at the join -- and the two buffers serialise.)
    // Early exit therefore has a granularity of 2U operands.  Past the end the clamped index
    // re-loads the last operand (idempotent; <= 2U L2-resident blocks per column).
    Part<ROWS> A[U], B[U];
    u32 k = U;                      // first operand of the batch to issue next
    fetch(0);
    issue(A);
    fetch(k);
    bool zero;
    do {
        issue(B); fetch(k + U);
        bool z1 = consume(A);
        issue(A); fetch(k + 2 * U);
        bool z2 = consume(B);
        zero = z1 | z2;
        k += 2 * U;
    } while (!zero && k < n + U);   // batch starting at k-U has been issued into A: consume it next time
    return zero;
}
