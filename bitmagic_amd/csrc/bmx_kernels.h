// bmx_kernels.h -- HIP kernels of the bvector/aggregator hot path (gfx950, wave64).
// One wavefront owns one 64 Kbit block (see bmx_device.h).  No MFMA: the work is
// bitwise and HBM-bound; what matters is coalesced 16 B/lane loads, enough
// independent loads in flight, and never reading an operand twice.
#pragma once
#include "bmx_device.h"

// descriptor: device pointer in the low 48 bits, kind in bits 62..63
#define DESC_MAKE(ptr, kind) ((u64)(uintptr_t)(ptr) | ((u64)(kind) << 62))
#define DESC_K(d) ((u32)((d) >> 62))
#define DESC_P(d) ((d) & 0x0000FFFFFFFFFFFFull)
// GAP descriptors also carry (len << 1 | start bit) of the block in bits 48..60 (GMETA, bmx_device.h): the
// run ends of a block can be requested without first waiting for its header word
#define DESC_MAKE_GAP(ptr, len, sbit) (DESC_MAKE(ptr, K_GAP) | ((u64)((((u32)(len)) << 1) | ((u32)(sbit) & 1u)) << 48))
enum { K_NULL = 0, K_FULL = 1, K_BIT = 2, K_GAP = 3 };

// pipeline row flags
#define ROW_EMPTY 1ull
#define ROW_FULL  2ull
#define ROW_ONES  4ull    // accumulator starts as all-ones (no bit-block AND operand)

struct BlockStat { u32 pop; u32 runs; u32 first; u32 kind; };

// ---------------------------------------------------------------------------
// synthetic data: out = nblocks * 1024 64-bit words, bits >= nbits are zero
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_generate(u64 seed, u32 vec_id, int with_common, u32 d, u64 nbits, u64* __restrict__ out, u64 nwords64, u64 word0)
{
    // out[i] = 64-bit word (word0 + i) of the logical vector: word0 != 0 generates a block-range shard
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < nwords64; i += stride) {
        u64 w = word0 + i;
        u64 v = gen_word64(seed, vec_id, w, d);
        if (with_common) v |= gen_word64(seed, 0xFFFFFFFFu, w, d);
        u64 bit0 = w * 64u;
        if (bit0 >= nbits) v = 0;
        else if (nbits - bit0 < 64u) v &= (~0ull) >> (64u - (nbits - bit0));
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------
// import, step 1: per raw block popcount / run count / first bit and the storage
// decision of blocks_manager::optimize_bit_block (src/bmblocks.h:1412-1436):
//   optimize: runs == 1 -> NULL or FULL;  runs < 1276 -> GAP;  else BIT.   no optimize: BIT.
// One wave per block, 4 blocks per workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_block_stats(const uint4* __restrict__ raw, u32 nblocks, int optimize, BlockStat* __restrict__ st)
{
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    Blk b, t;
    blk_load(b, as_gc4(raw + (size_t)nb * 512u), lane);
    u32 pop = wave_sum(blk_lane_popcount(b));
    u32 runs = 1u + wave_sum(blk_transitions(b, t, lane));
    u32 first = __shfl(b.r[0].x, 0, 64) & 1u;
    if (lane == 0) {
        // without optimize every block is stored as a bit-block, an all-zero / all-ones one too (copy_bit_block,
        // src/bmbvimport.h:46 + src/bmblocks.h:1340); optimize_bit_block (:1412) is what turns them into NULL / FULL / GAP
        u32 kind = !optimize ? K_BIT
                 : (runs == 1u) ? (first ? K_FULL : K_NULL)
                 : (runs < 1276u ? K_GAP : K_BIT);
        st[nb] = BlockStat{pop, runs, first, kind};
    }
}

// step 2: exclusive scan of storage sizes (single workgroup; nblocks <= 2^20)
// offs[nb] = bit-block ordinal or GAP u16-word offset (multiple of 8: every GAP block starts on a
// 16-byte boundary so a lane can fetch it with dwordx4 loads); totals[0]=n_bit, [1]=gap_words,
// totals[2..5] = blocks per kind
// 8,192 blocks per pass (SCAN_LAYOUT_PER per thread), 2 barriers per pass.  A 1024-thread workgroup may hold 128 VGPRs per
// thread: the round-1 form (16 items per thread, two DPP scans per step, four per-item arrays) spilled 136 B per thread to
// scratch and took 18.9 us for 15,259 blocks.  Now: kinds counted with ballots, the bit-block ordinal is a masked bit count,
// one DPP scan per step, one offset array -- 12.6 us at 8 items per thread (4: 14.1 us, 16: 14.2 us).
#define SCAN_PER 16u
#define SCAN_LAYOUT_PER 8u
// Wave w of the workgroup owns blocks [base + w*1024, +1024) of a pass and walks them 64 at a time
// (lane = consecutive block: every load / store instruction is one contiguous 1 KiB / 256 B piece),
// carrying its running sums across the 16 steps; wave totals meet in LDS once per pass.
__global__ __launch_bounds__(1024)
void k_scan_layout(const BlockStat* __restrict__ st, u32 nblocks, u32* __restrict__ offs, u64* __restrict__ totals)
{
    __shared__ u32 sm[32];
    __shared__ u32 kcnt[4];
    u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid < 4) kcnt[tid] = 0;
    __syncthreads();
    u32 carry_bit = 0, carry_gap = 0;
    u32 kc[4] = {0, 0, 0, 0};
    const uint4* st4 = reinterpret_cast<const uint4*>(st);
    for (u32 base = 0; base < nblocks; base += 1024u * SCAN_LAYOUT_PER) {
        u32 nb0 = base + w * (64u * SCAN_LAYOUT_PER) + lane;
        u32 kind[SCAN_LAYOUT_PER], e[SCAN_LAYOUT_PER];        // e: offset inside the wave's stretch (bit ordinal or GAP words, by kind)
        uint4 x[SCAN_LAYOUT_PER];
#pragma unroll
        for (u32 i = 0; i < SCAN_LAYOUT_PER; ++i) {
            u32 nb = nb0 + i * 64u;
            x[i] = nb < nblocks ? st4[nb] : make_uint4(0u, 0u, 0u, 0xFFu);      // {pop, runs, first, kind}
        }
        u32 run_b = 0, run_g = 0;                         // wave-local running sums over the steps
#pragma unroll
        for (u32 i = 0; i < SCAN_LAYOUT_PER; ++i) {
            kind[i] = x[i].w;
            u32 vg = kind[i] == K_GAP ? ((x[i].y + 1u + 7u) & ~7u) : 0u;         // GAP blocks start 16-B aligned
            // kinds are counted with ballots (scalar popcounts), the bit-block ordinal is a masked bit count: one DPP scan
            // per step (the GAP words) instead of two
            u64 mb = __ballot(kind[i] == K_BIT), mg = __ballot(kind[i] == K_GAP);
            kc[K_BIT] += (u32)__popcll(mb); kc[K_GAP] += (u32)__popcll(mg);
            kc[K_FULL] += (u32)__popcll(__ballot(kind[i] == K_FULL)); kc[K_NULL] += (u32)__popcll(__ballot(kind[i] == K_NULL));
            u32 ig = wave_scan_incl(vg, lane);
            u32 eb = run_b + __builtin_amdgcn_mbcnt_hi((u32)(mb >> 32), __builtin_amdgcn_mbcnt_lo((u32)mb, 0u));
            e[i] = kind[i] == K_BIT ? eb : run_g + ig - vg;
            run_b += (u32)__popcll(mb); run_g += __shfl(ig, 63, 64);
        }
        if (lane == 0) { sm[w] = run_b; sm[16 + w] = run_g; }
        __syncthreads();
        u32 ob = carry_bit, og = carry_gap, tb = 0, tg = 0;
#pragma unroll
        for (u32 i = 0; i < 16; ++i) { u32 a = sm[i], b = sm[16 + i]; if (i < w) { ob += a; og += b; } tb += a; tg += b; }
        __syncthreads();
#pragma unroll
        for (u32 i = 0; i < SCAN_LAYOUT_PER; ++i) {
            u32 nb = nb0 + i * 64u;
            if (nb < nblocks) offs[nb] = kind[i] == K_BIT ? ob + e[i] : (kind[i] == K_GAP ? og + e[i] : 0u);
        }
        carry_bit += tb; carry_gap += tg;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)                          // (wave-uniform counts) one LDS atomic per wave
        if (lane == 0 && kc[k]) atomicAdd(&kcnt[k], kc[k]);
    __syncthreads();
    if (tid == 0) { totals[0] = carry_bit; totals[1] = carry_gap; }
    if (tid < 4) totals[2 + tid] = kcnt[tid];
}

// step 3: write each block in its final form + the descriptor table.
// GAP conversion = bit_block_to_gap (src/bmfunc.h:5542): run k ends just before
// the k-th transition; index of a transition = 1 + number of earlier transitions.
__global__ __launch_bounds__(256)
void k_emit_blocks(const uint4* __restrict__ raw, u32 nblocks, const BlockStat* __restrict__ st,
                   const u32* __restrict__ offs, uint4* __restrict__ bit_slab, u16* __restrict__ gap_slab,
                   u64* __restrict__ desc)
{
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    u32 kind = uniform32(st[nb].kind);
    if (kind == K_NULL || kind == K_FULL) { if (lane == 0) desc[nb] = DESC_MAKE(0, kind); return; }
    Blk b;
    blk_load(b, as_gc4(raw + (size_t)nb * 512u), lane);
    if (kind == K_BIT) {
        uint4* dst = bit_slab + (size_t)offs[nb] * 512u;
        blk_store(b, as_g4(dst), lane);
        if (lane == 0) desc[nb] = DESC_MAKE(dst, K_BIT);
        return;
    }
    // GAP
    u16* g = gap_slab + offs[nb];
    Blk t;
    (void)blk_transitions(b, t, lane);
    u32 len = uniform32(st[nb].runs);
    u32 idx_base = 1u;                  // first run-end slot
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 c = __popc(t.r[i].x) + __popc(t.r[i].y) + __popc(t.r[i].z) + __popc(t.r[i].w);
        u32 incl = wave_scan_incl(c, lane);
        u32 idx = idx_base + incl - c;
        u32 wbase = (u32)i * 256u + lane * 4u;
        u32 tw[4] = {t.r[i].x, t.r[i].y, t.r[i].z, t.r[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 m = tw[j];
            while (m) {
                u32 k = __builtin_ctz(m); m &= m - 1u;
                g[idx++] = (u16)((wbase + j) * 32u + k - 1u);
            }
        }
        idx_base += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        u32 level = len <= 124u ? 0u : len <= 252u ? 1u : len <= 508u ? 2u : 3u;   // gap_calc_level src/bmfunc.h:5418
        g[0] = (u16)((len << 3) | (level << 1) | st[nb].first);
        g[len] = 65535u;
        desc[nb] = DESC_MAKE_GAP(g, len, st[nb].first);
    }
    // padding words up to the next 16-byte boundary read 0xFFFF: no run end but a block's last has that value, which is how
    // k_agg_or_rows (bmx_kernels7.h) tells a run from padding without the block's length
    if (lane >= 1u && lane <= 7u && len + lane < ((len + 1u + 7u) & ~7u)) g[len + lane] = 0xFFFFu;
}

// ---------------------------------------------------------------------------
// upload: GAP blocks of the host slab (raw, back to back as in a freeze()d arena,
// src/bmblocks.h:2614-2655) -> 16-byte aligned blocks of the device slab, validated
// on the way (the kernels index LDS with the run ends, so a malformed block must never
// get in): word[len] == 65535 (gap_max_bits - 1) and run ends strictly ascending
// (gap block invariant, src/bmfunc.h:1844-1896).  One wave per block; *err |= 1 on a violation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_gap_repack(const u16* __restrict__ raw, const u32* __restrict__ src_off, const u64* __restrict__ desc,
                  u32 nblocks, u64* __restrict__ err)
{
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    u64 d = uniform64(desc[nb]);
    if (DESC_K(d) != K_GAP) return;
    const u16* src = raw + src_off[nb];
    u16* dst = (u16*)(uintptr_t)DESC_P(d);
    u32 len = GMETA(d) >> 1;
    u32 padded = (len + 1u + 7u) & ~7u;
    bool bad = false;
    for (u32 k = lane; k < padded; k += 64u) {
        u32 cur = k <= len ? (u32)src[k] : 0xFFFFu;                // padding words: 0xFFFF (what k_agg_or_rows reads as "no run", bmx_kernels7.h)
        if (k >= 2u && k <= len && (u32)src[k - 1u] >= cur) bad = true;
        if (k == len && cur != 65535u) bad = true;
        dst[k] = (u16)cur;
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(reinterpret_cast<unsigned long long*>(err), 1ull);
}

// ---------------------------------------------------------------------------
// load any block kind into registers (NULL -> zeros, FULL -> ones, GAP decoded)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void blk_from_desc(u64 d, Blk& b, u32* lds, u32 lane)
{
    u32 k = DESC_K(d);
    if (k == K_BIT) blk_load(b, as_gc4(DESC_P(d)), lane);
    else if (k == K_GAP) gap_decode(as_gc16(DESC_P(d)), lds, b, lane, GMETA(d));
    else blk_fill(b, k == K_FULL ? ~0u : 0u);
}

// Count fan-in.  One device-scope atomic per wave on a single word serialises at ~12 ns each
// (15,259 waves of a 1 Gbit vector = 0.18 ms, 4x the streaming time of the kernel itself), so short
// counting kernels add into COUNT_SLOTS words, one 128-B line apart, picked by workgroup id, after a
// workgroup-level LDS reduce; k_sum_slots folds them.  (Integer adds: order-independent, exact.)
#define COUNT_SLOTS 64u
#define COUNT_SLOT_STRIDE 16u      // u64 words = 128 B
__device__ __forceinline__ void count_fanin(u32 wave_count, u64* __restrict__ slots, u32 lane, u32 wave)
{
    __shared__ u32 part[16];
    u32 nw = blockDim.x >> 6;
    if (lane == 0) part[wave] = wave_count;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (u32 i = 0; i < nw; ++i) t += part[i];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(slots + (blockIdx.x % COUNT_SLOTS) * COUNT_SLOT_STRIDE), (unsigned long long)t);
    }
}
__global__ __launch_bounds__(64)
void k_sum_slots(u64* __restrict__ slots, u64* __restrict__ out)
{
    u32 lane = threadIdx.x;
    u64 v = slots[lane * COUNT_SLOT_STRIDE];
    slots[lane * COUNT_SLOT_STRIDE] = 0;            // leave the slots clean for the next launch
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) *out = v;
}

// Fan-in with the fold inside the kernel: the LAST workgroup to arrive sums the 64 slots, leaves them clean, and
// writes the total to *out -- device memory or pinned host memory (hipHostMalloc is device-visible), so a
// host-synchronous count needs no second launch and no copy: launch, synchronise, read.
// No fences: on a multi-XCD part an agent-scope release / acquire pair costs an L2 write-back + invalidate per
// workgroup (measured: the 45 us count_and kernel became 215 us).  Everything the workgroups exchange goes through
// device-scope atomics, which execute at the memory side and are coherent across XCDs by themselves; program order
// between a workgroup's slot update and its ticket is enforced by waiting for the returning atomic (vmcnt).
// Tickets are two-level -- one counter per slot (its own 128-B line), then one global counter hit 64 times -- so no
// single word sees thousands of serialised atomics.
struct FoldOut { u64* slots; u32* done; u64* out; };     // done: [0] global ticket, [32 * (1 + s)] ticket of slot s
#define FOLD_DONE_WORDS (32u * (COUNT_SLOTS + 1u))

// adds `v` to the workgroup's slot; true for the single thread that must fold (all slots are final then)
// The ordering below (slot add performed before the ticket is drawn) rests on gfx9 semantics: a returning device-scope
// atomic is counted by vmcnt, so `s_waitcnt vmcnt(0)` after it means "performed at L2".  gfx10+ count stores / atomics
// without return in vscnt instead, and the HIP memory model as such gives relaxed atomics no order: refuse to build this
// scheme for anything but the gfx9 family (this library targets gfx950 only) rather than count wrongly there.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "fold_publish orders relaxed atomics with s_waitcnt vmcnt(0): valid on the gfx9 family (gfx90a/gfx942/gfx950) only"
#endif
__device__ __forceinline__ bool fold_publish(u64 v, FoldOut f)
{
    u32 s_ = blockIdx.x % COUNT_SLOTS;
    u64* slot = f.slots + s_ * COUNT_SLOT_STRIDE;
    if (v) {
        (void)__hip_atomic_fetch_add(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the add has been performed before the ticket is drawn
    }
    u32 expect = gridDim.x / COUNT_SLOTS + (s_ < gridDim.x % COUNT_SLOTS ? 1u : 0u);
    u32 k = __hip_atomic_fetch_add(f.done + 32u * (1u + s_), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k != expect - 1u) return false;
    __hip_atomic_store(f.done + 32u * (1u + s_), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 used = gridDim.x < COUNT_SLOTS ? gridDim.x : COUNT_SLOTS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32 g = __hip_atomic_fetch_add(f.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g != used - 1u) return false;
    __hip_atomic_store(f.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

__device__ __forceinline__ void count_fanin_fold(u32 wave_count, FoldOut f, u32 lane, u32 wave)
{
    __shared__ u32 part[16];
    u32 nw = blockDim.x >> 6;
    if (lane == 0) part[wave] = wave_count;
    __syncthreads();
    if (wave == 0) {
        u32 folder = 0;
        if (lane == 0) {
            u64 t = 0;
            for (u32 i = 0; i < nw; ++i) t += part[i];
            folder = fold_publish(t, f) ? 1u : 0u;
        }
        if (__shfl(folder, 0, 64)) {                            // the whole wave folds: one slot per lane, one round trip
            u64 v = __hip_atomic_exchange(f.slots + lane * COUNT_SLOT_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) __hip_atomic_store(f.out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Same pattern for the block KINDS of a result vector (kind counts packed 4 x 16 bit per slot: a slot sees at most
// nblocks / 64 <= 16,384 blocks): out[k] = number of result blocks of kind k.  Lets an operation that cannot produce GAP
// blocks skip the layout scan (k_scan_layout) altogether.
// packed: the wave hands over the kinds of ALL the blocks it produced (4 x 16-bit counters)
// extra != null: a device counter the workgroups bumped before their tickets (a bump allocator's cursor): the folding
// workgroup hands its value to out[4] and leaves it at zero for the next launch
__device__ __forceinline__ void kind_fanin_fold_packed(u64 wave_kinds, FoldOut f, u32 lane, u32 wave, u64* extra = nullptr)
{
    __shared__ u64 wk[16];
    u32 nw = blockDim.x >> 6;
    if (lane == 0) wk[wave] = wave_kinds;
    __syncthreads();
    if (wave == 0) {
        u32 folder = 0;
        if (lane == 0) {
            u64 t = 0;
            for (u32 i = 0; i < nw; ++i) t += wk[i];
            folder = fold_publish(t, f) ? 1u : 0u;
        }
        if (__shfl(folder, 0, 64)) {
            u64 v = __hip_atomic_exchange(f.slots + lane * COUNT_SLOT_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32 c = (u32)((v >> (16 * k)) & 0xFFFFull);
                c = wave_sum(c);
                if (lane == 0) __hip_atomic_store(f.out + k, (u64)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (extra && lane == 0) {
                const u64 cur = __hip_atomic_exchange(extra, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(f.out + 4, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
__device__ __forceinline__ void kind_fanin_fold(u32 kind /* 0..3, 4 = none */, FoldOut f, u32 lane, u32 wave)
{
    kind_fanin_fold_packed(kind < 4u ? 1ull << (16u * kind) : 0ull, f, lane, wave);
}

// bvector::count()  src/bm.h:2431 -> block_bitcount src/bmblocks.h:1710
__global__ __launch_bounds__(256)
void k_vec_count(const u64* __restrict__ desc, u32 nblocks, FoldOut fold)
{
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    u32 c = 0;
    if (nb < nblocks) {
        u64 d = uniform64(desc[nb]);
        u32 k = DESC_K(d);
        if (k == K_FULL) c = 65536u;
        else if (k == K_BIT) { Blk b; blk_load(b, as_gc4(DESC_P(d)), lane); c = wave_sum(blk_lane_popcount(b)); }
        else if (k == K_GAP) c = wave_sum(gap_lane_popcount(as_gc16(DESC_P(d)), lane, GMETA(d)));
    }
    count_fanin_fold(c, fold, lane, wave);
}

// expand a vector into raw words (nblocks_out blocks; blocks past the table are zero)
__global__ __launch_bounds__(256)
void k_vec_expand(const u64* __restrict__ desc, u32 nblocks, u32 nblocks_out, uint4* __restrict__ out)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks_out) return;
    u64 d = nb < nblocks ? uniform64(desc[nb]) : 0ull;
    Blk b;
    // the export path decodes GAP blocks with the toggle / prefix-XOR decoder: every content check of the parity tests
    // (to_words) thereby cross-checks it against the run-parallel decoder the operations use
    if (DESC_K(d) == K_GAP) gap_decode_xor(as_gc16(DESC_P(d)), lds + wave * 2048u, b, lane, GMETA(d));
    else blk_from_desc(d, b, lds + wave * 2048u, lane);
    blk_store(b, as_g4(out + (size_t)nb * 512u), lane);
}

// ---------------------------------------------------------------------------
// Pipeline "complete()": for every block column and arg-group, classify the
// operands once (aggregator::sort_input_blocks_and / _or, src/bmaggregator.h:
// 2315,2278) into a row  [hdr, flags, AND region (n_and), SUB region (n_sub)]:
// bit-block pointers are packed from the front of a region, GAP pointers from
// its back.  hdr = nbit_and | ngap_and<<16 | nbit_sub<<32 | ngap_sub<<48.
// One thread per (column, group).
// ---------------------------------------------------------------------------
struct PipeOperands {
    const u64* const* desc;     // per operand: descriptor table
    const u32* nblocks;         // per operand: table length
    const u32* and_off; const u32* and_n;   // per group: slice of the operand arrays (AND)
    const u32* sub_off; const u32* sub_n;   // per group: (SUB)
    const u32* row_off;         // per group: offset of its row inside a column record
};

// One WAVE per (column, group): lane k classifies operand k (64 at a time), list positions come from
// ballots (bit pointers keep operand order from the front, GAP pointers from the back).  Every operand's
// descriptor is read in the same round trip -- a thread walking its operands one by one paid three dependent
// reads per operand (0.4 ms for 256 operands however short the vectors are).
// `drop` = kind that is skipped (NULL in a SUB / OR list, FULL in an AND list), `kill` = kind that decides the
// whole row (NULL in an AND list, FULL in a SUB / OR list).
__device__ __forceinline__ bool sort_operands(const u64* const* __restrict__ desc, const u32* __restrict__ nblocks,
                                              u32 off, u32 n, u32 c, u32 kill, u64* __restrict__ region,
                                              u32& nbit, u32& ngap, u32 lane)
{
    bool killed = false;
    nbit = 0; ngap = 0;
    for (u32 base = 0; base < n; base += 64u) {
        u32 k = base + lane;
        bool valid = k < n;
        u32 op = off + (valid ? k : n - 1u);
        u32 nb = nblocks[op];
        const u64* dp = desc[op];
        u64 d = dp[c < nb ? c : 0u];                         // every table has at least one entry
        if (!valid || c >= nb) d = 0ull;
        u32 kd = valid ? DESC_K(d) : 4u;
        if (__ballot(valid && kd == kill) != 0ull) killed = true;
        u64 bit_m = __ballot(kd == K_BIT), gap_m = __ballot(kd == K_GAP);
        u64 lt = (1ull << lane) - 1ull;
        if (kd == K_BIT) region[nbit + (u32)__popcll(bit_m & lt)] = DESC_P(d);
        if (kd == K_GAP) region[n - 1u - (ngap + (u32)__popcll(gap_m & lt))] = d & 0x3FFFFFFFFFFFFFFFull;   // pointer | GMETA << 48 (len, start bit): consumers mask (as_gc16) or use it (bmx_kernels9.h)
        nbit += (u32)__popcll(bit_m); ngap += (u32)__popcll(gap_m);
    }
    return killed;
}

__global__ __launch_bounds__(256)
void k_pipe_sort(PipeOperands po, u32 ngroups, u32 ncols, u32 col_stride, u64* __restrict__ dmat)
{
    u32 lane = lane_id();
    u64 wid = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= (u64)ncols * ngroups) return;
    u32 c = (u32)(wid / ngroups), g = (u32)(wid - (u64)c * ngroups);
    u64* row = dmat + (size_t)c * col_stride + po.row_off[g];
    u32 na = po.and_n[g], ns = po.sub_n[g];
    u32 ao = po.and_off[g], so = po.sub_off[g];
    u32 nbit = 0, ngap = 0, sbit = 0, sgap = 0;
    bool empty = (na == 0);
    if (na && sort_operands(po.desc, po.nblocks, ao, na, c, K_NULL, row + 2, nbit, ngap, lane)) empty = true;   // any NULL => empty column (:2327); FULL operands are dropped (:2346)
    if (!empty && ns && sort_operands(po.desc, po.nblocks, so, ns, c, K_FULL, row + 2 + na, sbit, sgap, lane)) empty = true;   // FULL in the SUB group => empty (:1746)
    u64 flags = 0;
    if (empty) flags = ROW_EMPTY;
    else if (!nbit && !ngap) { if (!sbit && !sgap) flags = ROW_FULL; else flags = ROW_ONES; }  // all FULL (:1751)
    else if (!nbit) flags = ROW_ONES;                        // GAP-only AND group (:2033)
    if (lane == 0) {
        row[0] = (u64)nbit | ((u64)ngap << 16) | ((u64)sbit << 32) | ((u64)sgap << 48);
        row[1] = flags;
    }
}

// XCD-aware workgroup remap (bijective for any grid size): hardware places
// workgroup b on XCD b % 8; give every XCD one contiguous slice of the work so
// neighbouring block columns (same 2 MiB pages of every operand slab) stay on
// one XCD's L2/TLB.
__device__ __forceinline__ u32 xcd_remap(u32 b, u32 nwg)
{
    u32 q = nwg >> 3, rem = nwg & 7u;
    u32 x = b & 7u, i = b >> 3;
    return x * q + (x < rem ? x : rem) + i;
}

// ---------------------------------------------------------------------------
// THE hot kernel: fused N-way AND(-SUB) + COUNT, counts only
// (aggregator::combine_and_sub(pipe) with agg_opt_only_counts,
//  src/bmaggregator.h:1292-1399; per column :1720; bit-block chain :1994,2125;
//  GAP operands :1820,1854;  count[g] += is_full ? 65536 : popcount(result)).
// One wave = one (column, group) work item; accumulator = 32 VGPRs; U operand
// blocks (U x 8 KiB) are in flight per wave; the running result is tested for
// all-zero after every batch (the reference's digest==0 early exit, :2052).
// ---------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256)
void k_pipe_counts(const u64* __restrict__ dmat, const u32* __restrict__ row_off,
                   const u32* __restrict__ and_n, const u32* __restrict__ sub_n, u32 col_stride,
                   u32 ngroups, u32 col_from, u32 nitems, int xcd_swz, u64* __restrict__ counts)
{
    extern __shared__ u32 lds_dyn[];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 item = uniform32(bid * 4u + wave);
    if (item >= nitems) return;
    u32 c = item / ngroups, g = item - c * ngroups;
    const u64* row = dmat + (size_t)(col_from + c) * col_stride + row_off[g];
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) return;
    if (flags & ROW_FULL) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), 65536ull); return; }
    u32 nba = (u32)(hdr & 0xFFFFu), nga = (u32)((hdr >> 16) & 0xFFFFu);
    u32 nbs = (u32)((hdr >> 32) & 0xFFFFu), ngs = (u32)(hdr >> 48);
    u32 na = uniform32(and_n[g]), ns = uniform32(sub_n[g]);
    const u64* pa = row + 2;
    const u64* ps = pa + na;
    u32* lds = lds_dyn + wave * 2048u;

    Blk acc;
    blk_fill(acc, ~0u);
    // bit-block operands: software-pipelined fold (bmx_device.h pipe_chain), AND group then SUB group
    if (pipe_chain<U, true, 0>(acc, pa, nba, lane)) return;
    if (pipe_chain<U, true, 1>(acc, ps, nbs, lane)) return;
    // GAP operands (packed from the back of each region): applied run-by-run to the accumulator in LDS
    if (nga | ngs) {
        blk_to_lds(acc, lds, lane);
        if (nga && gap_apply_list<GAP_AND>(pa + na - 1u, nga, lds, lane)) return;
        if (ngs && gap_apply_list<GAP_SUB>(ps + ns - 1u, ngs, lds, lane)) return;
        blk_from_lds(acc, lds, lane);
    }
    u32 cnt = wave_sum(blk_lane_popcount(acc));
    if (lane == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)cnt);
}

// ---------------------------------------------------------------------------
// Bit-block-only fast path of the same computation (pipelines whose operands
// contain no GAP block: the headline case), software-pipelined (pipe_chain):
//  * operand pointers are wave-uniform and read with scalar loads one batch ahead;
//  * two register buffers: the loads of batch n+1 are issued before batch n is
//    consumed, so a wave always has U..2U slices in flight;
//  * the tail batch re-uses its last operand (AND / AND-NOT are idempotent), so
//    there is no remainder loop.
// A work item is one ROWS/8 slice of a (column, group): ROWS register rows = ROWS KiB of
// every operand block, slices of one column adjacent in the item order (the waves of a
// workgroup read one contiguous stretch of every operand).  ROWS = 8 (whole blocks) is the
// measured best when there are >= ~12 k columns (tools/tune_pipe.py); a block-range shard
// of a multi-GPU job (1,907 columns per GPU for 1e9 bits on 8 GPUs) is cut into smaller
// slices so that the chip still sees thousands of independent waves (bmx.hip pipe_rows_auto);
// the early-exit test is then per slice (finer than the reference's digest, same result).
// Measured alternatives that did NOT win on MI355X at full size (same box, interleaved A/B,
// 256 x 1e9 bits): occupancy pinned to 8 waves/SIMD with a one-block-in-flight loop (-7 %),
// hand-placed asm loads with counted vmcnt (-5 %, and hipcc may copy an asm-loaded register
// before the wait), dropping the early-exit test (-2 %), a persistent grid drawing columns
// from per-XCD ticket counters (+-0.5 %: turnover and tail are not the gap).
// ---------------------------------------------------------------------------
template <int U, bool NT, int WG = 256, int ROWS = 8>
__global__ __launch_bounds__(WG)
void k_pipe_counts_bits2(const u64* __restrict__ dmat, const u32* __restrict__ row_off,
                         const u32* __restrict__ and_n, u32 col_stride,
                         u32 ngroups, u32 col_from, u32 nitems, int xcd_swz, u64* __restrict__ counts)
{
    constexpr u32 PARTS = 8 / ROWS;
    u32 lane = lane_id(), wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 item = uniform32(bid * wpb + wave);
    if (item >= nitems) return;
    u32 part = item % PARTS, cg = item / PARTS;
    u32 c = cg / ngroups, g = cg - c * ngroups;
    const u64* row = dmat + (size_t)(col_from + c) * col_stride + row_off[g];
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) return;
    if (flags & ROW_FULL) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)(65536u / PARTS)); return; }
    u32 nba = (u32)(hdr & 0xFFFFu), nbs = (u32)((hdr >> 32) & 0xFFFFu);
    u32 na = uniform32(and_n[g]);
    const u64* pa = row + 2;
    const u64* ps = pa + na;
    const u32 poff = part * ROWS * 64u;          // in 16-byte units
    Part<ROWS> acc;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc.r[i] = (u32x4)(~0u);
    if (pipe_chain<U, NT, 0, ROWS>(acc, pa, nba, lane, poff)) return;
    if (pipe_chain<U, NT, 1, ROWS>(acc, ps, nbs, lane, poff)) return;
    u32 cnt = 0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        cnt += __popcll(((u64)acc.r[i].y << 32) | acc.r[i].x);
        cnt += __popcll(((u64)acc.r[i].w << 32) | acc.r[i].z);
    }
    cnt = wave_sum(cnt);
    if (lane == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)cnt);
}

#ifdef BMX_DIAG
// same read pattern through raw buffer loads with an explicit cache policy (AUX: 1 = sc0, 2 = nt, 16 = sc1):
// measures what the memory system does with each policy for a pure stream
template <int AUX>
__global__ __launch_bounds__(256)
void k_diag_stream_read_buf(const uint4* __restrict__ buf, u64 nblocks8k, u32 blocks_per_wave, int pattern, int xcd_swz, u64* __restrict__ sink)
{
    u32 lane = lane_id();
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u64 w = (u64)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    u64 nwaves = nblocks8k / blocks_per_wave;
    if (w >= nwaves) return;
    u64 b0 = pattern ? w : w * blocks_per_wave;
    u64 step = pattern ? nwaves : 1;
    u32x4 acc = (u32x4)(0u);
    for (u32 j = 0; j < blocks_per_wave; ++j) {
        u64 addr = uniform64((u64)(uintptr_t)(buf + (b0 + j * step) * 512u));
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)addr, 0, 8192, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 1024 + lane * 16), 0, AUX);
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) sink[0] = 1;
}

#endif  // BMX_DIAG

// algorithmic operand bytes of the rows in [col_from, col_from+ncols)
__global__ __launch_bounds__(256)
void k_pipe_bytes(const u64* __restrict__ dmat, const u32* __restrict__ row_off, const u32* __restrict__ and_n,
                  const u32* __restrict__ sub_n, u32 col_stride, u32 ngroups, u32 col_from, u32 nitems,
                  u64* __restrict__ total)
{
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 bytes = 0;
    if (tid < nitems) {
        u32 c = (u32)(tid / ngroups), g = (u32)(tid - (u64)c * ngroups);
        const u64* row = dmat + (size_t)(col_from + c) * col_stride + row_off[g];
        u64 hdr = row[0], flags = row[1];
        if (!(flags & (ROW_EMPTY | ROW_FULL))) {
            u32 nba = (u32)(hdr & 0xFFFFu), nga = (u32)((hdr >> 16) & 0xFFFFu);
            u32 nbs = (u32)((hdr >> 32) & 0xFFFFu), ngs = (u32)(hdr >> 48);
            bytes = (u64)(nba + nbs) * 8192ull;
            const u64* pa = row + 2; const u64* ps = pa + and_n[g];
            for (u32 i = 0; i < nga; ++i) bytes += 2ull * ((as_gc16(pa[and_n[g] - 1u - i])[0] >> 3) + 1u);
            for (u32 i = 0; i < ngs; ++i) bytes += 2ull * ((as_gc16(ps[sub_n[g] - 1u - i])[0] >> 3) + 1u);
        }
    }
    // block reduce
    __shared__ u64 s[256];
    s[threadIdx.x] = bytes; __syncthreads();
    for (u32 o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0 && s[0]) atomicAdd(reinterpret_cast<unsigned long long*>(total), (unsigned long long)s[0]);
}

#ifdef BMX_DIAG
// ---------------------------------------------------------------------------
// diagnostics: plain streaming read of a large buffer (practical HBM ceiling of
// the box, measured next to the product kernels by tools/tune_pipe.py)
// mode 0: one wave per contiguous CHUNK (8 KiB x chunk_blocks), plain loads
// mode 1: same with non-temporal loads
// ---------------------------------------------------------------------------
// pattern 0: wave w reads blocks [w*bpw, (w+1)*bpw) (contiguous); pattern 1: wave w reads block
// j*nwaves + w for j < bpw (the access pattern of an N-way aggregation over N separate vectors)
template <bool NT>
__global__ __launch_bounds__(256)
void k_diag_stream_read(const uint4* __restrict__ buf, u64 nblocks8k, u32 blocks_per_wave, int pattern, int xcd_swz, u64* __restrict__ sink)
{
    u32 lane = lane_id();
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u64 w = (u64)bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    u64 nwaves = nblocks8k / blocks_per_wave;
    if (w >= nwaves) return;
    u64 b0 = pattern ? w : w * blocks_per_wave;
    u64 step = pattern ? nwaves : 1;
    u32x4 acc = (u32x4)(0u);
    for (u32 j = 0; j < blocks_per_wave; ++j) {
        Part<8> t;
        part_load<8, NT>(t, as_gc4(buf + (b0 + j * step) * 512u), lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc |= t.r[i];
    }
    if ((acc.x | acc.y | acc.z | acc.w) == 0x12345678u) sink[0] = 1;   // never true for the memset pattern; defeats DCE
}
#endif  // BMX_DIAG
