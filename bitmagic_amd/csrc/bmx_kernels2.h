// bmx_kernels2.h -- pairwise set algebra, aggregator with materialised results,
// rank/select.  Same wave-per-block register image as bmx_kernels.h.
#pragma once
#include "bmx_kernels.h"

// ---------------------------------------------------------------------------
// Result store.  A materialised result vector owns a full-size slab (one 8 KiB
// slot per block column: HBM is plentiful, no compaction pass, no second read);
// the block rule of blocks_manager::opt_copy_bit_block (src/bmblocks.h:1355):
//   runs == 1 -> NULL / FULL (nothing stored)
//   opt_compress && runs < 1276 -> GAP candidate: raw bits parked in the slot,
//                                  converted by k_emit_gaps afterwards
//   else BIT in its slot.
// All 64 lanes call; st[nb] and desc[nb] are written by lane 0.
// ---------------------------------------------------------------------------
// How a produced block is classified.  ST_OPT: the opt_copy_bit_block rule above.  The 3-operand ops of the
// reference do NOT treat every block alike (combine_operation_block_*, src/bm.h:7100-7340, Appendix A.1):
//   ST_FORCE_BIT   a bit-block that is only copied (x op NULL, x AND FULL, ~x for XOR FULL) stays a bit-block,
//                  whatever it holds and whatever opt_mode says (clone_assign_block, src/bmblocks.h:894)
//   ST_FORCE_GAP   a copied GAP block and a GAP x GAP result stay GAP in either mode (clone_gap_block :865:
//                  all-zero -> NULL, too long -> bit-block)
//   ST_TEST_ZERO / ST_TEST_ONE   without opt_compress a computed block is stored as a bit-block unless the
//                  operation itself tests it: all-zero -> NULL (B x B except OR, any AND, GAP - B), all-ones -> FULL (B OR B)
//   ST_GAP_RESULT  a GAP x GAP result goes through clone_gap_block(i, j, tmp_buf, len) (src/bmblocks.h:865-889,
//                  called from combine_operation_block_* src/bm.h:7133,6975,7050,7319): only an all-ZERO level-0
//                  result is dropped; an all-ones result stays a GAP block of ONE run (it does not become FULL)
enum { ST_OPT = 1, ST_FORCE_BIT = 2, ST_TEST_ZERO = 4, ST_TEST_ONE = 8, ST_FORCE_GAP = 16, ST_GAP_RESULT = 32 };

// bit_block_to_gap (src/bmfunc.h:5542) from the transition masks of a block (blk_transitions): run k ends just before the
// k-th transition.  Writes the block's len + 1 words at g (16-byte aligned) and its 0xFFFF padding; all 64 lanes call.
// The words are collected in `stage` (2,560 bytes of LDS private to the wave: 2-byte writes at the index a wave scan gives
// every lane) and leave as whole 16-byte chunks -- written straight to memory, one 2-byte store per run end, a 1,270-run
// block took ~12 us (round 5: measured behind k_op2_loop, profiles/r05_pair).
__device__ __forceinline__ void gap_emit_from_transitions(const Blk& t, u32 len, u32 first, u16* __restrict__ g, u32 lane, u16* __restrict__ stage)
{
    u32 idx_base = 1u;
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
                stage[idx++] = (u16)((wbase + j) * 32u + k - 1u);
            }
        }
        idx_base += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
        u32 level = len <= 124u ? 0u : len <= 252u ? 1u : len <= 508u ? 2u : 3u;
        stage[0] = (u16)((len << 3) | (level << 1) | first);
        stage[len] = 65535u;
    }
    // padding words up to the next 16-byte boundary read 0xFFFF: no run end but a block's last has that value, which is how
    // k_agg_or_rows (bmx_kernels7.h) tells a run from padding without the block's length
    if (lane >= 1u && lane <= 7u && len + lane < ((len + 1u + 7u) & ~7u)) stage[len + lane] = 0xFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const u32 nch = (len + 8u) >> 3;                              // 16-byte chunks holding words 0..len
    const u32x4* s4 = reinterpret_cast<const u32x4*>(stage);
    u32x4* g4 = reinterpret_cast<u32x4*>(g);
    for (u32 ci = lane; ci < nch; ci += 64u) g4[ci] = s4[ci];
    __builtin_amdgcn_wave_barrier();                              // (the stage is the wave's again)
}

// gap_offs != null: the producing kernel also lays its GAP candidates out -- a bump allocation of their 16-byte-padded words
// from *gap_cursor (low 40 bits: words, above: blocks), the offset left in gap_offs[nb] -- and converts them itself before it
// ends (k_op2_loop's tail), into a slab the host sized at the operands' bound: neither a layout scan nor a conversion kernel runs
// behind it (the order of the blocks in the GAP slab is then the order of arrival).
template <bool SNT = false>
__device__ __forceinline__ u32 store_result_mode(const Blk& acc, u32 nb, u32 mode,
                                                  uint4* __restrict__ slab, u64* __restrict__ desc,
                                                  BlockStat* __restrict__ st, u32 lane,
                                                  u32* __restrict__ gap_offs = nullptr, u64* __restrict__ gap_cursor = nullptr,
                                                  u32* pop_out = nullptr, u32* gap_info = nullptr /* K_GAP with gap_offs: {offset, len << 1 | first} */)
{
    Blk t;
    u32 pop = wave_sum(blk_lane_popcount(acc));
    if (pop_out) *pop_out = pop;
    u32 runs = 1u + wave_sum(blk_transitions(acc, t, lane));
    u32 first = __shfl(acc.r[0].x, 0, 64) & 1u;
    u32 kind;
    if (mode & ST_FORCE_BIT) kind = K_BIT;
    else if (mode & (ST_FORCE_GAP | ST_OPT)) {
        kind = (runs == 1u) ? (first ? K_FULL : K_NULL) : (runs < 1276u ? K_GAP : K_BIT);
        if ((mode & ST_GAP_RESULT) && runs == 1u && first) kind = K_GAP;      // 1-run GAP block [hdr, 65535]
    } else {
        kind = K_BIT;
        if ((mode & ST_TEST_ZERO) && runs == 1u && !first) kind = K_NULL;
        if ((mode & ST_TEST_ONE) && runs == 1u && first) kind = K_FULL;
    }
    uint4* slot = slab + (size_t)nb * 512u;
    if (kind == K_BIT || kind == K_GAP) {
        if (SNT) {                                               // written once, read by a later kernel: keep it out of the way of the operand streams
            gptr4 p = as_g4(slot);
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(acc.r[i], &p[i * 64 + lane]);
        } else blk_store(acc, as_g4(slot), lane);
    }
    u32 off = 0u;
    if (lane == 0) {
        st[nb] = BlockStat{pop, runs, first, kind};
        desc[nb] = (kind == K_BIT) ? DESC_MAKE(slot, K_BIT) : DESC_MAKE(0, kind == K_GAP ? K_NULL : kind);
        if (kind == K_GAP && gap_offs) {
            // one atomic for both: the low 40 bits of the cursor count padded words, the bits above count candidates
            const u64 cur = __hip_atomic_fetch_add(gap_cursor, (1ull << 40) | (u64)((runs + 1u + 7u) & ~7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gap_offs[nb] = off = (u32)(cur & 0xFFFFFFFFFFull);
        }
    }
    if (gap_info) { gap_info[0] = off; gap_info[1] = (runs << 1) | first; }     // (lane 0's values are the ones that count)
    return kind;
}

__device__ __forceinline__ void store_result(const Blk& acc, u32 nb, int opt_compress,
                                             uint4* __restrict__ slab, u64* __restrict__ desc,
                                             BlockStat* __restrict__ st, u32 lane, bool full_as_bit = false)
{
    // aggregator results: opt_copy_bit_block(.., opt_mode, ..); without opt_compress a block is stored as it is
    // except that an empty / full one carries no storage here (full_as_bit: copy_bit_block keeps all-ones as bits)
    Blk t;
    u32 pop = wave_sum(blk_lane_popcount(acc));
    u32 runs = 1u + wave_sum(blk_transitions(acc, t, lane));
    u32 first = __shfl(acc.r[0].x, 0, 64) & 1u;
    u32 kind = (runs == 1u) ? (first ? K_FULL : K_NULL)
             : ((opt_compress && runs < 1276u) ? K_GAP : K_BIT);
    if (full_as_bit && kind == K_FULL) kind = K_BIT;         // copy_bit_block (src/bmblocks.h:1340) keeps all-ones as bits
    uint4* slot = slab + (size_t)nb * 512u;
    if (kind == K_BIT || kind == K_GAP) blk_store(acc, as_g4(slot), lane);
    if (lane == 0) {
        st[nb] = BlockStat{pop, runs, first, kind};
        desc[nb] = (kind == K_BIT) ? DESC_MAKE(slot, K_BIT) : DESC_MAKE(0, kind == K_GAP ? K_NULL : kind);
    }
}

__device__ __forceinline__ void store_trivial(u32 kind, u32 nb, u64* __restrict__ desc,
                                              BlockStat* __restrict__ st, u32 lane)
{
    if (lane == 0) {
        st[nb] = BlockStat{kind == K_FULL ? 65536u : 0u, 1u, kind == K_FULL ? 1u : 0u, kind};
        desc[nb] = DESC_MAKE(0, kind);
    }
}

// GAP conversion of the parked candidates
__global__ __launch_bounds__(256)
void k_emit_gaps(const uint4* __restrict__ slab, u32 nblocks, const BlockStat* __restrict__ st,
                 const u32* __restrict__ offs, u16* __restrict__ gap_slab, u64* __restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) u16 stage[4][1280];
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    if (uniform32(st[nb].kind) != K_GAP) return;
    Blk b, t;
    blk_load(b, as_gc4(slab + (size_t)nb * 512u), lane);
    (void)blk_transitions(b, t, lane);
    u16* g = gap_slab + offs[nb];
    u32 len = uniform32(st[nb].runs);
    u32 first = uniform32(st[nb].first);
    gap_emit_from_transitions(t, len, first, g, lane, stage[threadIdx.x >> 6]);
    if (lane == 0) desc[nb] = DESC_MAKE_GAP(g, len, first);
}

// ord[nb] = number of bit-blocks before block nb (what k_scan_layout leaves in offs[] for bit-blocks), from the descriptor
// table alone: one workgroup, 1024 blocks per step
__global__ __launch_bounds__(1024)
void k_ord_from_desc(const u64* __restrict__ desc, u32 nblocks, u32* __restrict__ ord)
{
    __shared__ u32 wsum[16];
    const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u32 carry = 0;
    for (u32 base = 0; base < nblocks; base += 1024u) {
        const u32 nb = base + tid;
        const bool bit = nb < nblocks && DESC_K(desc[nb]) == K_BIT;
        const u64 m = __ballot(bit);
        const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        if (lane == 0) wsum[w] = (u32)__popcll(m);
        __syncthreads();
        u32 off = carry, tot = 0;
#pragma unroll
        for (u32 i = 0; i < 16; ++i) { const u32 x = wsum[i]; if (i < w) off += x; tot += x; }
        __syncthreads();
        if (nb < nblocks) ord[nb] = bit ? off + before : 0u;
        carry += tot;
    }
}

// result slab -> right-sized slab: bit-block nb moves to ordinal offs[nb] (from k_scan_layout) and its
// descriptor follows.  One wave per block column.
__global__ __launch_bounds__(256)
void k_compact_bits(const uint4* __restrict__ slab, u32 nblocks, const BlockStat* __restrict__ st,
                    const u32* __restrict__ offs, uint4* __restrict__ packed, u64* desc)
{
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    if (uniform32(st ? st[nb].kind : DESC_K(desc[nb])) != K_BIT) return;    // (st == null: the kinds as the descriptor table has them)
    Blk b;
    blk_load(b, as_gc4(slab + (size_t)nb * 512u), lane);
    uint4* dst = packed + (size_t)uniform32(offs[nb]) * 512u;
    blk_store(b, as_g4(dst), lane);
    if (lane == 0) desc[nb] = DESC_MAKE(dst, K_BIT);
}

// download of a slab with unused slots: bit-block nb -> out[ord[nb]]
__global__ __launch_bounds__(256)
void k_gather_bits(const u64* __restrict__ desc, const u32* __restrict__ ord, u32 nblocks, uint4* __restrict__ out)
{
    u32 lane = lane_id();
    u32 nb = uniform32(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (nb >= nblocks) return;
    u64 d = uniform64(desc[nb]);
    if (DESC_K(d) != K_BIT) return;
    Blk b;
    blk_load(b, as_gc4(DESC_P(d)), lane);
    blk_store(b, as_g4(out + (size_t)uniform32(ord[nb]) * 512u), lane);
}

// descriptors of a cloned vector: same kinds, pointers moved into the clone's slabs
__global__ __launch_bounds__(256)
void k_rebase_desc(const u64* in, u64* out, u32 n, u64 old_bits, u64 new_bits, u64 old_gaps, u64 new_gaps)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 d = in[i];
    u32 k = DESC_K(d);
    if (k == K_BIT) d = (d & ~0x0000FFFFFFFFFFFFull) | (DESC_P(d) - old_bits + new_bits);
    else if (k == K_GAP) d = (d & ~0x0000FFFFFFFFFFFFull) | (DESC_P(d) - old_gaps + new_gaps);
    out[i] = d;
}

// ---------------------------------------------------------------------------
// Pairwise ops.  bvector::bit_and/or/xor/sub(bv1, bv2)  src/bm.h:6185,5973,6072,6403
// per block: combine_operation_block_* (:7100,6945,7018,7285); NULL/FULL
// shortcuts of SURVEY Appendix A.1 never touch block data.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 desc_at(const u64* __restrict__ d, u32 n, u32 nb)
{
    return nb < n ? uniform64(d[nb]) : 0ull;
}

// NULL / FULL results that need no block data; 4 = the block has to be computed
__device__ __forceinline__ u32 op2_trivial(int op, u32 ka, u32 kb)
{
    if (op == BMX_AND) {
        if (ka == K_NULL || kb == K_NULL) return K_NULL;
        if (ka == K_FULL && kb == K_FULL) return K_FULL;
    } else if (op == BMX_OR) {
        if (ka == K_FULL || kb == K_FULL) return K_FULL;
        if (ka == K_NULL && kb == K_NULL) return K_NULL;
    } else if (op == BMX_XOR) {
        if ((ka == K_NULL && kb == K_NULL) || (ka == K_FULL && kb == K_FULL)) return K_NULL;
        if ((ka == K_NULL && kb == K_FULL) || (ka == K_FULL && kb == K_NULL)) return K_FULL;
    } else {
        if (ka == K_NULL || kb == K_FULL) return K_NULL;
        if (ka == K_FULL && kb == K_NULL) return K_FULL;
    }
    return 4u;
}

// how the produced block is classified: the reference case by case (see ST_* above)
__device__ __forceinline__ u32 op2_store_mode(int op, u32 ka, u32 kb, int opt_compress)
{
    u32 copy_of = 4u;                                        // kind of the operand that is merely copied (4 = none)
    if (op == BMX_AND) { if (ka == K_FULL) copy_of = kb; else if (kb == K_FULL) copy_of = ka; }
    else if (op == BMX_OR) { if (ka == K_NULL) copy_of = kb; else if (kb == K_NULL) copy_of = ka; }
    else if (op == BMX_XOR) { if (ka == K_NULL || ka == K_FULL) copy_of = kb; else if (kb == K_NULL || kb == K_FULL) copy_of = ka; }
    else { if (kb == K_NULL) copy_of = ka; }
    if (copy_of == K_BIT) return ST_FORCE_BIT;
    if (copy_of == K_GAP) return ST_FORCE_GAP;
    if (ka == K_GAP && kb == K_GAP) return ST_FORCE_GAP | ST_GAP_RESULT;
    if (opt_compress) return ST_OPT;
    const bool bb = ka != K_GAP && kb != K_GAP;               // bit x bit (FULL in SUB counts as a real all-ones block)
    u32 mode = 0u;
    if ((bb && op != BMX_OR) || op == BMX_AND || (ka == K_GAP && op == BMX_SUB)) mode |= ST_TEST_ZERO;
    if (bb && op == BMX_OR) mode |= ST_TEST_ONE;
    return mode;
}

// one result block of a pairwise operation; returns its kind
__device__ __forceinline__ u32 op2_block(int op, u64 a, u64 b, u32 nb, int opt_compress, u32* l,
                                         uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, u32 lane, u32* pop_out = nullptr)
{
    u32 ka = DESC_K(a), kb = DESC_K(b);
    // shortcuts that produce NULL / FULL without reading anything
    const u32 trivial = op2_trivial(op, ka, kb);
    if (trivial != 4u) { store_trivial(trivial, nb, desc, st, lane); if (pop_out) *pop_out = trivial == K_FULL ? 65536u : 0u; return trivial; }
    Blk x, y;
    blk_from_desc(a, x, l, lane);
    blk_from_desc(b, y, l, lane);
    blk_op(op, x, y);
    return store_result_mode(x, nb, op2_store_mode(op, ka, kb, opt_compress), slab, desc, st, lane, nullptr, nullptr, pop_out);
}

// kinds.slots != null: the kind counts of the result are folded inside the kernel (kind_fanin_fold) -- used when no GAP
// block can come out (neither operand holds one and opt_compress is off), so that no layout scan is needed.
__global__ __launch_bounds__(256)
void k_op2(int op, const u64* __restrict__ da, u32 na, const u64* __restrict__ db, u32 nbk,
           u32 nblocks, int opt_compress, uint4* __restrict__ slab, u64* __restrict__ desc,
           BlockStat* __restrict__ st, FoldOut kinds, FoldOut total /* slots != null: the popcount of the result is folded too (bit_and + count() in one launch) */)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    u32 kind = 4u, pop = 0u;
    if (nb < nblocks)
        kind = op2_block(op, desc_at(da, na, nb), desc_at(db, nbk, nb), nb, opt_compress, lds + wave * 2048u, slab, desc, st, lane, &pop);
    if (kinds.slots) kind_fanin_fold(kind, kinds, lane, wave);
    if (total.slots) count_fanin_fold(pop, total, lane, wave);
}

// bm::count_and/or/xor/sub  src/bmalgo.h:49,149,81,115 (distance_operation,
// src/bmalgo_impl.h:766,853): popcount(a OP b) without materialising.
__global__ __launch_bounds__(256)
void k_count_op2(int op, const u64* __restrict__ da, u32 na, const u64* __restrict__ db, u32 nbk,
                 u32 nblocks, FoldOut fold)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    u32 c = 0;
    if (nb < nblocks) {
        u64 a = desc_at(da, na, nb), b = desc_at(db, nbk, nb);
        u32 ka = DESC_K(a), kb = DESC_K(b);
        bool skip = (ka == K_NULL && kb == K_NULL) || (op == BMX_AND && (ka == K_NULL || kb == K_NULL)) ||
                    (op == BMX_SUB && (ka == K_NULL || kb == K_FULL));
        if (!skip) {
            Blk x, y;
            u32* l = lds + wave * 2048u;
            blk_from_desc(a, x, l, lane);
            blk_from_desc(b, y, l, lane);
            blk_op(op, x, y);
            c = wave_sum(blk_lane_popcount(x));
        }
    }
    count_fanin_fold(c, fold, lane, wave);
}

// bm::count_* over operands of ANY block kinds, persistent form (round 3).  k_count_op2 above pays, per column, a wave
// launch, a descriptor round trip, the block loads, and -- for a GAP operand -- one more round trip per 512 runs (2-byte
// gathers, batch after batch) before anything is counted; the mixed 1 % case of BASELINE configs[1] (two thirds bit-blocks,
// one third ~1,300-run GAP blocks) ran at 53 % of HBM with less data than the all-bit case.  Here a wave stays and walks
// every (grid waves)-th column: the descriptors of its next column are fetched while the current one is counted, and ALL
// the loads of a column are issued before anything is decoded -- 8 x 16 B per lane for a bit-block, the whole run list of
// a GAP block as 3 x 16 B per lane (<= 1,280 words; lanes past the end re-read chunk 0) -- so a column costs ONE memory
// round trip whatever its kinds; the GAP block is then set into the wave's LDS block straight from those registers
// (gap_apply_chunk: the 4 wanted runs of a 16-B chunk, the first word of the next chunk comes from the neighbour lane).
// The loads of an operand are issued WITHOUT control flow, whatever its kind: eight 16-byte loads per lane from one base
// pointer -- the eight rows of a bit-block; for a GAP block its run list in the first three (chunk index clamped to the
// block, the other five re-read chunk 0: one cached line); NULL / FULL read `safe` (any 16 readable bytes).  (With a branch
// per kind around the loads hipcc merges the two register images of the operand with copies that WAIT for the loads inside
// the issue phase: the second operand's loads then leave after the first operand's have landed.)
template <bool NT>
__device__ __forceinline__ void op2_issue(u64 d, const u64* __restrict__ safe, Blk& x, u32 lane)
{
    const u32 k = DESC_K(d);
    const bool bit = k == K_BIT, gap = k == K_GAP;
    const u64 base = (bit || gap) ? DESC_P(d) : (u64)(uintptr_t)safe;
    const u32 nch = gap ? ((GMETA(d) >> 1) + 8u) >> 3 : 0u;       // 16-B chunks holding words 0..len of a GAP block
    gcptr4 p = as_gc4(base);
#pragma unroll
    for (u32 j = 0; j < 8; ++j) {
        const u32 ci = j * 64u + lane;
        const u32 at = bit ? ci : (ci < nch ? ci : 0u);
        x.r[j] = NT ? __builtin_nontemporal_load(&p[at]) : p[at];
    }
}

__device__ __forceinline__ void op2_finish(u64 d, Blk& x, u32* l, u32 lane)
{
    const u32 k = DESC_K(d);
    if (k == K_BIT) return;
    if (k != K_GAP) { blk_fill(x, k == K_FULL ? ~0u : 0u); return; }
    const u32 meta = GMETA(d), len = meta >> 1;
    const bool odd_runs = (meta & 1u) != 0u;                     // wanted (1-)runs are 1,3,5,.. when the block starts with 1
    const u32 nch = (len + 8u) >> 3;
    u32x4* l4 = reinterpret_cast<u32x4*>(l);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 64 + lane] = (u32x4)(0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (u32 j = 0; j < 3; ++j) {
        if (j * 64u >= nch) break;                                // wave-uniform
        const u32 ci = j * 64u + lane;
        u32 nx = __shfl_down(x.r[j].x, 1, 64);
        const u32 wrap = j < 2u ? __builtin_amdgcn_readfirstlane(x.r[j < 2u ? j + 1u : j].x) : 0u;
        if (lane == 63u) nx = wrap;
        if (ci < nch) {
            const u32 xs[5] = {x.r[j].x, x.r[j].y, x.r[j].z, x.r[j].w, nx};
            gap_apply_chunk<GAP_OR>(l, xs, ci, len, odd_runs);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) x.r[i] = l4[i * 64 + lane];
    __builtin_amdgcn_wave_barrier();
}

// NT: non-temporal loads.  Register allocation held to four waves per SIMD (128 VGPRs: two block images + the decode)
template <int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4)))
void k_count_op2_loop(int op, const u64* __restrict__ da, u32 na, const u64* __restrict__ db, u32 nbk, u32 nblocks, FoldOut fold)
{
    __shared__ u32 lds[WAVES * 2048];
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32* l = lds + wave * 2048u;
    const u32 total = gridDim.x * (u32)WAVES;
    u32 cnt = 0;
    u32 c = uniform32(blockIdx.x * (u32)WAVES + wave);
    // descriptors of the wave's NEXT column: requested with plain (per-lane, same address) loads one iteration ahead and
    // made wave-uniform only when they are needed -- a readfirstlane right after the load would wait for it on the spot
    auto raw = [&](const u64* __restrict__ d, u32 n, u32 col) -> u64 { return col < n ? d[col] : 0ull; };
    u64 ar = raw(da, na, c), br = raw(db, nbk, c);
    for (; c < nblocks; c += total) {
        const u64 a = uniform64(ar), b = uniform64(br);
        ar = raw(da, na, c + total); br = raw(db, nbk, c + total);
        const u32 ka = DESC_K(a), kb = DESC_K(b);
        const bool skip = (ka == K_NULL && kb == K_NULL) || (op == BMX_AND && (ka == K_NULL || kb == K_NULL)) ||
                          (op == BMX_SUB && (ka == K_NULL || kb == K_FULL));
        if (!skip) {
            Blk x, y;
            op2_issue<NT>(a, da, x, lane);
            op2_issue<NT>(b, da, y, lane);
            __builtin_amdgcn_sched_barrier(0);                    // (every load of the column leaves before anything is decoded)
            op2_finish(a, x, l, lane);
            op2_finish(b, y, l, lane);
            blk_op(op, x, y);
            cnt += blk_lane_popcount(x);
        }
    }
    cnt = wave_sum(cnt);
    count_fanin_fold(cnt, fold, lane, wave);
}

// bit_and / bit_or / bit_xor / bit_sub over operands of ANY block kinds, persistent form (round 4): the materialising twin of
// k_count_op2_loop.  k_op2 pays, per column, a wave launch, a descriptor round trip, the block loads and -- for a GAP operand
// -- one more round trip per 512 runs before anything is combined.  Here a wave stays and walks every (grid waves)-th column:
// next descriptors one column ahead, ALL loads of a column issued without control flow (op2_issue), GAP operands decoded from
// those registers (op2_finish), the result classified exactly as op2_block does and stored with non-temporal stores, the
// kinds of everything the wave produced folded once at the end (kinds.slots != null: no layout scan when no GAP block came out).
template <int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4)))
void k_op2_loop(int op, const u64* __restrict__ da, u32 na, const u64* __restrict__ db, u32 nbk, u32 nblocks, int opt_compress,
                uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, FoldOut kinds,
                u32* __restrict__ gap_offs, u64* __restrict__ gap_cursor, u16* __restrict__ gap_slab)
{
    // gap_offs != null: this kernel lays the GAP candidates out itself (bump allocation from *gap_cursor, offsets in gap_offs[];
    // the last workgroup of the fold hands the cursor to kinds.out[4] and leaves it at zero) and converts them in its tail, into
    // gap_slab (sized by the host at the operands' bound): no layout scan, no k_emit_gaps behind it
    __shared__ u32 lds[WAVES * 2048];
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32* l = lds + wave * 2048u;
    const u32 total = gridDim.x * (u32)WAVES;
    // the workgroup's GAP candidates, for its tail: {column, offset in the GAP slab, len << 1 | first} of the first PEND of them
    constexpr u32 PEND = 20u * (u32)WAVES;
    __shared__ u32 pend[PEND * 3u];
    __shared__ u32 npend_wg;
    if (threadIdx.x == 0) npend_wg = 0u;
    __syncthreads();
    u64 kc = 0ull;
    const u32 c0 = uniform32(blockIdx.x * (u32)WAVES + wave);
    u32 c = c0;
    auto raw = [&](const u64* __restrict__ d, u32 n, u32 col) -> u64 { return col < n ? d[col] : 0ull; };
    u64 ar = raw(da, na, c), br = raw(db, nbk, c);
    for (; c < nblocks; c += total) {
        const u64 a = uniform64(ar), b = uniform64(br);
        ar = raw(da, na, c + total); br = raw(db, nbk, c + total);
        const u32 ka = DESC_K(a), kb = DESC_K(b);
        u32 kind = op2_trivial(op, ka, kb);
        if (kind != 4u) store_trivial(kind, c, desc, st, lane);
        else {
            Blk x, y;
            op2_issue<NT>(a, da, x, lane);
            op2_issue<NT>(b, da, y, lane);
            __builtin_amdgcn_sched_barrier(0);                    // (every load of the column leaves before anything is decoded)
            op2_finish(a, x, l, lane);
            op2_finish(b, y, l, lane);
            blk_op(op, x, y);
            u32 gi[2];
            kind = store_result_mode<true>(x, c, op2_store_mode(op, ka, kb, opt_compress), slab, desc, st, lane, gap_offs, gap_cursor, nullptr, gi);
            if (gap_offs && uniform32(kind) == K_GAP && lane == 0) {
                const u32 at = atomicAdd(&npend_wg, 1u);
                if (at < PEND) { pend[at * 3u] = c; pend[at * 3u + 1u] = gi[0]; pend[at * 3u + 2u] = gi[1]; }
            }
        }
        kc += 1ull << (16u * kind);
    }
    if (kinds.slots) kind_fanin_fold_packed(kc, kinds, lane, wave, gap_offs ? gap_cursor : nullptr);
    // Tail (round 5): the workgroup converts the GAP candidates its waves parked -- as bits in their slots of the result slab,
    // which the loads of the same CU see (stores drained, then the workgroup barrier) -- into GAP blocks at the offsets they
    // drew from the cursor.  Here and not where the block is classified: two block images are alive there at the kernel's
    // 128-register cap, and every variant that converted in place spilled.  (This replaces a conversion kernel behind this
    // one: 10-18 us and a launch gap behind a 60-us kernel.)  The candidates of the workgroup's list are dealt round the waves
    // (a 1,270-run block takes ~5 us to convert: what ends the kernel is the wave with the most of them), the next one's
    // block is requested before the current one is converted; a workgroup with more than PEND of them finds them again
    // through st[] / gap_offs[].
    if (gap_offs) {                                               // (kernel argument: the whole workgroup takes the branch)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const u32 npend = uniform32(npend_wg);
        auto convert = [&](const Blk& b, u32 col, u32 off, u32 lf) {
            Blk t;
            (void)blk_transitions(b, t, lane);
            u16* g = gap_slab + off;
            gap_emit_from_transitions(t, lf >> 1, lf & 1u, g, lane, reinterpret_cast<u16*>(l));
            if (lane == 0) desc[col] = DESC_MAKE_GAP(g, lf >> 1, lf & 1u);
        };
        if (npend <= PEND) {
            if (wave < npend) {
                Blk b0, b1;
                blk_load(b0, as_gc4(slab + (size_t)uniform32(pend[wave * 3u]) * 512u), lane);
                for (u32 i = wave; i < npend; i += (u32)WAVES) {
                    const u32 nx = i + (u32)WAVES < npend ? i + (u32)WAVES : i;
                    blk_load(b1, as_gc4(slab + (size_t)uniform32(pend[nx * 3u]) * 512u), lane);
                    convert(b0, uniform32(pend[i * 3u]), uniform32(pend[i * 3u + 1u]), uniform32(pend[i * 3u + 2u]));
                    b0 = b1;
                }
            }
        } else {
            for (c = c0; c < nblocks; c += total) {
                if (uniform32(st[c].kind) != K_GAP) continue;
                Blk b;
                blk_load(b, as_gc4(slab + (size_t)c * 512u), lane);
                convert(b, c, uniform32(gap_offs[c]), (uniform32(st[c].runs) << 1) | uniform32(st[c].first));
            }
        }
    }
}

// bm::count_* when BOTH operands consist of bit-blocks only (the 10 % / 50 % cases of BASELINE configs[1]): the launch is
// one machine-load of waves (one workgroup per CU) and a wave streams a CONTIGUOUS stretch of block columns with two
// columns in flight -- the loads of column c+1 are issued before column c is counted, descriptor pairs are fetched two
// columns ahead with scalar loads -- instead of one short-lived wave per column (descriptor -> data -> count -> exit,
// three workgroup rounds per CU).  Same result: a sum of per-column popcounts.
template <int WAVES, bool NT, int NOPS = 2>
__global__ __launch_bounds__(WAVES * 64)
void k_count_op2_stream(int op, const u64* __restrict__ da, const u64* __restrict__ db, u32 nblocks, u32 per_wave, FoldOut fold)
{
    // NOPS = 1: bvector::count() of an all-bit-block vector (db unused): the same stream with one operand
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 w = uniform32(blockIdx.x * (u32)WAVES + wave);
    u32 c0 = w * per_wave;
    u32 c1 = c0 + per_wave < nblocks ? c0 + per_wave : nblocks;
    u32 cnt = 0;
    if (c0 < c1) {
        const u32 last = nblocks - 1u;
        auto ptr = [&](const u64* __restrict__ d, u32 c) { return DESC_P(uniform64(d[c < last ? c : last])); };   // (clamped: prefetch past the end is harmless)
        auto load = [&](Blk& x, Blk& y, u64 pa, u64 pb) {
            part_load<8, NT>(x, as_gc4(pa), lane);
            if (NOPS == 2) part_load<8, NT>(y, as_gc4(pb), lane);
        };
        auto eat = [&](Blk& x, const Blk& y) { if (NOPS == 2) blk_op(op, x, y); cnt += blk_lane_popcount(x); };
        Blk x0, y0, x1, y1;
        u32 c = c0;
        load(x0, y0, ptr(da, c), NOPS == 2 ? ptr(db, c) : 0ull);
        u64 a1 = ptr(da, c + 1u), b1 = NOPS == 2 ? ptr(db, c + 1u) : 0ull, a2 = ptr(da, c + 2u), b2 = NOPS == 2 ? ptr(db, c + 2u) : 0ull;
        for (; c + 2u < c1; c += 2u) {                       // columns c (in buffer 0), c+1, c+2 exist
            load(x1, y1, a1, b1);
            u64 a3 = ptr(da, c + 3u), b3 = NOPS == 2 ? ptr(db, c + 3u) : 0ull;
            eat(x0, y0);
            load(x0, y0, a2, b2);
            u64 a4 = ptr(da, c + 4u), b4 = NOPS == 2 ? ptr(db, c + 4u) : 0ull;
            eat(x1, y1);
            a1 = a3; b1 = b3; a2 = a4; b2 = b4;
        }
        if (c + 1u < c1) { load(x1, y1, a1, b1); eat(x0, y0); eat(x1, y1); }
        else eat(x0, y0);
        cnt = wave_sum(cnt);
    }
    count_fanin_fold(cnt, fold, lane, wave);
}

// bit_and/or/xor/sub(bv1, bv2) under the same conditions (bit-blocks only on both sides, opt_none): the materialising twin of
// the stream above.  A computed B x B block is stored as a bit-block unless the operation itself tests it (all-zero -> NULL
// for AND / XOR / SUB, all-ones -> FULL for OR: ST_TEST_ZERO / ST_TEST_ONE of op2_block), so the classification is two wave
// votes -- no popcount, no run count (st[].pop / runs of a bit-block are read by nothing) -- and the kinds of the wave's
// whole stretch are folded once at the end.  Results are written with non-temporal stores.
template <int WAVES, bool LNT, bool SNT>
__global__ __launch_bounds__(WAVES * 64)
void k_op2_stream(int op, const u64* __restrict__ da, const u64* __restrict__ db, u32 nblocks, u32 per_wave,
                  uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st, FoldOut kinds)
{
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 w = uniform32(blockIdx.x * (u32)WAVES + wave);
    u32 c0 = w * per_wave;
    u32 c1 = c0 + per_wave < nblocks ? c0 + per_wave : nblocks;
    u64 kc = 0ull;
    if (c0 < c1) {
        const u32 last = nblocks - 1u;
        auto ptr = [&](const u64* __restrict__ d, u32 c) { return DESC_P(uniform64(d[c < last ? c : last])); };
        auto load = [&](Blk& x, Blk& y, u64 pa, u64 pb) { part_load<8, LNT>(x, as_gc4(pa), lane); part_load<8, LNT>(y, as_gc4(pb), lane); };
        auto eat = [&](Blk& x, const Blk& y, u32 c) {
            blk_op(op, x, y);
            u32 o = 0u, a = ~0u;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o |= x.r[i].x | x.r[i].y | x.r[i].z | x.r[i].w; a &= x.r[i].x & x.r[i].y & x.r[i].z & x.r[i].w; }
            const bool zero = __ballot(o != 0u) == 0ull, ones = __ballot(a != ~0u) == 0ull;
            u32 kind = K_BIT;
            if (op != BMX_OR && zero) kind = K_NULL;
            if (op == BMX_OR && ones) kind = K_FULL;
            uint4* slot = slab + (size_t)c * 512u;
            if (kind == K_BIT) {
                gptr4 p = as_g4(slot);
#pragma unroll
                for (int i = 0; i < 8; ++i) { if (SNT) __builtin_nontemporal_store(x.r[i], &p[i * 64 + lane]); else p[i * 64 + lane] = x.r[i]; }
            }
            if (lane == 0) {
                st[c] = BlockStat{0u, kind == K_BIT ? 2u : 1u, ones ? 1u : 0u, kind};
                desc[c] = (kind == K_BIT) ? DESC_MAKE(slot, K_BIT) : DESC_MAKE(0, kind);
            }
            kc += 1ull << (16u * kind);
        };
        Blk x0, y0, x1, y1;
        u32 c = c0;
        load(x0, y0, ptr(da, c), ptr(db, c));
        u64 a1 = ptr(da, c + 1u), b1 = ptr(db, c + 1u), a2 = ptr(da, c + 2u), b2 = ptr(db, c + 2u);
        for (; c + 2u < c1; c += 2u) {
            load(x1, y1, a1, b1);
            u64 a3 = ptr(da, c + 3u), b3 = ptr(db, c + 3u);
            eat(x0, y0, c);
            load(x0, y0, a2, b2);
            u64 a4 = ptr(da, c + 4u), b4 = ptr(db, c + 4u);
            eat(x1, y1, c + 1u);
            a1 = a3; b1 = b3; a2 = a4; b2 = b4;
        }
        if (c + 1u < c1) { load(x1, y1, a1, b1); eat(x0, y0, c); eat(x1, y1, c + 1u); }
        else eat(x0, y0, c);
    }
    kind_fanin_fold_packed(kc, kinds, lane, wave);
}

// The yardstick of k_op2_stream: the same launch shape (a wave per stretch of 8-KiB blocks, two blocks in flight, non-temporal
// 16-byte loads and stores) doing nothing but c = a & b -- no descriptors, no classification, no per-block records.  What this
// box gives a 2-read : 1-write stream of that shape (bmx_probe_stream_rw).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void k_probe_rw(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, u32 nblocks, u32 per_wave)
{
    const u32 lane = lane_id(), wave = threadIdx.x >> 6;
    const u32 w = uniform32(blockIdx.x * (u32)WAVES + wave);
    const u32 c0 = w * per_wave;
    const u32 c1 = c0 + per_wave < nblocks ? c0 + per_wave : nblocks;
    if (c0 >= c1) return;
    auto load = [&](Blk& x, Blk& y, u32 col) {
        const u32 cc = col < nblocks ? col : nblocks - 1u;
        part_load<8, true>(x, as_gc4(a + (size_t)cc * 512u), lane); part_load<8, true>(y, as_gc4(b + (size_t)cc * 512u), lane);
    };
    auto eat = [&](Blk& x, const Blk& y, u32 col) {
        blk_and(x, y);
        gptr4 p = as_g4(c + (size_t)col * 512u);
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(x.r[i], &p[i * 64 + lane]);
    };
    Blk x0, y0, x1, y1;
    u32 col = c0;
    load(x0, y0, col);
    for (; col + 2u < c1; col += 2u) {
        load(x1, y1, col + 1u);
        eat(x0, y0, col);
        load(x0, y0, col + 2u);
        eat(x1, y1, col + 1u);
    }
    if (col + 1u < c1) { load(x1, y1, col + 1u); eat(x0, y0, col); eat(x1, y1, col + 1u); }
    else eat(x0, y0, col);
}

// ---------------------------------------------------------------------------
// OR-group classification (aggregator::sort_input_blocks_or src/bmaggregator.h:2278):
// row = [hdr, flags, region(n)]; hdr = nbit | ngap<<16; any FULL => ROW_FULL;
// nothing => ROW_EMPTY.  One wave per column.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_or_sort(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 n, u32 ncols,
               u64* __restrict__ dmat)
{
    u32 lane = lane_id();
    u32 c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // one wave per column (sort_operands, bmx_kernels.h)
    if (c >= ncols) return;
    u64* row = dmat + (size_t)c * (n + 2u);
    u32 nbit = 0, ngap = 0;
    bool full = sort_operands(descs, nblk, 0u, n, c, K_FULL, row + 2, nbit, ngap, lane);
    if (lane == 0) {
        row[0] = (u64)nbit | ((u64)ngap << 16);
        row[1] = full ? ROW_FULL : ((nbit | ngap) ? 0ull : ROW_EMPTY);
    }
}

// aggregator::combine_or(i, j, ...)  src/bmaggregator.h:1626; bit-block chain
// process_bit_blocks_or :1924 (saturation to all-ones => FULL, :1951);
// GAP operands process_gap_blocks_or :1808.
template <int U>
__global__ __launch_bounds__(256)
void k_agg_or(const u64* __restrict__ dmat, u32 n, u32 ncols, int opt_compress, int xcd_swz,
              uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st)
{
    extern __shared__ u32 lds_dyn[];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 c = uniform32(bid * 4u + wave);
    if (c >= ncols) return;
    const u64* row = dmat + (size_t)c * (n + 2u);
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) { store_trivial(K_NULL, c, desc, st, lane); return; }
    if (flags & ROW_FULL) { store_trivial(K_FULL, c, desc, st, lane); return; }
    u32 nbit = (u32)(hdr & 0xFFFFu), ngap = (u32)((hdr >> 16) & 0xFFFFu);
    const u64* p = row + 2;
    Blk acc;
    blk_fill(acc, 0u);
    // saturated (:1951) -- tested by the reference after every OR step, i.e. never when a single bit-block is just copied (:1936)
    if (pipe_chain<U, true, 2>(acc, p, nbit, lane) && nbit >= 2u) { store_trivial(K_FULL, c, desc, st, lane); return; }
    if (ngap) {                                          // process_gap_blocks_or (:1808), run-parallel in LDS
        u32* lds = lds_dyn + wave * 2048u;
        blk_to_lds(acc, lds, lane);
        (void)gap_apply_list<GAP_OR>(p + n - 1u, ngap, lds, lane);
        blk_from_lds(acc, lds, lane);
    }
    // opt_copy_bit_block(.., opt_mode_, ..) (:1658): without opt_compress the block is copied as it is, empty or not
    store_result_mode(acc, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);
}

// ---------------------------------------------------------------------------
// combine_or over MANY GAP-only operands (BASELINE configs[4]: thousands of sparse
// vectors).  Profiling the column-per-wave kernel on 4096 x 4e9-bit vectors showed
// 94 % UTCL1 (TLB) misses: 64 lanes of a wave touched 64 different vectors = 64
// different pages per load.  Here a 1024-thread workgroup owns a TILE of 16
// consecutive block columns with 16 accumulators in LDS (128 KiB), and 16 adjacent
// lanes read the 16 consecutive GAP blocks of ONE operand (contiguous in its slab,
// descriptors contiguous too): 4 pages per wave load instead of 64, full cache
// lines.  Operands come straight from the vectors' descriptor tables (no sort pass).
// Requires: no operand holds a bit-block (checked on the host; FULL/NULL are fine).
// ---------------------------------------------------------------------------
#define OR_TILE 16u
// What bounds it (round 2, profiles/r02f + tools/gpu_runs/r02_or.sh): the memory side of THIS access pattern.  With the
// run application compiled out (VAR 9, tuning build) the kernel still takes 3.65 ms of the 4.05 ms -- 4.4 TB/s for 16 GB
// read as 4096 streams in 1-KiB pieces (16 columns x 64 B per operand visit), whatever the load instructions look like:
// lane-per-block (4 x 16 B per lane), one fully coalesced 1-KiB load per operand (a row-coalesced kernel was written and
// measured: same floor, and slower in total because 24 % of the blocks exceed 64 B), one or two operands in flight per
// lane.  Also measured without effect: launch windows (the kernel is per-CU bound: half the CUs = half the rate), a
// bank-conflict-free interleaved tile (-6 %), fewer VALU instructions per run (single-bit fast path, VAR 1: +2 %).
// A 1-KiB visit is 256 B per HBM channel per row activation; the 8-KiB visits of the headline kernel stream at 6.9 TB/s.
// The tile cannot grow: 16 accumulators are 128 KiB of the 160 KiB LDS.
template <int VAR, int NOPS>
__global__ __launch_bounds__(1024)
void k_agg_or_gap_tiled(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 n, u32 ncols,
                        int opt_compress, uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                        u32 tile_base)
{
    extern __shared__ u32 lds_dyn[];                 // OR_TILE x 2048 u32 accumulators + OR_TILE flags
    u32* full = lds_dyn + OR_TILE * 2048u;
    u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    u32 c0 = (tile_base + blockIdx.x) * OR_TILE;
    // zero the accumulators: 1024 threads x 128 B
    u32x4* l4 = reinterpret_cast<u32x4*>(lds_dyn);
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i * 1024 + tid] = (u32x4)(0u);
    if (tid < OR_TILE) full[tid] = 0u;
    __syncthreads();
    u32 t = lane & (OR_TILE - 1u), grp = lane / OR_TILE;          // 4 operands per wave
    u32 col = c0 + t;
    const u32 S = (blockDim.x >> 6) * (64u / OR_TILE);             // operands one wave-step of the workgroup covers
    u32* acc = lds_dyn + t * 2048u;
    u32 op = wave * (64u / OR_TILE) + grp;                        // this lane's operands: op + (j + NOPS * step) * S
    // Three dependent reads per operand (table pointer + length -> descriptor -> block head) are spread
    // over three loop iterations, each issued unconditionally (indices clamped, results masked), so that
    // no iteration waits for a load it issued itself: stage A runs 3 steps ahead, B 2, C 1.
    typedef const __attribute__((address_space(1))) u64* gcptr64;
    typedef const __attribute__((address_space(1))) u32* gcptr32_;
    gcptr64 g_descs = (gcptr64)(uintptr_t)descs;
    gcptr32_ g_nblk = (gcptr32_)(uintptr_t)nblk;
    const u32 nm1 = n - 1u;
    bool colok = col < ncols;
#define TILE_STAGE_A(OP, PA, NB) { u32 oc_ = (OP) < n ? (OP) : nm1; PA = g_descs[oc_]; NB = ((OP) < n && colok) ? g_nblk[oc_] : 0u; }
#define TILE_STAGE_B(PA, NB, D)  { u32 cc_ = col < (NB) ? col : 0u; u64 v_ = ((gcptr64)(uintptr_t)(PA))[cc_]; D = col < (NB) ? v_ : 0ull; }
    u64 pa1[NOPS], pa2[NOPS], pa3[NOPS]; u32 nb1[NOPS], nb2[NOPS], nb3[NOPS];
    u64 d0[NOPS], d1[NOPS], d2[NOPS];
    GapHead h0[NOPS], h1[NOPS];
    const u32 STEP = S * NOPS;
#pragma unroll
    for (int j = 0; j < NOPS; ++j) {   // prologue: fill the pipeline
        u64 pa0; u32 nb0;
        u32 o = op + (u32)j * S;
        TILE_STAGE_A(o, pa0, nb0);
        TILE_STAGE_A(o + STEP, pa1[j], nb1[j]);
        TILE_STAGE_A(o + 2u * STEP, pa2[j], nb2[j]);
        TILE_STAGE_B(pa0, nb0, d0[j]);
        TILE_STAGE_B(pa1[j], nb1[j], d1[j]);
        gap_head_fetch(h0[j], DESC_P(d0[j]), DESC_K(d0[j]) == K_GAP);
    }
    for (; op < n; op += STEP) {
#pragma unroll
        for (int j = 0; j < NOPS; ++j) {
            TILE_STAGE_A(op + (u32)j * S + 3u * STEP, pa3[j], nb3[j]);
            TILE_STAGE_B(pa2[j], nb2[j], d2[j]);
            gap_head_fetch(h1[j], DESC_P(d1[j]), DESC_K(d1[j]) == K_GAP);
        }
#pragma unroll
        for (int j = 0; j < NOPS; ++j) {
            if constexpr (VAR == 9) {                              // memory-pattern probe (tuning build): loads only
                u32x4 z = h0[j].c[0] ^ h0[j].c[1] ^ h0[j].c[2] ^ h0[j].c[3];
                if ((z.x ^ z.y ^ z.z ^ z.w) == 0x12345679u) acc[0] = 1u;
            }
            else if constexpr (VAR == 1) gap_or_lane_v2(h0[j], DESC_P(d0[j]), acc);
            else gap_or_lane_fast(h0[j], DESC_P(d0[j]), acc);
            if (DESC_K(d0[j]) == K_FULL) full[t] = 1u;
        }
#pragma unroll
        for (int j = 0; j < NOPS; ++j) {
            d0[j] = d1[j]; d1[j] = d2[j]; h0[j] = h1[j];
            pa2[j] = pa3[j]; nb2[j] = nb3[j];
        }
    }
#undef TILE_STAGE_A
#undef TILE_STAGE_B
    __syncthreads();
    // one wave per column of the tile: classify + store (opt_copy_bit_block rule)
    for (u32 tc = wave; tc < OR_TILE; tc += (blockDim.x >> 6)) {
        u32 c = c0 + tc;
        if (c >= ncols) break;
        if (full[tc]) { store_trivial(K_FULL, c, desc, st, lane); continue; }
        Blk b;
        blk_from_lds(b, lds_dyn + tc * 2048u, lane);
        store_result_mode(b, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);
    }
}

// aggregator::combine_and_sub(target, ...)  src/bmaggregator.h:1162 with result
// blocks (always stored with opt_compress, :1210-1211).  Shares the row format
// and evaluation order of k_pipe_counts (single group).
// WG / col_base: the launch plan of the headline kernel applies here too -- a bit-block-only aggregation over many
// columns is launched in windows of one machine-load of waves (one 640-thread workgroup per CU, bmx.hip
// agg_and_sub_launch): 256 x 1e9-bit combine_and materialised 5.2 -> 4.8 ms.
template <int U, int WG = 256>
__global__ __launch_bounds__(WG)
void k_agg_and_sub(const u64* __restrict__ dmat, const u32* __restrict__ and_n_p, const u32* __restrict__ sub_n_p,
                   u32 col_stride, u32 ncols, int opt_compress, int xcd_swz,
                   uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
                   u32 col_from, u32 col_to, u32 col_base)
{
    extern __shared__ u32 lds_dyn[];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    u32 c = uniform32(col_base + bid * (u32)(WG / 64) + wave);
    if (c >= ncols) return;
    if (c < col_from || c >= col_to) { store_trivial(K_NULL, c, desc, st, lane); return; }     // outside the range hint: not visited (:1339-1346)
    const u64* row = dmat + (size_t)c * col_stride;
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) { store_trivial(K_NULL, c, desc, st, lane); return; }
    if (flags & ROW_FULL) { store_trivial(K_FULL, c, desc, st, lane); return; }
    u32 nba = (u32)(hdr & 0xFFFFu), nga = (u32)((hdr >> 16) & 0xFFFFu);
    u32 nbs = (u32)((hdr >> 32) & 0xFFFFu), ngs = (u32)(hdr >> 48);
    u32 na = uniform32(and_n_p[0]), ns = uniform32(sub_n_p[0]);
    const u64* pa = row + 2;
    const u64* ps = pa + na;
    u32* lds = lds_dyn + wave * 2048u;
    Blk acc;
    blk_fill(acc, ~0u);
    bool zero = pipe_chain<U, true, 0>(acc, pa, nba, lane);
    if (!zero) zero = pipe_chain<U, true, 1>(acc, ps, nbs, lane);
    // (an instantiation WITHOUT this GAP branch for the bit-block-only launch was measured: 5.4 against 4.85 ms on the
    // 256 x 1e9-bit combine_and -- the compiler schedules the fold differently; the dead branch stays)
    if (!zero && (nga | ngs)) {
        blk_to_lds(acc, lds, lane);
        zero = nga && gap_apply_list<GAP_AND>(pa + na - 1u, nga, lds, lane);
        if (!zero) zero = ngs && gap_apply_list<GAP_SUB>(ps + ns - 1u, ngs, lds, lane);
        if (!zero) { blk_from_lds(acc, lds, lane); zero = blk_is_zero(acc); }
    }
    if (zero) { store_trivial(K_NULL, c, desc, st, lane); return; }
    store_result(acc, c, opt_compress, slab, desc, st, lane);
}

// ---------------------------------------------------------------------------
// Few block columns, long operand lists (BASELINE configs[0] scale: 256 vectors of 1 Mbit = 16 columns): with one
// wave per (column, group) the chip would run 16 waves, each folding 256 blocks one after the other.  Here a
// workgroup of SPLIT waves owns the item: wave w folds its contiguous share of the AND / SUB bit-block lists into
// its own register accumulator (x & ~s is associative, so the shares combine with AND), the partial results meet in
// LDS, wave 0 reduces them, applies the GAP operands and finishes like k_pipe_counts / k_agg_and_sub.
// mode 0: counts[g] += popcount (counts-only pipeline); mode 1: store the block (single group: combine_and_sub).
// ---------------------------------------------------------------------------
template <int U, int SPLIT>
__global__ __launch_bounds__(SPLIT * 64)
void k_pipe_split(const u64* __restrict__ dmat, const u32* __restrict__ row_off, const u32* __restrict__ and_n,
                  const u32* __restrict__ sub_n, u32 col_stride, u32 ngroups, u32 col_from, u32 col_to, int mode,
                  u64* __restrict__ counts, int opt_compress, uint4* __restrict__ slab, u64* __restrict__ desc,
                  BlockStat* __restrict__ st)
{
    extern __shared__ u32 lds_dyn[];                               // SPLIT x 2048 u32
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 item = blockIdx.x;
    u32 c = col_from + item / ngroups, g = item % ngroups;
    if (mode == 1 && c >= col_to) { if (wave == 0) store_trivial(K_NULL, c, desc, st, lane); return; }   // outside a range hint
    const u64* row = dmat + (size_t)c * col_stride + row_off[g];
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) { if (mode == 1 && wave == 0) store_trivial(K_NULL, c, desc, st, lane); return; }
    if (flags & ROW_FULL) {
        if (wave == 0) {
            if (mode == 1) store_trivial(K_FULL, c, desc, st, lane);
            else if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), 65536ull);
        }
        return;
    }
    u32 nba = (u32)(hdr & 0xFFFFu), nga = (u32)((hdr >> 16) & 0xFFFFu);
    u32 nbs = (u32)((hdr >> 32) & 0xFFFFu), ngs = (u32)(hdr >> 48);
    u32 na = uniform32(and_n[g]), ns = uniform32(sub_n[g]);
    const u64* pa = row + 2;
    const u64* ps = pa + na;
    Blk acc;
    blk_fill(acc, ~0u);
    {
        u32 a0 = (u32)(((u64)nba * wave) / SPLIT), a1 = (u32)(((u64)nba * (wave + 1u)) / SPLIT);
        u32 s0 = (u32)(((u64)nbs * wave) / SPLIT), s1 = (u32)(((u64)nbs * (wave + 1u)) / SPLIT);
        bool zero = pipe_chain<U, false, 0>(acc, pa + a0, a1 - a0, lane);
        if (!zero) zero = pipe_chain<U, false, 1>(acc, ps + s0, s1 - s0, lane);
        if (zero) blk_fill(acc, 0u);
    }
    u32* mine = lds_dyn + wave * 2048u;
    blk_to_lds(acc, mine, lane);
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (u32 k = 1; k < (u32)SPLIT; ++k) { Blk t; blk_from_lds(t, lds_dyn + k * 2048u, lane); blk_and(acc, t); }
    bool zero = blk_is_zero(acc);
    if (!zero && (nga | ngs)) {
        blk_to_lds(acc, mine, lane);
        zero = nga && gap_apply_list<GAP_AND>(pa + na - 1u, nga, mine, lane);
        if (!zero) zero = ngs && gap_apply_list<GAP_SUB>(ps + ns - 1u, ngs, mine, lane);
        if (!zero) { blk_from_lds(acc, mine, lane); zero = blk_is_zero(acc); }
    }
    if (mode == 1) {
        if (zero) store_trivial(K_NULL, c, desc, st, lane);
        else store_result(acc, c, opt_compress, slab, desc, st, lane);
    } else if (!zero) {
        u32 cnt = wave_sum(blk_lane_popcount(acc));
        if (lane == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)cnt);
    }
}

// ---------------------------------------------------------------------------
// LDS-staged counts pipeline for MANY arg-groups over FEW distinct vectors -- the
// sparse_vector_scanner call pattern (bit-sliced search: every query is an AND-SUB
// group over the same bit-plane vectors, src/bmsparsevec_algo.h:2400-2630; the
// reference batches such groups per block column for L2 reuse, src/bmaggregator.h:
// 1326-1351,2900-2923).  Here one 1024-thread workgroup owns a block column:
// 16 plane blocks at a time are expanded into LDS (16 x 8 KiB; NULL -> zeros,
// FULL -> ones, GAP decoded) and each of the 16 waves folds them into the register
// accumulator of ITS group, selected by a 16+16-bit AND/SUB mask per (group, chunk).
// Every plane block is fetched once per 16 groups instead of once per group, and the
// operand traffic moves from L2 (~34 TB/s) to LDS (~150 TB/s).
// ---------------------------------------------------------------------------
// SLOTS = plane blocks staged at a time = waves per workgroup = groups per pass.  16 slots (1024 threads,
// 128 KiB, one workgroup per CU) fetch every block once per 16 groups; 8 slots (512 threads, 64 KiB) let two
// workgroups share a CU so one stages while the other computes.  Masks are always stored 16 planes per word.
template <int SLOTS>
__global__ __launch_bounds__(SLOTS * 64)
void k_pipe_counts_staged(const u64* const* __restrict__ udesc, const u32* __restrict__ unblk, u32 nplanes,
                          const u32* __restrict__ gmask /* [ngroups][nchunks]: and | sub << 16 */,
                          const u32* __restrict__ gskip /* [ngroups]: 1 = empty AND group */,
                          u32 ngroups, u32 col_from, u32 ncols_run, int xcd_swz, u64* __restrict__ counts)
{
    extern __shared__ u32 lds_dyn[];                        // SLOTS x 2048 u32
    constexpr u32 HALVES = 16u / SLOTS;                     // sub-steps per 16-plane mask word
    constexpr u32 HMASK = (1u << SLOTS) - 1u;
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 bid = xcd_swz ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    if (bid >= ncols_run) return;
    u32 c = col_from + bid;
    u32 nchunks = (nplanes + 15u) / 16u;
    u32* slot = lds_dyn + wave * 2048u;
    for (u32 g0 = 0; g0 < ngroups; g0 += SLOTS) {
        u32 g = g0 + wave;
        bool live = g < ngroups && !gskip[g < ngroups ? g : 0];
        Blk acc;
        blk_fill(acc, ~0u);
        bool zero = !live;
        for (u32 ch = 0; ch < nchunks; ++ch) {
            u32 m = zero ? 0u : uniform32(gmask[(size_t)g * nchunks + ch]);
#pragma unroll
            for (u32 h = 0; h < HALVES; ++h) {
                if (ch * 16u + h * SLOTS >= nplanes) break;  // workgroup-uniform
                __syncthreads();                             // everybody is done with the previous slots
                u32 pl = ch * 16u + h * SLOTS + wave;        // this wave stages plane pl into its slot
                if (pl < nplanes) {
                    u64 d = c < unblk[pl] ? uniform64(udesc[pl][c]) : 0ull;
                    Blk b;
                    blk_from_desc(d, b, slot, lane);         // GAP: the slot doubles as decode scratch
                    blk_to_lds(b, slot, lane);
                }
                __syncthreads();
                if (!zero) {
                    u32 am = (m >> (h * SLOTS)) & HMASK, sm = (m >> (16u + h * SLOTS)) & HMASK;
                    while (am) {
                        u32 p = (u32)__builtin_ctz(am); am &= am - 1u;
                        Blk t; blk_from_lds(t, lds_dyn + p * 2048u, lane);
                        blk_and(acc, t);
                    }
                    while (sm) {
                        u32 p = (u32)__builtin_ctz(sm); sm &= sm - 1u;
                        Blk t; blk_from_lds(t, lds_dyn + p * 2048u, lane);
                        blk_andn(acc, t);
                    }
                    zero = blk_is_zero(acc);
                }
            }
        }
        if (live && !zero) {
            u32 cnt = wave_sum(blk_lane_popcount(acc));
            if (lane == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&counts[g]), (unsigned long long)cnt);
        }
    }
}

// rows [mask_from, mask_to] of a block (range_gap_blk_, src/bmaggregator.h:980-988)
__device__ __forceinline__ void blk_bit_range(Blk& acc, u32 mask_from, u32 mask_to, u32 lane)
{
    u32 r = mask_to + 1u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 ws[4] = {acc.r[i].x, acc.r[i].y, acc.r[i].z, acc.r[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 lo = ((u32)i * 256u + lane * 4u + (u32)j) << 5;
            u32 hi_m = r >= lo + 32u ? ~0u : (r <= lo ? 0u : ((1u << (r - lo)) - 1u));
            u32 lo_m = mask_from <= lo ? ~0u : (mask_from >= lo + 32u ? 0u : (~0u << (mask_from - lo)));
            ws[j] &= hi_m & lo_m;
        }
        acc.r[i].x = ws[0]; acc.r[i].y = ws[1]; acc.r[i].z = ws[2]; acc.r[i].w = ws[3];
    }
}

// bit_find_first (src/bmfunc.h:9499): smallest bit index of the block, 0xFFFFFFFF = none (same value in every lane)
__device__ __forceinline__ u32 blk_first_bit(const Blk& acc, u32 lane)
{
    u32 mine = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        u32 w[4] = {acc.r[i].x, acc.r[i].y, acc.r[i].z, acc.r[i].w};
#pragma unroll
        for (int j = 3; j >= 0; --j)
            if (w[j]) mine = (((u32)i * 256u + lane * 4u + (u32)j) << 5) + (u32)__builtin_ctz(w[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { u32 t = __shfl_xor(mine, o, 64); mine = t < mine ? t : mine; }
    return mine;
}

// aggregator::find_first_and_sub  src/bmaggregator.h:1458: index of the first set bit of
// AND(group 0) AND NOT OR(group 1) without materialising the result.  Columns are visited in
// ascending order by the dispatcher; a wave gives up as soon as an earlier column already has a hit
// (*best holds the smallest global bit index found so far, ~0 = none).
template <int U>
__global__ __launch_bounds__(256)
void k_find_first_and_sub(const u64* __restrict__ dmat, const u32* __restrict__ and_n_p, const u32* __restrict__ sub_n_p,
                          u32 col_stride, u32 col_from, u32 ncols, int has_mask, u32 mask_from, u32 mask_to,
                          u64* __restrict__ best)
{
    // [col_from, ncols): aggregator::set_range_hint (src/bmaggregator.h:974): the search visits the block columns of
    // the hint only (:1470-1512); has_mask: both ends of the hint lie in this one block, which is then AND-ed with
    // the bit range [mask_from, mask_to] (range_gap_blk_, :980-988, 2354-2358)
    extern __shared__ u32 lds_dyn[];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 c = uniform32(col_from + blockIdx.x * 4u + wave);
    if (c >= ncols) return;
    u64 cur = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (((u64)c << 16) > cur) return;
    const u64* row = dmat + (size_t)c * col_stride;
    u64 hdr = uniform64(row[0]), flags = uniform64(row[1]);
    if (flags & ROW_EMPTY) return;
    if (flags & ROW_FULL) { if (lane == 0) atomicMin(reinterpret_cast<unsigned long long*>(best), ((unsigned long long)c << 16) + (has_mask ? mask_from : 0u)); return; }
    u32 nba = (u32)(hdr & 0xFFFFu), nga = (u32)((hdr >> 16) & 0xFFFFu);
    u32 nbs = (u32)((hdr >> 32) & 0xFFFFu), ngs = (u32)(hdr >> 48);
    u32 na = uniform32(and_n_p[0]), ns = uniform32(sub_n_p[0]);
    const u64* pa = row + 2;
    const u64* ps = pa + na;
    u32* lds = lds_dyn + wave * 2048u;
    Blk acc;
    blk_fill(acc, ~0u);
    if (pipe_chain<U, false, 0>(acc, pa, nba, lane)) return;
    if (pipe_chain<U, false, 1>(acc, ps, nbs, lane)) return;
    if (nga | ngs) {
        blk_to_lds(acc, lds, lane);
        if (nga && gap_apply_list<GAP_AND>(pa + na - 1u, nga, lds, lane)) return;
        if (ngs && gap_apply_list<GAP_SUB>(ps + ns - 1u, ngs, lds, lane)) return;
        blk_from_lds(acc, lds, lane);
    }
    if (has_mask) blk_bit_range(acc, mask_from, mask_to, lane);
    u32 mine = blk_first_bit(acc, lane);
    if (lane == 0 && mine != 0xFFFFFFFFu)
        atomicMin(reinterpret_cast<unsigned long long*>(best), ((unsigned long long)c << 16) + mine);
}

// ---------------------------------------------------------------------------
// Aggregation over a SMALL collection in ONE launch (configs[0] scale: few block columns, many operands): no row
// table, no sort pass.  A workgroup owns a block column: its threads classify the operands straight from the vectors'
// descriptor tables (sort_input_blocks_and / _or rules, src/bmaggregator.h:2315,2278), bit-block and GAP pointers are
// collected in LDS lists, the SPLIT waves fold their shares of the bit-blocks, wave 0 reduces, applies the GAP lists
// and finishes.  Host side: one staged copy of the operand table, this kernel (and the layout scan for a stored result).
//   MODE 0  combine_and_sub  (:1163): NULL in the AND group or FULL in the SUB group empties the column, FULL AND
//           operands and NULL SUB operands are dropped; result stored with opt_compress (:1210)
//   MODE 1  find_first_and_sub (:1458): same fold over columns [col_from, ncols), first bit -> atomicMin(best)
//   MODE 2  combine_or (:1626): any FULL operand => FULL column; saturation to all-ones after >= 2 bit-blocks => FULL (:1951)
// LDS: SPLIT x 8 KiB partials + one pointer slot per operand.
// ---------------------------------------------------------------------------
#define DIRECT_MAX_OPS 1024u
enum { DIRECT_AND_SUB = 0, DIRECT_FIND_FIRST = 1, DIRECT_OR = 2 };
template <int SPLIT, int MODE>
__global__ __launch_bounds__(SPLIT * 64)
void k_direct(const u64* const* __restrict__ descs, const u32* __restrict__ nblk, u32 n_and, u32 n_sub, u32 col_from, u32 ncols,
              int opt_compress, uint4* __restrict__ slab, u64* __restrict__ desc, BlockStat* __restrict__ st,
              int has_mask, u32 mask_from, u32 mask_to, u64* __restrict__ best)
{
    extern __shared__ u32 lds_dyn[];                               // SPLIT x 2048 u32, then the lists
    u64* bitA = reinterpret_cast<u64*>(lds_dyn + SPLIT * 2048u);   // group-0 bit-blocks from the front, GAP blocks from the back
    u64* gapA = bitA + n_and - 1u;                                 // (gap_apply_list walks backwards); nbit + ngap <= n_and
    u64* bitS = bitA + n_and;
    u64* gapS = bitS + n_sub - 1u;
    __shared__ u32 cnt[8];                                         // nbitA, ngapA, nbitS, ngapS, decided
    u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    u32 c = col_from + blockIdx.x;
    if (tid < 8) cnt[tid] = 0u;
    if (MODE == DIRECT_FIND_FIRST) {
        // an earlier column already has a hit: the whole workgroup leaves.  ONE thread reads the best hit so far and the
        // workgroup branches on that single value after a barrier -- *best only moves down, so per-wave loads could see
        // different values and part of a workgroup would skip the barriers below
        if (tid == 0) cnt[5] = (((u64)c << 16) > __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1u : 0u;
        __syncthreads();
        if (cnt[5]) return;
    } else __syncthreads();
    for (u32 i = tid; i < n_and + n_sub; i += blockDim.x) {
        bool is_sub = i >= n_and;
        u32 nb = nblk[i];
        u64 d = c < nb ? descs[i][c] : 0ull;
        u32 k = DESC_K(d);
        if (MODE == DIRECT_OR) {
            if (k == K_FULL) cnt[4] = 1u;                                          // any FULL => FULL column (:2292)
            else if (k == K_BIT) { u32 p = atomicAdd(&cnt[0], 1u); bitA[p] = DESC_P(d); }
            else if (k == K_GAP) { u32 p = atomicAdd(&cnt[1], 1u); *(gapA - p) = DESC_P(d); }
        } else if (!is_sub) {
            if (k == K_NULL) cnt[4] = 1u;                                          // any NULL => empty column (:2327)
            else if (k == K_BIT) { u32 p = atomicAdd(&cnt[0], 1u); bitA[p] = DESC_P(d); }
            else if (k == K_GAP) { u32 p = atomicAdd(&cnt[1], 1u); *(gapA - p) = DESC_P(d); }
        } else {
            if (k == K_FULL) cnt[4] = 1u;                                          // FULL in the SUB group => empty (:1746)
            else if (k == K_BIT) { u32 p = atomicAdd(&cnt[2], 1u); bitS[p] = DESC_P(d); }
            else if (k == K_GAP) { u32 p = atomicAdd(&cnt[3], 1u); *(gapS - p) = DESC_P(d); }
        }
    }
    __syncthreads();
    const u32 nba = cnt[0], nga = cnt[1], nbs = cnt[2], ngs = cnt[3];
    const bool nothing = !(nba | nga | nbs | ngs);
    if (MODE == DIRECT_OR) {
        if (cnt[4]) { if (wave == 0) store_trivial(K_FULL, c, desc, st, lane); return; }
        if (nothing) { if (wave == 0) store_trivial(K_NULL, c, desc, st, lane); return; }
    } else {
        if (cnt[4]) { if (MODE == DIRECT_AND_SUB && wave == 0) store_trivial(K_NULL, c, desc, st, lane); return; }
        if (nothing) {                                             // all FULL, nothing subtracted (:1751)
            if (MODE == DIRECT_AND_SUB) { if (wave == 0) store_trivial(K_FULL, c, desc, st, lane); }
            else if (tid == 0) atomicMin(reinterpret_cast<unsigned long long*>(best), ((unsigned long long)c << 16) + (has_mask ? mask_from : 0u));
            return;
        }
    }
    Blk acc;
    blk_fill(acc, MODE == DIRECT_OR ? 0u : ~0u);
    {
        // blocks in flight per wave: a lone wave per column (short lists, any number of columns) streams 4 at a time
        constexpr u32 DEPTH = SPLIT == 1 ? 4u : 2u;
        u32 a0 = (u32)(((u64)nba * wave) / SPLIT), a1 = (u32)(((u64)nba * (wave + 1u)) / SPLIT);
        for (u32 k = a0; k < a1; k += DEPTH) {
            Blk x[DEPTH];
#pragma unroll
            for (u32 j = 0; j < DEPTH; ++j) blk_load(x[j], as_gc4(uniform64(bitA[k + j < a1 ? k + j : a1 - 1u])), lane);   // (a repeated operand changes nothing)
#pragma unroll
            for (u32 j = 0; j < DEPTH; ++j) { if (MODE == DIRECT_OR) blk_or(acc, x[j]); else blk_and(acc, x[j]); }
            if (MODE == DIRECT_OR ? blk_is_ones(acc) : blk_is_zero(acc)) break;   // saturated / digest went to zero (:1951, :2081)
        }
        if (MODE != DIRECT_OR) {
            u32 s0 = (u32)(((u64)nbs * wave) / SPLIT), s1 = (u32)(((u64)nbs * (wave + 1u)) / SPLIT);
            for (u32 k = s0; k < s1; k += DEPTH) {
                Blk x[DEPTH];
#pragma unroll
                for (u32 j = 0; j < DEPTH; ++j) blk_load(x[j], as_gc4(uniform64(bitS[k + j < s1 ? k + j : s1 - 1u])), lane);
#pragma unroll
                for (u32 j = 0; j < DEPTH; ++j) blk_andn(acc, x[j]);
                if (blk_is_zero(acc)) break;
            }
        }
    }
    u32* mine = lds_dyn + wave * 2048u;
    blk_to_lds(acc, mine, lane);
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (u32 k = 1; k < (u32)SPLIT; ++k) {
        Blk t; blk_from_lds(t, lds_dyn + k * 2048u, lane);
        if (MODE == DIRECT_OR) blk_or(acc, t); else blk_and(acc, t);
    }
    if (MODE == DIRECT_OR) {
        if (nba >= 2u && blk_is_ones(acc)) { store_trivial(K_FULL, c, desc, st, lane); return; }   // saturated (:1951)
        if (nga) {
            blk_to_lds(acc, mine, lane);
            (void)gap_apply_list<GAP_OR>(gapA, nga, mine, lane);
            blk_from_lds(acc, mine, lane);
        }
        store_result_mode(acc, c, opt_compress ? ST_OPT : ST_FORCE_BIT, slab, desc, st, lane);   // :1658
        return;
    }
    bool zero = blk_is_zero(acc);
    if (!zero && (nga | ngs)) {
        blk_to_lds(acc, mine, lane);
        zero = nga && gap_apply_list<GAP_AND>(gapA, nga, mine, lane);
        if (!zero) zero = ngs && gap_apply_list<GAP_SUB>(gapS, ngs, mine, lane);
        if (!zero) { blk_from_lds(acc, mine, lane); zero = blk_is_zero(acc); }
    }
    if (MODE == DIRECT_AND_SUB) {
        if (zero) store_trivial(K_NULL, c, desc, st, lane);
        else store_result(acc, c, opt_compress, slab, desc, st, lane);
    } else if (!zero) {
        if (has_mask) blk_bit_range(acc, mask_from, mask_to, lane);
        u32 first = blk_first_bit(acc, lane);
        if (lane == 0 && first != 0xFFFFFFFFu)
            atomicMin(reinterpret_cast<unsigned long long*>(best), ((unsigned long long)c << 16) + first);
    }
}

// ---------------------------------------------------------------------------
// Rank / select.
// Device index (MI355X-first, sized for HBM not for a CPU cache): per block
//   rcount[nb]  u64  ones in blocks [0..nb]              (rs_index::rcount, src/bmrs.h:361)
//   cum[nb][64] u16  ones before each 1024-bit wave      (one 128 B line per block)
// so rank(n) = rcount[nb-1] + cum[nb][w] + popcount of <= 128 B: three
// INDEPENDENT loads whose addresses all follow from n (no pointer chase).
// The reference-compatible bcount / sub_count words (src/bm.h:2646-2656) are
// produced alongside so a host rs_index can be filled (bmx_rs_export).
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 word_count_to(u32 w, u32 widx, u32 R)
{
    // ones of word `widx` that lie at block positions <= R
    u32 lo = widx * 32u;
    if (lo > R) return 0u;
    u32 n = R - lo;                      // last included bit index, may be >= 31
    u32 m = n >= 31u ? ~0u : ((2u << n) - 1u);
    return __popc(w & m);
}

__device__ __forceinline__ u32 blk_lane_count_to(const Blk& b, u32 R, u32 lane)
{
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 w0 = (u32)i * 256u + lane * 4u;
        c += word_count_to(b.r[i].x, w0, R) + word_count_to(b.r[i].y, w0 + 1u, R)
           + word_count_to(b.r[i].z, w0 + 2u, R) + word_count_to(b.r[i].w, w0 + 3u, R);
    }
    return c;
}

// ones of a GAP block at positions <= R (gap_bit_count_to src/bmfunc.h:3499), lane partial
__device__ __forceinline__ u32 gap_lane_count_to(gcptr16 g, u32 R, u32 lane)
{
    u32 hdr = g[0]; u32 len = hdr >> 3, s = hdr & 1u; u32 c = 0;
    for (u32 k = 1 + lane; k <= len; k += 64) {
        if ((s ^ ((k - 1u) & 1u)) != 0u) {
            u32 e = g[k];
            u32 start = (k == 1u) ? 0u : (u32)g[k - 1] + 1u;
            if (start <= R) c += (e < R ? e : R) - start + 1u;
        }
    }
    return c;
}
// gap_bfind (src/bmfunc.h:1844): smallest k with g[k] >= pos, lane partial = #{k: g[k] < pos}
__device__ __forceinline__ u32 gap_lane_below(gcptr16 g, u32 pos, u32 lane)
{
    u32 len = (u32)g[0] >> 3; u32 c = 0;
    for (u32 k = 1 + lane; k <= len; k += 64) c += ((u32)g[k] < pos);
    return c;
}

// bvector::build_rs_index  src/bm.h:2531, one wave per block
__global__ __launch_bounds__(256)
void k_rs_build(const u64* __restrict__ desc, u32 nblocks, u32* __restrict__ bcount, u64* __restrict__ sub,
                u16* __restrict__ cum, u16* __restrict__ gidx)
{
    __shared__ u32 lds[4 * 2048];
    u32 lane = lane_id(), wave = threadIdx.x >> 6;
    u32 nb = uniform32(blockIdx.x * 4u + wave);
    if (nb >= nblocks) return;
    u64 d = uniform64(desc[nb]);
    u32 k = DESC_K(d);
    u16* crow = cum + (size_t)nb * 64u;
    if (k == K_NULL) { crow[lane] = 0; if (lane == 0) { bcount[nb] = 0; sub[nb] = 0; } return; }
    if (k == K_FULL) {
        crow[lane] = (u16)(lane * 1024u);
        if (lane == 0) { bcount[nb] = 65536u; sub[nb] = 21825ull | (21824ull << 16) | (32737ull << 32) | (54561ull << 48); }
        return;
    }
    u32 c0, c1, total; u64 aux0, aux1;
    Blk b;
    if (k == K_BIT) {
        blk_load(b, as_gc4(DESC_P(d)), lane);
        aux0 = wave_sum(blk_lane_count_to(b, 21824u + 10912u, lane));
        aux1 = wave_sum(blk_lane_count_to(b, 43648u + 10912u, lane));
    } else {
        gcptr16 g = as_gc16(DESC_P(d));
        gap_decode(g, lds + wave * 2048u, b, lane, GMETA(d));
        u32 s = GMETA(d) & 1u;
        {   // gidx[w] = first run that reaches bit w*1024 (gap_bfind, src/bmfunc.h:1844), one wave of 1024 bits per lane
            u32 glen = GMETA(d) >> 1, from = lane << 10, lo = 1, hi = glen;
            while (lo < hi) { u32 mid = (lo + hi) >> 1; if ((u32)g[mid] < from) lo = mid + 1; else hi = mid; }
            gidx[(size_t)nb * 64u + lane] = (u16)lo;
        }
        u32 i0 = 1u + wave_sum(gap_lane_below(g, 21825u, lane));
        u32 i1 = 1u + wave_sum(gap_lane_below(g, 43649u, lane));
        aux0 = ((u64)i0 << 1) | (s ^ ((i0 - 1u) & 1u));
        aux1 = ((u64)i1 << 1) | (s ^ ((i1 - 1u) & 1u));
    }
    c0 = wave_sum(blk_lane_count_to(b, 21824u, lane));
    c1 = wave_sum(blk_lane_count_to(b, 43648u, lane));
    // per-1024-bit wave counts: 1024 bits = 32 words = 8 lanes x 4 words of one register row
    u32 mine = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 pc = __popc(b.r[i].x) + __popc(b.r[i].y) + __popc(b.r[i].z) + __popc(b.r[i].w);
        pc += __shfl_xor(pc, 1, 64); pc += __shfl_xor(pc, 2, 64); pc += __shfl_xor(pc, 4, 64);
        // lanes 8q..8q+7 now hold the count of digest wave (i*8 + q); hand it to lane i*8+q
        u32 v = __shfl(pc, (lane & 7u) * 8u, 64);
        if ((lane >> 3) == (u32)i) mine = v;
    }
    u32 incl = wave_scan_incl(mine, lane);
    crow[lane] = (u16)(incl - mine);
    total = __shfl(incl, 63, 64);
    if (lane == 0) {
        bcount[nb] = total;
        sub[nb] = (u64)(c0 | ((c1 - c0) << 16)) | ((aux0 & 0xFFFFull) << 32) | ((aux1 & 0xFFFFull) << 48);
    }
}

// inclusive running count over blocks (single workgroup, SCAN_PER blocks per thread per pass)
__global__ __launch_bounds__(1024)
void k_rs_scan(const u32* __restrict__ bcount, u32 nblocks, u64* __restrict__ rcount, u64* __restrict__ total)
{
    __shared__ u32 sm[16];
    u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    u64 carry = 0;
    for (u32 base = 0; base < nblocks; base += 1024u * SCAN_PER) {
        u32 nb0 = base + w * (64u * SCAN_PER) + lane;             // lane = consecutive block: coalesced loads / stores
        u32 v[SCAN_PER], inc[SCAN_PER];
#pragma unroll
        for (u32 i = 0; i < SCAN_PER; ++i) { u32 nb = nb0 + i * 64u; v[i] = nb < nblocks ? bcount[nb] : 0u; }
        // one pass covers <= 16,384 blocks x 65,536 bits = 2^30 < 2^32: u32 partials are exact
        u32 run = 0;
#pragma unroll
        for (u32 i = 0; i < SCAN_PER; ++i) {
            u32 iv = wave_scan_incl(v[i], lane);
            inc[i] = run + iv;
            run += __shfl(iv, 63, 64);
        }
        if (lane == 0) sm[w] = run;
        __syncthreads();
        u32 off = 0, tot = 0;
#pragma unroll
        for (u32 i = 0; i < 16; ++i) { u32 a = sm[i]; if (i < w) off += a; tot += a; }
        __syncthreads();
#pragma unroll
        for (u32 i = 0; i < SCAN_PER; ++i) { u32 nb = nb0 + i * 64u; if (nb < nblocks) rcount[nb] = carry + off + inc[i]; }
        carry += tot;
    }
    if (tid == 0) *total = carry;
}

// top level of the select search: every (1 << shift)-th inclusive running count, small enough for LDS
__global__ __launch_bounds__(256)
void k_rs_sample(const u64* __restrict__ rcount, u32 nblocks, u32 shift, u32 nsamples, u64* __restrict__ sample)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamples) return;
    u64 last = (((u64)i + 1ull) << shift) - 1ull;
    sample[i] = rcount[last < nblocks ? last : nblocks - 1u];
}

// 8 lanes cooperate on one query (8 queries per wave step): each lane owns 16 B
// of the 128 B line that holds the query's 1024-bit wave.
__device__ __forceinline__ u32 group_sum8(u32 v)
{
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

// ones of GAP block g at positions in [from..to] (to inclusive), summed over a group of 8 lanes
template <u32 LPQ = 8>
__device__ __forceinline__ u32 gap_group_count_range(gcptr16 g, u32 meta, u32 lo, u32 from, u32 to, u32 sub)
{
    // lo = index of the first run that reaches `from` (from the rs-index: gidx[nb][wave]); meta = GMETA of the descriptor.
    // Each lane takes FOUR consecutive runs per round (five run ends read together, unconditionally, index clamped):
    // the 8 lanes cover 32 runs, more than a 1024-bit wave of a sparse block holds, so one memory round trip
    // usually answers the query (a lane walking every 8th run needed one round trip per 8 runs).
    u32 len = meta >> 1, s = meta & 1u;
    u32 c = 0;
    for (u32 k0 = lo + sub * 4u; k0 <= len; k0 += 4u * LPQ) {
        u32 ev[5];
#pragma unroll
        for (u32 j = 0; j < 5; ++j) { u32 kk = k0 - 1u + j; ev[j] = (u32)g[kk <= len ? kk : len]; }
        if (((k0 == 1u) ? 0u : ev[0] + 1u) > to) break;
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            u32 k = k0 + j;
            u32 start = (k == 1u) ? 0u : ev[j] + 1u;
            bool one = k <= len && start <= to && (s ^ ((k - 1u) & 1u)) != 0u;
            u32 a = start > from ? start : from, b = ev[j + 1u] < to ? ev[j + 1u] : to;
            c += one ? b - a + 1u : 0u;
        }
    }
    return c;
}

// Random 128-byte-line gather over a scratch buffer: the transaction-rate ceiling rank / select are measured against
// (bmx_probe_random_lines).  Same access shape as k_rank's bit-line read: 8 lanes share one line, 16 B each; the line
// index comes from a counter hash, four independent lines per lane per round.
__global__ __launch_bounds__(256)
void k_probe_lines(const uint4* __restrict__ buf, u64 nlines_buf, u64 nq, u64 seed, u64* __restrict__ sink)
{
    u32 sub = threadIdx.x & 7u;
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    u64 stride = ((u64)gridDim.x * blockDim.x) >> 3;
    u32 acc = 0;
    for (; qi < nq; qi += 4ull * stride) {
        uint4 v[4];
#pragma unroll
        for (u32 j = 0; j < 4; ++j) {
            u64 z = (qi + j * stride) * 0x9E3779B97F4A7C15ull + seed;
            z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
            u64 line = (u64)(((unsigned __int128)z * nlines_buf) >> 64);
            v[j] = buf[line * 8u + sub];
        }
#pragma unroll
        for (u32 j = 0; j < 4; ++j) acc += (qi + j * stride < nq) ? (v[j].x ^ v[j].y ^ v[j].z ^ v[j].w) : 0u;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;                  // keeps the loads alive; the buffer never holds that pattern
}

// bvector::count_to / rank(n, rs)  src/bm.h:3120 -- ones in [0..n]
__global__ __launch_bounds__(256)
void k_rank(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, const u16* __restrict__ cum,
            const u16* __restrict__ gidx, u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ out)
{
    u32 sub = threadIdx.x & 7u;
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    u64 stride = ((u64)gridDim.x * blockDim.x) >> 3;
    u64 nq_round = (nq + 7ull) & ~7ull;
    for (; qi < nq_round; qi += stride) {          // whole groups stay converged
        bool live = qi < nq;
        u64 n = live ? q[qi] : 0ull;
        u64 nb64 = n >> 16; u32 nbit = (u32)(n & 0xFFFFu);
        u64 res = 0;
        u32 part = 0;
        if (nb64 >= nblocks) res = total;           // rs.get_total() rule, src/bm.h:3133
        else {
            u32 nb = (u32)nb64;
            u64 d = desc[nb];
            u32 kd = DESC_K(d);
            res = nb ? rcount[nb - 1] : 0ull;
            u32 w = nbit >> 10;                     // digest wave
            if (kd == K_FULL) res += (u64)nbit + 1u;
            else if (kd == K_BIT) {
                res += cum[(size_t)nb * 64u + w];
                u32x4 v = as_gc4(DESC_P(d))[w * 8u + sub];
                u32 base = w * 32u + sub * 4u;      // word index of v.x
                part = word_count_to(v.x, base, nbit) + word_count_to(v.y, base + 1u, nbit)
                     + word_count_to(v.z, base + 2u, nbit) + word_count_to(v.w, base + 3u, nbit);
            } else if (kd == K_GAP) {
                res += cum[(size_t)nb * 64u + w];
                part = gap_group_count_range(as_gc16(DESC_P(d)), GMETA(d), gidx[(size_t)nb * 64u + w], w << 10, nbit, sub);
            }
        }
        part = group_sum8(part);
        if (live && sub == 0) out[qi] = res + part;
    }
}

// ---- LPQ lanes per query (round 3) ----
// k_rank gives a query 8 lanes (16 B of its bit line each): 8 queries per wave step, and a step is a chain of dependent
// reads (query -> descriptor / counts -> bit line), so a CU's 32 waves keep only 256 queries in flight and the kernel is
// latency-bound at half the box's random-line rate (bmx_probe_random_lines).  With LPQ = 2 or 4 lanes per query a lane
// reads 64 / 32 B of the line with 4 / 2 loads and a wave carries 32 / 16 queries per step.  (Tried first and dropped:
// Q queries per group of 8 lanes walked through the stages together -- hipcc sinks part of the batched loads back into
// the per-query branches and the step time did not move.)
template <u32 LPQ>
__global__ __launch_bounds__(256)
void k_rank_l(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, const u16* __restrict__ cum,
              const u16* __restrict__ gidx, u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ out)
{
    constexpr u32 NV = 8u / LPQ;                   // 16-byte pieces of the 128-B line per lane
    const u32 sub = threadIdx.x & (LPQ - 1u);
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / LPQ;
    const u64 stride = ((u64)gridDim.x * blockDim.x) / LPQ;
    const u64 nq_round = (nq + (64u / LPQ) - 1ull) / (64u / LPQ) * (64u / LPQ);      // whole waves stay converged
    for (; qi < nq_round; qi += stride) {
        bool live = qi < nq;
        u64 n = live ? q[qi] : 0ull;
        u64 nb64 = n >> 16; u32 nbit = (u32)(n & 0xFFFFu);
        bool in = live && nb64 < nblocks;
        u32 nb = in ? (u32)nb64 : 0u;
        u32 w = nbit >> 10;
        // the three index reads are independent and unconditional
        u64 d = desc[nb];
        u64 prev = rcount[nb ? nb - 1u : 0u];
        u32 cw = cum[(size_t)nb * 64u + w];
        u32 kd = in ? DESC_K(d) : K_NULL;
        // bit-blocks: my share of the line of the query's 1024-bit wave; everything else reads its (valid) cumulative row
        gcptr4 p = kd == K_BIT ? as_gc4(DESC_P(d)) + w * 8u + sub * NV : as_gc4(cum + (size_t)nb * 64u) + sub * NV;
        u32x4 v[NV];
#pragma unroll
        for (u32 i = 0; i < NV; ++i) v[i] = p[i];
        u64 res = nb ? prev : 0ull;
        u32 part = 0;
        if (kd == K_FULL) res += (u64)nbit + 1u;
        else if (kd == K_BIT) {
            res += cw;
#pragma unroll
            for (u32 i = 0; i < NV; ++i) {
                u32 base = w * 32u + (sub * NV + i) * 4u;
                part += word_count_to(v[i].x, base, nbit) + word_count_to(v[i].y, base + 1u, nbit)
                      + word_count_to(v[i].z, base + 2u, nbit) + word_count_to(v[i].w, base + 3u, nbit);
            }
        } else if (kd == K_GAP) {
            res += cw;
            part = gap_group_count_range<LPQ>(as_gc16(DESC_P(d)), GMETA(d), gidx[(size_t)nb * 64u + w], w << 10, nbit, sub);
        }
        if (live && !in) res = total;               // rs.get_total() rule, src/bm.h:3133
#pragma unroll
        for (u32 o = 1; o < LPQ; o <<= 1) part += __shfl_xor(part, o, 64);
        if (live && sub == 0) out[qi] = res + part;
    }
}

// bvector::select(rank, pos, rs)  src/bm.h:5350: position of the rank-th (1-based) set bit
__global__ __launch_bounds__(256)
void k_select(const u64* __restrict__ desc, u32 nblocks, const u64* __restrict__ rcount, const u16* __restrict__ cum,
              const u16* __restrict__ gidx, const u64* __restrict__ sample, u32 nsamples, u32 shift,
              u64 total, const u64* __restrict__ q, u64 nq, u64* __restrict__ pos, u8* __restrict__ found)
{
    // top level of rs_index::find in LDS (<= 2048 samples = 16 KiB): 11 LDS probes replace as many dependent
    // global loads; the bottom level searches one group of (1 << shift) blocks in global memory
    __shared__ u64 s_sample[2048];
    for (u32 i = threadIdx.x; i < nsamples; i += blockDim.x) s_sample[i] = sample[i];
    __syncthreads();
    u32 lane = lane_id();
    u32 sub = lane & 7u;
    u64 qi = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    u64 stride = ((u64)gridDim.x * blockDim.x) >> 3;
    u64 nq_round = (nq + 7ull) & ~7ull;
    for (; qi < nq_round; qi += stride) {
        bool live = qi < nq;
        u64 r = live ? q[qi] : 0ull;
        bool ok = live && r != 0ull && r <= total && nblocks != 0u;
        u64 result = 0;
        u32 kd = K_NULL, w = 0, rr = 0, nb = 0, sr_lo = 0, sr_hi = 0; u64 d = 0;
        if (ok) {
            // rs_index::find (src/bmrs.h:492): first block whose running count reaches r
            u32 glo = 0, ghi = nsamples - 1u;
            while (glo < ghi) { u32 mid = glo + ((ghi - glo) >> 1); if (s_sample[mid] < r) glo = mid + 1u; else ghi = mid; }
            u32 lo = glo << shift, hi = ((glo + 1u) << shift) - 1u;
            if (hi > nblocks - 1u) hi = nblocks - 1u;
            while (hi - lo >= 32u) { u32 mid = lo + ((hi - lo) >> 1); if (rcount[mid] < r) lo = mid + 1u; else hi = mid; }
            sr_lo = lo; sr_hi = hi;
        }
        // The last <= 32 running counts are read by the 8 lanes of the query's group in ONE round trip (4 each)
        // and the block index is the number of entries below r; the 64-entry cumulative row of the block is
        // searched the same way (8 x 16 B = its 128-byte line).  Three dependent memory round trips per query
        // (running counts -> previous count + descriptor + row -> bit line) instead of ~13 for two scalar binary searches.
        {
            u32 below = 0;
            u64 rc[4];
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {                               // unconditional reads (index clamped), counted afterwards
                u32 idx = sr_lo + sub * 4u + j;
                rc[j] = rcount[idx <= sr_hi ? idx : sr_hi];
            }
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                u32 idx = sr_lo + sub * 4u + j;
                below += (ok && idx <= sr_hi && rc[j] < r) ? 1u : 0u;
            }
            below = group_sum8(below);
            // previous running count, descriptor and the 64-entry row of block nb: three independent reads, issued
            // together and unconditionally (nb = 0 for a dead query: every table has at least one entry)
            nb = ok ? sr_lo + below : 0u;
            u64 prev = rcount[nb ? nb - 1u : 0u];
            d = desc[nb];
            u32x4 cv = as_gc4(cum + (size_t)nb * 64u)[sub];
            kd = ok ? DESC_K(d) : K_NULL;
            if (ok) {
                rr = (u32)(r - (nb ? prev : 0ull));                   // 1..65536 inside the block
                if (kd == K_FULL) result = ((u64)nb << 16) + rr - 1u;
            }
            // digest wave: last w with cum[w] < rr  (cum[0] = 0 < rr always; the row is non-decreasing)
            bool need_row = ok && kd != K_FULL;
            u32 c16[8] = {cv.x & 0xFFFFu, cv.x >> 16, cv.y & 0xFFFFu, cv.y >> 16, cv.z & 0xFFFFu, cv.z >> 16, cv.w & 0xFFFFu, cv.w >> 16};
            u32 nlt = 0, best = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { bool lt = c16[j] < rr; nlt += lt ? 1u : 0u; best = (lt && c16[j] > best) ? c16[j] : best; }
            nlt = group_sum8(nlt);
            { u32 t; t = __shfl_xor(best, 1, 64); best = t > best ? t : best; t = __shfl_xor(best, 2, 64); best = t > best ? t : best;
              t = __shfl_xor(best, 4, 64); best = t > best ? t : best; }
            if (need_row) { w = nlt - 1u; rr -= best; }             // 1..1024 inside the wave
        }
        // bit-blocks: 8 lanes x 16 B = the wave's 128 B line (block_find_rank src/bmfunc.h:9754)
        u32x4 v = (u32x4)(0u);
        if (ok && kd == K_BIT) v = as_gc4(DESC_P(d))[w * 8u + sub];
        u32 p0 = __popc(v.x), p1 = __popc(v.y), p2 = __popc(v.z), p3 = __popc(v.w);
        u32 mine = p0 + p1 + p2 + p3;
        // exclusive prefix inside the group of 8
        u32 incl = mine;
        { u32 t;
          t = __shfl_up(incl, 1, 64); if (sub >= 1u) incl += t;
          t = __shfl_up(incl, 2, 64); if (sub >= 2u) incl += t;
          t = __shfl_up(incl, 4, 64); if (sub >= 4u) incl += t; }
        u32 excl = incl - mine;
        bool hit = ok && kd == K_BIT && rr > excl && rr <= incl;
        if (hit) {
            u32 need = rr - excl;                               // 1..mine within my 4 words
            u32 word, wi;
            if (need <= p0) { word = v.x; wi = 0; }
            else if (need <= p0 + p1) { word = v.y; wi = 1; need -= p0; }
            else if (need <= p0 + p1 + p2) { word = v.z; wi = 2; need -= p0 + p1; }
            else { word = v.w; wi = 3; need -= p0 + p1 + p2; }
            for (u32 s = 1; s < need; ++s) word &= word - 1u;   // word_select (src/bmfunc.h:1084)
            u32 bit = (w * 32u + sub * 4u + wi) * 32u + (u32)__builtin_ctz(word);
            pos[qi] = ((u64)nb << 16) + bit;
        }
        {
            // gap_find_rank (src/bmfunc.h:3457) restricted to the digest wave: the 8 lanes of the query take 8
            // consecutive runs per round (run ends read unconditionally, index clamped), a prefix sum over the
            // group finds the run that holds the rr-th bit.  All lanes of the wave walk the loop together
            // (a group without a GAP query idles through it with gq = false).
            bool gq = ok && kd == K_GAP;
            gcptr16 g = gq ? as_gc16(DESC_P(d)) : (gcptr16)(uintptr_t)cum;       // idle lanes read a valid dummy
            u32 len = 0, s0 = 0, lo = 1, from = w << 10, need = rr;
            if (gq) { len = GMETA(d) >> 1; s0 = GMETA(d) & 1u; lo = gidx[(size_t)nb * 64u + w]; }
            bool searching = gq;
            for (u32 k0 = lo; __ballot(searching && k0 <= len) != 0ull; k0 += 32u) {
                // four consecutive runs per lane: five run ends read together (unconditionally, index clamped)
                u32 kf = k0 + sub * 4u;
                u32 ev[5];
#pragma unroll
                for (u32 j = 0; j < 5; ++j) { u32 kk = kf - 1u + j; ev[j] = (u32)g[(searching && kk <= len) ? kk : 1u]; }
                u32 cnt[4], st[4], mine = 0;
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {
                    u32 k = kf + j;
                    bool one = searching && k <= len && (s0 ^ ((k - 1u) & 1u)) != 0u;
                    u32 start = (k == 1u) ? 0u : ev[j] + 1u;
                    if (start < from) start = from;
                    st[j] = start;
                    cnt[j] = one ? ev[j + 1u] - start + 1u : 0u;
                    mine += cnt[j];
                }
                u32 incl = mine;
                { u32 t;
                  t = __shfl_up(incl, 1, 64); if (sub >= 1u) incl += t;
                  t = __shfl_up(incl, 2, 64); if (sub >= 2u) incl += t;
                  t = __shfl_up(incl, 4, 64); if (sub >= 4u) incl += t; }
                u32 total = __shfl(incl, (lane_id() & ~7u) + 7u, 64);
                u32 excl = incl - mine;
                if (searching && mine != 0u && need > excl && need <= incl) {
                    u32 rem = need - excl;                                  // 1..mine inside this lane's four runs
#pragma unroll
                    for (u32 j = 0; j < 4; ++j) {
                        if (rem != 0u && rem <= cnt[j]) { pos[qi] = ((u64)nb << 16) + st[j] + rem - 1u; rem = 0u; }
                        else if (rem != 0u) rem -= cnt[j];
                    }
                }
                if (searching && need <= total) searching = false;       // some lane of the group had the hit
                else need -= total;
            }
        }
        if (live && sub == 0) {
            found[qi] = ok ? 1 : 0;
            if (!ok) pos[qi] = 0;
            else if (kd != K_BIT && kd != K_GAP) pos[qi] = result;
        }
    }
}
