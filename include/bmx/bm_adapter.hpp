// bmx/bm_adapter.hpp -- bridge between a host bm::bvector<> (the real BitMagic
// container) and a device-resident bmx::bvector.  Include AFTER "bm.h".
//
// This is the code a BitMagic maintainer adds to route the vector-level loops of
// the hot path to the GPU: it walks blocks_manager exactly like
// bvector<>::count() does (src/bm.h:2436-2474), hands the
// flattened block table to bmx_vec_upload, and turns a downloaded result back
// into host blocks through the reference's own blocks_manager, so ownership and
// allocator rules of bm::bvector<> are untouched.
#pragma once

#include <cstring>
#include <vector>

#include "bvector.hpp"
#include "group.hpp"

namespace bmx {

struct block_table {
    uint64_t nbits = 0;
    std::vector<uint8_t> kinds;
    std::vector<uint32_t> offs;
    std::vector<uint32_t> bit_slab;     // n_bit * 2048 words
    std::vector<uint16_t> gap_slab;
};

/// flatten a bm::bvector<> (any allocator) into a block table of `nblocks` blocks
template <class BMBV>
void flatten(const BMBV& bv, uint32_t nblocks, block_table& t)
{
    const typename BMBV::blocks_manager_type& bman = bv.get_blocks_manager();
    t.nbits = bv.size();
    t.kinds.assign(nblocks, BMX_NULL); t.offs.assign(nblocks, 0);
    t.bit_slab.clear(); t.gap_slab.clear();
    if (!bman.is_init()) return;
    {   // size the staging slabs once (a growing std::vector would re-copy them ~log2(n) times)
        size_t nbit = 0, ngapw = 0;
        for (uint32_t nb = 0; nb < nblocks; ++nb) {
            unsigned i = nb >> bm::set_array_shift, j = nb & bm::set_array_mask;
            if (i >= bman.top_block_size()) break;
            const bm::word_t* p = bman.get_block_ptr(i, j);
            if (!p || p == FULL_BLOCK_FAKE_ADDR || p == FULL_BLOCK_REAL_ADDR) continue;
            if (BM_IS_GAP(p)) ngapw += (size_t)(BMGAP_PTR(p)[0] >> 3) + 1u; else ++nbit;
        }
        t.bit_slab.reserve(nbit * bm::set_block_size); t.gap_slab.reserve(ngapw);
    }
    uint32_t n_bit = 0;
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        unsigned i = nb >> bm::set_array_shift, j = nb & bm::set_array_mask;
        if (i >= bman.top_block_size()) break;
        const bm::word_t* p = bman.get_block_ptr(i, j);
        if (!p) continue;
        if (p == FULL_BLOCK_FAKE_ADDR || p == FULL_BLOCK_REAL_ADDR) { t.kinds[nb] = BMX_FULL; continue; }
        if (BM_IS_GAP(p)) {
            const bm::gap_word_t* g = BMGAP_PTR(p);
            unsigned n = (unsigned)(g[0] >> 3) + 1u;
            t.kinds[nb] = BMX_GAP; t.offs[nb] = (uint32_t)t.gap_slab.size();
            t.gap_slab.insert(t.gap_slab.end(), g, g + n);
        } else {
            t.kinds[nb] = BMX_BIT; t.offs[nb] = n_bit++;
            t.bit_slab.insert(t.bit_slab.end(), p, p + bm::set_block_size);
        }
    }
}

/// number of blocks needed to cover every stored block of bv
template <class BMBV>
uint32_t effective_blocks(const BMBV& bv)
{
    typename BMBV::size_type last = 0;
    if (!bv.find_reverse(last)) return 0;
    return (uint32_t)(last >> bm::set_block_shift) + 1u;
}

/// View of a vector whose blocks already lie back to back in memory -- what freeze() / optimize_freeze()
/// produce (blocks_manager::alloc_arena + copy_to_arena, src/bmblocks.h:2614-2655,2692-2770: every
/// bit-block in block order, then the top/sub pointer arrays, then every GAP block in block order).
/// Fills kinds / offs only and points at the arena memory; false when the blocks are scattered.
template <class BMBV>
bool flatten_view(const BMBV& bv, uint32_t nblocks, block_table& t,
                  const uint32_t*& bit_base, uint32_t& n_bit, const uint16_t*& gap_base, uint64_t& gap_words)
{
    const typename BMBV::blocks_manager_type& bman = bv.get_blocks_manager();
    t.nbits = bv.size();
    t.kinds.assign(nblocks, BMX_NULL); t.offs.assign(nblocks, 0);
    t.bit_slab.clear(); t.gap_slab.clear();
    bit_base = nullptr; gap_base = nullptr; n_bit = 0; gap_words = 0;
    if (!bman.is_init()) return true;
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        unsigned i = nb >> bm::set_array_shift, j = nb & bm::set_array_mask;
        if (i >= bman.top_block_size()) break;
        const bm::word_t* p = bman.get_block_ptr(i, j);
        if (!p) continue;
        if (p == FULL_BLOCK_FAKE_ADDR || p == FULL_BLOCK_REAL_ADDR) { t.kinds[nb] = BMX_FULL; continue; }
        if (BM_IS_GAP(p)) {
            const bm::gap_word_t* g = BMGAP_PTR(p);
            if (!gap_base) gap_base = g;
            if (g != gap_base + gap_words) return false;
            t.kinds[nb] = BMX_GAP; t.offs[nb] = (uint32_t)gap_words;
            gap_words += (uint64_t)(g[0] >> 3) + 1u;
        } else {
            if (!bit_base) bit_base = p;
            if (p != bit_base + (size_t)n_bit * bm::set_block_size) return false;
            t.kinds[nb] = BMX_BIT; t.offs[nb] = n_bit++;
        }
    }
    return true;
}

/// host bm::bvector<>  ->  device bmx::bvector.  A frozen (read-only, arena-backed) vector is uploaded
/// straight from its arena -- no host-side gather; returns true when that path was taken.
/// DST = bmx::bvector (one GPU) or bmx::gbvector (block-range shards over a device group).
template <class BMBV, class DST>
bool upload(const BMBV& src, DST& dst, uint32_t nblocks = 0)
{
    if (!nblocks) nblocks = effective_blocks(src);
    uint64_t nbits = (uint64_t)nblocks * BMX_BLOCK_BITS;
    block_table t;
    const uint32_t* bit_base; const uint16_t* gap_base; uint32_t n_bit; uint64_t gap_words;
    if (src.is_ro() && flatten_view(src, nblocks, t, bit_base, n_bit, gap_base, gap_words)) {
        dst.assign_block_table(nbits, nblocks, t.kinds.data(), t.offs.data(), bit_base, n_bit, gap_base, gap_words);
        return true;
    }
    flatten(src, nblocks, t);
    dst.assign_block_table(nbits, nblocks, t.kinds.data(), t.offs.data(), t.bit_slab.data(),
                           (uint32_t)(t.bit_slab.size() / BMX_BLOCK_WORDS), t.gap_slab.data(), t.gap_slab.size());
    return false;
}

/// slices of a bit-sliced container (bm::sparse_vector<>: get_slice(i), effective_slices(),
/// src/bmbmatrix.h:739,756) -> device vectors for bmx::slice_scanner (bmx/scanner.hpp).  `store` owns the
/// device vectors; slices[i] is nullptr where the host plane does not exist.  All slices are uploaded
/// with the same block count (the container's size) so that NULL tails behave as in the reference.
template <class SV>
void upload_slices(const SV& sv, context& ctx, std::vector<bvector>& store, std::vector<const bvector*>& slices,
                   const bvector** not_null = nullptr)
{
    unsigned planes = sv.effective_slices();
    uint32_t nblocks = (uint32_t)(((uint64_t)sv.size() + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS);
    store.clear(); store.reserve(planes + 1u);
    std::vector<int> slot(planes, -1);
    for (unsigned i = 0; i < planes; ++i) {
        if (const typename SV::bvector_type* bv = sv.get_slice(i)) {
            store.emplace_back(ctx);
            upload(*bv, store.back(), nblocks ? nblocks : 1);
            slot[i] = (int)store.size() - 1;
        }
    }
    int null_slot = -1;
    if (not_null) {                                  // sv.get_null_bvector(): the NOT-NULL flags of a nullable container
        *not_null = nullptr;
        if (const typename SV::bvector_type* bn = sv.get_null_bvector()) {
            store.emplace_back(ctx);
            upload(*bn, store.back(), nblocks ? nblocks : 1);
            null_slot = (int)store.size() - 1;
        }
    }
    slices.assign(planes, nullptr);
    for (unsigned i = 0; i < planes; ++i) if (slot[i] >= 0) slices[i] = &store[(size_t)slot[i]];
    if (null_slot >= 0) *not_null = &store[(size_t)null_slot];
}

/// install a block table into a host bm::bvector<> through the reference's own
/// blocks_manager (FULL sentinel / clone_gap_block src/bmblocks.h:865 / copy_bit_block :1340)
template <class BMBV>
void install(BMBV& dst, uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
             const uint32_t* bit_slab, const uint16_t* gap_slab)
{
    dst.clear(true);
    dst.init();
    typename BMBV::blocks_manager_type& bman = dst.get_blocks_manager();
    BM_DECLARE_TEMP_BLOCK(tb)          // SIMD builds stream-copy from an aligned source (src/bmfunc.h:7573)
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        unsigned i = nb >> bm::set_array_shift, j = nb & bm::set_array_mask;
        if (kinds[nb] == BMX_NULL) continue;
        bman.reserve_top_blocks(i + 1);
        bman.check_alloc_top_subblock(i);
        switch (kinds[nb]) {
        case BMX_FULL:
            bman.set_block_ptr(i, j, FULL_BLOCK_FAKE_ADDR);
            break;
        case BMX_BIT:
            std::memcpy(tb.begin(), bit_slab + (size_t)offs[nb] * BMX_BLOCK_WORDS, BMX_BLOCK_WORDS * 4);
            bman.copy_bit_block(i, j, tb.begin());
            break;
        default: {
            const bm::gap_word_t* g = gap_slab + offs[nb];
            bman.clone_gap_block(i, j, g, (unsigned)(g[0] >> 3));
            break; }
        }
    }
}

/// device bmx::bvector  ->  host bm::bvector<>
template <class BMBV>
void download(const bvector& src, BMBV& dst)
{
    if (src.empty_handle()) { dst.clear(true); return; }
    uint64_t nbits = 0, gap_words = 0; uint32_t nblocks = 0, slab_blocks = 0; uint32_t counts[4];
    check(bmx_vec_info(src.handle(), &nbits, &nblocks, counts, &slab_blocks, &gap_words));
    std::vector<uint8_t> kinds(nblocks ? nblocks : 1); std::vector<uint32_t> offs(nblocks ? nblocks : 1);
    std::vector<uint32_t> bits((size_t)slab_blocks * BMX_BLOCK_WORDS); std::vector<uint16_t> gaps(gap_words);
    check(bmx_vec_download(src.get_context().handle(), src.handle(), kinds.data(), offs.data(),
                           bits.empty() ? nullptr : bits.data(), gaps.empty() ? nullptr : gaps.data()));
    install(dst, nblocks, kinds.data(), offs.data(), bits.data(), gaps.data());
}

/// sharded device vector (device group)  ->  host bm::bvector<>: the shards are gathered into one block table
template <class BMBV>
void download(const gbvector& src, BMBV& dst)
{
    if (src.empty_handle()) { dst.clear(true); return; }
    uint64_t nbits = 0, gap_words = 0; uint32_t nblocks = 0, slab_blocks = 0; uint32_t counts[4];
    check(bmx_gvec_info(src.handle(), &nbits, &nblocks, counts, &slab_blocks, &gap_words));
    std::vector<uint8_t> kinds(nblocks ? nblocks : 1); std::vector<uint32_t> offs(nblocks ? nblocks : 1);
    std::vector<uint32_t> bits((size_t)slab_blocks * BMX_BLOCK_WORDS); std::vector<uint16_t> gaps(gap_words);
    check(bmx_gvec_download(src.get_group().handle(), src.handle(), kinds.data(), offs.data(),
                            bits.empty() ? nullptr : bits.data(), gaps.empty() ? nullptr : gaps.data()));
    install(dst, nblocks, kinds.data(), offs.data(), bits.data(), gaps.data());
}

} // namespace bmx
