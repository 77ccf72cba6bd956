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

#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>
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
/// (CTX / DV = context + bvector, or device_group + gbvector: the planes are then sharded by block range)
template <class SV, class CTX, class DV>
void upload_slices(const SV& sv, CTX& ctx, std::vector<DV>& store, std::vector<const DV*>& slices,
                   const DV** not_null = nullptr)
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

/// bm::bvector<>::build_rs_index(&rs) with the block popcounts computed on the GPU: `dev` is the device copy of `bv`
/// (bmx::upload), the index words come back through bmx_rs_export in the reference's own layout and are installed with
/// the calls build_rs_index itself uses (src/bm.h:2531-2660: init / set_total / resize / resize_effective_super_blocks /
/// set_null_super_block / set_full_super_block / register_super_block).  The HOST index then answers single
/// count_to / rank / select calls at CPU latency (~50 ns; a device launch per single query costs ~20 us), while batches
/// of queries go to bmx::bvector::count_to / select on the device copy.
template <class BMBV>
void build_rs_index(const BMBV& bv, const bvector& dev, typename BMBV::rs_index_type* rs_idx)
{
    typedef typename BMBV::size_type bm_size_type;
    rs_idx->init();
    const typename BMBV::blocks_manager_type& bman = bv.get_blocks_manager();
    if (!bman.is_init()) return;
    bm_size_type last_bit;
    if (!bv.find_reverse(last_bit)) return;
    uint64_t nb = (uint64_t)(last_bit >> bm::set_block_shift);
    const unsigned real_top_blocks = bman.find_real_top_blocks();
    const unsigned max_top_blocks = bman.find_max_top_blocks();
    if (nb < (uint64_t)max_top_blocks * bm::set_sub_array_size) nb = (uint64_t)max_top_blocks * bm::set_sub_array_size;
    rs_idx->set_total((bm_size_type)(nb + 1));
    rs_idx->resize((typename BMBV::block_idx_type)(nb + 1));
    rs_idx->resize_effective_super_blocks(real_top_blocks);
    rs_index drs;
    dev.build_rs_index(&drs);
    const uint32_t dev_blocks = dev.block_count();
    std::vector<uint32_t> bcount; std::vector<uint64_t> sub;
    drs.export_blocks(bcount, sub, dev_blocks);
    bm::word_t*** blk_root = bman.top_blocks_root();
    unsigned bc[bm::set_sub_array_size]; bm::id64_t sc[bm::set_sub_array_size];
    for (unsigned i = 0; i < max_top_blocks; ++i) {
        bm::word_t** blk_blk = blk_root[i];
        if (!blk_blk) { rs_idx->set_null_super_block(i); continue; }
        if ((bm::word_t*)blk_blk == FULL_BLOCK_FAKE_ADDR) { rs_idx->set_full_super_block(i); continue; }
        for (unsigned j = 0; j < bm::set_sub_array_size; ++j) {
            const uint64_t b = (uint64_t)i * bm::set_sub_array_size + j;
            bc[j] = b < dev_blocks ? bcount[(size_t)b] : 0u;
            sc[j] = b < dev_blocks ? (bm::id64_t)sub[(size_t)b] : 0ull;
        }
        rs_idx->register_super_block(i, &bc[0], &sc[0]);
    }
}

/// process-wide default context for code that constructs its aggregators without arguments (device: BMX_DEVICE, default 0)
inline context& default_context()
{
    static context ctx(std::getenv("BMX_DEVICE") ? std::atoi(std::getenv("BMX_DEVICE")) : 0);
    return ctx;
}

/// bm::aggregator<BV> DROP-IN over HOST vectors: same constructor, add(const BV*, group) / reset() / combine_or /
/// combine_and / combine_and_sub / find_first_and_sub / combine_shift_right_and / set_optimization / set_range_hint /
/// pipeline<Opt> as bm::aggregator<BV> (src/bmaggregator.h:120-854, 1013-1079) -- application code changes ONE
/// typedef:   bm::aggregator<bm::bvector<> > agg;   ->   bmx::device_aggregator<bm::bvector<> > agg;
/// (tests/cpp builds /root/reference/samples/bvsample16/sample16.cpp exactly like that and compares its output).
///
/// add() takes host `const BV*`; operands are uploaded on first use and results are installed into the host target
/// through the reference's own blocks_manager (bmx::download).  Upload cache, by vector address:
///   * a freeze()d vector (is_ro(): immutable by construction, its arena is uploaded without a host-side gather) stays
///     resident until invalidate() / the aggregator dies -- build the index once, query many times;
///   * a mutable vector is uploaded again at every combine_* call unless set_cache_mutable(true) promises that
///     invalidate(bv) is called after every change (the reference reads the live blocks at combine time, so silently
///     re-using an old upload would not be a drop-in).
/// A cached entry is also dropped when the vector's block-tree shape changed (size, top size, sub-array pointers).
template <class BV>
class device_aggregator {
public:
    typedef BV bvector_type;
    typedef typename BV::size_type size_type;
    typedef const bvector_type* bvector_type_const_ptr;

    struct arg_groups {                                   // aggregator::arg_groups (src/bmaggregator.h:2925)
        std::vector<bvector_type_const_ptr> arg_bv0, arg_bv1;
        void reset() { arg_bv0.clear(); arg_bv1.clear(); }
        size_t add(const BV* bv, unsigned agr_group)
        {
            if (agr_group > 1) throw error(BMX_ERR_RANGE, "BMX-03: Incorrect range or index [agr_group > 1]");   // BM_ERR_RANGE :2934
            if (!bv) return 0;                                                                                  // ignored :2939
            std::vector<bvector_type_const_ptr>& v = agr_group ? arg_bv1 : arg_bv0;
            v.push_back(bv);
            return v.size();
        }
    };

    /// aggregator::pipeline<Opt> (src/bmaggregator.h:222-341) over host vectors; run with combine_and_sub(pipe)
    template <class Opt = agg_run_options<> >
    class pipeline {
    public:
        typedef Opt options_type;
        pipeline() {}
        ~pipeline() { for (size_t i = 0; i < groups_.size(); ++i) delete groups_[i]; for (size_t i = 0; i < results_.size(); ++i) delete results_[i]; }
        pipeline(const pipeline&) = delete;
        pipeline& operator=(const pipeline&) = delete;
        arg_groups* add() { if (complete_) throw error(BMX_ERR_BADARG, "pipeline already complete()"); groups_.push_back(new arg_groups()); return groups_.back(); }
        void complete() { complete_ = true; counts_.assign(groups_.size(), 0); }
        bool is_complete() const noexcept { return complete_; }
        size_t size() const noexcept { return groups_.size(); }
        void set_or_target(BV* bv_or) noexcept { or_target_ = bv_or; }                      // :245
        void set_search_count_limit(size_type limit) noexcept { limit_ = limit; }          // :255 (forwarded to the device pipeline)
        std::vector<BV*>& get_bv_res_vector() noexcept { return results_; }                // nullptr where a group found nothing
        std::vector<size_type>& get_bv_count_vector() noexcept { return counts_; }
    private:
        friend class device_aggregator;
        std::vector<arg_groups*> groups_;
        std::vector<size_type> counts_;
        std::vector<BV*> results_;
        BV* or_target_ = nullptr;
        size_type limit_ = ~size_type(0);
        bool complete_ = false;
    };

    device_aggregator() : ctx_(&default_context()), agg_(*ctx_) {}
    explicit device_aggregator(context& ctx) : ctx_(&ctx), agg_(ctx) {}
    device_aggregator(const device_aggregator&) = delete;
    device_aggregator& operator=(const device_aggregator&) = delete;

    // ---- bm::aggregator surface ----
    size_t add(const BV* bv, unsigned agr_group = 0) { return ag_.add(bv, agr_group); }     // :1013
    void reset() { ag_.reset(); agg_.reset_range_hint(); }                                  // :941 (also clears the range hint, :944)
    void set_optimization(typename BV::optmode opt = BV::opt_compress) { agg_.set_optimization(opt == BV::opt_compress); }   // :359
    void set_compute_count(bool count_mode) { agg_.set_compute_count(count_mode); }         // :363
    size_type count() const { return (size_type)agg_.count(); }                            // :488
    bool set_range_hint(size_type from, size_type to) noexcept { return agg_.set_range_hint(from, to); }   // :481
    void reset_range_hint() noexcept { agg_.reset_range_hint(); }

    void combine_or(BV& bv_target) { combine_or(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size()); }                          // :1021
    void combine_and(BV& bv_target) { (void)combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), nullptr, 0, false); }   // :1030
    bool combine_and_sub(BV& bv_target)                                                                                       // :1044
    { return combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size(), false); }
    bool combine_and_sub(BV& bv_target, bool any)
    { return combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size(), any); }
    bool find_first_and_sub(size_type& idx)                                                                                   // :1079
    { return find_first_and_sub(idx, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size()); }
    void combine_shift_right_and(BV& bv_target)                                                                               // :473,1089
    { (void)combine_shift_right_and(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), false); }

    // C-style forms (:1101,1127,1162,1458,552)
    void combine_or(BV& bv_target, const bvector_type_const_ptr* bv_src, size_t src_size)
    {
        op_scope scope(this);
        std::vector<const bvector*> d = resident(bv_src, src_size);
        ag_.reset();                                     // the reference clears the member arg-groups here (:1110)
        bvector t(*ctx_);
        agg_.combine_or(t, d.data(), d.size());
        download(t, bv_target);
    }
    void combine_and(BV& bv_target, const bvector_type_const_ptr* bv_src, size_t src_size)
    {
        op_scope scope(this);
        std::vector<const bvector*> d = resident(bv_src, src_size);
        if (src_size > 1) ag_.reset();
        bvector t(*ctx_);
        (void)agg_.combine_and_sub(t, d.data(), d.size(), nullptr, 0, false);
        download(t, bv_target);
    }
    bool combine_and_sub(BV& bv_target, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                         const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size, bool any)
    {
        op_scope scope(this);
        std::vector<const bvector*> a = resident(bv_src_and, src_and_size), s = resident(bv_src_sub, src_sub_size);
        bvector t(*ctx_);
        bool found = agg_.combine_and_sub(t, a.data(), a.size(), s.data(), s.size(), any);
        download(t, bv_target);
        return found;
    }
    bool find_first_and_sub(size_type& idx, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                            const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size)
    {
        op_scope scope(this);
        std::vector<const bvector*> a = resident(bv_src_and, src_and_size), s = resident(bv_src_sub, src_sub_size);
        bmx::size_type p = 0;
        bool found = agg_.find_first_and_sub(p, a.data(), a.size(), s.data(), s.size());
        if (found) idx = (size_type)p;
        return found;
    }
    bool combine_shift_right_and(BV& bv_target, const bvector_type_const_ptr* bv_src_and, size_t src_and_size, bool any)
    {
        op_scope scope(this);
        std::vector<const bvector*> d = resident(bv_src_and, src_and_size);
        bvector t(*ctx_);
        bool found = agg_.combine_shift_right_and(t, d.data(), d.size(), any);
        if (!t.empty_handle()) download(t, bv_target);   // count mode leaves the target untouched (:2593)
        return found;
    }

    /// combine_and_sub(pipe)  :1292 -- counts land in pipe.get_bv_count_vector(), result vectors (host BV, owned by the
    /// pipeline) in get_bv_res_vector(), the OR target is updated in place
    template <class TPipe>
    void combine_and_sub(TPipe& pipe)
    {
        if (!pipe.is_complete()) throw error(BMX_ERR_BADARG, "pipeline is not complete()");
        if (!pipe.size()) return;
        op_scope scope(this);
        typedef typename TPipe::options_type opt;
        typename aggregator<bvector>::template pipeline<opt> dp(*ctx_);
        for (size_t g = 0; g < pipe.groups_.size(); ++g) {
            typename aggregator<bvector>::arg_groups* dg = dp.add();
            std::vector<const bvector*> a = resident(pipe.groups_[g]->arg_bv0.data(), pipe.groups_[g]->arg_bv0.size());
            std::vector<const bvector*> s = resident(pipe.groups_[g]->arg_bv1.data(), pipe.groups_[g]->arg_bv1.size());
            for (size_t i = 0; i < a.size(); ++i) dg->add(a[i], 0);
            for (size_t i = 0; i < s.size(); ++i) dg->add(s[i], 1);
        }
        bvector ort(*ctx_);
        if (pipe.or_target_) { upload(*pipe.or_target_, ort, common_blocks_); dp.set_or_target(&ort); }
        if (pipe.limit_ != ~size_type(0)) dp.set_search_count_limit((bvector::size_type)pipe.limit_);   // :255
        dp.complete();
        agg_.combine_and_sub(dp);
        if (opt::is_compute_counts())
            for (size_t g = 0; g < pipe.size(); ++g) pipe.counts_[g] = (size_type)dp.get_bv_count_vector()[g];
        if (opt::is_make_results()) {
            for (size_t i = 0; i < pipe.results_.size(); ++i) delete pipe.results_[i];
            pipe.results_.assign(pipe.size(), nullptr);
            for (size_t g = 0; g < pipe.size(); ++g)
                if (bvector* r = dp.get_bv_res_vector()[g]) { pipe.results_[g] = new BV(); download(*r, *pipe.results_[g]); }
        }
        if (pipe.or_target_) download(ort, *pipe.or_target_);
    }

    // ---- upload cache control ----
    // Device copies are keyed by the host vector's ADDRESS plus a stamp (size, top size, sub-array pointers, a sampled
    // content hash).  The reference aggregator reads live blocks; this drop-in does so for mutable vectors by uploading
    // them again on every operation.  Copies that are KEPT across operations -- frozen (read-only) vectors, and mutable
    // ones under set_cache_mutable(true) -- must be invalidate()d before the vector is destroyed or changed in place.
    void set_cache_mutable(bool on) noexcept { cache_mutable_ = on; }
    void invalidate(const BV* bv) { cache_.erase(bv); }
    void invalidate_all() { cache_.clear(); }
    size_t cached_vectors() const noexcept { return cache_.size(); }
    size_t uploads() const noexcept { return uploads_; }          ///< uploads performed so far (tests / tuning)
    context& get_context() noexcept { return *ctx_; }

private:
    struct stamp {
        uint64_t size = 0, top = 0, h = 0;
        bool operator==(const stamp& o) const noexcept { return size == o.size && top == o.top && h == o.h; }
    };
    static stamp stamp_of(const BV& bv)
    {
        stamp st;
        const typename BV::blocks_manager_type& bman = bv.get_blocks_manager();
        st.size = bv.size();
        if (!bman.is_init()) return st;
        st.top = bman.top_block_size();
        uint64_t h = 1469598103934665603ull;
        bm::word_t*** root = bman.top_blocks_root();
        for (unsigned i = 0; i < bman.top_block_size(); ++i) { h ^= (uint64_t)(uintptr_t)root[i]; h *= 1099511628211ull; }
        // a cheap CONTENT sample on top of the shape: up to 64 blocks spread over the vector, pointer + three words each.
        // It catches a vector that was destroyed and replaced by another one at the same address with the allocator
        // handing out the same sub-arrays; it cannot prove equality -- call invalidate(bv) before a cached
        // vector is destroyed or (with set_cache_mutable) changed in place.
        const unsigned nb_total = bman.top_block_size() * bm::set_sub_array_size;
        const unsigned step = nb_total > 64u ? nb_total / 64u : 1u;
        for (unsigned nb = 0; nb < nb_total; nb += step) {
            bm::word_t** sub = root[nb >> bm::set_array_shift];
            if (!sub || sub == (bm::word_t**)FULL_BLOCK_FAKE_ADDR) continue;
            const bm::word_t* blk = sub[nb & bm::set_array_mask];
            h ^= (uint64_t)(uintptr_t)blk + nb; h *= 1099511628211ull;
            if (!blk || blk == FULL_BLOCK_FAKE_ADDR) continue;
            if (BM_IS_GAP(blk)) { const bm::gap_word_t* g = BMGAP_PTR(blk); h ^= ((uint64_t)g[0] << 16) | g[1]; }
            else h ^= ((uint64_t)blk[0] << 32) ^ ((uint64_t)blk[bm::set_block_size / 2] << 16) ^ blk[bm::set_block_size - 1];
            h *= 1099511628211ull;
        }
        st.h = h;
        return st;
    }
    struct entry { std::unique_ptr<bvector> dev; stamp st; bool ro = false; uint32_t nblocks = 0; uint64_t epoch = 0; };

    /// device copies of the operands (uploading what is not resident); every operand of one operation is uploaded
    /// over the same block count, so that NULL tails behave as in the reference
    std::vector<const bvector*> resident(const bvector_type_const_ptr* src, size_t n)
    {
        std::vector<const bvector*> out;
        out.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const BV* bv = src[i];
            if (!bv) throw error(BMX_ERR_BADARG, "BMX-02: null operand");
            stamp st = stamp_of(*bv);
            bool ro = bv->is_ro();
            typename std::unordered_map<const BV*, entry>::iterator it = cache_.find(bv);
            // uploaded earlier in THIS operation (the same vector in two lists): the copy is current by definition
            bool keep = it != cache_.end() && it->second.st == st && it->second.ro == ro &&
                        (ro || cache_mutable_ || it->second.epoch == epoch_);
            if (!keep) {
                entry e;
                e.dev.reset(new bvector(*ctx_));
                e.nblocks = effective_blocks(*bv);
                upload(*bv, *e.dev, e.nblocks ? e.nblocks : 1u);
                e.st = st; e.ro = ro; e.epoch = epoch_;
                ++uploads_;
                if (it != cache_.end()) it->second = std::move(e); else it = cache_.emplace(bv, std::move(e)).first;
            }
            if (it->second.nblocks > common_blocks_) common_blocks_ = it->second.nblocks;
            out.push_back(it->second.dev.get());
        }
        return out;
    }

    context* ctx_;
    aggregator<bvector> agg_;
    arg_groups ag_;
    std::unordered_map<const BV*, entry> cache_;
    bool cache_mutable_ = false;
    size_t uploads_ = 0;
    uint32_t common_blocks_ = 1;
    uint64_t epoch_ = 1;                 // one per public operation
    struct op_scope { device_aggregator* a; explicit op_scope(device_aggregator* x) : a(x) { ++a->epoch_; } };
};

} // namespace bmx
