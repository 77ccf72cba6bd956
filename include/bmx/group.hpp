// bmx/group.hpp -- multi-GPU facade: the same bm::bvector<> / bm::aggregator<> method names over a GROUP of
// devices (include/bmx.h "device groups").  Every block column (i,j) is independent for AND/OR/XOR/SUB/COUNT
// (src/bmaggregator.h:1113-1121,1184-1218; src/bm.h:6226-6271), so a vector is sharded by block range over the
// members, every member runs the single-device engine over its shard, and the only exchange is the sum of the
// popcounts (host sum, or RCCL over xGMI with BMX_GROUP_RCCL).  Header-only, over the C-ABI.
//
//   bmx::device_group grp({0,1,2,3,4,5,6,7});              // one process, 8 GPUs
//   bmx::gbvector a(grp), b(grp), t(grp);
//   bmx::upload(host_a, a);  bmx::upload(host_b, b);        // bm_adapter.hpp: the block table is cut at the shard borders
//   t.bit_and(a, b);   auto n = bmx::count_and(a, b);
//   bmx::aggregator<bmx::gbvector> agg(grp);                // same calls as bm::aggregator<bm::bvector<>>
//   agg.add(&a); agg.add(&b); agg.combine_and(t);
#pragma once

#include <initializer_list>

#include "bvector.hpp"
#include "scanner.hpp"

namespace bmx {

class device_group {
public:
    explicit device_group(const std::vector<int>& devices, bool rccl = false)
    { check(bmx_group_create(devices.data(), (int)devices.size(), rccl ? BMX_GROUP_RCCL : BMX_GROUP_HOST_SUM, &h_)); }
    device_group(std::initializer_list<int> devices, bool rccl = false) : device_group(std::vector<int>(devices), rccl) {}
    ~device_group() { if (h_) bmx_group_destroy(h_); }
    device_group(const device_group&) = delete;
    device_group& operator=(const device_group&) = delete;
    bmx_group* handle() const noexcept { return h_; }
    int size() const { int n = 0; check(bmx_group_size(h_, &n)); return n; }
    void shard_range(uint32_t nblocks, int member, uint32_t& nb_from, uint32_t& nb_to) const
    { check(bmx_group_shard_range(h_, nblocks, member, &nb_from, &nb_to)); }
    void set_tuning(const char* key, int value)
    {
        for (int m = 0; m < size(); ++m) { bmx_ctx* c = nullptr; check(bmx_group_ctx(h_, m, &c)); check(bmx_ctx_set_tuning(c, key, value)); }
    }
    /// byte-weighted shard borders: weight[nb] = operand bytes of block column nb summed over the
    /// collection (bmx_block_table_weights per vector); must precede the first vector of that length.  -> n + 1 borders
    std::vector<uint32_t> partition_by_weight(const std::vector<uint64_t>& weight)
    {
        std::vector<uint32_t> b((size_t)size() + 1, 0);
        check(bmx_group_partition_by_weight(h_, (uint32_t)weight.size(), weight.data(), b.data()));
        return b;
    }
    void set_partition(uint32_t nblocks, const std::vector<uint32_t>& bounds)
    {
        if ((int)bounds.size() != size() + 1) check(BMX_ERR_BADARG);
        check(bmx_group_set_partition(h_, nblocks, bounds.data()));
    }
    /// ranks of the in-library RCCL communicator (0 for a host-sum group)
    int rccl_ranks() const { int n = 0; check(bmx_group_rccl_ranks(h_, &n)); return n; }
private:
    bmx_group* h_ = nullptr;
};

class gbvector;

/// bm::rs_index twin of a sharded vector: one index per shard + the ones before each shard (bmx_grs_*)
class grs_index {
public:
    grs_index() = default;
    ~grs_index() { reset(); }
    grs_index(const grs_index&) = delete;
    grs_index& operator=(const grs_index&) = delete;
    size_type count() const { uint64_t c = 0; if (h_) check(bmx_grs_count(h_, &c)); return c; }
    bmx_grs* handle() const noexcept { return h_; }
private:
    friend class gbvector;
    void reset() { if (h_) { bmx_grs_free(grp_, h_); h_ = nullptr; } }
    bmx_group* grp_ = nullptr;
    bmx_grs* h_ = nullptr;
};

/// bm::bvector<> twin sharded over the members of a device_group
class gbvector {
public:
    typedef bmx::size_type size_type;
    typedef bvector::optmode optmode;
    explicit gbvector(device_group& g) : grp_(&g) {}
    ~gbvector() { clear(); }
    gbvector(const gbvector&) = delete;
    gbvector& operator=(const gbvector&) = delete;
    gbvector(gbvector&& o) noexcept : grp_(o.grp_), h_(o.h_) { o.h_ = nullptr; }
    gbvector& operator=(gbvector&& o) noexcept { if (this != &o) { clear(); grp_ = o.grp_; h_ = o.h_; o.h_ = nullptr; } return *this; }

    void clear() { if (h_) { bmx_gvec_free(grp_->handle(), h_); h_ = nullptr; } }
    bool empty_handle() const noexcept { return h_ == nullptr; }
    device_group& get_group() const noexcept { return *grp_; }
    bmx_gvec* handle() const noexcept { return h_; }
    void adopt(bmx_gvec* h) { clear(); h_ = h; }

    void assign_block_table(uint64_t nbits, uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
                            const uint32_t* bit_slab, uint32_t n_bit_blocks, const uint16_t* gap_slab, uint64_t gap_words)
    {
        bmx_gvec* h = nullptr;
        check(bmx_gvec_upload(grp_->handle(), nbits, nblocks, kinds, offs, bit_slab, n_bit_blocks, gap_slab, gap_words, &h));
        adopt(h);
    }
    size_type size() const { uint64_t n = 0; if (h_) check(bmx_gvec_info(h_, &n, nullptr, nullptr, nullptr, nullptr)); return n; }
    uint32_t block_count() const { uint32_t n = 0; if (h_) check(bmx_gvec_info(h_, nullptr, &n, nullptr, nullptr, nullptr)); return n; }
    void calc_stat(bvector::statistics* st) const
    {
        uint32_t c[4] = {0, 0, 0, 0};
        if (h_) check(bmx_gvec_info(h_, nullptr, nullptr, c, nullptr, nullptr));
        st->null_blocks = c[BMX_NULL]; st->full_blocks = c[BMX_FULL]; st->bit_blocks = c[BMX_BIT]; st->gap_blocks = c[BMX_GAP];
    }
    size_type count() const { uint64_t c = 0; if (h_) check(bmx_gvec_count(grp_->handle(), h_, &c)); return c; }
    bool any() const { return count() != 0; }

    gbvector& bit_and(const gbvector& a, const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_AND, a, b, opt); }
    gbvector& bit_or(const gbvector& a, const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_OR, a, b, opt); }
    gbvector& bit_xor(const gbvector& a, const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_XOR, a, b, opt); }
    gbvector& bit_sub(const gbvector& a, const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_SUB, a, b, opt); }
    gbvector& bit_and(const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_AND, *this, b, opt); }
    gbvector& bit_or(const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_OR, *this, b, opt); }
    gbvector& bit_xor(const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_XOR, *this, b, opt); }
    gbvector& bit_sub(const gbvector& b, optmode opt = bvector::opt_none) { return op3(BMX_SUB, *this, b, opt); }
    bool equal(const gbvector& bv) const
    {
        if (!h_ || !bv.h_) return count() == 0 && bv.count() == 0;
        uint64_t c = 0; check(bmx_gvec_count_op2(grp_->handle(), BMX_XOR, h_, bv.h_, &c)); return c == 0;
    }
    // ---- rank / select (same method set as bmx::bvector; queries are routed to the member that owns the block) ----
    void build_rs_index(grs_index* rs) const
    {
        rs->reset(); rs->grp_ = grp_->handle(); require();
        check(bmx_grs_build(grp_->handle(), h_, &rs->h_));
    }
    size_type count_to(size_type n, const grs_index& rs) const
    { uint64_t out = 0; require(); check(bmx_grank_batch(grp_->handle(), h_, rs.h_, &n, 1, &out)); return out; }
    size_type rank(size_type n, const grs_index& rs) const { return count_to(n, rs); }
    bool select(size_type rank_in, size_type& pos, const grs_index& rs) const
    {
        uint64_t p = 0; uint8_t f = 0; require();
        check(bmx_gselect_batch(grp_->handle(), h_, rs.h_, &rank_in, 1, &p, &f));
        if (f) pos = p;
        return f != 0;
    }
    size_type count_range(size_type left, size_type right, const grs_index& rs) const        // src/bm.h:3548
    {
        if (left > right) std::swap(left, right);
        size_type q[2] = {right, left ? left - 1 : 0}, out[2];
        count_to(q, 2, out, rs);
        return out[0] - (left ? out[1] : 0);
    }
    bool find_rank(size_type rank_in, size_type from, size_type& pos, const grs_index& rs) const   // src/bm.h:5279
    {
        if (!rank_in) return false;
        size_type before = from ? count_to(from - 1, rs) : 0;
        return select(rank_in + before, pos, rs);
    }
    void count_to(const size_type* n, size_t q, size_type* out, const grs_index& rs) const
    { require(); check(bmx_grank_batch(grp_->handle(), h_, rs.h_, n, q, out)); }
    void select(const size_type* rank_in, size_t q, size_type* pos, uint8_t* found, const grs_index& rs) const
    { require(); check(bmx_gselect_batch(grp_->handle(), h_, rs.h_, rank_in, q, pos, found)); }
private:
    void require() const { if (!h_) throw error(BMX_ERR_BADARG, "BMX-02: vector holds no device data"); }
    gbvector& op3(int op, const gbvector& a, const gbvector& b, optmode opt)
    {
        a.require(); b.require();
        bmx_gvec* r = nullptr;
        check(bmx_gvec_op2(grp_->handle(), op, a.h_, b.h_, opt == bvector::opt_compress, &r));
        adopt(r);
        return *this;
    }
    device_group* grp_;
    bmx_gvec* h_ = nullptr;
};

namespace detail {
inline size_type gcount_op(int op, const gbvector& a, const gbvector& b)
{
    if (a.empty_handle() || b.empty_handle()) {
        if (op == BMX_AND) return 0;
        if (a.empty_handle()) return op == BMX_SUB ? 0 : b.count();
        return a.count();
    }
    uint64_t c = 0;
    check(bmx_gvec_count_op2(a.get_group().handle(), op, a.handle(), b.handle(), &c));
    return c;
}
} // namespace detail
inline size_type count_and(const gbvector& a, const gbvector& b) { return detail::gcount_op(BMX_AND, a, b); }
inline size_type count_or(const gbvector& a, const gbvector& b) { return detail::gcount_op(BMX_OR, a, b); }
inline size_type count_xor(const gbvector& a, const gbvector& b) { return detail::gcount_op(BMX_XOR, a, b); }
inline size_type count_sub(const gbvector& a, const gbvector& b) { return detail::gcount_op(BMX_SUB, a, b); }

/// bmx_gcollection_prepare: every member of the group transposes its block range of the vectors into a packed collection
/// (role BMX_ROLE_OR / BMX_ROLE_AND / BMX_ROLE_SUB); the group's aggregations and pipelines over those vectors then use it
inline void collection_prepare(device_group& g, const std::vector<const gbvector*>& vecs, int role)
{
    std::vector<const bmx_gvec*> h(vecs.size());
    for (size_t i = 0; i < vecs.size(); ++i) h[i] = vecs[i]->handle();
    check(bmx_gcollection_prepare(g.handle(), h.data(), h.size(), role));
}

/// bm::aggregator<BV> over sharded vectors: the overload of bmx::aggregator that takes a device group
template <>
class aggregator<gbvector> {
public:
    typedef gbvector bvector_type;
    typedef const gbvector* bvector_type_const_ptr;
    struct arg_groups {
        std::vector<bvector_type_const_ptr> arg_bv0, arg_bv1;
        void reset() { arg_bv0.clear(); arg_bv1.clear(); }
        size_t add(const gbvector* bv, unsigned agr_group)
        {
            if (agr_group > 1) throw error(BMX_ERR_RANGE, "BMX-03: Incorrect range or index [agr_group > 1]");
            if (!bv) return 0;
            std::vector<bvector_type_const_ptr>& v = agr_group ? arg_bv1 : arg_bv0;
            v.push_back(bv);
            return v.size();
        }
    };
    /// counts-only pipeline (src/bmaggregator.h:222-341 with agg_opt_only_counts)
    template <class Opt = agg_opt_only_counts>
    class pipeline {
    public:
        typedef Opt options_type;
        explicit pipeline(device_group& g) : grp_(&g) {}
        ~pipeline()
        {
            if (h_) bmx_gpipeline_destroy(grp_->handle(), h_);
            for (size_t i = 0; i < groups_.size(); ++i) delete groups_[i];
        }
        pipeline(const pipeline&) = delete;
        pipeline& operator=(const pipeline&) = delete;
        arg_groups* add() { if (h_) throw error(BMX_ERR_BADARG, "pipeline already complete()"); groups_.push_back(new arg_groups()); return groups_.back(); }
        size_t size() const noexcept { return groups_.size(); }
        bool is_complete() const noexcept { return h_ != nullptr; }
        void complete()
        {
            std::vector<const bmx_gvec*> al, sl; std::vector<uint32_t> an, sn;
            for (size_t g = 0; g < groups_.size(); ++g) {
                an.push_back((uint32_t)groups_[g]->arg_bv0.size()); sn.push_back((uint32_t)groups_[g]->arg_bv1.size());
                for (size_t i = 0; i < groups_[g]->arg_bv0.size(); ++i) al.push_back(groups_[g]->arg_bv0[i]->handle());
                for (size_t i = 0; i < groups_[g]->arg_bv1.size(); ++i) sl.push_back(groups_[g]->arg_bv1[i]->handle());
            }
            check(bmx_gpipeline_create(grp_->handle(), al.data(), an.data(), sl.data(), sn.data(), groups_.size(), &h_));
            counts_.assign(groups_.size(), 0);
            if (limit_ != ~0ull) check(bmx_gpipeline_set_search_count_limit(grp_->handle(), h_, limit_));
        }
        /// pipeline::set_search_count_limit (src/bmaggregator.h:255): every member searches its shard under the same limit
        void set_search_count_limit(size_type limit)
        {
            limit_ = limit;
            if (h_) check(bmx_gpipeline_set_search_count_limit(grp_->handle(), h_, limit_));
        }
        const std::vector<size_type>& get_bv_count_vector() const noexcept { return counts_; }
        /// device time each member spent in the last run (HIP events), ms
        std::vector<float> last_ms() const { std::vector<float> ms((size_t)grp_->size()); check(bmx_gpipeline_last_ms(grp_->handle(), h_, ms.data())); return ms; }
    private:
        friend class aggregator;
        device_group* grp_;
        std::vector<arg_groups*> groups_;
        std::vector<size_type> counts_;
        bmx_gpipeline* h_ = nullptr;
        uint64_t limit_ = ~0ull;
    };

    explicit aggregator(device_group& g) : grp_(&g) {}
    size_t add(const gbvector* bv, unsigned agr_group = 0) { return ag_.add(bv, agr_group); }
    void reset() { ag_.reset(); }
    void set_optimization(bool opt_compress = true) { opt_compress_ = opt_compress; }

    void combine_or(gbvector& bv_target) { combine_or(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size()); }
    void combine_and(gbvector& bv_target) { combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), nullptr, 0, false); }
    bool combine_and_sub(gbvector& bv_target)
    { return combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size(), false); }
    void combine_or(gbvector& bv_target, const bvector_type_const_ptr* bv_src, size_t src_size)
    {
        std::vector<const bmx_gvec*> h(src_size);
        for (size_t i = 0; i < src_size; ++i) h[i] = bv_src[i]->handle();
        ag_.reset();
        bmx_gvec* r = nullptr;
        check(bmx_gagg_or(grp_->handle(), h.data(), src_size, opt_compress_ ? 1 : 0, &r));
        bv_target.adopt(r);
    }
    bool combine_and_sub(gbvector& bv_target, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                         const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size, bool /*any*/)
    {
        std::vector<const bmx_gvec*> a(src_and_size), s(src_sub_size);
        for (size_t i = 0; i < src_and_size; ++i) a[i] = bv_src_and[i]->handle();
        for (size_t i = 0; i < src_sub_size; ++i) s[i] = bv_src_sub[i]->handle();
        bmx_gvec* r = nullptr; int any = 0;
        check(bmx_gagg_and_sub(grp_->handle(), a.data(), src_and_size, s.data(), src_sub_size, &r, &any));
        bv_target.adopt(r);
        return any != 0;
    }
    /// find_first_and_sub(idx)  src/bmaggregator.h:1079 / C-style :1458
    bool find_first_and_sub(size_type& idx)
    { return find_first_and_sub(idx, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size()); }
    bool find_first_and_sub(size_type& idx, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                            const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size)
    {
        std::vector<const bmx_gvec*> a(src_and_size), s(src_sub_size);
        for (size_t i = 0; i < src_and_size; ++i) a[i] = bv_src_and[i]->handle();
        for (size_t i = 0; i < src_sub_size; ++i) s[i] = bv_src_sub[i]->handle();
        int found = 0; uint64_t p = 0;
        check(bmx_gfind_first_and_sub(grp_->handle(), a.data(), src_and_size, s.data(), src_sub_size, &found, &p));
        if (found) idx = p;
        return found != 0;
    }
    template <class TPipe>
    void combine_and_sub(TPipe& pipe)
    {
        if (!pipe.is_complete()) throw error(BMX_ERR_BADARG, "pipeline is not complete()");
        if (!pipe.size()) return;
        check(bmx_gpipeline_run_counts(grp_->handle(), pipe.h_, pipe.counts_.data()));
    }
private:
    device_group* grp_;
    arg_groups ag_;
    bool opt_compress_ = false;
};

// ---- the scanner call pattern over sharded bit-planes (scanner.hpp): bmx::gslice_scanner ----
template <> struct scanner_traits<gbvector> {
    typedef device_group ctx_type;
    typedef bmx_gvec handle_type;
    static int compare(ctx_type& g, const handle_type* const* h, size_t n, int pred, uint64_t v0, uint64_t v1, uint64_t size,
                       const handle_type* nn, handle_type** r, uint64_t* cnt)
    { return bmx_gslice_compare(g.handle(), h, n, pred, v0, v1, size, nn, r, cnt); }
    static int eq_counts(ctx_type& g, const handle_type* const* h, size_t n, const uint64_t* values, size_t nv, uint64_t size,
                         const handle_type* nn, uint64_t* counts)
    { return bmx_gslice_eq_counts(g.handle(), h, n, values, nv, size, nn, counts); }
};
typedef basic_slice_scanner<gbvector> gslice_scanner;

} // namespace bmx
