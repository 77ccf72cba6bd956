// bmx/bvector.hpp -- header-only C++ host facade over the C-ABI of include/bmx.h.
//
// Mirrors the part of the reference API that lies on the hot path (same method
// names, argument meaning and return values) so code written against
// bm::bvector<> / bm::aggregator<> / bm::count_* reads the same:
//
//   bm::bvector<>::bit_and/bit_or/bit_xor/bit_sub (2- and 3-operand)   src/bm.h:1745-1850
//   bm::bvector<>::count / build_rs_index / count_to / rank / select   src/bm.h:2431,2531,3120,1449,5350
//   bm::bit_import_u32                                                   src/bmbvimport.h:46
//   bm::count_and/or/xor/sub                                             src/bmalgo.h:49,149,81,115
//   bm::rs_index::count                                                  src/bmrs.h:340
//   bm::aggregator<BV>::add/reset/combine_or/combine_and/combine_and_sub src/bmaggregator.h:1013-1079
//   bm::aggregator<BV>::pipeline<agg_opt_only_counts> + combine_and_sub(pipe)  :222-341,1292
//
// Differences that follow from device residency: a bmx::bvector is immutable
// once filled (set_bit & friends stay with the host bm::bvector<>; see
// bmx/bm_adapter.hpp for the bridge), and every object belongs to a bmx::context
// (one GPU + one HIP stream).  Errors: C status codes are turned into
// bmx::error (the reference throws std::bad_alloc / std::range_error too).
// There is no CPU fallback.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../bmx.h"

namespace bmx {

class error : public std::runtime_error {
public:
    error(int status, const std::string& what) : std::runtime_error(what), status_(status) {}
    int status() const noexcept { return status_; }
private:
    int status_;
};

inline void check(int rc)
{
    if (rc != BMX_OK) {
        std::string m = bmx_error_msg(rc);
        const char* d = bmx_last_error();
        if (d && *d) { m += " ["; m += d; m += "]"; }
        throw error(rc, m);
    }
}

typedef uint64_t size_type;

/// one device + one HIP stream
class context {
public:
    explicit context(int device = 0, void* hip_stream = nullptr) { check(bmx_ctx_create(device, hip_stream, &h_)); }
    ~context() { if (h_) bmx_ctx_destroy(h_); }
    context(const context&) = delete;
    context& operator=(const context&) = delete;
    bmx_ctx* handle() const noexcept { return h_; }
    void synchronize() { check(bmx_ctx_synchronize(h_)); }
    uint64_t mem_used() const { uint64_t b = 0; check(bmx_ctx_mem_used(h_, &b)); return b; }
private:
    bmx_ctx* h_ = nullptr;
};

class bvector;

/// bm::rs_index twin (built on the device; exportable in the reference layout)
class rs_index {
public:
    rs_index() = default;
    ~rs_index() { reset(); }
    rs_index(const rs_index&) = delete;
    rs_index& operator=(const rs_index&) = delete;
    size_type count() const { uint64_t c = 0; if (h_) check(bmx_rs_count(h_, &c)); return c; }
    /// what the index holds on the device: total bytes, whether the vector was laid out as rank lines (one line per rank query) and
    /// as select lines (offset width 16 | 32, 0 = none: select then searches the rank lines' directory or the block tables)
    struct device_layout { uint64_t bytes = 0; bool rank_lines = false; int select_offset_bits = 0; uint64_t select_lines_bytes = 0; };
    device_layout layout() const
    {
        device_layout l;
        if (!h_) return l;
        int has = 0;
        check(bmx_rs_info(h_, &l.bytes, &has)); l.rank_lines = has != 0;
        check(bmx_rs_select_format(h_, &l.select_offset_bits, &l.select_lines_bytes));
        return l;
    }
    /// bcount[nb] and sub_count[nb] = first | second<<16 | aux0<<32 | aux1<<48 (src/bm.h:2646-2656)
    void export_blocks(std::vector<uint32_t>& bcount, std::vector<uint64_t>& sub_count, uint32_t nblocks) const
    {
        bcount.assign(nblocks, 0); sub_count.assign(nblocks, 0);
        if (h_ && nblocks) check(bmx_rs_export(ctx_, h_, bcount.data(), sub_count.data()));
    }
    bmx_rs* handle() const noexcept { return h_; }
private:
    friend class bvector;
    void reset() { if (h_) { bmx_rs_free(ctx_, h_); h_ = nullptr; } }
    bmx_ctx* ctx_ = nullptr;
    bmx_rs* h_ = nullptr;
};

class bvector {
public:
    typedef bmx::size_type size_type;
    /// bvector<>::optmode (src/bm.h:129-135)
    enum optmode { opt_none = 0, opt_free_0 = 1, opt_free_01 = 2, opt_compress = 3 };
    struct statistics { uint32_t bit_blocks, gap_blocks, full_blocks, null_blocks; };

    explicit bvector(context& ctx) : ctx_(&ctx) {}
    ~bvector() { clear(); }
    bvector(const bvector&) = delete;
    bvector& operator=(const bvector&) = delete;
    bvector(bvector&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    bvector& operator=(bvector&& o) noexcept { if (this != &o) { clear(); ctx_ = o.ctx_; h_ = o.h_; o.h_ = nullptr; } return *this; }

    void clear() { if (h_) { bmx_vec_free(ctx_->handle(), h_); h_ = nullptr; } }
    bool empty_handle() const noexcept { return h_ == nullptr; }
    context& get_context() const noexcept { return *ctx_; }
    bmx_vec* handle() const noexcept { return h_; }
    void adopt(bmx_vec* h) { clear(); h_ = h; }

    /// upload of a flattened block table (walk of blocks_manager::top_blocks_root())
    void assign_block_table(uint64_t nbits, uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
                            const uint32_t* bit_slab, uint32_t n_bit_blocks, const uint16_t* gap_slab, uint64_t gap_words)
    {
        bmx_vec* h = nullptr;
        check(bmx_vec_upload(ctx_->handle(), nbits, nblocks, kinds, offs, bit_slab, n_bit_blocks, gap_slab, gap_words, &h));
        adopt(h);
    }

    size_type size() const { uint64_t n = 0; if (h_) check(bmx_vec_info(h_, &n, nullptr, nullptr, nullptr, nullptr)); return n; }
    uint32_t block_count() const { uint32_t n = 0; if (h_) check(bmx_vec_info(h_, nullptr, &n, nullptr, nullptr, nullptr)); return n; }
    /// bvector<>::calc_stat (src/bm.h:4010)
    void calc_stat(statistics* st) const
    {
        uint32_t c[4] = {0, 0, 0, 0};
        if (h_) check(bmx_vec_info(h_, nullptr, nullptr, c, nullptr, nullptr));
        st->null_blocks = c[BMX_NULL]; st->full_blocks = c[BMX_FULL]; st->bit_blocks = c[BMX_BIT]; st->gap_blocks = c[BMX_GAP];
    }

    /// bvector<>::count()  src/bm.h:2431
    size_type count() const { uint64_t c = 0; if (h_) check(bmx_count(ctx_->handle(), h_, &c)); return c; }
    /// the sorted positions of the set bits (device compaction, bmx_vec_to_indices): what bm::bvector<>::enumerator or
    /// bm::for_each_bit would feed into a container, in one call
    void to_indices(std::vector<size_type>& out) const
    {
        out.clear();
        if (!h_) return;
        uint64_t n = count();
        out.resize(n);
        if (n) check(bmx_vec_to_indices(ctx_->handle(), h_, 8, out.data(), n, &n));
    }
    /// feeds every set bit, ascending, into a back-insert style iterator (*bi = idx)
    template <class BII> void copy_to(BII bi) const
    {
        std::vector<size_type> idx; to_indices(idx);
        for (size_type p : idx) { *bi = p; ++bi; }
    }
    /// bvector<>::any() / find(pos) (src/bm.h:1593): first set bit = find_first_and_sub over the one-vector AND group
    bool find(size_type& pos) const
    {
        if (!h_) return false;
        const bmx_vec* one[1] = {h_};
        int found = 0; uint64_t p = 0;
        check(bmx_find_first_and_sub(ctx_->handle(), one, 1, nullptr, 0, &found, &p));
        if (found) pos = p;
        return found != 0;
    }
    bool any() const { size_type p; return find(p); }

    // ---- 3-operand set algebra: *this = bv1 OP bv2   (src/bm.h:6185,5973,6072,6403) ----
    bvector& bit_and(const bvector& bv1, const bvector& bv2, optmode opt = opt_none) { return op3(BMX_AND, bv1, bv2, opt); }
    bvector& bit_or(const bvector& bv1, const bvector& bv2, optmode opt = opt_none) { return op3(BMX_OR, bv1, bv2, opt); }
    bvector& bit_xor(const bvector& bv1, const bvector& bv2, optmode opt = opt_none) { return op3(BMX_XOR, bv1, bv2, opt); }
    bvector& bit_sub(const bvector& bv1, const bvector& bv2, optmode opt = opt_none) { return op3(BMX_SUB, bv1, bv2, opt); }
    // ---- 2-operand forms: *this OP= bv   (src/bm.h:1811-1850) ----
    bvector& bit_and(const bvector& bv, optmode opt = opt_none) { return op2(BMX_AND, bv, opt); }
    bvector& bit_or(const bvector& bv, optmode opt = opt_none) { return op2(BMX_OR, bv, opt); }
    bvector& bit_xor(const bvector& bv, optmode opt = opt_none) { return op2(BMX_XOR, bv, opt); }
    bvector& bit_sub(const bvector& bv, optmode opt = opt_none) { return op2(BMX_SUB, bv, opt); }
    bvector& operator&=(const bvector& bv) { return bit_and(bv); }
    bvector& operator|=(const bvector& bv) { return bit_or(bv); }
    bvector& operator^=(const bvector& bv) { return bit_xor(bv); }
    bvector& operator-=(const bvector& bv) { return bit_sub(bv); }

    /// representation-agnostic equality (bvector<>::equal / compare()==0)
    bool equal(const bvector& bv) const
    {
        if (!h_ || !bv.h_) return count() == 0 && bv.count() == 0;
        uint64_t c = 0; check(bmx_count_op2(ctx_->handle(), BMX_XOR, h_, bv.h_, &c)); return c == 0;
    }

    // ---- rank / select ----
    /// build_rs_index(&rs)  src/bm.h:2531
    void build_rs_index(rs_index* rs) const
    {
        rs->reset();
        rs->ctx_ = ctx_->handle();
        require();
        check(bmx_rs_build(ctx_->handle(), h_, &rs->h_));
    }
    /// count_to(n, rs): ones in [0..n]  src/bm.h:3120
    size_type count_to(size_type n, const rs_index& rs) const
    {
        uint64_t out = 0; require(); check(bmx_rank_batch(ctx_->handle(), h_, rs.h_, &n, 1, &out)); return out;
    }
    size_type rank(size_type n, const rs_index& rs) const { return count_to(n, rs); }      // src/bm.h:1449
    /// select(rank, pos, rs): rank is 1-based  src/bm.h:5350
    bool select(size_type rank_in, size_type& pos, const rs_index& rs) const
    {
        uint64_t p = 0; uint8_t f = 0; require();
        check(bmx_select_batch(ctx_->handle(), h_, rs.h_, &rank_in, 1, &p, &f));
        if (f) pos = p;
        return f != 0;
    }
    /// rank_corrected(n) = rank(n) - bit(n)  src/bm.h:3229
    size_type rank_corrected(size_type n, const rs_index& rs) const { size_type b, r; bit_and_rank(n, rs, b, r); return r - b; }
    /// count_to_test(n) = bit(n) ? rank(n) : 0  src/bm.h:3173
    size_type count_to_test(size_type n, const rs_index& rs) const { size_type b, r; bit_and_rank(n, rs, b, r); return b ? r : 0; }
    /// count_range(left, right): ones in [left..right], arguments swapped when left > right  src/bm.h:3548
    size_type count_range(size_type left, size_type right, const rs_index& rs) const
    {
        if (left > right) std::swap(left, right);
        size_type q[2] = {right, left ? left - 1 : 0}, out[2];
        count_to(q, 2, out, rs);
        return out[0] - (left ? out[1] : 0);
    }
    /// find_rank(rank, from, pos, rs): rank-th set bit at or after `from`  src/bm.h:5279
    bool find_rank(size_type rank_in, size_type from, size_type& pos, const rs_index& rs) const
    {
        if (!rank_in) return false;
        size_type before = from ? count_to(from - 1, rs) : 0;
        return select(rank_in + before, pos, rs);
    }
    /// batched forms (one launch for q queries)
    void count_to(const size_type* n, size_t q, size_type* out, const rs_index& rs) const
    { require(); check(bmx_rank_batch(ctx_->handle(), h_, rs.h_, n, q, out)); }
    void select(const size_type* rank_in, size_t q, size_type* pos, uint8_t* found, const rs_index& rs) const
    { require(); check(bmx_select_batch(ctx_->handle(), h_, rs.h_, rank_in, q, pos, found)); }

    /// export to raw words (twin of bit_import_u32)
    void export_words(uint32_t* words, uint64_t nwords) const
    {
        if (!h_) { for (uint64_t i = 0; i < nwords; ++i) words[i] = 0; return; }
        check(bmx_vec_to_words(ctx_->handle(), h_, words, nwords));
    }

private:
    void require() const { if (!h_) throw error(BMX_ERR_BADARG, "BMX-02: vector holds no device data"); }
    void bit_and_rank(size_type n, const rs_index& rs, size_type& bit, size_type& rank_n) const
    {
        size_type q[2] = {n, n ? n - 1 : 0}, out[2];
        count_to(q, 2, out, rs);
        rank_n = out[0]; bit = out[0] - (n ? out[1] : 0);
    }
    bvector& op3(int op, const bvector& a, const bvector& b, optmode opt)
    {
        a.require(); b.require();
        bmx_vec* r = nullptr;
        check(bmx_op2(ctx_->handle(), op, a.h_, b.h_, opt == opt_compress, &r));
        adopt(r);                      // also correct when this == &a or this == &b: the result is built first
        return *this;
    }
    bvector& op2(int op, const bvector& b, optmode opt)
    {
        require(); b.require();
        bmx_vec* r = nullptr;
        check(bmx_op2(ctx_->handle(), op, h_, b.h_, opt == opt_compress, &r));
        adopt(r);
        return *this;
    }
    context* ctx_;
    bmx_vec* h_ = nullptr;
};

/// An asynchronous result (bmx_op2_dev): complete on the context's stream, not yet an ordinary vector.  It is accepted as an
/// operand by bit_op_async() and by nothing else; wait() waits for it and moves it into a bvector.  No counterpart in the
/// reference (its vectors live on the host); the operations are those of bvector::bit_and/or/xor/sub, src/bm.h:6185,5973,6072,6403.
class pending {
public:
    pending(context& ctx, bmx_pending* h) : ctx_(&ctx), h_(h) {}
    pending(const pending&) = delete;
    pending& operator=(const pending&) = delete;
    pending(pending&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    ~pending() { if (h_) bmx_pending_free(ctx_->handle(), h_); }
    const bmx_pending* handle() const noexcept { return h_; }
    context& get_context() const noexcept { return *ctx_; }
    /// waits for this result only; the handle is consumed
    void wait(bvector& target)
    {
        bmx_vec* v = nullptr;
        bmx_pending* h = h_; h_ = nullptr;
        check(bmx_pending_wait(ctx_->handle(), h, &v));
        target.adopt(v);
    }
private:
    context* ctx_;
    bmx_pending* h_;
};

/// op = BMX_AND / BMX_OR / BMX_XOR / BMX_SUB (opt_none) over vectors of any block kinds or unresolved results: enqueued, not waited for
inline pending bit_op_async(int op, const bvector& a, const bvector& b)
{
    bmx_pending* p = nullptr;
    check(bmx_op2_dev(a.get_context().handle(), op, a.handle(), nullptr, b.handle(), nullptr, &p));
    return pending(a.get_context(), p);
}
inline pending bit_op_async(int op, const pending& a, const bvector& b)
{
    bmx_pending* p = nullptr;
    check(bmx_op2_dev(a.get_context().handle(), op, nullptr, a.handle(), b.handle(), nullptr, &p));
    return pending(a.get_context(), p);
}
inline pending bit_op_async(int op, const bvector& a, const pending& b)
{
    bmx_pending* p = nullptr;
    check(bmx_op2_dev(a.get_context().handle(), op, a.handle(), nullptr, nullptr, b.handle(), &p));
    return pending(a.get_context(), p);
}
inline pending bit_op_async(int op, const pending& a, const pending& b)
{
    bmx_pending* p = nullptr;
    check(bmx_op2_dev(a.get_context().handle(), op, nullptr, a.handle(), nullptr, b.handle(), &p));
    return pending(a.get_context(), p);
}

/// bm::bit_import_u32(bv, bit_arr, bit_arr_size, optimize)  src/bmbvimport.h:46
inline void bit_import_u32(bvector& bv, const unsigned int* bit_arr, size_type bit_arr_size, bool optimize)
{
    bmx_vec* h = nullptr;
    check(bmx_vec_import_bits(bv.get_context().handle(), bit_arr, bit_arr_size, optimize, &h));
    bv.adopt(h);
}

namespace detail {
inline size_type count_op(int op, const bvector& a, const bvector& b)
{
    if (a.empty_handle() || b.empty_handle()) {
        if (op == BMX_AND) return 0;
        if (a.empty_handle()) return op == BMX_SUB ? 0 : b.count();
        return a.count();
    }
    uint64_t c = 0;
    check(bmx_count_op2(a.get_context().handle(), op, a.handle(), b.handle(), &c));
    return c;
}
} // namespace detail
inline size_type count_and(const bvector& a, const bvector& b) { return detail::count_op(BMX_AND, a, b); }   // src/bmalgo.h:49
inline size_type count_or(const bvector& a, const bvector& b) { return detail::count_op(BMX_OR, a, b); }     // :149
inline size_type count_xor(const bvector& a, const bvector& b) { return detail::count_op(BMX_XOR, a, b); }   // :81
inline size_type count_sub(const bvector& a, const bvector& b) { return detail::count_op(BMX_SUB, a, b); }   // :115

/// run options (src/bmaggregator.h:62-103)
template <bool OBvects = true, bool OCounts = false, bool OSearchMasks = false>
struct agg_run_options {
    static constexpr bool is_make_results() noexcept { return OBvects; }
    static constexpr bool is_compute_counts() noexcept { return OCounts; }
    static constexpr bool is_masks() noexcept { return OSearchMasks; }          // :78: honours set_range_hint
};
typedef agg_run_options<false, false> agg_opt_disable_bvects_and_counts;   // :84
typedef agg_run_options<false, true> agg_opt_only_counts;                  // :92
typedef agg_run_options<true, true> agg_opt_bvect_and_counts;              // :100

/// bm::aggregator<BV> twin (src/bmaggregator.h:120)
template <class BV = bvector>
class aggregator {
public:
    typedef BV bvector_type;
    typedef const BV* bvector_type_const_ptr;

    /// aggregator::arg_groups (src/bmaggregator.h:2925): group 0 = AND, group 1 = SUB
    struct arg_groups {
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

    /// aggregator::pipeline<Opt> (src/bmaggregator.h:222-341)
    template <class Opt = agg_run_options<> >
    class pipeline {
    public:
        typedef Opt options_type;
        explicit pipeline(context& ctx) : ctx_(&ctx) {}
        ~pipeline()
        {
            if (h_) bmx_pipeline_destroy(ctx_->handle(), h_);
            for (size_t i = 0; i < groups_.size(); ++i) delete groups_[i];
            for (size_t i = 0; i < results_.size(); ++i) delete results_[i];
        }
        /// set_or_target (:245): group results are OR-ed into *bv_or (re-seated to the updated device vector)
        void set_or_target(BV* bv_or) noexcept { or_target_ = bv_or; }
        /// set_search_count_limit (:255, honoured at :1365): a group needs no more than `limit` hits ("can find more, cannot find
        /// less"): the counts run stops launching block-column windows once every group has enough
        void set_search_count_limit(size_type limit)
        {
            search_count_limit_ = limit;
            if (h_) check(bmx_pipeline_set_search_count_limit(ctx_->handle(), h_, (uint64_t)limit));
        }
        /// result vectors (nullptr where a group found nothing, :1406-1415); owned by the pipeline
        std::vector<BV*>& get_bv_res_vector() noexcept { return results_; }
        pipeline(const pipeline&) = delete;
        pipeline& operator=(const pipeline&) = delete;
        arg_groups* add() { if (h_) throw error(BMX_ERR_BADARG, "pipeline already complete()"); groups_.push_back(new arg_groups()); return groups_.back(); }
        size_t size() const noexcept { return groups_.size(); }
        bool is_complete() const noexcept { return h_ != nullptr; }
        void complete()
        {
            std::vector<const bmx_vec*> al, sl; std::vector<uint32_t> an, sn;
            for (size_t g = 0; g < groups_.size(); ++g) {
                an.push_back((uint32_t)groups_[g]->arg_bv0.size()); sn.push_back((uint32_t)groups_[g]->arg_bv1.size());
                for (size_t i = 0; i < groups_[g]->arg_bv0.size(); ++i) al.push_back(groups_[g]->arg_bv0[i]->handle());
                for (size_t i = 0; i < groups_[g]->arg_bv1.size(); ++i) sl.push_back(groups_[g]->arg_bv1[i]->handle());
            }
            check(bmx_pipeline_create(ctx_->handle(), al.data(), an.data(), sl.data(), sn.data(), groups_.size(), &h_));
            if (search_count_limit_ != ~size_type(0)) check(bmx_pipeline_set_search_count_limit(ctx_->handle(), h_, (uint64_t)search_count_limit_));
            counts_.assign(groups_.size(), 0);
        }
        const std::vector<size_type>& get_bv_count_vector() const noexcept { return counts_; }
        bmx_pipeline* handle() const noexcept { return h_; }
    private:
        friend class aggregator;
        context* ctx_;
        std::vector<arg_groups*> groups_;
        std::vector<size_type> counts_;
        std::vector<BV*> results_;
        BV* or_target_ = nullptr;
        size_type search_count_limit_ = ~size_type(0);
        bmx_pipeline* h_ = nullptr;
    };

    explicit aggregator(context& ctx) : ctx_(&ctx) {}

    size_t add(const BV* bv, unsigned agr_group = 0) { return ag_.add(bv, agr_group); }   // :1013
    void reset() { ag_.reset(); }                                                          // :941

    void combine_or(BV& bv_target) { combine_or(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size()); }      // :1021
    void combine_and(BV& bv_target) { combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), nullptr, 0, false); } // :1030
    bool combine_and_sub(BV& bv_target)                                                                       // :1044
    { return combine_and_sub(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size(), false); }

    /// C-style overloads (src/bmaggregator.h:1101,1162)
    void combine_or(BV& bv_target, const bvector_type_const_ptr* bv_src, size_t src_size)
    {
        std::vector<const bmx_vec*> h(src_size);
        for (size_t i = 0; i < src_size; ++i) h[i] = bv_src[i]->handle();
        ag_.reset();                 // the reference clears the member arg-groups here (src/bmaggregator.h:1110)
        bmx_vec* r = nullptr;
        check(bmx_agg_or_opt(ctx_->handle(), h.data(), src_size, opt_compress_ ? 1 : 0, &r));
        bv_target.adopt(r);
    }
    /// combine_and(target, src, n)  src/bmaggregator.h:1127 (the older C-style path; it also resets the member
    /// arg-groups, :1143).  Same content as the AND-SUB path without a SUB group; blocks are stored compressed
    /// here where the reference stores them with opt_mode_ (representation only).
    void combine_and(BV& bv_target, const bvector_type_const_ptr* bv_src, size_t src_size)
    {
        if (src_size > 1) ag_.reset();
        (void)combine_and_sub(bv_target, bv_src, src_size, nullptr, 0, false);
    }
    bool combine_and_sub(BV& bv_target, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                         const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size, bool /*any*/)
    {
        std::vector<const bmx_vec*> a(src_and_size), s(src_sub_size);
        for (size_t i = 0; i < src_and_size; ++i) a[i] = bv_src_and[i]->handle();
        for (size_t i = 0; i < src_sub_size; ++i) s[i] = bv_src_sub[i]->handle();
        bmx_vec* r = nullptr; int any = 0;
        check(bmx_agg_and_sub(ctx_->handle(), a.data(), src_and_size, s.data(), src_sub_size, &r, &any));
        bv_target.adopt(r);
        return any != 0;
    }
    /// combine_and_sub_bi(bi) :450,1068 / combine_and_sub(bi, and, n, sub, n) :533,1226: the AND-SUB result as sorted
    /// positions fed into a back-insert iterator instead of a target vector (bmx_agg_and_sub_indices: the result is
    /// compacted to positions on the device, only those cross PCIe).  @return true when anything was found
    template <class BII> bool combine_and_sub_bi(BII bi)
    { return combine_and_sub(bi, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size()); }
    template <class BII, class = decltype(*std::declval<BII&>() = size_type(0))>
    bool combine_and_sub(BII bi, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                         const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size)
    {
        if (!bv_src_and || !src_and_size) return false;
        BV t(*ctx_);
        bool any = combine_and_sub(t, bv_src_and, src_and_size, bv_src_sub, src_sub_size, false);
        if (any) t.copy_to(bi);
        return any;
    }
    /// set_range_hint(from, to) :481,974 -- where results need to be searched: find_first_and_sub visits the block
    /// columns of the range only (one-block ranges are also bit-masked), combine_and_sub(pipe) honours it when the
    /// pipeline options enable search masks (is_masks(), :1312-1346).  @return true if the range is one-block bound
    bool set_range_hint(size_type from, size_type to) noexcept
    { range_set_ = true; range_from_ = from; range_to_ = to; return (from >> 16) == (to >> 16); }
    void reset_range_hint() noexcept { range_set_ = false; }                                 // :486,962
    /// set_optimization :359, set_compute_count :363, count() :488
    void set_optimization(bool opt_compress = true) { opt_compress_ = opt_compress; }
    void set_compute_count(bool count_mode) { compute_count_ = count_mode; count_ = 0; }
    size_type count() const { return count_; }

    /// combine_shift_right_and  src/bmaggregator.h:473,1089 (member form) / :552,2494 (C-style):
    /// T_0 = src[0], T_k = (T_{k-1} >> 1) & src[k]; stored with the aggregator's optimisation mode.
    /// Under set_compute_count(true) the target is left untouched and count() holds the population (:2593).
    void combine_shift_right_and(BV& bv_target)
    {
        count_ = 0;
        (void)combine_shift_right_and(bv_target, ag_.arg_bv0.data(), ag_.arg_bv0.size(), false);
    }
    bool combine_shift_right_and(BV& bv_target, const bvector_type_const_ptr* bv_src_and, size_t src_and_size, bool any)
    {
        std::vector<const bmx_vec*> h(src_and_size);
        for (size_t i = 0; i < src_and_size; ++i) h[i] = bv_src_and[i]->handle();
        if (compute_count_) {
            uint64_t c = 0;
            check(bmx_agg_shift_right_and_count(ctx_->handle(), h.data(), src_and_size, &c));
            count_ += (size_type)c;
            return count_ != 0;
        }
        bmx_vec* r = nullptr; int found = 0;
        check(bmx_agg_shift_right_and(ctx_->handle(), h.data(), src_and_size, opt_compress_ ? 1 : 0, any ? 1 : 0, &r, &found));
        bv_target.adopt(r);
        return found != 0;
    }

    /// find_first_and_sub(idx)  src/bmaggregator.h:1079 / C-style :1458
    bool find_first_and_sub(size_type& idx)
    { return find_first_and_sub(idx, ag_.arg_bv0.data(), ag_.arg_bv0.size(), ag_.arg_bv1.data(), ag_.arg_bv1.size()); }
    bool find_first_and_sub(size_type& idx, const bvector_type_const_ptr* bv_src_and, size_t src_and_size,
                            const bvector_type_const_ptr* bv_src_sub, size_t src_sub_size)
    {
        std::vector<const bmx_vec*> a(src_and_size), s(src_sub_size);
        for (size_t i = 0; i < src_and_size; ++i) a[i] = bv_src_and[i]->handle();
        for (size_t i = 0; i < src_sub_size; ++i) s[i] = bv_src_sub[i]->handle();
        int found = 0; uint64_t p = 0;
        if (range_set_) check(bmx_find_first_and_sub_range(ctx_->handle(), a.data(), src_and_size, s.data(), src_sub_size,
                                                           range_from_, range_to_, &found, &p));
        else check(bmx_find_first_and_sub(ctx_->handle(), a.data(), src_and_size, s.data(), src_sub_size, &found, &p));
        if (found) idx = p;
        return found != 0;
    }
    /// combine_and_sub(pipe)  src/bmaggregator.h:1292 (counts land in pipe.get_bv_count_vector())
    template <class TPipe>
    void combine_and_sub(TPipe& pipe)
    {
        if (!pipe.is_complete()) throw error(BMX_ERR_BADARG, "pipeline is not complete()");
        if (!pipe.size()) return;
        typedef typename TPipe::options_type opt;
        // search masks enabled + a range hint: only the block columns of the hint are visited (:1312-1346); a hint inside
        // ONE block is bit-masked as well (range_gap_blk_, :980-988) -- bmx_pipeline_run_results_hint does both
        const bool hinted = opt::is_masks() && range_set_;
        if (opt::is_make_results() || pipe.or_target_) {
            std::vector<bmx_vec*> res(pipe.size(), nullptr);
            bmx_vec* ort = nullptr;
            const bmx_vec* ort_in = (pipe.or_target_ && !pipe.or_target_->empty_handle()) ? pipe.or_target_->handle() : nullptr;
            bmx_vec** rp = opt::is_make_results() ? res.data() : nullptr;
            uint64_t* cp = (opt::is_make_results() && opt::is_compute_counts()) ? pipe.counts_.data() : nullptr;
            if (hinted) check(bmx_pipeline_run_results_hint(ctx_->handle(), pipe.h_, range_from_, range_to_, rp, cp, ort_in, pipe.or_target_ ? &ort : nullptr));
            else check(bmx_pipeline_run_results_range(ctx_->handle(), pipe.h_, 0u, 0xFFFFFFFFu, rp, cp, ort_in, pipe.or_target_ ? &ort : nullptr));
            for (size_t i = 0; i < pipe.results_.size(); ++i) delete pipe.results_[i];
            pipe.results_.assign(pipe.size(), nullptr);
            for (size_t g = 0; g < res.size(); ++g)
                if (res[g]) { pipe.results_[g] = new BV(*ctx_); pipe.results_[g]->adopt(res[g]); }
            if (pipe.or_target_) pipe.or_target_->adopt(ort);
            if (opt::is_compute_counts() && !opt::is_make_results()) {
                if (hinted) check(bmx_pipeline_run_results_hint(ctx_->handle(), pipe.h_, range_from_, range_to_, nullptr, pipe.counts_.data(), nullptr, nullptr));
                else check(bmx_pipeline_run_counts(ctx_->handle(), pipe.h_, 0u, 0xFFFFFFFFu, pipe.counts_.data()));
            }
        } else if (opt::is_compute_counts()) {
            if (hinted) check(bmx_pipeline_run_results_hint(ctx_->handle(), pipe.h_, range_from_, range_to_, nullptr, pipe.counts_.data(), nullptr, nullptr));
            else check(bmx_pipeline_run_counts(ctx_->handle(), pipe.h_, 0u, 0xFFFFFFFFu, pipe.counts_.data()));
        }
    }

private:
    context* ctx_;
    arg_groups ag_;
    bool opt_compress_ = false;          // opt_mode_ = opt_none, src/bmaggregator.h:917
    bool compute_count_ = false;
    size_type count_ = 0;
    bool range_set_ = false;             // :837-839
    size_type range_from_ = 0, range_to_ = 0;
};

} // namespace bmx
