// bmx/scanner.hpp -- bit-sliced equality search over device-resident slices: the aggregator call
// pattern of bm::sparse_vector_scanner<SV>, header-only over bmx.h.
//
//   reference                                                     here
//   ------------------------------------------------------------  ----------------------------------
//   sparse_vector_scanner::find_eq(sv, value, bv_out)             slice_scanner::find_eq(value, bv_out)
//        src/bmsparsevec_algo.h:1083,4356 -> find_eq_with_nulls :2387-2412
//   sparse_vector_scanner::find_eq(sv, value, pos) :1111,4434     slice_scanner::find_first_eq(value, idx)
//        -> find_first_eq :2417
//   prepare_and_sub_aggregator(sv, value) :2593-2640              slice_scanner::add_groups()
//        AND group = slices of the set bits of the value (high bit first),
//        SUB group = every other existing slice below effective_slices(); a set bit
//        without a slice => nothing can match
//   pipeline of many searches (batch of arg-groups, :3236,3408)   slice_scanner::find_eq_counts(values, n, counts)
//   find_gt / find_ge / find_lt / find_le / find_range             slice_scanner::find_gt(value, bv_out) ...
//        :1135-1174 -> :2690-2880, find_gt_horizontal_u :2914         ONE pass over the planes (bmx_slice_compare,
//   find_zero :2290, find_nonzero :4464, find_eq(sv, 0, ..) :4366     bmx_kernels4.h) instead of a chain of vector ops
//
// The slices stay resident in HBM; a batch of searches is ONE counts-only pipeline launch (the
// LDS-staged kernel when many groups share the slices).  The container
// (bm::sparse_vector<>) stays on the host: bmx::upload_slices (bm_adapter.hpp) uploads its slices.
#pragma once

#include <vector>

#include "bvector.hpp"

namespace bmx {

/// what the scanner needs from a vector family: its context type and the one-pass comparison entry of the C-ABI
template <class BV> struct scanner_traits;
template <> struct scanner_traits<bvector> {
    typedef context ctx_type;
    typedef bmx_vec handle_type;
    static int compare(ctx_type& c, const handle_type* const* h, size_t n, int pred, uint64_t v0, uint64_t v1, uint64_t size,
                       const handle_type* nn, handle_type** r, uint64_t* cnt)
    { return bmx_slice_compare(c.handle(), h, n, pred, v0, v1, size, nn, r, cnt); }
    static int eq_counts(ctx_type& c, const handle_type* const* h, size_t n, const uint64_t* values, size_t nv, uint64_t size,
                         const handle_type* nn, uint64_t* counts)
    { return bmx_slice_eq_counts(c.handle(), h, n, values, nv, size, nn, counts); }
};

/// BV = bmx::bvector (one device, `slice_scanner`) or bmx::gbvector (a device group, `gslice_scanner` in group.hpp:
/// planes sharded by block range, every member searches its own rows)
template <class BV>
class basic_slice_scanner {
    typedef scanner_traits<BV> traits;
    typedef typename traits::ctx_type ctx_type;
    typedef BV bvector;                     // (the method bodies below are written against this name)
public:
    explicit basic_slice_scanner(ctx_type& ctx) : ctx_(&ctx), agg_(ctx) {}

    /// slice i holds bit i of every element; nullptr = the plane does not exist (sv.get_slice(i) == 0).
    /// slices.size() plays effective_slices(); size = sv.size() (rows; 0 = the longest slice);
    /// not_null = device copy of sv.get_null_bvector() (nullptr: the container has no NULLs).
    void bind(const std::vector<const bvector*>& slices, size_type size = 0, const bvector* not_null = nullptr)
    {
        slices_ = slices; size_ = size; not_null_ = not_null;
        if (!size_) for (size_t i = 0; i < slices_.size(); ++i) if (slices_[i] && slices_[i]->size() > size_) size_ = slices_[i]->size();
    }
    size_t effective_slices() const noexcept { return slices_.size(); }
    size_type size() const noexcept { return size_; }

    // ---- comparison searches (unsigned values): one pass over the planes ----
    void find_gt(uint64_t value, bvector& bv_out) { compare(BMX_CMP_GT, value, 0, &bv_out); }            // :2690
    void find_ge(uint64_t value, bvector& bv_out) { compare(BMX_CMP_GE, value, 0, &bv_out); }            // :2717
    void find_lt(uint64_t value, bvector& bv_out) { compare(BMX_CMP_LT, value, 0, &bv_out); }            // :2790
    void find_le(uint64_t value, bvector& bv_out) { compare(BMX_CMP_LE, value, 0, &bv_out); }            // :2824
    void find_range(uint64_t from, uint64_t to, bvector& bv_out) { compare(BMX_CMP_RANGE, from, to, &bv_out); }   // :2862
    void find_zero(bvector& bv_out) { compare(BMX_CMP_ZERO, 0, 0, &bv_out); }                            // :2290 (null_correct)
    void find_nonzero(bvector& bv_out) { compare(BMX_CMP_NONZERO, 0, 0, &bv_out); }                      // :4464
    /// popcount of a comparison search, nothing materialised (pred = BMX_CMP_*)
    size_type count(int pred, uint64_t v0, uint64_t v1 = 0) { return compare(pred, v0, v1, nullptr); }

    /// rows equal to `value` -> bv_out; false when nothing was found (value 0: find_zero, :4366)
    bool find_eq(uint64_t value, bvector& bv_out)
    {
        if (!value) { compare(BMX_CMP_EQ, 0, 0, &bv_out); return bv_out.any(); }
        typename aggregator<bvector>::arg_groups g;
        if (!add_groups(value, g)) { bv_out.clear(); return false; }
        return agg_.combine_and_sub(bv_out, g.arg_bv0.data(), g.arg_bv0.size(), g.arg_bv1.data(), g.arg_bv1.size(), false);
    }

    /// find_eq(sv, value, bi)  src/bmsparsevec_algo.h:1096: the matching rows as sorted indices fed into a back-insert iterator
    template <class BII, class = decltype(*std::declval<BII&>() = size_type(0))>
    bool find_eq(uint64_t value, BII bi)
    {
        bvector t(*ctx_);
        bool f = find_eq(value, t);
        if (f) t.copy_to(bi);
        return f;
    }

    /// index of the first row equal to `value`
    bool find_first_eq(uint64_t value, size_type& idx)
    {
        if (!value) return false;                                        // :2428
        typename aggregator<bvector>::arg_groups g;
        if (!add_groups(value, g)) return false;
        return agg_.find_first_and_sub(idx, g.arg_bv0.data(), g.arg_bv0.size(), g.arg_bv1.data(), g.arg_bv1.size());
    }

    /// counts[q] = number of rows equal to values[q].  Up to 32 planes: ONE pass over the planes whatever n is
    /// (bmx_slice_eq_counts: bit-matrix transposition + hash lookup); more planes, or use_pipeline: one AND-SUB group
    /// per value in one counts-only pipeline (the reference's formulation, :3236,3408)
    void find_eq_counts(const uint64_t* values, size_t n, uint64_t* counts, bool use_pipeline = false)
    {
        if (!use_pipeline && slices_.size() <= 32) {
            typedef typename traits::handle_type handle_type;
            std::vector<const handle_type*> h(slices_.size() ? slices_.size() : 1, nullptr);
            for (size_t i = 0; i < slices_.size(); ++i) h[i] = slices_[i] ? slices_[i]->handle() : nullptr;
            check(traits::eq_counts(*ctx_, h.data(), slices_.size(), values, n, size_,
                                    (not_null_ && !not_null_->empty_handle()) ? not_null_->handle() : nullptr, counts));
            return;
        }
        typedef typename aggregator<bvector>::template pipeline<agg_opt_only_counts> pipe_t;
        pipe_t pipe(*ctx_);
        std::vector<size_t> slot(n, ~size_t(0));
        for (size_t q = 0; q < n; ++q) {
            counts[q] = 0;
            if (!values[q]) { counts[q] = count(BMX_CMP_EQ, 0); continue; }
            typename aggregator<bvector>::arg_groups g;
            if (!add_groups(values[q], g)) continue;                     // impossible value: count 0
            typename aggregator<bvector>::arg_groups* pg = pipe.add();
            *pg = g;
            slot[q] = pipe.size() - 1;
        }
        if (!pipe.size()) return;
        pipe.complete();
        agg_.combine_and_sub(pipe);
        for (size_t q = 0; q < n; ++q) if (slot[q] != ~size_t(0)) counts[q] = pipe.get_bv_count_vector()[slot[q]];
    }

private:
    size_type compare(int pred, uint64_t v0, uint64_t v1, bvector* out)
    {
        typedef typename traits::handle_type handle_type;
        std::vector<const handle_type*> h(slices_.size() ? slices_.size() : 1, nullptr);
        for (size_t i = 0; i < slices_.size(); ++i) h[i] = slices_[i] ? slices_[i]->handle() : nullptr;
        handle_type* r = nullptr; uint64_t c = 0;
        check(traits::compare(*ctx_, h.data(), slices_.size(), pred, v0, v1, size_,
                              (not_null_ && !not_null_->empty_handle()) ? not_null_->handle() : nullptr,
                              out ? &r : nullptr, out ? nullptr : &c));
        if (out) out->adopt(r);
        return c;
    }
    // prepare_and_sub_aggregator (src/bmsparsevec_algo.h:2593-2640)
    bool add_groups(uint64_t value, typename aggregator<bvector>::arg_groups& g) const
    {
        for (int bit = 63; bit >= 0; --bit) {                            // backward order (:2614)
            if (!((value >> bit) & 1u)) continue;
            if ((size_t)bit >= slices_.size() || !slices_[(size_t)bit]) return false;      // :2621
            g.add(slices_[(size_t)bit], 0);
        }
        for (size_t i = 0; i < slices_.size(); ++i)
            if (slices_[i] && (i >= 64 || !((value >> i) & 1u))) g.add(slices_[i], 1);  // :2626-2631
        return true;
    }

    ctx_type* ctx_;
    aggregator<bvector> agg_;
    std::vector<const bvector*> slices_;
    size_type size_ = 0;
    const bvector* not_null_ = nullptr;
};

typedef basic_slice_scanner<bvector> slice_scanner;

} // namespace bmx
