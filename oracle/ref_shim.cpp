// ref_shim.cpp -- thin extern "C" shim over the UNMODIFIED reference headers.
//
// TEST INFRASTRUCTURE ONLY.  Compiled from the sources where they lie
// (-I/root/reference/src) by oracle/Makefile into oracle/_ref/ (git-ignored),
// once scalar and once with -DBMAVX2OPT.  Used to (1) validate the C
// restatement in bmx_oracle.c, (2) generate tests/golden/, (3) serve as the
// "reference" CPU baseline in bench.py.  No reference source is copied here:
// this file only calls the reference's public API.
//
// The entry points mirror bmx_oracle.h one-for-one (prefix ref_ instead of
// bmo_) so the tests can run the same checks against either implementation.
#include <cstdint>
#include <cstring>
#include <vector>

#include "bm.h"
#include "bmaggregator.h"
#include "bmalgo.h"
#include "bmbvimport.h"
#include "bmsparsevec.h"
#include "bmsparsevec_algo.h"
#include "bmserial.h"
#include <chrono>

typedef bm::bvector<> bvect;
typedef bm::aggregator<bvect> agg_t;

namespace {

enum { K_NULL = 0, K_FULL = 1, K_BIT = 2, K_GAP = 3 };

inline const bm::word_t* block_ptr(const bvect& bv, unsigned nb)
{
    const bvect::blocks_manager_type& bman = bv.get_blocks_manager();
    if (!bman.is_init()) return 0;
    unsigned i = nb >> bm::set_array_shift, j = nb & bm::set_array_mask;
    if (i >= bman.top_block_size()) return 0;
    return bman.get_block_ptr(i, j);
}

inline int block_kind(const bm::word_t* p)
{
    if (!p) return K_NULL;
    if (p == FULL_BLOCK_FAKE_ADDR || p == FULL_BLOCK_REAL_ADDR) return K_FULL;
    if (BM_IS_GAP(p)) return K_GAP;
    return K_BIT;
}

} // namespace

#pragma GCC visibility push(default)
extern "C" {

int ref_simd_version() { return bm::simd_version(); }

void* ref_vec_new() { return new bvect(); }

void* ref_vec_import(const uint32_t* words, uint64_t nwords, int optimize)
{
    bvect* bv = new bvect();
    bm::bit_import_u32(*bv, words, bvect::size_type(nwords), optimize != 0);
    return bv;
}

void ref_vec_free(void* v) { delete static_cast<bvect*>(v); }

void ref_vec_set_bit(void* v, uint64_t n) { static_cast<bvect*>(v)->set(bvect::size_type(n)); }
void ref_vec_set_range(void* v, uint64_t l, uint64_t r)
{ static_cast<bvect*>(v)->set_range(bvect::size_type(l), bvect::size_type(r)); }
void ref_vec_optimize(void* v)
{
    BM_DECLARE_TEMP_BLOCK(tb)
    static_cast<bvect*>(v)->optimize(tb, bvect::opt_compress);
}
int ref_vec_get_bit(void* v, uint64_t n) { return static_cast<bvect*>(v)->test(bvect::size_type(n)); }

void ref_vec_stat(void* v, uint32_t nblocks, uint32_t counts[4], uint64_t* gap_words)
{
    const bvect& bv = *static_cast<bvect*>(v);
    counts[0] = counts[1] = counts[2] = counts[3] = 0; *gap_words = 0;
    for (unsigned nb = 0; nb < nblocks; ++nb) {
        const bm::word_t* p = block_ptr(bv, nb);
        int k = block_kind(p);
        counts[k]++;
        if (k == K_GAP) *gap_words += (BMGAP_PTR(p)[0] >> 3) + 1;
    }
}

void ref_vec_flatten(void* v, uint32_t nblocks, uint8_t* kinds, uint32_t* offs,
                     uint32_t* bit_slab, uint16_t* gap_slab)
{
    const bvect& bv = *static_cast<bvect*>(v);
    uint32_t nbit = 0; uint64_t ngap = 0;
    for (unsigned nb = 0; nb < nblocks; ++nb) {
        const bm::word_t* p = block_ptr(bv, nb);
        int k = block_kind(p);
        kinds[nb] = uint8_t(k); offs[nb] = 0;
        if (k == K_BIT) {
            std::memcpy(bit_slab + uint64_t(nbit) * bm::set_block_size, p, bm::set_block_size * 4);
            offs[nb] = nbit++;
        } else if (k == K_GAP) {
            const bm::gap_word_t* g = BMGAP_PTR(p);
            unsigned n = (g[0] >> 3) + 1;
            std::memcpy(gap_slab + ngap, g, n * 2);
            offs[nb] = uint32_t(ngap); ngap += n;
        }
    }
}

void ref_vec_to_words(void* v, uint32_t* out, uint64_t nwords)
{
    const bvect& bv = *static_cast<bvect*>(v);
    BM_DECLARE_TEMP_BLOCK(tb)
    uint64_t nblocks = (nwords + bm::set_block_size - 1) / bm::set_block_size;
    for (uint64_t nb = 0; nb < nblocks; ++nb) {
        const bm::word_t* p = block_ptr(bv, unsigned(nb));
        int k = block_kind(p);
        uint64_t off = nb * bm::set_block_size;
        uint64_t n = nwords - off < bm::set_block_size ? nwords - off : bm::set_block_size;
        if (k == K_NULL) std::memset(out + off, 0, n * 4);
        else if (k == K_FULL) std::memset(out + off, 0xFF, n * 4);
        else if (k == K_BIT) std::memcpy(out + off, p, n * 4);
        else { bm::gap_convert_to_bitset(tb.begin(), BMGAP_PTR(p)); std::memcpy(out + off, tb.begin(), n * 4); }
    }
}

uint64_t ref_vec_count(void* v) { return static_cast<bvect*>(v)->count(); }

int ref_vec_equal(void* a, void* b)
{ return static_cast<bvect*>(a)->compare(*static_cast<bvect*>(b)) == 0; }

// 3-operand pairwise ops: bm.h:6185 (and), :5973 (or), :6072 (xor), :6403 (sub)
void* ref_op2(int op, void* a, void* b, int opt_compress)
{
    bvect* t = new bvect();
    const bvect& x = *static_cast<bvect*>(a);
    const bvect& y = *static_cast<bvect*>(b);
    bvect::optmode om = opt_compress ? bvect::opt_compress : bvect::opt_none;
    switch (op) {
    case 0: t->bit_and(x, y, om); break;
    case 1: t->bit_or(x, y, om); break;
    case 2: t->bit_xor(x, y, om); break;
    default: t->bit_sub(x, y, om); break;
    }
    return t;
}

// bmalgo.h:49 count_and, :149 count_or, :81 count_xor, :115 count_sub
uint64_t ref_count_op2(int op, void* a, void* b)
{
    const bvect& x = *static_cast<bvect*>(a);
    const bvect& y = *static_cast<bvect*>(b);
    switch (op) {
    case 0: return bm::count_and(x, y);
    case 1: return bm::count_or(x, y);
    case 2: return bm::count_xor(x, y);
    default: return bm::count_sub(x, y);
    }
}

// aggregator: bmaggregator.h:1101 combine_or, :1162 combine_and_sub
void* ref_agg_or(void* const* src, size_t n)
{
    bvect* t = new bvect();
    agg_t agg;
    agg.combine_or(*t, reinterpret_cast<const bvect* const*>(src), n);
    return t;
}

void* ref_agg_or_opt(void* const* src, size_t n, int opt_compress)
{
    bvect* t = new bvect();
    agg_t agg;
    agg.set_optimization(opt_compress ? bvect::opt_compress : bvect::opt_none);
    agg.combine_or(*t, reinterpret_cast<const bvect* const*>(src), n);
    return t;
}

void* ref_agg_and_sub(void* const* src_and, size_t n_and, void* const* src_sub, size_t n_sub)
{
    bvect* t = new bvect();
    agg_t agg;
    agg.combine_and_sub(*t, reinterpret_cast<const bvect* const*>(src_and), n_and,
                        reinterpret_cast<const bvect* const*>(src_sub), n_sub, false);
    return t;
}

// aggregator::combine_shift_right_and (bmaggregator.h:552,2494); count form via set_compute_count (:363)
void* ref_agg_shift_right_and(void* const* src, size_t n, int opt_compress, int any, int* found)
{
    bvect* t = new bvect();
    agg_t agg;
    agg.set_optimization(opt_compress ? bvect::opt_compress : bvect::opt_none);
    bool f = agg.combine_shift_right_and(*t, reinterpret_cast<const bvect* const*>(src), n, any != 0);
    if (found) *found = f ? 1 : 0;
    return t;
}

uint64_t ref_agg_shift_right_and_count(void* const* src, size_t n)
{
    bvect t;
    agg_t agg;
    agg.set_compute_count(true);
    (void)agg.combine_shift_right_and(t, reinterpret_cast<const bvect* const*>(src), n, false);
    return agg.count();
}

// member-style API: add() + combine_and / combine_or  (bmaggregator.h:1013,1021,1030)
void* ref_agg_member(int kind, void* const* src, size_t n)
{
    bvect* t = new bvect();
    agg_t agg;
    for (size_t i = 0; i < n; ++i) agg.add(static_cast<const bvect*>(src[i]));
    if (kind == 0) agg.combine_and(*t); else agg.combine_or(*t);
    return t;
}

// counts-only pipeline: bmaggregator.h:1292 with agg_opt_only_counts (:62-103).
// The block-range arguments of the oracle twin are not supported by the
// reference API; nb_from/nb_to must cover everything (0, UINT32_MAX).
void ref_agg_pipeline_counts(void* const* and_list, const uint32_t* and_n,
                             void* const* sub_list, const uint32_t* sub_n,
                             size_t ngroups, uint32_t /*nb_from*/, uint32_t /*nb_to*/,
                             uint64_t* counts_out)
{
    agg_t agg;
    agg_t::pipeline<bm::agg_opt_only_counts> pipe;
    size_t ao = 0, so = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        agg_t::arg_groups* ag = pipe.add();
        for (uint32_t k = 0; k < and_n[g]; ++k) ag->add(static_cast<const bvect*>(and_list[ao + k]), 0);
        for (uint32_t k = 0; k < sub_n[g]; ++k) ag->add(static_cast<const bvect*>(sub_list[so + k]), 1);
        ao += and_n[g]; so += sub_n[g];
    }
    pipe.complete();
    agg.combine_and_sub(pipe);
    auto& cnt = pipe.get_bv_count_vector();
    for (size_t g = 0; g < ngroups; ++g) counts_out[g] = cnt[g];
}

// the same with pipeline::set_search_count_limit (bmaggregator.h:255-261; honoured per block at :1361-1367: a group whose
// count has reached the limit is not evaluated on the blocks that follow)
void ref_agg_pipeline_counts_limit(void* const* and_list, const uint32_t* and_n,
                                   void* const* sub_list, const uint32_t* sub_n,
                                   size_t ngroups, uint64_t limit, uint64_t* counts_out)
{
    agg_t agg;
    agg_t::pipeline<bm::agg_opt_only_counts> pipe;
    pipe.set_search_count_limit(bvect::size_type(limit));
    size_t ao = 0, so = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        agg_t::arg_groups* ag = pipe.add();
        for (uint32_t k = 0; k < and_n[g]; ++k) ag->add(static_cast<const bvect*>(and_list[ao + k]), 0);
        for (uint32_t k = 0; k < sub_n[g]; ++k) ag->add(static_cast<const bvect*>(sub_list[so + k]), 1);
        ao += and_n[g]; so += sub_n[g];
    }
    pipe.complete();
    agg.combine_and_sub(pipe);
    auto& cnt = pipe.get_bv_count_vector();
    for (size_t g = 0; g < ngroups; ++g) counts_out[g] = cnt[g];
}

// full pipeline: pipeline<agg_opt_bvect_and_counts> with an OR target (bmaggregator.h:62-103,222-341,1292-1449)
// results_out[g] = new bvector (caller frees with ref_vec_free) or NULL when the group found nothing;
// *or_target_out = new bvector holding the OR of all group results.
void ref_agg_pipeline_results(void* const* and_list, const uint32_t* and_n,
                              void* const* sub_list, const uint32_t* sub_n, size_t ngroups,
                              void** results_out, uint64_t* counts_out, void** or_target_out)
{
    agg_t agg;
    agg_t::pipeline<bm::agg_opt_bvect_and_counts> pipe;
    bvect* ort = new bvect();
    pipe.set_or_target(ort);
    size_t ao = 0, so = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        agg_t::arg_groups* ag = pipe.add();
        for (uint32_t k = 0; k < and_n[g]; ++k) ag->add(static_cast<const bvect*>(and_list[ao + k]), 0);
        for (uint32_t k = 0; k < sub_n[g]; ++k) ag->add(static_cast<const bvect*>(sub_list[so + k]), 1);
        ao += and_n[g]; so += sub_n[g];
    }
    pipe.complete();
    agg.combine_and_sub(pipe);
    auto& res = pipe.get_bv_res_vector();
    auto& cnt = pipe.get_bv_count_vector();
    for (size_t g = 0; g < ngroups; ++g) {
        counts_out[g] = cnt[g];
        results_out[g] = res[g] ? new bvect(*res[g]) : nullptr;
    }
    *or_target_out = ort;
}

// rank / select: bm.h:2531 build_rs_index, :3120 count_to, :5350 select
struct ref_rs { bvect::rs_index_type rs; };

void* ref_rs_build(void* v)
{
    ref_rs* r = new ref_rs();
    static_cast<bvect*>(v)->build_rs_index(&r->rs);
    return r;
}
void ref_rs_free(void* r) { delete static_cast<ref_rs*>(r); }
uint64_t ref_rs_count(void* r) { return static_cast<ref_rs*>(r)->rs.count(); }
uint32_t ref_rs_total_blocks(void* r) { return uint32_t(static_cast<ref_rs*>(r)->rs.get_total()); }
void ref_rs_export(void* r, uint32_t nblocks, uint32_t* bcount, uint64_t* sub_count)
{
    const bvect::rs_index_type& rs = static_cast<ref_rs*>(r)->rs;
    for (uint32_t nb = 0; nb < nblocks; ++nb) { bcount[nb] = rs.count(nb); sub_count[nb] = rs.sub_count(nb); }
}
uint64_t ref_rank(void* v, void* r, uint64_t n)
{ return static_cast<bvect*>(v)->count_to(bvect::size_type(n), static_cast<ref_rs*>(r)->rs); }
int ref_select(void* v, void* r, uint64_t rank, uint64_t* pos)
{
    bvect::size_type p = 0;
    bool f = static_cast<bvect*>(v)->select(bvect::size_type(rank), p, static_cast<ref_rs*>(r)->rs);
    *pos = p; return f;
}
void ref_rank_batch(void* v, void* r, const uint64_t* n, size_t q, uint64_t* out)
{
    const bvect& bv = *static_cast<bvect*>(v); const bvect::rs_index_type& rs = static_cast<ref_rs*>(r)->rs;
    for (size_t i = 0; i < q; ++i) out[i] = bv.count_to(bvect::size_type(n[i]), rs);
}
void ref_select_batch(void* v, void* r, const uint64_t* rk, size_t q, uint64_t* pos, uint8_t* found)
{
    const bvect& bv = *static_cast<bvect*>(v); const bvect::rs_index_type& rs = static_cast<ref_rs*>(r)->rs;
    for (size_t i = 0; i < q; ++i) {
        bvect::size_type p = 0;
        found[i] = bv.select(bvect::size_type(rk[i]), p, rs); pos[i] = p;
    }
}

// bvector::find (first set bit), aggregator::find_first_and_sub (bmaggregator.h:1458)
int ref_vec_find_first(void* v, uint64_t* pos)
{
    bvect::size_type p = 0; bool f = static_cast<bvect*>(v)->find(p); *pos = p; return f;
}
int ref_find_first_and_sub(void* const* src_and, size_t n_and, void* const* src_sub, size_t n_sub, uint64_t* idx)
{
    agg_t agg; bvect::size_type i = 0;
    bool f = agg.find_first_and_sub(i, reinterpret_cast<const bvect* const*>(src_and), n_and,
                                    reinterpret_cast<const bvect* const*>(src_sub), n_sub);
    *idx = i; return f;
}
// the same under aggregator::set_range_hint(from, to) (bmaggregator.h:481,974): *hint_ok = what set_range_hint returned
int ref_find_first_and_sub_range(void* const* src_and, size_t n_and, void* const* src_sub, size_t n_sub,
                                 uint64_t from, uint64_t to, uint64_t* idx, int* hint_ok)
{
    agg_t agg; bvect::size_type i = 0;
    for (size_t k = 0; k < n_and; ++k) agg.add(static_cast<const bvect*>(src_and[k]), 0);
    for (size_t k = 0; k < n_sub; ++k) agg.add(static_cast<const bvect*>(src_sub[k]), 1);
    bool ok = agg.set_range_hint(bvect::size_type(from), bvect::size_type(to));
    if (hint_ok) *hint_ok = ok;
    bool f = agg.find_first_and_sub(i);
    *idx = i; return f;
}

// pipeline whose options enable search masks (agg_run_options<.., .., true>::is_masks(), bmaggregator.h:65,78) run under
// set_range_hint(from, to): result vectors + counts (:1312-1346)
void ref_agg_pipeline_masks(void* const* and_list, const uint32_t* and_n, void* const* sub_list, const uint32_t* sub_n,
                            size_t ngroups, uint64_t from, uint64_t to, void** results_out, uint64_t* counts_out)
{
    typedef bm::agg_run_options<true, true, true> opt_t;
    agg_t agg;
    agg_t::pipeline<opt_t> pipe;
    size_t ao = 0, so = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        agg_t::arg_groups* ag = pipe.add();
        for (uint32_t k = 0; k < and_n[g]; ++k) ag->add(static_cast<const bvect*>(and_list[ao + k]), 0);
        for (uint32_t k = 0; k < sub_n[g]; ++k) ag->add(static_cast<const bvect*>(sub_list[so + k]), 1);
        ao += and_n[g]; so += sub_n[g];
    }
    pipe.complete();
    agg.set_range_hint(bvect::size_type(from), bvect::size_type(to));
    agg.combine_and_sub(pipe);
    auto& res = pipe.get_bv_res_vector();
    auto& cnt = pipe.get_bv_count_vector();
    for (size_t g = 0; g < ngroups; ++g) {
        counts_out[g] = cnt[g];
        results_out[g] = res[g] ? new bvect(*res[g]) : nullptr;
    }
}

// ---- bm::sparse_vector<unsigned, bvector<>> + bm::sparse_vector_scanner<> (bmsparsevec_algo.h:931, 1083-1174, 2290,
// 2690-2880, 4464): the caller of the aggregator that SURVEY section 8(f)-1 names.  values[i] is stored at row i;
// is_null (may be NULL) marks rows left unassigned in a use_null vector. ----
typedef bm::sparse_vector<unsigned, bvect> svect_t;

void* ref_sv_new(const uint32_t* values, const uint8_t* is_null, uint64_t n)
{
    svect_t* sv = is_null ? new svect_t(bm::use_null) : new svect_t();
    if (is_null) {
        for (uint64_t i = 0; i < n; ++i) if (!is_null[i]) sv->set(svect_t::size_type(i), values[i]);
        if (n && is_null[n - 1]) sv->set_null(svect_t::size_type(n - 1));          // make size() == n
    } else {
        svect_t::back_insert_iterator bi = sv->get_back_inserter();
        for (uint64_t i = 0; i < n; ++i) bi = values[i];
        bi.flush();
    }
    sv->optimize();
    return sv;
}
void ref_sv_free(void* sv) { delete static_cast<svect_t*>(sv); }
uint64_t ref_sv_size(void* sv) { return static_cast<svect_t*>(sv)->size(); }
uint32_t ref_sv_effective_slices(void* sv) { return static_cast<svect_t*>(sv)->effective_slices(); }
// copy of bit-plane i (NULL when the plane does not exist); caller frees with ref_vec_free
void* ref_sv_slice(void* sv, uint32_t i)
{
    const bvect* p = static_cast<svect_t*>(sv)->get_slice(i);
    return p ? new bvect(*p) : nullptr;
}
void* ref_sv_not_null(void* sv)
{
    const bvect* p = static_cast<svect_t*>(sv)->get_null_bvector();
    return p ? new bvect(*p) : nullptr;
}
// pred: BMX_CMP_* of include/bmx.h (GT 0, GE 1, LT 2, LE 3, RANGE 4, EQ 5, ZERO 6, NONZERO 7)
void* ref_sv_compare(void* svp, int pred, uint32_t v0, uint32_t v1)
{
    const svect_t& sv = *static_cast<svect_t*>(svp);
    bm::sparse_vector_scanner<svect_t> sc;
    bvect* r = new bvect();
    switch (pred) {
    case 0: sc.find_gt(sv, v0, *r); break;
    case 1: sc.find_ge(sv, v0, *r); break;
    case 2: sc.find_lt(sv, v0, *r); break;
    case 3: sc.find_le(sv, v0, *r); break;
    case 4: sc.find_range(sv, v0, v1, *r); break;
    case 5: sc.find_eq(sv, v0, *r); break;
    case 6: sc.find_zero(sv, *r); break;
    default: sc.find_nonzero(sv, *r); break;
    }
    return r;
}
int ref_sv_find_first_eq(void* svp, uint32_t v, uint64_t* pos)
{
    const svect_t& sv = *static_cast<svect_t*>(svp);
    bm::sparse_vector_scanner<svect_t> sc;
    svect_t::size_type p = 0;
    bool f = sc.find_eq(sv, v, p);
    *pos = p; return f;
}

// ---- signed containers: bm::sparse_vector<int, bvector<>> (sign plane 0 + magnitude planes, s2u encoding) ----
typedef bm::sparse_vector<int, bvect> svects_t;

void* ref_svs_new(const int32_t* values, const uint8_t* is_null, uint64_t n)
{
    svects_t* sv = is_null ? new svects_t(bm::use_null) : new svects_t();
    if (is_null) {
        for (uint64_t i = 0; i < n; ++i) if (!is_null[i]) sv->set(svects_t::size_type(i), values[i]);
        if (n && is_null[n - 1]) sv->set_null(svects_t::size_type(n - 1));
    } else {
        svects_t::back_insert_iterator bi = sv->get_back_inserter();
        for (uint64_t i = 0; i < n; ++i) bi = values[i];
        bi.flush();
    }
    sv->optimize();
    return sv;
}
void ref_svs_free(void* sv) { delete static_cast<svects_t*>(sv); }
uint64_t ref_svs_size(void* sv) { return static_cast<svects_t*>(sv)->size(); }
uint32_t ref_svs_effective_slices(void* sv) { return static_cast<svects_t*>(sv)->effective_slices(); }
void* ref_svs_slice(void* sv, uint32_t i)
{
    const bvect* p = static_cast<svects_t*>(sv)->get_slice(i);
    return p ? new bvect(*p) : nullptr;
}
void* ref_svs_compare(void* svp, int pred, int32_t v0, int32_t v1)
{
    const svects_t& sv = *static_cast<svects_t*>(svp);
    bm::sparse_vector_scanner<svects_t> sc;
    bvect* r = new bvect();
    switch (pred) {
    case 0: sc.find_gt(sv, v0, *r); break;
    case 1: sc.find_ge(sv, v0, *r); break;
    case 2: sc.find_lt(sv, v0, *r); break;
    case 3: sc.find_le(sv, v0, *r); break;
    case 4: sc.find_range(sv, v0, v1, *r); break;
    case 5: sc.find_eq(sv, v0, *r); break;
    case 6: sc.find_zero(sv, *r); break;
    default: sc.find_nonzero(sv, *r); break;
    }
    return r;
}
// scanner leftovers on the unsigned container: IN-list find_eq(sv, start, end, bv_out) (bmsparsevec_algo.h:1399) and invert (:2321)
void* ref_sv_find_eq_in(void* svp, const uint32_t* values, size_t n)
{
    const svect_t& sv = *static_cast<svect_t*>(svp);
    bm::sparse_vector_scanner<svect_t> sc;
    bvect* r = new bvect();
    sc.find_eq(sv, values, values + n, *r);
    return r;
}
void* ref_sv_invert(void* svp, void* bv)
{
    const svect_t& sv = *static_cast<svect_t*>(svp);
    bm::sparse_vector_scanner<svect_t> sc;
    bvect* r = new bvect(*static_cast<bvect*>(bv));
    sc.invert(sv, *r);
    return r;
}

// rank variants: bm.h:3548 count_range, :3229 rank_corrected, :3173 count_to_test, :5279 find_rank
uint64_t ref_count_range(void* v, void* r, uint64_t left, uint64_t right)
{ return static_cast<bvect*>(v)->count_range(bvect::size_type(left), bvect::size_type(right), static_cast<ref_rs*>(r)->rs); }
uint64_t ref_rank_corrected(void* v, void* r, uint64_t n)
{ return static_cast<bvect*>(v)->rank_corrected(bvect::size_type(n), static_cast<ref_rs*>(r)->rs); }
uint64_t ref_count_to_test(void* v, void* r, uint64_t n)
{ return static_cast<bvect*>(v)->count_to_test(bvect::size_type(n), static_cast<ref_rs*>(r)->rs); }
int ref_find_rank(void* v, void* r, uint64_t rank, uint64_t from, uint64_t* pos)
{
    bvect::size_type p = 0;
    bool f = static_cast<bvect*>(v)->find_rank(bvect::size_type(rank), bvect::size_type(from), p, static_cast<ref_rs*>(r)->rs);
    *pos = p; return f;
}

// block-level known-answer helpers (bmfunc.h) used by the golden generator
uint32_t ref_bit_block_count(const uint32_t* blk) { return bm::bit_block_count(blk); }
uint64_t ref_calc_block_digest0(const uint32_t* blk) { return bm::calc_block_digest0(blk); }
uint32_t ref_bit_block_calc_change(const uint32_t* blk) { return bm::bit_block_calc_change(blk); }
unsigned ref_bit_to_gap(uint16_t* dest, const uint32_t* blk)
{ return bm::bit_to_gap(dest, blk, bm::gap_equiv_len * 2); }
void ref_gap_convert_to_bitset(uint32_t* dest, const uint16_t* gap) { bm::gap_convert_to_bitset(dest, gap); }
uint32_t ref_gap_bit_count(const uint16_t* gap) { return bm::gap_bit_count_unr(gap); }

// ---- serialisation (src/bmserial.h) -- only to MEASURE what SURVEY section 8(f)-3's second leg would be about: the size of a
// BLOB next to the vector's own blocks and how fast the reference's own deserializer turns it back into blocks on a host core
// (tools/cpu_blob_decode_rate.py).  Nothing in the product path touches BLOBs.
// level: serializer::set_compression_level (5 = the default: BIC + digests).  Returns the BLOB size; the bytes go to *out
// when cap is large enough.
uint64_t ref_serialize(void* v, unsigned level, unsigned char* out, uint64_t cap)
{
    bm::serializer<bvect> ser;
    ser.set_compression_level(level);
    bm::serializer<bvect>::buffer buf;
    ser.serialize(*static_cast<bvect*>(v), buf, 0);
    if (out && cap >= buf.size()) memcpy(out, buf.data(), buf.size());
    return buf.size();
}

// deserialises `reps` times into fresh vectors; returns the last one, *best_seconds = the fastest pass
void* ref_deserialize_timed(const unsigned char* blob, unsigned reps, double* best_seconds)
{
    bvect* last = nullptr; double best = 1e30;
    for (unsigned r = 0; r < (reps ? reps : 1u); ++r) {
        delete last;
        last = new bvect();
        auto t0 = std::chrono::steady_clock::now();
        bm::deserialize(*last, blob);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
    }
    if (best_seconds) *best_seconds = best;
    return last;
}

} // extern "C"
#pragma GCC visibility pop
