// GPU test of the header-only C++ facade (include/bmx/bvector.hpp) against the C oracle.
// Built by tests/cpp/Makefile, run by tests/test_cpp_facade.py (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bmx/bvector.hpp"
extern "C" {
#include "../../oracle/bmx_oracle.h"
}

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static const uint64_t SEED = 0xB17A61C;

int main()
{
    bmx::context ctx(0);
    const uint64_t nbits = 11 * 65536 + 777;
    const unsigned NV = 10;
    std::vector<std::vector<uint32_t>> words(NV);
    std::vector<bmo_vec*> pv(NV);
    std::vector<bmx::bvector> gv;
    for (unsigned v = 0; v < NV; ++v) {
        uint32_t dq = v < 4 ? 6554u : (v < 7 ? 655u : 120u);          // bit, mixed, GAP operands
        uint64_t nw = ((nbits + 63) / 64) * 2;
        words[v].resize(nw);
        bmo_gen_words(SEED, v, v < 4, dq, nbits, 0, nw, words[v].data());
        pv[v] = bmo_vec_import(words[v].data(), nw, 1);
        gv.emplace_back(ctx);
        bmx::bit_import_u32(gv[v], words[v].data(), nw, true);
        REQUIRE(gv[v].count() == bmo_vec_count(pv[v]));
        bmx::bvector::statistics st; gv[v].calc_stat(&st);
        uint32_t c[4]; uint64_t gw; bmo_vec_stat(pv[v], c, &gw);
        REQUIRE(st.bit_blocks == c[BMO_BIT] && st.gap_blocks == c[BMO_GAP]);
    }
    // pairwise, 3-operand and 2-operand forms
    for (int op = 0; op < 4; ++op) {
        bmx::bvector t(ctx), u(ctx);
        bmo_vec* e = bmo_op2(op, pv[1], pv[5], 0);
        switch (op) {
        case 0: t.bit_and(gv[1], gv[5]); REQUIRE(bmx::count_and(gv[1], gv[5]) == bmo_vec_count(e)); break;
        case 1: t.bit_or(gv[1], gv[5], bmx::bvector::opt_compress); REQUIRE(bmx::count_or(gv[1], gv[5]) == bmo_vec_count(e)); break;
        case 2: t.bit_xor(gv[1], gv[5]); REQUIRE(bmx::count_xor(gv[1], gv[5]) == bmo_vec_count(e)); break;
        default: t.bit_sub(gv[1], gv[5]); REQUIRE(bmx::count_sub(gv[1], gv[5]) == bmo_vec_count(e)); break;
        }
        REQUIRE(t.count() == bmo_vec_count(e));
        std::vector<uint32_t> w1(12 * 2048), w2(12 * 2048);
        t.export_words(w1.data(), w1.size()); bmo_vec_to_words(e, w2.data(), w2.size());
        REQUIRE(w1 == w2);
        // in-place form must give the same vector
        bmx::bit_import_u32(u, words[1].data(), words[1].size(), true);
        switch (op) { case 0: u &= gv[5]; break; case 1: u |= gv[5]; break; case 2: u ^= gv[5]; break; default: u -= gv[5]; break; }
        REQUIRE(u.equal(t));
        bmo_vec_free(e);
    }
    // asynchronous chain (bmx_op2_dev; dense vectors here): ((a & b) | c) - a, one wait at the end = the synchronous result
    {
        std::vector<bmx::bvector> dv; std::vector<bmo_vec*> dp;
        for (unsigned v = 0; v < 3; ++v) {
            std::vector<uint32_t> w((nbits + 31) / 32);
            bmo_gen_words(SEED, 50 + v, 0, 20000 + 3000 * v, nbits, 0, w.size(), w.data());
            dv.emplace_back(ctx); bmx::bit_import_u32(dv[v], w.data(), w.size(), false);
            dp.push_back(bmo_vec_import(w.data(), w.size(), 0));
        }
        bmx::pending p1 = bmx::bit_op_async(BMX_AND, dv[0], dv[1]);
        bmx::pending p2 = bmx::bit_op_async(BMX_OR, p1, dv[2]);
        bmx::pending p3 = bmx::bit_op_async(BMX_SUB, p2, dv[0]);
        bmx::bvector r(ctx), s1(ctx), s2(ctx), s3(ctx);
        p3.wait(r);
        s1.bit_and(dv[0], dv[1]); s2.bit_or(s1, dv[2]); s3.bit_sub(s2, dv[0]);
        REQUIRE(r.equal(s3));
        bmo_vec* e1 = bmo_op2(0, dp[0], dp[1], 0); bmo_vec* e2 = bmo_op2(1, e1, dp[2], 0); bmo_vec* e3 = bmo_op2(3, e2, dp[0], 0);
        REQUIRE(r.count() == bmo_vec_count(e3));
        bmo_vec_free(e1); bmo_vec_free(e2); bmo_vec_free(e3);
        for (auto* q : dp) bmo_vec_free(q);
    }
    // aggregator, member API
    bmx::aggregator<bmx::bvector> agg(ctx);
    for (unsigned v = 0; v < 4; ++v) agg.add(&gv[v]);
    agg.add(nullptr);                                   // ignored
    bool threw = false;
    try { agg.add(&gv[0], 2); } catch (const bmx::error& e) { threw = e.status() == BMX_ERR_RANGE; }
    REQUIRE(threw);
    agg.add(&gv[8], 1);
    bmx::bvector t(ctx);
    bool any = agg.combine_and_sub(t);
    const bmo_vec* a4[4] = {pv[0], pv[1], pv[2], pv[3]}; const bmo_vec* s1[1] = {pv[8]};
    bmo_vec* e = bmo_agg_and_sub(a4, 4, s1, 1);
    REQUIRE(t.count() == bmo_vec_count(e) && any == (bmo_vec_count(e) != 0));
    {
        bmx::size_type idx = 0; uint64_t pidx = 0;
        bool f = agg.find_first_and_sub(idx);
        int pf = bmo_find_first_and_sub(a4, 4, s1, 1, &pidx);
        REQUIRE(f == (pf != 0) && (!f || idx == pidx));
    }
    agg.combine_or(t);                                  // also clears the arg-groups (reference :1110)
    bmo_vec* eo = bmo_agg_or(a4, 4);
    REQUIRE(t.count() == bmo_vec_count(eo));
    { bmx::size_type idx = 0; REQUIRE(!agg.find_first_and_sub(idx)); }
    bmo_vec_free(e); bmo_vec_free(eo);
    // combine_shift_right_and (member form on a fresh group, then the count-only mode)
    {
        bmx::aggregator<bmx::bvector> sagg(ctx);
        const bmo_vec* s3[3] = {pv[0], pv[1], pv[2]};
        for (unsigned v = 0; v < 3; ++v) sagg.add(&gv[v]);
        bmx::bvector st(ctx);
        sagg.combine_shift_right_and(st);
        int pf = 0;
        bmo_vec* es = bmo_agg_shift_right_and(s3, 3, 0, 0, &pf);
        REQUIRE(st.count() == bmo_vec_count(es));
        std::vector<uint32_t> w1(12 * 2048), w2(12 * 2048);
        st.export_words(w1.data(), w1.size()); bmo_vec_to_words(es, w2.data(), w2.size());
        REQUIRE(w1 == w2);
        sagg.set_compute_count(true);
        sagg.combine_shift_right_and(st);
        REQUIRE(sagg.count() == bmo_agg_shift_right_and_count(s3, 3));
        bmo_vec_free(es);
    }
    // counts-only pipeline
    bmx::aggregator<bmx::bvector>::pipeline<bmx::agg_opt_only_counts> pipe(ctx);
    {
        auto* g0 = pipe.add(); for (unsigned v = 0; v < NV; ++v) g0->add(&gv[v], 0);
        auto* g1 = pipe.add(); g1->add(&gv[0], 0); g1->add(&gv[1], 0); g1->add(&gv[9], 1);
        auto* g2 = pipe.add(); g2->add(&gv[4], 0); g2->add(&gv[7], 0);
    }
    pipe.complete();
    agg.combine_and_sub(pipe);
    const bmo_vec* al[14]; for (unsigned v = 0; v < NV; ++v) al[v] = pv[v];
    al[10] = pv[0]; al[11] = pv[1]; al[12] = pv[4]; al[13] = pv[7];
    const bmo_vec* sl[1] = {pv[9]};
    uint32_t an[3] = {NV, 2, 2}, sn[3] = {0, 1, 0}; uint64_t exp[3];
    bmo_agg_pipeline_counts(al, an, sl, sn, 3, 0, 12, exp);
    for (int g = 0; g < 3; ++g) REQUIRE(pipe.get_bv_count_vector()[g] == exp[g]);
    // rank / select
    bmx::rs_index rs; gv[5].build_rs_index(&rs);
    bmo_rs* prs = bmo_rs_build(pv[5]);
    REQUIRE(rs.count() == bmo_rs_count(prs));
    for (uint64_t n = 0; n < nbits; n += 9973) REQUIRE(gv[5].count_to(n, rs) == bmo_rank(pv[5], prs, n));
    for (uint64_t r = 1; r <= rs.count(); r += 1 + rs.count() / 97) {
        uint64_t p1 = 0, p2 = 0;
        REQUIRE(gv[5].select(r, p1, rs) && bmo_select(pv[5], prs, r, &p2) && p1 == p2);
    }
    for (uint64_t n = 3; n + 5000 < nbits; n += 77777) {
        REQUIRE(gv[5].count_range(n, n + 5000, rs) == bmo_count_range(pv[5], prs, n, n + 5000));
        REQUIRE(gv[5].count_range(n + 5000, n, rs) == bmo_count_range(pv[5], prs, n, n + 5000));
        REQUIRE(gv[5].rank_corrected(n, rs) == bmo_rank_corrected(pv[5], prs, n));
        REQUIRE(gv[5].count_to_test(n, rs) == bmo_count_to_test(pv[5], prs, n));
        uint64_t p1 = 0, p2 = 0;
        bool f1 = gv[5].find_rank(7, n, p1, rs); int f2 = bmo_find_rank(pv[5], prs, 7, n, &p2);
        REQUIRE(f1 == (f2 != 0) && (!f1 || p1 == p2));
    }
    uint64_t dummy;
    REQUIRE(!gv[5].select(0, dummy, rs) && !gv[5].select(rs.count() + 1, dummy, rs));
    {   // round 6: what the index holds on the device (bmx_rs_info + bmx_rs_select_format through the facade)
        const bmx::rs_index::device_layout l = rs.layout();
        REQUIRE(l.bytes > 0 && (l.select_offset_bits == 0 || l.select_offset_bits == 16 || l.select_offset_bits == 32));
        REQUIRE((l.select_offset_bits == 0) == (l.select_lines_bytes == 0));
        if (l.select_offset_bits) REQUIRE(l.select_lines_bytes == (rs.count() + (l.select_offset_bits == 16 ? 59 : 29)) / (l.select_offset_bits == 16 ? 60 : 30) * 128);
    }
    bmo_rs_free(prs);
    for (unsigned v = 0; v < NV; ++v) bmo_vec_free(pv[v]);
    std::printf("test_facade ok\n");
    return 0;
}
