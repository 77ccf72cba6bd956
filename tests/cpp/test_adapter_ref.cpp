// End-to-end drop-in check with the REAL reference container: build bm::bvector<>s with
// BitMagic, upload through bmx/bm_adapter.hpp, run the path on the GPU, download into
// bm::bvector<> and compare with BitMagic's own result (compare() == 0).
// Compiled ONLY where /root/reference exists (tests/cpp/Makefile -> oracle/_ref/, git-ignored);
// the binary travels to the GPU box and is run by tests/test_cpp_facade.py (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <vector>

#include "bm.h"
#include "bmaggregator.h"
#include "bmalgo.h"
#include "bmbvimport.h"
#include "bmsparsevec.h"
#include "bmsparsevec_algo.h"

#include "bmx/bm_adapter.hpp"
#include "bmx/scanner.hpp"
#include "bmx/group.hpp"
extern "C" {
#include "../../oracle/bmx_oracle.h"     // only for the deterministic input generator
}

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

typedef bm::bvector<> bvect;

int main()
{
    bmx::context ctx(0);
    const uint64_t nbits = 9 * 65536 + 123;
    const unsigned NV = 8;
    const uint32_t NB = 10;
    std::vector<bvect> hv(NV);
    std::vector<bmx::bvector> gv;
    for (unsigned v = 0; v < NV; ++v) {
        uint32_t dq = v < 3 ? 6554u : (v < 6 ? 655u : 100u);
        uint64_t nw = ((nbits + 63) / 64) * 2;
        std::vector<uint32_t> w(nw);
        bmo_gen_words(0xB17A61C, v, v < 3, dq, nbits, 0, nw, w.data());
        bm::bit_import_u32(hv[v], w.data(), bvect::size_type(nw), true);
        if (v == 2) hv[v].set_range(65536 * 3, 65536 * 5 - 1);          // FULL blocks
        if (v == 4) hv[v].set_range(65536 * 6, 65536 * 7 - 1, false);   // a NULL block
        gv.emplace_back(ctx);
        bmx::upload(hv[v], gv[v], NB);
        REQUIRE(gv[v].count() == hv[v].count());
        bvect back; bmx::download(gv[v], back);
        REQUIRE(back.compare(hv[v]) == 0);
    }
    // pairwise ops vs the reference's own results
    for (int op = 0; op < 4; ++op) {
        for (unsigned i = 0; i + 1 < NV; i += 2) {
            bvect ref; bmx::bvector t(ctx);
            switch (op) {
            case 0: ref.bit_and(hv[i], hv[i + 1]); t.bit_and(gv[i], gv[i + 1], bmx::bvector::opt_compress); REQUIRE(bmx::count_and(gv[i], gv[i + 1]) == bm::count_and(hv[i], hv[i + 1])); break;
            case 1: ref.bit_or(hv[i], hv[i + 1]); t.bit_or(gv[i], gv[i + 1]); REQUIRE(bmx::count_or(gv[i], gv[i + 1]) == bm::count_or(hv[i], hv[i + 1])); break;
            case 2: ref.bit_xor(hv[i], hv[i + 1]); t.bit_xor(gv[i], gv[i + 1], bmx::bvector::opt_compress); REQUIRE(bmx::count_xor(gv[i], gv[i + 1]) == bm::count_xor(hv[i], hv[i + 1])); break;
            default: ref.bit_sub(hv[i], hv[i + 1]); t.bit_sub(gv[i], gv[i + 1]); REQUIRE(bmx::count_sub(gv[i], gv[i + 1]) == bm::count_sub(hv[i], hv[i + 1])); break;
            }
            bvect got; bmx::download(t, got);
            REQUIRE(got.compare(ref) == 0);
        }
    }
    // aggregator vs bm::aggregator
    {
        bm::aggregator<bvect> ragg; bmx::aggregator<bmx::bvector> gagg(ctx);
        for (unsigned v = 0; v < 5; ++v) { ragg.add(&hv[v]); gagg.add(&gv[v]); }
        ragg.add(&hv[6], 1); gagg.add(&gv[6], 1);
        bvect r1, r2, g1, g2; bmx::bvector t(ctx);
        ragg.combine_and_sub(r1); bool any = gagg.combine_and_sub(t); bmx::download(t, g1);
        REQUIRE(g1.compare(r1) == 0 && any == r1.any());
        ragg.combine_or(r2); gagg.combine_or(t); bmx::download(t, g2);
        REQUIRE(g2.compare(r2) == 0);
        // counts-only pipeline vs the reference pipeline
        bm::aggregator<bvect>::pipeline<bm::agg_opt_only_counts> rp;
        bmx::aggregator<bmx::bvector>::pipeline<bmx::agg_opt_only_counts> gp(ctx);
        for (unsigned g = 0; g < 3; ++g) {
            auto* ra = rp.add(); auto* ga = gp.add();
            for (unsigned v = g; v < NV; v += (g + 1)) { ra->add(&hv[v], 0); ga->add(&gv[v], 0); }
            if (g) { ra->add(&hv[7], 1); ga->add(&gv[7], 1); }
        }
        rp.complete(); gp.complete();
        ragg.combine_and_sub(rp); gagg.combine_and_sub(gp);
        for (unsigned g = 0; g < 3; ++g) REQUIRE(gp.get_bv_count_vector()[g] == rp.get_bv_count_vector()[g]);
        // results + counts + OR target vs the reference pipeline<agg_opt_bvect_and_counts>
        bm::aggregator<bvect>::pipeline<bm::agg_opt_bvect_and_counts> rp2;
        bmx::aggregator<bmx::bvector>::pipeline<bmx::agg_opt_bvect_and_counts> gp2(ctx);
        bvect r_or; bmx::bvector g_or(ctx);
        rp2.set_or_target(&r_or); gp2.set_or_target(&g_or);
        for (unsigned g = 0; g < 4; ++g) {
            auto* ra = rp2.add(); auto* ga = gp2.add();
            for (unsigned v = g; v < NV; v += 2) { ra->add(&hv[v], 0); ga->add(&gv[v], 0); }
            if (g & 1) { ra->add(&hv[0], 1); ga->add(&gv[0], 1); }
        }
        rp2.complete(); gp2.complete();
        ragg.combine_and_sub(rp2); gagg.combine_and_sub(gp2);
        for (unsigned g = 0; g < 4; ++g) {
            REQUIRE(gp2.get_bv_count_vector()[g] == rp2.get_bv_count_vector()[g]);
            const bvect* rr = rp2.get_bv_res_vector()[g]; bmx::bvector* gr = gp2.get_bv_res_vector()[g];
            REQUIRE((rr == nullptr) == (gr == nullptr));
            if (rr) { bvect got; bmx::download(*gr, got); REQUIRE(got.compare(*rr) == 0); }
        }
        bvect got_or; bmx::download(g_or, got_or);
        REQUIRE(got_or.compare(r_or) == 0);
        // find_first_and_sub vs the reference
        bvect::size_type ri = 0; bmx::size_type gi = 0;
        bool rf = ragg.find_first_and_sub(ri), gf = gagg.find_first_and_sub(gi);
        if (!(rf == gf && (!rf || ri == gi))) std::fprintf(stderr, "find_first_and_sub: reference %d %llu, device %d %llu\n",
                                                           (int)rf, (unsigned long long)ri, (int)gf, (unsigned long long)gi);
        REQUIRE(rf == gf && (!rf || ri == gi));
    }
    // frozen vectors (freeze() arena, src/bmblocks.h:2572-2770) upload straight from the arena
    {
        for (unsigned v = 0; v < NV; v += 3) {
            bvect fz(hv[v]);
            fz.freeze();
            REQUIRE(fz.is_ro());
            bmx::bvector g(ctx);
            bool zero_copy = bmx::upload(fz, g, NB);
            REQUIRE(zero_copy);
            REQUIRE(g.count() == hv[v].count() && g.equal(gv[v]));
            bvect back; bmx::download(g, back);
            REQUIRE(back.compare(hv[v]) == 0);
        }
        bmx::bvector g2(ctx);
        REQUIRE(!bmx::upload(hv[1], g2, NB));           // mutable vector: gathered copy
    }
    // combine_shift_right_and vs the reference (dense operands so that the chain survives), member form,
    // opt_compress and count-only modes
    {
        std::vector<bvect> dv(6); std::vector<bmx::bvector> dg; dg.reserve(6);
        for (unsigned v = 0; v < 6; ++v) {
            dv[v].set_range(10, bvect::size_type(nbits - 20));
            for (bvect::size_type k = 100 + v; k < nbits - 20; k += 997 + 131 * v) dv[v].clear_bit(k);
            if (v & 1) dv[v].optimize();
            dg.emplace_back(ctx); bmx::upload(dv[v], dg.back(), NB);
        }
        bm::aggregator<bvect> ragg; bmx::aggregator<bmx::bvector> gagg(ctx);
        for (unsigned v = 0; v < 6; ++v) { ragg.add(&dv[v]); gagg.add(&dg[v]); }
        for (int opt = 0; opt < 2; ++opt) {
            bvect r; bmx::bvector t(ctx); bvect g;
            if (opt) { ragg.set_optimization(); gagg.set_optimization(); }
            ragg.combine_shift_right_and(r); gagg.combine_shift_right_and(t); bmx::download(t, g);
            REQUIRE(r.any() && g.compare(r) == 0);
        }
        ragg.set_compute_count(true); gagg.set_compute_count(true);
        bvect r; bmx::bvector t(ctx);
        ragg.combine_shift_right_and(r); gagg.combine_shift_right_and(t);
        REQUIRE(ragg.count() == gagg.count() && gagg.count() != 0);
    }
    // bit-sliced search: bm::sparse_vector_scanner::find_eq / find_first_eq vs bmx::slice_scanner on the
    // uploaded slices (group rule src/bmsparsevec_algo.h:2593-2640)
    {
        typedef bm::sparse_vector<unsigned, bvect> svect;
        svect sv;
        const unsigned N = 300000;
        uint64_t x = 88172645463325252ull;
        {
            svect::back_insert_iterator bi = sv.get_back_inserter();
            for (unsigned i = 0; i < N; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                unsigned v = (unsigned)(x % 97u);                    // 7 slices, value 0 occurs too
                if (i % 1000 == 0) v = 70000u + (i % 7u);            // rare wide values: sparse high slices
                bi = v;
            }
            bi.flush();
        }
        sv.optimize();
        std::vector<bmx::bvector> store; std::vector<const bmx::bvector*> slices;
        bmx::upload_slices(sv, ctx, store, slices);
        REQUIRE(slices.size() == sv.effective_slices());
        bmx::slice_scanner gs(ctx);
        gs.bind(slices);
        bm::sparse_vector_scanner<svect> rs;
        std::vector<uint64_t> vals, ref_counts, got;
        for (unsigned v : {1u, 2u, 17u, 64u, 96u, 97u, 120u, 70000u, 70003u, 70006u, 131072u, 5000000u}) {
            bvect r; bmx::bvector g(ctx); bvect gh;
            rs.find_eq(sv, v, r);                                    // :1083 -> find_eq_with_nulls :2387
            bool rf = r.any();
            bool gf = gs.find_eq(v, g);
            if (!g.empty_handle()) bmx::download(g, gh);
            REQUIRE(rf == gf && gh.compare(r) == 0);
            bvect::size_type ri = 0; bmx::size_type gi = 0;
            bool rff = rs.find_eq(sv, v, ri), gff = gs.find_first_eq(v, gi);       // :1111 -> find_first_eq :2417
            REQUIRE(rff == gff && (!rff || ri == gi));
            // index-list output: find_eq(sv, value, BII) :1096 -- sorted positions through a back-insert iterator
            { std::vector<bvect::size_type> rpos; std::vector<bmx::size_type> gpos;
              rs.find_eq(sv, v, std::back_inserter(rpos)); bool gfi = gs.find_eq(v, std::back_inserter(gpos));
              REQUIRE(gfi == !gpos.empty() && rpos.size() == gpos.size());
              for (size_t k = 0; k < rpos.size(); ++k) REQUIRE((bmx::size_type)rpos[k] == gpos[k]); }
            vals.push_back(v); ref_counts.push_back(r.count());
        }
        got.resize(vals.size());
        gs.find_eq_counts(vals.data(), vals.size(), got.data());                     // one pass: transposition + hash lookup
        REQUIRE(got == ref_counts);
        std::fill(got.begin(), got.end(), ~0ull);
        gs.find_eq_counts(vals.data(), vals.size(), got.data(), true);               // one AND-SUB group per value (pipeline)
        REQUIRE(got == ref_counts);
        // range search: find_gt / find_ge / find_lt / find_le / find_range / find_zero / find_nonzero / find_eq(0)
        // vs the real scanner (src/bmsparsevec_algo.h:1135-1174, 2290, 2690-2880, 4464), with and without NULLs
        typedef bm::sparse_vector<unsigned, bvect> svect_t;
        svect_t svn(bm::use_null);
        for (unsigned i = 0; i < N; ++i) { unsigned v = sv.get(i); if (i % 11u == 3u) continue; svn.set(i, v); }   // gaps = NULL elements
        svn.optimize();
        for (int with_null = 0; with_null < 2; ++with_null) {
            const svect_t& S = with_null ? svn : sv;
            std::vector<bmx::bvector> st2; std::vector<const bmx::bvector*> sl2; const bmx::bvector* nn = nullptr;
            bmx::upload_slices(S, ctx, st2, sl2, &nn);
            REQUIRE((nn != nullptr) == (with_null != 0));
            bmx::slice_scanner g2(ctx);
            g2.bind(sl2, S.size(), nn);
            bm::sparse_vector_scanner<svect_t> r2;
            auto same = [&](bvect& r, bmx::bvector& g) { bvect gh; if (!g.empty_handle()) bmx::download(g, gh); return gh.compare(r) == 0; };
            for (unsigned v : {0u, 1u, 2u, 50u, 95u, 96u, 97u, 128u, 69999u, 70000u, 70003u, 70006u, 70007u, 131071u, 131072u, 4000000u}) {
                bvect r; bmx::bvector g(ctx);
                r2.find_gt(S, v, r); g2.find_gt(v, g); REQUIRE(same(r, g));
                r.clear(); r2.find_ge(S, v, r); g2.find_ge(v, g); REQUIRE(same(r, g));
                r.clear(); r2.find_lt(S, v, r); g2.find_lt(v, g); REQUIRE(same(r, g));
                r.clear(); r2.find_le(S, v, r); g2.find_le(v, g); REQUIRE(same(r, g));
                REQUIRE(g2.count(BMX_CMP_LE, v) == r.count());
            }
            for (auto pr : {std::pair<unsigned, unsigned>{0u, 0u}, {0u, 5u}, {3u, 3u}, {90u, 10u}, {10u, 69999u}, {96u, 70003u}, {70001u, 70005u}, {1u, 4000000u}}) {
                bvect r; bmx::bvector g(ctx);
                r2.find_range(S, pr.first, pr.second, r); g2.find_range(pr.first, pr.second, g);
                REQUIRE(same(r, g));
            }
            { bvect r; bmx::bvector g(ctx); r2.find_zero(S, r); g2.find_zero(g); REQUIRE(same(r, g)); }
            { bvect r; bmx::bvector g(ctx); r2.find_nonzero(S, r); g2.find_nonzero(g); REQUIRE(same(r, g)); }
            { bvect r; bmx::bvector g(ctx); r2.find_eq(S, 0u, r); bool gf = g2.find_eq(0, g); REQUIRE(r.any() == gf && same(r, g)); }
            // the same searches over a device GROUP: planes sharded over three members (bmx::gslice_scanner, group.hpp)
            bmx::device_group grp3({0, 0, 0});
            std::vector<bmx::gbvector> st3; std::vector<const bmx::gbvector*> sl3; const bmx::gbvector* nn3 = nullptr;
            bmx::upload_slices(S, grp3, st3, sl3, &nn3);
            bmx::gslice_scanner g3(grp3);
            g3.bind(sl3, S.size(), nn3);
            auto gsame = [&](bvect& r, bmx::gbvector& g) { bvect gh; if (!g.empty_handle()) bmx::download(g, gh); return gh.compare(r) == 0; };
            for (unsigned v : {0u, 2u, 96u, 70003u, 131072u}) {
                bvect r; bmx::gbvector g(grp3);
                r2.find_gt(S, v, r); g3.find_gt(v, g); REQUIRE(gsame(r, g));
                r.clear(); r2.find_le(S, v, r); g3.find_le(v, g); REQUIRE(gsame(r, g));
                REQUIRE(g3.count(BMX_CMP_LE, v) == r.count());
                r.clear(); r2.find_eq(S, v, r); bool gf = g3.find_eq(v, g); REQUIRE(r.any() == gf && gsame(r, g));
                bvect::size_type ri = 0; bmx::size_type gi = 0;
                if (v) { bool rff = r2.find_eq(S, v, ri), gff = g3.find_first_eq(v, gi); REQUIRE(rff == gff && (!rff || ri == gi)); }
            }
            { bvect r; bmx::gbvector g(grp3); r2.find_range(S, 10u, 69999u, r); g3.find_range(10u, 69999u, g); REQUIRE(gsame(r, g)); }
            { uint64_t vals3[3] = {17u, 70003u, 5000000u}, got3[3]; g3.find_eq_counts(vals3, 3, got3);
              for (int k = 0; k < 3; ++k) { bvect r; r2.find_eq(S, (unsigned)vals3[k], r); REQUIRE(got3[k] == r.count()); } }
        }
    }
    // host rs_index filled from the GPU-built index (bmx::build_rs_index): identical answers to the reference's own index
    for (unsigned v = 0; v < 4; ++v) {
        bvect::rs_index_type rs_ref, rs_gpu;
        hv[v].build_rs_index(&rs_ref);
        bmx::build_rs_index(hv[v], gv[v], &rs_gpu);
        REQUIRE(rs_ref.count() == rs_gpu.count() && rs_ref.get_total() == rs_gpu.get_total());
        uint64_t x = 0x2545F4914F6CDD1Dull + v;
        for (int it = 0; it < 3000; ++it) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            bvect::size_type n = (bvect::size_type)(x % nbits);
            REQUIRE(hv[v].count_to(n, rs_ref) == hv[v].count_to(n, rs_gpu));
            bvect::size_type cnt = rs_ref.count();
            if (cnt) {
                bvect::size_type r = (bvect::size_type)(1 + (x >> 20) % cnt), p1 = 0, p2 = 0;
                bool f1 = hv[v].select(r, p1, rs_ref), f2 = hv[v].select(r, p2, rs_gpu);
                REQUIRE(f1 == f2 && p1 == p2);
            }
        }
    }
    // set_range_hint + find_first_and_sub vs the real aggregator (src/bmaggregator.h:974,1458)
    {
        bm::aggregator<bvect> ragg; bmx::aggregator<bmx::bvector> gagg(ctx);
        ragg.add(&hv[0]); ragg.add(&hv[1]); ragg.add(&hv[2], 1);
        gagg.add(&gv[0]); gagg.add(&gv[1]); gagg.add(&gv[2], 1);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (int it = 0; it < 40; ++it) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint64_t a = x % nbits; x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint64_t b = (it & 1) ? std::min<uint64_t>(nbits - 1, a + x % 3000) : x % nbits;
            if (a > b) std::swap(a, b);
            bool r1 = ragg.set_range_hint(bvect::size_type(a), bvect::size_type(b));
            bool g1 = gagg.set_range_hint(a, b);
            bvect::size_type ri = 0; bmx::size_type gi = 0;
            bool rf = ragg.find_first_and_sub(ri), gf = gagg.find_first_and_sub(gi);
            REQUIRE(r1 == g1 && rf == gf && (!rf || ri == gi));
        }
        ragg.reset_range_hint(); gagg.reset_range_hint();
        bvect::size_type ri = 0; bmx::size_type gi = 0;
        REQUIRE(ragg.find_first_and_sub(ri) == gagg.find_first_and_sub(gi) && ri == gi);
    }
    // bmx::device_aggregator<bm::bvector<>>: the drop-in over HOST vectors vs bm::aggregator<bm::bvector<>>
    // (add / combine_or / combine_and / combine_and_sub / find_first_and_sub / pipeline), and its upload cache
    {
        bm::aggregator<bvect> ragg; bmx::device_aggregator<bvect> dagg(ctx);
        std::vector<bvect> fz(4);                         // frozen copies: cached across operations
        for (unsigned v = 0; v < 4; ++v) { fz[v] = hv[v]; fz[v].freeze(); }
        for (int round = 0; round < 2; ++round) {
            for (unsigned v = 0; v < 3; ++v) { ragg.add(&hv[v]); dagg.add(round ? &fz[v] : &hv[v]); }
            ragg.add(&hv[3], 1); dagg.add(round ? &fz[3] : &hv[3], 1);
            bvect r, g;
            bool rf = ragg.combine_and_sub(r), gf = dagg.combine_and_sub(g);
            REQUIRE(rf == gf && g.compare(r) == 0);
            if (round == 0) {
                // combine_and_sub_bi(BII) :450,1068 on resident vectors: sorted positions = the reference's bulk-insert output
                bmx::aggregator<bmx::bvector> bagg(ctx);
                for (unsigned v = 0; v < 3; ++v) bagg.add(&gv[v]);
                bagg.add(&gv[3], 1);
                std::vector<bvect::size_type> rpos; std::vector<bmx::size_type> gpos;
                bool rb = ragg.combine_and_sub_bi(std::back_inserter(rpos)), gb = bagg.combine_and_sub_bi(std::back_inserter(gpos));
                REQUIRE(rb == gb && rpos.size() == gpos.size() && rpos.size() == r.count());
                for (size_t k = 0; k < rpos.size(); ++k) REQUIRE((bmx::size_type)rpos[k] == gpos[k]);
                std::vector<bmx::size_type> all; gv[1].to_indices(all);
                REQUIRE(all.size() == hv[1].count());
                bvect::enumerator en = hv[1].first();
                for (size_t k = 0; k < all.size(); ++k, ++en) REQUIRE(en.valid() && (bmx::size_type)*en == all[k]);
            }
            bvect::size_type ri = 0, gi = 0;
            REQUIRE(ragg.find_first_and_sub(ri) == dagg.find_first_and_sub(gi) && ri == gi);
            ragg.combine_and(r); dagg.combine_and(g); REQUIRE(g.compare(r) == 0);
            ragg.combine_or(r); dagg.combine_or(g); REQUIRE(g.compare(r) == 0);          // clears the groups (:1110)
            ragg.combine_and(r); dagg.combine_and(g); REQUIRE(!r.any() && !g.any());      // empty AND group => cleared target
            ragg.reset(); dagg.reset();
        }
        size_t up0 = dagg.uploads();
        for (unsigned v = 0; v < 4; ++v) dagg.add(&fz[v]);
        bvect g; dagg.combine_or(g);
        REQUIRE(dagg.uploads() == up0);                   // frozen vectors were resident already
        dagg.add(&hv[0]); dagg.add(&hv[1]); dagg.combine_and(g);
        REQUIRE(dagg.uploads() == up0 + 2);               // mutable vectors are uploaded again ...
        hv[0].set(12345); hv[0].clear_bit(12345);         // (a change the cache could not see)
        dagg.add(&hv[0]); dagg.add(&hv[1]); dagg.combine_and(g);
        REQUIRE(dagg.uploads() == up0 + 4);
        dagg.set_cache_mutable(true);                     // ... unless the caller takes responsibility
        dagg.add(&hv[0]); dagg.add(&hv[1]); dagg.combine_and(g);
        dagg.add(&hv[0]); dagg.add(&hv[1]); dagg.combine_and(g);
        REQUIRE(dagg.uploads() == up0 + 4);               // the copies of the previous operation are re-used
        dagg.invalidate(&hv[0]);
        dagg.add(&hv[0]); dagg.add(&hv[1]); dagg.combine_and(g);
        REQUIRE(dagg.uploads() == up0 + 5);
        { bm::aggregator<bvect> ra2; ra2.add(&hv[0]); ra2.add(&hv[1]); bvect r; ra2.combine_and(r); REQUIRE(g.compare(r) == 0); }
        // pipeline over host vectors: counts + result vectors + OR target
        typedef bm::agg_run_options<true, true> ropt; typedef bmx::agg_run_options<true, true> gopt;
        bm::aggregator<bvect>::pipeline<ropt> rp; bmx::device_aggregator<bvect>::pipeline<gopt> gp;
        bvect r_or, g_or; rp.set_or_target(&r_or); gp.set_or_target(&g_or);
        for (unsigned q = 0; q < 5; ++q) {
            auto* ra = rp.add(); auto* ga = gp.add();
            ra->add(&hv[q % 4], 0); ga->add(&hv[q % 4], 0);
            ra->add(&hv[(q + 1) % 4], 0); ga->add(&hv[(q + 1) % 4], 0);
            if (q & 1) { ra->add(&hv[(q + 2) % 4], 1); ga->add(&hv[(q + 2) % 4], 1); }
        }
        rp.complete(); gp.complete();
        bm::aggregator<bvect> ra3; ra3.combine_and_sub(rp); dagg.combine_and_sub(gp);
        for (unsigned q = 0; q < 5; ++q) {
            REQUIRE(rp.get_bv_count_vector()[q] == gp.get_bv_count_vector()[q]);
            const bvect* rr = rp.get_bv_res_vector()[q]; const bvect* gr = gp.get_bv_res_vector()[q];
            REQUIRE((rr == nullptr) == (gr == nullptr) && (!rr || gr->compare(*rr) == 0));
        }
        REQUIRE(g_or.compare(r_or) == 0 && g_or.any());
    }
    // multi-GPU facade (bmx/group.hpp): the same calls over a device GROUP -- three members on this one GPU; vectors are
    // sharded by block range, results gathered on download -- against bm::aggregator / bm::bvector on the host
    {
        bmx::device_group grp({0, 0, 0});
        REQUIRE(grp.size() == 3);
        std::vector<bmx::gbvector> gg; gg.reserve(6);
        for (unsigned v = 0; v < 6; ++v) { gg.emplace_back(grp); bmx::upload(hv[v], gg.back(), NB); REQUIRE(gg[v].count() == hv[v].count()); }
        { bvect back; bmx::download(gg[2], back); REQUIRE(back.compare(hv[2]) == 0); }
        bmx::gbvector t(grp); bvect ref, got;
        t.bit_and(gg[0], gg[1], bmx::bvector::opt_compress); ref.bit_and(hv[0], hv[1]); bmx::download(t, got); REQUIRE(got.compare(ref) == 0);
        t.bit_xor(gg[2], gg[3]); ref.bit_xor(hv[2], hv[3]); bmx::download(t, got); REQUIRE(got.compare(ref) == 0);
        REQUIRE(bmx::count_and(gg[0], gg[1]) == bm::count_and(hv[0], hv[1]) && bmx::count_sub(gg[4], gg[5]) == bm::count_sub(hv[4], hv[5]));
        bm::aggregator<bvect> ragg; bmx::aggregator<bmx::gbvector> gagg(grp);
        for (unsigned v = 0; v < 3; ++v) { ragg.add(&hv[v]); gagg.add(&gg[v]); }
        ragg.add(&hv[3], 1); gagg.add(&gg[3], 1);
        bool rf = ragg.combine_and_sub(ref), gf = gagg.combine_and_sub(t);
        bmx::download(t, got); REQUIRE(rf == gf && got.compare(ref) == 0);
        { bvect::size_type ri = 0; uint64_t gi = 0;
          bool rff = ragg.find_first_and_sub(ri), gff = gagg.find_first_and_sub(gi);
          REQUIRE(rff == gff && (!rff || ri == gi)); }
        ragg.combine_or(ref); gagg.combine_or(t); bmx::download(t, got); REQUIRE(got.compare(ref) == 0);
        typedef bm::aggregator<bvect>::pipeline<bm::agg_opt_only_counts> rpipe_t;
        typedef bmx::aggregator<bmx::gbvector>::pipeline<bmx::agg_opt_only_counts> gpipe_t;
        rpipe_t rp; gpipe_t gp(grp);
        for (unsigned q = 0; q < 4; ++q) {
            auto* ra = rp.add(); auto* ga = gp.add();
            ra->add(&hv[q], 0); ga->add(&gg[q], 0); ra->add(&hv[q + 1], 0); ga->add(&gg[q + 1], 0);
            if (q & 1) { ra->add(&hv[q + 2], 1); ga->add(&gg[q + 2], 1); }
        }
        rp.complete(); gp.complete();
        bm::aggregator<bvect> ra2; ra2.combine_and_sub(rp); gagg.combine_and_sub(gp);
        for (unsigned q = 0; q < 4; ++q) REQUIRE(rp.get_bv_count_vector()[q] == gp.get_bv_count_vector()[q]);
        REQUIRE(gp.last_ms().size() == 3);
        // rank / select over the sharded vector against the reference rs_index on the host
        bvect::rs_index_type rrs; hv[3].build_rs_index(&rrs);
        bmx::grs_index grs; gg[3].build_rs_index(&grs);
        REQUIRE(grs.count() == rrs.count());
        for (uint64_t n = 0; n < nbits; n += 6007) REQUIRE(gg[3].count_to(n, grs) == hv[3].count_to(bvect::size_type(n), rrs));
        for (uint64_t r = 1; r <= grs.count(); r += 1 + grs.count() / 97) {
            uint64_t p = 0; bvect::size_type q = 0;
            REQUIRE(gg[3].select(r, p, grs) && hv[3].select(bvect::size_type(r), q, rrs) && p == q);
        }
        { uint64_t p = 0; REQUIRE(!gg[3].select(grs.count() + 1, p, grs) && !gg[3].select(0, p, grs)); }
        REQUIRE(gg[3].count_range(70000, 300000, grs) == hv[3].count_range(70000, 300000, rrs));
    }
    // rank / select vs bvector<>::count_to / select with the reference rs_index
    {
        bvect::rs_index_type rrs; hv[3].build_rs_index(&rrs);
        bmx::rs_index grs; gv[3].build_rs_index(&grs);
        REQUIRE(grs.count() == rrs.count());
        for (uint64_t n = 0; n < nbits; n += 7919) REQUIRE(gv[3].count_to(n, grs) == hv[3].count_to(bvect::size_type(n), rrs));
        for (uint64_t r = 1; r <= grs.count(); r += 1 + grs.count() / 131) {
            uint64_t p = 0; bvect::size_type q = 0;
            REQUIRE(gv[3].select(r, p, grs) && hv[3].select(bvect::size_type(r), q, rrs) && p == q);
        }
        std::vector<uint32_t> bc; std::vector<uint64_t> sub;
        grs.export_blocks(bc, sub, NB);
        for (uint32_t nb = 0; nb < NB; ++nb) REQUIRE(bc[nb] == rrs.count(nb) && sub[nb] == rrs.sub_count(nb));
    }
    std::printf("test_adapter_ref ok (simd_version %d)\n", bm::simd_version());
    return 0;
}
