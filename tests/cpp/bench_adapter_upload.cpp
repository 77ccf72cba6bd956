// Host -> device bridge timing with the REAL reference container: a 1e9-bit bm::bvector<>
// uploaded (a) from a mutable vector (blocks gathered into a staging slab on the host) and (b) from a
// freeze()d vector (arena handed over as is).  Built only where /root/reference exists (-> oracle/_ref/).
#include <chrono>
#include <cstdio>
#include <vector>

#include "bm.h"
#include "bmbvimport.h"

#include "bmx/bm_adapter.hpp"
extern "C" {
#include "../../oracle/bmx_oracle.h"     // deterministic input generator only
}

typedef bm::bvector<> bvect;
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    bmx::context ctx(0);
    const uint64_t nbits = 1000000000ull;
    const uint32_t NB = (uint32_t)((nbits + 65535) / 65536);
    for (uint32_t dq : {6554u, 655u}) {
        uint64_t nw = ((nbits + 63) / 64) * 2;
        std::vector<uint32_t> w(nw);
        bmo_gen_words(0xB17A61C, 7, 0, dq, nbits, 0, nw, w.data());
        bvect bv;
        bm::bit_import_u32(bv, w.data(), bvect::size_type(nw), true);
        bvect fz(bv);
        fz.freeze();
        double best[2] = {1e30, 1e30};
        uint64_t cnt[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            for (int mode = 0; mode < 2; ++mode) {
                bmx::bvector g(ctx);
                double t0 = now_ms();
                bool zc = bmx::upload(mode ? fz : bv, g, NB);
                ctx.synchronize();
                double dt = now_ms() - t0;
                if (zc != (mode == 1)) { std::fprintf(stderr, "unexpected upload path\n"); return 1; }
                if (dt < best[mode]) best[mode] = dt;
                cnt[mode] = g.count();
            }
        }
        if (cnt[0] != cnt[1] || cnt[0] != bv.count()) { std::fprintf(stderr, "count mismatch\n"); return 1; }
        bvect::statistics st; bv.calc_stat(&st);
        double mb = (st.bit_blocks * 8192.0 + st.gap_blocks * 0.0) / 1e6;
        std::printf("{\"density_q16\": %u, \"bit_blocks\": %u, \"gap_blocks\": %u, \"upload_gather_ms\": %.2f, \"upload_frozen_ms\": %.2f, \"bit_MB\": %.1f}\n",
                    dq, (unsigned)st.bit_blocks, (unsigned)st.gap_blocks, best[0], best[1], mb);
    }
    return 0;
}
