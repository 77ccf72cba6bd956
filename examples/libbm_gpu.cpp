// libbm_gpu.cpp -- a GPU-enabled edition of BitMagic's own C wrapper (lang-maps/libbm): the pairwise + count surface the
// wrapper already exports (BM_bvector_count_AND/OR/XOR/SUB libbm.h:644-677, BM_bvector_combine_AND/OR/XOR/SUB :439-472,
// BM_bvector_count :314) routed to libbmx.so.  Handles stay what they are in libbm -- pointers to
// bm::bvector<libbm::standard_allocator> -- so every other BM_ call keeps working on them.
//
// This file is the binding INTEGRATION.md describes, for real: it is compiled together with the UNMODIFIED
// lang-maps/libbm/src/libbm.cpp from /root/reference (tests/cpp/Makefile, output under the git-ignored oracle/_ref/) and
// checked by tests/cpp/test_libbm_gpu.c, a plain C program that builds vectors through the libbm API and compares every
// BMX_ call with its BM_ twin.  The BMX_ prefix keeps both editions linkable side by side; a maintainer who ships
// a GPU build renames them to BM_ and drops the CPU bodies.
//
// Residency: a vector is uploaded on first use and stays resident, keyed by handle, until BMX_bvector_invalidate(h) /
// BMX_bvector_release(h) -- the C API has no change notification, so the contract is the same one
// bmx::device_aggregator offers with set_cache_mutable(true): read-only (BM_bvector_freeze) vectors need nothing,
// a vector the application mutates must be invalidated before the next GPU call.
#include "libbm.h"
#include "try_throw_catch.h"
extern __thread jmp_buf ex_buf__;               /* defined by libbm.cpp: the wrapper's setjmp / longjmp error channel */

#define BM_NO_STL
#define BM_NO_CXX11
#define BMALLOC__H__INCLUDED__
#define BM_ASSERT_THROW(x, xerrcode)            /* the GPU edition reports through status codes only */
#include "bmdef.h"
#include "bmconst.h"
#include "bmsimd.h"
#include "bmcalloc.h"
#include "bm.h"
#include "bmalgo.h"

#include "bmx/bm_adapter.hpp"

#include <unordered_map>

typedef bm::bvector<libbm::standard_allocator> TBM_bvector;     // == libbm.cpp's TBM_bvector

namespace {
struct gpu_state {
    bmx::context ctx{0};
    std::unordered_map<const void*, std::unique_ptr<bmx::bvector>> resident;
};
gpu_state& G() { static gpu_state g; return g; }

const bmx::bvector& device_copy(BM_BVHANDLE h)
{
    gpu_state& g = G();
    auto it = g.resident.find(h);
    if (it == g.resident.end()) {
        std::unique_ptr<bmx::bvector> d(new bmx::bvector(g.ctx));
        const TBM_bvector* bv = (const TBM_bvector*)h;
        uint32_t nb = bmx::effective_blocks(*bv);
        bmx::upload(*bv, *d, nb ? nb : 1u);
        it = g.resident.emplace(h, std::move(d)).first;
    }
    return *it->second;
}

int status_of(const bmx::error& e)
{
    switch (e.status()) {                                        // bmx.h uses libbm's numeric codes where a twin exists
    case BMX_ERR_BADALLOC: return BM_ERR_BADALLOC;
    case BMX_ERR_BADARG: return BM_ERR_BADARG;
    case BMX_ERR_RANGE: return BM_ERR_RANGE;
    default: return BM_ERR_CPU;
    }
}

int count_op(int op, BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* pcount)
{
    if (!h1 || !h2 || !pcount) return BM_ERR_BADARG;
    try {
        const bmx::bvector& a = device_copy(h1);
        const bmx::bvector& b = device_copy(h2);
        uint64_t c = 0;
        bmx::check(bmx_count_op2(G().ctx.handle(), op, a.handle(), b.handle(), &c));
        *pcount = (unsigned int)c;
    } catch (const bmx::error& e) { return status_of(e); } catch (...) { return BM_ERR_BADALLOC; }
    return BM_OK;
}

int combine_op(int op, BM_BVHANDLE hdst, BM_BVHANDLE hsrc)
{
    if (!hdst || !hsrc) return BM_ERR_BADARG;
    try {
        const bmx::bvector& a = device_copy(hdst);
        const bmx::bvector& b = device_copy(hsrc);
        bmx::bvector t(G().ctx);
        switch (op) {
        case BMX_AND: t.bit_and(a, b); break;
        case BMX_OR: t.bit_or(a, b); break;
        case BMX_XOR: t.bit_xor(a, b); break;
        default: t.bit_sub(a, b); break;
        }
        bmx::download(t, *(TBM_bvector*)hdst);                   // dst OP= src, installed through the reference's blocks_manager
        G().resident.erase(hdst);                                // dst changed: its old upload is stale
    } catch (const bmx::error& e) { return status_of(e); } catch (...) { return BM_ERR_BADALLOC; }
    return BM_OK;
}
} // namespace

extern "C" {

int BMX_bvector_count(BM_BVHANDLE h, unsigned int* pcount)                           /* BM_bvector_count, libbm.h:314 */
{
    if (!h || !pcount) return BM_ERR_BADARG;
    try { *pcount = (unsigned int)device_copy(h).count(); }
    catch (const bmx::error& e) { return status_of(e); } catch (...) { return BM_ERR_BADALLOC; }
    return BM_OK;
}
int BMX_bvector_count_AND(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p) { return count_op(BMX_AND, h1, h2, p); }   /* :644 */
int BMX_bvector_count_OR(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p) { return count_op(BMX_OR, h1, h2, p); }     /* :677 */
int BMX_bvector_count_XOR(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p) { return count_op(BMX_XOR, h1, h2, p); }   /* :655 */
int BMX_bvector_count_SUB(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p) { return count_op(BMX_SUB, h1, h2, p); }   /* :666 */
int BMX_bvector_combine_AND(BM_BVHANDLE hdst, BM_BVHANDLE hsrc) { return combine_op(BMX_AND, hdst, hsrc); }           /* :439 */
int BMX_bvector_combine_OR(BM_BVHANDLE hdst, BM_BVHANDLE hsrc) { return combine_op(BMX_OR, hdst, hsrc); }             /* :450 */
int BMX_bvector_combine_XOR(BM_BVHANDLE hdst, BM_BVHANDLE hsrc) { return combine_op(BMX_XOR, hdst, hsrc); }           /* :472 */
int BMX_bvector_combine_SUB(BM_BVHANDLE hdst, BM_BVHANDLE hsrc) { return combine_op(BMX_SUB, hdst, hsrc); }           /* :461 */
/* the application changed h through the CPU API: forget its upload */
int BMX_bvector_invalidate(BM_BVHANDLE h) { if (!h) return BM_ERR_BADARG; G().resident.erase(h); return BM_OK; }
/* call before BM_bvector_free(h) */
int BMX_bvector_release(BM_BVHANDLE h) { return BMX_bvector_invalidate(h); }
int BMX_simd_version(void) { return bmx_simd_version(); }

} // extern "C"
