// bmx.hip -- C-ABI (include/bmx.h) of the MI355X-native bit-vector engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared bmx.hip -o libbmx.so
#include "bmx_internal.h"
#ifdef BMX_DIAG
#include "bmx_diag.h"
#endif
#include "bmx_kernels2.h"
#include "bmx_kernels3.h"
#include "bmx_kernels4.h"
#include "bmx_kernels5.h"
#include "bmx_kernels6.h"
#include "bmx_kernels7.h"
#include "bmx_kernels8.h"
#include "bmx_kernels9.h"
#include "bmx_kernels10.h"
#include "bmx_kernels11.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------
// error plumbing: no exception crosses the ABI (lang-maps/libbm conventions)
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

int bmx_fail_hip(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    const char* base = strrchr(file, '/');
    snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, base ? base + 1 : file, line, hipGetErrorString(e));
    g_last_error = buf;
    return e == hipErrorOutOfMemory ? BMX_ERR_BADALLOC : BMX_ERR_DEVICE;
}
void bmx_set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
static int fail_hip(hipError_t e, const char* what, int line) { return bmx_fail_hip(e, what, "bmx.hip", line); }

// The exception barrier of the C-ABI (ABI_TRY / ABI_END around every extern "C" body, bmx_internal.h): the reference's C
// wrapper guarantees that no C++ exception reaches a C caller (lang-maps/libbm/src/libbm.cpp:28-35: every body is a
// try / catch that turns std::bad_alloc into BM_ERR_BADALLOC); the host side here uses std::vector / std::map / std::string.
int bmx_abi_caught(int kind, const char* what)
{
    if (kind == 0) { g_last_error = "out of host memory (std::bad_alloc)"; return BMX_ERR_BADALLOC; }
    try { g_last_error = std::string("unexpected C++ exception inside the library: ") + (what ? what : "(not a std::exception)"); } catch (...) {}
    return BMX_ERR_DEVICE;
}
// debug fault injection (bmx_debug_inject_failure): the ABI entry `after` calls from now on this thread throws
static thread_local int g_inject_kind = 0;
static thread_local long long g_inject_after = -1;
void bmx_abi_enter()
{
    if (g_inject_after < 0) return;
    if (g_inject_after-- > 0) return;
    const int k = g_inject_kind; g_inject_kind = 0;
    if (k == 1) throw std::bad_alloc();
    if (k == 2) throw std::length_error("injected std::length_error");
    if (k == 3) throw 42;
}

static void coll_free(bmx_ctx* ctx, size_t idx);
static int vec_build_tdir(bmx_ctx* ctx, bmx_vec* v);
static void coll_drop_vector(bmx_ctx* ctx, uint64_t uid);
static bool coll_evict_one(bmx_ctx* ctx);
static int set_dev(const bmx_ctx* ctx) { HIPCHK(hipSetDevice(ctx->device)); return BMX_OK; }

static size_t pool_round(size_t bytes)
{
    if (bytes < 256) bytes = 256;
    size_t g = bytes >= (2u << 20) ? (2u << 20) : (bytes >= (64u << 10) ? (64u << 10) : 256u);
    return (bytes + g - 1) / g * g;
}

// ---------------------------------------------------------------------------
// red zones (debug, BMX_DEBUG_REDZONE=1): see bmx_ctx in bmx_internal.h
// ---------------------------------------------------------------------------
#define RZ_BYTES 4096u
#define RZ_PATTERN 0xA5u
struct RzSeg { const u8* p; u64 n; };
__global__ __launch_bounds__(256)
void k_rz_check(const RzSeg* __restrict__ seg, u32* __restrict__ bad /* per segment: damaged bytes; first damaged offset */)
{
    const RzSeg sg = seg[blockIdx.x];
    u32 cnt = 0, first = 0xFFFFFFFFu;
    for (u64 i = threadIdx.x; i < sg.n; i += 256u)
        if (sg.p[i] != (u8)RZ_PATTERN) { ++cnt; if (first == 0xFFFFFFFFu) first = i < 0xFFFFFFFEull ? (u32)i : 0xFFFFFFFEu; }
    if (cnt) { atomicAdd(&bad[2 * blockIdx.x], cnt); atomicMin(&bad[2 * blockIdx.x + 1], first); }
}
static size_t rz_user_end(size_t bytes) { return (bytes + 15u) & ~(size_t)15u; }
// paint the zones around a block of `block` bytes at raw that serves a request of `bytes`; returns the user pointer
static void* rz_arm(bmx_ctx* ctx, void* raw, size_t block, size_t bytes, int line)
{
    u8* user = (u8*)raw + RZ_BYTES;
    const size_t end = rz_user_end(bytes);
    (void)hipMemsetAsync(raw, RZ_PATTERN, RZ_BYTES, ctx->stream);
    (void)hipMemsetAsync(user + end, RZ_PATTERN, block - RZ_BYTES - end, ctx->stream);
    ctx->rz_live[user] = bmx_ctx::RzInfo{raw, block, bytes, line};
    return user;
}
// verify the zones of one allocation (user != null) or of every live one; damaged allocations are counted, described in
// ctx->rz_report (and on stderr) and repainted so that the same damage is reported once
static void rz_verify(bmx_ctx* ctx, void* user_or_null)
{
    if (!ctx->redzone || ctx->rz_live.empty()) return;
    std::vector<RzSeg> seg; std::vector<void*> who;
    auto add = [&](void* user, const bmx_ctx::RzInfo& in) {
        const size_t end = rz_user_end(in.bytes);
        seg.push_back(RzSeg{(const u8*)in.raw, RZ_BYTES}); seg.push_back(RzSeg{(const u8*)user + end, in.block - RZ_BYTES - end});
        who.push_back(user);
    };
    if (user_or_null) { auto it = ctx->rz_live.find(user_or_null); if (it == ctx->rz_live.end()) return; add(it->first, it->second); }
    else for (auto& kv : ctx->rz_live) add(kv.first, kv.second);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    RzSeg* d_seg = nullptr; u32* d_bad = nullptr;
    std::vector<u32> bad(seg.size() * 2);
    for (size_t i = 0; i < seg.size(); ++i) { bad[2 * i] = 0; bad[2 * i + 1] = 0xFFFFFFFFu; }
    if (hipMalloc((void**)&d_seg, seg.size() * sizeof(RzSeg)) != hipSuccess || hipMalloc((void**)&d_bad, bad.size() * 4) != hipSuccess) {
        (void)hipGetLastError(); if (d_seg) (void)hipFree(d_seg); return;
    }
    (void)hipMemcpy(d_seg, seg.data(), seg.size() * sizeof(RzSeg), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bad, bad.data(), bad.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rz_check, dim3((u32)seg.size()), dim3(256), 0, ctx->stream, (const RzSeg*)d_seg, d_bad);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(bad.data(), d_bad, bad.size() * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_seg); (void)hipFree(d_bad);
    for (size_t a = 0; a < who.size(); ++a) {
        const u32 front = bad[4 * a], rear = bad[4 * a + 2];
        if (!front && !rear) continue;
        const bmx_ctx::RzInfo& in = ctx->rz_live[who[a]];
        char buf[320];
        snprintf(buf, sizeof(buf), "[bmx redzone] allocation of %zu bytes made at bmx.hip:%d: %u byte(s) damaged in FRONT of it (first at -%u), %u byte(s) BEHIND it (first at +%zu past the requested size)\n",
                 in.bytes, in.line, front, front ? RZ_BYTES - bad[4 * a + 1] : 0u, rear, rear ? (size_t)bad[4 * a + 3] + (rz_user_end(in.bytes) - in.bytes) : (size_t)0);
        fputs(buf, stderr);
        ++ctx->rz_hits;
        if (ctx->rz_report.size() < 16384) ctx->rz_report += buf;
        (void)rz_arm(ctx, in.raw, in.block, in.bytes, in.line);
    }
    (void)hipStreamSynchronize(ctx->stream);
}

static int dmalloc_(bmx_ctx* ctx, void** p, size_t bytes, int line);
static int dmalloc_at(bmx_ctx* ctx, void** p, size_t bytes, int line)
{
    static const bool trace = getenv("BMX_TRACE_ALLOC") != nullptr;
    if (!trace || bytes < (64u << 20)) return dmalloc_(ctx, p, bytes, line);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t cached = ctx->pool_cached;
    int rc = dmalloc_(ctx, p, bytes, line);
    fprintf(stderr, "[bmx] dmalloc %.1f MB: %.2f ms (%s)\n", bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ctx->pool_cached < cached ? "pool" : "hipMalloc");
    return rc;
}
#define dmalloc(ctx, p, bytes) dmalloc_at((ctx), (p), (bytes), __LINE__)
static int dmalloc_(bmx_ctx* ctx, void** p, size_t bytes, int line)
{
    *p = nullptr;
    if (ctx->fail_dmalloc_after >= 0 && ctx->fail_dmalloc_after-- == 0) {      // debug fault injection
        g_last_error = "injected device allocation failure"; return BMX_ERR_BADALLOC;
    }
    size_t sz = pool_round(ctx->redzone ? rz_user_end(bytes) + 2u * RZ_BYTES : bytes);
    auto it = ctx->pool_free.lower_bound(sz);
    if (it != ctx->pool_free.end() && it->first <= sz + sz / 4) {          // best fit within 25 % slack
        *p = it->second; sz = it->first;
        ctx->pool_cached -= sz;
        ctx->pool_free.erase(it);
    } else {
        hipError_t e = hipMalloc(p, sz);
        if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {          // give the cache back and retry
            (void)hipGetLastError();
            if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);        // (a pooled block may still be read by enqueued work)
            for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
            ctx->pool_free.clear(); ctx->pool_cached = 0;
            e = hipMalloc(p, sz);
        }
        // packed collections are copies the library made for speed: under memory pressure they go, least recently used first
        while (e == hipErrorOutOfMemory && coll_evict_one(ctx)) { (void)hipGetLastError(); e = hipMalloc(p, sz); }
        if (e != hipSuccess) return fail_hip(e, "hipMalloc", __LINE__);
    }
    if (ctx->redzone) *p = rz_arm(ctx, *p, sz, bytes, line);
    ctx->pool_live[*p] = sz;
    ctx->mem_used += sz;
    return BMX_OK;
}

// Work enqueued on the context's stream may still use p: a pooled block is only ever handed to work that is enqueued behind
// it on the same stream, and a block that really leaves (pool full, foreign pointer) waits for the stream first.
static void dfree(bmx_ctx* ctx, void* p)
{
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) { if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); (void)hipFree(p); return; }
    size_t sz = it->second;
    ctx->pool_live.erase(it);
    ctx->mem_used -= std::min<uint64_t>(ctx->mem_used, sz);
    if (ctx->redzone) {
        auto rz = ctx->rz_live.find(p);
        if (rz != ctx->rz_live.end()) { rz_verify(ctx, p); p = rz->second.raw; ctx->rz_live.erase(rz); }
    }
    if (ctx->pool_cached + sz <= ctx->pool_cap) { ctx->pool_free.emplace(sz, p); ctx->pool_cached += sz; }
    else { if (ctx->stream) (void)hipStreamSynchronize(ctx->stream); (void)hipFree(p); }
}

static void pool_trim(bmx_ctx* ctx)
{
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear(); ctx->pool_cached = 0;
}

// the grow-only buffers (scratch, aux)
static void ensure_release(bmx_ctx* ctx, void** buf, size_t* cur)
{
    if (!*buf) return;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void* raw = *buf;
    auto rz = ctx->rz_live.find(*buf);
    if (rz != ctx->rz_live.end()) { rz_verify(ctx, *buf); raw = rz->second.raw; ctx->rz_live.erase(rz); }
    (void)hipFree(raw); *buf = nullptr; *cur = 0;
}
static int ensure_at(bmx_ctx* ctx, void** buf, size_t* cur, size_t need, int line)
{
    if (*cur >= need) return BMX_OK;
    ensure_release(ctx, buf, cur);
    if (ctx->fail_dmalloc_after >= 0 && ctx->fail_dmalloc_after-- == 0) { g_last_error = "injected device allocation failure"; return BMX_ERR_BADALLOC; }
    if (ctx->redzone) {
        const size_t block = rz_user_end(need) + 2u * RZ_BYTES;
        void* raw = nullptr;
        HIPCHK(hipMalloc(&raw, block));
        *buf = rz_arm(ctx, raw, block, need, line);
    }
    else HIPCHK(hipMalloc(buf, need));
    *cur = need;
    return BMX_OK;
}
#define ensure(ctx, buf, cur, need) ensure_at((ctx), (buf), (cur), (need), __LINE__)

// ---------------------------------------------------------------------------
// column-major packed GAP collections (bmx_kernels6.h, member directory bmx_kernels8.h)
// ---------------------------------------------------------------------------
static void coll_free(bmx_ctx* ctx, size_t idx)
{
    bmx_coll* c = ctx->colls[idx];
    dfree(ctx, c->d_runs); dfree(ctx, c->d_off); dfree(ctx, c->d_cnt); dfree(ctx, c->d_flags); dfree(ctx, c->d_cnt_s);
    dfree(ctx, c->d_dir); dfree(ctx, c->d_dir_s); dfree(ctx, c->d_bt);
    ctx->pack_bytes -= std::min<uint64_t>(ctx->pack_bytes, c->bytes);
    ctx->colls.erase(ctx->colls.begin() + (long)idx);
    ++ctx->coll_gen;                                       // pipelines that resolved their groups against a collection look again
    delete c->index;
    delete c;
}

// a vector is going away (the caller has synchronised the stream): every collection that holds its runs goes with it
static void coll_drop_vector(bmx_ctx* ctx, uint64_t uid)
{
    for (size_t i = ctx->colls.size(); i-- > 0;)
        if (ctx->colls[i]->index->count(uid)) coll_free(ctx, i);
}

// device memory is short: the least recently used collection goes (the caller retries its allocation); false = none left
static bool coll_evict_one(bmx_ctx* ctx)
{
    if (ctx->colls.empty() || ctx->coll_building) return false;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;    // (nothing may still read it)
    // (a collection a running call resolved and still holds a pointer to is pinned: bmx_agg_and_sub / agg_or_impl allocate
    // their result slab and member tables AFTER resolving)
    size_t lru = ctx->colls.size();
    for (size_t i = 0; i < ctx->colls.size(); ++i)
        if (!ctx->colls[i]->pins && (lru == ctx->colls.size() || ctx->colls[i]->last_use < ctx->colls[lru]->last_use)) lru = i;
    if (lru == ctx->colls.size()) return false;
    coll_free(ctx, lru);
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);             // its blocks went to the pool: give them to the driver
    ctx->pool_free.clear(); ctx->pool_cached = 0;
    return true;
}

// holds collections against eviction for the duration of a scope (dmalloc under memory pressure evicts the least recently
// used UNPINNED collection)
struct CollPin {
    bmx_coll* c[2] = {nullptr, nullptr};
    void pin(bmx_coll* a, bmx_coll* b = nullptr) { release(); c[0] = a; c[1] = b; for (bmx_coll* x : c) if (x) ++x->pins; }
    void release() { for (bmx_coll*& x : c) if (x) { --x->pins; x = nullptr; } }
    ~CollPin() { release(); }
};

// may this operand list be packed at all?  GAP / NULL / FULL blocks only (and at least one GAP block)
static bool coll_packable(const bmx_ctx* ctx, const bmx_vec* const* v, size_t n)
{
    bool any_gap = false;
    for (size_t i = 0; i < n; ++i) {
        if (!v[i] || v[i]->ctx != ctx || v[i]->counts[BMX_BIT] != 0) return false;
        any_gap |= v[i]->counts[BMX_GAP] != 0;
    }
    return any_gap;
}

// A collection of the wanted polarity that holds EVERY vector of the list (any order, repeats allowed: the unions are
// idempotent).  members[k] = member index of list element k; full = the list names every member, i.e. the whole column
// regions can be streamed.  Among several covering collections the one the list fills completely wins, then the smallest.
static bmx_coll* coll_cover(bmx_ctx* ctx, const uint64_t* uids, size_t n, int polarity, std::vector<u32>* members, bool* full)
{
    bmx_coll* best = nullptr; bool best_full = false;
    std::vector<u32> idx(n);
    for (bmx_coll* c : ctx->colls) {
        if (c->polarity != polarity || !n || c->has_bit) continue;
        if (!c->index->count(uids[0])) continue;
        bool all = true;
        for (size_t i = 0; i < n && all; ++i) {
            auto it = c->index->find(uids[i]);
            if (it == c->index->end()) all = false; else idx[i] = it->second;
        }
        if (!all) continue;
        bool f = false;
        if (n >= c->index->size()) {                       // names every distinct member?
            std::vector<u8> seen(c->nvec, 0); size_t distinct = 0;
            for (size_t i = 0; i < n; ++i) if (!seen[idx[i]]) { seen[idx[i]] = 1; ++distinct; }
            f = distinct == c->index->size();
        }
        if (!best || (f && !best_full) || (f == best_full && c->nvec < best->nvec)) {
            best = c; best_full = f;
            if (members) *members = idx;
        }
    }
    if (best) { best->last_use = ++ctx->coll_tick; if (full) *full = best_full; }
    return best;
}

// transposes the GAP blocks of the operand set into column-major interval bags.  Everything runs on the context's stream.
static int coll_build(bmx_ctx* ctx, const bmx_vec* const* v, size_t n, int polarity, bmx_coll** out, const bmx_coll* keep)
{
    *out = nullptr;
    int rc;
    uint32_t ncols = 0; uint64_t alg = 0;
    std::vector<const u64*> descs(n); std::vector<uint32_t> nblk(n);
    for (size_t i = 0; i < n; ++i) {
        descs[i] = v[i]->d_desc; nblk[i] = v[i]->nblocks; ncols = std::max(ncols, v[i]->nblocks);
        alg += 2ull * v[i]->gap_words;                    // (device slabs pad blocks to 16 B: an upper bound, refined below)
    }
    if (!ncols || (ncols + 3u) / 4u > 65535u) return BMX_OK;        // (grid.y of the scatter pass; longer vectors keep the table kernels)
    bmx_coll* c = new (std::nothrow) bmx_coll();
    if (!c) return BMX_ERR_BADALLOC;
    c->polarity = polarity; c->ncols = ncols; c->nvec = (uint32_t)n;
    c->d_runs = nullptr; c->d_off = nullptr; c->d_cnt = nullptr; c->d_flags = nullptr; c->d_cnt_s = nullptr; c->d_dir = nullptr; c->d_dir_s = nullptr;
    c->entries = 0; c->bytes = 0; c->has_bit = false; c->build_ms = 0.f; c->alg_bytes = alg; c->prepared = false;
    c->key.resize(n);
    c->index = new (std::nothrow) std::unordered_map<uint64_t, uint32_t>();
    if (!c->index) { delete c; return BMX_ERR_BADALLOC; }
    for (size_t i = 0; i < n; ++i) { c->key[i] = v[i]->uid; c->index->emplace(v[i]->uid, (uint32_t)i); }      // (a repeated vector keeps its first index)
    struct BuildGuard { bmx_ctx* c; BuildGuard(bmx_ctx* x) : c(x) { ++c->coll_building; } ~BuildGuard() { --c->coll_building; } } guard(ctx);   // (no eviction from under a build)
    void* d_descs = nullptr; void* d_nblk = nullptr; u32* d_pre = nullptr; u32* d_sgl = nullptr; u32* d_words = nullptr;
    void* d_optab = nullptr; u32* d_bt = nullptr;
    const bool split = polarity == 1 && ctx->coll_split != 0;
    // sparse operands, OR / SUB role: the tile build (bmx_kernels10.h).  coll_build: -1 = where the operands average <= 4.1
    // 16-byte chunks per GAP block (the rows of 14 columns fit one wave load nearly always), 0 = never, 1 = whenever possible
    bool tiles = false;
    if (split && ctx->coll_build != 0 && n <= C2_MAX_N) {
        uint64_t gw = 0, gb = 0;
        for (size_t i = 0; i < n; ++i) { gw += v[i]->gap_words; gb += v[i]->counts[BMX_GAP]; }
        tiles = ctx->coll_build == 1 || (gb && gw * 10ull <= gb * 328ull);
    }
    u64 total = 0;
    const size_t dir_bytes = ((size_t)n + 1) * ncols * 4;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_descs); dfree(ctx, d_nblk); dfree(ctx, d_pre); dfree(ctx, d_sgl); dfree(ctx, d_words); dfree(ctx, d_optab); dfree(ctx, d_bt);
        dfree(ctx, c->d_runs); dfree(ctx, c->d_off); dfree(ctx, c->d_cnt); dfree(ctx, c->d_flags); dfree(ctx, c->d_cnt_s);
        dfree(ctx, c->d_dir); dfree(ctx, c->d_dir_s);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        delete c->index;
        delete c;
        return code;
    };
    if (tiles) {
        // ---- the tile build: two passes over the run lists, a workgroup per tile of 14 columns (bmx_kernels10.h) ----
        const u32 ntiles = (ncols + ORR_TILE - 1u) / ORR_TILE, ngroups = ((u32)n + C2_GROUP - 1u) / C2_GROUP;
        std::vector<u64> tab(n * 4, 0ull);
        for (size_t i = 0; i < n; ++i) {
            const bmx_vec* o = v[i];
            if (!o->d_tdir && (rc = vec_build_tdir(ctx, const_cast<bmx_vec*>(o)))) return fail(rc);      // (a cache of the immutable vector's layout)
            tab[i * 4] = (u64)(uintptr_t)o->d_tdir; tab[i * 4 + 1] = (u64)(uintptr_t)o->d_gaps;
            tab[i * 4 + 2] = (u64)(uintptr_t)o->d_desc; tab[i * 4 + 3] = (u64)o->nblocks;
        }
        if ((rc = dmalloc(ctx, &d_optab, std::max<size_t>(tab.size() * 8, 64))) ||
            (rc = dmalloc(ctx, (void**)&d_bt, (size_t)ntiles * ngroups * 16 * 4)) || (rc = dmalloc(ctx, (void**)&d_words, (size_t)ncols * 4)) ||
            (rc = dmalloc(ctx, (void**)&c->d_off, ((size_t)ncols + 1) * 8)) || (rc = dmalloc(ctx, (void**)&c->d_cnt, (size_t)ncols * 4)) ||
            (rc = dmalloc(ctx, (void**)&c->d_flags, (size_t)ncols * 4)) || (rc = dmalloc(ctx, (void**)&c->d_cnt_s, (size_t)ncols * 4))) return fail(rc);
        // (the member directory -- 8 B per member and column, 2 GB for configs[4] -- is built by coll_ensure_dir when a call that
        // names only some of the members first needs it: a list of all members streams the column regions without it)
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_optab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(fail_hip(e, "coll_build (tiles)", __LINE__));
        hipLaunchKernelGGL(k_coll2_count, dim3(ntiles), dim3(1024), 0, ctx->stream, (const u32x4*)d_optab, (u32)n, ncols, ngroups, ctx->xcd_swz,
                           C2CountOut{c->d_cnt, c->d_cnt_s, c->d_flags, d_words, d_bt});
        hipLaunchKernelGGL(k_coll_offsets, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)d_words, ncols, c->d_off);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&total, c->d_off + ncols, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // (the operand table came from pageable memory: done too)
        float ms_count = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_count, e0, e1);
        if (e != hipSuccess) return fail(fail_hip(e, "coll_build (tile count)", __LINE__));
        c->entries = total;
        // (build_ms is the device time of the two passes: the allocation between them -- gigabytes the driver may have to map and
        // clear, 0.1 to 120 ms on the boxes of this pool -- is host time the caller sees in the call's wall time)
        if ((rc = dmalloc(ctx, (void**)&c->d_runs, std::max<size_t>((size_t)total * 4, 64)))) return fail(rc);
        e = hipEventRecord(e0, ctx->stream);
        if (e != hipSuccess) return fail(fail_hip(e, "coll_build (tile scatter)", __LINE__));
        const size_t lds = (size_t)ngroups * 16 * 4 * 2;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_coll2_scatter<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll2_scatter<8, 4>), dim3(ntiles), dim3(256), lds, ctx->stream, (const u32x4*)d_optab, (u32)n, ncols, ngroups, ctx->xcd_swz,
                               (const u32*)d_bt, (const u64*)c->d_off, (const u32*)c->d_cnt, (const u32*)c->d_cnt_s, c->d_runs, (u32*)nullptr, (u32*)nullptr, C2_SKIP_DIR);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&c->build_ms, e0, e1);
        if (e != hipSuccess) return fail(fail_hip(e, "coll_build (tile scatter)", __LINE__));
        c->build_ms += ms_count;
        c->d_bt = d_bt; d_bt = nullptr; c->dir_pending = true;
    } else {
    if ((rc = dmalloc(ctx, &d_descs, n * 8)) || (rc = dmalloc(ctx, &d_nblk, n * 4)) ||
        (rc = dmalloc(ctx, (void**)&d_pre, (size_t)n * ncols * 4)) ||
        (rc = dmalloc(ctx, (void**)&c->d_off, ((size_t)ncols + 1) * 8)) || (rc = dmalloc(ctx, (void**)&c->d_cnt, (size_t)ncols * 4)) ||
        (rc = dmalloc(ctx, (void**)&c->d_flags, (size_t)ncols * 4))) return fail(rc);
    if (split && ((rc = dmalloc(ctx, (void**)&d_sgl, (size_t)n * ncols * 4)) || (rc = dmalloc(ctx, (void**)&d_words, (size_t)ncols * 4)) ||
                  (rc = dmalloc(ctx, (void**)&c->d_cnt_s, (size_t)ncols * 4)))) return fail(rc);
    // lanes per block in the passes that walk run lists: by the average block length
    uint64_t nblocks_gap = 0, gap_words_all = 0;
    for (size_t i = 0; i < n; ++i) { nblocks_gap += v[i]->counts[BMX_GAP]; gap_words_all += v[i]->gap_words; }
    const bool short_blocks = nblocks_gap && gap_words_all / nblocks_gap <= 56;          // (<= ~24 runs of one polarity per block)
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return fail(fail_hip(e, "coll_build", __LINE__));
    hipLaunchKernelGGL(k_coll_count, dim3((ncols + 255) / 256), dim3(256), 0, ctx->stream, (const u64* const*)d_descs,
                       (const u32*)d_nblk, (u32)n, ncols, (u32)polarity, d_pre, c->d_cnt, c->d_flags);
    if (split) {
        // single-bit runs per (operand, column) -> their prefix per column -> the column's size in 32-bit words
        if (short_blocks) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_count_singles<8>), dim3((u32)n, ((ncols + 31) / 32 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                             (const u64* const*)d_descs, (const u32*)d_nblk, ncols, d_sgl);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_count_singles<64>), dim3((u32)n, ((ncols + 3) / 4 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                (const u64* const*)d_descs, (const u32*)d_nblk, ncols, d_sgl);
        hipLaunchKernelGGL(k_coll_prefix_singles, dim3((ncols + 255) / 256), dim3(256), 0, ctx->stream, d_sgl, (u32)n, ncols,
                           (const u32*)c->d_cnt, c->d_cnt_s, d_words);
        hipLaunchKernelGGL(k_coll_offsets, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)d_words, ncols, c->d_off);
    } else
    hipLaunchKernelGGL(k_coll_offsets, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)c->d_cnt, ncols, c->d_off);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&total, c->d_off + ncols, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // (descs / nblk are read from pageable memory: they are done too)
    float ms_count = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms_count, e0, e1);
    if (e != hipSuccess) return fail(fail_hip(e, "coll_build (count)", __LINE__));
    c->entries = total;
    // (build_ms = device time of the passes: the allocations between them are host time, see the tile build above)
    const size_t dir_b = ((size_t)n + 1) * ncols * 4;
    if ((rc = dmalloc(ctx, (void**)&c->d_runs, std::max<size_t>((size_t)total * 4, 64))) ||
        (rc = dmalloc(ctx, (void**)&c->d_dir, dir_b)) || (split && (rc = dmalloc(ctx, (void**)&c->d_dir_s, dir_b)))) return fail(rc);
    e = hipEventRecord(e0, ctx->stream);
    if (e != hipSuccess) return fail(fail_hip(e, "coll_build (scatter)", __LINE__));
    if (total && split) {
        if (short_blocks) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_scatter_split<8>), dim3((u32)n, ((ncols + 31) / 32 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                             (const u64* const*)d_descs, (const u32*)d_nblk, ncols, (const u32*)d_pre, (const u32*)d_sgl,
                                             (const u32*)c->d_cnt, (const u32*)c->d_cnt_s, (const u64*)c->d_off, c->d_runs);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_scatter_split<64>), dim3((u32)n, ((ncols + 3) / 4 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                (const u64* const*)d_descs, (const u32*)d_nblk, ncols, (const u32*)d_pre, (const u32*)d_sgl,
                                (const u32*)c->d_cnt, (const u32*)c->d_cnt_s, (const u64*)c->d_off, c->d_runs);
        e = hipGetLastError();
    } else if (total) {
        if (short_blocks) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_scatter<8>), dim3((u32)n, ((ncols + 31) / 32 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                          (const u64* const*)d_descs, (const u32*)d_nblk, ncols, (u32)polarity, (const u32*)d_pre, (const u64*)c->d_off, c->d_runs);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_scatter<64>), dim3((u32)n, ((ncols + 3) / 4 + COLL_YT - 1) / COLL_YT), dim3(256), 0, ctx->stream,
                                (const u64* const*)d_descs, (const u32*)d_nblk, ncols, (u32)polarity, (const u32*)d_pre, (const u64*)c->d_off, c->d_runs);
        e = hipGetLastError();
    }
    // the member directory (bmx_kernels8.h): the prefixes of the passes above, column-major, with the members' block kinds
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_coll_dir, dim3((ncols + 31) / 32, ((u32)n + 1 + 31) / 32), dim3(1024), 0, ctx->stream, (const u32*)d_pre, (const u32*)d_sgl,
                           (const u32*)c->d_cnt, (const u32*)c->d_cnt_s, (u32)n, ncols, c->d_dir, c->d_dir_s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&c->build_ms, e0, e1);
    if (e != hipSuccess) return fail(fail_hip(e, "coll_build (scatter)", __LINE__));
    c->build_ms += ms_count;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dfree(ctx, d_descs); dfree(ctx, d_nblk); dfree(ctx, d_pre); dfree(ctx, d_sgl); dfree(ctx, d_words); dfree(ctx, d_optab); dfree(ctx, d_bt);
    c->bytes = (uint64_t)total * 4 + (uint64_t)ncols * (split ? 20 : 16) + 8 + (c->dir_pending ? (uint64_t)((ncols + ORR_TILE - 1u) / ORR_TILE) * ((n + C2_GROUP - 1) / C2_GROUP) * 64 : (uint64_t)dir_bytes * (split ? 2 : 1));
    c->run_bytes = (uint64_t)total * 4;
    c->id = ++ctx->coll_next_id;
    c->last_use = ++ctx->coll_tick;
    ctx->last_pack_ms = c->build_ms;
    // make room: least recently used collections go first
    while (ctx->pack_bytes + c->bytes > ctx->pack_cap) {
        size_t lru = ctx->colls.size();
        for (size_t i = 0; i < ctx->colls.size(); ++i)
            if (ctx->colls[i] != keep && !ctx->colls[i]->pins && (lru == ctx->colls.size() || ctx->colls[i]->last_use < ctx->colls[lru]->last_use)) lru = i;
        if (lru == ctx->colls.size()) break;
        coll_free(ctx, lru);
    }
    ctx->colls.push_back(c);
    ctx->pack_bytes += c->bytes;
    ++ctx->coll_gen;
    *out = c;
    return BMX_OK;
}

// Which collection serves this operand list?  One that holds every vector of it (coll_cover) -- prepared by
// bmx_collection_prepare, or, with gap_pack 1, built here at the first use of a list of >= min_n packable vectors.
// *out = nullptr: none, the descriptor-table kernels take the call.  members / full as coll_cover.
static int coll_resolve(bmx_ctx* ctx, const bmx_vec* const* v, size_t n, int polarity, size_t min_n, bool may_build, bmx_coll** out,
                        std::vector<u32>* members, bool* full, const bmx_coll* keep = nullptr)
{
    *out = nullptr; *full = false;
    if (ctx->gap_pack == 0 || !n || (ctx->colls.empty() && !(ctx->gap_pack == 1 && may_build))) return BMX_OK;
    std::vector<uint64_t> uids(n);
    for (size_t i = 0; i < n; ++i) { if (!v[i] || v[i]->ctx != ctx || v[i]->counts[BMX_BIT]) return BMX_OK; uids[i] = v[i]->uid; }
    if (bmx_coll* c = coll_cover(ctx, uids.data(), n, polarity, members, full)) { *out = c; return BMX_OK; }
    if (ctx->gap_pack != 1 || !may_build || n < min_n || !coll_packable(ctx, v, n)) return BMX_OK;
    uint64_t need = 0;
    for (size_t i = 0; i < n; ++i) need += 2ull * v[i]->gap_words;
    if (need > ctx->pack_cap) return BMX_OK;
    int rc = coll_build(ctx, v, n, polarity, out, keep);
    if (!rc && *out) { *full = true; if (members) { members->resize(n); for (size_t i = 0; i < n; ++i) (*members)[i] = (*(*out)->index)[uids[i]]; } }
    return rc;
}

// (tuning build only: BMX_DIAG_COLL = 512 -> loads alone, 1024 -> no fold / store, 1536 -> both; results are then meaningless)
static int coll_diag_bits()
{
#ifdef BMX_DIAG
    if (const char* e = getenv("BMX_DIAG_COLL")) return atoi(e) & (512 | 1024);
#endif
    return 0;
}

// one workgroup per block column over [col_from, col_to); a = the AND / OR bag, s = the SUB bag (may be null)
static int coll_launch(int mode, bmx_ctx* ctx, const bmx_coll* a, const bmx_coll* s, u32 col_from, u32 col_to, int opt_compress,
                       u64* d_counts, bmx_vec* v, BlockStat* st, u32 hint_from, u32 hint_to, FoldOut kinds = FoldOut{nullptr, nullptr, nullptr})
{
    if (col_to <= col_from) return BMX_OK;
    const u32 grid_all = col_to - col_from;
#define COLL_ARGS_WG(W) dim3(grid), dim3(W), 0, ctx->stream, (const u32*)a->d_runs, (const u64*)a->d_off, (const u32*)a->d_cnt, \
        (const u32*)a->d_flags, a->ncols, (const u32*)(s ? s->d_runs : nullptr), (const u64*)(s ? s->d_off : nullptr), \
        (const u32*)(s ? s->d_cnt : nullptr), (const u32*)(s ? s->d_flags : nullptr), s ? s->ncols : 0u, cbase, col_to, opt_compress | coll_diag_bits(), \
        d_counts, v ? v->d_bits : (uint4*)nullptr, v ? v->d_desc : (u64*)nullptr, st, hint_from, hint_to, kinds, \
        (const u32*)a->d_cnt_s, (const u32*)(s ? s->d_cnt_s : nullptr)
    // coll_shape (tuning): 0 = 256 threads, 1 = 256 threads + prefetch, 2 = 512 threads, 3 = 512 threads + prefetch
#define COLL_LAUNCH(M) do { \
        switch (ctx->coll_shape) { \
        case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 256, true>), COLL_ARGS_WG(256)); break; \
        case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 512, false>), COLL_ARGS_WG(512)); break; \
        case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 512, true>), COLL_ARGS_WG(512)); break; \
        case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 512, false, true>), COLL_ARGS_WG(512)); break; \
        case 5: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 512, true, true>), COLL_ARGS_WG(512)); break; \
        default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_apply<M, 256, false>), COLL_ARGS_WG(256)); break; } } while (0)
    // launch windows (coll_window columns per launch; 0 = one launch)
    const u32 win = ctx->coll_window > 0 ? (u32)ctx->coll_window : grid_all;
    for (u32 w0 = 0; w0 < grid_all; w0 += win) {
        const u32 grid = std::min(win, grid_all - w0);
        const u32 cbase = col_from + w0;
        if (mode == COLL_OR) COLL_LAUNCH(COLL_OR);
        else if (mode == COLL_AND_STORE) COLL_LAUNCH(COLL_AND_STORE);
        else COLL_LAUNCH(COLL_AND_COUNT);
    }
#undef COLL_LAUNCH
#undef COLL_ARGS_WG
    KCHK();
    return BMX_OK;
}

// Is the member-directory path (k_coll_members) the better one for lists that name only SOME vectors of their collections?
// Measured (tools/bench_pipeline_coll.py, profiles/r04_coll): it wins where the blocks are sparse and the lists long -- 16
// groups x 257 vectors of configs[4]'s density: 11.8 against 58.7 ms -- and loses to the descriptor-table kernels where a
// member's piece of a column is a few hundred bytes or the lists are short (64 groups x 32 vectors at 0.1 %: 8.5 against 6.1 ms).
static bool coll_members_wanted(const bmx_ctx* ctx, uint64_t gap_words, uint64_t gap_blocks, uint64_t ops, uint64_t ngroups)
{
    if (ctx->coll_members >= 0) return ctx->coll_members != 0;
    return gap_blocks && gap_words <= 48ull * gap_blocks && ops >= 32ull * ngroups;
}

// the collections of an AND list + SUB list, if both lists are served by one: *a = nullptr otherwise.  full = both lists
// name their whole collection (the streaming kernel applies); else ma / ms are the member indices for k_coll_members
static int coll_resolve_and_sub(bmx_ctx* ctx, const bmx_vec* const* va, size_t na, const bmx_vec* const* vs, size_t ns,
                                bmx_coll** a, bmx_coll** s, std::vector<u32>* ma, std::vector<u32>* ms, bool* full)
{
    *a = nullptr; *s = nullptr; *full = false;
    bool fa = false, fs = true;
    int rc = coll_resolve(ctx, va, na, 0, 64, true, a, ma, &fa);
    if (rc || !*a) return rc;
    if (ns) {
        rc = coll_resolve(ctx, vs, ns, 1, 1, true, s, ms, &fs, *a);       // (built with gap_pack 1: must not evict its AND partner)
        if (rc) return rc;
        if (!*s) *a = nullptr;
    }
    *full = fa && fs;
    return BMX_OK;
}

// The member directory of a collection built through the tile directories: one more pass over the members' rows (counts and
// prefixes only: k_coll2_scatter without its run stores), the first time a call names only SOME of the members.
static int coll_ensure_dir(bmx_ctx* ctx, bmx_coll* c)
{
    if (!c || !c->dir_pending) return BMX_OK;
    int rc;
    const size_t n = c->nvec;
    const u32 ncols = c->ncols, ntiles = (ncols + ORR_TILE - 1u) / ORR_TILE, ngroups = ((u32)n + C2_GROUP - 1u) / C2_GROUP;
    std::vector<u64> tab(n * 4, 0ull);
    for (size_t i = 0; i < n; ++i) {
        auto it = ctx->live_vecs.find(c->key[i]);
        if (it == ctx->live_vecs.end()) { g_last_error = "a member of the collection is gone"; return BMX_ERR_BADARG; }
        const bmx_vec* o = it->second;
        tab[i * 4] = (u64)(uintptr_t)o->d_tdir; tab[i * 4 + 1] = (u64)(uintptr_t)o->d_gaps;
        tab[i * 4 + 2] = (u64)(uintptr_t)o->d_desc; tab[i * 4 + 3] = (u64)o->nblocks;
    }
    const size_t dir_bytes = (n + 1) * (size_t)ncols * 4;
    void* d_optab = nullptr;
    CollPin pin; pin.pin(c);                                                  // (the allocations below must not evict it)
    // the directory counts against the packing budget like the runs do: least recently used, unpinned collections make room first
    while (ctx->pack_bytes + 2ull * dir_bytes > ctx->pack_cap) {
        size_t lru = ctx->colls.size();
        for (size_t i = 0; i < ctx->colls.size(); ++i)
            if (!ctx->colls[i]->pins && (lru == ctx->colls.size() || ctx->colls[i]->last_use < ctx->colls[lru]->last_use)) lru = i;
        if (lru == ctx->colls.size()) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) break;              // (nothing may still read it)
        coll_free(ctx, lru);
    }
    if ((rc = dmalloc(ctx, &d_optab, std::max<size_t>(tab.size() * 8, 64))) || (rc = dmalloc(ctx, (void**)&c->d_dir, dir_bytes)) ||
        (rc = dmalloc(ctx, (void**)&c->d_dir_s, dir_bytes))) {
        dfree(ctx, d_optab); dfree(ctx, c->d_dir); dfree(ctx, c->d_dir_s); c->d_dir = c->d_dir_s = nullptr; return rc;
    }
    hipError_t e = hipMemcpyAsync(d_optab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    const size_t lds = (size_t)ngroups * 16 * 4 * 2;
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_coll2_scatter<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll2_scatter<8, 4>), dim3(ntiles), dim3(256), lds, ctx->stream, (const u32x4*)d_optab, (u32)n, ncols, ngroups, ctx->xcd_swz,
                           (const u32*)c->d_bt, (const u64*)c->d_off, (const u32*)c->d_cnt, (const u32*)c->d_cnt_s, c->d_runs, c->d_dir, c->d_dir_s, C2_SKIP_RUNS);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);                // (the table came from pageable memory)
    dfree(ctx, d_optab);
    if (e != hipSuccess) { dfree(ctx, c->d_dir); dfree(ctx, c->d_dir_s); c->d_dir = c->d_dir_s = nullptr; return fail_hip(e, "coll_ensure_dir", __LINE__); }
    dfree(ctx, c->d_bt); c->d_bt = nullptr; c->dir_pending = false;
    c->bytes += 2ull * dir_bytes; ctx->pack_bytes += 2ull * dir_bytes;
    return BMX_OK;
}

static CollView coll_view(const bmx_coll* c)
{
    if (!c) return CollView{nullptr, nullptr, nullptr, nullptr, 0u, 0u};
    return CollView{c->d_runs, c->d_off, c->d_dir, c->d_dir_s, c->nvec, c->ncols};
}

// k_coll_members over [col_from, col_to): d_midx / d_groups are device arrays (member indices, arg-groups)
static int coll_members_launch(int mode, bmx_ctx* ctx, const bmx_coll* a, const bmx_coll* s, const u32* d_midx, const CollGroup* d_groups, u32 ngroups,
                               u32 col_from, u32 col_to, int opt_compress, u64* d_counts, bmx_vec* v, BlockStat* st)
{
    if (col_to <= col_from) return BMX_OK;
    // both collections stay for the whole call: building the directory of one allocates GBs, and an allocation under memory
    // pressure evicts the least recently used UNPINNED collection -- which must not be the other one (the pipeline callers hold
    // a and s unpinned)
    CollPin pin_both; pin_both.pin(const_cast<bmx_coll*>(a), const_cast<bmx_coll*>(s));
    { int rce; if ((rce = coll_ensure_dir(ctx, const_cast<bmx_coll*>(a))) || (rce = coll_ensure_dir(ctx, const_cast<bmx_coll*>(s)))) return rce; }
    const u64 nitems = (u64)(col_to - col_from) * ngroups;
    if ((nitems + CM_WAVES - 1) / CM_WAVES > 0x7FFFFFFFull) { g_last_error = "too many (column, group) items in one run"; return BMX_ERR_RANGE; }
#define CM_ARGS dim3((u32)((nitems + CM_WAVES - 1) / CM_WAVES)), dim3(CM_WAVES * 64), 0, ctx->stream, coll_view(a), coll_view(s), d_midx, d_groups, ngroups, col_from, col_to, opt_compress, \
        d_counts, v ? v->d_bits : (uint4*)nullptr, v ? v->d_desc : (u64*)nullptr, st
    if (mode == CM_OR_STORE) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_members<CM_OR_STORE>), CM_ARGS);
    else if (mode == CM_AND_STORE) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_members<CM_AND_STORE>), CM_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_coll_members<CM_AND_COUNT>), CM_ARGS);
#undef CM_ARGS
    KCHK();
    return BMX_OK;
}

// one arg-group for a one-shot call: member indices (A list, then S list) + the group record, staged to the device
static int coll_members_upload(bmx_ctx* ctx, const std::vector<u32>& ma, const std::vector<u32>& ms, void** d_buf, const u32** d_midx, const CollGroup** d_groups);

// Small host tables (operand pointer lists, pipeline metadata) go through a pinned ring: the copy is truly
// asynchronous, the caller's buffer may die on return, and nobody has to synchronise the stream for it.  A region of
// the ring is reused only after the event recorded behind the last copy has completed.
#define STAGE_BYTES (1u << 20)
static int h2d_staged(bmx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return BMX_OK;
    if (bytes > STAGE_BYTES / 4) {                      // big table: plain copy, wait for it
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return BMX_OK;
    }
    size_t need = (bytes + 63u) & ~(size_t)63u;
    if (ctx->stage_off + need > STAGE_BYTES) { HIPCHK(hipEventSynchronize(ctx->ev_stage)); ctx->stage_off = 0; }
    char* s = ctx->h_stage + ctx->stage_off;
    memcpy(s, src, bytes);
    ctx->stage_off += need;
    HIPCHK(hipMemcpyAsync(dst, s, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_stage, ctx->stream));
    return BMX_OK;
}

static int coll_members_upload(bmx_ctx* ctx, const std::vector<u32>& ma, const std::vector<u32>& ms, void** d_buf, const u32** d_midx, const CollGroup** d_groups)
{
    const size_t nm = ma.size() + ms.size(), goff = (nm * 4 + 15) & ~(size_t)15;
    std::vector<u8> h(goff + sizeof(CollGroup));
    if (!ma.empty()) memcpy(h.data(), ma.data(), ma.size() * 4);
    if (!ms.empty()) memcpy(h.data() + ma.size() * 4, ms.data(), ms.size() * 4);
    const CollGroup g{0u, (u32)ma.size(), (u32)ma.size(), (u32)ms.size()};
    memcpy(h.data() + goff, &g, sizeof(g));
    *d_buf = nullptr;
    int rc;
    if ((rc = dmalloc(ctx, d_buf, h.size())) || (rc = h2d_staged(ctx, *d_buf, h.data(), h.size()))) { dfree(ctx, *d_buf); *d_buf = nullptr; return rc; }
    *d_midx = (const u32*)*d_buf; *d_groups = (const CollGroup*)((const char*)*d_buf + goff);
    return BMX_OK;
}

// few (column, group) items with long operand lists: a workgroup of 8 waves per item (k_pipe_split)
static bool use_split(const bmx_ctx* ctx, const bmx_pipeline* p, u64 nitems)
{
    if (ctx->pipe_split == 0 || !nitems) return false;
    if (ctx->pipe_split == 1) return nitems <= 65535u * 16u;
    return nitems <= 384u && (u64)p->n_ops >= 24ull * p->ngroups;
}
#define SPLIT_WAVES 8

// ---- launch shapes of the bit-only counts kernel --------------------------------------------------
typedef void (*pipe_bits_fn)(const u64*, const u32*, const u32*, u32, u32, u32, u32, int, u64*);

// Launch plan of the bit-only counts kernel for a run of nitems (column, group) items of whole blocks.
// One WINDOW = one machine-load of waves: 256 CUs x the waves one CU holds at this workgroup size.  Measured on
// the headline (tools/tune_pipe.py, profiles/r02): one workgroup of 8-12 waves per CU, windows cut evenly --
//   wg 640 / window 2560: 86.2 %, wg 576: 86.2 %, wg 512 / 2048: 85.7 %, wg 768 / 3072: 85.3 %, wg 256 / 2048: 84.9 %
//   of the HBM peak, against 79.1 % for the single launch (wg 384) on the same box.
// The shape is picked by how evenly the run fills its windows (a 1,908-column shard of an 8-GPU job fills 93 %
// of a 2,048-wave window and runs at 84 %; 3,815 columns -> 2 x 1,908 at 85.7 %).  Runs much shorter than a
// window are cut into smaller slices (rows < 8) so the chip still sees >= ~1.4 k waves.
// Explicit knobs override: pipe_rows, pipe_wg, pipe_unroll, pipe_window (-1 = single launch).
static u32 pipe_window_cap(u32 wg)
{
    u32 wpb = wg / 64u;
    u32 per_cu = wpb >= 7u ? 1u : 12u / wpb;                          // 147 VGPRs: 12 waves per CU
    return 256u * per_cu * wpb;
}
static void pipe_plan(const bmx_ctx* ctx, u64 nitems, u32 ngroups, u32& rows, u32& wg, u32& unroll, u32& window)
{
    rows = (u32)ctx->pipe_rows;
    if (!rows) rows = nitems >= 1400u ? 8u : nitems >= 700u ? 4u : nitems >= 350u ? 2u : 1u;
    unroll = ctx->pipe_unroll ? (u32)ctx->pipe_unroll : ((rows >= 4u || !ctx->pipe_nt) ? 4u : 8u);      // (plain loads: four slices in flight is the only compiled shape)
    u64 n = nitems * (8u / rows);
    wg = (u32)ctx->pipe_wg;
    if (!wg) {
        wg = 256u;
        if (rows == 8u) {
            static const u32 cand[3] = {640u, 768u, 512u};               // preference order on equal fill
            double best = -1.0;
            for (u32 c : cand) {
                u64 cap = pipe_window_cap(c), nwin = (n + cap - 1) / cap, per = (n + nwin - 1) / nwin;
                double fill = (double)per / (double)cap;
                if (fill > best + 0.02) { best = fill; wg = c; }
            }
        }
    }
    if (!ctx->pipe_nt && !ctx->pipe_wg && rows == 8u && wg > 512u) wg = 512u;               // (plain loads: the 640 / 768-thread shapes exist with non-temporal loads only)
    if (ctx->pipe_window < 0 || (ctx->pipe_window == 0 && ngroups > 4u)) window = 0u;     // many groups: operand re-use in L2 is what matters
    else if (ctx->pipe_window > 0) window = (u32)ctx->pipe_window;
    else window = (u32)std::max<u64>(pipe_window_cap(wg) / ((u64)ngroups * (8u / rows)), 1u);
}

template <int ROWS>
static pipe_bits_fn pipe_bits_rows(u32 unroll, bool nt, u32 wg)
{
#define B2(U, NT, WG) (pipe_bits_fn)k_pipe_counts_bits2<U, NT, WG, ROWS>
    if (wg == 256 && nt) {
        if (unroll == 4) return B2(4, true, 256);
        if constexpr (ROWS <= 4) { if (unroll == 8) return B2(8, true, 256); }
    }
    if (wg == 256 && !nt && unroll == 4) return B2(4, false, 256);
    if constexpr (ROWS == 8) {
        if (unroll == 4 && nt) switch (wg) { case 384: return B2(4, true, 384); case 512: return B2(4, true, 512);
                                             case 640: return B2(4, true, 640); case 768: return B2(4, true, 768); default: break; }
        if (unroll == 4 && !nt && wg == 512) return B2(4, false, 512);      // (plain loads need ~196 VGPRs: two waves per SIMD, i.e. <= 512 threads; <4, false, 640> spilled)
    }
#ifdef BMX_TUNE   // shapes of the tuning sweeps (tools/tune_pipe.py): make -C bitmagic_amd/csrc tune
    if constexpr (ROWS == 8) {
        if (nt && unroll == 4) switch (wg) { case 192: return B2(4, true, 192); case 320: return B2(4, true, 320); case 448: return B2(4, true, 448);
                                             case 576: return B2(4, true, 576); default: break; }
        if (nt && unroll == 2) switch (wg) { case 256: return B2(2, true, 256); case 384: return B2(2, true, 384); case 640: return B2(2, true, 640);
                                             case 1024: return B2(2, true, 1024); default: break; }
        if (nt && unroll == 1 && wg == 256) return B2(1, true, 256);
    } else {
        if (nt && wg == 384) {
            if (unroll == 4) return B2(4, true, 384);
            if constexpr (ROWS <= 4) { if (unroll == 8) return B2(8, true, 384); }
            if constexpr (ROWS <= 2) { if (unroll == 16) return B2(16, true, 384); }
        }
        if (nt && wg == 512 && unroll == 4) return B2(4, true, 512);
    }
#endif
#undef B2
    return nullptr;
}
static pipe_bits_fn pipe_bits_kernel(u32 rows, u32 unroll, bool nt, u32 wg)
{
    switch (rows) {
    case 8: return pipe_bits_rows<8>(unroll, nt, wg);
    case 4: return pipe_bits_rows<4>(unroll, nt, wg);
    case 2: return pipe_bits_rows<2>(unroll, nt, wg);
    case 1: return pipe_bits_rows<1>(unroll, nt, wg);
    default: return nullptr;
    }
}

extern "C" {

const char* bmx_error_msg(int status)
{
    switch (status) {
    case BMX_OK: return "BMX-00: All correct";
    case BMX_ERR_BADALLOC: return "BMX-01: Allocation error (HBM or host)";
    case BMX_ERR_BADARG: return "BMX-02: Invalid or missing function argument";
    case BMX_ERR_RANGE: return "BMX-03: Incorrect range or index";
    case BMX_ERR_DEVICE: return "BMX-04: No usable gfx950 device or HIP runtime failure";
    default: return "BMX-XX: Unknown error";
    }
}
const char* bmx_last_error(void) { return g_last_error.c_str(); }
int bmx_simd_version(void) { return 950; }

int bmx_device_count(int* n)
{ ABI_TRY
    ARGCHK(n);
    *n = 0;
    HIPCHK(hipGetDeviceCount(n));
    return BMX_OK;
ABI_END }

int bmx_ctx_create(int device, void* stream, bmx_ctx** out)
{ ABI_TRY
    ARGCHK(out);
    *out = nullptr;
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { g_last_error = "device index out of range"; return BMX_ERR_RANGE; }
    HIPCHK(hipSetDevice(device));
    bmx_ctx* ctx = new (std::nothrow) bmx_ctx();
    if (!ctx) return BMX_ERR_BADALLOC;
    ctx->device = device;
#define CTXCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_ctx_destroy(ctx); return r_; } } while (0)
    if (stream) ctx->stream = (hipStream_t)stream;
    else { CTXCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    CTXCHK(hipEventCreate(&ctx->ev0));
    CTXCHK(hipEventCreate(&ctx->ev1));
    CTXCHK(hipMalloc((void**)&ctx->d_small, 64 * sizeof(u64)));
    CTXCHK(hipHostMalloc((void**)&ctx->h_small, 64 * sizeof(u64)));
    CTXCHK(hipHostMalloc((void**)&ctx->h_stage, STAGE_BYTES));
    CTXCHK(hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming));
    CTXCHK(hipMalloc((void**)&ctx->d_slots, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64)));
    CTXCHK(hipMemsetAsync(ctx->d_slots, 0, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64), ctx->stream));
    CTXCHK(hipMalloc((void**)&ctx->d_zero, 256));
    CTXCHK(hipMemsetAsync(ctx->d_zero, 0, 256, ctx->stream));
    CTXCHK(hipMalloc((void**)&ctx->d_done, FOLD_DONE_WORDS * 4));
    CTXCHK(hipMemsetAsync(ctx->d_done, 0, FOLD_DONE_WORDS * 4, ctx->stream));
    CTXCHK(hipMalloc((void**)&ctx->d_slots2, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64)));
    CTXCHK(hipMemsetAsync(ctx->d_slots2, 0, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64), ctx->stream));
    CTXCHK(hipMalloc((void**)&ctx->d_done2, FOLD_DONE_WORDS * 4));
    CTXCHK(hipMemsetAsync(ctx->d_done2, 0, FOLD_DONE_WORDS * 4, ctx->stream));
    CTXCHK(hipHostMalloc((void**)&ctx->h_pend, PEND_SLOTS * 8 * sizeof(u64)));
    CTXCHK(hipMalloc((void**)&ctx->d_cursor, 64));
    CTXCHK(hipMemsetAsync(ctx->d_cursor, 0, 64, ctx->stream));
#undef CTXCHK
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) ctx->max_lds_bytes = (uint32_t)v; else (void)hipGetLastError(); }
    { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr) ctx->pack_cap = (uint64_t)fr / 4; else (void)hipGetLastError(); }   // packed copies: at most a quarter of what is free now
    if (const char* e = getenv("BMX_DEBUG_REDZONE")) ctx->redzone = atoi(e) != 0;
    if (const char* e = getenv("BMX_PACK_MAX_MB")) { long long mb = atoll(e); if (mb >= 0) ctx->pack_cap = (uint64_t)mb << 20; }
    if (const char* e = getenv("BMX_POOL_MAX_MB")) { long long mb = atoll(e); if (mb >= 0) ctx->pool_cap = (uint64_t)mb << 20; }
    // launch-shape knobs from the environment go through the same validation as bmx_ctx_set_tuning;
    // an invalid value is ignored (the default stays)
    static const char* const env_keys[][2] = {
        {"BMX_PIPE_UNROLL", "pipe_unroll"}, {"BMX_PIPE_ROWS", "pipe_rows"}, {"BMX_PIPE_NT", "pipe_nt"},
        {"BMX_PIPE_WG", "pipe_wg"}, {"BMX_PIPE_WINDOW", "pipe_window"}, {"BMX_PIPE_SPLIT", "pipe_split"}, {"BMX_OR_TILE", "or_tile"}, {"BMX_OR_ROWS", "or_rows"}, {"BMX_OR_DEPTH", "or_depth"}, {"BMX_OR_WINDOW", "or_window"}, {"BMX_DIRECT_COLS", "direct_cols"}, {"BMX_FF_WINDOW", "ff_window"}, {"BMX_GAP_COUNT", "gap_count"}, {"BMX_AND_ROWS", "and_rows"}, {"BMX_AGG_SHAPE", "agg_shape"}, {"BMX_AND_ROWS_WG", "and_rows_wg"}, {"BMX_AND_ROWS_DEPTH", "and_rows_depth"}, {"BMX_AND_ROWS_NT", "and_rows_nt"}, {"BMX_AND_ROWS_IPW", "and_rows_ipw"}, {"BMX_RANGE_HALVES", "range_halves"}, {"BMX_PAIR_STREAM", "pair_stream"}, {"BMX_PAIR_WGS", "pair_wgs"}, {"BMX_RS_LANES", "rs_lanes"}, {"BMX_RS_SELECT_TOP", "rs_select_top"}, {"BMX_RS_SELECT_SEL", "rs_select_sel"}, {"BMX_RS_LINES", "rs_lines"}, {"BMX_RS_SELECT_LINES", "rs_select_lines"}, {"BMX_RS_SDIR_SHIFT", "rs_sdir_shift"}, {"BMX_COLL_SHAPE", "coll_shape"}, {"BMX_COLL_WINDOW", "coll_window"}, {"BMX_COLL_SPLIT", "coll_split"}, {"BMX_COLL_BUILD", "coll_build"}, {"BMX_EQ_BIG", "eq_big"}, {"BMX_PAIR_LOOP", "pair_loop"}, {"BMX_PAIR_NT", "pair_nt"}, {"BMX_EQ_BIG_SHAPE", "eq_big_shape"}, {"BMX_OP2_WGS", "op2_wgs"}, {"BMX_OP2_LOOP", "op2_loop"}, {"BMX_OP2_NT", "op2_nt"}, {"BMX_GAP_PACK", "gap_pack"}, {"BMX_COLL_MEMBERS", "coll_members"}, {"BMX_XCD_SWIZZLE", "xcd_swizzle"}};
    for (auto& kv : env_keys)
        if (const char* e = getenv(kv[0])) (void)bmx_ctx_set_tuning(ctx, kv[1], atoi(e));
    g_last_error.clear();
    *out = ctx;
    return BMX_OK;
ABI_END }

int bmx_ctx_destroy(bmx_ctx* ctx)
{ ABI_TRY
    if (!ctx) return BMX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    while (!ctx->colls.empty()) coll_free(ctx, ctx->colls.size() - 1);
    pool_trim(ctx);
    // vectors / pipelines the caller never freed: their handles die with the context, the device memory must not leak
    rz_verify(ctx, nullptr);
    if (ctx->rz_hits) fprintf(stderr, "[bmx redzone] context destroyed with %llu damaged allocation(s) found during its life\n", (unsigned long long)ctx->rz_hits);
    for (auto& kv : ctx->pool_live) { auto rz = ctx->rz_live.find(kv.first); (void)hipFree(rz != ctx->rz_live.end() ? rz->second.raw : kv.first); if (rz != ctx->rz_live.end()) ctx->rz_live.erase(rz); }
    ctx->pool_live.clear();
    ensure_release(ctx, &ctx->scratch, &ctx->scratch_bytes);
    ensure_release(ctx, &ctx->aux, &ctx->aux_bytes);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->d_done) (void)hipFree(ctx->d_done);
    if (ctx->d_slots2) (void)hipFree(ctx->d_slots2);
    if (ctx->d_done2) (void)hipFree(ctx->d_done2);
    if (ctx->d_cursor) (void)hipFree(ctx->d_cursor);
    if (ctx->h_pend) (void)hipHostFree(ctx->h_pend);
    if (ctx->d_zero) (void)hipFree(ctx->d_zero);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->ev_stage) (void)hipEventDestroy(ctx->ev_stage);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return BMX_OK;
ABI_END }

int bmx_ctx_set_tuning(bmx_ctx* ctx, const char* key, int value)
{ ABI_TRY
    ARGCHK(ctx && key);
    std::string k(key);
    if (k == "pipe_unroll") { ARGCHK(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16); ctx->pipe_unroll = value; }
    else if (k == "pipe_rows") { ARGCHK(value == 0 || value == 8 || value == 4 || value == 2 || value == 1); ctx->pipe_rows = value; }
    else if (k == "pipe_nt") ctx->pipe_nt = value != 0;
    else if (k == "pipe_lds") { ARGCHK(value >= 0 && value <= 160 * 1024); ctx->pipe_lds = value; }
    else if (k == "pipe_slots") { ARGCHK(value == 8 || value == 16); ctx->pipe_slots = value; }
    else if (k == "pipe_staged") { ARGCHK(value >= -1 && value <= 1); ctx->pipe_staged = value; }
    else if (k == "pipe_split") { ARGCHK(value >= -1 && value <= 1); ctx->pipe_split = value; }
    else if (k == "pipe_window") { ARGCHK(value >= -1); ctx->pipe_window = value; }
    else if (k == "or_tile") { ARGCHK(value >= 0 && value <= 3); ctx->or_tile = value; }
    else if (k == "or_rows") { ARGCHK(value >= -1 && value <= 1); ctx->or_rows = value; }
    else if (k == "or_depth") { ARGCHK(value == 4 || value == 8); ctx->or_depth = value; }
    else if (k == "direct_cols") { ARGCHK(value >= 0); ctx->direct_cols = value; }
    else if (k == "pair_stream") { ARGCHK(value == -1 || value == 0 || value == 2 || value == 4 || value == 8); ctx->pair_stream = value; }
    else if (k == "pair_wgs") { ARGCHK(value >= 1 && value <= 8); ctx->pair_wgs = value; }
    else if (k == "range_halves") { ARGCHK(value == 0 || value == 1); ctx->range_halves = value; }
    else if (k == "gap_count") { ARGCHK(value >= -1 && value <= 1); ctx->gap_count = value; }
    else if (k == "agg_shape") { ARGCHK(value == 0 || value == 1); ctx->agg_shape = value; }
    else if (k == "and_rows") { ARGCHK(value >= -1 && value <= 1); ctx->and_rows = value; }
    else if (k == "and_rows_wg") { ARGCHK(value == 128 || value == 256 || value == 512); ctx->and_rows_wg = value; }
    else if (k == "and_rows_depth") { ARGCHK(value == 2 || value == 3 || value == 4 || value == 8); ctx->and_rows_depth = value; }
    else if (k == "and_rows_nt") ctx->and_rows_nt = value != 0;
    else if (k == "and_rows_ipw") { ARGCHK(value >= 0 && value <= 64); ctx->and_rows_ipw = value; }
    else if (k == "ff_window") { ARGCHK(value >= -1); ctx->ff_window = value; }
    else if (k == "or_window") { ARGCHK(value >= -9); ctx->or_window = value; }
    else if (k == "gap_pack") { ARGCHK(value >= -1 && value <= 1); ctx->gap_pack = value; }
    else if (k == "coll_members") { ARGCHK(value >= -1 && value <= 1); ctx->coll_members = value; ++ctx->coll_gen; }
    else if (k == "coll_shape") { ARGCHK(value >= 0 && value <= 5); ctx->coll_shape = value; }
    else if (k == "op2_nt") { ARGCHK(value >= 0 && value <= 3); ctx->op2_nt = value; }
    else if (k == "op2_loop") { ARGCHK(value >= -1 && value <= 8); ctx->op2_loop = value; }
    else if (k == "op2_wgs") { ARGCHK(value >= 1 && value <= 8); ctx->op2_wgs = value; }
    else if (k == "pair_nt") { ARGCHK(value >= 0 && value <= 1); ctx->pair_nt = value; }
    else if (k == "pair_loop") { ARGCHK(value >= -1 && value <= 5); ctx->pair_loop = value; }
    else if (k == "eq_big_shape") { ARGCHK(value >= 0 && value <= 2); ctx->eq_big_shape = value; }
    else if (k == "eq_big") { ARGCHK(value >= -1 && value <= 1); ctx->eq_big = value; }
    else if (k == "coll_split") { ARGCHK(value == 0 || value == 1); ctx->coll_split = value; }
    else if (k == "coll_build") { ARGCHK(value >= -1 && value <= 1); ctx->coll_build = value; }
    else if (k == "coll_window") { ARGCHK(value >= 0); ctx->coll_window = value; }
    else if (k == "rs_select_lines") { ARGCHK(value >= 0 && value <= 2); ctx->rs_select_lines = value; }
    else if (k == "rs_sdir_shift") { ARGCHK(value == 0 || (value >= 6 && value <= 20)); ctx->rs_sdir_shift = value; }
    else if (k == "rs_lines") { ARGCHK(value >= 0 && value <= 2); ctx->rs_lines = value; }
    else if (k == "rs_sorted_hint") { ARGCHK(value == 0 || value == 1); ctx->rs_sorted_hint = value; }
    else if (k == "rs_select_sel") { ARGCHK(value >= -1 && value <= 2); ctx->rs_select_sel = value; }
    else if (k == "rs_select_top") { ARGCHK(value >= -1 && value <= 1); ctx->rs_select_top = value; }
    else if (k == "rs_lanes") { ARGCHK(value == 0 || value == 2 || value == 4 || value == 8); ctx->rs_lanes = value; }
    else if (k == "pipe_wg") { ARGCHK(value == 0 || (value >= 64 && value <= 1024 && value % 64 == 0)); ctx->pipe_wg = value; }
    else if (k == "xcd_swizzle") ctx->xcd_swz = value != 0;
    else { g_last_error = "unknown tuning key"; return BMX_ERR_BADARG; }
    return BMX_OK;
ABI_END }

// Measurement helper: ms of one pass of c = a & b over three buffers of `bytes` each, in the launch shape of k_op2_stream
// (a wave per stretch of 8-KiB blocks, non-temporal 16-byte loads and stores): the yardstick bench.py --config 1 puts next
// to the materialised pairwise operations.  The passes rotate over `sets` buffer triples so that nothing is served by the
// Infinity Cache.
int bmx_probe_stream_rw(bmx_ctx* ctx, uint64_t bytes, int sets, int wgs_per_cu, int iters, float* ms_per_pass)
{ ABI_TRY
    ARGCHK(ctx && ms_per_pass && bytes >= 8192 && sets >= 1 && sets <= 8 && wgs_per_cu >= 1 && wgs_per_cu <= 8 && iters >= 1);
    int rc = set_dev(ctx); if (rc) return rc;
    const u64 nblk = bytes / 8192;
    if (nblk > 0x7FFFFFFFull) { g_last_error = "probe buffer too large"; return BMX_ERR_RANGE; }
    void* buf = nullptr;
    HIPCHK(hipMalloc(&buf, (size_t)nblk * 8192 * 3 * sets));
    hipError_t e = hipMemsetAsync(buf, 0x5A, (size_t)nblk * 8192 * 3 * sets, ctx->stream);
    const u32 waves = 4u, total = 256u * waves * (u32)wgs_per_cu;
    const u32 per_wave = (u32)((nblk + total - 1u) / total);
    const u32 grid = (u32)(((nblk + per_wave - 1u) / per_wave + waves - 1u) / waves);
    for (int it = -2; it < iters && e == hipSuccess; ++it) {
        if (it == 0) e = hipEventRecord(ctx->ev0, ctx->stream);
        uint4* base = (uint4*)buf + (size_t)((it + 2) % sets) * nblk * 512u * 3u;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_probe_rw<4>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)base, (const uint4*)(base + nblk * 512u),
                           base + 2u * nblk * 512u, (u32)nblk, per_wave);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail_hip(e, "bmx_probe_stream_rw", __LINE__);
    *ms_per_pass = ms / iters;
    return BMX_OK;
ABI_END }

#ifdef BMX_DIAG
int bmx_diag_stream_read(bmx_ctx* ctx, uint64_t bytes, int nt, uint32_t blocks_per_wave, int pattern, int iters, float* ms_per_pass)
{
    ARGCHK(ctx && ms_per_pass && bytes >= 8192 && blocks_per_wave >= 1 && iters >= 1);
    int rc = set_dev(ctx); if (rc) return rc;
    void* buf = nullptr;
    u64 nblk = bytes / 8192;
    HIPCHK(hipMalloc(&buf, nblk * 8192));
    hipError_t e = hipMemsetAsync(buf, 0x5A, nblk * 8192, ctx->stream);
    u64 waves = (nblk + blocks_per_wave - 1) / blocks_per_wave;
    u32 grid = (u32)((waves + 3) / 4);
    for (int it = -1; it < iters && e == hipSuccess; ++it) {
        if (it == 0) e = hipEventRecord(ctx->ev0, ctx->stream);
#define DIAG_BUF(AUX) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_diag_stream_read_buf<AUX>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)buf, nblk, blocks_per_wave, pattern, ctx->xcd_swz, ctx->d_small)
        if (nt >= 100) {                         // 100 + cache-policy bits of a raw buffer load (1 = sc0, 2 = nt, 16 = sc1)
            switch (nt - 100) { case 0: DIAG_BUF(0); break; case 1: DIAG_BUF(1); break; case 2: DIAG_BUF(2); break; case 3: DIAG_BUF(3); break;
                                case 16: DIAG_BUF(16); break; case 17: DIAG_BUF(17); break; case 18: DIAG_BUF(18); break; default: DIAG_BUF(19); break; }
        }
        else if (nt) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_diag_stream_read<true>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)buf, nblk, blocks_per_wave, pattern, ctx->xcd_swz, ctx->d_small);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_diag_stream_read<false>), dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)buf, nblk, blocks_per_wave, pattern, ctx->xcd_swz, ctx->d_small);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail_hip(e, "bmx_diag_stream_read", __LINE__);
    *ms_per_pass = ms / iters;
    return BMX_OK;
}

#endif  // BMX_DIAG

int bmx_ctx_synchronize(bmx_ctx* ctx)
{ ABI_TRY
    ARGCHK(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->redzone) {                                          // debug: every live allocation's red zones
        int rc = set_dev(ctx); if (rc) return rc;
        const uint64_t before = ctx->rz_hits;
        rz_verify(ctx, nullptr);
        if (ctx->rz_hits != before) { g_last_error = ctx->rz_report; return BMX_ERR_DEVICE; }
    }
    return BMX_OK;
ABI_END }

int bmx_debug_redzone_check(bmx_ctx* ctx, int* enabled, uint64_t* hits, char* report, size_t report_len)
{ ABI_TRY
    ARGCHK(ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    rz_verify(ctx, nullptr);
    if (enabled) *enabled = ctx->redzone ? 1 : 0;
    if (hits) *hits = ctx->rz_hits;
    if (report && report_len) { size_t n = std::min(report_len - 1, ctx->rz_report.size()); memcpy(report, ctx->rz_report.data(), n); report[n] = 0; }
    return BMX_OK;
ABI_END }

int bmx_debug_inject_failure(bmx_ctx* ctx, int kind, long long after)
{ ABI_TRY
    ARGCHK(kind >= 0 && kind <= 5 && after >= 0);
    if (kind == 4) { ARGCHK(ctx); ctx->fail_dmalloc_after = after; return BMX_OK; }
    if (kind == 5) {                                             // red-zone self-test: one byte written just past the requested end of a fresh block
        ARGCHK(ctx && ctx->redzone);
        int rc = set_dev(ctx); if (rc) return rc;
        void* d = nullptr;
        if ((rc = dmalloc(ctx, &d, 1000 + (size_t)after))) return rc;
        HIPCHK(hipMemsetAsync((u8*)d + rz_user_end(1000 + (size_t)after), 0, 1, ctx->stream));
        dfree(ctx, d);
        return BMX_OK;
    }
    if (kind == 0) { g_inject_after = -1; g_inject_kind = 0; if (ctx) ctx->fail_dmalloc_after = -1; return BMX_OK; }
    g_inject_kind = kind; g_inject_after = after;
    return BMX_OK;
ABI_END }

int bmx_ctx_trim(bmx_ctx* ctx)
{ ABI_TRY
    ARGCHK(ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    pool_trim(ctx);
    return BMX_OK;
ABI_END }

int bmx_ctx_mem_used(const bmx_ctx* ctx, uint64_t* bytes)
{ ABI_TRY
    ARGCHK(ctx && bytes);
    *bytes = ctx->mem_used;
    return BMX_OK;
ABI_END }

int bmx_collection_prepare(bmx_ctx* ctx, const bmx_vec* const* vecs, size_t n, int role)
{ ABI_TRY
    ARGCHK(ctx && n > 0 && vecs && (role == BMX_ROLE_OR || role == BMX_ROLE_AND || role == BMX_ROLE_SUB));
    int rc = set_dev(ctx); if (rc) return rc;
    for (size_t i = 0; i < n; ++i) if (!vecs[i] || vecs[i]->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
    if (ctx->gap_pack == 0 || !coll_packable(ctx, vecs, n) || n > 65535) {
        g_last_error = "not packable: needs 1..65535 vectors made of GAP / NULL / FULL blocks only (at least one GAP block), and gap_pack != 0";
        return BMX_ERR_BADARG;
    }
    const int polarity = role == BMX_ROLE_AND ? 0 : 1;
    for (bmx_coll* c : ctx->colls) {                         // the same list in the same role is already there
        if (c->polarity != polarity || c->key.size() != n) continue;
        bool same = true;
        for (size_t i = 0; i < n && same; ++i) same = c->key[i] == vecs[i]->uid;
        if (same) { c->prepared = true; c->last_use = ++ctx->coll_tick; return BMX_OK; }
    }
    uint64_t need = 0;
    for (size_t i = 0; i < n; ++i) need += 2ull * vecs[i]->gap_words;
    if (need > ctx->pack_cap) { g_last_error = "collection larger than the packing budget (BMX_PACK_MAX_MB)"; return BMX_ERR_BADALLOC; }
    bmx_coll* c = nullptr;
    rc = coll_build(ctx, vecs, n, polarity, &c, nullptr);
    if (!rc && !c) { g_last_error = "vectors too long for a packed collection"; return BMX_ERR_RANGE; }
    if (!rc) c->prepared = true;
    return rc;
ABI_END }

int bmx_ctx_pack_stats(const bmx_ctx* ctx, uint32_t* n_collections, uint64_t* bytes, float* last_build_ms)
{ ABI_TRY
    ARGCHK(ctx);
    if (n_collections) *n_collections = (uint32_t)ctx->colls.size();
    if (bytes) *bytes = ctx->pack_bytes;
    if (last_build_ms) *last_build_ms = ctx->last_pack_ms;
    return BMX_OK;
ABI_END }

int bmx_ctx_pack_run_bytes(const bmx_ctx* ctx, uint64_t* bytes)
{ ABI_TRY
    ARGCHK(ctx && bytes);
    uint64_t b = 0;
    for (const bmx_coll* c : ctx->colls) b += c->run_bytes;
    *bytes = b;
    return BMX_OK;
ABI_END }

int bmx_timer_start(bmx_ctx* ctx) { ABI_TRY ARGCHK(ctx); HIPCHK(hipEventRecord(ctx->ev0, ctx->stream)); return BMX_OK; ABI_END }
int bmx_timer_stop_ms(bmx_ctx* ctx, float* ms)
{ ABI_TRY
    ARGCHK(ctx && ms);
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return BMX_OK;
ABI_END }

// Measurement helper (like the timer): how many random 128-byte lines per second this box gathers from a buffer of
// buf_bytes -- the bound of rank / select ("random access => bound by HBM transaction rate").
int bmx_probe_random_lines(bmx_ctx* ctx, uint64_t buf_bytes, uint64_t nlines, int iters, float* ms_per_pass)
{ ABI_TRY
    ARGCHK(ctx && ms_per_pass && buf_bytes >= 128 && nlines >= 1 && iters >= 1);
    int rc = set_dev(ctx); if (rc) return rc;
    void* buf = nullptr;
    u64 nl = buf_bytes / 128;
    HIPCHK(hipMalloc(&buf, nl * 128));
    hipError_t e = hipMemsetAsync(buf, 0x5A, nl * 128, ctx->stream);
    u32 grid = (u32)std::min<u64>((nlines * 8 + 255) / 256, 256u * 16u);
    for (int it = -1; it < iters && e == hipSuccess; ++it) {
        if (it == 0) e = hipEventRecord(ctx->ev0, ctx->stream);
        hipLaunchKernelGGL(k_probe_lines, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)buf, nl, nlines,
                           0xB17A61Cull + (u64)(it + 1) * 7919ull, ctx->d_small);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail_hip(e, "bmx_probe_random_lines", __LINE__);
    *ms_per_pass = ms / iters;
    return BMX_OK;
ABI_END }

// ---------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------
static bmx_vec* vec_alloc_host(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks)
{
    bmx_vec* v = new (std::nothrow) bmx_vec();
    if (!v) return nullptr;
    memset(v, 0, sizeof(*v));
    static std::atomic<uint64_t> next_uid{1};
    v->uid = next_uid.fetch_add(1);
    v->ctx = ctx; v->nbits = nbits; v->nblocks = nblocks;
    ctx->live_vecs[v->uid] = v;
    return v;
}

static int vec_alloc_device(bmx_vec* v, uint32_t n_bit, uint64_t gap_words)
{
    bmx_ctx* ctx = v->ctx;
    int rc;
    v->n_bit = n_bit; v->gap_words = gap_words;
    // + 64-byte guard: gap_apply_lds_lane requests the first 64 B of a GAP block before it knows its length
    size_t b_desc = (size_t)std::max<uint32_t>(v->nblocks, 1) * 8, b_bits = (size_t)n_bit * 8192, b_gaps = (size_t)gap_words * 2 + 64;
    if ((rc = dmalloc(ctx, (void**)&v->d_desc, b_desc))) return rc;
    if ((rc = dmalloc(ctx, (void**)&v->d_bits, b_bits))) return rc;
    if ((rc = dmalloc(ctx, (void**)&v->d_gaps, b_gaps))) return rc;
    // (padding words between GAP blocks read 0xFFFF -- what k_agg_or_rows relies on, bmx_kernels7.h: the kernels that write
    // GAP blocks into the slab, k_emit_blocks / k_emit_gaps / k_gap_repack, write them)
    v->bytes = std::max<size_t>(b_desc, 16) + std::max<size_t>(b_bits, 16) + std::max<size_t>(b_gaps, 16);
    return BMX_OK;
}

// tile directory (bmx_kernels7.h): built on the stream once the vector's descriptors are final; vectors without GAP / FULL
// blocks have none (they contribute nothing to a union and are left out of its operand table)
static int vec_build_tdir(bmx_ctx* ctx, bmx_vec* v)
{
    if (v->d_tdir || !v->nblocks || !(v->counts[BMX_GAP] | v->counts[BMX_FULL])) return BMX_OK;
    const u32 ntiles = (v->nblocks + ORR_TILE - 1u) / ORR_TILE;
    // 16 B per tile + 4 B per block: behind the directory, the blocks' (multi-bit | single-bit << 16) 1-run counts
    const size_t tbytes = (size_t)ntiles * 16 + (size_t)v->nblocks * 4;
    int rc = dmalloc(ctx, &v->d_tdir, tbytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_build_tdir, dim3((ntiles + 255) / 256), dim3(256), 0, ctx->stream, (const u64*)v->d_desc, v->nblocks,
                       (u64)(uintptr_t)v->d_gaps, (u32x4*)v->d_tdir, ntiles, (u32*)((char*)v->d_tdir + (size_t)ntiles * 16));
    KCHK();
    v->bytes += tbytes;
    return BMX_OK;
}

int bmx_vec_free(bmx_ctx* ctx, bmx_vec* v)
{ ABI_TRY
    if (!v) return BMX_OK;
    ARGCHK(ctx && v->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    coll_drop_vector(ctx, v->uid);
    ctx->live_vecs.erase(v->uid);
    dfree(ctx, v->d_desc); dfree(ctx, v->d_bits); dfree(ctx, v->d_gaps); dfree(ctx, v->d_ord); dfree(ctx, v->d_tdir);
    delete v;
    return BMX_OK;
ABI_END }

int bmx_vec_upload(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks,
                   const uint8_t* kinds, const uint32_t* offs,
                   const uint32_t* bit_slab, uint32_t n_bit_blocks,
                   const uint16_t* gap_slab, uint64_t gap_words, bmx_vec** out)
{ ABI_TRY
    ARGCHK(ctx && out && (nblocks == 0 || (kinds && offs)));
    ARGCHK(n_bit_blocks == 0 || bit_slab);
    ARGCHK(gap_words == 0 || gap_slab);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    // Host work is O(nblocks): kinds / offsets / GAP headers.  The two slabs are handed to the copy engine as
    // they are (for a freeze()d vector that is the arena memory itself: no per-block host copy of bit OR GAP
    // blocks); GAP blocks are re-packed onto 16-byte boundaries and validated (terminator, strictly ascending
    // run ends) by k_gap_repack on the device.
    std::vector<u64> desc(std::max<uint32_t>(nblocks, 1), 0);
    std::vector<u32> gsrc;                                     // per block: source word offset of a GAP block (others: 0)
    uint32_t counts[4] = {0, 0, 0, 0};
    uint64_t gpad_words = 0; bool any_gap = false;
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        uint8_t k = kinds[nb];
        if (k > BMX_GAP) { g_last_error = "bad block kind"; return BMX_ERR_BADARG; }
        counts[k]++;
        if (k == BMX_BIT && offs[nb] >= n_bit_blocks) { g_last_error = "bit-block offset out of range"; return BMX_ERR_RANGE; }
        if (k != BMX_GAP) continue;
        uint64_t o = offs[nb];
        if (o >= gap_words) { g_last_error = "GAP offset out of range"; return BMX_ERR_RANGE; }
        uint32_t len = gap_slab[o] >> 3;
        // len <= 1279 = capacity of the top GAP level (glen(3) = 1280 words incl. the header, src/bmconst.h:81-87)
        if (len == 0 || len > 1279u || o + len + 1u > gap_words) { g_last_error = "malformed GAP block (length)"; return BMX_ERR_RANGE; }
        if (!any_gap) { gsrc.assign(nblocks, 0); any_gap = true; }
        gsrc[nb] = (u32)o;
        desc[nb] = gpad_words;                                 // destination word offset, turned into a descriptor below
        gpad_words += ((uint64_t)len + 1u + 7u) & ~7ull;
    }
    bmx_vec* v = vec_alloc_host(ctx, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
#define UPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_vec_free(ctx, v); return r_; } } while (0)
    if ((rc = vec_alloc_device(v, n_bit_blocks, gpad_words))) { bmx_vec_free(ctx, v); return rc; }
    memcpy(v->counts, counts, sizeof(counts));
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        uint8_t k = kinds[nb];
        if (k == BMX_BIT) desc[nb] = DESC_MAKE(v->d_bits + (size_t)offs[nb] * 512u, K_BIT);
        else if (k == BMX_GAP) { u16 h = gap_slab[offs[nb]]; desc[nb] = DESC_MAKE_GAP(v->d_gaps + desc[nb], h >> 3, h & 1u); }
        else desc[nb] = DESC_MAKE(0, k);
    }
    if (nblocks) UPCHK(hipMemcpyAsync(v->d_desc, desc.data(), (size_t)nblocks * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_bit_blocks) UPCHK(hipMemcpyAsync(v->d_bits, bit_slab, (size_t)n_bit_blocks * 8192, hipMemcpyHostToDevice, ctx->stream));
    if (any_gap) {
        size_t raw_bytes = ((size_t)gap_words * 2 + 15u) & ~(size_t)15u, src_bytes = (size_t)nblocks * 4;
        if ((rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, raw_bytes + src_bytes))) { bmx_vec_free(ctx, v); return rc; }
        u16* d_raw = (u16*)ctx->scratch; u32* d_src = (u32*)((char*)ctx->scratch + raw_bytes);
        UPCHK(hipMemcpyAsync(d_raw, gap_slab, (size_t)gap_words * 2, hipMemcpyHostToDevice, ctx->stream));
        UPCHK(hipMemcpyAsync(d_src, gsrc.data(), src_bytes, hipMemcpyHostToDevice, ctx->stream));
        UPCHK(hipMemsetAsync(ctx->d_small, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_gap_repack, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const u16*)d_raw, (const u32*)d_src, (const u64*)v->d_desc, nblocks, ctx->d_small);
        UPCHK(hipGetLastError());
        UPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if ((rc = vec_build_tdir(ctx, v))) { (void)hipStreamSynchronize(ctx->stream); bmx_vec_free(ctx, v); return rc; }
    UPCHK(hipStreamSynchronize(ctx->stream));
#undef UPCHK
    if (any_gap && ctx->h_small[0]) {
        bmx_vec_free(ctx, v);
        g_last_error = "malformed GAP block (terminator / run ends not strictly ascending)";
        return BMX_ERR_RANGE;
    }
    *out = v;
    return BMX_OK;
ABI_END }

// raw block slab (in ctx->scratch, nblocks x 8 KiB) -> classified / compressed vector
static int vec_from_raw(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks, int optimize, bmx_vec** out)
{
    int rc;
    size_t aux_need = (size_t)nblocks * (sizeof(BlockStat) + 4 + 4) + 64;      // st[], offs[], and a block list behind them (result_list())
    if ((rc = ensure(ctx, &ctx->aux, &ctx->aux_bytes, aux_need))) return rc;
    BlockStat* st = (BlockStat*)ctx->aux;
    u32* offs = (u32*)((char*)ctx->aux + (size_t)nblocks * sizeof(BlockStat));
    const uint4* raw = (const uint4*)ctx->scratch;
    bmx_vec* v = vec_alloc_host(ctx, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
#define RAWCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_vec_free(ctx, v); return r_; } } while (0)
    if (nblocks) {
        hipLaunchKernelGGL(k_block_stats, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, raw, nblocks, optimize, st);
        RAWCHK(hipGetLastError());
        hipLaunchKernelGGL(k_scan_layout, dim3(1), dim3(1024), 0, ctx->stream, st, nblocks, offs, ctx->d_small);
        RAWCHK(hipGetLastError());
        RAWCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 6 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        RAWCHK(hipStreamSynchronize(ctx->stream));
    } else memset(ctx->h_small, 0, 6 * sizeof(u64));
    uint32_t n_bit = (uint32_t)ctx->h_small[0]; uint64_t gap_words = ctx->h_small[1];
    for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
    if ((rc = vec_alloc_device(v, n_bit, gap_words))) { bmx_vec_free(ctx, v); return rc; }
    if (nblocks) {
        hipLaunchKernelGGL(k_emit_blocks, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           raw, nblocks, st, offs, v->d_bits, v->d_gaps, v->d_desc);
        RAWCHK(hipGetLastError());
        if ((rc = vec_build_tdir(ctx, v))) { (void)hipStreamSynchronize(ctx->stream); bmx_vec_free(ctx, v); return rc; }
        RAWCHK(hipStreamSynchronize(ctx->stream));
    }
#undef RAWCHK
    *out = v;
    return BMX_OK;
}

int bmx_vec_import_bits(bmx_ctx* ctx, const uint32_t* words, uint64_t nwords, int optimize, bmx_vec** out)
{ ABI_TRY
    ARGCHK(ctx && out && (nwords == 0 || words));
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    uint64_t nblocks64 = (nwords + BMX_BLOCK_WORDS - 1) / BMX_BLOCK_WORDS;
    if (nblocks64 > 65536ull * 16) { g_last_error = "vector too long"; return BMX_ERR_RANGE; }
    uint32_t nblocks = (uint32_t)nblocks64;
    size_t raw_bytes = (size_t)nblocks * 8192;
    if ((rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, std::max<size_t>(raw_bytes, 8192)))) return rc;
    if (nwords) HIPCHK(hipMemcpyAsync(ctx->scratch, words, nwords * 4, hipMemcpyHostToDevice, ctx->stream));
    if (raw_bytes > nwords * 4)
        HIPCHK(hipMemsetAsync((char*)ctx->scratch + nwords * 4, 0, raw_bytes - nwords * 4, ctx->stream));
    return vec_from_raw(ctx, nwords * 32ull, nblocks, optimize, out);
ABI_END }

int bmx_vec_generate_shard(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                           uint32_t density_q16, uint64_t nbits, uint32_t nb_from, uint32_t nb_to,
                           int optimize, bmx_vec** out)
{ ABI_TRY
    ARGCHK(ctx && out && density_q16 <= 65536u);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    uint64_t nblocks64 = (nbits + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks64 > 65536ull * 16) { g_last_error = "vector too long"; return BMX_ERR_RANGE; }
    if (nb_to > nblocks64) nb_to = (uint32_t)nblocks64;
    if (nb_from > nb_to) { g_last_error = "nb_from > nb_to"; return BMX_ERR_RANGE; }
    uint32_t nblocks = nb_to - nb_from;
    uint64_t lo = (uint64_t)nb_from * BMX_BLOCK_BITS, hi = std::min<uint64_t>(nbits, (uint64_t)nb_to * BMX_BLOCK_BITS);
    uint64_t shard_bits = hi > lo ? hi - lo : 0;
    size_t raw_bytes = (size_t)nblocks * 8192;
    if ((rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, std::max<size_t>(raw_bytes, 8192)))) return rc;
    u64 nwords64 = (u64)nblocks * 1024u;
    if (nwords64) {
        u32 grid = (u32)std::min<u64>((nwords64 + 255) / 256, 256u * 32u);
        hipLaunchKernelGGL(k_generate, dim3(grid), dim3(256), 0, ctx->stream, seed, vec_id, with_common,
                           density_q16, nbits, (u64*)ctx->scratch, nwords64, (u64)nb_from * 1024u);
        KCHK();
    }
    return vec_from_raw(ctx, shard_bits, nblocks, optimize, out);
ABI_END }

int bmx_vec_generate(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                     uint32_t density_q16, uint64_t nbits, int optimize, bmx_vec** out)
{ ABI_TRY
    return bmx_vec_generate_shard(ctx, seed, vec_id, with_common, density_q16, nbits, 0u, 0xFFFFFFFFu, optimize, out);
ABI_END }

} // extern "C"

// ordinals of the bit-blocks of a result whose slab has unused slots (what the layout scan would have left in d_ord): computed
// when a download first needs them, so that the operation that produced the vector does not pay a scan for it
static int vec_build_ord(bmx_ctx* ctx, bmx_vec* v)
{
    if (v->d_ord || !v->nblocks) { v->ord_lazy = false; return BMX_OK; }
    int rc;
    if ((rc = dmalloc(ctx, (void**)&v->d_ord, (size_t)v->nblocks * 4))) return rc;
    hipLaunchKernelGGL(k_ord_from_desc, dim3(1), dim3(1024), 0, ctx->stream, (const u64*)v->d_desc, v->nblocks, v->d_ord);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dfree(ctx, v->d_ord); v->d_ord = nullptr; return fail_hip(e, "k_ord_from_desc", __LINE__); }
    v->ord_lazy = false;
    return BMX_OK;
}

extern "C" {

int bmx_vec_info(const bmx_vec* v, uint64_t* nbits, uint32_t* nblocks, uint32_t counts[4],
                 uint32_t* bit_slab_blocks, uint64_t* gap_words)
{ ABI_TRY
    ARGCHK(v);
    if (nbits) *nbits = v->nbits;
    if (nblocks) *nblocks = v->nblocks;
    if (counts) memcpy(counts, v->counts, sizeof(v->counts));
    if (bit_slab_blocks) *bit_slab_blocks = (v->d_ord || v->ord_lazy) ? v->counts[BMX_BIT] : v->n_bit;   // a slab with unused slots is gathered on download
    if (gap_words) *gap_words = v->gap_words;
    return BMX_OK;
ABI_END }

int bmx_vec_operand_bytes(bmx_ctx* ctx, const bmx_vec* v, uint64_t* bytes)
{ ABI_TRY
    ARGCHK(ctx && v && v->ctx == ctx && bytes);
    int rc = set_dev(ctx); if (rc) return rc;
    *bytes = 0;
    if (!v->nblocks) return BMX_OK;
    HIPCHK(hipMemsetAsync(ctx->d_small, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_vec_alg_bytes, dim3((v->nblocks + 255) / 256), dim3(256), 0, ctx->stream, (const u64*)v->d_desc, v->nblocks, ctx->d_small);
    KCHK();
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *bytes = ctx->h_small[0];
    return BMX_OK;
ABI_END }

// sorted positions of the set bits: device compaction (bmx_kernels6.h k_block_counts / k_rs_scan / k_expand_indices)
static int vec_indices_impl(bmx_ctx* ctx, const bmx_vec* v, int width, void* out, bool out_is_host, uint64_t cap, uint64_t* n)
{
    ARGCHK(ctx && v && v->ctx == ctx && n && (width == 4 || width == 8) && (cap == 0 || out));
    int rc = set_dev(ctx); if (rc) return rc;
    *n = 0;
    const uint32_t nblocks = v->nblocks;
    if (!nblocks) return BMX_OK;
    if (width == 4 && (uint64_t)nblocks > 65536ull) { g_last_error = "32-bit positions cannot address this vector: use width 8"; return BMX_ERR_RANGE; }
    u32* d_bc = nullptr; u64* d_rc = nullptr; void* d_out = nullptr;
    if ((rc = dmalloc(ctx, (void**)&d_bc, (size_t)nblocks * 4)) || (rc = dmalloc(ctx, (void**)&d_rc, (size_t)nblocks * 8))) { dfree(ctx, d_bc); return rc; }
    hipLaunchKernelGGL(k_block_counts, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, (const u64*)v->d_desc, nblocks, d_bc);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, ctx->stream, (const u32*)d_bc, nblocks, d_rc, ctx->d_small);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    uint64_t total = e == hipSuccess ? ctx->h_small[0] : 0;
    if (e == hipSuccess) {
        *n = total;
        if (total > cap) { rc = BMX_ERR_RANGE; g_last_error = "output buffer too small for the positions (n holds the number needed)"; }
        else if (total) {
            d_out = out_is_host ? nullptr : out;
            if (out_is_host) rc = dmalloc(ctx, &d_out, (size_t)total * (size_t)width);
            if (!rc) {
                if (width == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expand_indices<u64>), dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                                                   (const u64*)v->d_desc, nblocks, (const u64*)d_rc, (u64*)d_out, total, 0ull);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_expand_indices<u32>), dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                                        (const u64*)v->d_desc, nblocks, (const u64*)d_rc, (u32*)d_out, total, 0ull);
                e = hipGetLastError();
                if (e == hipSuccess && out_is_host) e = hipMemcpyAsync(out, d_out, (size_t)total * (size_t)width, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            }
        }
    }
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); rc = fail_hip(e, "bmx_vec_to_indices", __LINE__); }
    if (out_is_host) dfree(ctx, d_out);
    dfree(ctx, d_bc); dfree(ctx, d_rc);
    return rc;
}

int bmx_vec_to_indices(bmx_ctx* ctx, const bmx_vec* v, int width, void* out, uint64_t cap, uint64_t* n)
{ ABI_TRY
    return vec_indices_impl(ctx, v, width, out, true, cap, n);
ABI_END }

int bmx_vec_to_indices_dev(bmx_ctx* ctx, const bmx_vec* v, int width, void* d_out, uint64_t cap, uint64_t* n)
{ ABI_TRY
    return vec_indices_impl(ctx, v, width, d_out, false, cap, n);
ABI_END }

int bmx_vec_download(bmx_ctx* ctx, const bmx_vec* v, uint8_t* kinds, uint32_t* offs,
                     uint32_t* bit_slab, uint16_t* gap_slab)
{ ABI_TRY
    ARGCHK(ctx && v && v->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    if (v->ord_lazy && !v->d_ord && (rc = vec_build_ord(ctx, const_cast<bmx_vec*>(v)))) return rc;   // (a cache of the immutable vector's layout: logically const)
    if (kinds || offs) {
        std::vector<u64> desc(std::max<uint32_t>(v->nblocks, 1));
        std::vector<u32> ord;
        HIPCHK(hipMemcpyAsync(desc.data(), v->d_desc, (size_t)v->nblocks * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (offs && v->d_ord && v->nblocks) {
            ord.resize(v->nblocks);
            HIPCHK(hipMemcpyAsync(ord.data(), v->d_ord, (size_t)v->nblocks * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
            u32 k = DESC_K(desc[nb]);
            if (kinds) kinds[nb] = (uint8_t)k;
            if (offs) {
                if (k == K_BIT) offs[nb] = v->d_ord ? ord[nb] : (uint32_t)((DESC_P(desc[nb]) - (u64)(uintptr_t)v->d_bits) / 8192u);
                else if (k == K_GAP) offs[nb] = (uint32_t)((DESC_P(desc[nb]) - (u64)(uintptr_t)v->d_gaps) / 2u);
                else offs[nb] = 0;
            }
        }
    }
    if (bit_slab && v->d_ord && v->counts[BMX_BIT]) {
        // result slab with unused slots: only the live blocks cross PCIe, gathered into their ordinals first
        size_t bytes = (size_t)v->counts[BMX_BIT] * 8192;
        if ((rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes))) return rc;
        hipLaunchKernelGGL(k_gather_bits, dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const u64*)v->d_desc, (const u32*)v->d_ord, v->nblocks, (uint4*)ctx->scratch);
        KCHK();
        HIPCHK(hipMemcpyAsync(bit_slab, ctx->scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    } else if (bit_slab && v->n_bit && v->d_bits)
        HIPCHK(hipMemcpyAsync(bit_slab, v->d_bits, (size_t)v->n_bit * 8192, hipMemcpyDeviceToHost, ctx->stream));
    if (gap_slab && v->gap_words) HIPCHK(hipMemcpyAsync(gap_slab, v->d_gaps, (size_t)v->gap_words * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
ABI_END }

int bmx_vec_to_words(bmx_ctx* ctx, const bmx_vec* v, uint32_t* words, uint64_t nwords)
{ ABI_TRY
    ARGCHK(ctx && v && v->ctx == ctx && (nwords == 0 || words));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!nwords) return BMX_OK;
    uint32_t nb_out = (uint32_t)((nwords + BMX_BLOCK_WORDS - 1) / BMX_BLOCK_WORDS);
    if ((rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, (size_t)nb_out * 8192))) return rc;
    hipLaunchKernelGGL(k_vec_expand, dim3((nb_out + 3) / 4), dim3(256), 0, ctx->stream,
                       v->d_desc, v->nblocks, nb_out, (uint4*)ctx->scratch);
    KCHK();
    HIPCHK(hipMemcpyAsync(words, ctx->scratch, nwords * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
ABI_END }

} // extern "C"

int bmx_i_count_async(bmx_ctx* ctx, const bmx_vec* a, int slot)
{
    ARGCHK(ctx && a && a->ctx == ctx && slot >= 0 && slot < 64);
    int rc = set_dev(ctx); if (rc) return rc;
    if (!a->nblocks) { ctx->h_small[slot] = 0; return BMX_OK; }
    // the last workgroup folds the partial counts and writes the total straight into the pinned word the host reads
    if (ctx->pair_stream != 0 && a->counts[BMX_BIT] == a->nblocks && a->nblocks >= 2048u) {     // bit-blocks only: the streaming form
        const u32 total = 256u * 4u * (u32)std::max(ctx->pair_wgs, 1);
        u32 per_wave = (a->nblocks + total - 1u) / total;
        u32 grid = ((a->nblocks + per_wave - 1u) / per_wave + 3u) / 4u;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_op2_stream<4, true, 1>), dim3(grid), dim3(256), 0, ctx->stream, 0, a->d_desc, a->d_desc,
                           a->nblocks, per_wave, FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + slot});
    } else
    hipLaunchKernelGGL(k_vec_count, dim3((a->nblocks + 3) / 4), dim3(256), 0, ctx->stream, a->d_desc, a->nblocks,
                       FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + slot});
    KCHK();
    return BMX_OK;
}

extern "C" {

int bmx_count(bmx_ctx* ctx, const bmx_vec* a, uint64_t* count)
{ ABI_TRY
    ARGCHK(ctx && a && count && a->ctx == ctx);
    if (a->count_valid) { *count = a->count; return BMX_OK; }      // folded by the kernel that produced the vector
    int rc = bmx_i_count_async(ctx, a, 0); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count = ctx->h_small[0];
    return BMX_OK;
ABI_END }

// ---------------------------------------------------------------------------
// pipeline
// ---------------------------------------------------------------------------
int bmx_pipeline_create(bmx_ctx* ctx, const bmx_vec* const* and_list, const uint32_t* and_n,
                        const bmx_vec* const* sub_list, const uint32_t* sub_n,
                        size_t ngroups, bmx_pipeline** out)
{ ABI_TRY
    ARGCHK(ctx && out && ngroups > 0 && ngroups < (1u << 20) && and_n && sub_n);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    size_t tot_and = 0, tot_sub = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        if (and_n[g] > 65535u || sub_n[g] > 65535u) { g_last_error = "more than 65535 operands in one arg-group"; return BMX_ERR_RANGE; }
        tot_and += and_n[g]; tot_sub += sub_n[g];
    }
    ARGCHK(tot_and == 0 || and_list);
    ARGCHK(tot_sub == 0 || sub_list);
    size_t n_ops = tot_and + tot_sub;
    std::vector<const u64*> descs(std::max<size_t>(n_ops, 1), nullptr);
    std::vector<u32> meta(5 * ngroups + std::max<size_t>(n_ops, 1), 0);
    u32* row_off = meta.data(); u32* m_and_n = row_off + ngroups; u32* m_sub_n = m_and_n + ngroups;
    u32* and_off = m_sub_n + ngroups; u32* sub_off = and_off + ngroups; u32* nblk = sub_off + ngroups;
    // row offsets / operand offsets are 32-bit on the device: refuse what does not fit instead of wrapping
    if (n_ops > 0xFFFFFFF0ull || 2ull * ngroups + n_ops > 0xFFFFFFF0ull) { g_last_error = "pipeline too large: operand count exceeds 32 bits"; return BMX_ERR_RANGE; }
    uint32_t ncols = 0, col_stride = 0; bool has_gap = false, has_bit = false; uint64_t max_bits = 0;
    uint64_t gap_words_sum = 0, gap_blocks_sum = 0;
    size_t ia = 0, is = 0;
    uint32_t null_row_off = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        row_off[g] = col_stride; col_stride += 2 + and_n[g] + sub_n[g];
        m_and_n[g] = and_n[g]; m_sub_n[g] = sub_n[g];
        and_off[g] = (u32)ia; sub_off[g] = (u32)(tot_and + is);
        for (uint32_t k = 0; k < and_n[g]; ++k, ++ia) {
            const bmx_vec* v = and_list[ia];
            if (!v || v->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
            descs[ia] = v->d_desc; nblk[ia] = v->nblocks; ncols = std::max(ncols, v->nblocks); has_gap |= v->counts[BMX_GAP] != 0; has_bit |= v->counts[BMX_BIT] != 0; gap_words_sum += v->gap_words; gap_blocks_sum += v->counts[BMX_GAP];
            max_bits = std::max(max_bits, v->nbits);
        }
        for (uint32_t k = 0; k < sub_n[g]; ++k, ++is) {
            const bmx_vec* v = sub_list[is];
            if (!v || v->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
            descs[tot_and + is] = v->d_desc; nblk[tot_and + is] = v->nblocks; ncols = std::max(ncols, v->nblocks); has_gap |= v->counts[BMX_GAP] != 0; has_bit |= v->counts[BMX_BIT] != 0; gap_words_sum += v->gap_words; gap_blocks_sum += v->counts[BMX_GAP];
            max_bits = std::max(max_bits, v->nbits);
        }
    }
    null_row_off = col_stride; col_stride += 2;               // the always-empty row behind the groups' rows (k_limit_null)
#define PIPECHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_pipeline_destroy(ctx, p); return r_; } } while (0)
    bmx_pipeline* p = new (std::nothrow) bmx_pipeline();
    if (!p) return BMX_ERR_BADALLOC;
    memset(p, 0, sizeof(*p));
    p->search_limit = ~0ull; p->cm_gen = ~0ull; p->cm_tried_gen = ~0ull - 1;     // (the memset above wiped the member initialisers)
    p->ctx = ctx; p->ngroups = (uint32_t)ngroups; p->ncols = ncols; p->col_stride = col_stride; p->null_row_off = null_row_off; p->n_ops = (uint32_t)n_ops; p->has_gap = has_gap; p->has_bit = has_bit; p->gap_avg_words = gap_blocks_sum ? (uint32_t)(gap_words_sum / gap_blocks_sum) : 0u;
    p->nbits = max_bits;
    p->h_row_off = new std::vector<u32>(row_off, row_off + ngroups);
    p->h_and_n = new std::vector<u32>(m_and_n, m_and_n + ngroups);
    p->h_sub_n = new std::vector<u32>(m_sub_n, m_sub_n + ngroups);
    if (has_gap && !has_bit && tot_and) {
        // GAP-only operands: remember WHICH vectors (uids, not pointers) so that a run can be served by the packed collections
        // that hold them (the reference's pipeline keeps bvector pointers, :2939; here a freed vector is just not found)
        p->h_uids = new std::vector<uint64_t>();
        p->h_uids->reserve(tot_and + tot_sub);
        for (size_t i = 0; i < tot_and; ++i) p->h_uids->push_back(and_list[i]->uid);
        for (size_t i = 0; i < tot_sub; ++i) p->h_uids->push_back(sub_list[i]->uid);
    }
    // distinct vectors of the pipeline (pipeline::unique_vectors(), src/bmaggregator.h:301) and the
    // (AND | SUB << 16) plane masks of every group, 16 planes per chunk -- only where the staged kernel can be chosen
    // (>= 32 groups, or forced by the knob): a single-group combine_and_sub pays neither the hashing nor the uploads
    if (ngroups >= 32 || ctx->pipe_staged == 1)
    {
        std::unordered_map<const u64*, u32> plane_of;
        std::vector<const u64*> udesc; std::vector<u32> unblk;
        auto plane = [&](size_t op) {
            auto it = plane_of.find(descs[op]);
            if (it != plane_of.end()) return it->second;
            u32 id = (u32)udesc.size(); plane_of.emplace(descs[op], id);
            udesc.push_back(descs[op]); unblk.push_back(nblk[op]);
            return id;
        };
        std::vector<u32> plane_id(std::max<size_t>(n_ops, 1));
        for (size_t op = 0; op < n_ops; ++op) plane_id[op] = plane(op);
        p->nplanes = (uint32_t)udesc.size();
        p->nchunks = (p->nplanes + 15u) / 16u;
        p->staged_ok = p->nplanes > 0 && (size_t)ngroups * p->nchunks < (1u << 28);
        if (p->staged_ok) {
            std::vector<u32> gmask((size_t)ngroups * std::max<u32>(p->nchunks, 1), 0), gskip(ngroups, 0);
            size_t a0 = 0, s0 = 0;
            for (size_t g = 0; g < ngroups; ++g) {
                gskip[g] = and_n[g] == 0;
                for (uint32_t k = 0; k < and_n[g]; ++k) { u32 pl = plane_id[a0 + k]; gmask[g * p->nchunks + pl / 16] |= 1u << (pl % 16); }
                for (uint32_t k = 0; k < sub_n[g]; ++k) { u32 pl = plane_id[tot_and + s0 + k]; gmask[g * p->nchunks + pl / 16] |= 1u << (16 + pl % 16); }
                a0 += and_n[g]; s0 += sub_n[g];
            }
            if ((rc = dmalloc(ctx, (void**)&p->d_udesc, udesc.size() * 8)) || (rc = dmalloc(ctx, (void**)&p->d_unblk, unblk.size() * 4)) ||
                (rc = dmalloc(ctx, (void**)&p->d_gmask, gmask.size() * 4)) || (rc = dmalloc(ctx, (void**)&p->d_gskip, gskip.size() * 4))) { bmx_pipeline_destroy(ctx, p); return rc; }
            PIPECHK(hipMemcpyAsync(p->d_udesc, udesc.data(), udesc.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            PIPECHK(hipMemcpyAsync(p->d_unblk, unblk.data(), unblk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            PIPECHK(hipMemcpyAsync(p->d_gmask, gmask.data(), gmask.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            PIPECHK(hipMemcpyAsync(p->d_gskip, gskip.data(), gskip.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            PIPECHK(hipStreamSynchronize(ctx->stream));     // the host vectors die at the end of this scope
        }
    }
    size_t b_dmat = (size_t)std::max<uint32_t>(ncols, 1) * col_stride * 8, b_meta = meta.size() * 4, b_descs = descs.size() * 8;
    if ((rc = dmalloc(ctx, (void**)&p->d_dmat, b_dmat)) || (rc = dmalloc(ctx, (void**)&p->d_meta, b_meta)) ||
        (rc = dmalloc(ctx, (void**)&p->d_descs, b_descs))) { bmx_pipeline_destroy(ctx, p); return rc; }
    p->bytes = b_dmat + b_meta + b_descs;
    if ((rc = h2d_staged(ctx, p->d_meta, meta.data(), b_meta)) || (rc = h2d_staged(ctx, (void*)p->d_descs, descs.data(), b_descs))) { bmx_pipeline_destroy(ctx, p); return rc; }
    if (ncols) {
        PipeOperands po;
        po.desc = (const u64* const*)p->d_descs;
        po.nblocks = p->d_meta + 5 * ngroups;
        po.row_off = p->d_meta; po.and_n = p->d_meta + ngroups; po.sub_n = p->d_meta + 2 * ngroups;
        po.and_off = p->d_meta + 3 * ngroups; po.sub_off = p->d_meta + 4 * ngroups;
        u64 nrows = (u64)ncols * ngroups;                  // one wave per (column, group) row
        hipLaunchKernelGGL(k_pipe_sort, dim3((u32)((nrows + 3) / 4)), dim3(256), 0, ctx->stream,
                           po, (u32)ngroups, ncols, col_stride, p->d_dmat);
        PIPECHK(hipGetLastError());
        hipLaunchKernelGGL(k_pipe_null_rows, dim3((ncols + 255u) / 256u), dim3(256), 0, ctx->stream, p->d_dmat, ncols, col_stride, null_row_off);
        PIPECHK(hipGetLastError());
    }
    // no synchronise: the tables went through the pinned ring and everything that uses the rows is ordered behind
    // k_pipe_sort on the context's stream
    *out = p;
    return BMX_OK;
#undef PIPECHK
ABI_END }

int bmx_pipeline_destroy(bmx_ctx* ctx, bmx_pipeline* p)
{ ABI_TRY
    if (!p) return BMX_OK;
    ARGCHK(ctx && p->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(ctx, p->d_dmat); dfree(ctx, p->d_meta); dfree(ctx, (void*)p->d_descs);
    dfree(ctx, (void*)p->d_udesc); dfree(ctx, p->d_unblk); dfree(ctx, p->d_gmask); dfree(ctx, p->d_gskip);
    dfree(ctx, p->cm_buf);
    delete p->h_row_off; delete p->h_and_n; delete p->h_sub_n; delete p->h_uids; delete p->h_win_groups; delete p->h_stop;
    delete p;
    return BMX_OK;
ABI_END }

static int pipe_range(const bmx_pipeline* p, uint32_t& nb_from, uint32_t& nb_to)
{
    if (nb_to > p->ncols) nb_to = p->ncols;
    if (nb_from > nb_to) { g_last_error = "nb_from > nb_to"; return BMX_ERR_RANGE; }
    return BMX_OK;
}

} // extern "C"
// GAP-only pipelines: the union-of-0-runs kernel over the operands' own slabs (k_agg_and_rows, bmx_kernels9.h).  A workgroup
// per (column, group) whose waves share the group's operand list.  Measured against the wave-per-item kernel k_pipe_counts
// (16 groups of n operands over 1e9-bit vectors at 0.1 % / 0.3 %, tools/bench_and_rows.py CROSS=1, profiles/r05_and_rows):
// n = 2: 1.49 / 1.53 against 1.64 / 1.62 ms, 4: 1.55 / 1.66 against 1.53 / 1.72, 8: 1.62 / 1.78 against 1.74 / 2.31,
// 16: 1.74 / 1.91 against 2.40 / 3.50, 32 (8 groups): 1.57 / 1.87 against 7.3 / 8.1 (the counting kernel), 256 (1 group):
// 0.51 / 1.35 against 1.56 / 2.10 -- a tie up to 4 operands, a win from 8 on: and_rows -1 takes it from 8 operands per group.
static bool use_and_rows(const bmx_ctx* ctx, const bmx_pipeline* p, uint64_t ops_of_group = 0)
{
    if (ctx->and_rows == 0 || ctx->gap_count == 1 || !p->has_gap || p->has_bit) return false;
    if (ctx->and_rows > 0) return true;
    if (ops_of_group) return ops_of_group >= 8u;
    return (uint64_t)p->n_ops >= 8ull * p->ngroups;
}
typedef void (*and_rows_fn)(const u64*, const u32*, const u32*, const u32*, u32, u32, u32, u32, int, u64*, uint4*, u64*, BlockStat*, u32, u32, int, u32);
// (tuning build only: BMX_DIAG_AROWS = 1 -> the row loads alone, 2 -> no fold / count; results are then meaningless)
static int and_rows_diag_bits()
{
#ifdef BMX_DIAG
    if (const char* e = getenv("BMX_DIAG_AROWS")) return atoi(e) & 15;
#endif
    return 0;
}
template <int MODE, bool NT>
static and_rows_fn and_rows_kernel_nt(const bmx_ctx* ctx)
{
    const int d = ctx->and_rows_depth;
    switch (ctx->and_rows_wg) {
    case 128:  return d == 2 ? k_agg_and_rows<MODE, 128, 2, NT> : d == 8 ? k_agg_and_rows<MODE, 128, 8, NT> : d == 3 ? k_agg_and_rows<MODE, 128, 3, NT> : k_agg_and_rows<MODE, 128, 4, NT>;
    case 256:  return d == 2 ? k_agg_and_rows<MODE, 256, 2, NT> : d == 8 ? k_agg_and_rows<MODE, 256, 8, NT> : d == 3 ? k_agg_and_rows<MODE, 256, 3, NT> : k_agg_and_rows<MODE, 256, 4, NT>;
    default:   return d == 2 ? k_agg_and_rows<MODE, 512, 2, NT> : d == 8 ? k_agg_and_rows<MODE, 512, 8, NT> : d == 3 ? k_agg_and_rows<MODE, 512, 3, NT> : k_agg_and_rows<MODE, 512, 4, NT>;
    }
}
template <int MODE>
static and_rows_fn and_rows_kernel(const bmx_ctx* ctx) { return ctx->and_rows_nt ? and_rows_kernel_nt<MODE, true>(ctx) : and_rows_kernel_nt<MODE, false>(ctx); }

// GAP-only pipelines with long operand lists: count the covering operands per position (k_pipe_counts_gapcount) instead
// of applying them one by one.  gap_count: -1 = automatic (>= 32 operands per group on average AND GAP blocks of >= 240 words on average), 0 = off, 1 = whenever it applies
static bool use_gapcount(const bmx_ctx* ctx, const bmx_pipeline* p, uint64_t ops_of_group = 0)
{
    if (ctx->gap_count == 0 || !p->has_gap || p->has_bit) return false;
    // k_pipe_counts_gapcount needs 136 KiB of dynamic LDS (+ its static arrays): a device / runtime that does not grant
    // that per workgroup keeps the run-by-run kernel instead of failing the run
    if (ctx->max_lds_bytes < (16384u * 2u + 2048u) * 4u + 1024u) return false;
    if (ctx->gap_count > 0) return true;
    if (ops_of_group) return ops_of_group >= 32u && p->gap_avg_words >= 240u;          // one group (materialising form)
    // measured on the 256-way AND over 1e9-bit vectors: blocks of ~780 words (0.3 %) 2.71 -> 2.09 ms, ~390 words (0.15 %)
    // 1.65 -> 1.57 ms, ~265 words (0.1 %) 1.58 -> 1.54 ms -- below that the per-column scan and zeroing stop paying off
    return (uint64_t)p->n_ops >= 32ull * p->ngroups && p->gap_avg_words >= 240u;
}
// The arg-groups of a GAP-only pipeline against the collections in force: every AND list inside ONE polarity-0 collection
// and every SUB list inside ONE polarity-1 collection => the groups become member-index lists (k_coll_members), or -- one
// group naming whole collections -- the streaming kernel.  Looked at again whenever a collection appeared or went.
// may_build: the synchronous entries may build the collections of a one-group pipeline under gap_pack 1; the asynchronous
// (_dev) entry never builds (it must not block the host or put a transposition into the caller's timed region).
static bmx_coll* coll_by_id(bmx_ctx* ctx, uint64_t id)
{
    if (!id) return nullptr;
    for (bmx_coll* c : ctx->colls) if (c->id == id) return c;
    return nullptr;
}
static int pipe_resolve_colls(bmx_ctx* ctx, bmx_pipeline* p, bool may_build, bmx_coll** a, bmx_coll** s)
{
    *a = nullptr; *s = nullptr;
    if (!p->h_uids || ctx->gap_pack == 0) return BMX_OK;
    size_t tot_and = 0, tot_sub = 0;
    for (u32 g = 0; g < p->ngroups; ++g) { tot_and += (*p->h_and_n)[g]; tot_sub += (*p->h_sub_n)[g]; }
    if (p->cm_gen != ctx->coll_gen || (may_build && ctx->gap_pack == 1 && !p->cm_a_id && p->cm_tried_gen != ctx->coll_gen)) {
        int rc;
        // (no synchronise: a pooled block is only ever handed to work enqueued behind its last reader on this stream, see dfree --
        // the asynchronous entry must not block the host here; a build that was refused is not attempted again until a collection
        // appears or goes)
        if (p->cm_buf) { dfree(ctx, p->cm_buf); p->cm_buf = nullptr; }
        if (may_build && ctx->gap_pack == 1) p->cm_tried_gen = ~0ull;
        p->cm_a_id = p->cm_s_id = 0; p->cm_full = false;
        std::vector<u32> ma, ms; bool fa = false, fs = true;
        const uint64_t* uids = p->h_uids->data();
        bmx_coll* ca = coll_cover(ctx, uids, tot_and, 0, &ma, &fa);
        bmx_coll* cs = (ca && tot_sub) ? coll_cover(ctx, uids + tot_and, tot_sub, 1, &ms, &fs) : nullptr;
        if (!ca && may_build && ctx->gap_pack == 1 && p->ngroups == 1 && tot_and >= 64) {
            // one arg-group of >= 64 packable vectors, first synchronous run: build its collections now (gap_pack 1)
            std::vector<const bmx_vec*> va, vs;
            bool ok = true;
            for (size_t i = 0; i < tot_and + tot_sub && ok; ++i) {
                auto it = ctx->live_vecs.find(uids[i]);
                if (it == ctx->live_vecs.end()) ok = false; else (i < tot_and ? va : vs).push_back(it->second);
            }
            if (ok) {
                bool full = false;
                if ((rc = coll_resolve_and_sub(ctx, va.data(), va.size(), vs.data(), vs.size(), &ca, &cs, &ma, &ms, &full))) return rc;
                fa = fs = full;
            }
        }
        p->cm_gen = ctx->coll_gen;
        if (may_build && ctx->gap_pack == 1) p->cm_tried_gen = ctx->coll_gen;
        const bool whole = p->ngroups == 1 && fa && (!tot_sub || fs);
        if (ca && !whole && !coll_members_wanted(ctx, (uint64_t)p->gap_avg_words, 1, p->n_ops, p->ngroups)) ca = nullptr;      // (the table kernels serve these groups better)
        if (ca && (!tot_sub || cs)) {
            p->cm_a_id = ca->id; p->cm_s_id = cs ? cs->id : 0;
            p->cm_full = p->ngroups == 1 && fa && (!tot_sub || fs);
            // member indices (AND lists of all groups, then SUB lists) + one CollGroup per arg-group
            const size_t goff = ((tot_and + tot_sub) * 4 + 15) & ~(size_t)15;
            std::vector<u8> h(goff + (size_t)p->ngroups * sizeof(CollGroup));
            if (tot_and) memcpy(h.data(), ma.data(), tot_and * 4);
            if (tot_sub) memcpy(h.data() + tot_and * 4, ms.data(), tot_sub * 4);
            CollGroup* gr = reinterpret_cast<CollGroup*>(h.data() + goff);
            u32 ao = 0, so = (u32)tot_and;
            for (u32 g = 0; g < p->ngroups; ++g) {
                gr[g] = CollGroup{ao, (*p->h_and_n)[g], so, (*p->h_sub_n)[g]};
                ao += (*p->h_and_n)[g]; so += (*p->h_sub_n)[g];
            }
            if ((rc = dmalloc(ctx, &p->cm_buf, h.size())) || (rc = h2d_staged(ctx, p->cm_buf, h.data(), h.size()))) {
                dfree(ctx, p->cm_buf); p->cm_buf = nullptr; p->cm_a_id = p->cm_s_id = 0; return rc;
            }
            p->cm_groups_off = goff;
        }
    }
    *a = coll_by_id(ctx, p->cm_a_id);
    *s = coll_by_id(ctx, p->cm_s_id);
    if (!*a || (p->cm_s_id && !*s)) { *a = nullptr; *s = nullptr; return BMX_OK; }
    (*a)->last_use = ++ctx->coll_tick; if (*s) (*s)->last_use = ctx->coll_tick;
    return BMX_OK;
}

// a subset of a pipeline's arg-groups for one counts launch: the compacted per-group tables k_limit_step wrote
struct GroupView { const u32* row_off; const u32* and_n; const u32* sub_n; u32 ngroups; const u32* gmask; const u32* gskip; const CollGroup* cgroups; };
static int pipeline_run_counts_impl(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts, bool may_build, const GroupView* gv = nullptr);
static int limit_counts_run_dev(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts);

extern "C" {

int bmx_pipeline_run_counts_dev(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx && d_counts);
    if (p->search_limit != ~0ull) return limit_counts_run_dev(ctx, p, nb_from, nb_to, d_counts);
    return pipeline_run_counts_impl(ctx, p, nb_from, nb_to, d_counts, false);
ABI_END }

static int pipeline_run_counts_impl(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts, bool may_build, const GroupView* gv)
{
    ARGCHK(ctx && p && p->ctx == ctx && d_counts);
    int rc = set_dev(ctx); if (rc) return rc;
    if ((rc = pipe_range(p, nb_from, nb_to))) return rc;
    // gv: the run covers gv->ngroups arg-groups of the pipeline (those still below their search limit); counts[k] belongs to the
    // k-th of them.  The kernel is chosen as for the whole pipeline.
    const u32 ngroups = gv ? gv->ngroups : p->ngroups;
    HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)std::max(ngroups, 1u) * 8, ctx->stream));
    u64 nitems64 = (u64)(nb_to - nb_from) * ngroups;
    if (!nitems64) return BMX_OK;
    const u32* row_off = gv ? gv->row_off : p->d_meta; const u32* and_n = gv ? gv->and_n : p->d_meta + p->ngroups; const u32* sub_n = gv ? gv->sub_n : p->d_meta + 2 * p->ngroups;
    if (p->h_uids) {
        // GAP-only operands held by packed collections: the column regions of the collections instead of the operands' slabs
        bmx_coll *ca = nullptr, *cs = nullptr;
        if ((rc = pipe_resolve_colls(ctx, p, may_build, &ca, &cs))) return rc;
        if (ca && p->cm_full) return coll_launch(COLL_AND_COUNT, ctx, ca, cs, nb_from, nb_to, 1, (u64*)d_counts, nullptr, nullptr, 0u, 0xFFFFFFFFu);
        if (ca) return coll_members_launch(CM_AND_COUNT, ctx, ca, cs, (const u32*)p->cm_buf, gv ? gv->cgroups : (const CollGroup*)((const char*)p->cm_buf + p->cm_groups_off),
                                           ngroups, nb_from, nb_to, 1, (u64*)d_counts, nullptr, nullptr);
    }
    {
        // many groups over few distinct vectors: every plane block is re-used >= 8 times per column
        bool reuse = p->ngroups >= 32 && (uint64_t)p->n_ops >= 8ull * p->nplanes;
        bool use_staged = p->staged_ok && (ctx->pipe_staged == 1 || (ctx->pipe_staged < 0 && reuse));
        if (use_staged) {
            size_t lds = (size_t)ctx->pipe_slots * 8192;
#define LAUNCH_STG(S) do { \
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_counts_staged<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_staged<S>), dim3(nb_to - nb_from), dim3(S * 64), lds, ctx->stream, \
                               (const u64* const*)p->d_udesc, (const u32*)p->d_unblk, p->nplanes, gv ? gv->gmask : (const u32*)p->d_gmask, gv ? gv->gskip : (const u32*)p->d_gskip, \
                               ngroups, nb_from, nb_to - nb_from, ctx->xcd_swz, (u64*)d_counts); } while (0)
            if (ctx->pipe_slots == 8) LAUNCH_STG(8); else LAUNCH_STG(16);
#undef LAUNCH_STG
            KCHK();
            return BMX_OK;
        }
    }
    if (use_split(ctx, p, nitems64)) {
        // few (column, group) items with long operand lists: 8 waves share an item (k_pipe_split, counts mode)
        size_t lds = (size_t)SPLIT_WAVES * 8192;
        auto fn = k_pipe_split<2, SPLIT_WAVES>;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, dim3((u32)nitems64), dim3(SPLIT_WAVES * 64), lds, ctx->stream,
                           (const u64*)p->d_dmat, row_off, and_n, sub_n, p->col_stride, ngroups, nb_from, nb_to, 0, (u64*)d_counts,
                           0, (uint4*)nullptr, (u64*)nullptr, (BlockStat*)nullptr);
        KCHK();
        return BMX_OK;
    }
    if (!p->has_gap) {
        // bit-block-only fast path: (column, group, slice) items
        // Launch plan (pipe_plan): the run is issued as a sequence of launches over column WINDOWS of at most one
        // machine-load of waves each.  Inside a window every wave is in flight at once and the whole chip walks
        // the operand list in step: at any moment HBM serves ONE contiguous stretch of one operand slab (open-page
        // locality) instead of 256 unrelated streams, which is what a single launch over all columns degenerates
        // into after its first round of workgroups (measured: +5 % on the headline, tools/chunk_probe.py; a
        // 1,908-column shard = one window runs at a higher rate than the full-size steady state).
        u32 rows, wg, unroll, window;
        pipe_plan(ctx, nitems64, ngroups, rows, wg, unroll, window);
        u32 parts = 8u / rows;
        if (nitems64 * parts > 0xFFFFFFF0ull) { g_last_error = "too many work items in one run"; return BMX_ERR_RANGE; }
        pipe_bits_fn fn = pipe_bits_kernel(rows, unroll, ctx->pipe_nt != 0, wg);
        if (!fn) { g_last_error = "this (pipe_rows, pipe_unroll, pipe_nt, pipe_wg) shape is not compiled into the library"; return BMX_ERR_BADARG; }
        u32 wpb = wg / 64u;
        u32 ncols = nb_to - nb_from;
        u32 nwin = window ? (ncols + window - 1u) / window : 1u;
        u32 per = (ncols + nwin - 1u) / nwin;                      // even split: no short last window
        for (u32 c0 = 0; c0 < ncols; c0 += per) {
            u32 cols = std::min(per, ncols - c0);
            u32 nitems = cols * ngroups * parts, grid = (nitems + wpb - 1) / wpb;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(wg), (size_t)ctx->pipe_lds, ctx->stream,
                               (const u64*)p->d_dmat, row_off, and_n, p->col_stride, ngroups, nb_from + c0, nitems, ctx->xcd_swz, (u64*)d_counts);
            KCHK();
        }
        return BMX_OK;
    }
    if (nitems64 > 0xFFFFFFF0ull) { g_last_error = "too many (column, group) items in one run"; return BMX_ERR_RANGE; }
    if (use_and_rows(ctx, p)) {
        // every operand block is GAP (or NULL / FULL): the union of the operands' 0-runs read straight from their slabs
        // (and_rows_ipw: several consecutive items -- arg-groups of one column -- per workgroup.  Measured on 16 groups x 2 .. 64 operands
        // over 1e9-bit vectors: 1, 4 and 8 items per workgroup run within 2 % of each other -- such a run is bound by the chain of
        // dependent round trips inside an item (row header, entries, pieces, fold), not by the dispatch rate -- so the default is 1)
        const u32 ipw = ctx->and_rows_ipw > 0 ? (u32)ctx->and_rows_ipw : 1u;
        hipLaunchKernelGGL(and_rows_kernel<AR_COUNT>(ctx), dim3((u32)((nitems64 + ipw - 1u) / ipw)), dim3((u32)ctx->and_rows_wg), 0, ctx->stream,
                           (const u64*)p->d_dmat, row_off, and_n, sub_n, p->col_stride, ngroups, nb_from, (u32)nitems64, ctx->xcd_swz, (u64*)d_counts,
                           (uint4*)nullptr, (u64*)nullptr, (BlockStat*)nullptr, 0u, 0xFFFFFFFFu, and_rows_diag_bits(), ipw);
        KCHK();
        return BMX_OK;
    }
    if (use_gapcount(ctx, p)) {
        // every operand block is GAP (or NULL / FULL): the counting formulation, one 1024-thread workgroup per (column, group)
        size_t lds = (size_t)(16384 * 2 + 2048) * 4;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_counts_gapcount<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_gapcount<false>), dim3((u32)nitems64), dim3(1024), lds, ctx->stream,
                           p->d_dmat, row_off, and_n, sub_n, p->col_stride, ngroups, nb_from, (u32)nitems64, (u64*)d_counts,
                           (uint4*)nullptr, (u64*)nullptr, (BlockStat*)nullptr, 0u, 0xFFFFFFFFu);
        KCHK();
        return BMX_OK;
    }
    // general kernel (GAP operands present): ONE launch unless pipe_window asks for windows.  Measured (256-way AND, mixed
    // 1 % and all-GAP 0.3 %): 2,048-column windows cost 19-38 %, 3,072 are neutral -- columns with GAP operands take
    // unequal time, so a window boundary idles most of the chip while the slowest columns finish.
    size_t lds = 4 * 2048 * 4;
    u32 ncols = nb_to - nb_from;
    u32 window = ctx->pipe_window > 0 ? (u32)ctx->pipe_window : 0u;
    u32 nwin = window ? (ncols + window - 1u) / window : 1u;
    u32 per = (ncols + nwin - 1u) / nwin;
    for (u32 c0 = 0; c0 < ncols; c0 += per) {
        u32 cols = std::min(per, ncols - c0);
        u32 nitems = cols * ngroups;
        u32 grid = (nitems + 3) / 4;
#define LAUNCH_PIPE(U) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts<U>), dim3(grid), dim3(256), lds, ctx->stream, \
        p->d_dmat, row_off, and_n, sub_n, p->col_stride, ngroups, nb_from + c0, nitems, ctx->xcd_swz, (u64*)d_counts)
        switch (ctx->pipe_unroll) {
        case 1: LAUNCH_PIPE(1); break;
        case 2: LAUNCH_PIPE(2); break;
        default: LAUNCH_PIPE(4); break;
        }
#undef LAUNCH_PIPE
        KCHK();
    }
    return BMX_OK;
}

int bmx_pipeline_describe(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, char* buf, size_t buf_len, uint32_t* n_launches)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx && buf && buf_len > 0);
    if (n_launches) *n_launches = 1;
    int rc = pipe_range(p, nb_from, nb_to); if (rc) return rc;
    u64 nitems64 = (u64)(nb_to - nb_from) * p->ngroups;
    bool reuse = p->ngroups >= 32 && (uint64_t)p->n_ops >= 8ull * p->nplanes;
    if (p->h_uids && ctx->gap_pack != 0 && p->cm_gen == ctx->coll_gen && coll_by_id(ctx, p->cm_a_id) && (!p->cm_s_id || coll_by_id(ctx, p->cm_s_id))) {
        if (p->cm_full) snprintf(buf, buf_len, "k_coll_apply<AND_COUNT,512> x 1 launch, %llu workgroups (packed collection of the operand set)", (unsigned long long)nitems64);
        else snprintf(buf, buf_len, "k_coll_members<AND_COUNT> x 1 launch, a wave per (column, group): %u x %u items (members of a packed collection)", nb_to - nb_from, p->ngroups);
    }
    else if (p->staged_ok && (ctx->pipe_staged == 1 || (ctx->pipe_staged < 0 && reuse)))
        snprintf(buf, buf_len, "k_pipe_counts_staged<%d> x 1 launch, %u workgroups", ctx->pipe_slots, nb_to - nb_from);
    else if (use_split(ctx, p, nitems64))
        snprintf(buf, buf_len, "k_pipe_split<2,%d> x 1 launch, %llu workgroups", SPLIT_WAVES, (unsigned long long)nitems64);
    else if (use_and_rows(ctx, p))
        snprintf(buf, buf_len, "k_agg_and_rows<COUNT,%d,%d> x 1 launch, %llu workgroups", ctx->and_rows_wg, ctx->and_rows_depth, (unsigned long long)nitems64);
    else if (use_gapcount(ctx, p))
        snprintf(buf, buf_len, "k_pipe_counts_gapcount x 1 launch, %llu workgroups", (unsigned long long)nitems64);
    else if (p->has_gap)
        snprintf(buf, buf_len, "k_pipe_counts<%d> x 1 launch", ctx->pipe_unroll == 1 ? 1 : ctx->pipe_unroll == 2 ? 2 : 4);
    else {
        u32 rows, wg, unroll, window;
        pipe_plan(ctx, nitems64, p->ngroups, rows, wg, unroll, window);
        u32 ncols = nb_to - nb_from, nwin = (window && ncols) ? (ncols + window - 1u) / window : 1u;
        u32 per = nwin ? (ncols + nwin - 1u) / nwin : 0u;
        snprintf(buf, buf_len, "k_pipe_counts_bits2<%u,%s,%u,%u> x %u launch%s of <= %u columns", unroll, ctx->pipe_nt ? "true" : "false",
                 wg, rows, nwin, nwin == 1 ? "" : "es", per);
        if (n_launches) *n_launches = nwin;
    }
    return BMX_OK;
ABI_END }

int bmx_pipeline_set_search_count_limit(bmx_ctx* ctx, bmx_pipeline* p, uint64_t limit)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx);
    // 0, bm::id_max and the 48-bit id_max all mean "no limit" (the reference's default is id_max, src/bmaggregator.h:338)
    p->search_limit = (limit == 0 || limit == 0xFFFFFFFFull || limit == 0xFFFFFFFFFFFFull) ? ~0ull : limit;
    return BMX_OK;
ABI_END }

int bmx_pipeline_last_windows(const bmx_pipeline* p, uint32_t* launched, uint32_t* planned)
{ ABI_TRY
    ARGCHK(p);
    if (launched) *launched = p->last_windows;
    if (planned) *planned = p->last_windows_planned;
    return BMX_OK;
ABI_END }

} // extern "C"

// The counts run under pipeline::set_search_count_limit (src/bmaggregator.h:255, honoured PER ARG-GROUP at :1362-1367: a group
// whose count has reached the limit is skipped on every following block -- "can find more, cannot find less").  Here: ascending
// launch windows of block columns (each 4 x the one before, as find_first_and_sub walks them); the per-group totals stay on the
// device; after a window k_limit_step (bmx_kernels9.h) drops the groups that have enough from the tables the next window is
// launched over and writes the number of groups left into pinned memory -- the one word the host waits for.  A group returns
// >= min(limit, its true count) and never more than its true count; h_stop[g] = the column at which it stopped.
static int limit_counts_run(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* counts_out)
{
    int rc;
    uint32_t f = nb_from, t = nb_to;
    if ((rc = pipe_range(p, f, t))) return rc;
    const u32 ng = p->ngroups, ncols = t - f;
    if (!p->h_win_groups) p->h_win_groups = new std::vector<uint32_t>();
    if (!p->h_stop) p->h_stop = new std::vector<uint32_t>();
    p->h_win_groups->clear(); p->h_stop->assign(ng, 0xFFFFFFFFu);
    u32 w = std::max<u32>(ncols / 64u, 16u), planned = 0;
    for (u32 c = 0, ww = w; c < ncols; c += ww, ww *= 4u) ++planned;
    p->last_windows_planned = std::max(planned, 1u); p->last_windows = 0;
    if (counts_out) for (u32 g = 0; g < ng; ++g) counts_out[g] = 0;
    if (!ncols) return BMX_OK;
    // collections / staged tables in force (resolved once, before the first window)
    bmx_coll *ca = nullptr, *cs = nullptr;
    if (p->h_uids && (rc = pipe_resolve_colls(ctx, p, true, &ca, &cs))) return rc;
    CollPin pin; pin.pin(ca, cs);                              // (the member table captured below must stay what the windows resolve: no eviction mid-run)
    const bool have_cg = ca && !p->cm_full && p->cm_buf;
    const bool have_masks = p->staged_ok && p->d_gmask && p->d_gskip;
    const size_t nch = std::max<u32>(p->nchunks, 1u);
    // device state: totals, the window's compact counts, stop columns, two active lists, the compacted tables
    const size_t words = (size_t)ng * (2 + 2 + 1 + 2 + 3 + (have_cg ? 4 : 0) + (have_masks ? nch + 1 : 0)) + 16;
    u32* buf = nullptr;
    if ((rc = dmalloc(ctx, (void**)&buf, words * 4))) return rc;
    u64* d_tot = (u64*)buf; u64* d_cc = d_tot + ng;
    u32* d_stop = (u32*)(d_cc + ng); u32* d_act[2] = {d_stop + ng, d_stop + 2 * (size_t)ng};
    u32* d_ro = d_act[1] + ng; u32* d_an = d_ro + ng; u32* d_sn = d_an + ng;
    u32* q = d_sn + ng;
    q = (u32*)(((uintptr_t)q + 15u) & ~(uintptr_t)15u);
    CollGroup* d_cg = nullptr; u32* d_gm = nullptr; u32* d_gs = nullptr;
    if (have_cg) { d_cg = (CollGroup*)q; q += (size_t)ng * 4; }
    if (have_masks) { d_gm = q; q += (size_t)ng * nch; d_gs = q; q += ng; }
    u32* d_nact = q;
    hipError_t e = hipMemsetAsync(d_tot, 0, (size_t)ng * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_stop, 0xFF, (size_t)ng * 4, ctx->stream);
    if (e != hipSuccess) { dfree(ctx, buf); return fail_hip(e, "limit_counts_run", __LINE__); }
    LimitTables lt{p->d_meta, p->d_meta + ng, p->d_meta + 2 * (size_t)ng, d_ro, d_an, d_sn,
                   have_cg ? (const CollGroup*)((const char*)p->cm_buf + p->cm_groups_off) : nullptr, d_cg,
                   have_masks ? (const u32*)p->d_gmask : nullptr, have_masks ? (const u32*)p->d_gskip : nullptr, d_gm, d_gs, (u32)nch};
    GroupView gv{d_ro, d_an, d_sn, ng, d_gm, d_gs, d_cg};
    u32 n_active = ng; int cur = 0; bool first = true;
    for (u32 c = 0; c < ncols && n_active && !rc; c += w, w *= 4u) {
        const u32 c1 = (u32)std::min<u64>((u64)c + w, ncols);
        gv.ngroups = n_active;
        rc = pipeline_run_counts_impl(ctx, p, f + c, f + c1, d_cc, false, first ? nullptr : &gv);
        if (rc) break;
        hipLaunchKernelGGL(k_limit_step, dim3(1), dim3(1024), 0, ctx->stream, d_tot, (const u64*)d_cc, first ? (const u32*)nullptr : (const u32*)d_act[cur], n_active,
                           p->search_limit, f + c1, d_stop, d_act[cur ^ 1], lt, d_nact, ctx->h_small + 60);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);        // (the word in pinned memory is all the host reads)
        if (e != hipSuccess) { rc = fail_hip(e, "k_limit_step", __LINE__); break; }
        p->h_win_groups->push_back(n_active);
        ++p->last_windows;
        n_active = (u32)ctx->h_small[60];
        cur ^= 1; first = false;
    }
    if (!rc) {
        e = hipMemcpyAsync(p->h_stop->data(), d_stop, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && counts_out) e = hipMemcpyAsync(counts_out, d_tot, (size_t)ng * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail_hip(e, "limit_counts_run readback", __LINE__);
    } else (void)hipStreamSynchronize(ctx->stream);
    dfree(ctx, buf);
    return rc;
}

// The asynchronous entry under a limit: the same ascending windows, ALL enqueued, nothing read back.  Every window runs over all
// arg-groups through private copies of the group tables; k_limit_null (bmx_kernels9.h) points the groups that have enough at null
// entries after each window, so their items of the later windows end at a header.  (A pipeline served as ONE whole packed
// collection -- cm_full -- has a single fused group with no table entry to null: it runs to the end and returns its true count.)
static int limit_counts_run_dev(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts)
{
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t f = nb_from, t = nb_to;
    if ((rc = pipe_range(p, f, t))) return rc;
    const u32 ng = p->ngroups, ncols = t - f;
    HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)std::max(ng, 1u) * 8, ctx->stream));
    if (!ncols || !ng) return BMX_OK;
    bmx_coll *ca = nullptr, *cs = nullptr;
    if (p->h_uids && (rc = pipe_resolve_colls(ctx, p, false, &ca, &cs))) return rc;
    CollPin pin; pin.pin(ca, cs);
    const bool have_cg = ca && !p->cm_full && p->cm_buf;
    const bool have_masks = p->staged_ok && p->d_gmask && p->d_gskip;
    // totals, the window's counts, row offsets, (member ranges), (gskip)
    const size_t words = (size_t)ng * (2 + 2 + 1 + (have_cg ? 4 : 0) + (have_masks ? 1 : 0)) + 16;
    u32* buf = nullptr;
    if ((rc = dmalloc(ctx, (void**)&buf, words * 4))) return rc;
    u64* d_tot = (u64*)buf; u64* d_cc = d_tot + ng;
    u32* d_ro = (u32*)(d_cc + ng);
    u32* q = d_ro + ng;
    q = (u32*)(((uintptr_t)q + 15u) & ~(uintptr_t)15u);
    CollGroup* d_cg = nullptr; u32* d_gs = nullptr;
    if (have_cg) { d_cg = (CollGroup*)q; q += (size_t)ng * 4; }
    if (have_masks) { d_gs = q; q += ng; }
    hipError_t e = hipMemsetAsync(d_tot, 0, (size_t)ng * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ro, p->d_meta, (size_t)ng * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && have_cg) e = hipMemcpyAsync(d_cg, (const char*)p->cm_buf + p->cm_groups_off, (size_t)ng * sizeof(CollGroup), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && have_masks) e = hipMemcpyAsync(d_gs, p->d_gskip, (size_t)ng * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { dfree(ctx, buf); return fail_hip(e, "limit_counts_run_dev", __LINE__); }
    GroupView gv{d_ro, p->d_meta + ng, p->d_meta + 2 * (size_t)ng, ng, have_masks ? (const u32*)p->d_gmask : nullptr, d_gs, d_cg};
    u32 w = std::max<u32>(ncols / 64u, 16u);
    for (u32 c = 0; c < ncols && !rc; c += w, w *= 4u) {
        const u32 c1 = (u32)std::min<u64>((u64)c + w, ncols);
        rc = pipeline_run_counts_impl(ctx, p, f + c, f + c1, d_cc, false, &gv);
        if (rc) break;
        hipLaunchKernelGGL(k_limit_null, dim3((ng + 255u) / 256u), dim3(256), 0, ctx->stream, d_tot, (const u64*)d_cc, ng, p->search_limit, p->null_row_off,
                           d_ro, d_cg, d_gs, c1 >= ncols ? (u64*)d_counts : (u64*)nullptr);
        e = hipGetLastError();
        if (e != hipSuccess) { rc = fail_hip(e, "k_limit_null", __LINE__); break; }
    }
    dfree(ctx, buf);                                               // (pooled: handed only to work enqueued behind this run)
    return rc;
}

extern "C" {

// groups every launched window of the last synchronous counts run under a limit ran over (out[0 .. min(cap, n))), n = windows launched
int bmx_pipeline_last_window_groups(const bmx_pipeline* p, uint32_t* out, uint32_t cap, uint32_t* n)
{ ABI_TRY
    ARGCHK(p && (out || !cap));
    const uint32_t have = p->h_win_groups ? (uint32_t)p->h_win_groups->size() : 0u;
    for (uint32_t i = 0; i < have && i < cap; ++i) out[i] = (*p->h_win_groups)[i];
    if (n) *n = have;
    return BMX_OK;
ABI_END }

int bmx_pipeline_run_counts(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* counts_out)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx && counts_out);
    int rc = set_dev(ctx); if (rc) return rc;
    p->last_windows = p->last_windows_planned = 1;
    if (p->search_limit != ~0ull) return limit_counts_run(ctx, p, nb_from, nb_to, counts_out);
    u64* d_counts = nullptr;
    size_t bytes = (size_t)p->ngroups * 8;
    if (p->ngroups <= 64) d_counts = ctx->d_small;
    else if ((rc = dmalloc(ctx, (void**)&d_counts, bytes))) return rc;
    rc = pipeline_run_counts_impl(ctx, p, nb_from, nb_to, d_counts, true);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(counts_out, d_counts, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail_hip(e, "counts readback", __LINE__);
    }
    if (p->ngroups > 64) dfree(ctx, d_counts);
    return rc;
ABI_END }

int bmx_pipeline_operand_bytes(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* bytes)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx && bytes);
    int rc = set_dev(ctx); if (rc) return rc;
    if ((rc = pipe_range(p, nb_from, nb_to))) return rc;
    *bytes = 0;
    u64 nitems64 = (u64)(nb_to - nb_from) * p->ngroups;
    if (!nitems64) return BMX_OK;
    HIPCHK(hipMemsetAsync(ctx->d_small, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_pipe_bytes, dim3((u32)((nitems64 + 255) / 256)), dim3(256), 0, ctx->stream,
                       p->d_dmat, p->d_meta, p->d_meta + p->ngroups, p->d_meta + 2 * p->ngroups,
                       p->col_stride, p->ngroups, nb_from, (u32)nitems64, ctx->d_small);
    KCHK();
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *bytes = ctx->h_small[0];
    return BMX_OK;
ABI_END }

} // extern "C"

// ---------------------------------------------------------------------------
// materialised results: full-size slab (one slot per block column), see
// store_result() in bmx_kernels2.h
// ---------------------------------------------------------------------------
static int result_begin(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks, bmx_vec** out, BlockStat** st, u32** offs)
{
    int rc;
    // st[nblocks] + offs[nblocks] (+ nblocks words of slack behind them: round 4's candidate list lived there)
    size_t aux_need = (size_t)nblocks * (sizeof(BlockStat) + 4 + 4) + 64;
    if ((rc = ensure(ctx, &ctx->aux, &ctx->aux_bytes, aux_need))) return rc;
    *st = (BlockStat*)ctx->aux;
    *offs = (u32*)((char*)ctx->aux + (size_t)nblocks * sizeof(BlockStat));
    bmx_vec* v = vec_alloc_host(ctx, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    size_t b_desc = (size_t)std::max<uint32_t>(nblocks, 1) * 8, b_bits = (size_t)nblocks * 8192;
    if ((rc = dmalloc(ctx, (void**)&v->d_desc, b_desc)) || (rc = dmalloc(ctx, (void**)&v->d_bits, b_bits))) { bmx_vec_free(ctx, v); return rc; }
    v->n_bit = nblocks;
    v->bytes = std::max<size_t>(b_desc, 16) + std::max<size_t>(b_bits, 16);
    *out = v;
    return BMX_OK;
}

// Memory of a result: the full-size slab lives only while the result is being produced.  When fewer than 7/8 of
// its slots hold a bit-block the survivors are moved into a right-sized slab (k_compact_bits; the ordinals come
// from the layout scan) and the big one goes back to the pool: a sparse result of a 1e9-bit operation costs what
// it holds, not 125 MB, and bmx_pipeline_run_results over G groups needs (sum of the live blocks) + ONE transient
// slab.  A nearly full slab is kept as it is (no second pass over the blocks); its ordinals are kept in d_ord so
// that bmx_vec_download moves only live blocks.
static int result_finish(bmx_ctx* ctx, bmx_vec* v, BlockStat* st, u32* offs, bool gaps_done = false /* the kernel wrote its GAP blocks itself: only the bit slab is laid out */)
{
    int rc;
    uint32_t nblocks = v->nblocks;
    if (!nblocks) return BMX_OK;
    hipLaunchKernelGGL(k_scan_layout, dim3(1), dim3(1024), 0, ctx->stream, st, nblocks, offs, ctx->d_small);
    KCHK();
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 6 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t gap_words = gaps_done ? 0 : ctx->h_small[1];
    for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
    bool pending = false;
    if (gap_words) {
        size_t b_gaps = (size_t)gap_words * 2 + 64;      // + guard, see vec_alloc_device
        if ((rc = dmalloc(ctx, (void**)&v->d_gaps, b_gaps))) return rc;
        v->bytes += std::max<size_t>(b_gaps, 16);
        v->gap_words = gap_words;
        hipLaunchKernelGGL(k_emit_gaps, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           v->d_bits, nblocks, st, offs, v->d_gaps, v->d_desc);
        KCHK();
        pending = true;
    }
    uint32_t live = v->counts[BMX_BIT];
    uint4* old_slab = nullptr;
    if (live && (uint64_t)live * 8u < (uint64_t)nblocks * 7u) {
        uint4* packed = nullptr;
        if ((rc = dmalloc(ctx, (void**)&packed, (size_t)live * 8192))) return rc;
        hipLaunchKernelGGL(k_compact_bits, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint4*)v->d_bits, nblocks, (const BlockStat*)st, (const u32*)offs, packed, v->d_desc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { dfree(ctx, packed); return fail_hip(e, "k_compact_bits", __LINE__); }
        old_slab = v->d_bits; v->d_bits = packed; v->n_bit = live;
        pending = true;
    } else if (live && live < nblocks) {
        if ((rc = dmalloc(ctx, (void**)&v->d_ord, (size_t)nblocks * 4))) return rc;
        HIPCHK(hipMemcpyAsync(v->d_ord, offs, (size_t)nblocks * 4, hipMemcpyDeviceToDevice, ctx->stream));
        pending = true;
    }
    // Nothing enqueued above is waited for: everything it reads (st / offs in the context's scratch, the old slab) is next
    // touched by work that is enqueued BEHIND it on the same stream -- the scratch by the next operation (ensure() synchronises
    // before it ever re-allocates), a pooled block by whoever is handed it next (hipFree, when the pool overflows, synchronises
    // the device) -- and the host needs nothing more from the device here: the kinds came with the synchronise above.
    (void)pending;
    if (old_slab) dfree(ctx, old_slab);
    if (live == 0) {                          // nothing lives in the slab: give it back
        dfree(ctx, v->d_bits); v->d_bits = nullptr; v->n_bit = 0;
    }
    return BMX_OK;
}

extern "C" {

// block-for-block copy of a device vector (bvector::operator=): same kinds, own slabs
static int vec_clone(bmx_ctx* ctx, const bmx_vec* a, bmx_vec** out)
{
    bmx_vec* v = vec_alloc_host(ctx, a->nbits, a->nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    int rc;
    size_t b_bits = a->d_bits ? (size_t)a->n_bit * 8192 : 0, b_gaps = a->d_gaps ? (size_t)a->gap_words * 2 + 64 : 0;
    size_t b_desc = (size_t)std::max<uint32_t>(a->nblocks, 1) * 8;
    if ((rc = dmalloc(ctx, (void**)&v->d_desc, b_desc)) || (b_bits && (rc = dmalloc(ctx, (void**)&v->d_bits, b_bits))) ||
        (b_gaps && (rc = dmalloc(ctx, (void**)&v->d_gaps, b_gaps)))) { bmx_vec_free(ctx, v); return rc; }
    v->n_bit = a->d_bits ? a->n_bit : 0; v->gap_words = a->d_gaps ? a->gap_words : 0;
    memcpy(v->counts, a->counts, sizeof(v->counts));
    v->bytes = std::max<size_t>(b_desc, 16) + std::max<size_t>(b_bits, 16) + std::max<size_t>(b_gaps, 16);
    hipError_t e = hipSuccess;
    if (b_bits) e = hipMemcpyAsync(v->d_bits, a->d_bits, b_bits, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && b_gaps) e = hipMemcpyAsync(v->d_gaps, a->d_gaps, b_gaps, hipMemcpyDeviceToDevice, ctx->stream);
    v->ord_lazy = a->ord_lazy;
    if (e == hipSuccess && a->d_ord && a->nblocks) {
        if ((rc = dmalloc(ctx, (void**)&v->d_ord, (size_t)a->nblocks * 4))) { bmx_vec_free(ctx, v); return rc; }
        e = hipMemcpyAsync(v->d_ord, a->d_ord, (size_t)a->nblocks * 4, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess && a->nblocks) {
        hipLaunchKernelGGL(k_rebase_desc, dim3((a->nblocks + 255) / 256), dim3(256), 0, ctx->stream, a->d_desc, v->d_desc, a->nblocks,
                           (u64)(uintptr_t)a->d_bits, (u64)(uintptr_t)v->d_bits, (u64)(uintptr_t)a->d_gaps, (u64)(uintptr_t)v->d_gaps);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "vec_clone", __LINE__); }
    if (a->count_valid) { v->count = a->count; v->count_valid = true; }
    *out = v;
    return BMX_OK;
}

} // extern "C"

// k_op2_loop's launch: op2_loop N > 0 = N workgroups of 4 waves per CU, -1 = 4 of them.  (Measured and dropped, profiles/r05_pair: ONE
// 16-wave workgroup per CU when GAP results can come out, so that the conversions of its tail spread over 16 waves -- the tail
// then starts when the slowest of 16 waves has left the column loop: AND 0.078 -> 0.082 ms, SUB 0.085 -> 0.086 ms.)
static void op2_loop_launch(bmx_ctx* ctx, int op, const bmx_vec* va, const bmx_vec* vb, u32 nblocks, int opt_compress, bmx_vec* v, BlockStat* st,
                            FoldOut fo, u32* offs, u16* gap_slab)
{
    const u32 wgs = (u32)(ctx->op2_loop > 0 ? ctx->op2_loop : 4);        // workgroups per CU = waves per SIMD
    const u32 grid = std::min<u32>((nblocks + 3u) / 4u, 256u * wgs);
    auto fn = (ctx->op2_nt & 1) ? k_op2_loop<4, true> : k_op2_loop<4, false>;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, ctx->stream, op, va->d_desc, va->nblocks, vb->d_desc, vb->nblocks, nblocks, opt_compress,
                       v->d_bits, v->d_desc, st, fo, offs, ctx->d_cursor, gap_slab);
}

// A GAP slab allocated at the operands' bound (the words a pairwise result can hold at most: a copied GAP block, a GAP x GAP
// result of at most len(a) + len(b) runs) and filled by the kernel up to `used`: given back when nothing landed in it, moved
// into a slab of its own size -- a copy and a descriptor rebase, enqueued -- when the slack is worth it (more than
// `slack_ok` bytes), kept otherwise.  v->bytes does not hold the slab yet.
static int gap_slab_trim(bmx_ctx* ctx, bmx_vec* v, uint64_t bound, uint64_t used, size_t slack_ok)
{
    int rc;
    if (!used) { dfree(ctx, v->d_gaps); v->d_gaps = nullptr; v->gap_words = 0; return BMX_OK; }
    if (bound > 2 * used + 4096 && (size_t)(bound - used) * 2 > slack_ok) {
        u16* small_ = nullptr;
        if ((rc = dmalloc(ctx, (void**)&small_, (size_t)used * 2 + 64))) return rc;
        hipError_t e = hipMemcpyAsync(small_, v->d_gaps, (size_t)used * 2, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_rebase_desc, dim3((v->nblocks + 255) / 256), dim3(256), 0, ctx->stream, (const u64*)v->d_desc, v->d_desc, v->nblocks,
                               (u64)(uintptr_t)v->d_bits, (u64)(uintptr_t)v->d_bits, (u64)(uintptr_t)v->d_gaps, (u64)(uintptr_t)small_);
            e = hipGetLastError();
        }
        if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); dfree(ctx, small_); return fail_hip(e, "gap_slab_trim", __LINE__); }
        dfree(ctx, v->d_gaps);
        v->d_gaps = small_; v->gap_words = used; v->bytes += (size_t)used * 2 + 64;
    } else { v->gap_words = used; v->bytes += (size_t)bound * 2 + 64; }
    return BMX_OK;
}

// The tail of a pairwise operation whose kernel wrote its GAP blocks itself (k_op2_loop with a cursor and a slab at the operands'
// bound; the stream has been synchronised): kinds in h_small[2..5], the cursor in h_small[6].  A bit slab sparse enough to be
// compacted takes the scan path (result_finish) as before; a nearly full one is kept, its ordinals left to the first download --
// and with it a GAP slab whose slack stays under a quarter of the bytes the vector keeps in bit-blocks anyway.
static int op2_finish_laid_out(bmx_ctx* ctx, bmx_vec* v, BlockStat* st, u32* offs, uint64_t bound)
{
    const uint32_t nblocks = v->nblocks;
    uint32_t counts[4];
    for (int k = 0; k < 4; ++k) counts[k] = (uint32_t)ctx->h_small[2 + k];
    const uint64_t used = ctx->h_small[6] & 0xFFFFFFFFFFull, ncand = ctx->h_small[6] >> 40;      // the cursor: words | GAP blocks << 40
    if ((uint64_t)counts[0] + counts[1] + counts[2] + counts[3] != nblocks || ncand != counts[BMX_GAP] || (used == 0) != (ncand == 0) || used > bound) {
        g_last_error = "bmx_op2: inconsistent fold of the result block kinds"; return BMX_ERR_DEVICE;
    }
    const uint32_t live = counts[BMX_BIT];
    const bool sparse = live && live < nblocks && (uint64_t)live * 8u < (uint64_t)nblocks * 7u;
    int rc = gap_slab_trim(ctx, v, bound, used, sparse ? 0 : (size_t)live * 2048u);
    if (rc) return rc;
    if (sparse) return result_finish(ctx, v, st, offs, true);
    memcpy(v->counts, counts, sizeof(counts));
    if (live == 0) { dfree(ctx, v->d_bits); v->d_bits = nullptr; v->n_bit = 0; }
    else if (live < nblocks) v->ord_lazy = true;
    return BMX_OK;
}

extern "C" {

int bmx_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int opt_compress, bmx_vec** result)
{ ABI_TRY
    ARGCHK(ctx && a && b && result && a->ctx == ctx && b->ctx == ctx && op >= BMX_AND && op <= BMX_SUB);
    *result = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t nblocks = std::max(a->nblocks, b->nblocks);
    uint64_t nbits = std::max(a->nbits, b->nbits);                  // src/bm.h:6219-6221
    if (a == b) {
        // aliasing as the reference handles it up front: AND / OR of a vector with itself is a block-for-block copy
        // (src/bm.h:6191-6195, 5984-5988), XOR / SUB are empty (:6081, 6412); nothing is re-optimised
        if (op == BMX_AND || op == BMX_OR) return vec_clone(ctx, a, result);
        bmx_vec* v; BlockStat* st; u32* offs;
        if ((rc = result_begin(ctx, nbits, nblocks, &v, &st, &offs))) return rc;
        if (nblocks) {
            hipError_t e = hipMemsetAsync(v->d_desc, 0, (size_t)nblocks * 8, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "bmx_op2", __LINE__); }
        }
        v->counts[BMX_NULL] = nblocks;
        dfree(ctx, v->d_bits); v->d_bits = nullptr; v->n_bit = 0;
        *result = v;
        return BMX_OK;
    }
    bmx_vec* v; BlockStat* st; u32* offs;
    if ((rc = result_begin(ctx, nbits, nblocks, &v, &st, &offs))) return rc;
    // no GAP block can come out (neither operand holds one, no re-compression): the kernel folds the kind counts itself
    // and the layout scan is skipped -- k_op2, one synchronise, done -- unless result blocks vanished (then the scan /
    // compaction path below decides what to do with the slab)
    bool no_gap = !opt_compress && a->counts[BMX_GAP] == 0 && b->counts[BMX_GAP] == 0, folded = false, emit = false, counted = false;
    uint64_t gap_bound = 0;
    if (nblocks) {
        // bit-blocks only on both sides: the streaming form (one machine-load of waves, each owning a stretch of columns)
        const bool stream = no_gap && ctx->pair_stream != 0 && a->nblocks == b->nblocks && a->counts[BMX_BIT] == nblocks &&
                            b->counts[BMX_BIT] == nblocks && nblocks >= 2048u;
        if (stream) {
            const u32 waves = 4u, total = 256u * waves * (u32)std::max(ctx->op2_wgs, 1);
            const u32 per_wave = (nblocks + total - 1u) / total;
            const u32 grid = ((nblocks + per_wave - 1u) / per_wave + waves - 1u) / waves;
            auto fn = ctx->op2_nt == 3 ? k_op2_stream<4, true, true> : ctx->op2_nt == 2 ? k_op2_stream<4, false, true>
                    : ctx->op2_nt == 1 ? k_op2_stream<4, true, false> : k_op2_stream<4, false, false>;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, ctx->stream, op, a->d_desc, b->d_desc, nblocks, per_wave,
                               v->d_bits, v->d_desc, st, FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + 2});
        } else if (ctx->op2_loop != 0 && nblocks >= 2048u) {
            // any block kinds, long vectors: the persistent form (one memory round trip per column, GAP blocks decoded from registers)
            // Without re-compression the kernel also lays its GAP candidates out (a bump cursor instead of the layout scan), converts
            // them in its tail into a slab sized at the operands' bound, and folds the kinds: nothing is left after the one
            // synchronise -- unless the bit slab turns out sparse enough to be compacted, which takes the scan path as before.
            // (The 16-bit kind counters of a fold slot hold 64 x 65,535 blocks.)
            emit = !opt_compress && !no_gap && nblocks <= 2000000u;
            if (emit) {
                gap_bound = (a->counts[BMX_GAP] ? a->gap_words : 0) + (b->counts[BMX_GAP] ? b->gap_words : 0) + 8u;
                if ((rc = dmalloc(ctx, (void**)&v->d_gaps, (size_t)gap_bound * 2 + 64))) { bmx_vec_free(ctx, v); return rc; }
            }
            // the kinds are folded whatever the operands hold: when every block came out as a bit-block (OR / XOR of two 1 % vectors:
            // their GAP x GAP results pass the 1,276-run limit) there is nothing for the layout scan to lay out
            op2_loop_launch(ctx, op, a, b, nblocks, opt_compress, v, st,
                            (emit || no_gap || op == BMX_OR || op == BMX_XOR) ? FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + 2} : FoldOut{nullptr, nullptr, nullptr},
                            emit ? offs : (u32*)nullptr, emit ? v->d_gaps : (u16*)nullptr);
            folded = emit || op == BMX_OR || op == BMX_XOR;                 // (with re-compression AND / SUB go straight to the layout scan, no extra synchronise)
        } else {
        // short vectors: a wave per column; the kernel also folds the popcount of its result (bvector::bit_and + count(), the
        // plumbing case of BASELINE configs[0], is then ONE launch and one synchronise: bmx_count finds the count with the vector)
        hipLaunchKernelGGL(k_op2, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, op,
                           a->d_desc, a->nblocks, b->d_desc, b->nblocks, nblocks, opt_compress,
                           v->d_bits, v->d_desc, st,
                           no_gap ? FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + 2} : FoldOut{nullptr, nullptr, nullptr},
                           FoldOut{ctx->d_slots2, ctx->d_done2, ctx->h_small + 8});
        counted = true;
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && (no_gap || folded)) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "k_op2", __LINE__); }
        if (emit) {
            rc = op2_finish_laid_out(ctx, v, st, offs, gap_bound);
            if (rc) { bmx_vec_free(ctx, v); return rc; }
            *result = v;
            return BMX_OK;
        }
        if ((no_gap || folded) && ctx->h_small[2 + BMX_GAP] == 0 && ctx->h_small[2 + BMX_BIT] == nblocks) {
            for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
            if (counted) { v->count = ctx->h_small[8]; v->count_valid = true; }
            *result = v;
            return BMX_OK;
        }
    }
    if ((rc = result_finish(ctx, v, st, offs))) { bmx_vec_free(ctx, v); return rc; }       // (synchronises: the folded count has arrived too)
    if (counted) { v->count = ctx->h_small[8]; v->count_valid = true; }
    *result = v;
    return BMX_OK;
ABI_END }

// bit_and/or/xor/sub + count() in one call (bmx_op2(ctx, op, hA, hB, want_result, &hR, &count)): result
// may be NULL (count only: bm::count_*, src/bmalgo.h:49-149).  Short vectors take one launch for both; long ones the
// streaming kernels followed by the count pass.
int bmx_op2_count(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int opt_compress, bmx_vec** result, uint64_t* count)
{ ABI_TRY
    ARGCHK(ctx && a && b && count);
    if (!result) return bmx_count_op2(ctx, op, a, b, count);
    int rc = bmx_op2(ctx, op, a, b, opt_compress, result);
    if (rc) return rc;
    rc = bmx_count(ctx, *result, count);
    if (rc) { bmx_vec_free(ctx, *result); *result = nullptr; }
    return rc;
ABI_END }

// ---- asynchronous 3-operand operations (bmx_op2_dev / bmx_pending_wait / bmx_pending_free) ----
// What the synchronous bmx_op2 waits for is not the result -- that is complete on the stream when the kernel ends -- but the
// COUNTS of its block kinds, which decide on the host how the next operation over it is dispatched.  Operands without GAP
// blocks under opt_none cannot produce a GAP block, so their result needs no layout pass at all: the kernel folds the kind
// counts into a pinned slot, the call returns at once, and a later operation that takes the unresolved result as an operand
// simply runs the kernel that asks nothing of its operands' kinds (k_op2_loop / k_op2).  One synchronise resolves a whole chain.
int bmx_op2_dev(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_pending* pa, const bmx_vec* b, const bmx_pending* pb, bmx_pending** out)
{ ABI_TRY
    ARGCHK(ctx && out && op >= BMX_AND && op <= BMX_SUB && ((a != nullptr) != (pa != nullptr)) && ((b != nullptr) != (pb != nullptr)));
    *out = nullptr;
    ARGCHK((!a || a->ctx == ctx) && (!b || b->ctx == ctx) && (!pa || pa->ctx == ctx) && (!pb || pb->ctx == ctx));
    // operands that may hold GAP blocks: the result may hold GAP blocks too -- at most the operands' GAP words together
    // (a copied GAP block, a GAP x GAP result of at most len(a) + len(b) runs) -- so its GAP slab is allocated at that bound
    // and the kernel converts its GAP candidates into it before it ends: the descriptors are complete on the stream and the
    // result can be an operand at once; bmx_pending_wait trims the slab
    const uint64_t gap_bound = (a ? (a->counts[BMX_GAP] ? a->gap_words : 0) : pa->gap_bound) + (b ? (b->counts[BMX_GAP] ? b->gap_words : 0) : pb->gap_bound);
    int rc = set_dev(ctx); if (rc) return rc;
    const bmx_vec* va = a ? a : pa->v; const bmx_vec* vb = b ? b : pb->v;
    const uint32_t nblocks = std::max(va->nblocks, vb->nblocks);
    const uint64_t nbits = std::max(va->nbits, vb->nbits);
    if (nblocks > 2000000u) { g_last_error = "bmx_op2_dev: more than 2,000,000 blocks (the in-kernel fold of the kind counts holds 64 x 65,535): use bmx_op2"; return BMX_ERR_RANGE; }
    int slot = -1;
    for (int i = 0; i < PEND_SLOTS; ++i) if (!(ctx->pend_used[i >> 6] >> (i & 63) & 1ull)) { slot = i; break; }
    if (slot < 0) { g_last_error = "bmx_op2_dev: 1,024 unresolved results are outstanding (bmx_pending_wait / bmx_pending_free them)"; return BMX_ERR_RANGE; }
    bmx_pending* p = new (std::nothrow) bmx_pending();
    if (!p) return BMX_ERR_BADALLOC;
    p->ctx = ctx; p->v = nullptr; p->slot = slot; p->ev = nullptr; p->gap_bound = 0; p->scratch = nullptr; p->resolved = false;
    if (((a && a == b) || (pa && pa == pb)) && (op == BMX_AND || op == BMX_OR)) {
        // aliasing as the reference handles it up front (src/bm.h:6191-6195, 5984-5988): x & x, x | x are block-for-block copies,
        // nothing is re-classified.  An unresolved operand is waited for here (its kind counts are the copy's).
        const bmx_vec* src = a ? a : pa->v;
        uint32_t cnt[4]; uint64_t used = src->gap_words;
        memcpy(cnt, src->counts, sizeof(cnt));
        if (pa && !pa->resolved) {
            hipError_t ew = hipEventSynchronize(pa->ev);
            if (ew != hipSuccess) { delete p; return fail_hip(ew, "bmx_op2_dev (alias)", __LINE__); }
            const u64* ps = ctx->h_pend + (size_t)pa->slot * 8;
            for (int k = 0; k < 4; ++k) cnt[k] = (uint32_t)ps[k];
            used = ps[4] & 0xFFFFFFFFFFull;
        }
        bmx_vec* c = nullptr;
        if ((rc = vec_clone(ctx, src, &c))) { delete p; return rc; }
        memcpy(c->counts, cnt, sizeof(cnt));
        c->count_valid = false;
        if (c->d_gaps) c->gap_words = used;                                  // (an unresolved source: its slab is sized at the bound, its data end at `used`)
        if (c->d_bits && c->n_bit == c->nblocks && cnt[BMX_BIT] < c->nblocks && !c->d_ord) c->ord_lazy = true;
        hipError_t ee = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
        if (ee == hipSuccess) ee = hipEventRecord(p->ev, ctx->stream);
        if (ee != hipSuccess) { bmx_vec_free(ctx, c); delete p; return fail_hip(ee, "bmx_op2_dev (alias)", __LINE__); }
        p->v = c; p->resolved = true; p->gap_bound = c->d_gaps ? c->gap_words : 0;
        ctx->pend_used[slot >> 6] |= 1ull << (slot & 63);
        *out = p;
        return BMX_OK;
    }
    bmx_vec* v; BlockStat* st; u32* offs;
    if ((rc = result_begin(ctx, nbits, nblocks, &v, &st, &offs))) { delete p; return rc; }
    u16* gap_slab = nullptr;
    if (gap_bound && nblocks) {
        // this result's own st[] / offs[] / candidate list (the context's scratch is the next operation's) and its GAP slab
        const uint64_t bound = gap_bound + 8u;
        if ((rc = dmalloc(ctx, &p->scratch, (size_t)nblocks * (sizeof(BlockStat) + 8) + 64)) || (rc = dmalloc(ctx, (void**)&gap_slab, (size_t)bound * 2 + 64))) {
            dfree(ctx, p->scratch); bmx_vec_free(ctx, v); delete p; return rc;
        }
        st = (BlockStat*)p->scratch; offs = (u32*)((char*)p->scratch + (size_t)nblocks * sizeof(BlockStat));
        v->d_gaps = gap_slab; v->gap_words = bound; v->bytes += (size_t)bound * 2 + 64;
        p->gap_bound = bound;
    }
    u64* hs = ctx->h_pend + (size_t)slot * 8;
    for (int k = 0; k < 8; ++k) hs[k] = 0;
    hipError_t e = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
    if (e != hipSuccess) { dfree(ctx, p->scratch); bmx_vec_free(ctx, v); delete p; return fail_hip(e, "hipEventCreate", __LINE__); }
    const FoldOut fo{ctx->d_slots, ctx->d_done, hs};
    const bool same = (a && b && a == b) || (pa && pb && pa == pb);
    if (!nblocks) hs[BMX_NULL] = 0;
    else if (same && (op == BMX_XOR || op == BMX_SUB)) {                    // x ^ x, x - x: empty (src/bm.h:6081, 6412)
        e = hipMemsetAsync(v->d_desc, 0, (size_t)nblocks * 8, ctx->stream);
        hs[BMX_NULL] = nblocks;
    } else {
        const bool stream = a && b && ctx->pair_stream != 0 && a->nblocks == b->nblocks && a->counts[BMX_BIT] == nblocks &&
                            b->counts[BMX_BIT] == nblocks && nblocks >= 2048u;
        if (gap_slab) {
            op2_loop_launch(ctx, op, va, vb, nblocks, 0, v, st, fo, offs, gap_slab);
        } else if (stream) {
            const u32 waves = 4u, total = 256u * waves * (u32)std::max(ctx->op2_wgs, 1);
            const u32 per_wave = (nblocks + total - 1u) / total;
            const u32 grid = ((nblocks + per_wave - 1u) / per_wave + waves - 1u) / waves;
            auto fn = ctx->op2_nt == 3 ? k_op2_stream<4, true, true> : ctx->op2_nt == 2 ? k_op2_stream<4, false, true>
                    : ctx->op2_nt == 1 ? k_op2_stream<4, true, false> : k_op2_stream<4, false, false>;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, ctx->stream, op, va->d_desc, vb->d_desc, nblocks, per_wave, v->d_bits, v->d_desc, st, fo);
        } else if (ctx->op2_loop != 0 && nblocks >= 2048u) {
            op2_loop_launch(ctx, op, va, vb, nblocks, 0, v, st, fo, nullptr, nullptr);
        } else
            hipLaunchKernelGGL(k_op2, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, op, va->d_desc, va->nblocks, vb->d_desc, vb->nblocks, nblocks, 0,
                               v->d_bits, v->d_desc, st, fo, FoldOut{nullptr, nullptr, nullptr});
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(p->ev, ctx->stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); (void)hipEventDestroy(p->ev); dfree(ctx, p->scratch); bmx_vec_free(ctx, v); delete p; return fail_hip(e, "bmx_op2_dev", __LINE__); }
    ctx->pend_used[slot >> 6] |= 1ull << (slot & 63);
    p->v = v;
    *out = p;
    return BMX_OK;
ABI_END }

int bmx_pending_wait(bmx_ctx* ctx, bmx_pending* p, bmx_vec** out)
{ ABI_TRY
    ARGCHK(ctx && p && out && p->ctx == ctx && p->v);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipEventSynchronize(p->ev));
    bmx_vec* v = p->v;
    if (p->resolved) {
        ctx->pend_used[p->slot >> 6] &= ~(1ull << (p->slot & 63));
        (void)hipEventDestroy(p->ev);
        p->v = nullptr;
        delete p;
        *out = v;
        return BMX_OK;
    }
    const u64* hs = ctx->h_pend + (size_t)p->slot * 8;
    const uint32_t nblocks = v->nblocks;
    for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)hs[k];
    const uint64_t used = hs[4] & 0xFFFFFFFFFFull, ncand = hs[4] >> 40, bound = p->gap_bound;
    const bool ok = (uint64_t)v->counts[0] + v->counts[1] + v->counts[2] + v->counts[3] == nblocks && ncand == v->counts[BMX_GAP] &&
                    used <= bound && (used == 0) == (ncand == 0);
    ctx->pend_used[p->slot >> 6] &= ~(1ull << (p->slot & 63));
    (void)hipEventDestroy(p->ev);
    dfree(ctx, p->scratch);                                           // (the kernel that wrote it ran before the event)
    p->v = nullptr;
    delete p;
    if (!ok) { (void)hipStreamSynchronize(ctx->stream); bmx_vec_free(ctx, v); g_last_error = "bmx_pending_wait: inconsistent fold of the result block kinds"; return BMX_ERR_DEVICE; }
    if (bound) {
        v->bytes -= std::min<size_t>(v->bytes, (size_t)bound * 2 + 64);
        if ((rc = gap_slab_trim(ctx, v, bound, used, 0))) { bmx_vec_free(ctx, v); return rc; }
    }
    // the slab, as result_finish treats it: nothing alive -> back to the pool; sparse -> the survivors into a right-sized slab
    // (ordinals from the descriptor table; enqueued, not waited for); nearly full -> kept, ordinals at the first download
    const uint32_t live = v->counts[BMX_BIT];
    if (live == 0) { dfree(ctx, v->d_bits); v->d_bits = nullptr; v->n_bit = 0; }
    else if ((uint64_t)live * 8u < (uint64_t)nblocks * 7u) {
        uint4* packed = nullptr; u32* ord = nullptr;
        if ((rc = dmalloc(ctx, (void**)&packed, (size_t)live * 8192)) || (rc = dmalloc(ctx, (void**)&ord, (size_t)nblocks * 4))) {
            dfree(ctx, packed); bmx_vec_free(ctx, v); return rc;
        }
        hipLaunchKernelGGL(k_ord_from_desc, dim3(1), dim3(1024), 0, ctx->stream, (const u64*)v->d_desc, nblocks, ord);
        hipLaunchKernelGGL(k_compact_bits, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint4*)v->d_bits, nblocks, (const BlockStat*)nullptr, (const u32*)ord, packed, v->d_desc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); dfree(ctx, packed); dfree(ctx, ord); bmx_vec_free(ctx, v); return fail_hip(e, "bmx_pending_wait (compaction)", __LINE__); }
        dfree(ctx, ord); dfree(ctx, v->d_bits);                          // (stream-ordered, see dfree)
        v->d_bits = packed; v->n_bit = live;
    } else if (live < nblocks) v->ord_lazy = true;
    *out = v;
    return BMX_OK;
ABI_END }

int bmx_pending_free(bmx_ctx* ctx, bmx_pending* p)
{ ABI_TRY
    if (!p) return BMX_OK;
    ARGCHK(ctx && p->ctx == ctx);
    (void)hipEventSynchronize(p->ev);
    (void)hipEventDestroy(p->ev);
    ctx->pend_used[p->slot >> 6] &= ~(1ull << (p->slot & 63));
    dfree(ctx, p->scratch);
    int rc = p->v ? bmx_vec_free(ctx, p->v) : BMX_OK;
    delete p;
    return rc;
ABI_END }

static int count_op2_launch(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, u64* out, bool out_is_host)
{
    ARGCHK(ctx && a && b && out && a->ctx == ctx && b->ctx == ctx && op >= BMX_AND && op <= BMX_SUB);
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t nblocks = std::max(a->nblocks, b->nblocks);
    if (!nblocks) {
        if (out_is_host) *out = 0; else HIPCHK(hipMemsetAsync(out, 0, 8, ctx->stream));
        return BMX_OK;
    }
    // bit-blocks only on both sides, same length: the streaming form (a wave per stretch of columns, one workgroup per CU)
    if (ctx->pair_stream != 0 && a->nblocks == b->nblocks && a->counts[BMX_BIT] == nblocks && b->counts[BMX_BIT] == nblocks &&
        nblocks >= 2048u) {
        const u32 waves = ctx->pair_stream > 0 ? (u32)ctx->pair_stream : 4u;      // per workgroup
        const u32 total = 256u * waves * (u32)std::max(ctx->pair_wgs, 1);        // waves of the launch
        u32 per_wave = (nblocks + total - 1u) / total;
        u32 grid = ((nblocks + per_wave - 1u) / per_wave + waves - 1u) / waves;
        FoldOut fo{ctx->d_slots, ctx->d_done, out};
        if (waves == 2u) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_op2_stream<2, true>), dim3(grid), dim3(128), 0, ctx->stream, op, a->d_desc, b->d_desc, nblocks, per_wave, fo);
        else if (waves == 8u) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_op2_stream<8, true>), dim3(grid), dim3(512), 0, ctx->stream, op, a->d_desc, b->d_desc, nblocks, per_wave, fo);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_op2_stream<4, true>), dim3(grid), dim3(256), 0, ctx->stream, op, a->d_desc, b->d_desc, nblocks, per_wave, fo);
        KCHK();
        return BMX_OK;
    }
    // (round 3: a streaming form for operands of ANY block kinds was built and measured -- a wave owning a stretch of
    // columns, sorted by load shape so that every pipelined loop issues a uniform number of loads, GAP blocks prefetched
    // into registers and decoded from there: 52-58 us against the 45.9 us of this kernel on the 1 % mixed case at one to
    // eight workgroups per CU (profiles/r03g, r03h); each of the nine shape loops fills and drains its own pipeline over
    // ~4 columns.  Dropped.)
    // mixed kinds, long vectors: the persistent form (one memory round trip per column, GAP blocks decoded from registers)
    if (ctx->pair_loop != 0 && nblocks >= 2048u) {
        const u32 wgs = (u32)(ctx->pair_loop > 0 ? ctx->pair_loop : 4);     // workgroups per CU = waves per SIMD
        const u32 grid = std::min<u32>((nblocks + 3u) / 4u, 256u * wgs);
        auto fn = ctx->pair_nt ? k_count_op2_loop<4, true> : k_count_op2_loop<4, false>;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, ctx->stream, op,
                           a->d_desc, a->nblocks, b->d_desc, b->nblocks, nblocks, FoldOut{ctx->d_slots, ctx->d_done, out});
        KCHK();
        return BMX_OK;
    }
    hipLaunchKernelGGL(k_count_op2, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, op,
                       a->d_desc, a->nblocks, b->d_desc, b->nblocks, nblocks, FoldOut{ctx->d_slots, ctx->d_done, out});
    KCHK();
    return BMX_OK;
}

int bmx_count_op2_dev(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* d_count)
{ ABI_TRY
    return count_op2_launch(ctx, op, a, b, (u64*)d_count, false);
ABI_END }

} // extern "C"

int bmx_i_count_op2_async(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int slot)
{
    ARGCHK(ctx && slot >= 0 && slot < 64);
    return count_op2_launch(ctx, op, a, b, ctx->h_small + slot, true);
}

extern "C" {

int bmx_count_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* count)
{ ABI_TRY
    ARGCHK(ctx && count);
    int rc = bmx_i_count_op2_async(ctx, op, a, b, 0);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count = ctx->h_small[0];
    return BMX_OK;
ABI_END }

// ---- small collections: aggregation in one launch straight from the descriptor tables (k_direct, bmx_kernels2.h) ----
// waves per column of the one-launch path, 0 = take the row-table pipeline instead.  Long operand lists (24..1024) over few
// columns: 8 waves share the list.  Short lists (< 24 operands): one wave per column -- what the direct kernel saves
// there is the pipeline object and its sort launch (~0.04 ms of host calls per aggregation).
static int use_direct(const bmx_ctx* ctx, uint32_t ncols, size_t n_ops)
{
    if (ctx->pipe_split == 0 || ctx->direct_cols <= 0 || !ncols || !n_ops) return 0;
    // (measured, 1e9-bit operands: 2 operands 0.112 against 0.149 ms, 4: 0.163 / 0.170, 16: 0.449 / 0.421 -- beyond
    // ~0.5 GB of operand blocks the pipelined row kernel streams faster than the saved host calls are worth)
    if (n_ops < 24u) return (uint64_t)ncols * n_ops <= 65536u ? 1 : 0;
    return (ncols <= (uint32_t)ctx->direct_cols && n_ops <= DIRECT_MAX_OPS) ? SPLIT_WAVES : 0;
}

// operand table on the device: n descriptor-table pointers, then n block counts (u32); one staged copy
static int direct_table(bmx_ctx* ctx, const bmx_vec* const* a, size_t na, const bmx_vec* const* b, size_t nb, void** d_tab)
{
    size_t n = na + nb;
    std::vector<u64> tab(n + (n + 1) / 2);
    u32* nb32 = reinterpret_cast<u32*>(tab.data() + n);
    for (size_t i = 0; i < n; ++i) {
        const bmx_vec* o = i < na ? a[i] : b[i - na];
        if (!o || o->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
        tab[i] = (u64)(uintptr_t)o->d_desc; nb32[i] = o->nblocks;
    }
    int rc;
    *d_tab = nullptr;
    if ((rc = dmalloc(ctx, d_tab, tab.size() * 8)) || (rc = h2d_staged(ctx, *d_tab, tab.data(), tab.size() * 8))) { dfree(ctx, *d_tab); *d_tab = nullptr; }
    return rc;
}

static int direct_launch(int mode, bmx_ctx* ctx, const void* d_tab, size_t n_and, size_t n_sub, u32 col_from, u32 col_to, int opt_compress,
                         bmx_vec* v, BlockStat* st, int has_mask, u32 mf, u32 mt)
{
    if (col_to <= col_from) return BMX_OK;
    size_t n = n_and + n_sub;
    const int split = n < 24u ? 1 : SPLIT_WAVES;
    size_t lds = (size_t)split * 8192 + 2 * n * 8;                          // partials + bit / GAP lists of both groups
    auto fn = split == 1 ? (mode == DIRECT_AND_SUB ? k_direct<1, DIRECT_AND_SUB> : mode == DIRECT_OR ? k_direct<1, DIRECT_OR> : k_direct<1, DIRECT_FIND_FIRST>)
                         : (mode == DIRECT_AND_SUB ? k_direct<SPLIT_WAVES, DIRECT_AND_SUB> : mode == DIRECT_OR ? k_direct<SPLIT_WAVES, DIRECT_OR> : k_direct<SPLIT_WAVES, DIRECT_FIND_FIRST>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fn, dim3(col_to - col_from), dim3(split * 64), lds, ctx->stream,
                           (const u64* const*)d_tab, (const u32*)((const u64*)d_tab + n), (u32)n_and, (u32)n_sub, col_from, col_to,
                           opt_compress, v ? v->d_bits : nullptr, v ? v->d_desc : nullptr, st, has_mask, mf, mt, ctx->d_small);
        e = hipGetLastError();
    }
    return e == hipSuccess ? BMX_OK : fail_hip(e, "k_direct", __LINE__);
}

static int agg_or_impl(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result);
static bool coll_members_wanted_list(const bmx_ctx* ctx, const bmx_vec* const* src, size_t n)
{
    uint64_t gw = 0, gb = 0;
    for (size_t i = 0; i < n; ++i) { gw += src[i]->gap_words; gb += src[i]->counts[BMX_GAP]; }
    return coll_members_wanted(ctx, gw, gb, n, 1);
}

// combine_or over >= 64 GAP-only operands: the row kernel (bmx_kernels7.h) when the operands are sparse enough for a tile of
// ORR_TILE = 14 blocks to fit one 1-KiB row (<= 64 chunks of 16 B) nearly always, i.e. <= 4.1 chunks per GAP block on average
// (tuning build only: BMX_DIAG_ROWS = 512 -> the row loads alone; results are then meaningless)
static int or_rows_diag_bits()
{
#ifdef BMX_DIAG
    if (const char* e = getenv("BMX_DIAG_ROWS")) return atoi(e) & (512 | 1024 | 2048);
#endif
    return 0;
}
static bool or_rows_wanted(const bmx_ctx* ctx, const bmx_vec* const* src, size_t n)
{
    if (ctx->or_rows == 0) return false;
    uint64_t words = 0, blocks = 0;
    for (size_t i = 0; i < n; ++i) {
        words += src[i]->gap_words; blocks += src[i]->counts[BMX_GAP];
    }
    return ctx->or_rows == 1 || (blocks && words * 10ull <= blocks * 328ull);     // <= 4.1 chunks of 8 words per GAP block
}

int bmx_agg_or(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, bmx_vec** result)
{ ABI_TRY
    return agg_or_impl(ctx, src, n, 0 /* opt_mode_ = opt_none, src/bmaggregator.h:917 */, result);
ABI_END }

int bmx_agg_or_opt(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result)
{ ABI_TRY
    return agg_or_impl(ctx, src, n, opt_compress ? 1 : 0, result);
ABI_END }

static int agg_or_impl(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result)
{
    ARGCHK(ctx && result && (n == 0 || src) && n <= 65535);
    *result = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t ncols = 0; uint64_t nbits = 0; bool has_gap = false, has_bit = false;
    std::vector<const u64*> descs(std::max<size_t>(n, 1), nullptr);
    std::vector<u32> nblk(std::max<size_t>(n, 1), 0);
    for (size_t i = 0; i < n; ++i) {
        if (!src[i] || src[i]->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
        descs[i] = src[i]->d_desc; nblk[i] = src[i]->nblocks;
        ncols = std::max(ncols, src[i]->nblocks); nbits = std::max(nbits, src[i]->nbits);
        has_gap |= src[i]->counts[BMX_GAP] != 0;
        has_bit |= src[i]->counts[BMX_BIT] != 0;
    }
    bmx_vec* v; BlockStat* st; u32* offs;
    bmx_coll* packed = nullptr; std::vector<u32> members; bool packed_full = false;
    CollPin pin;
    if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) return rc;      // empty list => cleared target (:1105)
    if (use_direct(ctx, ncols, n)) {
        void* d_tab = nullptr;
        if ((rc = direct_table(ctx, src, n, nullptr, 0, &d_tab))) { bmx_vec_free(ctx, v); return rc; }
        rc = direct_launch(DIRECT_OR, ctx, d_tab, n, 0, 0u, ncols, opt_compress, v, st, 0, 0u, 0u);
        if (!rc) rc = result_finish(ctx, v, st, offs);
        else (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_tab);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
    } else if (n >= 16 && ncols && has_gap && !has_bit && (rc = coll_resolve(ctx, src, n, 1, 64, true, &packed, &members, &packed_full)) == BMX_OK && packed && !packed_full &&
               !(ctx->coll_members < 0 && n >= 64 && or_rows_wanted(ctx, src, n)) && coll_members_wanted_list(ctx, src, n)) {
        pin.pin(packed);                                       // (coll_members_upload allocates: the collection must outlive it)
        // GAP-only operands that are SOME of the vectors of a packed collection: their pieces of its column regions
        // (k_coll_members, bmx_kernels8.h).  Sparse lists of >= 64 vectors take the row kernel below instead: a member's piece
        // of a column is ~26 bytes there, and reading them one by one (9.1 ms for 2,048 of configs[4]'s 4,096 vectors) loses
        // to the rows of 14 columns the vectors' own tile directories give (1.7 ms, profiles/r04i)
        void* d_buf = nullptr; const u32* d_midx = nullptr; const CollGroup* d_groups = nullptr;
        if ((rc = coll_members_upload(ctx, members, std::vector<u32>(), &d_buf, &d_midx, &d_groups))) { bmx_vec_free(ctx, v); return rc; }
        rc = coll_members_launch(CM_OR_STORE, ctx, packed, nullptr, d_midx, d_groups, 1u, 0u, ncols, opt_compress, nullptr, v, st);
        if (!rc) rc = result_finish(ctx, v, st, offs);
        else (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_buf);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
    } else if (rc) { bmx_vec_free(ctx, v); return rc;
    } else if (packed && packed_full) {
        pin.pin(packed);
        // GAP-only operands = ALL the vectors of a packed collection: one sequential stream per block column (bmx_kernels6.h)
        // without opt_compress no GAP block can come out: the kernel folds the kind counts of its result itself and, when every
        // block turned out to be a bit-block (the OR of thousands of sparse vectors), the layout scan is skipped (one launch
        // window only: the fold's tickets count the workgroups of ONE launch)
        const bool fold = !opt_compress && ctx->coll_window == 0;
        rc = coll_launch(COLL_OR, ctx, packed, nullptr, 0u, ncols, opt_compress, nullptr, v, st, 0u, 0xFFFFFFFFu,
                         fold ? FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + 2} : FoldOut{nullptr, nullptr, nullptr});
        bool done = false;
        if (!rc && fold) {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail_hip(e, "k_coll_apply", __LINE__);
            else if (ctx->h_small[2 + BMX_GAP] == 0 && ctx->h_small[2 + BMX_BIT] == ncols) {
                for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
                done = true;
            }
        }
        if (!rc && !done) rc = result_finish(ctx, v, st, offs);
        else if (rc) (void)hipStreamSynchronize(ctx->stream);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
    } else if (rc) { bmx_vec_free(ctx, v); return rc;
    } else if (n >= 64 && ncols && has_gap && !has_bit && or_rows_wanted(ctx, src, n)) {
        // many SPARSE GAP-only operands: rows of 16 block columns read through the vectors' tile directories (k_agg_or_rows,
        // bmx_kernels7.h); the kernel folds the block kinds and the popcount of its result (no layout scan when every block
        // came out as a bit-block, no count pass later)
        std::vector<u64> tab; tab.reserve(n * 4);
        for (size_t i = 0; i < n; ++i) {
            const bmx_vec* o = src[i];
            // uploaded / imported / generated vectors carry their tile directory; a RESULT vector (or a clone) gets its own the
            // first time it is an operand here (the directory is a cache of the immutable vector's layout: logically const)
            if (!o->d_tdir && (rc = vec_build_tdir(ctx, const_cast<bmx_vec*>(o)))) { (void)hipStreamSynchronize(ctx->stream); bmx_vec_free(ctx, v); return rc; }
            if (!o->d_tdir) continue;                                          // NULL blocks only: contributes nothing
            tab.push_back((u64)(uintptr_t)o->d_tdir); tab.push_back((u64)(uintptr_t)o->d_gaps);
            tab.push_back((u64)(uintptr_t)o->d_desc); tab.push_back((u64)o->nblocks);
        }
        const u32 nops = (u32)(tab.size() / 4);
        void* d_tab = nullptr;
        if ((rc = dmalloc(ctx, &d_tab, std::max<size_t>(tab.size() * 8, 64))) || (rc = h2d_staged(ctx, d_tab, tab.data(), tab.size() * 8))) {
            dfree(ctx, d_tab); bmx_vec_free(ctx, v); return rc;
        }
        const size_t lds = (size_t)ORR_TILE * 8192 + 64;
        const u32 ntiles = (ncols + ORR_TILE - 1) / ORR_TILE;
        auto rows = ctx->or_depth == 8 ? k_agg_or_rows<8> : k_agg_or_rows<4>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rows, dim3(ntiles), dim3(1024), lds, ctx->stream, (const u32x4*)d_tab, nops, ncols, opt_compress | or_rows_diag_bits(), ctx->xcd_swz,
                               v->d_bits, v->d_desc, st, FoldOut{ctx->d_slots, ctx->d_done, ctx->h_small + 2},
                               FoldOut{ctx->d_slots2, ctx->d_done2, ctx->h_small + 8});
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_tab);
        if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "bmx_agg_or (rows)", __LINE__); }
        const uint64_t total = ctx->h_small[8];
        if (ctx->h_small[2 + BMX_GAP] == 0 && ctx->h_small[2 + BMX_BIT] == ncols) {
            for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
        } else if ((rc = result_finish(ctx, v, st, offs))) { bmx_vec_free(ctx, v); return rc; }
        v->count = total; v->count_valid = true;
    } else if (n >= 64 && ncols && has_gap && !has_bit) {
        // many GAP-only operands: column-tile kernel straight from the descriptor tables (no sort pass)
        void* d_descs = nullptr; void* d_nblk = nullptr;
        if ((rc = dmalloc(ctx, &d_descs, n * 8)) || (rc = dmalloc(ctx, &d_nblk, n * 4))) { dfree(ctx, d_descs); bmx_vec_free(ctx, v); return rc; }
        hipError_t e = hipMemcpyAsync(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        size_t lds = (size_t)OR_TILE * 8192 + OR_TILE * 4;
        // or_tile: 0 = single-bit fast path (default), 1 = the round-1 run code, 2 = two operands per lane per step;
        // tuning build, or_window = -9: loads only (the memory floor of the access pattern)
        u32 ntiles = (ncols + OR_TILE - 1) / OR_TILE;
#ifdef BMX_TUNE
        auto tiled = ctx->or_window == -9 ? (ctx->or_tile == 2 ? k_agg_or_gap_tiled<9, 2> : k_agg_or_gap_tiled<9, 1>) :
                     ctx->or_tile == 0 ? k_agg_or_gap_tiled<1, 1> : ctx->or_tile == 2 ? k_agg_or_gap_tiled<1, 2> : k_agg_or_gap_tiled<0, 1>;
#else
        auto tiled = ctx->or_tile == 0 ? k_agg_or_gap_tiled<1, 1> : ctx->or_tile == 2 ? k_agg_or_gap_tiled<1, 2> : k_agg_or_gap_tiled<0, 1>;
#endif
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiled), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            u32 win = ctx->or_window <= 0 ? ntiles : (u32)ctx->or_window;        // windows measured: no gain here
            u32 nwin = (ntiles + win - 1) / win, per = (ntiles + nwin - 1) / nwin;
            for (u32 t0 = 0; t0 < ntiles && e == hipSuccess; t0 += per) {
                hipLaunchKernelGGL(tiled, dim3(std::min(per, ntiles - t0)), dim3(1024), lds, ctx->stream,
                                   (const u64* const*)d_descs, (const u32*)d_nblk, (u32)n, ncols, opt_compress, v->d_bits, v->d_desc, st, t0);
                e = hipGetLastError();
            }
        }
        if (e == hipSuccess) rc = result_finish(ctx, v, st, offs);
        else rc = fail_hip(e, "bmx_agg_or (tiled)", __LINE__);
        (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_descs); dfree(ctx, d_nblk);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
    } else if (n && ncols) {
        void* d_descs = nullptr; void* d_nblk = nullptr; void* d_dmat = nullptr;
        size_t b_dmat = (size_t)ncols * (n + 2) * 8;
        if ((rc = dmalloc(ctx, &d_descs, n * 8)) || (rc = dmalloc(ctx, &d_nblk, n * 4)) || (rc = dmalloc(ctx, &d_dmat, b_dmat))) {
            dfree(ctx, d_descs); dfree(ctx, d_nblk); bmx_vec_free(ctx, v); return rc;
        }
        hipError_t e = hipMemcpyAsync(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_or_sort, dim3((ncols + 3) / 4), dim3(256), 0, ctx->stream,
                               (const u64* const*)d_descs, (const u32*)d_nblk, (u32)n, ncols, (u64*)d_dmat);
            size_t lds = has_gap ? 4 * 2048 * 4 : 0;
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_or<2>), dim3((ncols + 3) / 4), dim3(256), lds, ctx->stream,
                               (const u64*)d_dmat, (u32)n, ncols, opt_compress, ctx->xcd_swz,
                               v->d_bits, v->d_desc, st);
            e = hipGetLastError();
        }
        if (e == hipSuccess) rc = result_finish(ctx, v, st, offs);
        else rc = fail_hip(e, "bmx_agg_or", __LINE__);
        (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_descs); dfree(ctx, d_nblk); dfree(ctx, d_dmat);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
    } else if (ncols) {
        HIPCHK(hipMemsetAsync(v->d_desc, 0, (size_t)ncols * 8, ctx->stream));
        v->counts[BMX_NULL] = ncols;
    }
    *result = v;
    return BMX_OK;
}

} // extern "C"

// one (column-per-wave) materialising aggregation of group g of a pipeline.  Bit-block operands only and enough columns:
// the launch plan of the headline kernel (windows of one 640-thread workgroup per CU, 4 blocks in flight per wave);
// otherwise one launch of 256-thread workgroups (GAP operands park the accumulator in LDS: 8 KiB per wave).
static int agg_and_sub_launch(bmx_ctx* ctx, const bmx_pipeline* p, uint32_t g, bmx_vec* v, BlockStat* st, uint32_t nb_from, uint32_t nb_to)
{
    const u64* rows = p->d_dmat + (*p->h_row_off)[g];
    const u32* an = p->d_meta + p->ngroups + g;
    const u32* sn = p->d_meta + 2 * p->ngroups + g;
    const u32 ncols = p->ncols;
    hipError_t e = hipSuccess;
    if (ncols && use_and_rows(ctx, p, (uint64_t)(*p->h_and_n)[g] + (*p->h_sub_n)[g])) {
        // GAP-only operands: the union of 0-runs over the operands' own slabs, result block stored (bmx_kernels9.h)
        hipLaunchKernelGGL(and_rows_kernel<AR_STORE>(ctx), dim3(ncols), dim3((u32)ctx->and_rows_wg), 0, ctx->stream,
                           rows, p->d_meta /* row_off[0] = 0: rows already points at group g */, an, sn, p->col_stride, 1u, 0u, ncols, ctx->xcd_swz, (u64*)nullptr,
                           v->d_bits, v->d_desc, st, nb_from, nb_to, 0, 1u);
        e = hipGetLastError();
        return e == hipSuccess ? BMX_OK : fail_hip(e, "k_agg_and_rows", __LINE__);
    }
    if (ncols && use_gapcount(ctx, p, (*p->h_and_n)[g])) {
        // GAP-only operands, a long list: the counting formulation, one 1024-thread workgroup per column, result block stored
        size_t lds = (size_t)(16384 * 2 + 2048) * 4;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_counts_gapcount<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_gapcount<true>), dim3(ncols), dim3(1024), lds, ctx->stream,
                               rows, p->d_meta /* row_off[0] is added below: rows already points at group g */, an, sn, p->col_stride, 1u, 0u, ncols, (u64*)nullptr,
                               v->d_bits, v->d_desc, st, nb_from, nb_to);
            e = hipGetLastError();
        }
        return e == hipSuccess ? BMX_OK : fail_hip(e, "k_pipe_counts_gapcount", __LINE__);
    }
    if (!p->has_gap && ncols >= 2560u && ctx->pipe_window >= 0 && (ctx->pipe_wg == 0 || ctx->pipe_wg == 640)) {
        // (640 threads = 10 waves per CU; 512 measured: 4.98 against 4.86 ms on the 256 x 1e9-bit combine_and.  Four operand
        // blocks in flight per wave need ~180 VGPRs, more than the 168 a 640-thread workgroup leaves a wave: <4, 640> spilled
        // 80 B per lane to scratch.  agg_shape 0 = three blocks in flight at 640 threads (143 VGPRs), 1 = four at
        // 512 threads (183 VGPRs, two waves per SIMD); neither touches scratch -- tests/test_abi_host.py checks the whole library)
        const bool s512 = ctx->agg_shape == 1;
        const u32 wpb = s512 ? 8u : 10u, wgs = wpb * 64u;
        const u32 cap = ctx->pipe_window > 0 ? (u32)ctx->pipe_window : pipe_window_cap(wgs);
        u32 nwin = (ncols + cap - 1u) / cap, per = (ncols + nwin - 1u) / nwin;
        per = (per + wpb - 1u) / wpb * wpb;                               // whole workgroups
        for (u32 c0 = 0; c0 < ncols && e == hipSuccess; c0 += per) {
            u32 n = std::min(per, ncols - c0);
            if (s512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_and_sub<4, 512>), dim3((n + wpb - 1u) / wpb), dim3(512), 0, ctx->stream,
                               rows, an, sn, p->col_stride, std::min(ncols, c0 + n), 1 /* opt_compress, :1210,1421 */, 0,
                               v->d_bits, v->d_desc, st, nb_from, nb_to, c0);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_and_sub<3, 640>), dim3((n + wpb - 1u) / wpb), dim3(640), 0, ctx->stream,
                               rows, an, sn, p->col_stride, std::min(ncols, c0 + n), 1 /* opt_compress, :1210,1421 */, 0,
                               v->d_bits, v->d_desc, st, nb_from, nb_to, c0);
            e = hipGetLastError();
        }
    } else {
        size_t lds = p->has_gap ? 4 * 2048 * 4 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_and_sub<2, 256>), dim3((ncols + 3) / 4), dim3(256), lds, ctx->stream,
                           rows, an, sn, p->col_stride, ncols, 1, ctx->xcd_swz, v->d_bits, v->d_desc, st, nb_from, nb_to, 0u);
        e = hipGetLastError();
    }
    return e == hipSuccess ? BMX_OK : fail_hip(e, "k_agg_and_sub", __LINE__);
}

extern "C" {

int bmx_agg_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                    const bmx_vec* const* src_sub, size_t n_sub, bmx_vec** result, int* any)
{ ABI_TRY
    ARGCHK(ctx && result && (n_and == 0 || src_and) && (n_sub == 0 || src_sub));
    *result = nullptr;
    if (any) *any = 0;
    int rc = set_dev(ctx); if (rc) return rc;
    uint64_t nbits = 0; uint32_t ncols = 0;
    for (size_t i = 0; i < n_and; ++i) { ARGCHK(src_and[i]); nbits = std::max(nbits, src_and[i]->nbits); ncols = std::max(ncols, src_and[i]->nblocks); }
    for (size_t i = 0; i < n_sub; ++i) { ARGCHK(src_sub[i]); nbits = std::max(nbits, src_sub[i]->nbits); ncols = std::max(ncols, src_sub[i]->nblocks); }
    bmx_vec* v; BlockStat* st; u32* offs;
    if (!n_and) {                                               // empty AND group => cleared target (:1170-1174)
        if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) return rc;
        if (ncols) HIPCHK(hipMemsetAsync(v->d_desc, 0, (size_t)ncols * 8, ctx->stream));
        v->counts[BMX_NULL] = ncols;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *result = v;
        return BMX_OK;
    }
    // small collection (few columns, many operands): one launch straight from the descriptor tables (k_direct)
    if (use_direct(ctx, ncols, n_and + n_sub)) {
        void* d_tab = nullptr;
        if ((rc = direct_table(ctx, src_and, n_and, src_sub, n_sub, &d_tab))) return rc;
        if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) { dfree(ctx, d_tab); return rc; }
        rc = direct_launch(DIRECT_AND_SUB, ctx, d_tab, n_and, n_sub, 0u, ncols, 1, v, st, 0, 0u, 0u);
        if (!rc) rc = result_finish(ctx, v, st, offs);              // synchronises
        else (void)hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_tab);
        if (rc) { bmx_vec_free(ctx, v); return rc; }
        if (any) *any = (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP]) != 0;
        *result = v;
        return BMX_OK;
    }
    {
        // GAP-only operands held by packed collections: the whole column regions (the lists name every vector of their
        // collections, bmx_kernels6.h) or the lists' pieces of them (k_coll_members, bmx_kernels8.h)
        bmx_coll *ca = nullptr, *cs = nullptr;
        std::vector<u32> ma, ms; bool full = false;
        bool ok = true;
        for (size_t i = 0; i < n_and && ok; ++i) ok = src_and[i]->ctx == ctx;
        for (size_t i = 0; i < n_sub && ok; ++i) ok = src_sub[i]->ctx == ctx;
        if (ok && ncols && (rc = coll_resolve_and_sub(ctx, src_and, n_and, src_sub, n_sub, &ca, &cs, &ma, &ms, &full))) return rc;
        if (ca && !full) {
            uint64_t gw = 0, gb = 0;
            for (size_t i = 0; i < n_and; ++i) { gw += src_and[i]->gap_words; gb += src_and[i]->counts[BMX_GAP]; }
            for (size_t i = 0; i < n_sub; ++i) { gw += src_sub[i]->gap_words; gb += src_sub[i]->counts[BMX_GAP]; }
            if (!coll_members_wanted(ctx, gw, gb, n_and + n_sub, 1)) ca = nullptr;
        }
        if (ca) {
            CollPin pin; pin.pin(ca, cs);                      // (result_begin / coll_members_upload allocate: no eviction from under the call)
            if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) return rc;
            void* d_buf = nullptr;
            if (full) rc = coll_launch(COLL_AND_STORE, ctx, ca, cs, 0u, ncols, 1, nullptr, v, st, 0u, 0xFFFFFFFFu);
            else {
                const u32* d_midx = nullptr; const CollGroup* d_groups = nullptr;
                rc = coll_members_upload(ctx, ma, ms, &d_buf, &d_midx, &d_groups);
                if (!rc) rc = coll_members_launch(CM_AND_STORE, ctx, ca, cs, d_midx, d_groups, 1u, 0u, ncols, 1, nullptr, v, st);
            }
            if (!rc) rc = result_finish(ctx, v, st, offs);
            else (void)hipStreamSynchronize(ctx->stream);
            dfree(ctx, d_buf);
            if (rc) { bmx_vec_free(ctx, v); return rc; }
            if (any) *any = (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP]) != 0;
            *result = v;
            return BMX_OK;
        }
    }
    uint32_t an = (uint32_t)n_and, sn = (uint32_t)n_sub;
    bmx_pipeline* p = nullptr;
    if ((rc = bmx_pipeline_create(ctx, src_and, &an, src_sub, &sn, 1, &p))) return rc;
    if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) { bmx_pipeline_destroy(ctx, p); return rc; }
    if (ncols && use_split(ctx, p, ncols)) {
        size_t lds = (size_t)SPLIT_WAVES * 8192;
        auto fn = k_pipe_split<2, SPLIT_WAVES>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(fn, dim3(ncols), dim3(SPLIT_WAVES * 64), lds, ctx->stream,
                               (const u64*)p->d_dmat, (const u32*)p->d_meta, (const u32*)(p->d_meta + 1), (const u32*)(p->d_meta + 2),
                               p->col_stride, 1u, 0u, ncols, 1, (u64*)nullptr, 1, v->d_bits, v->d_desc, st);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = fail_hip(e, "k_pipe_split", __LINE__);
    } else if (ncols) rc = agg_and_sub_launch(ctx, p, 0u, v, st, 0u, 0xFFFFFFFFu);
    if (!rc) rc = result_finish(ctx, v, st, offs);
    bmx_pipeline_destroy(ctx, p);
    if (rc) { bmx_vec_free(ctx, v); return rc; }
    if (any) *any = (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP]) != 0;
    *result = v;
    return BMX_OK;
ABI_END }

// combine_and_sub(pipe) with result vectors / counts / OR target (src/bmaggregator.h:1292-1449)
static int run_results_impl(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, bmx_vec** results_out, uint64_t* counts_out,
                            const bmx_vec* or_target_in, bmx_vec** or_target_out);

int bmx_pipeline_run_results(bmx_ctx* ctx, bmx_pipeline* p, bmx_vec** results_out, uint64_t* counts_out,
                             const bmx_vec* or_target_in, bmx_vec** or_target_out)
{ ABI_TRY
    return run_results_impl(ctx, p, 0u, 0xFFFFFFFFu, results_out, counts_out, or_target_in, or_target_out);
ABI_END }

int bmx_pipeline_run_results_range(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, bmx_vec** results_out,
                                   uint64_t* counts_out, const bmx_vec* or_target_in, bmx_vec** or_target_out)
{ ABI_TRY
    return run_results_impl(ctx, p, nb_from, nb_to, results_out, counts_out, or_target_in, or_target_out);
ABI_END }

static int run_results_impl(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, bmx_vec** results_out, uint64_t* counts_out,
                            const bmx_vec* or_target_in, bmx_vec** or_target_out)
{
    ARGCHK(ctx && p && p->ctx == ctx && (results_out || or_target_out));
    { int rcr = pipe_range(p, nb_from, nb_to); if (rcr) return rcr; }
    ARGCHK(!or_target_in || or_target_in->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    std::vector<bmx_vec*> res(p->ngroups, nullptr);
    auto cleanup = [&]() { for (bmx_vec* r : res) if (r) bmx_vec_free(ctx, r); };
    // set_search_count_limit applies whenever counts are computed, result vectors included (the test at src/bmaggregator.h:1362
    // sits in front of both branches): a counts run under the limit finds the column at which every group has enough, and the
    // group's vector is produced up to there -- its count is >= min(limit, true count), bits beyond are not searched
    const bool limited = counts_out && p->search_limit != ~0ull;
    if (limited && (rc = limit_counts_run(ctx, p, nb_from, nb_to, nullptr))) return rc;
    const uint32_t nb_to_all = nb_to;
    for (uint32_t g = 0; g < p->ngroups; ++g) {
        if (counts_out) counts_out[g] = 0;
        if (!(*p->h_and_n)[g] || !p->ncols) continue;                       // empty AND group: skipped (:1352)
        nb_to = limited ? std::min(nb_to_all, (*p->h_stop)[g]) : nb_to_all;
        bmx_vec* v; BlockStat* st; u32* offs;
        if ((rc = result_begin(ctx, p->nbits, p->ncols, &v, &st, &offs))) { cleanup(); return rc; }
        bmx_coll *ca = nullptr, *cs = nullptr;
        if (p->h_uids && nb_from == 0 && nb_to >= p->ncols && (rc = pipe_resolve_colls(ctx, p, false, &ca, &cs))) { bmx_vec_free(ctx, v); cleanup(); return rc; }
        if (ca) rc = coll_members_launch(CM_AND_STORE, ctx, ca, cs, (const u32*)p->cm_buf, (const CollGroup*)((const char*)p->cm_buf + p->cm_groups_off) + g,
                                         1u, 0u, p->ncols, 1, nullptr, v, st);       // (the group's members inside the packed collections)
        else rc = agg_and_sub_launch(ctx, p, g, v, st, nb_from, nb_to);
        if (rc) { bmx_vec_free(ctx, v); cleanup(); return rc; }
        if ((rc = result_finish(ctx, v, st, offs))) { bmx_vec_free(ctx, v); cleanup(); return rc; }
        if (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP] == 0) { bmx_vec_free(ctx, v); continue; }   // nothing found: stays NULL (:1406)
        res[g] = v;
        if (counts_out && (rc = bmx_count(ctx, v, &counts_out[g]))) { cleanup(); return rc; }
    }
    if (or_target_out) {
        std::vector<const bmx_vec*> src;
        if (or_target_in) src.push_back(or_target_in);
        for (bmx_vec* r : res) if (r) src.push_back(r);
        *or_target_out = nullptr;
        if ((rc = agg_or_impl(ctx, src.data(), src.size(), 1 /* optimised at the end, :1440-1447 */, or_target_out))) { cleanup(); return rc; }
    }
    if (results_out) for (uint32_t g = 0; g < p->ngroups; ++g) results_out[g] = res[g];
    else cleanup();
    return BMX_OK;
}

// ---- pipelines with search masks under aggregator::set_range_hint(from, to) ----
// The reference evaluates only the block columns of the hint (src/bmaggregator.h:1312-1346) and, when both ends lie in ONE
// block, ANDs that column with the bit range as well (range_gap_blk_, :980-988, 2354-2358) -- pinned by the reference-generated
// fixtures of tests/golden ("range_hint").  A hint across blocks is the block-range run.  The one-block case is a single
// column: it is evaluated by the ordinary kernels and the result block is AND-ed with a one-run mask vector afterwards.
static int hint_mask_vector(bmx_ctx* ctx, uint64_t nbits, uint32_t ncols, uint64_t from, uint64_t to, bmx_vec** out)
{
    const uint32_t nb = (uint32_t)(from >> 16);
    const uint32_t a = (uint32_t)(from & 65535u), b = (uint32_t)(to & 65535u);
    std::vector<uint8_t> kinds(ncols, BMX_NULL); std::vector<uint32_t> offs(ncols, 0);
    uint16_t gap[4]; uint32_t len = 0;
    // run ends of the block: [0-run to a-1], 1-run to b, [0-run to 65535]
    if (a > 0) gap[++len] = (uint16_t)(a - 1u);
    gap[++len] = (uint16_t)b;
    if (b < 65535u) gap[++len] = 65535u;
    gap[0] = (uint16_t)((len << 3) | (a == 0 ? 1u : 0u));
    kinds[nb] = BMX_GAP;
    return bmx_vec_upload(ctx, nbits, ncols, kinds.data(), offs.data(), nullptr, 0, gap, len + 1u, out);
}

int bmx_pipeline_run_results_hint(bmx_ctx* ctx, bmx_pipeline* p, uint64_t from, uint64_t to, bmx_vec** results_out,
                                  uint64_t* counts_out, const bmx_vec* or_target_in, bmx_vec** or_target_out)
{ ABI_TRY
    ARGCHK(ctx && p && p->ctx == ctx && (results_out || or_target_out || counts_out));
    if (from > to) { g_last_error = "range hint: from > to"; return BMX_ERR_RANGE; }
    const uint64_t nbf = from >> 16, nbt = to >> 16;
    const bool whole = (from & 65535u) == 0 && (to & 65535u) == 65535u;
    if (nbf != nbt || whole || nbf >= p->ncols) {
        uint32_t lo = (uint32_t)std::min<uint64_t>(nbf, 0xFFFFFFFEull), hi = (uint32_t)std::min<uint64_t>(nbt + 1, 0xFFFFFFFFull);
        if (!results_out && !or_target_out) return bmx_pipeline_run_counts(ctx, p, lo, hi, counts_out);
        return run_results_impl(ctx, p, lo, hi, results_out, counts_out, or_target_in, or_target_out);
    }
    // one block, partly covered: evaluate the column, then AND with the bit range
    std::vector<bmx_vec*> res(p->ngroups, nullptr);
    int rc = run_results_impl(ctx, p, (uint32_t)nbf, (uint32_t)nbf + 1u, res.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    auto cleanup = [&]() { for (bmx_vec*& r : res) if (r) { bmx_vec_free(ctx, r); r = nullptr; } };
    bmx_vec* mask = nullptr;
    if ((rc = hint_mask_vector(ctx, p->nbits, p->ncols, from, to, &mask))) { cleanup(); return rc; }
    for (uint32_t g = 0; g < p->ngroups && !rc; ++g) {
        if (counts_out) counts_out[g] = 0;
        if (!res[g]) continue;
        bmx_vec* m = nullptr;
        rc = bmx_op2(ctx, BMX_AND, res[g], mask, 1, &m);
        bmx_vec_free(ctx, res[g]); res[g] = nullptr;
        if (rc) break;
        if (m->counts[BMX_FULL] + m->counts[BMX_BIT] + m->counts[BMX_GAP] == 0) { bmx_vec_free(ctx, m); continue; }
        res[g] = m;
        if (counts_out) rc = bmx_count(ctx, m, &counts_out[g]);
    }
    bmx_vec_free(ctx, mask);
    if (!rc && or_target_out) {
        std::vector<const bmx_vec*> src;
        if (or_target_in) src.push_back(or_target_in);
        for (bmx_vec* r : res) if (r) src.push_back(r);
        *or_target_out = nullptr;
        rc = agg_or_impl(ctx, src.data(), src.size(), 1, or_target_out);
    }
    if (rc) { cleanup(); return rc; }
    if (results_out) for (uint32_t g = 0; g < p->ngroups; ++g) results_out[g] = res[g];
    else cleanup();
    return BMX_OK;
ABI_END }

static int find_first_impl(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and, const bmx_vec* const* src_sub, size_t n_sub,
                           bool ranged, uint64_t from, uint64_t to, int* found, uint64_t* idx)
{
    ARGCHK(ctx && found && idx && (n_and == 0 || src_and) && (n_sub == 0 || src_sub));
    *found = 0; *idx = 0;
    int rc = set_dev(ctx); if (rc) return rc;
    if (!n_and) return BMX_OK;
    if (ranged && from > to) { g_last_error = "range hint: from > to"; return BMX_ERR_RANGE; }
    uint32_t an = (uint32_t)n_and, sn = (uint32_t)n_sub;
    u32 ncols_all = 0;
    for (size_t i = 0; i < n_and; ++i) { ARGCHK(src_and[i]); ncols_all = std::max(ncols_all, src_and[i]->nblocks); }
    for (size_t i = 0; i < n_sub; ++i) { ARGCHK(src_sub[i]); ncols_all = std::max(ncols_all, src_sub[i]->nblocks); }
    u32 col_from = 0, col_to = ncols_all; int has_mask = 0; u32 mf = 0, mt = 65535u;
    if (ranged) {
        uint64_t nbf = from >> 16, nbt = to >> 16;
        col_from = (u32)std::min<uint64_t>(nbf, ncols_all);
        col_to = (u32)std::min<uint64_t>(nbt + 1u, ncols_all);
        if (nbf == nbt) { has_mask = 1; mf = (u32)(from & 65535u); mt = (u32)(to & 65535u); }
    }
    // The reference stops at the first column that holds a hit (src/bmaggregator.h:1470-1512).  Here the columns are
    // visited in ASCENDING WINDOWS that grow fourfold, all enqueued at once on the stream: every workgroup / wave first
    // looks at the best index found so far and leaves when its column lies behind it, so after a hit the remaining
    // windows cost a launch each and nothing else -- no host round trip between windows.  (One big launch cannot do
    // that: thousands of columns are resident before the first one finishes.)
    const u32 ncv = col_to > col_from ? col_to - col_from : 0u;
    // long lists: any number of columns (windows); short lists: the size rule of use_direct
    const bool direct = use_direct(ctx, (n_and + n_sub < 24u) ? ncv : (ncv ? 1u : 0u), n_and + n_sub) != 0;
    auto windows = [&](u32 first, auto&& launch) -> int {
        if (ctx->ff_window < 0) first = ncv;                                 // knob: one launch
        else if (ctx->ff_window > 0) first = (u32)ctx->ff_window;
        u32 c = col_from, w = std::max(first, 1u);
        while (c < col_to) {
            u32 e = (col_to - c <= w + w / 2u) ? col_to : c + w;             // no tiny last window
            int r = launch(c, e); if (r) return r;
            c = e; w *= 4u;
        }
        return BMX_OK;
    };
    if (direct) {
        // many operands: a workgroup of 8 waves per column straight from the descriptor tables (k_direct); 64 columns first
        // (a hit in the first blocks is answered after ~1/8 of what two workgroups per CU would read)
        void* d_tab = nullptr;
        if ((rc = direct_table(ctx, src_and, n_and, src_sub, n_sub, &d_tab))) return rc;
        hipError_t e = hipMemsetAsync(ctx->d_small, 0xFF, 8, ctx->stream);
        if (e == hipSuccess)
            rc = windows(n_and + n_sub < 24u ? 512u : 64u, [&](u32 c0, u32 c1) { return direct_launch(DIRECT_FIND_FIRST, ctx, d_tab, n_and, n_sub, c0, c1, 0, nullptr, nullptr, has_mask, mf, mt); });
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_tab);
        if (rc) return rc;
        if (e != hipSuccess || e2 != hipSuccess) return fail_hip(e != hipSuccess ? e : e2, "bmx_find_first_and_sub", __LINE__);
        if (ctx->h_small[0] != ~0ull) { *found = 1; *idx = ctx->h_small[0]; }
        return BMX_OK;
    }
    bmx_pipeline* p = nullptr;
    if ((rc = bmx_pipeline_create(ctx, src_and, &an, src_sub, &sn, 1, &p))) return rc;
    hipError_t e = hipMemsetAsync(ctx->d_small, 0xFF, 8, ctx->stream);
    if (e == hipSuccess) {
        size_t lds = p->has_gap ? 4 * 2048 * 4 : 0;
        rc = windows(512u, [&](u32 c0, u32 c1) {                             // a wave per column: first window = 2 waves per CU
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_find_first_and_sub<2>), dim3((c1 - c0 + 3) / 4), dim3(256), lds, ctx->stream,
                               p->d_dmat, p->d_meta + 1, p->d_meta + 2, p->col_stride, c0, c1, has_mask, mf, mt, ctx->d_small);
            hipError_t le = hipGetLastError();
            return le == hipSuccess ? BMX_OK : fail_hip(le, "k_find_first_and_sub", __LINE__);
        });
    }
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    bmx_pipeline_destroy(ctx, p);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail_hip(e != hipSuccess ? e : e2, "bmx_find_first_and_sub", __LINE__);
    if (ctx->h_small[0] != ~0ull) { *found = 1; *idx = ctx->h_small[0]; }
    return BMX_OK;
}

int bmx_agg_and_sub_indices(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                            const bmx_vec* const* src_sub, size_t n_sub, int width, void* out, uint64_t cap, uint64_t* n)
{ ABI_TRY
    ARGCHK(n);
    *n = 0;
    bmx_vec* t = nullptr; int any = 0;
    int rc = bmx_agg_and_sub(ctx, src_and, n_and, src_sub, n_sub, &t, &any);
    if (rc) return rc;
    if (any) rc = bmx_vec_to_indices(ctx, t, width, out, cap, n);
    bmx_vec_free(ctx, t);
    return rc;
ABI_END }

int bmx_find_first_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                           const bmx_vec* const* src_sub, size_t n_sub, int* found, uint64_t* idx)
{ ABI_TRY
    return find_first_impl(ctx, src_and, n_and, src_sub, n_sub, false, 0, 0, found, idx);
ABI_END }

int bmx_find_first_and_sub_range(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                                 const bmx_vec* const* src_sub, size_t n_sub, uint64_t from, uint64_t to,
                                 int* found, uint64_t* idx)
{ ABI_TRY
    return find_first_impl(ctx, src_and, n_and, src_sub, n_sub, true, from, to, found, idx);
ABI_END }

// aggregator::combine_shift_right_and  src/bmaggregator.h:552,2494 (count form: set_compute_count, :363,2595)
static int shift_right_and_impl(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, int any,
                                bmx_vec** result, int* found, uint64_t* count)
{
    ARGCHK(ctx && (n == 0 || src) && (result || count));
    if (result) *result = nullptr;
    if (found) *found = 0;
    if (count) *count = 0;
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t ncols = 0; uint64_t nbits = 0;
    std::vector<const u64*> descs(std::max<size_t>(n, 1), nullptr);
    std::vector<u32> nblk(std::max<size_t>(n, 1), 0);
    for (size_t i = 0; i < n; ++i) {
        if (!src[i] || src[i]->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
        descs[i] = src[i]->d_desc; nblk[i] = src[i]->nblocks;
        ncols = std::max(ncols, src[i]->nblocks); nbits = std::max(nbits, src[i]->nbits);
    }
    bmx_vec* v = nullptr; BlockStat* st = nullptr; u32* offs = nullptr;
    if (result && (rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) return rc;   // empty list => cleared target (:2499)
    if (!n || !ncols) {
        if (v && ncols) {
            hipError_t e = hipMemsetAsync(v->d_desc, 0, (size_t)ncols * 8, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "bmx_agg_shift_right_and", __LINE__); }
            v->counts[BMX_NULL] = ncols;
        }
        if (result) *result = v;
        return BMX_OK;
    }
    void* d_descs = nullptr; void* d_nblk = nullptr;
    if ((rc = dmalloc(ctx, &d_descs, n * 8)) || (rc = dmalloc(ctx, &d_nblk, n * 4))) { dfree(ctx, d_descs); if (v) bmx_vec_free(ctx, v); return rc; }
    hipError_t e = hipMemcpyAsync(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    size_t lds = 4 * 4096 * 4;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_shift_right_and, dim3((ncols + 3) / 4), dim3(256), lds, ctx->stream,
                           (const u64* const*)d_descs, (const u32*)d_nblk, (u32)n, ncols, opt_compress, result ? 0 : 1,
                           ctx->xcd_swz, v ? v->d_bits : nullptr, v ? v->d_desc : nullptr, st, ctx->d_slots);
        e = hipGetLastError();
    }
    if (e == hipSuccess && result && any) {
        hipLaunchKernelGGL(k_keep_first_block, dim3(1), dim3(1024), 0, ctx->stream, st, v->d_desc, ncols);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !result) {
        hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(64), 0, ctx->stream, ctx->d_slots, ctx->d_small);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess && result) rc = result_finish(ctx, v, st, offs);
    else if (e != hipSuccess) rc = fail_hip(e, "bmx_agg_shift_right_and", __LINE__);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (!rc && e2 != hipSuccess) rc = fail_hip(e2, "bmx_agg_shift_right_and", __LINE__);
    dfree(ctx, d_descs); dfree(ctx, d_nblk);
    if (rc) { if (v) bmx_vec_free(ctx, v); return rc; }
    if (result) {
        if (found) *found = (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP]) != 0;
        *result = v;
    } else {
        *count = ctx->h_small[0];
        if (found) *found = *count != 0;
    }
    return BMX_OK;
}

int bmx_agg_shift_right_and(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, int any,
                            bmx_vec** result, int* found)
{ ABI_TRY
    ARGCHK(result);
    return shift_right_and_impl(ctx, src, n, opt_compress, any, result, found, nullptr);
ABI_END }

int bmx_agg_shift_right_and_count(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, uint64_t* count)
{ ABI_TRY
    ARGCHK(count);
    return shift_right_and_impl(ctx, src, n, 0, 0, nullptr, nullptr, count);
ABI_END }

// sparse_vector_scanner<SV>::find_gt/ge/lt/le/range/eq/zero/nonzero over resident slices (bmx_kernels4.h)
// shared body of bmx_slice_compare / bmx_slice_compare_signed: `slices` are the planes the magnitude walk runs over;
// sign / sign_mode: the sign plane of a signed container and how the magnitude predicate combines with it (bmx_kernels4.h);
// null_correct: the predicate admits value 0, so NULL rows (stored as 0) must be removed
static int slice_compare_impl(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                              uint64_t size, const bmx_vec* not_null, int null_correct, const bmx_vec* sign, int sign_mode,
                              bmx_vec** result, uint64_t* count, uint64_t* plane_bytes)
{
    if (result) *result = nullptr;
    if (count) *count = 0;
    if (plane_bytes) *plane_bytes = 0;
    int rc = set_dev(ctx); if (rc) return rc;
    uint64_t nblocks64 = (size + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks64 > 65536ull * 16) { g_last_error = "vector too long"; return BMX_ERR_RANGE; }
    uint32_t ncols = (uint32_t)nblocks64;
    std::vector<const u64*> descs(std::max<size_t>(nslices, 1), nullptr);
    std::vector<u32> nblk(std::max<size_t>(nslices, 1), 0);
    for (size_t i = 0; i < nslices; ++i) {
        if (!slices[i]) continue;                                           // plane does not exist
        if (slices[i]->ctx != ctx) { g_last_error = "slice belongs to another context"; return BMX_ERR_BADARG; }
        descs[i] = slices[i]->d_desc; nblk[i] = slices[i]->nblocks;
    }
    const bool sgn = sign_mode != SIGN_NONE || pred == CMP_SRANGE;
    const bool two = pred == BMX_CMP_RANGE || pred == CMP_SRANGE;
    bmx_vec* v = nullptr; BlockStat* st = nullptr; u32* offs = nullptr;
    if (result && (rc = result_begin(ctx, size, ncols, &v, &st, &offs))) return rc;
    if (!ncols) { if (result) *result = v; return BMX_OK; }
    void* d_descs = nullptr; void* d_nblk = nullptr;
    size_t nal = std::max<size_t>(nslices, 1);
    if ((rc = dmalloc(ctx, &d_descs, nal * 8)) || (rc = dmalloc(ctx, &d_nblk, nal * 4))) { dfree(ctx, d_descs); if (v) bmx_vec_free(ctx, v); return rc; }
    hipError_t e = hipMemcpyAsync(d_descs, descs.data(), nal * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), nal * 4, hipMemcpyHostToDevice, ctx->stream);
    u64* d_stat = plane_bytes ? ctx->d_small + 8 : nullptr;
    if (e == hipSuccess && d_stat) e = hipMemsetAsync(d_stat, 0, 8, ctx->stream);
    if (e == hipSuccess) {
#define CMP_ARGS dim3((ncols + 3) / 4), dim3(256), 0, ctx->stream, \
            (const u64* const*)d_descs, (const u32*)d_nblk, (u32)nslices, ncols, pred, v0, v1, size, \
            not_null ? (const u64*)not_null->d_desc : nullptr, not_null ? not_null->nblocks : 0u, null_correct, \
            result ? 0 : 1, ctx->xcd_swz, v ? v->d_bits : nullptr, v ? v->d_desc : nullptr, st, ctx->d_slots, \
            sign ? (const u64*)sign->d_desc : nullptr, sign ? sign->nblocks : 0u, sign_mode, d_stat
        if (ctx->range_halves) {                                            // partial-block passes (fewer accumulators, more waves)
            // (measured: one bound -- half blocks 0.372 ms, quarter blocks 0.422; two bounds -- half blocks 0.536 ms, quarter blocks 0.473)
            if (sgn) { if (two) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare_halves<true, 2, true>), CMP_ARGS);
                       else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare_halves<false, 4, true>), CMP_ARGS); }
            else if (two) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare_halves<true, 2>), CMP_ARGS);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare_halves<false, 4>), CMP_ARGS);
        } else {
            if (sgn) { if (two) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare<true, true>), CMP_ARGS);
                       else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare<false, true>), CMP_ARGS); }
            else if (two) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare<true>), CMP_ARGS);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slice_compare<false>), CMP_ARGS);
        }
#undef CMP_ARGS
        e = hipGetLastError();
    }
    if (e == hipSuccess && !result) {
        hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(64), 0, ctx->stream, ctx->d_slots, ctx->d_small);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess && d_stat) e = hipMemcpyAsync(ctx->h_small + 8, d_stat, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && result) rc = result_finish(ctx, v, st, offs);
    else if (e != hipSuccess) rc = fail_hip(e, "bmx_slice_compare", __LINE__);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (!rc && e2 != hipSuccess) rc = fail_hip(e2, "bmx_slice_compare", __LINE__);
    dfree(ctx, d_descs); dfree(ctx, d_nblk);
    if (rc) { if (v) bmx_vec_free(ctx, v); return rc; }
    if (plane_bytes) *plane_bytes = ctx->h_small[8];
    if (result) {
        *result = v;
        if (count) { rc = bmx_count(ctx, v, count); if (rc) return rc; }
    } else *count = ctx->h_small[0];
    return BMX_OK;
}

static int slice_compare_unsigned(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                                  uint64_t size, const bmx_vec* not_null, bmx_vec** result, uint64_t* count, uint64_t* plane_bytes)
{
    ARGCHK(ctx && (nslices == 0 || slices) && nslices <= 64 && pred >= BMX_CMP_GT && pred <= BMX_CMP_NONZERO && (result || count));
    ARGCHK(!not_null || not_null->ctx == ctx);
    if (pred == BMX_CMP_RANGE && v1 < v0) std::swap(v0, v1);              // bm::xor_swap(from, to), :2872
    if (pred == BMX_CMP_ZERO || pred == BMX_CMP_NONZERO) v0 = 0;
    // which results can contain value 0 = where NULL elements hide (needs_null_correct_*, :1703-1735, unsigned)
    int null_correct = 0;
    switch (pred) {
    case BMX_CMP_GE: case BMX_CMP_RANGE: null_correct = v0 == 0; break;
    case BMX_CMP_LT: null_correct = v0 > 0; break;
    case BMX_CMP_LE: case BMX_CMP_ZERO: null_correct = 1; break;
    case BMX_CMP_EQ: null_correct = v0 == 0; break;
    default: break;
    }
    return slice_compare_impl(ctx, slices, nslices, pred, v0, v1, size, not_null, null_correct, nullptr, SIGN_NONE, result, count, plane_bytes);
}

int bmx_slice_compare(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                      uint64_t size, const bmx_vec* not_null, bmx_vec** result, uint64_t* count)
{ ABI_TRY
    return slice_compare_unsigned(ctx, slices, nslices, pred, v0, v1, size, not_null, result, count, nullptr);
ABI_END }

int bmx_slice_compare_stat(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                           uint64_t size, const bmx_vec* not_null, uint64_t* count, uint64_t* plane_bytes)
{ ABI_TRY
    ARGCHK(count && plane_bytes);
    return slice_compare_unsigned(ctx, slices, nslices, pred, v0, v1, size, not_null, nullptr, count, plane_bytes);
ABI_END }

// signed containers: slices[0] = the sign plane, slices[1..] = magnitude planes (s2u encoding, src/bmbmatrix.h:2536-2548)
int bmx_slice_compare_signed(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, int64_t v0, int64_t v1,
                             uint64_t size, const bmx_vec* not_null, bmx_vec** result, uint64_t* count)
{ ABI_TRY
    ARGCHK(ctx && (nslices == 0 || slices) && nslices <= 65 && pred >= BMX_CMP_GT && pred <= BMX_CMP_NONZERO && (result || count));
    ARGCHK(!not_null || not_null->ctx == ctx);
    const bmx_vec* sign = nslices ? slices[0] : nullptr;
    if (sign && sign->ctx != ctx) { g_last_error = "slice belongs to another context"; return BMX_ERR_BADARG; }
    const bmx_vec* const* mag = nslices ? slices + 1 : slices;
    const size_t nmag = nslices ? nslices - 1 : 0;
    if (pred == BMX_CMP_RANGE && v1 < v0) std::swap(v0, v1);
    // magnitude of a bound: v >= 0 -> v, v < 0 -> -(v + 1)
    auto magn = [](int64_t v) -> uint64_t { return v >= 0 ? (uint64_t)v : (uint64_t)(-(v + 1)); };
    // does value 0 satisfy the predicate?  (NULL rows are stored as sign 0, magnitude 0)
    bool admits0;
    switch (pred) {
    case BMX_CMP_GT: admits0 = 0 > v0; break;
    case BMX_CMP_GE: admits0 = 0 >= v0; break;
    case BMX_CMP_LT: admits0 = 0 < v0; break;
    case BMX_CMP_LE: admits0 = 0 <= v0; break;
    case BMX_CMP_RANGE: admits0 = v0 <= 0 && 0 <= v1; break;
    case BMX_CMP_EQ: admits0 = v0 == 0; break;
    case BMX_CMP_ZERO: admits0 = true; break;
    default: admits0 = false; break;
    }
    int kp = pred, mode = SIGN_NONE; uint64_t b0 = 0, b1 = 0;
    switch (pred) {
    case BMX_CMP_GT: if (v0 >= 0) { kp = CMP_GT; b0 = magn(v0); mode = SIGN_NONNEG_ONLY; } else { kp = CMP_LT; b0 = magn(v0); mode = SIGN_NONNEG_ALL; } break;
    case BMX_CMP_GE: if (v0 >= 0) { kp = CMP_GE; b0 = magn(v0); mode = SIGN_NONNEG_ONLY; } else { kp = CMP_LE; b0 = magn(v0); mode = SIGN_NONNEG_ALL; } break;
    case BMX_CMP_LT: if (v0 >= 0) { kp = CMP_LT; b0 = magn(v0); mode = SIGN_NEG_ALL; } else { kp = CMP_GT; b0 = magn(v0); mode = SIGN_NEG_ONLY; } break;
    case BMX_CMP_LE: if (v0 >= 0) { kp = CMP_LE; b0 = magn(v0); mode = SIGN_NEG_ALL; } else { kp = CMP_GE; b0 = magn(v0); mode = SIGN_NEG_ONLY; } break;
    case BMX_CMP_EQ: kp = CMP_EQ; b0 = magn(v0); mode = v0 >= 0 ? SIGN_NONNEG_ONLY : SIGN_NEG_ONLY; break;
    case BMX_CMP_RANGE:
        if (v0 >= 0) { kp = CMP_RANGE; b0 = magn(v0); b1 = magn(v1); mode = SIGN_NONNEG_ONLY; }
        else if (v1 < 0) { kp = CMP_RANGE; b0 = magn(v1); b1 = magn(v0); mode = SIGN_NEG_ONLY; }      // v0 <= v1 < 0: magnitudes swap order
        else { kp = CMP_SRANGE; b0 = magn(v1); b1 = magn(v0); mode = SIGN_NONE; }                   // v0 < 0 <= v1
        break;
    case BMX_CMP_ZERO: kp = CMP_ZERO; mode = SIGN_NONNEG_ONLY; break;                              // sign 0, magnitude 0
    default: kp = CMP_NONZERO; mode = SIGN_NEG_ALL; break;                                          // any magnitude bit, or negative (-1 = sign only)
    }
    return slice_compare_impl(ctx, mag, nmag, kp, b0, b1, size, not_null, admits0 ? 1 : 0, sign, mode, result, count, nullptr);
ABI_END }

// counts[q] = rows equal to values[q]: one pass over the planes whatever the number of queries (k_slice_eq_counts:
// bit-matrix transposition + hash lookup); identical to the per-query AND-SUB groups of the reference
// (src/bmsparsevec_algo.h:2593-2640).  Value 0 goes through bmx_slice_compare (NULL correction); a value with a bit
// above the planes or in an absent plane matches nothing (:2621).
int bmx_slice_eq_counts(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, const uint64_t* values, size_t n,
                        uint64_t size, const bmx_vec* not_null, uint64_t* counts)
{ ABI_TRY
    ARGCHK(ctx && (nslices == 0 || slices) && (n == 0 || (values && counts)));
    ARGCHK(!not_null || not_null->ctx == ctx);
    if (nslices > 32) { g_last_error = "more than 32 planes: use the pipeline form (bmx_pipeline_*)"; return BMX_ERR_RANGE; }
    int rc = set_dev(ctx); if (rc) return rc;
    for (size_t q = 0; q < n; ++q) counts[q] = 0;
    uint64_t nblocks64 = (size + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks64 > 65536ull * 16) { g_last_error = "vector too long"; return BMX_ERR_RANGE; }
    const uint32_t ncols = (uint32_t)nblocks64;
    uint32_t present = 0;
    for (size_t i = 0; i < nslices; ++i) if (slices[i]) {
        if (slices[i]->ctx != ctx) { g_last_error = "slice belongs to another context"; return BMX_ERR_BADARG; }
        present |= 1u << i;
    }
    // unique non-zero values that can match at all; value 0 through the comparison kernel
    std::vector<uint32_t> uniq; std::vector<int64_t> slot(n, -1);
    {
        std::unordered_map<uint32_t, uint32_t> seen;
        bool zero_done = false; uint64_t zero_count = 0;
        for (size_t q = 0; q < n; ++q) {
            uint64_t v = values[q];
            if (!v) {
                if (!zero_done) { if ((rc = bmx_slice_compare(ctx, slices, nslices, BMX_CMP_EQ, 0, 0, size, not_null, nullptr, &zero_count))) return rc; zero_done = true; }
                counts[q] = zero_count; continue;
            }
            if ((v >> 32) || ((uint32_t)v & ~present)) continue;                    // impossible value: 0 rows
            auto it = seen.find((uint32_t)v);
            if (it == seen.end()) { it = seen.emplace((uint32_t)v, (uint32_t)uniq.size()).first; uniq.push_back((uint32_t)v); }
            slot[q] = it->second;
        }
    }
    if (uniq.empty() || !ncols) return BMX_OK;
    // planes holding GAP blocks are expanded to raw bits once (k_vec_expand); the others are read through their tables
    EqPlanes pl; memset(&pl, 0, sizeof(pl));
    std::vector<void*> temps;
    auto free_temps = [&]() { for (void* t : temps) dfree(ctx, t); };
    for (size_t i = 0; i < nslices; ++i) {
        const bmx_vec* sv = slices[i];
        if (!sv) continue;
        pl.nblk[i] = sv->nblocks;
        if (sv->counts[BMX_GAP]) {
            void* raw = nullptr;
            if ((rc = dmalloc(ctx, &raw, (size_t)ncols * 8192))) { free_temps(); return rc; }
            temps.push_back(raw);
            hipLaunchKernelGGL(k_vec_expand, dim3((ncols + 3) / 4), dim3(256), 0, ctx->stream, sv->d_desc, sv->nblocks, ncols, (uint4*)raw);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); free_temps(); return fail_hip(e, "k_vec_expand", __LINE__); }
            pl.raw[i] = (const uint4*)raw;
        } else pl.desc[i] = sv->d_desc;
    }
    std::vector<uint64_t> ucount(uniq.size(), 0);
    // up to 2,048 values: the table with ordinals (2 workgroups per CU); more: the lean {key, count} table shared by 512
    // threads, up to EQB_MAX_VALUES per pass over the planes (eq_big: 0 = never, 1 = always, -1 = by the batch size)
    // (the lean table wants up to 156 KiB of LDS per workgroup: without that much the 2,048-value form takes every batch)
    // automatic: the lean table with 768 threads for every batch over more than 16 planes; up to 16 planes (dense value
    // spaces: most rows are looked up) the 2,048-value kernel keeps batches it can take in one pass (2.33 vs 2.86 ms)
    const bool big = (ctx->eq_big == 1 || (ctx->eq_big < 0 && uniq.size() > ((ctx->eq_big_shape == 2 && nslices > 16) ? 0u : 2048u))) && ctx->max_lds_bytes >= EQB_SLOTS(EQB_MAX_VALUES) * 8u + (1u << 15) + 8u * 512u * 4u;
    const size_t cap = big ? (ctx->eq_big_shape == 2 ? 8704u : EQB_MAX_VALUES) : 2048;       // (768 threads: 24 KiB of queues)
    const size_t npass = (uniq.size() + cap - 1) / cap;
    const size_t CHUNK = (uniq.size() + npass - 1) / npass;
    for (size_t u0 = 0; u0 < uniq.size() && !rc; u0 += CHUNK) {
        const uint32_t nv = (uint32_t)std::min(CHUNK, uniq.size() - u0);
        if (big) {
            const uint32_t tab = EQB_SLOTS(nv);
            std::vector<uint32_t> keys(tab, 0), where(nv);
            for (uint32_t k = 0; k < nv; ++k) {
                uint32_t v = uniq[u0 + k], h = (uint32_t)(((uint64_t)(uint32_t)(v * 0x9E3779B1u) * tab) >> 32);
                while (keys[h]) h = h + 1u == tab ? 0u : h + 1u;
                keys[h] = v; where[k] = h;
            }
            void* d_tab = nullptr; void* d_cnt = nullptr;
            if ((rc = dmalloc(ctx, &d_tab, (size_t)tab * 4)) || (rc = dmalloc(ctx, &d_cnt, (size_t)tab * 8))) { dfree(ctx, d_tab); break; }
            std::vector<uint64_t> scount(tab, 0);
            rc = h2d_staged(ctx, d_tab, keys.data(), (size_t)tab * 4);
            hipError_t e = rc ? hipSuccess : hipMemsetAsync(d_cnt, 0, (size_t)tab * 8, ctx->stream);
            if (!rc && e == hipSuccess) {
                // eq_big_shape: 0 = 512 threads, 16 KiB filter, 1,024-entry queues; 1 = 512 threads, 32 KiB filter, 512-entry queues;
                // 2 = 768 threads with the registers held to 3 waves per SIMD (8 filter reads in flight instead of 32), 32 KiB + 512
                const int shp = ctx->eq_big_shape;
                const u32 wg = shp == 2 ? 768u : 512u, nw = wg / 64u;
                size_t lds = (size_t)tab * 8 + (shp >= 1 ? (1u << 15) + nw * 512u * 4u : (1u << 14) + nw * 1024u * 4u);
                auto eqfn = shp == 2 ? (nslices <= 16 ? k_slice_eq_counts_big<16, 18, 512, 768, 3, 8> : k_slice_eq_counts_big<32, 18, 512, 768, 3, 4>)   /* (32 planes: 4 filter reads in flight -- 8 need 3 VGPRs more than three waves per SIMD leave: 12 B of scratch) */
                          : shp == 1 ? (nslices <= 16 ? k_slice_eq_counts_big<16, 18, 512> : k_slice_eq_counts_big<32, 18, 512>)
                                     : (nslices <= 16 ? k_slice_eq_counts_big<16, 17, 1024> : k_slice_eq_counts_big<32, 17, 1024>);
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(eqfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess) {
                    u32 grid = std::min<u32>((ncols + nw - 1u) / nw, 256u);
                    hipLaunchKernelGGL(eqfn, dim3(grid), dim3(wg), lds, ctx->stream, pl, (u32)nslices, ncols, size, (const u32*)d_tab, tab, (u64*)d_cnt);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipMemcpyAsync(scount.data(), d_cnt, (size_t)tab * 8, hipMemcpyDeviceToHost, ctx->stream);
            }
            hipError_t e2 = hipStreamSynchronize(ctx->stream);
            dfree(ctx, d_tab); dfree(ctx, d_cnt);
            if (!rc && (e != hipSuccess || e2 != hipSuccess)) rc = fail_hip(e != hipSuccess ? e : e2, "bmx_slice_eq_counts", __LINE__);
            if (!rc) for (uint32_t k = 0; k < nv; ++k) ucount[u0 + k] = scount[where[k]];
            continue;
        }
        uint32_t tab = 64; while (tab < 2u * nv) tab <<= 1;
        uint32_t shift = 32; for (uint32_t t = tab; t > 1; t >>= 1) --shift;
        // host-built open-addressing table: [keys u32 x tab][idx u16 x tab]
        std::vector<uint32_t> blob(tab + tab / 2, 0);
        uint32_t* keys = blob.data(); uint16_t* idx = reinterpret_cast<uint16_t*>(blob.data() + tab);
        for (uint32_t k = 0; k < nv; ++k) {
            uint32_t v = uniq[u0 + k], h = (v * 0x9E3779B1u) >> shift;
            while (keys[h]) h = (h + 1u) & (tab - 1u);
            keys[h] = v; idx[h] = (uint16_t)k;
        }
        void* d_tab = nullptr; void* d_cnt = nullptr;
        if ((rc = dmalloc(ctx, &d_tab, blob.size() * 4)) || (rc = dmalloc(ctx, &d_cnt, (size_t)nv * 8))) { dfree(ctx, d_tab); break; }
        rc = h2d_staged(ctx, d_tab, blob.data(), blob.size() * 4);
        hipError_t e = rc ? hipSuccess : hipMemsetAsync(d_cnt, 0, (size_t)nv * 8, ctx->stream);
        if (!rc && e == hipSuccess) {
            size_t lds = (size_t)tab * 4 + (size_t)nv * 4 + EQ_FILTER_WORDS * 4 + 4 * EQ_QUEUE * 4 + (size_t)tab * 2;
            auto eqfn = nslices <= 16 ? k_slice_eq_counts<16> : k_slice_eq_counts<32>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(eqfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) {
                u32 grid = std::min<u32>((ncols + 3u) / 4u, 512u);
                hipLaunchKernelGGL(eqfn, dim3(grid), dim3(256), lds, ctx->stream, pl, (u32)nslices, ncols, size,
                                   (const u32*)d_tab, (const u16*)((const u32*)d_tab + tab), tab, shift, nv, (u64*)d_cnt);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(ucount.data() + u0, d_cnt, (size_t)nv * 8, hipMemcpyDeviceToHost, ctx->stream);
        }
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        dfree(ctx, d_tab); dfree(ctx, d_cnt);
        if (!rc && (e != hipSuccess || e2 != hipSuccess)) rc = fail_hip(e != hipSuccess ? e : e2, "bmx_slice_eq_counts", __LINE__);
    }
    (void)hipStreamSynchronize(ctx->stream);
    free_temps();
    if (rc) return rc;
    for (size_t q = 0; q < n; ++q) if (slot[q] >= 0) counts[q] = ucount[(size_t)slot[q]];
    return BMX_OK;
ABI_END }

// ---------------------------------------------------------------------------
// rank / select
// ---------------------------------------------------------------------------
int bmx_rs_build(bmx_ctx* ctx, const bmx_vec* v, bmx_rs** out)
{ ABI_TRY
    ARGCHK(ctx && v && out && v->ctx == ctx);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    bmx_rs* rs = new (std::nothrow) bmx_rs();
    if (!rs) return BMX_ERR_BADALLOC;
    memset(rs, 0, sizeof(*rs));
    rs->ctx = ctx; rs->nblocks = v->nblocks;
    uint32_t n = std::max<uint32_t>(v->nblocks, 1);
    size_t b1 = (size_t)n * 4, b2 = (size_t)n * 8, b3 = (size_t)n * 8, b4 = (size_t)n * 128;
    if ((rc = dmalloc(ctx, (void**)&rs->d_bcount, b1)) || (rc = dmalloc(ctx, (void**)&rs->d_sub, b2)) ||
        (rc = dmalloc(ctx, (void**)&rs->d_rcount, b3)) || (rc = dmalloc(ctx, (void**)&rs->d_cum, b4)) ||
        (rc = dmalloc(ctx, (void**)&rs->d_gidx, b4))) { bmx_rs_free(ctx, rs); return rc; }
    rs->bytes = b1 + b2 + b3 + 2 * b4;
    if (v->nblocks) {
#define RSCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_rs_free(ctx, rs); return r_; } } while (0)
        hipLaunchKernelGGL(k_rs_build, dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           v->d_desc, v->nblocks, rs->d_bcount, rs->d_sub, rs->d_cum, rs->d_gidx);
        RSCHK(hipGetLastError());
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, ctx->stream, rs->d_bcount, v->nblocks, rs->d_rcount, ctx->d_small);
        RSCHK(hipGetLastError());
        uint32_t shift = 0;
        while (((v->nblocks + (1u << shift) - 1u) >> shift) > 2048u) ++shift;
        rs->sample_shift = shift; rs->nsamples = (v->nblocks + (1u << shift) - 1u) >> shift;
        if ((rc = dmalloc(ctx, (void**)&rs->d_sample, (size_t)rs->nsamples * 8))) { bmx_rs_free(ctx, rs); return rc; }
        hipLaunchKernelGGL(k_rs_sample, dim3((rs->nsamples + 255) / 256), dim3(256), 0, ctx->stream,
                           rs->d_rcount, v->nblocks, shift, rs->nsamples, rs->d_sample);
        RSCHK(hipGetLastError());
        // rank lines (bmx_kernels6.h): the vector once more, interleaved with its running counts -- one line per rank query.
        // Memory policy (rs_lines 1): they are built where they cost no more than 2 x what the vector itself holds on the
        // device, i.e. for vectors with bit-blocks in more than about half of their block columns (configs[3]'s 66%); a sparse vector (GAP / NULL / FULL blocks: a 4e9-bit
        // operand of configs[4] is 3.4 MB, its lines would be 539 MB) keeps the table kernels (k_rank_l / k_select_l over the
        // running counts, src/bmrs.h:39-155 is 0.7 MB for such a vector too).  Line numbers are 32-bit: 69 lines per block
        // pass 2^32 at 62.2 M blocks = 510 GB of bit-blocks under this policy, more than a device holds; refused anyway.
        const size_t lines_bytes = (size_t)v->nblocks * RL_LINES * 128u;
        const bool lines_fit = (uint64_t)v->nblocks * RL_LINES < 0xFFFFFFFFull;
        const bool want_lines = lines_fit && (ctx->rs_lines == 2 || (ctx->rs_lines == 1 && (double)lines_bytes <= 2.0 * (double)v->bytes));
        if (want_lines) {
            size_t bl = lines_bytes;
            if ((rc = dmalloc(ctx, (void**)&rs->d_lines, bl)) || (rc = dmalloc(ctx, (void**)&rs->d_dir8, (size_t)v->nblocks * 16u))) { bmx_rs_free(ctx, rs); return rc; }
            rs->bytes += bl + (size_t)v->nblocks * 16u;
            hipLaunchKernelGGL(k_rs_lines, dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                               v->d_desc, v->nblocks, (const u64*)rs->d_rcount, rs->d_lines, rs->d_dir8);
            RSCHK(hipGetLastError());
        }
        RSCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
        RSCHK(hipStreamSynchronize(ctx->stream));
        rs->count = ctx->h_small[0];
        if (rs->count && ctx->rs_select_sel != 0) {
            // select lines (bmx_kernels11.h): the ones' positions, 60 (16-bit offsets) or 30 (32-bit) per 128-byte line.  Memory
            // policy (rs_select_sel -1): where they cost no more than 2 x what the vector and its rank lines hold on the device.
            // The 16-bit form is tried unless the average spacing of the ones already says that 60 of them span half a block.
            const double budget = 2.0 * ((double)v->bytes + (rs->d_lines ? (double)lines_bytes : 0.0));
            const bool try16 = ctx->rs_select_sel != 2 && (double)v->nblocks * 65536.0 / (double)rs->count * 60.0 < 32768.0;
            for (int bits = try16 ? 16 : 32; bits <= 32 && !rs->d_sel; bits += 16) {
                const uint32_t K = bits == 16 ? 60u : 30u;
                const uint64_t nsel = (rs->count + K - 1u) / K;
                const size_t sb = (size_t)nsel * SL_BYTES;
                if (ctx->rs_select_sel == -1 && (double)sb > budget) break;
                u8* d = nullptr;
                if (dmalloc(ctx, (void**)&d, sb) != BMX_OK) break;           // (an optional index: the directory kernels serve)
                RSCHK(hipMemsetAsync(ctx->d_small + 32, 0, 8, ctx->stream));
                if (bits == 16) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_sel_build<u16>), dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks, (const u64*)rs->d_rcount, d);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_sel_check<u16>), dim3((u32)((nsel + 255u) / 256u)), dim3(256), 0, ctx->stream, (const u8*)d, (u64)nsel, (u64)rs->count, (const u64*)rs->d_rcount, v->nblocks, (u32*)(ctx->d_small + 32));
                } else {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_sel_build<u32>), dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks, (const u64*)rs->d_rcount, d);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rs_sel_check<u32>), dim3((u32)((nsel + 255u) / 256u)), dim3(256), 0, ctx->stream, (const u8*)d, (u64)nsel, (u64)rs->count, (const u64*)rs->d_rcount, v->nblocks, (u32*)(ctx->d_small + 32));
                }
                hipError_t e_ = hipGetLastError();
                if (e_ == hipSuccess) e_ = hipMemcpyAsync(ctx->h_small + 32, ctx->d_small + 32, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e_ == hipSuccess) e_ = hipStreamSynchronize(ctx->stream);
                if (e_ != hipSuccess) { dfree(ctx, d); int r_ = fail_hip(e_, "select lines", __LINE__); bmx_rs_free(ctx, rs); return r_; }
                if (ctx->h_small[32] != 0) { dfree(ctx, d); continue; }      // a line spans >= 2^bits bits: the wider form
                rs->d_sel = d; rs->sel_bits = (uint32_t)bits; rs->sel_lines = nsel; rs->bytes += sb;
            }
        }
        if (rs->d_lines && rs->count) {
            // select directory over the lines: the line of every 2^shift-th one (+ sentinel); k_select_sdir
            const uint64_t nlines = (uint64_t)v->nblocks * RL_LINES;
            uint32_t sh = (uint32_t)ctx->rs_sdir_shift;
            if (ctx->rs_sdir_shift <= 0) {                        // automatic: 2^sh next to 10 x (ones per line)
                const double want = 10.0 * (double)rs->count / (double)nlines;
                sh = 6u; while (sh < 20u && (double)(1ull << sh) * 1.4142 < want) ++sh;
            }
            while (((rs->count >> sh) + 2ull) * 4ull > (8ull << 20) && sh < 24u) ++sh;
            rs->sdir_shift = sh; rs->sdir_entries = ((rs->count + (1ull << sh) - 1ull) >> sh) + 1ull;
            if ((rc = dmalloc(ctx, (void**)&rs->d_sdir, (size_t)rs->sdir_entries * 4u + 16u))) { bmx_rs_free(ctx, rs); return rc; }
            rs->bytes += (size_t)rs->sdir_entries * 4u;
            hipLaunchKernelGGL(k_rs_sdir, dim3((u32)((nlines + 255u) / 256u)), dim3(256), 0, ctx->stream,
                               (const u32*)rs->d_lines, (u64)nlines, (u64)rs->count, sh, rs->d_sdir, (u64)rs->sdir_entries);
            RSCHK(hipGetLastError());
            // the directory's summary for LDS (k_select_top): one entry per 2^stop_shift ones, at most 65,535 + the sentinel; an entry
            // is the position of its one to 1 / 2^fb of a line (fb <= 3: as fine as the 16-bit offsets of a group of 64 entries allow)
            uint32_t ssh = sh;
            while (((rs->count + (1ull << ssh) - 1ull) >> ssh) + 1ull > STOP_ENTRIES && ssh < 40u) ++ssh;
            const uint32_t n_top = (uint32_t)(((rs->count + (1ull << ssh) - 1ull) >> ssh) + 1ull);
            u32* d_p8 = nullptr;
            if (nlines < (1ull << 28) && dmalloc(ctx, (void**)&d_p8, (size_t)n_top * 4u + 64u) == BMX_OK) {
                if ((rc = dmalloc(ctx, (void**)&rs->d_stop, STOP_BYTES + 64u))) { dfree(ctx, d_p8); bmx_rs_free(ctx, rs); return rc; }
                hipError_t e_ = hipMemsetAsync(rs->d_stop, 0, STOP_BYTES + 64u, ctx->stream);
                if (e_ == hipSuccess) e_ = hipMemsetAsync(ctx->d_small + 32, 0, 8, ctx->stream);
                if (e_ == hipSuccess) {
                    hipLaunchKernelGGL(k_rs_stop_pos, dim3((n_top + 255u) / 256u), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, (const u32*)rs->d_sdir, (u64)rs->sdir_entries,
                                       sh, ssh, n_top, (u64)rs->count, d_p8);
                    hipLaunchKernelGGL(k_rs_stop_range, dim3((n_top / STOP_GROUP + 256u) / 256u), dim3(256), 0, ctx->stream, (const u32*)d_p8, n_top, (u32*)(ctx->d_small + 32));
                    e_ = hipGetLastError();
                }
                if (e_ == hipSuccess) e_ = hipMemcpyAsync(ctx->h_small + 32, ctx->d_small + 32, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e_ == hipSuccess) e_ = hipStreamSynchronize(ctx->stream);
                if (e_ != hipSuccess) { dfree(ctx, d_p8); int r_ = fail_hip(e_, "select summary", __LINE__); bmx_rs_free(ctx, rs); return r_; }
                const uint32_t spread = (uint32_t)ctx->h_small[32];          // in eighths of a line
                uint32_t fb = 3u;
                while (fb > 0u && (spread >> (3u - fb)) > 65000u) --fb;
                if ((spread >> (3u - fb)) > 65000u) { dfree(ctx, rs->d_stop); rs->d_stop = nullptr; }   // (64 entries spread over more than 65,000 lines somewhere: the global directory serves)
                else {
                    hipLaunchKernelGGL(k_rs_stop_pack, dim3((n_top + 255u) / 256u), dim3(256), 0, ctx->stream, (const u32*)d_p8, n_top, 3u - fb, rs->d_stop, (u16*)(rs->d_stop + STOP_BASES));
                    e_ = hipGetLastError();
                    if (e_ != hipSuccess) { dfree(ctx, d_p8); int r_ = fail_hip(e_, "k_rs_stop_pack", __LINE__); bmx_rs_free(ctx, rs); return r_; }
                    rs->stop_shift = ssh; rs->stop_fb = fb; rs->bytes += STOP_BYTES;
                }
                dfree(ctx, d_p8);
            }
        }
#undef RSCHK
    }
    *out = rs;
    return BMX_OK;
ABI_END }

int bmx_rs_info(const bmx_rs* rs, uint64_t* bytes, int* has_lines)
{ ABI_TRY
    ARGCHK(rs);
    if (bytes) *bytes = rs->bytes;
    if (has_lines) *has_lines = rs->d_lines ? 1 : 0;
    return BMX_OK;
ABI_END }

int bmx_rs_select_format(const bmx_rs* rs, int* offset_bits, uint64_t* bytes)
{ ABI_TRY
    ARGCHK(rs);
    if (offset_bits) *offset_bits = rs->d_sel ? (int)rs->sel_bits : 0;
    if (bytes) *bytes = rs->d_sel ? rs->sel_lines * SL_BYTES : 0;
    return BMX_OK;
ABI_END }

int bmx_rs_free(bmx_ctx* ctx, bmx_rs* rs)
{ ABI_TRY
    if (!rs) return BMX_OK;
    ARGCHK(ctx && rs->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(ctx, rs->d_bcount); dfree(ctx, rs->d_sub); dfree(ctx, rs->d_rcount); dfree(ctx, rs->d_cum); dfree(ctx, rs->d_gidx); dfree(ctx, rs->d_sample); dfree(ctx, rs->d_lines); dfree(ctx, rs->d_dir8); dfree(ctx, rs->d_sdir); dfree(ctx, rs->d_stop); dfree(ctx, rs->d_sel);
    delete rs;
    return BMX_OK;
ABI_END }

int bmx_rs_count(const bmx_rs* rs, uint64_t* count) { ABI_TRY ARGCHK(rs && count); *count = rs->count; return BMX_OK; ABI_END }

int bmx_rs_export(bmx_ctx* ctx, const bmx_rs* rs, uint32_t* bcount, uint64_t* sub_count)
{ ABI_TRY
    ARGCHK(ctx && rs && rs->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    if (bcount && rs->nblocks) HIPCHK(hipMemcpyAsync(bcount, rs->d_bcount, (size_t)rs->nblocks * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (sub_count && rs->nblocks) HIPCHK(hipMemcpyAsync(sub_count, rs->d_sub, (size_t)rs->nblocks * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
ABI_END }

#define RS_LANES_DEFAULT 2
#define RS_SELECT_LANES_DEFAULT 4
static u32 query_grid(size_t q) { return (u32)std::min<size_t>((q * 8 + 255) / 256, 256u * 16u); }

int bmx_rank_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* d_n, size_t q, uint64_t* d_out)
{ ABI_TRY
    ARGCHK(ctx && v && rs && v->ctx == ctx && rs->ctx == ctx && rs->nblocks == v->nblocks && (q == 0 || (d_n && d_out)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    // lanes per query (rs_lanes: 0 = automatic; batches too small to fill the chip keep the 8-lane kernel)
    int lpq = ctx->rs_lanes ? ctx->rs_lanes : (q >= (1u << 16) ? RS_LANES_DEFAULT : 8);
    u32 grid = (u32)std::min<size_t>((q * (size_t)lpq + 255) / 256, 256u * 16u);
#define RANK_ARGS dim3(grid), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks, \
                  rs->d_rcount, rs->d_cum, rs->d_gidx, rs->count, (const u64*)d_n, (u64)q, (u64*)d_out
    if (rs->d_lines && lpq != 8) {
        if (lpq == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rank_lines<2>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, v->nblocks, rs->count, (const u64*)d_n, (u64)q, (u64*)d_out);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rank_lines<4>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, v->nblocks, rs->count, (const u64*)d_n, (u64)q, (u64*)d_out);
    }
    else if (lpq == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rank_l<2>), RANK_ARGS);
    else if (lpq == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rank_l<4>), RANK_ARGS);
    else hipLaunchKernelGGL(k_rank, RANK_ARGS);
#undef RANK_ARGS
    KCHK();
    return BMX_OK;
ABI_END }

int bmx_select_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* d_rank, size_t q,
                         uint64_t* d_pos, uint8_t* d_found)
{ ABI_TRY
    ARGCHK(ctx && v && rs && v->ctx == ctx && rs->ctx == ctx && rs->nblocks == v->nblocks && (q == 0 || (d_rank && d_pos && d_found)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    if (rs->d_sel && ctx->rs_select_sel != 0) {
        // select lines: one lane and one 128-byte line per query, no search (bmx_kernels11.h); any batch size, any order
        const u32 g = (u32)std::min<size_t>((q + 511) / 512, 256u * 8u);
        if (rs->sel_bits == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_sel<u16>), dim3(g), dim3(256), 0, ctx->stream, (const u8*)rs->d_sel, rs->count, (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_sel<u32>), dim3(g), dim3(256), 0, ctx->stream, (const u8*)rs->d_sel, rs->count, (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
        KCHK();
        return BMX_OK;
    }
    int lpq = ctx->rs_lanes ? ctx->rs_lanes : (q >= (1u << 16) ? RS_SELECT_LANES_DEFAULT : 8);
    // big batches over a vector whose directory summary fits LDS: k_select_top, two lanes per query (profiles/r05_select: 10 M random
    // selects on configs[3] 0.377 ms against 0.416 / 0.436 for the global-directory kernel with four / two lanes, 100 M: 3.63 against
    // 4.09; at 1 M the 129 KiB every workgroup copies first cost more than they save: 0.058 against 0.044 -- taken from 4 M queries)
    const bool top_ok = rs->d_stop && rs->d_sdir && ctx->rs_select_lines == 2 && ctx->rs_select_top != 0 &&
                        (ctx->rs_select_top == 1 || (q >= (1u << 22) && !ctx->rs_sorted_hint)) && q < (1ull << 32) && ctx->max_lds_bytes >= STOP_BYTES + 16384u;
    if (top_ok && !ctx->rs_lanes) lpq = 2;
    // ranks the caller says arrive in ascending order (cursor-style enumeration): neighbours share lines and directory entries;
    // the global-directory kernel with two lanes per query is the fastest there (10 M: 0.25 ms against 0.38 random)
    if (ctx->rs_sorted_hint && !ctx->rs_lanes && q >= (1u << 16)) lpq = 2;
    u32 grid = (u32)std::min<size_t>((q * (size_t)lpq + 255) / 256, 256u * 16u);
#define SEL_ARGS dim3(grid), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks, \
                 rs->d_rcount, rs->d_cum, rs->d_gidx, rs->d_sample, rs->nsamples, rs->sample_shift, rs->count, \
                 (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found
    const bool top = top_ok && lpq != 8;
    if (top) {
        // the directory's summary in LDS: one 1024-thread workgroup per CU (129 KiB of LDS each), one global read per query
        const size_t lds = STOP_BYTES + 16u * 64u * 16u;                          // the summary + a queue of 64 parked queries per wave
        auto fn = lpq == 2 ? k_select_top<2> : k_select_top<4>;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const u32 g = (u32)std::min<size_t>((q * (size_t)lpq + 1023) / 1024, 256u);
        hipLaunchKernelGGL(fn, dim3(g), dim3(1024), lds, ctx->stream, (const u32*)rs->d_lines, (const u32*)rs->d_stop, rs->stop_shift, rs->stop_fb, rs->count,
                           (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
    }
    else if (rs->d_sdir && lpq != 8 && ctx->rs_select_lines == 2) {
        if (lpq == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_sdir<2>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, (const u32*)rs->d_sdir,
                                         rs->sdir_shift, rs->count, (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_sdir<4>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, (const u32*)rs->d_sdir,
                                rs->sdir_shift, rs->count, (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
    }
    else if (rs->d_lines && lpq != 8 && ctx->rs_select_lines) {
        if (lpq == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_lines<2>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, (const u16*)rs->d_dir8,
                                         v->nblocks, (const u64*)rs->d_rcount, (const u64*)rs->d_sample, rs->nsamples, rs->sample_shift, rs->count,
                                         (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_lines<4>), dim3(grid), dim3(256), 0, ctx->stream, (const u32*)rs->d_lines, (const u16*)rs->d_dir8,
                                v->nblocks, (const u64*)rs->d_rcount, (const u64*)rs->d_sample, rs->nsamples, rs->sample_shift, rs->count,
                                (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
    }
    else if (lpq == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_l<2>), SEL_ARGS);
    else if (lpq == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_select_l<4>), SEL_ARGS);
    else hipLaunchKernelGGL(k_select, SEL_ARGS);
#undef SEL_ARGS
    KCHK();
    return BMX_OK;
ABI_END }

int bmx_rank_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* n, size_t q, uint64_t* out)
{ ABI_TRY
    ARGCHK(ctx && (q == 0 || (n && out)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    u64* d = nullptr;
    if ((rc = dmalloc(ctx, (void**)&d, q * 16))) return rc;
    hipError_t e = hipMemcpyAsync(d, n, q * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = bmx_rank_batch_dev(ctx, v, rs, d, q, d + q);
        if (!rc) e = hipMemcpyAsync(out, d + q, q * 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    dfree(ctx, d);
    if (e != hipSuccess) return fail_hip(e, "bmx_rank_batch", __LINE__);
    return rc;
ABI_END }

int bmx_select_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* rank, size_t q,
                     uint64_t* pos, uint8_t* found)
{ ABI_TRY
    ARGCHK(ctx && (q == 0 || (rank && pos && found)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    u64* d = nullptr;
    if ((rc = dmalloc(ctx, (void**)&d, q * 17))) return rc;
    hipError_t e = hipMemcpyAsync(d, rank, q * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        rc = bmx_select_batch_dev(ctx, v, rs, d, q, d + q, (uint8_t*)(d + 2 * q));
        if (!rc) e = hipMemcpyAsync(pos, d + q, q * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (!rc && e == hipSuccess) e = hipMemcpyAsync(found, d + 2 * q, q, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    dfree(ctx, d);
    if (e != hipSuccess) return fail_hip(e, "bmx_select_batch", __LINE__);
    return rc;
ABI_END }

} // extern "C"
