// bmx_group.hip -- multi-GPU group layer of libbmx.so (include/bmx.h "device groups").
//
// Reference: there is none to translate -- bm::aggregator is single-threaded (src/bmaggregator.h:824-853).
// What the layer relies on is the column independence of the reference's loops (combine_or :1113-1121,
// combine_and_sub :1184-1218, bit_and src/bm.h:6226-6271): result block (i,j) depends on operand blocks (i,j)
// only, so the linear block range is cut into one contiguous shard per device, every member runs the
// single-device engine (bmx.hip) over its shard, and the only exchange is the sum of the popcounts.
//
// Host-only code: it composes the C-ABI of bmx.hip plus the two asynchronous internals of bmx_internal.h.
#include "bmx_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <chrono>
#include <new>
#include <thread>

// a cut of the linear block range [0, nblocks) into one contiguous piece per member: member m holds blocks
// [bounds[m], bounds[m + 1]).  ONE partition per (group, nblocks): every vector of that length is cut at the same
// borders, so block columns stay aligned across the operands of an operation.
struct bmx_partition {
    uint32_t nblocks = 0;
    std::vector<uint32_t> bounds;         // n + 1 entries, bounds[0] = 0, bounds[n] = nblocks
};
typedef std::shared_ptr<const bmx_partition> part_ref;

// persistent per-member host workers (members 1..n-1; member 0 runs on the calling thread): the synchronous
// single-device entry points are dispatched to them instead of spawning n threads per call
struct bmx_workers {
    std::mutex call_mu;                         // serialises for_each_member callers
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> th;
    const std::function<int(int)>* fn = nullptr;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;
    std::vector<std::string> msg;
};

struct bmx_group {
    int n = 0, flags = 0;
    std::vector<bmx_ctx*> ctx;
    std::map<uint32_t, part_ref> parts;   // partition in force per vector length (default: equal block counts)
    bmx_workers* wk = nullptr;
    // RCCL (optional, loaded on demand: the library has no link-time dependency on librccl)
    void* rccl = nullptr;
    std::vector<void*> comm;
    int (*p_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*p_group_start)() = nullptr;
    int (*p_group_end)() = nullptr;
    int (*p_comm_destroy)(void*) = nullptr;
    int (*p_comm_count)(void*, int*) = nullptr;
    const char* (*p_errstr)(int) = nullptr;
};

struct bmx_gvec {
    bmx_group* g;
    uint64_t nbits; uint32_t nblocks;
    part_ref part;                        // the cut this vector was sharded with
    std::vector<bmx_vec*> shard;          // shard[m] lives on g->ctx[m], blocks [part->bounds[m], part->bounds[m + 1])
};

struct bmx_grs {
    bmx_group* g;
    const bmx_gvec* v;
    std::vector<bmx_rs*> rs;              // rs[m]: index of shard m (local block numbering)
    std::vector<uint64_t> before;         // ones in the shards before m (n + 1 entries; before[n] = total)
};

struct bmx_gpipeline {
    bmx_group* g;
    uint32_t ngroups;
    std::vector<bmx_pipeline*> pipe;
    std::vector<u64*> d_counts;           // per member: ngroups x u64 on the device
    u64* h_counts = nullptr;              // pinned, n x ngroups
    std::vector<hipEvent_t> ev0, ev1, ev2;
    std::vector<float> last_ms, last_xchg_ms;
    uint64_t search_limit = ~0ull;        // pipeline::set_search_count_limit: forwarded to every member's pipeline
};

static void equal_range(uint32_t nblocks, int m, int n, uint32_t* lo, uint32_t* hi)
{
    uint32_t q = nblocks / (uint32_t)n, r = nblocks % (uint32_t)n, mm = (uint32_t)m;
    *lo = mm * q + std::min(mm, r);
    *hi = *lo + q + (mm < r ? 1u : 0u);
}

// the partition in force for vectors of `nblocks` blocks (created on first use: equal block counts)
static part_ref part_for(bmx_group* g, uint32_t nblocks)
{
    auto it = g->parts.find(nblocks);
    if (it != g->parts.end()) return it->second;
    auto p = std::make_shared<bmx_partition>();
    p->nblocks = nblocks;
    p->bounds.resize((size_t)g->n + 1);
    for (int m = 0; m < g->n; ++m) { uint32_t lo, hi; equal_range(nblocks, m, g->n, &lo, &hi); p->bounds[(size_t)m] = lo; p->bounds[(size_t)m + 1] = hi; }
    part_ref r = p;
    g->parts[nblocks] = r;
    return r;
}

static inline void shard_of(const bmx_gvec* v, int m, uint32_t* lo, uint32_t* hi)
{
    *lo = v->part->bounds[(size_t)m]; *hi = v->part->bounds[(size_t)m + 1];
}

static void worker_main(bmx_group* g, int m)
{
    bmx_workers* w = g->wk;
    (void)hipSetDevice(g->ctx[(size_t)m]->device);
    uint64_t seen = 0;
    for (;;) {
        const std::function<int(int)>* fn;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv_go.wait(lk, [&] { return w->stop || w->gen != seen; });
            if (w->stop) return;
            seen = w->gen; fn = w->fn;
        }
        int rc;
        try { rc = (*fn)(m); }                                    // (a member lambda allocates: nothing may unwind out of a worker)
        catch (const std::bad_alloc&) { rc = BMX_ERR_BADALLOC; bmx_set_last_error("out of host memory in a group member"); }
        catch (...) { rc = BMX_ERR_DEVICE; bmx_set_last_error("unexpected exception in a group member"); }
        std::string msg = rc ? bmx_last_error() : "";
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->rc[(size_t)m] = rc; w->msg[(size_t)m] = std::move(msg);
            if (--w->pending == 0) w->cv_done.notify_one();
        }
    }
}

static int workers_start(bmx_group* g)
{
    if (g->n <= 1) return BMX_OK;
    g->wk = new (std::nothrow) bmx_workers();
    if (!g->wk) return BMX_ERR_BADALLOC;
    g->wk->rc.assign((size_t)g->n, BMX_OK); g->wk->msg.assign((size_t)g->n, std::string());
    for (int m = 1; m < g->n; ++m) g->wk->th.emplace_back(worker_main, g, m);
    return BMX_OK;
}

static void workers_stop(bmx_group* g)
{
    if (!g->wk) return;
    { std::lock_guard<std::mutex> lk(g->wk->mu); g->wk->stop = true; }
    g->wk->cv_go.notify_all();
    for (auto& t : g->wk->th) t.join();
    delete g->wk; g->wk = nullptr;
}

// run fn(m) for every member: member 0 on the calling thread, the others on their persistent workers (the
// single-device entry points are synchronous); returns the first non-zero status and carries that member's error
// text over to the caller's thread
static int for_each_member(bmx_group* g, const std::function<int(int)>& fn)
{
    auto guarded = [&](int m) -> int {
        try { return fn(m); }
        catch (const std::bad_alloc&) { bmx_set_last_error("out of host memory in a group member"); return BMX_ERR_BADALLOC; }
        catch (...) { bmx_set_last_error("unexpected exception in a group member"); return BMX_ERR_DEVICE; }
    };
    if (g->n == 1 || !g->wk) {
        for (int m = 0; m < g->n; ++m) { int rc = guarded(m); if (rc) return rc; }
        return BMX_OK;
    }
    bmx_workers* w = g->wk;
    // one driver at a time: the workers have ONE job slot (a second host thread calling into the same group waits here;
    // include/bmx.h asks for one driving thread per group anyway)
    std::lock_guard<std::mutex> call(w->call_mu);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->fn = &fn; w->pending = g->n - 1; ++w->gen;
    }
    w->cv_go.notify_all();
    int rc0 = guarded(0);                                        // (never unwinds: the workers still hold &fn until pending drains below)
    std::string msg0 = rc0 ? bmx_last_error() : "";
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv_done.wait(lk, [&] { return w->pending == 0; });
        w->fn = nullptr;
    }
    if (rc0) { bmx_set_last_error(msg0.c_str()); return rc0; }
    for (int m = 1; m < g->n; ++m)
        if (w->rc[(size_t)m]) { bmx_set_last_error(w->msg[(size_t)m].c_str()); return w->rc[(size_t)m]; }
    return BMX_OK;
}

static int sync_all(bmx_group* g)
{
    int rc = BMX_OK;
    for (int m = 0; m < g->n; ++m) { int r = bmx_ctx_synchronize(g->ctx[(size_t)m]); if (r && !rc) rc = r; }
    return rc;
}

static int load_rccl(bmx_group* g, const int* devices)
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) { g->rccl = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (g->rccl) break; }
    if (!g->rccl) { bmx_set_last_error("BMX_GROUP_RCCL: librccl.so could not be loaded"); return BMX_ERR_DEVICE; }
    auto p_init_all = (int (*)(void**, int, const int*))dlsym(g->rccl, "ncclCommInitAll");
    g->p_allreduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g->rccl, "ncclAllReduce");
    g->p_group_start = (int (*)())dlsym(g->rccl, "ncclGroupStart");
    g->p_group_end = (int (*)())dlsym(g->rccl, "ncclGroupEnd");
    g->p_comm_destroy = (int (*)(void*))dlsym(g->rccl, "ncclCommDestroy");
    g->p_errstr = (const char* (*)(int))dlsym(g->rccl, "ncclGetErrorString");
    g->p_comm_count = (int (*)(void*, int*))dlsym(g->rccl, "ncclCommCount");
    if (!p_init_all || !g->p_allreduce || !g->p_group_start || !g->p_group_end || !g->p_comm_destroy) {
        bmx_set_last_error("BMX_GROUP_RCCL: librccl.so lacks the expected entry points"); return BMX_ERR_DEVICE;
    }
    g->comm.assign((size_t)g->n, nullptr);
    // ncclCommInitAll can hang on a node whose fabric is not up (it has never run with > 1 rank in this repo's CI): run it on a
    // helper thread and give up with a clear message after BMX_RCCL_INIT_TIMEOUT_S seconds (default 120) instead of blocking
    // the caller for ever; the helper is left behind in that case (the communicators are not used)
    int timeout_s = 120;
    if (const char* e = getenv("BMX_RCCL_INIT_TIMEOUT_S")) { int t = atoi(e); if (t > 0) timeout_s = t; }
    struct InitState { std::mutex mu; std::condition_variable cv; bool done = false; int r = 0; std::vector<void*> comm; std::vector<int> devs; };
    auto st = std::make_shared<InitState>();
    st->comm.assign((size_t)g->n, nullptr); st->devs.assign(devices, devices + g->n);
    const int nn = g->n;
    std::thread([st, p_init_all, nn] {
        int r = p_init_all(st->comm.data(), nn, st->devs.data());
        std::lock_guard<std::mutex> lk(st->mu); st->r = r; st->done = true; st->cv.notify_all();
    }).detach();
    int r;
    {
        std::unique_lock<std::mutex> lk(st->mu);
        if (!st->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return st->done; })) {
            char buf[200];
            snprintf(buf, sizeof(buf), "ncclCommInitAll over %d devices did not return within %d s (BMX_RCCL_INIT_TIMEOUT_S): use BMX_GROUP_HOST_SUM", nn, timeout_s);
            bmx_set_last_error(buf); g->comm.clear(); return BMX_ERR_DEVICE;
        }
        r = st->r; g->comm = st->comm;
    }
    if (r != 0) {
        std::string m = "ncclCommInitAll failed: "; m += g->p_errstr ? g->p_errstr(r) : "?";
        bmx_set_last_error(m.c_str()); g->comm.clear(); return BMX_ERR_DEVICE;
    }
    return BMX_OK;
}

extern "C" {

int bmx_group_create(const int* devices, int n, int flags, bmx_group** out)
{ ABI_TRY
    ARGCHK(out && devices && n >= 1 && n <= 64 && (flags & ~BMX_GROUP_RCCL) == 0);
    *out = nullptr;
    if (flags & BMX_GROUP_RCCL)
        for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b)
            if (devices[a] == devices[b]) { bmx_set_last_error("BMX_GROUP_RCCL needs distinct devices"); return BMX_ERR_BADARG; }
    bmx_group* g = new (std::nothrow) bmx_group();
    if (!g) return BMX_ERR_BADALLOC;
    g->n = n; g->flags = flags;
    for (int m = 0; m < n; ++m) {
        bmx_ctx* c = nullptr;
        int rc = bmx_ctx_create(devices[m], nullptr, &c);
        if (rc) { bmx_group_destroy(g); return rc; }
        g->ctx.push_back(c);
    }
    if (flags & BMX_GROUP_RCCL) { int rc = load_rccl(g, devices); if (rc) { bmx_group_destroy(g); return rc; } }
    { int rc = workers_start(g); if (rc) { bmx_group_destroy(g); return rc; } }
    *out = g;
    return BMX_OK;
ABI_END }

int bmx_group_destroy(bmx_group* g)
{ ABI_TRY
    if (!g) return BMX_OK;
    workers_stop(g);
    for (size_t m = 0; m < g->comm.size(); ++m) if (g->comm[m] && g->p_comm_destroy) (void)g->p_comm_destroy(g->comm[m]);
    for (bmx_ctx* c : g->ctx) bmx_ctx_destroy(c);
    // librccl stays loaded: unloading a library that registered HIP fat binaries is not safe
    delete g;
    return BMX_OK;
ABI_END }

int bmx_group_size(const bmx_group* g, int* n) { ABI_TRY ARGCHK(g && n); *n = g->n; return BMX_OK; ABI_END }

int bmx_group_ctx(const bmx_group* g, int member, bmx_ctx** ctx)
{ ABI_TRY
    ARGCHK(g && ctx);
    if (member < 0 || member >= g->n) { bmx_set_last_error("member index out of range"); return BMX_ERR_RANGE; }
    *ctx = g->ctx[(size_t)member];
    return BMX_OK;
ABI_END }

int bmx_group_shard_range(const bmx_group* g, uint32_t nblocks, int member, uint32_t* nb_from, uint32_t* nb_to)
{ ABI_TRY
    ARGCHK(g && nb_from && nb_to);
    if (member < 0 || member >= g->n) { bmx_set_last_error("member index out of range"); return BMX_ERR_RANGE; }
    part_ref p = part_for(const_cast<bmx_group*>(g), nblocks);
    *nb_from = p->bounds[(size_t)member]; *nb_to = p->bounds[(size_t)member + 1];
    return BMX_OK;
ABI_END }

// ---- byte-weighted shard borders ("weighted by non-NULL operand bytes") ----
static bool part_in_use(const bmx_group* g, uint32_t nblocks)
{
    auto it = g->parts.find(nblocks);
    return it != g->parts.end() && it->second.use_count() > 1;       // a live vector / pipeline holds a reference
}

int bmx_group_set_partition(bmx_group* g, uint32_t nblocks, const uint32_t* bounds)
{ ABI_TRY
    ARGCHK(g && bounds);
    if (bounds[0] != 0 || bounds[g->n] != nblocks) { bmx_set_last_error("partition must start at 0 and end at nblocks"); return BMX_ERR_RANGE; }
    for (int m = 0; m < g->n; ++m)
        if (bounds[m] > bounds[m + 1]) { bmx_set_last_error("partition borders must not decrease"); return BMX_ERR_RANGE; }
    auto it = g->parts.find(nblocks);
    if (it != g->parts.end() && std::equal(it->second->bounds.begin(), it->second->bounds.end(), bounds)) return BMX_OK;
    if (part_in_use(g, nblocks)) {
        bmx_set_last_error("vectors of this length are already sharded with other borders: free them first");
        return BMX_ERR_BADARG;
    }
    auto p = std::make_shared<bmx_partition>();
    p->nblocks = nblocks;
    p->bounds.assign(bounds, bounds + g->n + 1);
    g->parts[nblocks] = p;
    return BMX_OK;
ABI_END }

int bmx_group_partition_by_weight(bmx_group* g, uint32_t nblocks, const uint64_t* weight, uint32_t* bounds_out)
{ ABI_TRY
    ARGCHK(g && (nblocks == 0 || weight));
    // border m = the first block at which the running weight reaches m/n of the total: every member gets the same
    // share of operand BYTES (not of block columns); empty stretches (NULL top-level ranges, src/bmblocks.h:556-564)
    // cost nothing and therefore do not count
    unsigned __int128 total = 0;
    for (uint32_t i = 0; i < nblocks; ++i) total += weight[i];
    std::vector<uint32_t> b((size_t)g->n + 1, 0);
    b[(size_t)g->n] = nblocks;
    if (total == 0) { for (int m = 0; m < g->n; ++m) { uint32_t lo, hi; equal_range(nblocks, m, g->n, &lo, &hi); b[(size_t)m] = lo; } }
    else {
        unsigned __int128 run = 0; uint32_t i = 0;
        for (int m = 1; m < g->n; ++m) {
            unsigned __int128 want = total * (unsigned)m / (unsigned)g->n;
            // advance while adding block i keeps the running weight closer to (or below) the target
            while (i < nblocks && run + weight[i] / 2 < want) { run += weight[i]; ++i; }
            b[(size_t)m] = i;
        }
    }
    int rc = bmx_group_set_partition(g, nblocks, b.data());
    if (rc) return rc;
    if (bounds_out) memcpy(bounds_out, b.data(), ((size_t)g->n + 1) * sizeof(uint32_t));
    return BMX_OK;
ABI_END }

int bmx_block_table_weights(uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
                            const uint16_t* gap_slab, uint64_t gap_words, uint64_t* weight)
{ ABI_TRY
    ARGCHK(weight && (nblocks == 0 || (kinds && offs)));
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        if (kinds[nb] == BMX_BIT) weight[nb] += 8192u;
        else if (kinds[nb] == BMX_GAP) {
            if (!gap_slab || offs[nb] >= gap_words) { bmx_set_last_error("GAP offset out of range"); return BMX_ERR_RANGE; }
            weight[nb] += 2u * ((uint64_t)(gap_slab[offs[nb]] >> 3) + 1u);
        } else if (kinds[nb] > BMX_GAP) { bmx_set_last_error("bad block kind"); return BMX_ERR_BADARG; }
    }
    return BMX_OK;
ABI_END }

int bmx_group_rccl_ranks(const bmx_group* g, int* n)
{ ABI_TRY
    ARGCHK(g && n);
    *n = 0;
    if (!(g->flags & BMX_GROUP_RCCL) || g->comm.empty() || !g->p_comm_count) return BMX_OK;
    int c = 0;
    int r = g->p_comm_count(g->comm[0], &c);
    if (r != 0) { bmx_set_last_error("ncclCommCount failed"); return BMX_ERR_DEVICE; }
    *n = c;
    return BMX_OK;
ABI_END }

static bmx_gvec* gvec_new(bmx_group* g, uint64_t nbits, uint32_t nblocks)
{
    bmx_gvec* v = new (std::nothrow) bmx_gvec();
    if (!v) return nullptr;
    v->g = g; v->nbits = nbits; v->nblocks = nblocks;
    v->part = part_for(g, nblocks);
    v->shard.assign((size_t)g->n, nullptr);
    return v;
}

int bmx_gvec_free(bmx_group* g, bmx_gvec* v)
{ ABI_TRY
    if (!v) return BMX_OK;
    ARGCHK(g && v->g == g);
    int rc = BMX_OK;
    for (int m = 0; m < g->n; ++m) { int r = bmx_vec_free(g->ctx[(size_t)m], v->shard[(size_t)m]); if (r && !rc) rc = r; }
    delete v;
    return rc;
ABI_END }

static uint64_t shard_bits(uint64_t nbits, uint32_t lo, uint32_t hi)
{
    uint64_t b0 = (uint64_t)lo * BMX_BLOCK_BITS, b1 = std::min<uint64_t>(nbits, (uint64_t)hi * BMX_BLOCK_BITS);
    return b1 > b0 ? b1 - b0 : 0;
}

int bmx_gvec_upload(bmx_group* g, uint64_t nbits, uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
                    const uint32_t* bit_slab, uint32_t n_bit_blocks, const uint16_t* gap_slab, uint64_t gap_words,
                    bmx_gvec** out)
{ ABI_TRY
    ARGCHK(g && out && (nblocks == 0 || (kinds && offs)));
    *out = nullptr;
    bmx_gvec* v = gvec_new(g, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    int rc = for_each_member(g, [&](int m) -> int {
        uint32_t lo, hi; shard_of(v, m, &lo, &hi);
        // the piece of each slab this shard references: [min offset, max end) -- exact for tables in block
        // order (what a tree walk or a frozen arena yields), still correct for any other order
        uint32_t bmin = 0xFFFFFFFFu, bmax = 0; uint64_t gmin = ~0ull, gmax = 0;
        for (uint32_t nb = lo; nb < hi; ++nb) {
            if (kinds[nb] == BMX_BIT) { bmin = std::min(bmin, offs[nb]); bmax = std::max(bmax, offs[nb] + 1u); }
            else if (kinds[nb] == BMX_GAP) {
                uint64_t o = offs[nb];
                if (!gap_slab || o >= gap_words) { bmx_set_last_error("GAP offset out of range"); return BMX_ERR_RANGE; }
                gmin = std::min(gmin, o); gmax = std::max(gmax, o + (uint64_t)(gap_slab[o] >> 3) + 1u);
            } else if (kinds[nb] > BMX_GAP) { bmx_set_last_error("bad block kind"); return BMX_ERR_BADARG; }
        }
        if (bmax > n_bit_blocks) { bmx_set_last_error("bit-block offset out of range"); return BMX_ERR_RANGE; }
        if (gmax > gap_words) { bmx_set_last_error("malformed GAP block (length)"); return BMX_ERR_RANGE; }
        if (bmin == 0xFFFFFFFFu) { bmin = 0; bmax = 0; }
        if (gmin == ~0ull) { gmin = 0; gmax = 0; }
        std::vector<uint32_t> so(std::max<uint32_t>(hi - lo, 1u), 0);
        for (uint32_t nb = lo; nb < hi; ++nb)
            so[nb - lo] = kinds[nb] == BMX_BIT ? offs[nb] - bmin : (kinds[nb] == BMX_GAP ? (uint32_t)(offs[nb] - gmin) : 0u);
        return bmx_vec_upload(g->ctx[(size_t)m], shard_bits(nbits, lo, hi), hi - lo, kinds + lo, so.data(),
                              bmax > bmin ? bit_slab + (size_t)bmin * BMX_BLOCK_WORDS : nullptr, bmax - bmin,
                              gmax > gmin ? gap_slab + gmin : nullptr, gmax - gmin, &v->shard[(size_t)m]);
    });
    if (rc) { std::string keep = bmx_last_error(); bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    *out = v;
    return BMX_OK;
ABI_END }

int bmx_gvec_generate(bmx_group* g, uint64_t seed, uint32_t vec_id, int with_common, uint32_t density_q16,
                      uint64_t nbits, int optimize, bmx_gvec** out)
{ ABI_TRY
    ARGCHK(g && out);
    *out = nullptr;
    uint64_t nblocks64 = (nbits + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks64 > 65536ull * 16) { bmx_set_last_error("vector too long"); return BMX_ERR_RANGE; }
    bmx_gvec* v = gvec_new(g, nbits, (uint32_t)nblocks64);
    if (!v) return BMX_ERR_BADALLOC;
    int rc = for_each_member(g, [&](int m) -> int {
        uint32_t lo, hi; shard_of(v, m, &lo, &hi);
        return bmx_vec_generate_shard(g->ctx[(size_t)m], seed, vec_id, with_common, density_q16, nbits, lo, hi, optimize,
                                      &v->shard[(size_t)m]);
    });
    if (rc) { std::string keep = bmx_last_error(); bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    *out = v;
    return BMX_OK;
ABI_END }

int bmx_gvec_info(const bmx_gvec* v, uint64_t* nbits, uint32_t* nblocks, uint32_t counts[4],
                  uint32_t* bit_slab_blocks, uint64_t* gap_words)
{ ABI_TRY
    ARGCHK(v);
    uint32_t c[4] = {0, 0, 0, 0}; uint32_t slab = 0; uint64_t gw = 0;
    for (bmx_vec* s : v->shard) {
        uint32_t sc[4], sb; uint64_t sg;
        int rc = bmx_vec_info(s, nullptr, nullptr, sc, &sb, &sg); if (rc) return rc;
        for (int k = 0; k < 4; ++k) c[k] += sc[k];
        slab += sb; gw += sg;
    }
    if (nbits) *nbits = v->nbits;
    if (nblocks) *nblocks = v->nblocks;
    if (counts) memcpy(counts, c, sizeof(c));
    if (bit_slab_blocks) *bit_slab_blocks = slab;
    if (gap_words) *gap_words = gw;
    return BMX_OK;
ABI_END }

int bmx_gvec_shard(const bmx_gvec* v, int member, const bmx_vec** shard)
{ ABI_TRY
    ARGCHK(v && shard);
    if (member < 0 || member >= (int)v->shard.size()) { bmx_set_last_error("member index out of range"); return BMX_ERR_RANGE; }
    *shard = v->shard[(size_t)member];
    return BMX_OK;
ABI_END }

int bmx_gvec_download(bmx_group* g, const bmx_gvec* v, uint8_t* kinds, uint32_t* offs, uint32_t* bit_slab, uint16_t* gap_slab)
{ ABI_TRY
    ARGCHK(g && v && v->g == g);
    uint32_t bbase = 0; uint64_t gbase = 0;
    for (int m = 0; m < g->n; ++m) {
        uint32_t lo, hi; shard_of(v, m, &lo, &hi);
        const bmx_vec* s = v->shard[(size_t)m];
        uint32_t sb; uint64_t sg;
        int rc = bmx_vec_info(s, nullptr, nullptr, nullptr, &sb, &sg); if (rc) return rc;
        std::vector<uint8_t> kk;
        uint8_t* kdst = kinds ? kinds + lo : nullptr;
        if (offs && !kinds) { kk.resize(std::max<uint32_t>(hi - lo, 1u)); kdst = kk.data(); }
        rc = bmx_vec_download(g->ctx[(size_t)m], s, kdst, offs ? offs + lo : nullptr,
                              bit_slab ? bit_slab + (size_t)bbase * BMX_BLOCK_WORDS : nullptr,
                              gap_slab ? gap_slab + gbase : nullptr);
        if (rc) return rc;
        if (offs) for (uint32_t nb = lo; nb < hi; ++nb) {
            if (kdst[nb - lo] == BMX_BIT) offs[nb] += bbase;
            else if (kdst[nb - lo] == BMX_GAP) {
                if (gbase + offs[nb] > 0xFFFFFFFFull) { bmx_set_last_error("gathered GAP slab exceeds 32-bit word offsets"); return BMX_ERR_RANGE; }
                offs[nb] += (uint32_t)gbase;
            }
        }
        bbase += sb; gbase += sg;
    }
    return BMX_OK;
ABI_END }

int bmx_gvec_count(bmx_group* g, const bmx_gvec* a, uint64_t* count)
{ ABI_TRY
    ARGCHK(g && a && count && a->g == g);
    for (int m = 0; m < g->n; ++m) { int rc = bmx_i_count_async(g->ctx[(size_t)m], a->shard[(size_t)m], 0); if (rc) { (void)sync_all(g); return rc; } }
    int rc = sync_all(g); if (rc) return rc;
    uint64_t t = 0;
    for (int m = 0; m < g->n; ++m) t += g->ctx[(size_t)m]->h_small[0];
    *count = t;
    return BMX_OK;
ABI_END }

int bmx_gvec_count_op2(bmx_group* g, int op, const bmx_gvec* a, const bmx_gvec* b, uint64_t* count)
{ ABI_TRY
    ARGCHK(g && a && b && count && a->g == g && b->g == g);
    if (a->nblocks != b->nblocks || a->part != b->part) { bmx_set_last_error("sharded operands must cover the same block range (upload them with the same nblocks)"); return BMX_ERR_BADARG; }
    for (int m = 0; m < g->n; ++m) {
        int rc = bmx_i_count_op2_async(g->ctx[(size_t)m], op, a->shard[(size_t)m], b->shard[(size_t)m], 0);
        if (rc) { (void)sync_all(g); return rc; }
    }
    int rc = sync_all(g); if (rc) return rc;
    uint64_t t = 0;
    for (int m = 0; m < g->n; ++m) t += g->ctx[(size_t)m]->h_small[0];
    *count = t;
    return BMX_OK;
ABI_END }

int bmx_gvec_op2(bmx_group* g, int op, const bmx_gvec* a, const bmx_gvec* b, int opt_compress, bmx_gvec** result)
{ ABI_TRY
    ARGCHK(g && a && b && result && a->g == g && b->g == g);
    *result = nullptr;
    if (a->nblocks != b->nblocks || a->part != b->part) { bmx_set_last_error("sharded operands must cover the same block range (upload them with the same nblocks)"); return BMX_ERR_BADARG; }
    bmx_gvec* v = gvec_new(g, std::max(a->nbits, b->nbits), a->nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    int rc = for_each_member(g, [&](int m) -> int {
        return bmx_op2(g->ctx[(size_t)m], op, a->shard[(size_t)m], b->shard[(size_t)m], opt_compress, &v->shard[(size_t)m]);
    });
    if (rc) { std::string keep = bmx_last_error(); bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    *result = v;
    return BMX_OK;
ABI_END }

// ---- rank / select over a sharded vector (per-shard index, one exchange of the shard totals,
// queries routed to the shard that owns the block) ----
int bmx_grs_free(bmx_group* g, bmx_grs* rs)
{ ABI_TRY
    if (!rs) return BMX_OK;
    ARGCHK(g && rs->g == g);
    int rc = BMX_OK;
    for (int m = 0; m < g->n; ++m) { int r = bmx_rs_free(g->ctx[(size_t)m], rs->rs[(size_t)m]); if (r && !rc) rc = r; }
    delete rs;
    return rc;
ABI_END }

int bmx_grs_build(bmx_group* g, const bmx_gvec* v, bmx_grs** out)
{ ABI_TRY
    ARGCHK(g && v && out && v->g == g);
    *out = nullptr;
    bmx_grs* rs = new (std::nothrow) bmx_grs();
    if (!rs) return BMX_ERR_BADALLOC;
    rs->g = g; rs->v = v;
    rs->rs.assign((size_t)g->n, nullptr);
    int rc = for_each_member(g, [&](int m) -> int { return bmx_rs_build(g->ctx[(size_t)m], v->shard[(size_t)m], &rs->rs[(size_t)m]); });
    rs->before.assign((size_t)g->n + 1, 0);
    for (int m = 0; m < g->n && !rc; ++m) {                     // the exchange: n x 8 bytes, exclusive scan on the host
        uint64_t c = 0;
        rc = bmx_rs_count(rs->rs[(size_t)m], &c);
        rs->before[(size_t)m + 1] = rs->before[(size_t)m] + c;
    }
    if (rc) { std::string keep = bmx_last_error(); bmx_grs_free(g, rs); bmx_set_last_error(keep.c_str()); return rc; }
    *out = rs;
    return BMX_OK;
ABI_END }

int bmx_grs_count(const bmx_grs* rs, uint64_t* count)
{ ABI_TRY
    ARGCHK(rs && count);
    *count = rs->before.back();
    return BMX_OK;
ABI_END }

// queries are split by owner on the host, every member answers its share (one batch call per member, in parallel),
// answers are put back in query order
int bmx_grank_batch(bmx_group* g, const bmx_gvec* v, const bmx_grs* rs, const uint64_t* n, size_t q, uint64_t* out)
{ ABI_TRY
    ARGCHK(g && v && rs && v->g == g && rs->g == g && rs->v == v && (q == 0 || (n && out)));
    std::vector<uint32_t> lo((size_t)g->n + 1, 0);
    for (int m = 0; m < g->n; ++m) { uint32_t a, b; shard_of(v, m, &a, &b); lo[(size_t)m] = a; lo[(size_t)m + 1] = b; }
    std::vector<std::vector<uint64_t>> qs((size_t)g->n), ans((size_t)g->n);
    std::vector<std::vector<size_t>> at((size_t)g->n);
    for (size_t i = 0; i < q; ++i) {
        uint64_t nb = n[i] >> 16;
        if (nb >= v->nblocks) { out[i] = rs->before.back(); continue; }            // past the end: the total (src/bm.h:3133)
        size_t m = (size_t)(std::upper_bound(lo.begin(), lo.end(), (uint32_t)nb) - lo.begin()) - 1;
        qs[m].push_back(n[i] - (uint64_t)lo[m] * BMX_BLOCK_BITS);
        at[m].push_back(i);
    }
    int rc = for_each_member(g, [&](int m) -> int {
        size_t k = qs[(size_t)m].size();
        if (!k) return BMX_OK;
        ans[(size_t)m].resize(k);
        return bmx_rank_batch(g->ctx[(size_t)m], v->shard[(size_t)m], rs->rs[(size_t)m], qs[(size_t)m].data(), k, ans[(size_t)m].data());
    });
    if (rc) return rc;
    for (int m = 0; m < g->n; ++m)
        for (size_t k = 0; k < at[(size_t)m].size(); ++k) out[at[(size_t)m][k]] = rs->before[(size_t)m] + ans[(size_t)m][k];
    return BMX_OK;
ABI_END }

int bmx_gselect_batch(bmx_group* g, const bmx_gvec* v, const bmx_grs* rs, const uint64_t* rank, size_t q,
                      uint64_t* pos, uint8_t* found)
{ ABI_TRY
    ARGCHK(g && v && rs && v->g == g && rs->g == g && rs->v == v && (q == 0 || (rank && pos && found)));
    std::vector<uint32_t> lo((size_t)g->n, 0);
    for (int m = 0; m < g->n; ++m) { uint32_t a, b; shard_of(v, m, &a, &b); lo[(size_t)m] = a; }
    std::vector<std::vector<uint64_t>> qs((size_t)g->n), ans((size_t)g->n);
    std::vector<std::vector<uint8_t>> fnd((size_t)g->n);
    std::vector<std::vector<size_t>> at((size_t)g->n);
    const uint64_t total = rs->before.back();
    for (size_t i = 0; i < q; ++i) {
        pos[i] = 0; found[i] = 0;
        if (!rank[i] || rank[i] > total) continue;                                  // rank is 1-based (src/bm.h:5350)
        // the shard that holds the rank-th one: before[m] < rank <= before[m + 1]
        size_t m = (size_t)(std::lower_bound(rs->before.begin(), rs->before.end(), rank[i]) - rs->before.begin()) - 1;
        qs[m].push_back(rank[i] - rs->before[m]);
        at[m].push_back(i);
    }
    int rc = for_each_member(g, [&](int m) -> int {
        size_t k = qs[(size_t)m].size();
        if (!k) return BMX_OK;
        ans[(size_t)m].resize(k); fnd[(size_t)m].resize(k);
        return bmx_select_batch(g->ctx[(size_t)m], v->shard[(size_t)m], rs->rs[(size_t)m], qs[(size_t)m].data(), k,
                                ans[(size_t)m].data(), fnd[(size_t)m].data());
    });
    if (rc) return rc;
    for (int m = 0; m < g->n; ++m)
        for (size_t k = 0; k < at[(size_t)m].size(); ++k) {
            size_t i = at[(size_t)m][k];
            found[i] = fnd[(size_t)m][k];
            pos[i] = found[i] ? ans[(size_t)m][k] + (uint64_t)lo[(size_t)m] * BMX_BLOCK_BITS : 0;
        }
    return BMX_OK;
ABI_END }

static int same_range(bmx_group* g, const bmx_gvec* const* src, size_t n, uint32_t* nblocks, uint64_t* nbits)
{
    for (size_t i = 0; i < n; ++i) {
        if (!src[i] || src[i]->g != g) { bmx_set_last_error("operand is null or belongs to another group"); return BMX_ERR_BADARG; }
        if (*nblocks == 0xFFFFFFFFu) *nblocks = src[i]->nblocks;
        else if (src[i]->nblocks != *nblocks || src[i]->part != part_for(g, *nblocks)) { bmx_set_last_error("sharded operands must cover the same block range (upload them with the same nblocks)"); return BMX_ERR_BADARG; }
        *nbits = std::max(*nbits, src[i]->nbits);
    }
    return BMX_OK;
}

// bmx_collection_prepare over shards: every member transposes ITS block range of the vectors (the collection of member m
// serves the aggregations member m runs: bmx_gagg_or / bmx_gagg_and_sub / bmx_gpipeline_* dispatch per member context)
int bmx_gcollection_prepare(bmx_group* g, const bmx_gvec* const* vecs, size_t n, int role)
{ ABI_TRY
    ARGCHK(g && vecs && n >= 1);
    uint32_t nblocks = 0xFFFFFFFFu; uint64_t nbits = 0;
    int rc = same_range(g, vecs, n, &nblocks, &nbits); if (rc) return rc;
    return for_each_member(g, [&](int m) -> int {
        std::vector<const bmx_vec*> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = vecs[i]->shard[(size_t)m];
        // a shard without a single GAP block (a member whose block range is empty or all NULL) has nothing to transpose
        bool any_gap = false;
        for (size_t i = 0; i < n && !any_gap; ++i) { uint32_t c[4] = {0, 0, 0, 0}; (void)bmx_vec_info(h[i], nullptr, nullptr, c, nullptr, nullptr); any_gap = c[BMX_GAP] != 0; }
        if (!any_gap) return BMX_OK;
        return bmx_collection_prepare(g->ctx[(size_t)m], h.data(), n, role);
    });
ABI_END }

int bmx_gagg_or(bmx_group* g, const bmx_gvec* const* src, size_t n, int opt_compress, bmx_gvec** result)
{ ABI_TRY
    ARGCHK(g && result && (n == 0 || src));
    *result = nullptr;
    uint32_t nblocks = 0xFFFFFFFFu; uint64_t nbits = 0;
    int rc = same_range(g, src, n, &nblocks, &nbits); if (rc) return rc;
    if (nblocks == 0xFFFFFFFFu) nblocks = 0;
    bmx_gvec* v = gvec_new(g, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    rc = for_each_member(g, [&](int m) -> int {
        std::vector<const bmx_vec*> h(std::max<size_t>(n, 1));
        for (size_t i = 0; i < n; ++i) h[i] = src[i]->shard[(size_t)m];
        return bmx_agg_or_opt(g->ctx[(size_t)m], h.data(), n, opt_compress, &v->shard[(size_t)m]);
    });
    if (rc) { std::string keep = bmx_last_error(); bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    *result = v;
    return BMX_OK;
ABI_END }

int bmx_gagg_and_sub(bmx_group* g, const bmx_gvec* const* src_and, size_t n_and,
                     const bmx_gvec* const* src_sub, size_t n_sub, bmx_gvec** result, int* any)
{ ABI_TRY
    ARGCHK(g && result && (n_and == 0 || src_and) && (n_sub == 0 || src_sub));
    *result = nullptr;
    if (any) *any = 0;
    uint32_t nblocks = 0xFFFFFFFFu; uint64_t nbits = 0;
    int rc = same_range(g, src_and, n_and, &nblocks, &nbits); if (rc) return rc;
    rc = same_range(g, src_sub, n_sub, &nblocks, &nbits); if (rc) return rc;
    if (nblocks == 0xFFFFFFFFu) nblocks = 0;
    bmx_gvec* v = gvec_new(g, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    std::vector<int> many((size_t)g->n, 0);
    rc = for_each_member(g, [&](int m) -> int {
        std::vector<const bmx_vec*> a(std::max<size_t>(n_and, 1)), s(std::max<size_t>(n_sub, 1));
        for (size_t i = 0; i < n_and; ++i) a[i] = src_and[i]->shard[(size_t)m];
        for (size_t i = 0; i < n_sub; ++i) s[i] = src_sub[i]->shard[(size_t)m];
        return bmx_agg_and_sub(g->ctx[(size_t)m], a.data(), n_and, s.data(), n_sub, &v->shard[(size_t)m], &many[(size_t)m]);
    });
    if (rc) { std::string keep = bmx_last_error(); bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    if (any) for (int m = 0; m < g->n; ++m) *any |= many[(size_t)m];
    *result = v;
    return BMX_OK;
ABI_END }

// aggregator::find_first_and_sub over sharded vectors (src/bmaggregator.h:1458): every member searches its own shard
// (ascending launch windows inside), the answer is the hit of the LOWEST member that found one
int bmx_gfind_first_and_sub(bmx_group* g, const bmx_gvec* const* src_and, size_t n_and,
                            const bmx_gvec* const* src_sub, size_t n_sub, int* found, uint64_t* idx)
{ ABI_TRY
    ARGCHK(g && found && idx && (n_and == 0 || src_and) && (n_sub == 0 || src_sub));
    *found = 0; *idx = 0;
    if (!n_and) return BMX_OK;
    uint32_t nblocks = 0xFFFFFFFFu; uint64_t nbits = 0;
    int rc = same_range(g, src_and, n_and, &nblocks, &nbits); if (rc) return rc;
    rc = same_range(g, src_sub, n_sub, &nblocks, &nbits); if (rc) return rc;
    std::vector<int> f((size_t)g->n, 0);
    std::vector<uint64_t> pos((size_t)g->n, 0);
    rc = for_each_member(g, [&](int m) -> int {
        std::vector<const bmx_vec*> a(std::max<size_t>(n_and, 1)), s(std::max<size_t>(n_sub, 1));
        for (size_t i = 0; i < n_and; ++i) a[i] = src_and[i]->shard[(size_t)m];
        for (size_t i = 0; i < n_sub; ++i) s[i] = src_sub[i]->shard[(size_t)m];
        return bmx_find_first_and_sub(g->ctx[(size_t)m], a.data(), n_and, s.data(), n_sub, &f[(size_t)m], &pos[(size_t)m]);
    });
    if (rc) return rc;
    for (int m = 0; m < g->n; ++m)
        if (f[(size_t)m]) {
            uint32_t lo = src_and[0]->part->bounds[(size_t)m];
            *found = 1; *idx = (uint64_t)lo * BMX_BLOCK_BITS + pos[(size_t)m];
            break;
        }
    return BMX_OK;
ABI_END }

// sparse_vector_scanner range search over SHARDED bit-planes (bmx_slice_compare per member over its rows): the planes
// (and the not-NULL vector) must cover the same block range and `size` must end in its last block; the result is sharded
// like the planes, the count is the sum over the members
int bmx_gslice_compare(bmx_group* g, const bmx_gvec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                       uint64_t size, const bmx_gvec* not_null, bmx_gvec** result, uint64_t* count)
{ ABI_TRY
    ARGCHK(g && (nslices == 0 || slices) && nslices <= 64 && (result || count));
    if (result) *result = nullptr;
    if (count) *count = 0;
    uint32_t nblocks = 0xFFFFFFFFu;
    for (size_t i = 0; i < nslices; ++i) {
        if (!slices[i]) continue;
        if (slices[i]->g != g) { bmx_set_last_error("slice belongs to another group"); return BMX_ERR_BADARG; }
        if (nblocks == 0xFFFFFFFFu) nblocks = slices[i]->nblocks;
        else if (nblocks != slices[i]->nblocks) { bmx_set_last_error("sharded planes must cover the same block range (upload them with the same nblocks)"); return BMX_ERR_BADARG; }
    }
    uint64_t need = (size + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks == 0xFFFFFFFFu) nblocks = (uint32_t)need;
    if (need != nblocks || (not_null && (not_null->g != g || not_null->nblocks != nblocks))) {
        bmx_set_last_error("size / not-NULL vector must span the block range of the planes"); return BMX_ERR_BADARG;
    }
    bmx_gvec* v = result ? gvec_new(g, size, nblocks) : nullptr;
    if (result && !v) return BMX_ERR_BADALLOC;
    part_ref pr = part_for(g, nblocks);
    std::vector<uint64_t> cnt((size_t)g->n, 0);
    int rc = for_each_member(g, [&](int m) -> int {
        uint32_t lo = pr->bounds[(size_t)m], hi = pr->bounds[(size_t)m + 1];
        std::vector<const bmx_vec*> sl(std::max<size_t>(nslices, 1), nullptr);
        for (size_t i = 0; i < nslices; ++i) sl[i] = slices[i] ? slices[i]->shard[(size_t)m] : nullptr;
        return bmx_slice_compare(g->ctx[(size_t)m], sl.data(), nslices, pred, v0, v1, shard_bits(size, lo, hi),
                                 not_null ? not_null->shard[(size_t)m] : nullptr, v ? &v->shard[(size_t)m] : nullptr,
                                 count ? &cnt[(size_t)m] : nullptr);
    });
    if (rc) { std::string keep = bmx_last_error(); if (v) bmx_gvec_free(g, v); bmx_set_last_error(keep.c_str()); return rc; }
    if (count) for (int m = 0; m < g->n; ++m) *count += cnt[(size_t)m];
    if (result) *result = v;
    return BMX_OK;
ABI_END }

// bmx_slice_eq_counts over sharded bit-planes: every member counts over its own rows, the counts are summed
int bmx_gslice_eq_counts(bmx_group* g, const bmx_gvec* const* slices, size_t nslices, const uint64_t* values, size_t n,
                         uint64_t size, const bmx_gvec* not_null, uint64_t* counts)
{ ABI_TRY
    ARGCHK(g && (nslices == 0 || slices) && (n == 0 || (values && counts)));
    uint32_t nblocks = 0xFFFFFFFFu;
    for (size_t i = 0; i < nslices; ++i) {
        if (!slices[i]) continue;
        if (slices[i]->g != g) { bmx_set_last_error("slice belongs to another group"); return BMX_ERR_BADARG; }
        if (nblocks == 0xFFFFFFFFu) nblocks = slices[i]->nblocks;
        else if (nblocks != slices[i]->nblocks) { bmx_set_last_error("sharded planes must cover the same block range (upload them with the same nblocks)"); return BMX_ERR_BADARG; }
    }
    uint64_t need = (size + BMX_BLOCK_BITS - 1) / BMX_BLOCK_BITS;
    if (nblocks == 0xFFFFFFFFu) nblocks = (uint32_t)need;
    if (need != nblocks || (not_null && (not_null->g != g || not_null->nblocks != nblocks))) {
        bmx_set_last_error("size / not-NULL vector must span the block range of the planes"); return BMX_ERR_BADARG;
    }
    for (size_t q = 0; q < n; ++q) counts[q] = 0;
    part_ref pr = part_for(g, nblocks);
    std::vector<std::vector<uint64_t>> part((size_t)g->n, std::vector<uint64_t>(std::max<size_t>(n, 1), 0));
    int rc = for_each_member(g, [&](int m) -> int {
        uint32_t lo = pr->bounds[(size_t)m], hi = pr->bounds[(size_t)m + 1];
        std::vector<const bmx_vec*> sl(std::max<size_t>(nslices, 1), nullptr);
        for (size_t i = 0; i < nslices; ++i) sl[i] = slices[i] ? slices[i]->shard[(size_t)m] : nullptr;
        return bmx_slice_eq_counts(g->ctx[(size_t)m], sl.data(), nslices, values, n, shard_bits(size, lo, hi),
                                   not_null ? not_null->shard[(size_t)m] : nullptr, part[(size_t)m].data());
    });
    if (rc) return rc;
    for (int m = 0; m < g->n; ++m) for (size_t q = 0; q < n; ++q) counts[q] += part[(size_t)m][q];
    return BMX_OK;
ABI_END }

int bmx_gpipeline_destroy(bmx_group* g, bmx_gpipeline* p)
{ ABI_TRY
    if (!p) return BMX_OK;
    ARGCHK(g && p->g == g);
    for (int m = 0; m < g->n; ++m) {
        bmx_ctx* c = g->ctx[(size_t)m];
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if ((size_t)m < p->pipe.size()) bmx_pipeline_destroy(c, p->pipe[(size_t)m]);
        if ((size_t)m < p->d_counts.size() && p->d_counts[(size_t)m]) (void)hipFree(p->d_counts[(size_t)m]);
        if ((size_t)m < p->ev0.size() && p->ev0[(size_t)m]) (void)hipEventDestroy(p->ev0[(size_t)m]);
        if ((size_t)m < p->ev1.size() && p->ev1[(size_t)m]) (void)hipEventDestroy(p->ev1[(size_t)m]);
        if ((size_t)m < p->ev2.size() && p->ev2[(size_t)m]) (void)hipEventDestroy(p->ev2[(size_t)m]);
    }
    if (p->h_counts) (void)hipHostFree(p->h_counts);
    delete p;
    return BMX_OK;
ABI_END }

int bmx_gpipeline_create(bmx_group* g, const bmx_gvec* const* and_list, const uint32_t* and_n,
                         const bmx_gvec* const* sub_list, const uint32_t* sub_n, size_t ngroups, bmx_gpipeline** out)
{ ABI_TRY
    ARGCHK(g && out && ngroups > 0 && ngroups < (1u << 20) && and_n && sub_n);
    *out = nullptr;
    size_t tot_and = 0, tot_sub = 0;
    for (size_t k = 0; k < ngroups; ++k) { tot_and += and_n[k]; tot_sub += sub_n[k]; }
    ARGCHK(tot_and == 0 || and_list);
    ARGCHK(tot_sub == 0 || sub_list);
    uint32_t nblocks = 0xFFFFFFFFu; uint64_t nbits = 0;
    int rc = same_range(g, and_list, tot_and, &nblocks, &nbits); if (rc) return rc;
    rc = same_range(g, sub_list, tot_sub, &nblocks, &nbits); if (rc) return rc;
    bmx_gpipeline* p = new (std::nothrow) bmx_gpipeline();
    if (!p) return BMX_ERR_BADALLOC;
    p->g = g; p->ngroups = (uint32_t)ngroups;
    p->pipe.assign((size_t)g->n, nullptr); p->d_counts.assign((size_t)g->n, nullptr);
    p->ev0.assign((size_t)g->n, nullptr); p->ev1.assign((size_t)g->n, nullptr); p->ev2.assign((size_t)g->n, nullptr);
    p->last_ms.assign((size_t)g->n, 0.f); p->last_xchg_ms.assign((size_t)g->n, 0.f);
    rc = for_each_member(g, [&](int m) -> int {
        bmx_ctx* c = g->ctx[(size_t)m];
        std::vector<const bmx_vec*> a(std::max<size_t>(tot_and, 1)), s(std::max<size_t>(tot_sub, 1));
        for (size_t i = 0; i < tot_and; ++i) a[i] = and_list[i]->shard[(size_t)m];
        for (size_t i = 0; i < tot_sub; ++i) s[i] = sub_list[i]->shard[(size_t)m];
        int r = bmx_pipeline_create(c, a.data(), and_n, s.data(), sub_n, ngroups, &p->pipe[(size_t)m]);
        if (r) return r;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMalloc((void**)&p->d_counts[(size_t)m], ngroups * 8));
        HIPCHK(hipEventCreate(&p->ev0[(size_t)m]));
        HIPCHK(hipEventCreate(&p->ev1[(size_t)m]));
        HIPCHK(hipEventCreate(&p->ev2[(size_t)m]));
        return BMX_OK;
    });
    if (!rc) {
        hipError_t e = hipHostMalloc((void**)&p->h_counts, (size_t)g->n * ngroups * 8);
        if (e != hipSuccess) rc = bmx_fail_hip(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    if (rc) { std::string keep = bmx_last_error(); bmx_gpipeline_destroy(g, p); bmx_set_last_error(keep.c_str()); return rc; }
    *out = p;
    return BMX_OK;
ABI_END }

// everything of one run that is enqueued: kernels on every member first, then the exchange.  A failure in the middle
// must not leave the other members' work in flight when the caller sees the error: the wrapper below drains every
// stream before it returns (no early return past enqueued work)
static int gpipeline_enqueue(bmx_group* g, bmx_gpipeline* p, bool rccl)
{
    const size_t ng = p->ngroups;
    // 1. the counts kernel on every member (asynchronous: all devices start before any is waited for)
    for (int m = 0; m < g->n; ++m) {
        bmx_ctx* c = g->ctx[(size_t)m];
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipEventRecord(p->ev0[(size_t)m], c->stream));
        int rc = bmx_pipeline_run_counts_dev(c, p->pipe[(size_t)m], 0u, 0xFFFFFFFFu, p->d_counts[(size_t)m]);
        if (rc) return rc;
        HIPCHK(hipEventRecord(p->ev1[(size_t)m], c->stream));
    }
    // 2. the only exchange: the popcounts (8 B per arg-group)
    if (rccl) {
        int r = g->p_group_start();
        for (int m = 0; m < g->n && r == 0; ++m) {
            bmx_ctx* c = g->ctx[(size_t)m];
            HIPCHK(hipSetDevice(c->device));
            r = g->p_allreduce(p->d_counts[(size_t)m], p->d_counts[(size_t)m], ng, 5 /* ncclUint64 */, 0 /* ncclSum */,
                               g->comm[(size_t)m], c->stream);
        }
        int r2 = g->p_group_end();
        if (r || r2) {
            std::string msg = "RCCL all-reduce failed: "; msg += g->p_errstr ? g->p_errstr(r ? r : r2) : "?";
            bmx_set_last_error(msg.c_str()); return BMX_ERR_DEVICE;
        }
        for (int m = 0; m < g->n; ++m) {
            bmx_ctx* c = g->ctx[(size_t)m];
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipEventRecord(p->ev2[(size_t)m], c->stream));
        }
        bmx_ctx* c0 = g->ctx[0];
        HIPCHK(hipSetDevice(c0->device));
        HIPCHK(hipMemcpyAsync(p->h_counts, p->d_counts[0], ng * 8, hipMemcpyDeviceToHost, c0->stream));
    } else {
        for (int m = 0; m < g->n; ++m) {
            bmx_ctx* c = g->ctx[(size_t)m];
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipMemcpyAsync(p->h_counts + (size_t)m * ng, p->d_counts[(size_t)m], ng * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipEventRecord(p->ev2[(size_t)m], c->stream));
        }
    }
    return BMX_OK;
}

// pipeline::set_search_count_limit over shards (src/bmaggregator.h:255,1365).  Every member runs its own windowed search with
// the SAME limit: either some member reaches the limit on its shard alone (then the sum has), or every member returns its
// full shard count (then the sum is the true count) -- sum >= min(limit, true count) and <= true count, the reference's
// contract ("can find more, cannot find less"), with no exchange between the windows.
int bmx_gpipeline_set_search_count_limit(bmx_group* g, bmx_gpipeline* p, uint64_t limit)
{ ABI_TRY
    ARGCHK(g && p && p->g == g);
    for (int m = 0; m < g->n; ++m) {
        int rc = bmx_pipeline_set_search_count_limit(g->ctx[(size_t)m], p->pipe[(size_t)m], limit);
        if (rc) return rc;
    }
    p->search_limit = limit == 0 ? ~0ull : limit;
    return BMX_OK;
ABI_END }

int bmx_gpipeline_run_counts(bmx_group* g, bmx_gpipeline* p, uint64_t* counts_out)
{ ABI_TRY
    ARGCHK(g && p && p->g == g && counts_out);
    const size_t ng = p->ngroups;
    if (p->search_limit != ~0ull) {
        // under a limit the members run their windowed (synchronous) searches side by side on the group's workers; counts are
        // summed on the host (8 bytes per arg-group and member)
        std::vector<uint64_t> part((size_t)g->n * std::max<size_t>(ng, 1), 0);
        int rc = for_each_member(g, [&](int m) -> int {
            return bmx_pipeline_run_counts(g->ctx[(size_t)m], p->pipe[(size_t)m], 0u, 0xFFFFFFFFu, part.data() + (size_t)m * ng);
        });
        if (rc) return rc;
        for (size_t k = 0; k < ng; ++k) { uint64_t t = 0; for (int m = 0; m < g->n; ++m) t += part[(size_t)m * ng + k]; counts_out[k] = t; }
        for (int m = 0; m < g->n; ++m) { p->last_ms[(size_t)m] = 0.f; p->last_xchg_ms[(size_t)m] = 0.f; }
        return BMX_OK;
    }
    const bool rccl = (g->flags & BMX_GROUP_RCCL) && !g->comm.empty();
    int rc = gpipeline_enqueue(g, p, rccl);
    if (rc) { std::string keep = bmx_last_error(); (void)sync_all(g); bmx_set_last_error(keep.c_str()); return rc; }
    rc = sync_all(g); if (rc) return rc;
    for (size_t k = 0; k < ng; ++k) {
        uint64_t t = p->h_counts[k];
        if (!rccl) for (int m = 1; m < g->n; ++m) t += p->h_counts[(size_t)m * ng + k];
        counts_out[k] = t;
    }
    for (int m = 0; m < g->n; ++m) {
        float ms = 0.f, xs = 0.f;
        (void)hipSetDevice(g->ctx[(size_t)m]->device);
        if (hipEventElapsedTime(&ms, p->ev0[(size_t)m], p->ev1[(size_t)m]) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
        if (hipEventElapsedTime(&xs, p->ev1[(size_t)m], p->ev2[(size_t)m]) != hipSuccess) { (void)hipGetLastError(); xs = 0.f; }
        p->last_ms[(size_t)m] = ms; p->last_xchg_ms[(size_t)m] = xs;
    }
    return BMX_OK;
ABI_END }

int bmx_gpipeline_last_ms(bmx_group* g, const bmx_gpipeline* p, float* ms)
{ ABI_TRY
    ARGCHK(g && p && p->g == g && ms);
    for (int m = 0; m < g->n; ++m) ms[m] = p->last_ms[(size_t)m];
    return BMX_OK;
ABI_END }

int bmx_gpipeline_last_exchange_ms(bmx_group* g, const bmx_gpipeline* p, float* ms)
{ ABI_TRY
    ARGCHK(g && p && p->g == g && ms);
    for (int m = 0; m < g->n; ++m) ms[m] = p->last_xchg_ms[(size_t)m];
    return BMX_OK;
ABI_END }

int bmx_gpipeline_operand_bytes(bmx_group* g, bmx_gpipeline* p, uint64_t* bytes_per_member)
{ ABI_TRY
    ARGCHK(g && p && p->g == g && bytes_per_member);
    for (int m = 0; m < g->n; ++m) {
        int rc = bmx_pipeline_operand_bytes(g->ctx[(size_t)m], p->pipe[(size_t)m], 0u, 0xFFFFFFFFu, &bytes_per_member[m]);
        if (rc) return rc;
    }
    return BMX_OK;
ABI_END }

int bmx_gpipeline_describe(bmx_group* g, bmx_gpipeline* p, int member, char* buf, size_t buf_len, uint32_t* n_launches)
{ ABI_TRY
    ARGCHK(g && p && p->g == g && member >= 0 && member < g->n);
    return bmx_pipeline_describe(g->ctx[(size_t)member], p->pipe[(size_t)member], 0u, 0xFFFFFFFFu, buf, buf_len, n_launches);
ABI_END }

} // extern "C"
