// bmx.hip -- C-ABI (include/bmx.h) of the MI355X-native bit-vector engine.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared bmx.hip -o libbmx.so
#include "../../include/bmx.h"
#include "bmx_kernels2.h"
#include "bmx_kernels3.h"

#include <algorithm>
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

static int fail_hip(hipError_t e, const char* what, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed at bmx.hip:%d: %s", what, line, hipGetErrorString(e));
    g_last_error = buf;
    return e == hipErrorOutOfMemory ? BMX_ERR_BADALLOC : BMX_ERR_DEVICE;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail_hip(e_, #call, __LINE__); } while (0)
#define ARGCHK(cond) do { if (!(cond)) { g_last_error = "bad argument: " #cond; return BMX_ERR_BADARG; } } while (0)
#define KCHK() HIPCHK(hipGetLastError())

struct bmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t mem_used = 0;
    // grow-only scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;      // raw block slab for import/generate
    void* aux = nullptr; size_t aux_bytes = 0;              // stats / offsets / totals
    u64* d_small = nullptr;                                 // 64 x u64 result words
    u64* d_slots = nullptr;                                 // COUNT_SLOTS striped count accumulators (kept zero between launches)
    u64* h_small = nullptr;                                 // pinned mirror
    // caching device allocator: results of same-shaped operations re-use their blocks instead of
    // paying hipMalloc / hipFree (which synchronises the device) on every call
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    uint64_t pool_cached = 0, pool_cap = 16ull << 30;
    int pipe_unroll = 4;       // operand blocks per batch (two batches in flight in the v2 kernel)
    int pipe_rows = 8;         // register rows per work item (8 = whole block, 4/2/1 = slices)
    int pipe_nt = 1;           // non-temporal operand loads (+4.5 % on the streamed-once headline case)
    int pipe_wg = 384;         // workgroup size of the bit-only counts kernel: 6 adjacent columns per workgroup, 2 workgroups per CU
                               // (+6 % over 256 in the A/B sweep: co-scheduled waves read one contiguous stretch of each operand)
    int pipe_ver = 2;          // 1 = k_pipe_counts_bits, 2 = software-pipelined k_pipe_counts_bits2
    int pipe_staged = -1;      // LDS-staged many-groups kernel: -1 auto, 0 never, 1 whenever possible
    int pipe_lds = 0;          // experiment: dynamic LDS bytes requested by the bit-only counts kernel (occupancy throttle)
    int pipe_slots = 16;       // plane blocks staged at a time (16: 1024-thread WG; 8: two 512-thread WGs per CU)
    int xcd_swz = 1;
};

struct bmx_vec {
    bmx_ctx* ctx;
    uint64_t nbits; uint32_t nblocks;
    uint32_t counts[4]; uint64_t gap_words; uint32_t n_bit;
    u64* d_desc; uint4* d_bits; u16* d_gaps;
    size_t bytes;
};

struct bmx_pipeline {
    bmx_ctx* ctx;
    uint32_t ngroups, ncols, col_stride, n_ops;
    bool has_gap;
    uint64_t nbits;                       // max size of the operands
    // LDS-staged path (k_pipe_counts_staged): distinct vectors ("planes") + per-group plane masks
    uint32_t nplanes, nchunks; bool staged_ok;
    const u64** d_udesc; u32* d_unblk; u32* d_gmask; u32* d_gskip;
    std::vector<u32>* h_row_off;          // host copy: row offset of each group inside a column record
    std::vector<u32>* h_and_n;            // host copy: AND operands per group
    u64* d_dmat;
    u32* d_meta;       // row_off | and_n | sub_n | and_off | sub_off (ngroups each) | nblocks (n_ops)
    const u64** d_descs;
    size_t bytes;
};

struct bmx_rs {
    bmx_ctx* ctx;
    uint32_t nblocks; uint64_t count;
    u32* d_bcount; u64* d_sub; u64* d_rcount; u16* d_cum;
    u16* d_gidx;                                          // GAP blocks: first run reaching each 1024-bit wave
    u64* d_sample; uint32_t nsamples, sample_shift;       // top level of the select search (<= 2048 entries)
    size_t bytes;
};

static int set_dev(const bmx_ctx* ctx) { HIPCHK(hipSetDevice(ctx->device)); return BMX_OK; }

static size_t pool_round(size_t bytes)
{
    if (bytes < 256) bytes = 256;
    size_t g = bytes >= (2u << 20) ? (2u << 20) : (bytes >= (64u << 10) ? (64u << 10) : 256u);
    return (bytes + g - 1) / g * g;
}

static int dmalloc(bmx_ctx* ctx, void** p, size_t bytes)
{
    *p = nullptr;
    size_t sz = pool_round(bytes);
    auto it = ctx->pool_free.lower_bound(sz);
    if (it != ctx->pool_free.end() && it->first <= sz + sz / 4) {          // best fit within 25 % slack
        *p = it->second; sz = it->first;
        ctx->pool_cached -= sz;
        ctx->pool_free.erase(it);
    } else {
        hipError_t e = hipMalloc(p, sz);
        if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {          // give the cache back and retry
            (void)hipGetLastError();
            for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
            ctx->pool_free.clear(); ctx->pool_cached = 0;
            e = hipMalloc(p, sz);
        }
        if (e != hipSuccess) return fail_hip(e, "hipMalloc", __LINE__);
    }
    ctx->pool_live[*p] = sz;
    ctx->mem_used += sz;
    return BMX_OK;
}

// the caller guarantees no kernel still uses p (handles are freed after a stream synchronise)
static void dfree(bmx_ctx* ctx, void* p)
{
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) { (void)hipFree(p); return; }
    size_t sz = it->second;
    ctx->pool_live.erase(it);
    ctx->mem_used -= std::min<uint64_t>(ctx->mem_used, sz);
    if (ctx->pool_cached + sz <= ctx->pool_cap) { ctx->pool_free.emplace(sz, p); ctx->pool_cached += sz; }
    else (void)hipFree(p);
}

static void pool_trim(bmx_ctx* ctx)
{
    for (auto& kv : ctx->pool_free) (void)hipFree(kv.second);
    ctx->pool_free.clear(); ctx->pool_cached = 0;
}

static int ensure(bmx_ctx* ctx, void** buf, size_t* cur, size_t need)
{
    if (*cur >= need) return BMX_OK;
    if (*buf) { HIPCHK(hipStreamSynchronize(ctx->stream)); (void)hipFree(*buf); *buf = nullptr; *cur = 0; }
    HIPCHK(hipMalloc(buf, need));
    *cur = need;
    return BMX_OK;
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
{
    ARGCHK(n);
    *n = 0;
    HIPCHK(hipGetDeviceCount(n));
    return BMX_OK;
}

int bmx_ctx_create(int device, void* stream, bmx_ctx** out)
{
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
    CTXCHK(hipMalloc((void**)&ctx->d_slots, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64)));
    CTXCHK(hipMemsetAsync(ctx->d_slots, 0, COUNT_SLOTS * COUNT_SLOT_STRIDE * sizeof(u64), ctx->stream));
#undef CTXCHK
    if (const char* e = getenv("BMX_POOL_MAX_MB")) ctx->pool_cap = (uint64_t)atoll(e) << 20;
    if (const char* e = getenv("BMX_PIPE_UNROLL")) ctx->pipe_unroll = atoi(e);
    if (const char* e = getenv("BMX_PIPE_ROWS")) ctx->pipe_rows = atoi(e);
    if (const char* e = getenv("BMX_PIPE_NT")) ctx->pipe_nt = atoi(e);
    if (const char* e = getenv("BMX_PIPE_VER")) ctx->pipe_ver = atoi(e);
    if (const char* e = getenv("BMX_PIPE_WG")) ctx->pipe_wg = atoi(e);
    if (const char* e = getenv("BMX_XCD_SWIZZLE")) ctx->xcd_swz = atoi(e);
    *out = ctx;
    return BMX_OK;
}

int bmx_ctx_destroy(bmx_ctx* ctx)
{
    if (!ctx) return BMX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    pool_trim(ctx);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return BMX_OK;
}

int bmx_ctx_set_tuning(bmx_ctx* ctx, const char* key, int value)
{
    ARGCHK(ctx && key);
    std::string k(key);
    if (k == "pipe_unroll") { ARGCHK(value == 1 || value == 2 || value == 4); ctx->pipe_unroll = value; }
    else if (k == "pipe_rows") { ARGCHK(value == 8 || value == 4 || value == 2 || value == 1); ctx->pipe_rows = value; }
    else if (k == "pipe_nt") ctx->pipe_nt = value != 0;
    else if (k == "pipe_lds") { ARGCHK(value >= 0 && value <= 160 * 1024); ctx->pipe_lds = value; }
    else if (k == "pipe_slots") { ARGCHK(value == 8 || value == 16); ctx->pipe_slots = value; }
    else if (k == "pipe_staged") { ARGCHK(value >= -1 && value <= 1); ctx->pipe_staged = value; }
    else if (k == "pipe_ver") { ARGCHK(value == 1 || value == 2); ctx->pipe_ver = value; }
    else if (k == "pipe_wg") { ARGCHK(value >= 64 && value <= 1024 && value % 64 == 0); ctx->pipe_wg = value; }
    else if (k == "xcd_swizzle") ctx->xcd_swz = value != 0;
    else { g_last_error = "unknown tuning key"; return BMX_ERR_BADARG; }
    return BMX_OK;
}

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

int bmx_ctx_synchronize(bmx_ctx* ctx)
{
    ARGCHK(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
}

int bmx_ctx_trim(bmx_ctx* ctx)
{
    ARGCHK(ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    pool_trim(ctx);
    return BMX_OK;
}

int bmx_ctx_mem_used(const bmx_ctx* ctx, uint64_t* bytes)
{
    ARGCHK(ctx && bytes);
    *bytes = ctx->mem_used;
    return BMX_OK;
}

int bmx_timer_start(bmx_ctx* ctx) { ARGCHK(ctx); HIPCHK(hipEventRecord(ctx->ev0, ctx->stream)); return BMX_OK; }
int bmx_timer_stop_ms(bmx_ctx* ctx, float* ms)
{
    ARGCHK(ctx && ms);
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return BMX_OK;
}

// ---------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------
static bmx_vec* vec_alloc_host(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks)
{
    bmx_vec* v = new (std::nothrow) bmx_vec();
    if (!v) return nullptr;
    memset(v, 0, sizeof(*v));
    v->ctx = ctx; v->nbits = nbits; v->nblocks = nblocks;
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
    v->bytes = std::max<size_t>(b_desc, 16) + std::max<size_t>(b_bits, 16) + std::max<size_t>(b_gaps, 16);
    return BMX_OK;
}

int bmx_vec_free(bmx_ctx* ctx, bmx_vec* v)
{
    if (!v) return BMX_OK;
    ARGCHK(ctx && v->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(ctx, v->d_desc); dfree(ctx, v->d_bits); dfree(ctx, v->d_gaps);
    delete v;
    return BMX_OK;
}

int bmx_vec_upload(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks,
                   const uint8_t* kinds, const uint32_t* offs,
                   const uint32_t* bit_slab, uint32_t n_bit_blocks,
                   const uint16_t* gap_slab, uint64_t gap_words, bmx_vec** out)
{
    ARGCHK(ctx && out && (nblocks == 0 || (kinds && offs)));
    ARGCHK(n_bit_blocks == 0 || bit_slab);
    ARGCHK(gap_words == 0 || gap_slab);
    *out = nullptr;
    int rc = set_dev(ctx); if (rc) return rc;
    // validate + re-pack the GAP slab so that every block starts 16-byte aligned on the device
    std::vector<u16> gpad;
    std::vector<u64> goff(std::max<uint32_t>(nblocks, 1), 0);
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        if (kinds[nb] > BMX_GAP) { g_last_error = "bad block kind"; return BMX_ERR_BADARG; }
        if (kinds[nb] == BMX_BIT && offs[nb] >= n_bit_blocks) { g_last_error = "bit-block offset out of range"; return BMX_ERR_RANGE; }
        if (kinds[nb] != BMX_GAP) continue;
        uint64_t o = offs[nb];
        if (o >= gap_words) { g_last_error = "GAP offset out of range"; return BMX_ERR_RANGE; }
        uint32_t len = gap_slab[o] >> 3;
        if (len == 0 || len > 1280u + 8u || o + len + 1u > gap_words || gap_slab[o + len] != 65535u) { g_last_error = "malformed GAP block"; return BMX_ERR_RANGE; }
        goff[nb] = gpad.size();
        gpad.insert(gpad.end(), gap_slab + o, gap_slab + o + len + 1u);
        gpad.resize((gpad.size() + 7u) & ~(size_t)7u, 0);
    }
    bmx_vec* v = vec_alloc_host(ctx, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    if ((rc = vec_alloc_device(v, n_bit_blocks, gpad.size()))) { bmx_vec_free(ctx, v); return rc; }
    std::vector<u64> desc(std::max<uint32_t>(nblocks, 1), 0);
    for (uint32_t nb = 0; nb < nblocks; ++nb) {
        uint8_t k = kinds[nb];
        v->counts[k]++;
        if (k == BMX_BIT) desc[nb] = DESC_MAKE(v->d_bits + (size_t)offs[nb] * 512u, K_BIT);
        else if (k == BMX_GAP) desc[nb] = DESC_MAKE_GAP(v->d_gaps + goff[nb], gpad[goff[nb]] >> 3, gpad[goff[nb]] & 1u);
        else desc[nb] = DESC_MAKE(0, k);
    }
    HIPCHK(hipMemcpyAsync(v->d_desc, desc.data(), (size_t)nblocks * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_bit_blocks) HIPCHK(hipMemcpyAsync(v->d_bits, bit_slab, (size_t)n_bit_blocks * 8192, hipMemcpyHostToDevice, ctx->stream));
    if (!gpad.empty()) HIPCHK(hipMemcpyAsync(v->d_gaps, gpad.data(), gpad.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = v;
    return BMX_OK;
}

// raw block slab (in ctx->scratch, nblocks x 8 KiB) -> classified / compressed vector
static int vec_from_raw(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks, int optimize, bmx_vec** out)
{
    int rc;
    size_t aux_need = (size_t)nblocks * (sizeof(BlockStat) + 4) + 64;
    if ((rc = ensure(ctx, &ctx->aux, &ctx->aux_bytes, aux_need))) return rc;
    BlockStat* st = (BlockStat*)ctx->aux;
    u32* offs = (u32*)((char*)ctx->aux + (size_t)nblocks * sizeof(BlockStat));
    const uint4* raw = (const uint4*)ctx->scratch;
    bmx_vec* v = vec_alloc_host(ctx, nbits, nblocks);
    if (!v) return BMX_ERR_BADALLOC;
    if (nblocks) {
        hipLaunchKernelGGL(k_block_stats, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, raw, nblocks, optimize, st);
        KCHK();
        hipLaunchKernelGGL(k_scan_layout, dim3(1), dim3(1024), 0, ctx->stream, st, nblocks, offs, ctx->d_small);
        KCHK();
        HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 6 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    } else memset(ctx->h_small, 0, 6 * sizeof(u64));
    uint32_t n_bit = (uint32_t)ctx->h_small[0]; uint64_t gap_words = ctx->h_small[1];
    for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
    if ((rc = vec_alloc_device(v, n_bit, gap_words))) { bmx_vec_free(ctx, v); return rc; }
    if (nblocks) {
        hipLaunchKernelGGL(k_emit_blocks, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           raw, nblocks, st, offs, v->d_bits, v->d_gaps, v->d_desc);
        KCHK();
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    *out = v;
    return BMX_OK;
}

int bmx_vec_import_bits(bmx_ctx* ctx, const uint32_t* words, uint64_t nwords, int optimize, bmx_vec** out)
{
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
}

int bmx_vec_generate_shard(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                           uint32_t density_q16, uint64_t nbits, uint32_t nb_from, uint32_t nb_to,
                           int optimize, bmx_vec** out)
{
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
}

int bmx_vec_generate(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                     uint32_t density_q16, uint64_t nbits, int optimize, bmx_vec** out)
{
    return bmx_vec_generate_shard(ctx, seed, vec_id, with_common, density_q16, nbits, 0u, 0xFFFFFFFFu, optimize, out);
}

int bmx_vec_info(const bmx_vec* v, uint64_t* nbits, uint32_t* nblocks, uint32_t counts[4],
                 uint32_t* bit_slab_blocks, uint64_t* gap_words)
{
    ARGCHK(v);
    if (nbits) *nbits = v->nbits;
    if (nblocks) *nblocks = v->nblocks;
    if (counts) memcpy(counts, v->counts, sizeof(v->counts));
    if (bit_slab_blocks) *bit_slab_blocks = v->n_bit;
    if (gap_words) *gap_words = v->gap_words;
    return BMX_OK;
}

int bmx_vec_download(bmx_ctx* ctx, const bmx_vec* v, uint8_t* kinds, uint32_t* offs,
                     uint32_t* bit_slab, uint16_t* gap_slab)
{
    ARGCHK(ctx && v && v->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    if (kinds || offs) {
        std::vector<u64> desc(std::max<uint32_t>(v->nblocks, 1));
        HIPCHK(hipMemcpyAsync(desc.data(), v->d_desc, (size_t)v->nblocks * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (uint32_t nb = 0; nb < v->nblocks; ++nb) {
            u32 k = DESC_K(desc[nb]);
            if (kinds) kinds[nb] = (uint8_t)k;
            if (offs) {
                if (k == K_BIT) offs[nb] = (uint32_t)((DESC_P(desc[nb]) - (u64)(uintptr_t)v->d_bits) / 8192u);
                else if (k == K_GAP) offs[nb] = (uint32_t)((DESC_P(desc[nb]) - (u64)(uintptr_t)v->d_gaps) / 2u);
                else offs[nb] = 0;
            }
        }
    }
    if (bit_slab && v->n_bit) HIPCHK(hipMemcpyAsync(bit_slab, v->d_bits, (size_t)v->n_bit * 8192, hipMemcpyDeviceToHost, ctx->stream));
    if (gap_slab && v->gap_words) HIPCHK(hipMemcpyAsync(gap_slab, v->d_gaps, (size_t)v->gap_words * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
}

int bmx_vec_to_words(bmx_ctx* ctx, const bmx_vec* v, uint32_t* words, uint64_t nwords)
{
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
}

int bmx_count(bmx_ctx* ctx, const bmx_vec* a, uint64_t* count)
{
    ARGCHK(ctx && a && count && a->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    if (a->nblocks) {
        hipLaunchKernelGGL(k_vec_count, dim3((a->nblocks + 3) / 4), dim3(256), 0, ctx->stream, a->d_desc, a->nblocks, ctx->d_slots);
        KCHK();
    }
    hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(64), 0, ctx->stream, ctx->d_slots, ctx->d_small);
    KCHK();
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count = ctx->h_small[0];
    return BMX_OK;
}

// ---------------------------------------------------------------------------
// pipeline
// ---------------------------------------------------------------------------
int bmx_pipeline_create(bmx_ctx* ctx, const bmx_vec* const* and_list, const uint32_t* and_n,
                        const bmx_vec* const* sub_list, const uint32_t* sub_n,
                        size_t ngroups, bmx_pipeline** out)
{
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
    uint32_t ncols = 0, col_stride = 0; bool has_gap = false; uint64_t max_bits = 0;
    size_t ia = 0, is = 0;
    for (size_t g = 0; g < ngroups; ++g) {
        row_off[g] = col_stride; col_stride += 2 + and_n[g] + sub_n[g];
        m_and_n[g] = and_n[g]; m_sub_n[g] = sub_n[g];
        and_off[g] = (u32)ia; sub_off[g] = (u32)(tot_and + is);
        for (uint32_t k = 0; k < and_n[g]; ++k, ++ia) {
            const bmx_vec* v = and_list[ia];
            if (!v || v->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
            descs[ia] = v->d_desc; nblk[ia] = v->nblocks; ncols = std::max(ncols, v->nblocks); has_gap |= v->counts[BMX_GAP] != 0;
            max_bits = std::max(max_bits, v->nbits);
        }
        for (uint32_t k = 0; k < sub_n[g]; ++k, ++is) {
            const bmx_vec* v = sub_list[is];
            if (!v || v->ctx != ctx) { g_last_error = "operand is null or belongs to another context"; return BMX_ERR_BADARG; }
            descs[tot_and + is] = v->d_desc; nblk[tot_and + is] = v->nblocks; ncols = std::max(ncols, v->nblocks); has_gap |= v->counts[BMX_GAP] != 0;
            max_bits = std::max(max_bits, v->nbits);
        }
    }
#define PIPECHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { int r_ = fail_hip(e_, #call, __LINE__); bmx_pipeline_destroy(ctx, p); return r_; } } while (0)
    bmx_pipeline* p = new (std::nothrow) bmx_pipeline();
    if (!p) return BMX_ERR_BADALLOC;
    memset(p, 0, sizeof(*p));
    p->ctx = ctx; p->ngroups = (uint32_t)ngroups; p->ncols = ncols; p->col_stride = col_stride; p->n_ops = (uint32_t)n_ops; p->has_gap = has_gap;
    p->nbits = max_bits;
    p->h_row_off = new std::vector<u32>(row_off, row_off + ngroups);
    p->h_and_n = new std::vector<u32>(m_and_n, m_and_n + ngroups);
    // distinct vectors of the pipeline (pipeline::unique_vectors(), src/bmaggregator.h:301) and the
    // (AND | SUB << 16) plane masks of every group, 16 planes per chunk
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
        // plane tables are only built where the staged kernel can be chosen (>= 32 groups, or forced by the knob):
        // a single-group combine_and_sub should not pay four uploads and a sync for them
        p->staged_ok = p->nplanes > 0 && (size_t)ngroups * p->nchunks < (1u << 28) && (ngroups >= 32 || ctx->pipe_staged == 1);
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
    PIPECHK(hipMemcpyAsync(p->d_meta, meta.data(), b_meta, hipMemcpyHostToDevice, ctx->stream));
    PIPECHK(hipMemcpyAsync(p->d_descs, descs.data(), b_descs, hipMemcpyHostToDevice, ctx->stream));
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
    }
    PIPECHK(hipStreamSynchronize(ctx->stream));
    *out = p;
    return BMX_OK;
#undef PIPECHK
}

int bmx_pipeline_destroy(bmx_ctx* ctx, bmx_pipeline* p)
{
    if (!p) return BMX_OK;
    ARGCHK(ctx && p->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(ctx, p->d_dmat); dfree(ctx, p->d_meta); dfree(ctx, (void*)p->d_descs);
    dfree(ctx, (void*)p->d_udesc); dfree(ctx, p->d_unblk); dfree(ctx, p->d_gmask); dfree(ctx, p->d_gskip);
    delete p->h_row_off; delete p->h_and_n;
    delete p;
    return BMX_OK;
}

static int pipe_range(const bmx_pipeline* p, uint32_t& nb_from, uint32_t& nb_to)
{
    if (nb_to > p->ncols) nb_to = p->ncols;
    if (nb_from > nb_to) { g_last_error = "nb_from > nb_to"; return BMX_ERR_RANGE; }
    return BMX_OK;
}

int bmx_pipeline_run_counts_dev(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* d_counts)
{
    ARGCHK(ctx && p && p->ctx == ctx && d_counts);
    int rc = set_dev(ctx); if (rc) return rc;
    if ((rc = pipe_range(p, nb_from, nb_to))) return rc;
    HIPCHK(hipMemsetAsync(d_counts, 0, (size_t)p->ngroups * 8, ctx->stream));
    u64 nitems64 = (u64)(nb_to - nb_from) * p->ngroups;
    if (!nitems64) return BMX_OK;
    const u32* row_off = p->d_meta; const u32* and_n = p->d_meta + p->ngroups; const u32* sub_n = p->d_meta + 2 * p->ngroups;
    {
        // many groups over few distinct vectors: every plane block is re-used >= 8 times per column
        bool reuse = p->ngroups >= 32 && (uint64_t)p->n_ops >= 8ull * p->nplanes;
        bool use_staged = p->staged_ok && (ctx->pipe_staged == 1 || (ctx->pipe_staged < 0 && reuse));
        if (use_staged) {
            size_t lds = (size_t)ctx->pipe_slots * 8192;
#define LAUNCH_STG(S) do { \
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pipe_counts_staged<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_staged<S>), dim3(nb_to - nb_from), dim3(S * 64), lds, ctx->stream, \
                               (const u64* const*)p->d_udesc, (const u32*)p->d_unblk, p->nplanes, (const u32*)p->d_gmask, (const u32*)p->d_gskip, \
                               p->ngroups, nb_from, nb_to - nb_from, ctx->xcd_swz, (u64*)d_counts); } while (0)
            if (ctx->pipe_slots == 8) LAUNCH_STG(8); else LAUNCH_STG(16);
#undef LAUNCH_STG
            KCHK();
            return BMX_OK;
        }
    }
    if (!p->has_gap) {
        // bit-block-only fast path: (column, group, slice) items
        u32 rows = (u32)ctx->pipe_rows, parts = 8u / rows;
        u64 n64 = nitems64 * parts;
        if (n64 > 0xFFFFFFF0ull) { g_last_error = "too many work items in one run"; return BMX_ERR_RANGE; }
        if (ctx->pipe_ver == 2) {
            u32 nitems = (u32)nitems64, wpb = (u32)ctx->pipe_wg / 64u, grid = (nitems + wpb - 1) / wpb;
#define LAUNCH_B2(U, NT, WG) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_bits2<U, NT, WG>), dim3(grid), dim3(ctx->pipe_wg), (size_t)ctx->pipe_lds, ctx->stream, \
        p->d_dmat, row_off, and_n, p->col_stride, p->ngroups, nb_from, nitems, ctx->xcd_swz, (u64*)d_counts)
            if (ctx->pipe_wg == 1024) { if (ctx->pipe_nt) LAUNCH_B2(2, true, 1024); else LAUNCH_B2(2, false, 1024); }   // 128 VGPRs max
            else if (ctx->pipe_wg == 768) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 768); else LAUNCH_B2(4, false, 768); }
            else if (ctx->pipe_wg == 640) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 640); else LAUNCH_B2(4, false, 640); }
            else if (ctx->pipe_wg == 576) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 576); else LAUNCH_B2(4, false, 576); }
            else if (ctx->pipe_wg == 448) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 448); else LAUNCH_B2(4, false, 448); }
            else if (ctx->pipe_wg == 384) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 384); else LAUNCH_B2(4, false, 384); }
            else if (ctx->pipe_wg == 320) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 320); else LAUNCH_B2(4, false, 320); }
            else if (ctx->pipe_wg == 192) { if (ctx->pipe_nt) LAUNCH_B2(4, true, 192); else LAUNCH_B2(4, false, 192); }
            else if (ctx->pipe_wg == 512) {
                if (ctx->pipe_nt) { switch (ctx->pipe_unroll) { case 1: LAUNCH_B2(1, true, 512); break; case 4: LAUNCH_B2(4, true, 512); break; default: LAUNCH_B2(2, true, 512); break; } }
                else { switch (ctx->pipe_unroll) { case 1: LAUNCH_B2(1, false, 512); break; case 4: LAUNCH_B2(4, false, 512); break; default: LAUNCH_B2(2, false, 512); break; } }
            }
            else if (ctx->pipe_wg > 256) { g_last_error = "unsupported pipe_wg"; return BMX_ERR_BADARG; }
            else if (ctx->pipe_nt) { switch (ctx->pipe_unroll) { case 1: LAUNCH_B2(1, true, 256); break; case 4: LAUNCH_B2(4, true, 256); break; default: LAUNCH_B2(2, true, 256); break; } }
            else { switch (ctx->pipe_unroll) { case 1: LAUNCH_B2(1, false, 256); break; case 4: LAUNCH_B2(4, false, 256); break; default: LAUNCH_B2(2, false, 256); break; } }
#undef LAUNCH_B2
            KCHK();
            return BMX_OK;
        }
        u32 wg1 = ctx->pipe_wg > 256 ? 256u : (u32)ctx->pipe_wg;          // v1 kernels are compiled for <= 256 threads
        u32 nitems = (u32)n64, wpb = wg1 / 64u, grid = (nitems + wpb - 1) / wpb;
#define LAUNCH_BITS(U, R, NT) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts_bits<U, R, NT>), dim3(grid), dim3(wg1), 0, ctx->stream, \
        p->d_dmat, row_off, and_n, p->col_stride, p->ngroups, nb_from, nitems, ctx->xcd_swz, (u64*)d_counts)
#define LAUNCH_BITS_R(U, NT) switch (rows) { case 1: LAUNCH_BITS(U, 1, NT); break; case 2: LAUNCH_BITS(U, 2, NT); break; \
        case 4: LAUNCH_BITS(U, 4, NT); break; default: LAUNCH_BITS(U, 8, NT); break; }
#define LAUNCH_BITS_U(NT) switch (ctx->pipe_unroll) { case 1: LAUNCH_BITS_R(1, NT); break; case 4: LAUNCH_BITS_R(4, NT); break; \
        default: LAUNCH_BITS_R(2, NT); break; }
        if (ctx->pipe_nt) { LAUNCH_BITS_U(true); } else { LAUNCH_BITS_U(false); }
#undef LAUNCH_BITS_U
#undef LAUNCH_BITS_R
#undef LAUNCH_BITS
        KCHK();
        return BMX_OK;
    }
    if (nitems64 > 0xFFFFFFF0ull) { g_last_error = "too many (column, group) items in one run"; return BMX_ERR_RANGE; }
    u32 nitems = (u32)nitems64;
    u32 grid = (nitems + 3) / 4;
    size_t lds = 4 * 2048 * 4;
#define LAUNCH_PIPE(U) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pipe_counts<U>), dim3(grid), dim3(256), lds, ctx->stream, \
        p->d_dmat, row_off, and_n, sub_n, p->col_stride, p->ngroups, nb_from, nitems, ctx->xcd_swz, (u64*)d_counts)
    switch (ctx->pipe_unroll) {
    case 1: LAUNCH_PIPE(1); break;
    case 4: LAUNCH_PIPE(4); break;
    default: LAUNCH_PIPE(2); break;
    }
#undef LAUNCH_PIPE
    KCHK();
    return BMX_OK;
}

int bmx_pipeline_run_counts(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* counts_out)
{
    ARGCHK(ctx && p && p->ctx == ctx && counts_out);
    int rc = set_dev(ctx); if (rc) return rc;
    u64* d_counts = nullptr;
    size_t bytes = (size_t)p->ngroups * 8;
    if (p->ngroups <= 64) d_counts = ctx->d_small;
    else if ((rc = dmalloc(ctx, (void**)&d_counts, bytes))) return rc;
    rc = bmx_pipeline_run_counts_dev(ctx, p, nb_from, nb_to, d_counts);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(counts_out, d_counts, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail_hip(e, "counts readback", __LINE__);
    }
    if (p->ngroups > 64) dfree(ctx, d_counts);
    return rc;
}

int bmx_pipeline_operand_bytes(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, uint64_t* bytes)
{
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
}

} // extern "C"

// ---------------------------------------------------------------------------
// materialised results: full-size slab (one slot per block column), see
// store_result() in bmx_kernels2.h
// ---------------------------------------------------------------------------
static int result_begin(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks, bmx_vec** out, BlockStat** st, u32** offs)
{
    int rc;
    size_t aux_need = (size_t)nblocks * (sizeof(BlockStat) + 4) + 64;
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

static int result_finish(bmx_ctx* ctx, bmx_vec* v, BlockStat* st, u32* offs)
{
    int rc;
    uint32_t nblocks = v->nblocks;
    if (!nblocks) return BMX_OK;
    hipLaunchKernelGGL(k_scan_layout, dim3(1), dim3(1024), 0, ctx->stream, st, nblocks, offs, ctx->d_small);
    KCHK();
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 6 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t gap_words = ctx->h_small[1];
    for (int k = 0; k < 4; ++k) v->counts[k] = (uint32_t)ctx->h_small[2 + k];
    if (gap_words) {
        size_t b_gaps = (size_t)gap_words * 2 + 64;      // + guard, see vec_alloc_device
        if ((rc = dmalloc(ctx, (void**)&v->d_gaps, b_gaps))) return rc;
        v->bytes += std::max<size_t>(b_gaps, 16);
        v->gap_words = gap_words;
        hipLaunchKernelGGL(k_emit_gaps, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           v->d_bits, nblocks, st, offs, v->d_gaps, v->d_desc);
        KCHK();
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    if (v->counts[BMX_BIT] == 0) {            // nothing lives in the slab: give it back
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
    if (e == hipSuccess && a->nblocks) {
        hipLaunchKernelGGL(k_rebase_desc, dim3((a->nblocks + 255) / 256), dim3(256), 0, ctx->stream, a->d_desc, v->d_desc, a->nblocks,
                           (u64)(uintptr_t)a->d_bits, (u64)(uintptr_t)v->d_bits, (u64)(uintptr_t)a->d_gaps, (u64)(uintptr_t)v->d_gaps);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { bmx_vec_free(ctx, v); return fail_hip(e, "vec_clone", __LINE__); }
    *out = v;
    return BMX_OK;
}

int bmx_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int opt_compress, bmx_vec** result)
{
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
    if (nblocks) {
        hipLaunchKernelGGL(k_op2, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, op,
                           a->d_desc, a->nblocks, b->d_desc, b->nblocks, nblocks, opt_compress,
                           v->d_bits, v->d_desc, st);
        KCHK();
    }
    if ((rc = result_finish(ctx, v, st, offs))) { bmx_vec_free(ctx, v); return rc; }
    *result = v;
    return BMX_OK;
}

int bmx_count_op2_dev(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* d_count)
{
    ARGCHK(ctx && a && b && d_count && a->ctx == ctx && b->ctx == ctx && op >= BMX_AND && op <= BMX_SUB);
    int rc = set_dev(ctx); if (rc) return rc;
    uint32_t nblocks = std::max(a->nblocks, b->nblocks);
    if (nblocks) {
        hipLaunchKernelGGL(k_count_op2, dim3((nblocks + 3) / 4), dim3(256), 0, ctx->stream, op,
                           a->d_desc, a->nblocks, b->d_desc, b->nblocks, nblocks, ctx->d_slots);
        KCHK();
    }
    hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(64), 0, ctx->stream, ctx->d_slots, (u64*)d_count);
    KCHK();
    return BMX_OK;
}

int bmx_count_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* count)
{
    ARGCHK(ctx && count);
    int rc = bmx_count_op2_dev(ctx, op, a, b, ctx->d_small);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *count = ctx->h_small[0];
    return BMX_OK;
}

static int agg_or_impl(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result);

int bmx_agg_or(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, bmx_vec** result)
{
    return agg_or_impl(ctx, src, n, 0 /* opt_mode_ = opt_none, src/bmaggregator.h:917 */, result);
}

int bmx_agg_or_opt(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result)
{
    return agg_or_impl(ctx, src, n, opt_compress ? 1 : 0, result);
}

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
    if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) return rc;      // empty list => cleared target (:1105)
    if (n >= 64 && ncols && has_gap && !has_bit) {
        // many GAP-only operands: column-tile kernel straight from the descriptor tables (no sort pass)
        void* d_descs = nullptr; void* d_nblk = nullptr;
        if ((rc = dmalloc(ctx, &d_descs, n * 8)) || (rc = dmalloc(ctx, &d_nblk, n * 4))) { dfree(ctx, d_descs); bmx_vec_free(ctx, v); return rc; }
        hipError_t e = hipMemcpyAsync(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_nblk, nblk.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        size_t lds = (size_t)OR_TILE * 8192 + OR_TILE * 4;
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_agg_or_gap_tiled), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_agg_or_gap_tiled, dim3((ncols + OR_TILE - 1) / OR_TILE), dim3(1024), lds, ctx->stream,
                               (const u64* const*)d_descs, (const u32*)d_nblk, (u32)n, ncols, opt_compress, v->d_bits, v->d_desc, st);
            e = hipGetLastError();
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

int bmx_agg_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                    const bmx_vec* const* src_sub, size_t n_sub, bmx_vec** result, int* any)
{
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
    uint32_t an = (uint32_t)n_and, sn = (uint32_t)n_sub;
    bmx_pipeline* p = nullptr;
    if ((rc = bmx_pipeline_create(ctx, src_and, &an, src_sub, &sn, 1, &p))) return rc;
    if ((rc = result_begin(ctx, nbits, ncols, &v, &st, &offs))) { bmx_pipeline_destroy(ctx, p); return rc; }
    if (ncols) {
        size_t lds = p->has_gap ? 4 * 2048 * 4 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_and_sub<2>), dim3((ncols + 3) / 4), dim3(256), lds, ctx->stream,
                           p->d_dmat, p->d_meta + 1, p->d_meta + 2, p->col_stride, ncols,
                           1 /* combine_and_sub always stores with opt_compress, :1210 */, ctx->xcd_swz,
                           v->d_bits, v->d_desc, st);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = fail_hip(e, "k_agg_and_sub", __LINE__);
    }
    if (!rc) rc = result_finish(ctx, v, st, offs);
    bmx_pipeline_destroy(ctx, p);
    if (rc) { bmx_vec_free(ctx, v); return rc; }
    if (any) *any = (v->counts[BMX_FULL] + v->counts[BMX_BIT] + v->counts[BMX_GAP]) != 0;
    *result = v;
    return BMX_OK;
}

// combine_and_sub(pipe) with result vectors / counts / OR target (src/bmaggregator.h:1292-1449)
int bmx_pipeline_run_results(bmx_ctx* ctx, bmx_pipeline* p, bmx_vec** results_out, uint64_t* counts_out,
                             const bmx_vec* or_target_in, bmx_vec** or_target_out)
{
    ARGCHK(ctx && p && p->ctx == ctx && (results_out || or_target_out));
    ARGCHK(!or_target_in || or_target_in->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    std::vector<bmx_vec*> res(p->ngroups, nullptr);
    auto cleanup = [&]() { for (bmx_vec* r : res) if (r) bmx_vec_free(ctx, r); };
    for (uint32_t g = 0; g < p->ngroups; ++g) {
        if (counts_out) counts_out[g] = 0;
        if (!(*p->h_and_n)[g] || !p->ncols) continue;                       // empty AND group: skipped (:1352)
        bmx_vec* v; BlockStat* st; u32* offs;
        if ((rc = result_begin(ctx, p->nbits, p->ncols, &v, &st, &offs))) { cleanup(); return rc; }
        size_t lds = p->has_gap ? 4 * 2048 * 4 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_agg_and_sub<2>), dim3((p->ncols + 3) / 4), dim3(256), lds, ctx->stream,
                           p->d_dmat + (*p->h_row_off)[g], p->d_meta + p->ngroups + g, p->d_meta + 2 * p->ngroups + g,
                           p->col_stride, p->ncols, 1 /* opt_compress, :1421 */, ctx->xcd_swz, v->d_bits, v->d_desc, st);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { bmx_vec_free(ctx, v); cleanup(); return fail_hip(e, "k_agg_and_sub", __LINE__); }
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

int bmx_find_first_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                           const bmx_vec* const* src_sub, size_t n_sub, int* found, uint64_t* idx)
{
    ARGCHK(ctx && found && idx && (n_and == 0 || src_and) && (n_sub == 0 || src_sub));
    *found = 0; *idx = 0;
    int rc = set_dev(ctx); if (rc) return rc;
    if (!n_and) return BMX_OK;
    uint32_t an = (uint32_t)n_and, sn = (uint32_t)n_sub;
    bmx_pipeline* p = nullptr;
    if ((rc = bmx_pipeline_create(ctx, src_and, &an, src_sub, &sn, 1, &p))) return rc;
    hipError_t e = hipMemsetAsync(ctx->d_small, 0xFF, 8, ctx->stream);
    if (e == hipSuccess && p->ncols) {
        size_t lds = p->has_gap ? 4 * 2048 * 4 : 0;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_find_first_and_sub<2>), dim3((p->ncols + 3) / 4), dim3(256), lds, ctx->stream,
                           p->d_dmat, p->d_meta + 1, p->d_meta + 2, p->col_stride, p->ncols, ctx->d_small);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    bmx_pipeline_destroy(ctx, p);
    if (e != hipSuccess) return fail_hip(e, "bmx_find_first_and_sub", __LINE__);
    if (ctx->h_small[0] != ~0ull) { *found = 1; *idx = ctx->h_small[0]; }
    return BMX_OK;
}

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
{
    ARGCHK(result);
    return shift_right_and_impl(ctx, src, n, opt_compress, any, result, found, nullptr);
}

int bmx_agg_shift_right_and_count(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, uint64_t* count)
{
    ARGCHK(count);
    return shift_right_and_impl(ctx, src, n, 0, 0, nullptr, nullptr, count);
}

// ---------------------------------------------------------------------------
// rank / select
// ---------------------------------------------------------------------------
int bmx_rs_build(bmx_ctx* ctx, const bmx_vec* v, bmx_rs** out)
{
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
    rs->bytes = b1 + b2 + b3 + b4;
    if (v->nblocks) {
        hipLaunchKernelGGL(k_rs_build, dim3((v->nblocks + 3) / 4), dim3(256), 0, ctx->stream,
                           v->d_desc, v->nblocks, rs->d_bcount, rs->d_sub, rs->d_cum, rs->d_gidx);
        KCHK();
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, ctx->stream, rs->d_bcount, v->nblocks, rs->d_rcount, ctx->d_small);
        KCHK();
        uint32_t shift = 0;
        while (((v->nblocks + (1u << shift) - 1u) >> shift) > 2048u) ++shift;
        rs->sample_shift = shift; rs->nsamples = (v->nblocks + (1u << shift) - 1u) >> shift;
        if ((rc = dmalloc(ctx, (void**)&rs->d_sample, (size_t)rs->nsamples * 8))) { bmx_rs_free(ctx, rs); return rc; }
        hipLaunchKernelGGL(k_rs_sample, dim3((rs->nsamples + 255) / 256), dim3(256), 0, ctx->stream,
                           rs->d_rcount, v->nblocks, shift, rs->nsamples, rs->d_sample);
        KCHK();
        HIPCHK(hipMemcpyAsync(ctx->h_small, ctx->d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        rs->count = ctx->h_small[0];
    }
    *out = rs;
    return BMX_OK;
}

int bmx_rs_free(bmx_ctx* ctx, bmx_rs* rs)
{
    if (!rs) return BMX_OK;
    ARGCHK(ctx && rs->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dfree(ctx, rs->d_bcount); dfree(ctx, rs->d_sub); dfree(ctx, rs->d_rcount); dfree(ctx, rs->d_cum); dfree(ctx, rs->d_gidx); dfree(ctx, rs->d_sample);
    delete rs;
    return BMX_OK;
}

int bmx_rs_count(const bmx_rs* rs, uint64_t* count) { ARGCHK(rs && count); *count = rs->count; return BMX_OK; }

int bmx_rs_export(bmx_ctx* ctx, const bmx_rs* rs, uint32_t* bcount, uint64_t* sub_count)
{
    ARGCHK(ctx && rs && rs->ctx == ctx);
    int rc = set_dev(ctx); if (rc) return rc;
    if (bcount && rs->nblocks) HIPCHK(hipMemcpyAsync(bcount, rs->d_bcount, (size_t)rs->nblocks * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (sub_count && rs->nblocks) HIPCHK(hipMemcpyAsync(sub_count, rs->d_sub, (size_t)rs->nblocks * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BMX_OK;
}

static u32 query_grid(size_t q) { return (u32)std::min<size_t>((q * 8 + 255) / 256, 256u * 16u); }

int bmx_rank_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* d_n, size_t q, uint64_t* d_out)
{
    ARGCHK(ctx && v && rs && v->ctx == ctx && rs->ctx == ctx && rs->nblocks == v->nblocks && (q == 0 || (d_n && d_out)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    hipLaunchKernelGGL(k_rank, dim3(query_grid(q)), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks,
                       rs->d_rcount, rs->d_cum, rs->d_gidx, rs->count, (const u64*)d_n, (u64)q, (u64*)d_out);
    KCHK();
    return BMX_OK;
}

int bmx_select_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* d_rank, size_t q,
                         uint64_t* d_pos, uint8_t* d_found)
{
    ARGCHK(ctx && v && rs && v->ctx == ctx && rs->ctx == ctx && rs->nblocks == v->nblocks && (q == 0 || (d_rank && d_pos && d_found)));
    int rc = set_dev(ctx); if (rc) return rc;
    if (!q) return BMX_OK;
    hipLaunchKernelGGL(k_select, dim3(query_grid(q)), dim3(256), 0, ctx->stream, v->d_desc, v->nblocks,
                       rs->d_rcount, rs->d_cum, rs->d_gidx, rs->d_sample, rs->nsamples, rs->sample_shift, rs->count,
                       (const u64*)d_rank, (u64)q, (u64*)d_pos, (u8*)d_found);
    KCHK();
    return BMX_OK;
}

int bmx_rank_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* n, size_t q, uint64_t* out)
{
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
}

int bmx_select_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs, const uint64_t* rank, size_t q,
                     uint64_t* pos, uint8_t* found)
{
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
}

} // extern "C"
