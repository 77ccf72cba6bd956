// bmx_internal.h -- host-side structures shared by the translation units of libbmx.so
// (bmx.hip: single-device engine + kernels; bmx_group.hip: multi-device group layer).
// Not part of the ABI: include/bmx.h is.
#pragma once
#include "../../include/bmx.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <unordered_map>
#include <vector>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// error plumbing: no exception crosses the ABI (lang-maps/libbm conventions)
int bmx_fail_hip(hipError_t e, const char* what, const char* file, int line);
void bmx_set_last_error(const char* msg);
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bmx_fail_hip(e_, #call, __FILE__, __LINE__); } while (0)
#define ARGCHK(cond) do { if (!(cond)) { bmx_set_last_error("bad argument: " #cond); return BMX_ERR_BADARG; } } while (0)
#define KCHK() HIPCHK(hipGetLastError())
#define PEND_SLOTS 1024
// the exception barrier around every extern "C" body: { ABI_TRY ... ABI_END }  (bmx.hip: bmx_abi_caught)
#include <new>
#include <stdexcept>
int bmx_abi_caught(int kind, const char* what);
void bmx_abi_enter();
#define ABI_TRY try { bmx_abi_enter();
#define ABI_END } catch (const std::bad_alloc&) { return bmx_abi_caught(0, nullptr); } catch (const std::exception& e_) { return bmx_abi_caught(1, e_.what()); } catch (...) { return bmx_abi_caught(2, nullptr); }

struct bmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t mem_used = 0;
    // grow-only scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;      // raw block slab for import/generate/upload/download staging
    void* aux = nullptr; size_t aux_bytes = 0;              // stats / offsets / totals
    u64* d_small = nullptr;                                 // 64 x u64 result words
    u64* d_slots = nullptr;                                 // COUNT_SLOTS striped count accumulators (kept zero between launches)
    u64* d_slots2 = nullptr; u32* d_done2 = nullptr;        // a second fold (slots + tickets) for kernels that fold block kinds AND a count
    u64* h_pend = nullptr; uint64_t pend_used[16] = {};    // PEND_SLOTS = 1,024 pinned slots (8 x u64) for the kind counts of unresolved asynchronous results (round 6: 64 before)
    u64* d_cursor = nullptr;                                // bump cursor of kernels that write GAP results themselves (k_op2_loop); zero between launches
    u64* d_zero = nullptr;                                  // 256 bytes of zeros: what an invalid slot of an unconditional load reads
    u32* d_done = nullptr;                                  // workgroup ticket of the in-kernel folds (kept zero between launches)
    u64* h_small = nullptr;                                 // pinned mirror
    char* h_stage = nullptr; size_t stage_off = 0;          // pinned ring for small host -> device tables (h2d_staged)
    hipEvent_t ev_stage = nullptr;
    // caching device allocator: results of same-shaped operations re-use their blocks instead of
    // paying hipMalloc / hipFree (which synchronises the device) on every call
    std::multimap<size_t, void*> pool_free;
    std::unordered_map<void*, size_t> pool_live;
    uint64_t pool_cached = 0, pool_cap = 16ull << 30;
    // debug: red zones (BMX_DEBUG_REDZONE=1 when the context is created).  Every device allocation of the library -- pooled
    // blocks (dmalloc: vectors' slabs, results, indexes, collections, tables) and the grow-only scratch / aux buffers -- gets
    // RZ_BYTES of a canary pattern in front of it and from the end of the REQUESTED bytes (rounded to 16) to the end of the
    // block behind it; verified when the block is freed, at bmx_ctx_synchronize, bmx_debug_redzone_check and context destroy.
    struct RzInfo { void* raw; size_t block; size_t bytes; int line; };
    bool redzone = false;
    std::unordered_map<void*, RzInfo> rz_live;               // user pointer -> what lies around it
    uint64_t rz_hits = 0; std::string rz_report;             // allocations found damaged so far, and where they came from
    long long fail_dmalloc_after = -1;                        // debug fault injection: the allocation this many dmallocs from now fails (bmx_debug_inject_failure kind 4)
    int pipe_unroll = 0;       // operand slices per batch (two batches in flight); 0 = the measured best for the slice size
    int pipe_rows = 0;         // register rows (KiB of a block) per work item: 8 = whole block, 4/2/1 = slices, 0 = auto by item count
    int pipe_nt = 1;           // non-temporal operand loads (+4.5 % on the streamed-once headline case)
    int pipe_wg = 0;           // workgroup size of the bit-only counts kernel (0 = plan default, see pipe_plan in bmx.hip)
    int pipe_window = 0;       // block columns per launch: 0 = one machine-load of waves (pipe_plan), -1 = a single launch, N = explicit
    int pipe_staged = -1;      // LDS-staged many-groups kernel: -1 auto, 0 never, 1 whenever possible
    int pipe_lds = 0;          // experiment: dynamic LDS bytes requested by the bit-only counts kernel (occupancy throttle)
    int pipe_slots = 16;       // plane blocks staged at a time (16: 1024-thread WG; 8: two 512-thread WGs per CU)
    int pipe_split = -1;       // few columns: waves of a workgroup share one column's operand list: -1 auto, 0 never, 1 always
    int or_tile = 0;           // k_agg_or_gap_tiled variant (0 = default)
    int direct_cols = 384;     // aggregation over <= this many block columns and 24..1024 operands: one launch from the descriptor tables (k_direct); 0 = off
    int pair_stream = -1;      // bm::count_* over two all-bit-block vectors: streaming kernel with this many waves per workgroup (-1 = 4, 0 = the column-per-wave kernel)
    int pair_wgs = 1;          // ... and this many workgroups per CU
    int range_halves = 1;      // comparison search in half-block passes (k_slice_compare_halves) instead of whole-block accumulators (k_slice_compare)
    int gap_count = -1;        // GAP-only counts pipelines: counting formulation (k_pipe_counts_gapcount): -1 = automatic, 0 = off, 1 = force
    int agg_shape = 0;         // materialised combine_and / combine_and_sub over bit-block-only operands (k_agg_and_sub): 0 = 640 threads, three operand blocks in flight per wave, 1 = 512 threads, four in flight
    int and_rows = -1;         // AND / AND-SUB over GAP-only operands straight from their slabs (k_agg_and_rows, bmx_kernels9.h): -1 = automatic (>= 8 operands per group on average), 0 = never, 1 = whenever the pipeline holds no bit-block
    int and_rows_wg = 256;     // ... threads per workgroup (128 / 256 / 512)
    int and_rows_ipw = 0;      // ... (column, group) items per workgroup: 0 = 1 (more measured no gain: profiles/r05_and_rows)
    int and_rows_nt = 0;       // ... non-temporal loads of the run lists
    int and_rows_depth = 3;    // ... 1-KiB pieces in flight per wave (2 / 3 / 4 / 8)
    int ff_window = 0;         // find_first_and_sub: block columns of the FIRST launch window (each next one is 4x larger): 0 = automatic, -1 = one launch
    int or_window = 0;         // column tiles per launch of k_agg_or_gap_tiled: 0 / -1 = all in one launch (windows measured: no gain)
    int or_rows = -1;          // combine_or over >= 64 GAP-only operands through the tile directories (k_agg_or_rows, bmx_kernels7.h): -1 = when the operands average <= 4.1 chunks per GAP block, 0 = never (k_agg_or_gap_tiled), 1 = always
    int or_depth = 4;          // ... rows (operands) in flight per wave: 4 or 8
    // column-major packed GAP collections (bmx_kernels6.h), cached by operand set
    std::vector<struct bmx_coll*> colls;
    int gap_pack = -1;         // aggregation over GAP-only operands through a packed collection: -1 = the collections bmx_collection_prepare built, 0 = never, 1 = also build one at the first use of a list of >= 64 packable vectors
    uint64_t pack_cap = 16ull << 30, pack_bytes = 0, coll_tick = 0;     // pack_cap: a quarter of the device's free memory at context creation (BMX_PACK_MAX_MB overrides)
    uint64_t coll_gen = 0, coll_next_id = 0;   // coll_gen changes whenever a collection appears or goes: pipelines re-resolve their groups then
    int coll_members = -1;     // lists naming only SOME vectors of their collection through the member directory (k_coll_members): -1 = where it pays (sparse blocks, >= 32 operands per group), 0 = never, 1 = always
    int coll_building = 0;                     // > 0 while a collection is being built (no eviction from under it)
    std::unordered_map<uint64_t, struct bmx_vec*> live_vecs;   // uid -> vector, for callers that hold uids (pipelines) instead of pointers
    float last_pack_ms = 0.f;
    uint32_t max_lds_bytes = 160u * 1024u;   // hipDeviceAttributeMaxSharedMemoryPerBlock of the device (queried at creation)
    int coll_split = 1;        // polarity-1 collections keep single-bit runs as 16-bit positions (half the bytes per isolated bit)
    int coll_build = -1;       // how a polarity-1 split collection is built: -1 = through the tile directories where the operands are sparse (bmx_kernels10.h), 0 = the (operand, column tile) passes of bmx_kernels6.h, 1 = tiles whenever possible
    int coll_shape = 4;        // k_coll_apply shape: 0 = 256 threads, 1 = 256 + prefetch, 2 = 512 (configs[4] 2.50 vs 2.61 ms, the AND cases equal), 3 = 512 + prefetch, 4 = 512 with a wave's batch as ONE 4-KiB piece (default: -2.8 % / -1.7 %, profiles/r03ag), 5 = 4 + prefetch
    int pair_nt = 1;           // ... with non-temporal loads
    int pair_loop = -1;        // pairwise counts over mixed block kinds: -1 = persistent kernel (4 workgroups per CU), 0 = a wave per column, N = workgroups per CU
    int op2_nt = 3;            // ... bit 0: non-temporal loads, bit 1: non-temporal stores
    int op2_loop = -1;         // materialised pairwise ops over mixed block kinds (>= 2,048 blocks): -1 = persistent kernel (k_op2_loop, 4 workgroups per CU), 0 = a wave per column (k_op2), N = workgroups per CU
    int op2_wgs = 4;           // workgroups per CU of the streaming materialised pairwise kernel (k_op2_stream)
    int eq_big_shape = 2;      // lean table: 2 = 768 threads at 3 waves per SIMD, 8 filter reads in flight, 256 Kbit filter, 512-entry queues -- taken for every batch size over more than 16 planes; 1 = 512 threads, 2 waves per SIMD; 0 = 128 Kbit filter + 1,024-entry queues
    int eq_big = -1;           // batched equality counts: -1 = lean 9,216-value table when the batch has more than 2,048 values, 0 = never, 1 = always
    int coll_window = 0;       // block columns per launch of k_coll_apply (0 = one launch)
    int rs_lines = 1;          // build_rs_index also lays the vector out as rank lines (one 128-B line per rank query; +108 % of the raw bits): 1 = where that is <= 1.2 x the vector's own device bytes (dense vectors), 2 = always, 0 = never
    int rs_sdir_shift = 0;     // ones per select-directory entry = 2^this; 0 = from the density (an entry per ~10 lines); grown when the directory would pass 8 MB
    int rs_select_lines = 2;   // select through the rank lines (octant directory + interpolated line guess verified by the line headers): 0 = k_select_l, 2 = select directory over the lines (k_select_sdir)
    int rs_select_top = -1;    // select with the 65,536-entry directory summary in LDS (k_select_top): -1 = batches of >= 4 M queries, 0 = never, 1 = always (where the summary exists)
    int rs_select_sel = -1;    // select lines (k_select_sel: the ones' positions laid out 60 / 30 per 128-byte line, one line per query, no search): -1 = built where they cost <= 2 x the vector + its rank lines, 0 = never built / never used, 1 = always (16-bit offsets, 32-bit if a line spans >= 2^16 bits), 2 = always with 32-bit offsets
    int rs_sorted_hint = 0;    // the caller's select batches arrive with ascending ranks (enumeration): the shape that is fastest for them
    int rs_lanes = 0;          // rank: lanes per query (k_rank_l): 0 = automatic, 8 = the original kernel, 2, 4
    int xcd_swz = 1;
};

struct bmx_vec {
    bmx_ctx* ctx;
    uint64_t uid;                                             // never reused: keys of the packed-collection cache
    uint64_t nbits; uint32_t nblocks;
    uint32_t counts[4]; uint64_t gap_words; uint32_t n_bit;   // n_bit = slots of d_bits
    u64* d_desc; uint4* d_bits; u16* d_gaps;
    u32* d_ord;        // result vectors whose slab has unused slots: ordinal of every bit-block (download gathers), else null
    bool ord_lazy;     // ... whose ordinals have not been computed yet (vec_build_ord does it when a download first needs them)
    void* d_tdir;      // tile directory (bmx_kernels7.h): 16 B per 14 blocks, vectors with GAP or FULL blocks only, else null
    uint64_t count; bool count_valid;                         // popcount of the vector when the kernel that produced it folded one
    size_t bytes;
};

struct bmx_pipeline {
    bmx_ctx* ctx;
    uint32_t ngroups, ncols, col_stride, n_ops;
    uint32_t null_row_off;                // every column record ends with a row that is always ROW_EMPTY: what the asynchronous counts run under a search limit points a finished group at
    bool has_gap;
    bool has_bit = false;      // any operand vector holds a bit-block
    uint32_t gap_avg_words = 0;  // average GAP block size of the operands (16-bit words incl. padding)
    uint64_t nbits;                       // max size of the operands
    // LDS-staged path (k_pipe_counts_staged): distinct vectors ("planes") + per-group plane masks
    uint32_t nplanes, nchunks; bool staged_ok;
    const u64** d_udesc; u32* d_unblk; u32* d_gmask; u32* d_gskip;
    std::vector<u32>* h_row_off;          // host copy: row offset of each group inside a column record
    std::vector<u32>* h_and_n;            // host copy: AND operands per group
    std::vector<u32>* h_sub_n = nullptr;  // host copy: SUB operands per group
    u64* d_dmat;
    u32* d_meta;       // row_off | and_n | sub_n | and_off | sub_off (ngroups each) | nblocks (n_ops)
    const u64** d_descs;
    size_t bytes;
    // GAP-only pipelines: the uids of the operand vectors (AND lists, then SUB lists, in group order) -- never the pointers: a
    // vector freed before the pipeline is simply not found in any collection any more -- and what they resolved to
    uint64_t search_limit = ~0ull;       // pipeline::set_search_count_limit (src/bmaggregator.h:255): a group needs no more than this many hits
    uint32_t last_windows = 0, last_windows_planned = 0;   // launch windows of the last synchronous counts run under a limit
    std::vector<uint32_t>* h_win_groups = nullptr;        // ... and the arg-groups every launched window ran over
    std::vector<uint32_t>* h_stop = nullptr;              // ... and, per group, the block column at which it reached the limit (0xFFFFFFFF: never)
    std::vector<uint64_t>* h_uids = nullptr;
    uint64_t cm_gen = ~0ull;            // ctx->coll_gen at the last resolution
    uint64_t cm_tried_gen = ~0ull - 1;  // ctx->coll_gen after the last attempt to build the group's collections (gap_pack 1): not retried until it changes
    uint64_t cm_a_id = 0, cm_s_id = 0;  // collections serving the AND lists / SUB lists (0 = none)
    bool cm_full = false;               // one group whose lists name their whole collections: the streaming kernel
    void* cm_buf = nullptr;             // device: member indices + CollGroup[ngroups]
    size_t cm_groups_off = 0;
};

// a column-major packed interval collection of a set of vectors (bmx_kernels6.h) with its member directory (bmx_kernels8.h)
struct bmx_coll {
    std::vector<uint64_t> key;            // member uids in member order
    std::unordered_map<uint64_t, uint32_t>* index;   // uid -> member index
    int polarity;
    uint32_t ncols, nvec;
    u32* d_runs; u64* d_off; u32* d_cnt; u32* d_flags;
    u32* d_cnt_s;                         // split bag (polarity 1): single-bit runs per column, kept as 16-bit positions behind the multi-bit runs; else null
    u32* d_dir; u32* d_dir_s;             // member directory [ncols][nvec + 1]: entries (split: multi-bit runs) before member i | kind << 30; singles before member i
    u32* d_bt = nullptr;                  // tile build (bmx_kernels10.h): the (tile, group of 64 members, column) run counts, kept so that the
    bool dir_pending = false;             // ... member directory can be built when a call first needs it (coll_ensure_dir)
    uint64_t entries, bytes, run_bytes, last_use, id;
    uint64_t alg_bytes;                   // algorithmic bytes of the GAP operands: sum of 2 x (len + 1)
    bool has_bit;                         // a bit-block was found while counting: unusable
    bool prepared;                        // built by bmx_collection_prepare (not by the gap_pack 1 policy)
    int pins = 0;                         // > 0 while a one-shot call holds a raw pointer to it across allocations: never evicted then
    float build_ms;
};

// an asynchronous result (bmx_op2_dev): the vector is complete on the stream, its block-kind counts are on their way into a
// pinned slot; bmx_pending_wait turns it into an ordinary vector.  A handle type of its own: no other entry can be handed one.
struct bmx_pending {
    bmx_ctx* ctx;
    bmx_vec* v;
    int slot;
    hipEvent_t ev;
    uint64_t gap_bound;       // upper bound of the GAP words the result holds (0: it cannot hold a GAP block); its GAP slab has that size
    void* scratch;            // st[] / offs[] / candidate list of the producing kernel (GAP path), until the result is resolved
    bool resolved;            // the vector is already an ordinary one (x & x, x | x: a block-for-block copy): the wait only hands it over
};

struct bmx_rs {
    bmx_ctx* ctx;
    uint32_t nblocks; uint64_t count;
    u32* d_bcount; u64* d_sub; u64* d_rcount; u16* d_cum;
    u16* d_gidx;                                          // GAP blocks: first run reaching each 1024-bit wave
    u64* d_sample; uint32_t nsamples, sample_shift;       // top level of the select search (<= 2048 entries)
    u32* d_lines;                                         // rank lines: 69 x 128 B per block (count before the line + 960 bits), or null
    u32* d_sdir; uint32_t sdir_shift; uint64_t sdir_entries;  // with rank lines: select directory (line of every 2^shift-th one) + sentinel
    u16* d_dir8;                                          // with rank lines: ones of a block before each of its eight 8,192-bit octants
    u8* d_sel = nullptr; uint32_t sel_bits = 0; uint64_t sel_lines = 0;  // select lines (bmx_kernels11.h): the positions of the ones, K = 60 (16-bit offsets) / 30 (32-bit) per 128-byte line, or null
    u32* d_stop = nullptr; uint32_t stop_shift = 0, stop_fb = 0;       // with the select directory: its 65,536-entry summary for LDS (k_select_top): positions to 1 / 2^stop_fb of a line as base[1024] + 16-bit offsets, or null
    size_t bytes;
};

// asynchronous building blocks the group layer composes (bmx.hip): everything is enqueued on ctx->stream,
// the 8-byte results land in ctx->h_small[slot] (pinned) after the stream has been synchronised
int bmx_i_count_async(bmx_ctx* ctx, const bmx_vec* a, int slot);
int bmx_i_count_op2_async(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int slot);
