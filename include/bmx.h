/*
 * bmx.h -- C-ABI of the MI355X-native bit-vector set-algebra engine (libbmx.so).
 *
 * This is the drop-in boundary for the bm::bvector<> / bm::aggregator<> hot path
 * of BitMagic (reference = /root/reference, v9.2.1).  The reference has no
 * run-time FFI for this path: its only seam is the compile-time VECT_* macro
 * table (src/bmavx2.h:3432-3587, selected by src/bmsimd.h:24-65), whose calls
 * cover one 8 KiB block each -- three orders of magnitude too fine for a GPU
 * launch.  The boundary therefore sits one level up, at the per-vector loops
 * that walk the (i,j) block tree; each entry point below names the reference
 * interface it replaces.  Conventions follow the reference's own C wrapper
 * (lang-maps/libbm/include/libbm.h:28-35,74-76,123-140): opaque handles, every
 * function returns an int status, outputs through pointer arguments, no C++
 * exceptions or STL types cross the ABI, plain pointers and sizes only.
 *
 * Data model handed across the boundary: a vector is a
 * flat BLOCK TABLE -- kinds[nb] in {NULL, FULL, BIT, GAP} plus one contiguous
 * slab of 8 KiB bit-blocks and one slab of GAP blocks (uint16 run-end lists) --
 * obtained by walking blocks_manager::top_blocks_root() (src/bmblocks.h:595)
 * exactly like bvector::count() does (src/bm.h:2436-2474).
 *
 * Threading: one bmx_ctx = one HIP stream; calls on a ctx are serialised by the
 * caller (the reference aggregator is not thread-safe either,
 * src/bmaggregator.h:824-853); distinct contexts may run concurrently.
 * Vectors are immutable once created and may be shared by any number of
 * operations on the context that owns them.
 */
#ifndef BMX_H
#define BMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: numeric values of libbm.h:28-35 where a twin exists */
#define BMX_OK            0
#define BMX_ERR_BADALLOC  1   /* BM_ERR_BADALLOC */
#define BMX_ERR_BADARG    2   /* BM_ERR_BADARG   */
#define BMX_ERR_RANGE     3   /* BM_ERR_RANGE    */
#define BMX_ERR_DEVICE    4   /* BM_ERR_CPU twin: no usable gfx950 device / HIP failure */

/* block kinds of the flat block table (tagged pointers of src/bmdef.h:165-199) */
#define BMX_NULL 0
#define BMX_FULL 1
#define BMX_BIT  2
#define BMX_GAP  3

/* set operations (src/bmconst.h set_operation: set_AND, set_OR, set_XOR, set_SUB) */
#define BMX_AND 0
#define BMX_OR  1
#define BMX_XOR 2
#define BMX_SUB 3

/* geometry (src/bmconst.h:55-87) */
#define BMX_BLOCK_WORDS 2048u
#define BMX_BLOCK_BITS  65536u

typedef struct bmx_ctx      bmx_ctx;
typedef struct bmx_vec      bmx_vec;
typedef struct bmx_pending  bmx_pending;  /* an asynchronous result that has not been resolved into a vector yet (bmx_op2_dev) */
typedef struct bmx_pipeline bmx_pipeline;
typedef struct bmx_rs       bmx_rs;

/* ---- library / context (BM_init, BM_error_msg, BM_simd_version: libbm.h:123-140) ---- */
const char* bmx_error_msg(int status);
/* last HIP/driver error text recorded on this thread ("" if none) */
const char* bmx_last_error(void);
/* 950 for gfx950: the analogue of bm::simd_version() (src/bmsimd.h:67-90) */
int bmx_simd_version(void);
int bmx_device_count(int* n);
/* stream: a hipStream_t owned by the caller, or NULL to let the context create its own */
int bmx_ctx_create(int device, void* stream, bmx_ctx** out);
int bmx_ctx_destroy(bmx_ctx* ctx);
int bmx_ctx_synchronize(bmx_ctx* ctx);
/* launch-shape knobs of the counts pipeline (results never depend on them; 0 / -1 = automatic):
 * "pipe_rows" 0|8|4|2|1 (KiB of a block per work item), "pipe_unroll" 0|1|2|4|8|16, "pipe_nt" 0|1,
 * "pipe_wg" 0|64..1024 (multiples of 64; the default build carries 256, 384, 512, 640, 768), "pipe_staged" -1|0|1, "pipe_slots" 8|16,
 * "pipe_window" -1|0|N (block columns per launch), "pipe_split" -1|0|1, "direct_cols" 0..N (one-launch aggregation over small collections), "ff_window" -1|0|N (first launch window of find_first_and_sub), "pair_stream" -1|0|2|4|8 and "pair_wgs" 1..8 (shape of the streaming pairwise count kernel), "pair_loop" -1|0|1..5 and "pair_nt" 0|1 (persistent pairwise count kernel for mixed block kinds: workgroups per CU, non-temporal loads), "eq_big_shape" 0|1|2, "gap_count" -1|0|1 (counting formulation for GAP-only counts pipelines), "range_halves" 0|1 (comparison search in half-block passes), "rs_lanes" 0|2|4|8 (lanes per rank query), "rs_lines" 0|1|2 (build_rs_index also lays the vector out as rank lines: one 128-byte line per rank query), "rs_select_lines" 0|1|2 (select through the block index | the lines + octant directory | the select directory over the lines) and "rs_sdir_shift" 0|6..20 (log2 of the ones per select-directory entry; 0 = an entry per ~10 lines), "gap_pack" -1|0|1 (packed collections, see below), "eq_big" -1|0|1 (table form of bmx_slice_eq_counts), "op2_wgs" 1..8 and "op2_nt" 0..3 (streaming bit_and/or/xor/sub kernel: workgroups per CU; bit 0 / 1 = non-temporal loads / stores), "op2_loop" -1|0|1..8 (persistent bit_and/or/xor/sub kernel for mixed block kinds: workgroups per CU), "or_rows" -1|0|1 and "or_depth" 4|8 (combine_or over >= 64 sparse GAP-only operands through the vectors' tile directories: automatic | never | always; rows in flight per wave), "coll_members" -1|0|1 (lists that are SOME vectors of a prepared collection: the member directory automatic | never | whenever covered), "or_tile" 0..3, "xcd_swizzle" 0|1.  Environment twins (BMX_PIPE_ROWS, ...) pass the same checks. */
int bmx_ctx_set_tuning(bmx_ctx* ctx, const char* key, int value);
/* The context keeps freed device blocks in a size-keyed cache (results of same-shaped
 * operations re-use them instead of paying hipMalloc/hipFree, which synchronises the
 * device); bmx_ctx_trim gives the cache back to the driver.  BMX_POOL_MAX_MB caps it. */
int bmx_ctx_trim(bmx_ctx* ctx);
/* bytes of HBM currently held by vectors/pipelines of this context */
int bmx_ctx_mem_used(const bmx_ctx* ctx, uint64_t* bytes);

/* ---- vectors ---- */
/* Upload a flattened block table (walk of top_blocks_root()).
 * offs[nb]: BIT -> ordinal of the block inside bit_slab; GAP -> uint16 word offset
 * of the block inside gap_slab.  The engine copies; the host keeps its blocks. */
int bmx_vec_upload(bmx_ctx* ctx, uint64_t nbits, uint32_t nblocks,
                   const uint8_t* kinds, const uint32_t* offs,
                   const uint32_t* bit_slab, uint32_t n_bit_blocks,
                   const uint16_t* gap_slab, uint64_t gap_words,
                   bmx_vec** out);
/* bm::bit_import_u32(bv, words, nwords, optimize)  src/bmbvimport.h:46.
 * Raw bits go to the device; classification into NULL/FULL/BIT and, with
 * optimize, GAP compression (optimize_bit_block, src/bmblocks.h:1412) run there. */
int bmx_vec_import_bits(bmx_ctx* ctx, const uint32_t* words, uint64_t nwords,
                        int optimize, bmx_vec** out);
/* Synthetic vector generated on the device with the counter-based generator of
 * SURVEY.md section 8(d) (normative arithmetic: oracle/bmx_oracle.c bmo_gen_word64):
 * per-bit Bernoulli(density_q16 / 65536); with_common ORs in the shared vector
 * (correlated data set, tests/perf/perf.cpp:234-267); then as bmx_vec_import_bits. */
int bmx_vec_generate(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                     uint32_t density_q16, uint64_t nbits, int optimize, bmx_vec** out);
/* Block-range shard of the same logical vector: blocks [nb_from, nb_to) only, as a vector of nb_to - nb_from
 * blocks (every GPU holds only its block range of every operand; the per-column loops
 * src/bmaggregator.h:1113-1121,1184-1218 are independent, so shard results add up / concatenate). */
int bmx_vec_generate_shard(bmx_ctx* ctx, uint64_t seed, uint32_t vec_id, int with_common,
                           uint32_t density_q16, uint64_t nbits, uint32_t nb_from, uint32_t nb_to,
                           int optimize, bmx_vec** out);
int bmx_vec_free(bmx_ctx* ctx, bmx_vec* v);
/* bvector::calc_stat (src/bm.h:4010): counts[kind]; bit_slab_blocks / gap_words =
 * sizes (8 KiB blocks / uint16 words) of the two slabs bmx_vec_download fills.
 * Result vectors keep one slab slot per block column, so bit_slab_blocks may
 * exceed counts[BMX_BIT]; offs[] always indexes the slab that is downloaded. */
int bmx_vec_info(const bmx_vec* v, uint64_t* nbits, uint32_t* nblocks,
                 uint32_t counts[4], uint32_t* bit_slab_blocks, uint64_t* gap_words);
/* algorithmic bytes an operation must read for this operand: 8,192 B per bit-block,
 * 2 x (len + 1) B per GAP block (len = buf[0] >> 3), nothing for NULL / FULL -- what benchmark reports divide by */
int bmx_vec_operand_bytes(bmx_ctx* ctx, const bmx_vec* v, uint64_t* bytes);
/* Download the block table (feeds blocks_manager on the host, src/bmblocks.h:1355).
 * Array sizes come from bmx_vec_info; any pointer may be NULL to skip that part. */
int bmx_vec_download(bmx_ctx* ctx, const bmx_vec* v, uint8_t* kinds, uint32_t* offs,
                     uint32_t* bit_slab, uint16_t* gap_slab);
/* The vector as the SORTED positions of its set bits -- the index-list form an inverted-index consumer reads back (what
 * bm::bvector<>::enumerator / bm::for_each_bit_blk feed into a back-insert iterator, src/bmaggregator.h:1226-1284,
 * src/bmalgo_impl.h for_each_bit).  width = 4 (uint32_t positions; vectors of <= 2^32 bits) or 8 (uint64_t).
 * *n = number of set bits; when it exceeds cap nothing is written and BMX_ERR_RANGE is returned (call again with a
 * buffer of *n entries; bmx_count gives the number in advance).  _dev: the buffer is device memory. */
int bmx_vec_to_indices(bmx_ctx* ctx, const bmx_vec* v, int width, void* out, uint64_t cap, uint64_t* n);
int bmx_vec_to_indices_dev(bmx_ctx* ctx, const bmx_vec* v, int width, void* d_out, uint64_t cap, uint64_t* n);
/* expand to raw words (export twin of bit_import_u32) */
int bmx_vec_to_words(bmx_ctx* ctx, const bmx_vec* v, uint32_t* words, uint64_t nwords);

/* ---- pairwise set algebra ---- */
/* bvector::count()  src/bm.h:2431 */
int bmx_count(bmx_ctx* ctx, const bmx_vec* a, uint64_t* count);
/* bvector::bit_and/bit_or/bit_xor/bit_sub(bv1, bv2, opt_mode)  src/bm.h:6185,5973,6072,6403
 * opt_compress != 0 re-compresses produced blocks (opt_compress, src/bm.h:6263). */
int bmx_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int opt_compress,
            bmx_vec** result);
/* bit_and/or/xor/sub + count() of the result in ONE call (the reference's callers write
 * bv.bit_and(a, b); bv.count();  src/bm.h:6185,2431 -- tests/perf and lang-maps/libbm BM_bvector_combine_AND + BM_bvector_count,
 * libbm.h:439,314): *count = popcount of the result.  result may be NULL: count only (bm::count_*, src/bmalgo.h:49-149).
 * Vectors of fewer than 2,048 blocks take one launch and one synchronise for both (the kernel folds the count; bmx_op2
 * followed by bmx_count costs the same: the count travels with the result vector). */
int bmx_op2_count(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, int opt_compress,
                  bmx_vec** result, uint64_t* count);
/* The same three-operand operations (src/bm.h:6185,5973,6072,6403; opt_none), ASYNCHRONOUS on the context's stream: the call
 * enqueues the kernel and returns.  What bmx_op2 waits for is not the result -- it is complete on the stream when the kernel
 * ends -- but the counts of its block kinds, which the host needs to dispatch later operations over it; here they travel to
 * pinned memory behind the kernel and are read when the result is resolved.  Each operand is EITHER a vector (a / b) OR an
 * unresolved result of an earlier bmx_op2_dev (pa / pb) -- exactly one of each pair is non-NULL -- so a chain of operations
 * stays on the stream and pays one synchronise at its end instead of one per operation.  Operands of any block kinds: where
 * they hold GAP blocks the result may too (at most the operands' GAP words together), so its GAP slab is allocated at that
 * bound, the kernel lays the GAP results out itself and their conversion is enqueued right behind it -- the descriptors are
 * complete on the stream, bmx_pending_wait trims the slab.
 * Aliased operands follow the reference's rule (src/bm.h:6191, 5984, 6081, 6412): x & x and x | x are block-for-block copies
 * (an unresolved x is waited for), x ^ x and x - x are empty.
 *   bmx_pending_wait   waits for THIS result only, turns it into an ordinary vector (*out; the handle is consumed)
 *   bmx_pending_free   drops an unresolved result (it may still be an operand of operations enqueued earlier)
 * At most 1,024 unresolved results per context (round 6; 64 before) and 2,000,000 blocks (1.3e11 bits) per operand (BMX_ERR_RANGE).  bmx_pending is a handle type of its own: no other entry point
 * accepts one, so a vector whose kind counts are not known yet can never reach a dispatch decision. */
int bmx_op2_dev(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_pending* pa, const bmx_vec* b, const bmx_pending* pb,
                bmx_pending** out);
int bmx_pending_wait(bmx_ctx* ctx, bmx_pending* p, bmx_vec** out);
int bmx_pending_free(bmx_ctx* ctx, bmx_pending* p);
/* bm::count_and/count_or/count_xor/count_sub  src/bmalgo.h:49,149,81,115 */
int bmx_count_op2(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* count);
/* same, asynchronous on the context's stream; d_count is DEVICE memory (one uint64) */
int bmx_count_op2_dev(bmx_ctx* ctx, int op, const bmx_vec* a, const bmx_vec* b, uint64_t* d_count);

/* ---- aggregator ---- */
/* aggregator::combine_or(target, src, n)  src/bmaggregator.h:1101 */
int bmx_agg_or(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, bmx_vec** result);
/* same after aggregator::set_optimization(opt) (src/bmaggregator.h:359): result blocks pass through
 * opt_copy_bit_block(.., opt_mode_, ..) (:1658) */
int bmx_agg_or_opt(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, bmx_vec** result);
/* aggregator::combine_and_sub(target, and, n_and, sub, n_sub, false)  src/bmaggregator.h:1162
 * (combine_and(target) == n_sub 0, :1030-1039).  *any = result is non-empty. */
int bmx_agg_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                    const bmx_vec* const* src_sub, size_t n_sub,
                    bmx_vec** result, int* any);
/* aggregator::combine_and_sub(BII bi, and, n_and, sub, n_sub) / combine_and_sub_bi(bi)  src/bmaggregator.h:450,533,1068,1226:
 * the AND-SUB result as sorted positions instead of a bit-vector (same buffer rules as bmx_vec_to_indices). */
int bmx_agg_and_sub_indices(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                            const bmx_vec* const* src_sub, size_t n_sub, int width, void* out, uint64_t cap, uint64_t* n);
/* aggregator::find_first_and_sub(idx, and, n_and, sub, n_sub)  src/bmaggregator.h:1458:
 * index of the first set bit of the AND-SUB result, nothing materialised. */
int bmx_find_first_and_sub(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                           const bmx_vec* const* src_sub, size_t n_sub, int* found, uint64_t* idx);
/* the same under aggregator::set_range_hint(from, to) (src/bmaggregator.h:481,974): only block columns
 * [from >> 16, to >> 16] are visited (:1470-1512); when both ends lie in ONE block that column is also AND-ed with
 * the bit range [from & 65535, to & 65535] (range_gap_blk_, :980-988, 2354-2358) -- a hint spanning several blocks is
 * block-granular, exactly like the reference. */
int bmx_find_first_and_sub_range(bmx_ctx* ctx, const bmx_vec* const* src_and, size_t n_and,
                                 const bmx_vec* const* src_sub, size_t n_sub, uint64_t from, uint64_t to,
                                 int* found, uint64_t* idx);
/* aggregator::combine_shift_right_and(bv_target, src, n, any)  src/bmaggregator.h:552,2494 (member form
 * :473,1089): T_0 = src[0], T_k = (T_{k-1} >> 1) & src[k] with ">>" moving bit p to p+1 across block
 * borders (process_shift_right_and :2618) -- result bit p is set iff src[k] has bit p-(n-1-k) for every k
 * (sequence search).  Stored with the aggregator's optimisation mode (opt_none by default, :917,2600).
 * any != 0: stop at the first block column that produced a result (:2519) -- the target then holds that one
 * block.  *found = target is non-empty.  Empty list => cleared target, found = 0 (:2499-2503). */
int bmx_agg_shift_right_and(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, int opt_compress, int any,
                            bmx_vec** result, int* found);
/* same under set_compute_count(true) (src/bmaggregator.h:363,2595): no target, *count = aggregator::count(). */
int bmx_agg_shift_right_and_count(bmx_ctx* ctx, const bmx_vec* const* src, size_t n, uint64_t* count);
/* ---- bit-sliced comparison search (the range-search half of bm::sparse_vector_scanner<SV>, unsigned values) ----
 * slices[i] = device vector of bit-plane i (sv.get_slice(i)), NULL where the plane does not exist; nslices plays
 * effective_slices(); size = sv.size() (rows); not_null = sv.get_null_bvector() or NULL.
 *   BMX_CMP_GT/GE/LT/LE (v0)   find_gt / find_ge / find_lt / find_le      src/bmsparsevec_algo.h:2690,2717,2790,2824
 *   BMX_CMP_RANGE [v0, v1]     find_range (closed, swapped when v1 < v0)   :2862
 *   BMX_CMP_EQ (v0)            find_eq incl. value 0                       :4356,2387
 *   BMX_CMP_ZERO / NONZERO     find_zero(null_correct = true) / find_nonzero :2290,4464
 * NULL elements are stored as 0: where the predicate admits 0 the result is AND-ed with not_null
 * (needs_null_correct_*, :1703-1735; correct_nulls :2376).  One pass over the planes (bmx_kernels4.h).
 * result may be NULL (count only: nothing is materialised); count may be NULL. */
#define BMX_CMP_GT 0
#define BMX_CMP_GE 1
#define BMX_CMP_LT 2
#define BMX_CMP_LE 3
#define BMX_CMP_RANGE 4
#define BMX_CMP_EQ 5
#define BMX_CMP_ZERO 6
#define BMX_CMP_NONZERO 7
int bmx_slice_compare(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                      uint64_t size, const bmx_vec* not_null, bmx_vec** result, uint64_t* count);

/* The same for SIGNED containers (bm::sparse_vector<int, ..>): slices[0] is the sign plane, slices[1..] the magnitude planes
 * of the reference's encoding (v >= 0 -> v << 1, v < 0 -> ((-(v + 1)) << 1) | 1: base_sparse_vector::s2u, src/bmbmatrix.h:2536);
 * bounds are signed.  What the reference builds from whole-vector passes (find_gt_horizontal_s, src/bmsparsevec_algo.h:1484,
 * 3033-3160, and find_ge / lt / le / range on top of it) is the same one pass over the magnitude planes combined with the sign
 * block in the kernel.  NULL rows (stored as +0) drop out wherever the predicate admits 0. */
int bmx_slice_compare_signed(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, int64_t v0, int64_t v1,
                             uint64_t size, const bmx_vec* not_null, bmx_vec** result, uint64_t* count);
/* bmx_slice_compare (count only) that also reports the plane bytes the walk actually had to read -- it stops in a block column
 * as soon as no row is "equal so far" -- i.e. the ALGORITHMIC bytes of this search (benchmark reports divide by them) */
int bmx_slice_compare_stat(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                           uint64_t size, const bmx_vec* not_null, uint64_t* count, uint64_t* plane_bytes);

/* a batch of equality searches, counts only: counts[q] = rows equal to values[q] -- what n pipelined AND-SUB groups of
 * prepare_and_sub_aggregator compute (src/bmsparsevec_algo.h:2593-2640,3236,3408), in ONE pass over the planes whatever n is
 * (bit-matrix transposition + hash lookup, bmx_kernels4.h).  nslices <= 32 (else BMX_ERR_RANGE: use the pipeline form). */
int bmx_slice_eq_counts(bmx_ctx* ctx, const bmx_vec* const* slices, size_t nslices, const uint64_t* values, size_t n,
                        uint64_t size, const bmx_vec* not_null, uint64_t* counts);

/* ---- packed collections: a column-major copy of the GAP run lists of a SET of vectors (bmx_kernels6.h, bmx_kernels8.h) ----
 * Vectors are immutable and device-resident, so the library owns their layout.  combine_or / combine_and / combine_and_sub
 * and pipelines over many operands made of GAP (+ NULL / FULL) blocks read thousands of separate slabs in small pieces
 * (src/bmaggregator.h:1808-1924 walks them operand by operand).  bmx_collection_prepare transposes the GAP blocks of a
 * list of vectors ONCE into column-major order -- one contiguous run list per block column plus a member directory (where
 * each vector's runs sit inside every column) -- and from then on EVERY aggregation whose operands are vectors of that
 * collection is served by it, whatever subset of them it names, in whatever order, with or without repeats:
 *   - a list naming all the collection's vectors streams the column regions (k_coll_apply);
 *   - any other list, and every arg-group of a pipeline (many groups over shared operands, :1292-1399), reads its members'
 *     pieces of the regions through the directory (k_coll_members).
 * Roles: an OR list or SUB list needs the vectors' 1-runs (BMX_ROLE_OR = BMX_ROLE_SUB), an AND list their 0-runs
 * (BMX_ROLE_AND: AND_i x_i = NOT OR_i NOT x_i); an index searched both ways prepares both.  Results are identical to the
 * descriptor-table kernels.  A collection lives until one of its vectors is freed, it is the least recently used one when
 * device memory runs short (budget: a quarter of the free HBM at context creation, BMX_PACK_MAX_MB), or the context goes.
 * Tuning key "gap_pack": -1 (default) = use what bmx_collection_prepare built, nothing is built on the side; 0 = never use
 * collections; 1 = also build one at the FIRST use of a list of >= 64 packable vectors (synchronous entries only).
 * "coll_split" 0|1: OR / SUB collections keep a single-bit run as one 16-bit position (default 1: half the bytes for sparse
 * vectors).
 *   bmx_collection_prepare  builds the collection of the list in the role (1..65535 vectors without bit-blocks)
 *   bmx_ctx_pack_stats      collections held, their bytes (runs + directories), device time of the last build
 *   bmx_ctx_pack_run_bytes  the run-list bytes alone: what a full aggregation over the collections streams */
#define BMX_ROLE_AND 0
#define BMX_ROLE_OR  1
#define BMX_ROLE_SUB 2
int bmx_collection_prepare(bmx_ctx* ctx, const bmx_vec* const* vecs, size_t n, int role);
int bmx_ctx_pack_stats(const bmx_ctx* ctx, uint32_t* n_collections, uint64_t* bytes, float* last_build_ms);
int bmx_ctx_pack_run_bytes(const bmx_ctx* ctx, uint64_t* bytes);

/* aggregator::pipeline<agg_opt_only_counts>  src/bmaggregator.h:62-103,222-341:
 * arg-groups are given as concatenated operand lists, and_n[g] / sub_n[g] per
 * group (pipeline::add() + arg_groups::add(bv, 0|1) + complete(), :2784-2931).
 * Lifetime: like the reference's pipeline, which stores bvector POINTERS (:2939), the object references its
 * operand vectors (their device tables): they must stay alive (not bmx_vec_free'd) until bmx_pipeline_destroy.
 * Limits: < 2^20 groups, <= 65535 operands per list, total operands + 2 x groups < 2^32 (else BMX_ERR_RANGE). */
int bmx_pipeline_create(bmx_ctx* ctx,
                        const bmx_vec* const* and_list, const uint32_t* and_n,
                        const bmx_vec* const* sub_list, const uint32_t* sub_n,
                        size_t ngroups, bmx_pipeline** out);
int bmx_pipeline_destroy(bmx_ctx* ctx, bmx_pipeline* p);
/* aggregator::combine_and_sub(pipe)  src/bmaggregator.h:1292 with counts only
 * (:1392-1399): counts_out[g] = popcount of group g's AND-SUB result restricted
 * to block columns [nb_from, nb_to) (nb_to = UINT32_MAX: all).  Synchronous. */
int bmx_pipeline_run_counts(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to,
                            uint64_t* counts_out);
/* pipeline::set_search_count_limit  src/bmaggregator.h:255, honoured PER ARG-GROUP at :1362-1367: a group whose count has
 * reached `limit` is not evaluated on the blocks that follow ("can find more, cannot find less": top-k searches stop paying
 * once they have enough).  bmx_pipeline_run_counts then walks the block columns in ascending launch windows (each 4 x the one
 * before); the per-group totals stay on the device, and after every window the groups that have enough are DROPPED from the
 * tables the next window is launched over (one word -- the number of groups left -- is what the host waits for): a later
 * window runs over the groups that still need hits only, and no window is launched once none is left.  counts_out[g] is
 * >= min(limit, the group's true count) and <= the true count.  limit 0 / bm::id_max (2^32 - 1, 2^48 - 1) / UINT64_MAX = no
 * limit (one run).  As in the reference the limit applies whenever counts are computed: bmx_pipeline_run_results* with
 * counts_out != NULL produces a group's vector up to the window at whose end the group had enough (its count is then
 * >= min(limit, true count)); result-only runs ignore it.  bmx_pipeline_run_counts_dev (asynchronous) honours it without a host
 * decision: the same windows are all enqueued over all groups, and after each one the groups that have enough are pointed at null
 * table entries on the device, so their items of the later windows end at a header read (same totals as the synchronous run; a
 * pipeline that is served as ONE whole packed collection runs to the end and returns its true count).
 * bmx_pipeline_last_windows: windows launched / planned by the last synchronous counts run;
 * bmx_pipeline_last_window_groups: out[w] = arg-groups window w of that run ran over (n = windows launched). */
int bmx_pipeline_set_search_count_limit(bmx_ctx* ctx, bmx_pipeline* p, uint64_t limit);
int bmx_pipeline_last_windows(const bmx_pipeline* p, uint32_t* launched, uint32_t* planned);
int bmx_pipeline_last_window_groups(const bmx_pipeline* p, uint32_t* out, uint32_t cap, uint32_t* n);
/* Same, asynchronous on the context's stream; d_counts is DEVICE memory
 * (ngroups x uint64) -- e.g. the buffer a following RCCL all-reduce sums. */
int bmx_pipeline_run_counts_dev(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to,
                                uint64_t* d_counts);
/* aggregator::combine_and_sub(pipe) for the other run options (src/bmaggregator.h:62-103):
 *   results_out != NULL  -> Opt::is_make_results(): one vector per arg-group, NULL where the group found
 *                           nothing (:1406-1415); blocks stored with opt_compress (:1421)
 *   counts_out  != NULL  -> Opt::is_compute_counts()
 *   or_target_out != NULL -> pipeline::set_or_target (:245): OR of or_target_in (may be NULL) and every
 *                           group result, optimised (:1440-1447)
 * set_search_count_limit (:255): with counts_out != NULL a group's vector is produced up to the window at whose end the group
 *                           had enough hits (see bmx_pipeline_set_search_count_limit); without counts the limit does not apply. */
int bmx_pipeline_run_results(bmx_ctx* ctx, bmx_pipeline* p, bmx_vec** results_out, uint64_t* counts_out,
                             const bmx_vec* or_target_in, bmx_vec** or_target_out);
/* same for a pipeline whose options enable search masks (agg_run_options<.., .., true>::is_masks(), :65,78) under
 * set_range_hint: only block columns [nb_from, nb_to) are visited (:1312-1346); results hold nothing outside */
int bmx_pipeline_run_results_range(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, bmx_vec** results_out,
                                   uint64_t* counts_out, const bmx_vec* or_target_in, bmx_vec** or_target_out);
/* the same given the aggregator's range hint itself (set_range_hint(from, to), src/bmaggregator.h:481,974): block columns
 * [from >> 16, to >> 16] are visited and -- exactly like the reference -- a hint whose ends lie in ONE block also restricts that
 * column to the bit range [from & 65535, to & 65535] (range_gap_blk_, :980-988,2354-2358); results_out / or_target_out may
 * be NULL (counts only). */
int bmx_pipeline_run_results_hint(bmx_ctx* ctx, bmx_pipeline* p, uint64_t from, uint64_t to, bmx_vec** results_out,
                                  uint64_t* counts_out, const bmx_vec* or_target_in, bmx_vec** or_target_out);
/* algorithmic operand bytes one run over [nb_from, nb_to) must read
 * (8192 B per bit-block operand, 2*(len+1) B per GAP operand; NULL/FULL: 0) */
int bmx_pipeline_operand_bytes(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to,
                               uint64_t* bytes);

/* which kernel and launch plan a counts run over [nb_from, nb_to) takes (for benchmark reports / profiles);
 * n_launches (may be NULL) = kernel launches of one run */
int bmx_pipeline_describe(bmx_ctx* ctx, bmx_pipeline* p, uint32_t nb_from, uint32_t nb_to, char* buf, size_t buf_len,
                          uint32_t* n_launches);

/* ---- rank / select ---- */
/* bvector::build_rs_index  src/bm.h:2531 */
int bmx_rs_build(bmx_ctx* ctx, const bmx_vec* v, bmx_rs** out);
int bmx_rs_free(bmx_ctx* ctx, bmx_rs* rs);
/* what the index holds on the device (rs_index itself is ~0.7 MB per 4e9 bits, src/bmrs.h:39-155; here: running counts and a
 * 256-byte row per block, plus -- where they cost no more than 2 x the vector's own device bytes (bit-blocks in over half of the block columns;
 * tuning key "rs_lines" 0 never | 1 this policy | 2 always) -- the rank lines:
 * the vector laid out once more with its running counts interleaved, +108 % of the raw bits) */
int bmx_rs_info(const bmx_rs* rs, uint64_t* bytes, int* has_lines);
/* the select lines of the index, if it has them (tuning key "rs_select_sel": -1 where they cost <= 2 x the vector and its rank
 * lines | 0 never | 1 always | 2 always with 32-bit offsets): the positions of the ones laid out 60 (16-bit offsets) or 30 (32-bit)
 * per 128-byte line, so that select(r) -- bvector::select src/bm.h:5350, rs_index::find src/bmrs.h:492 -- reads ONE line and
 * searches nothing.  offset_bits = 16 | 32 | 0 (none: select runs through the rank lines' directory or the block tables) */
int bmx_rs_select_format(const bmx_rs* rs, int* offset_bits, uint64_t* bytes);
/* rs_index::count()  src/bmrs.h:340 */
int bmx_rs_count(const bmx_rs* rs, uint64_t* count);
/* reference-compatible per-block arrays so a host rs_index can be filled:
 * bcount[nb] (rs_index::count(nb)) and sub_count[nb] packed as
 * first | second<<16 | aux0<<32 | aux1<<48 (src/bm.h:2646-2656, src/bmrs.h:688) */
int bmx_rs_export(bmx_ctx* ctx, const bmx_rs* rs, uint32_t* bcount, uint64_t* sub_count);
/* bvector::count_to / rank(n, rs): ones in [0..n] inclusive  src/bm.h:3120,1449 (batched) */
int bmx_rank_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs,
                   const uint64_t* n, size_t q, uint64_t* out);
/* bvector::select(rank, pos, rs): rank is 1-based  src/bm.h:5350 (batched) */
int bmx_select_batch(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs,
                     const uint64_t* rank, size_t q, uint64_t* pos, uint8_t* found);
/* device-resident query/answer buffers (no PCIe in the timed region) */
int bmx_rank_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs,
                       const uint64_t* d_n, size_t q, uint64_t* d_out);
int bmx_select_batch_dev(bmx_ctx* ctx, const bmx_vec* v, const bmx_rs* rs,
                         const uint64_t* d_rank, size_t q, uint64_t* d_pos, uint8_t* d_found);

/* ---- multi-GPU: device groups (SURVEY.md section 8(b) "init with a device list", section 8(e)) ----
 * Every block column (i,j) is independent for AND/OR/XOR/SUB/COUNT (src/bmaggregator.h:1113-1121,1184-1218,
 * src/bm.h:6226-6271), so a group of n devices shards the linear block range [0, nblocks) into n contiguous
 * pieces: member m holds blocks bmx_group_shard_range(nblocks, m) of EVERY vector (a "sharded vector",
 * bmx_gvec); operand bits never cross xGMI.  One host thread drives all members: kernels are enqueued on every
 * member's stream before the first one is waited for (calls that materialise vectors run the synchronous single-device
 * entry on persistent per-member worker threads); the only exchange is the sum of the per-member popcounts
 * (8 B per arg-group), done on the host (default) or by an RCCL all-reduce over xGMI (BMX_GROUP_RCCL; librccl is
 * loaded on demand, devices must be distinct).  Materialised results stay sharded; bmx_gvec_download gathers.
 * A device may appear several times in `devices` (several streams on one GPU; what the 1-GPU tests use). */
typedef struct bmx_group     bmx_group;
typedef struct bmx_gvec      bmx_gvec;
typedef struct bmx_gpipeline bmx_gpipeline;
typedef struct bmx_grs       bmx_grs;
#define BMX_GROUP_HOST_SUM 0
#define BMX_GROUP_RCCL     1
int bmx_group_create(const int* devices, int n, int flags, bmx_group** out);
int bmx_group_destroy(bmx_group* g);
int bmx_group_size(const bmx_group* g, int* n);
/* member m's context (owned by the group): e.g. to set tuning knobs or upload private vectors */
int bmx_group_ctx(const bmx_group* g, int member, bmx_ctx** ctx);
/* the cut in force for vectors of nblocks blocks: contiguous, exhaustive; by default balanced to within one block */
int bmx_group_shard_range(const bmx_group* g, uint32_t nblocks, int member, uint32_t* nb_from, uint32_t* nb_to);
/* Byte-weighted shard borders (shards "weighted by non-NULL operand bytes").  A group keeps ONE
 * cut of [0, nblocks) per vector length, so the block columns of all operands of an operation stay on the same member;
 * the default cut gives every member the same number of block columns.  An index with empty stretches (NULL top-level
 * ranges cost nothing: blocks_manager::get_block_ptr, src/bmblocks.h:556-564) would leave members idle under it:
 *   bmx_block_table_weights       adds the algorithmic operand bytes of one host block table to weight[nblocks]
 *                                 (8192 B per bit-block, 2 x (len + 1) B per GAP block, 0 for NULL / FULL); call it for
 *                                 every vector of the collection
 *   bmx_group_partition_by_weight cuts where the running weight reaches m/n of the total and makes that cut the one in
 *                                 force for vectors of nblocks blocks (bounds_out: n + 1 borders, may be NULL)
 *   bmx_group_set_partition       the same with explicit borders (bounds[0] = 0 <= ... <= bounds[n] = nblocks)
 * Both must be called before the first vector of that length is created in the group (BMX_ERR_BADARG while vectors cut
 * with other borders are alive).  bmx_group_shard_range reports the cut in force. */
int bmx_block_table_weights(uint32_t nblocks, const uint8_t* kinds, const uint32_t* offs,
                            const uint16_t* gap_slab, uint64_t gap_words, uint64_t* weight);
int bmx_group_partition_by_weight(bmx_group* g, uint32_t nblocks, const uint64_t* weight, uint32_t* bounds_out);
int bmx_group_set_partition(bmx_group* g, uint32_t nblocks, const uint32_t* bounds);
/* ranks of the group's RCCL communicator as RCCL reports them (ncclCommCount); 0 for a host-sum group */
int bmx_group_rccl_ranks(const bmx_group* g, int* n);
/* bmx_vec_upload for a group: the block table is cut at the shard borders, every member receives its piece */
int bmx_gvec_upload(bmx_group* g, uint64_t nbits, uint32_t nblocks,
                    const uint8_t* kinds, const uint32_t* offs,
                    const uint32_t* bit_slab, uint32_t n_bit_blocks,
                    const uint16_t* gap_slab, uint64_t gap_words, bmx_gvec** out);
/* bmx_vec_generate for a group: every member generates its own block range (bmx_vec_generate_shard) */
int bmx_gvec_generate(bmx_group* g, uint64_t seed, uint32_t vec_id, int with_common,
                      uint32_t density_q16, uint64_t nbits, int optimize, bmx_gvec** out);
int bmx_gvec_free(bmx_group* g, bmx_gvec* v);
/* totals over the shards; bit_slab_blocks / gap_words size the buffers of bmx_gvec_download */
int bmx_gvec_info(const bmx_gvec* v, uint64_t* nbits, uint32_t* nblocks, uint32_t counts[4],
                  uint32_t* bit_slab_blocks, uint64_t* gap_words);
/* member m's shard (owned by the gvec; lives on bmx_group_ctx(g, m)) */
int bmx_gvec_shard(const bmx_gvec* v, int member, const bmx_vec** shard);
/* gathers the shards into ONE block table (same layout as bmx_vec_download) */
int bmx_gvec_download(bmx_group* g, const bmx_gvec* v, uint8_t* kinds, uint32_t* offs,
                      uint32_t* bit_slab, uint16_t* gap_slab);
/* bvector::count() / bm::count_* over all shards: n kernels in flight at once, one sum */
int bmx_gvec_count(bmx_group* g, const bmx_gvec* a, uint64_t* count);
int bmx_gvec_count_op2(bmx_group* g, int op, const bmx_gvec* a, const bmx_gvec* b, uint64_t* count);
/* bvector::bit_and/or/xor/sub (3-operand), result sharded like the operands */
int bmx_gvec_op2(bmx_group* g, int op, const bmx_gvec* a, const bmx_gvec* b, int opt_compress, bmx_gvec** result);
/* build_rs_index / count_to / select over a sharded vector (src/bm.h:2531,3120,5350): every member indexes its own
 * shard, the shard totals (n x 8 B) are scanned on the host, a query goes to the member that owns its block (rank)
 * or holds the rank-th one (select).  The index refers to `v`: free it before the vector. */
int bmx_grs_build(bmx_group* g, const bmx_gvec* v, bmx_grs** out);
int bmx_grs_free(bmx_group* g, bmx_grs* rs);
int bmx_grs_count(const bmx_grs* rs, uint64_t* count);
int bmx_grank_batch(bmx_group* g, const bmx_gvec* v, const bmx_grs* rs, const uint64_t* n, size_t q, uint64_t* out);
int bmx_gselect_batch(bmx_group* g, const bmx_gvec* v, const bmx_grs* rs, const uint64_t* rank, size_t q,
                      uint64_t* pos, uint8_t* found);
/* aggregator::combine_or / combine_and_sub over sharded vectors (src/bmaggregator.h:1101,1162) */
int bmx_gagg_or(bmx_group* g, const bmx_gvec* const* src, size_t n, int opt_compress, bmx_gvec** result);
int bmx_gagg_and_sub(bmx_group* g, const bmx_gvec* const* src_and, size_t n_and,
                     const bmx_gvec* const* src_sub, size_t n_sub, bmx_gvec** result, int* any);
/* aggregator::find_first_and_sub over sharded vectors (src/bmaggregator.h:1458): the hit of the lowest shard that has one */
int bmx_gfind_first_and_sub(bmx_group* g, const bmx_gvec* const* src_and, size_t n_and,
                            const bmx_gvec* const* src_sub, size_t n_sub, int* found, uint64_t* idx);
/* bmx_slice_compare over sharded bit-planes (scanner range search on several GPUs): planes, not_null and `size` must
 * span the same block range; result sharded like the planes; count = sum over the members */
int bmx_gslice_compare(bmx_group* g, const bmx_gvec* const* slices, size_t nslices, int pred, uint64_t v0, uint64_t v1,
                       uint64_t size, const bmx_gvec* not_null, bmx_gvec** result, uint64_t* count);
/* bmx_slice_eq_counts over sharded bit-planes (same conditions as bmx_gslice_compare) */
int bmx_gslice_eq_counts(bmx_group* g, const bmx_gvec* const* slices, size_t nslices, const uint64_t* values, size_t n,
                         uint64_t size, const bmx_gvec* not_null, uint64_t* counts);
/* aggregator::pipeline + combine_and_sub(pipe), counts only (src/bmaggregator.h:1292-1399): member m runs the
 * pipeline over its shard of every operand, counts_out[g] = sum over the members */
int bmx_gpipeline_create(bmx_group* g,
                         const bmx_gvec* const* and_list, const uint32_t* and_n,
                         const bmx_gvec* const* sub_list, const uint32_t* sub_n,
                         size_t ngroups, bmx_gpipeline** out);
int bmx_gpipeline_destroy(bmx_group* g, bmx_gpipeline* p);
int bmx_gpipeline_run_counts(bmx_group* g, bmx_gpipeline* p, uint64_t* counts_out);
/* pipeline::set_search_count_limit over shards (src/bmaggregator.h:255,1365): every member searches its shard under the same
 * limit (ascending launch windows, see bmx_pipeline_set_search_count_limit), the counts are summed: counts_out[g] is
 * >= min(limit, the group's true count) and <= the true count.  limit 0 / UINT64_MAX = no limit. */
int bmx_gpipeline_set_search_count_limit(bmx_group* g, bmx_gpipeline* p, uint64_t limit);
/* bmx_collection_prepare over shards: member m transposes its block range of the vectors; the aggregations and pipelines of
 * the group then use it member by member (roles as BMX_ROLE_*) */
int bmx_gcollection_prepare(bmx_group* g, const bmx_gvec* const* vecs, size_t n, int role);
/* per-member device time of the last bmx_gpipeline_run_counts (HIP events on the member streams), ms[n] */
int bmx_gpipeline_last_ms(bmx_group* g, const bmx_gpipeline* p, float* ms);
/* per-member device time between the end of the kernel and the end of the exchange (RCCL all-reduce, or the 8-byte
 * copy to the host) of the last run, ms[n] */
int bmx_gpipeline_last_exchange_ms(bmx_group* g, const bmx_gpipeline* p, float* ms);
/* bmx_pipeline_operand_bytes / bmx_pipeline_describe per member (benchmark reports) */
int bmx_gpipeline_operand_bytes(bmx_group* g, bmx_gpipeline* p, uint64_t* bytes_per_member);
int bmx_gpipeline_describe(bmx_group* g, bmx_gpipeline* p, int member, char* buf, size_t buf_len, uint32_t* n_launches);

/* ---- debug aids (no reference twin) ----
 * Red zones: a context created while BMX_DEBUG_REDZONE=1 is set surrounds every device allocation of the library (slabs, results,
 * indexes, collections, tables, scratch) with 4 KiB of a canary pattern -- in front, and from the end of the requested bytes to
 * the end of the block -- and verifies them when the block is freed, at bmx_ctx_synchronize (which then returns BMX_ERR_DEVICE
 * with the report as bmx_last_error), here, and when the context is destroyed.  hits = damaged allocations found so far;
 * report = one line per damaged allocation naming the line of bmx.hip that made it. */
int bmx_debug_redzone_check(bmx_ctx* ctx, int* enabled, uint64_t* hits, char* report, size_t report_len);
/* Fault injection for tests of the error paths.  kind 1 | 2 | 3: the library entry `after` calls from now on this thread throws
 * std::bad_alloc | std::length_error | a non-standard exception at its first statement -- what must come back is a status
 * (BMX_ERR_BADALLOC | BMX_ERR_DEVICE | BMX_ERR_DEVICE), never an exception (libbm.cpp:28-35); ctx may be NULL.  kind 4: the
 * device allocation `after` allocations from now on ctx fails with BMX_ERR_BADALLOC.  kind 5 (red-zone contexts): writes one byte
 * behind a fresh allocation -- the checker's self-test.  kind 0: disarm. */
int bmx_debug_inject_failure(bmx_ctx* ctx, int kind, long long after);

/* ---- timing helper: HIP events on the context's stream ---- */
int bmx_timer_start(bmx_ctx* ctx);
int bmx_timer_stop_ms(bmx_ctx* ctx, float* ms);   /* synchronises on the stop event */

/* measurement helper: ms of one pass of `nlines` random 128-byte-line reads (8 lanes x 16 B per line, the access shape
 * of a rank query's bit line) over a scratch buffer of buf_bytes -- the gather ceiling that bench.py --config 3 divides
 * the rank / select rates by (random access is bound by the HBM transaction rate) */
int bmx_probe_random_lines(bmx_ctx* ctx, uint64_t buf_bytes, uint64_t nlines, int iters, float* ms_per_pass);

/* measurement helper: ms of one pass of c = a & b over three buffers of `bytes` each (rotating over `sets` triples so that the
 * Infinity Cache serves nothing) in the launch shape of the streaming pairwise kernel -- a wave per stretch of 8-KiB blocks,
 * wgs_per_cu workgroups of 4 waves per CU, non-temporal 16-byte loads and stores: what this box gives a 2-read : 1-write
 * stream, the yardstick bench.py --config 1 reports next to bvector::bit_and/or/xor/sub (src/bm.h:6185) */
int bmx_probe_stream_rw(bmx_ctx* ctx, uint64_t bytes, int sets, int wgs_per_cu, int iters, float* ms_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* BMX_H */
