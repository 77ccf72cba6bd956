/*
 * bmx_oracle.h -- CPU restatement of the BitMagic bvector/aggregator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The shipped path (bitmagic_amd/, include/)
 * never links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is validated in the build container
 * against the reference itself (oracle/ref_shim.cpp compiled against
 * /root/reference/src, scalar and -DBMAVX2OPT) and against the golden fixtures
 * under tests/golden/ that the reference generated (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line (relative to /root/reference/)
 * whose behaviour it restates.  Plain C99, no dependencies.
 */
#ifndef BMX_ORACLE_H
#define BMX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry (src/bmconst.h:55-87,115-124) ---- */
#define BMO_BLOCK_WORDS   2048u      /* set_block_size: 32-bit words per bit-block   */
#define BMO_BLOCK_BITS    65536u     /* gap_max_bits / bits_in_block                 */
#define BMO_WAVE_WORDS    32u        /* set_block_digest_wave_size (1024 bits)       */
#define BMO_GAP_MAX_LEN   1280u      /* gap_max_buff_len                             */
#define BMO_GAP_THRESHOLD 1276u      /* glen(gap_max_level)-4, src/bmblocks.h:1388   */
#define BMO_RS3_BORDER0   21824u
#define BMO_RS3_BORDER1   43648u
#define BMO_RS3_HALF_SPAN 10912u

/* block kinds of the flat block table (what a tagged pointer encodes in the
 * reference: src/bmdef.h:165-199) */
enum { BMO_NULL = 0, BMO_FULL = 1, BMO_BIT = 2, BMO_GAP = 3 };
/* set operations (src/bmconst.h set_operation subset) */
enum { BMO_AND = 0, BMO_OR = 1, BMO_XOR = 2, BMO_SUB = 3 };

typedef struct bmo_vec bmo_vec;

/* ---- synthetic input generator (normative spec shared with the HIP kernel;
 *      SURVEY.md section 8d).  Output word index is in 64-bit words. ---- */
uint64_t bmo_gen_word64(uint64_t seed, uint32_t vec_id, uint64_t w64, uint32_t density_q16);
/* fills nwords32 32-bit words starting at 32-bit word offset word_off (must be even);
 * bits at positions >= nbits (absolute) are zero; common_id = 0xFFFFFFFF.. see .c */
void bmo_gen_words(uint64_t seed, uint32_t vec_id, int with_common, uint32_t density_q16,
                   uint64_t nbits, uint64_t word_off, uint64_t nwords32, uint32_t* out);

/* ---- block primitives (src/bmfunc.h, table in SURVEY.md section 2c) ---- */
uint32_t bmo_bit_block_count(const uint32_t* blk);
uint32_t bmo_bit_block_count_digest(const uint32_t* blk, uint64_t digest);
uint64_t bmo_calc_block_digest0(const uint32_t* blk);
uint32_t bmo_bit_block_calc_change(const uint32_t* blk);
uint32_t bmo_bit_block_count_range(const uint32_t* blk, uint32_t left, uint32_t right);
unsigned bmo_bit_to_gap(uint16_t* dest, const uint32_t* blk);   /* returns len */
void     bmo_gap_convert_to_bitset(uint32_t* dest, const uint16_t* gap);
uint32_t bmo_gap_bit_count(const uint16_t* gap);
unsigned bmo_gap_test(const uint16_t* gap, uint32_t pos);
unsigned bmo_gap_bfind(const uint16_t* gap, uint32_t pos, unsigned* is_set);
uint32_t bmo_gap_bit_count_to(const uint16_t* gap, uint32_t right);
unsigned bmo_gap_op(int op, const uint16_t* a, const uint16_t* b, uint16_t* dest, unsigned dest_cap);
uint64_t bmo_and_2way(uint32_t* dst, const uint32_t* s1, const uint32_t* s2, uint64_t digest);
uint64_t bmo_and_5way(uint32_t* dst, const uint32_t* s0, const uint32_t* s1,
                      const uint32_t* s2, const uint32_t* s3, uint64_t digest);
uint64_t bmo_gap_and_to_bitset_digest(uint32_t* dst, const uint16_t* gap, uint64_t digest);
uint64_t bmo_gap_sub_to_bitset_digest(uint32_t* dst, const uint16_t* gap, uint64_t digest);

/* ---- vectors (flat block table mirror of bm::bvector<>, src/bm.h) ---- */
bmo_vec* bmo_vec_new(uint64_t nbits);
bmo_vec* bmo_vec_import(const uint32_t* words, uint64_t nwords, int optimize); /* src/bmbvimport.h:46 */
bmo_vec* bmo_vec_from_table(uint64_t nbits, uint32_t nblocks, const uint8_t* kinds,
                            const uint32_t* offs, const uint32_t* bit_slab,
                            const uint16_t* gap_slab);
void     bmo_vec_free(bmo_vec* v);
uint64_t bmo_vec_nbits(const bmo_vec* v);
uint32_t bmo_vec_nblocks(const bmo_vec* v);
/* counts[4] = number of NULL/FULL/BIT/GAP blocks; gap_words = sum of (len+1) */
void     bmo_vec_stat(const bmo_vec* v, uint32_t counts[4], uint64_t* gap_words);
/* flatten: kinds[nblocks]; offs[nblocks] = index of the block inside its slab
 * (bit slab: block ordinal; gap slab: u16 word offset); slabs sized per bmo_vec_stat */
void     bmo_vec_flatten(const bmo_vec* v, uint8_t* kinds, uint32_t* offs,
                         uint32_t* bit_slab, uint16_t* gap_slab);
void     bmo_vec_to_words(const bmo_vec* v, uint32_t* out, uint64_t nwords);
void     bmo_vec_optimize(bmo_vec* v);                      /* optimize(opt_compress) */
int      bmo_vec_equal(const bmo_vec* a, const bmo_vec* b); /* logical content, representation-agnostic */
uint64_t bmo_vec_count(const bmo_vec* v);                   /* src/bm.h:2431 */
int      bmo_vec_get_bit(const bmo_vec* v, uint64_t n);
void     bmo_vec_set_bit(bmo_vec* v, uint64_t n);           /* test helper (KATs) */
void     bmo_vec_set_range(bmo_vec* v, uint64_t l, uint64_t r);

/* pairwise (src/bm.h:6185,5973,6072,6403 and Appendix A.1) */
bmo_vec* bmo_op2(int op, const bmo_vec* a, const bmo_vec* b, int opt_compress);
/* count_and/or/xor/sub (src/bmalgo.h:49,149,81,115; Appendix A.2) */
uint64_t bmo_count_op2(int op, const bmo_vec* a, const bmo_vec* b);

/* aggregator (src/bmaggregator.h:1101,1162 ; Appendix A.3) */
bmo_vec* bmo_agg_or(const bmo_vec* const* src, size_t n);                      /* opt_mode_ = opt_none (:917) */
bmo_vec* bmo_agg_or_opt(const bmo_vec* const* src, size_t n, int opt_compress);  /* after set_optimization (:359) */
bmo_vec* bmo_agg_and_sub(const bmo_vec* const* src_and, size_t n_and,
                         const bmo_vec* const* src_sub, size_t n_sub);
/* aggregator::combine_shift_right_and (src/bmaggregator.h:2494-2669): T_0 = src[0],
 * T_k = shift_right_1(T_{k-1}) & src[k] with the carry crossing block borders; stored with opt mode
 * opt_compress (0 = opt_none, the aggregator default); any: stop at the first result block.
 * _count: set_compute_count(true) form (:363,2593). */
bmo_vec* bmo_agg_shift_right_and(const bmo_vec* const* src, size_t n, int opt_compress, int any, int* found);
uint64_t bmo_agg_shift_right_and_count(const bmo_vec* const* src, size_t n);
/* counts-only pipeline (src/bmaggregator.h:1292-1399): groups given as
 * concatenated pointer lists; and_n[g]/sub_n[g] operands per group.
 * block range [nb_from, nb_to) restricts the columns visited (shard support). */
void     bmo_agg_pipeline_counts(const bmo_vec* const* and_list, const uint32_t* and_n,
                                 const bmo_vec* const* sub_list, const uint32_t* sub_n,
                                 size_t ngroups, uint32_t nb_from, uint32_t nb_to,
                                 uint64_t* counts_out);

/* pipeline<agg_opt_bvect_and_counts> + OR target (src/bmaggregator.h:1292-1449): per group a result
 * vector (NULL when the group found nothing, :1406-1415), its count, and the OR of all results
 * (optimised, :1440-1447) */
void     bmo_agg_pipeline_results(const bmo_vec* const* and_list, const uint32_t* and_n,
                                  const bmo_vec* const* sub_list, const uint32_t* sub_n, size_t ngroups,
                                  bmo_vec** results_out, uint64_t* counts_out, bmo_vec** or_target_out);

/* first set bit of a vector (bvector::find) and of an AND-SUB aggregation without materialising
 * (aggregator::find_first_and_sub, src/bmaggregator.h:1458).  Logical definition; the reference
 * additionally narrows the searched sub-array range by the SUB group (its own ":1526 TODO"). */
int      bmo_vec_find_first(const bmo_vec* v, uint64_t* pos);
int      bmo_find_first_and_sub(const bmo_vec* const* src_and, size_t n_and,
                                const bmo_vec* const* src_sub, size_t n_sub, uint64_t* idx);

/* rank / select (src/bm.h:3120,5350 ; Appendix A.5) */
typedef struct bmo_rs bmo_rs;
bmo_rs*  bmo_rs_build(const bmo_vec* v);                    /* src/bm.h:2531 */
void     bmo_rs_free(bmo_rs* rs);
uint64_t bmo_rs_count(const bmo_rs* rs);
uint32_t bmo_rs_total_blocks(const bmo_rs* rs);
/* reference-compatible per-block arrays: bcount[nb], sub_count[nb] (packing src/bm.h:2646-2656) */
void     bmo_rs_export(const bmo_rs* rs, uint32_t* bcount, uint64_t* sub_count);
uint64_t bmo_rank(const bmo_vec* v, const bmo_rs* rs, uint64_t n);          /* count_to */
int      bmo_select(const bmo_vec* v, const bmo_rs* rs, uint64_t rank, uint64_t* pos);
/* src/bm.h:3548 count_range, :3229 rank_corrected, :3173 count_to_test, :5279 find_rank(rank, from) */
uint64_t bmo_count_range(const bmo_vec* v, const bmo_rs* rs, uint64_t left, uint64_t right);
uint64_t bmo_rank_corrected(const bmo_vec* v, const bmo_rs* rs, uint64_t n);
uint64_t bmo_count_to_test(const bmo_vec* v, const bmo_rs* rs, uint64_t n);
int      bmo_find_rank(const bmo_vec* v, const bmo_rs* rs, uint64_t rank, uint64_t from, uint64_t* pos);
void     bmo_rank_batch(const bmo_vec* v, const bmo_rs* rs, const uint64_t* n, size_t q, uint64_t* out);
void     bmo_select_batch(const bmo_vec* v, const bmo_rs* rs, const uint64_t* r, size_t q,
                          uint64_t* pos, uint8_t* found);

#ifdef __cplusplus
}
#endif
#endif
