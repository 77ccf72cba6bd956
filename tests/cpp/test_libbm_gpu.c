/* Plain C client of BitMagic's C wrapper (lang-maps/libbm) using the GPU edition of its pairwise / count surface
 * (examples/libbm_gpu.cpp): every BMX_ call must agree with its BM_ twin on the same handles. */
#include <stdio.h>
#include <stdlib.h>
#include "libbm.h"

int BMX_bvector_count(BM_BVHANDLE h, unsigned int* pcount);
int BMX_bvector_count_AND(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p);
int BMX_bvector_count_OR(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p);
int BMX_bvector_count_XOR(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p);
int BMX_bvector_count_SUB(BM_BVHANDLE h1, BM_BVHANDLE h2, unsigned int* p);
int BMX_bvector_combine_AND(BM_BVHANDLE hdst, BM_BVHANDLE hsrc);
int BMX_bvector_combine_OR(BM_BVHANDLE hdst, BM_BVHANDLE hsrc);
int BMX_bvector_combine_XOR(BM_BVHANDLE hdst, BM_BVHANDLE hsrc);
int BMX_bvector_combine_SUB(BM_BVHANDLE hdst, BM_BVHANDLE hsrc);
int BMX_bvector_invalidate(BM_BVHANDLE h);
int BMX_bvector_release(BM_BVHANDLE h);
int BMX_simd_version(void);

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(void)
{
    BM_BVHANDLE a = 0;                                  /* BM_BVHANDLE is a macro for void*: one declaration per handle */
    BM_BVHANDLE b = 0;
    BM_BVHANDLE c = 0;
    BM_BVHANDLE d = 0;
    unsigned int i, x = 2463534242u, r0, r1;
    int cmp = 1;
    REQUIRE(BM_init(0) == BM_OK && BMX_simd_version() == 950);
    REQUIRE(BM_bvector_construct(&a, 0) == BM_OK && BM_bvector_construct(&b, 0) == BM_OK);
    for (i = 0; i < 400000; ++i) {                      /* a: random bits (bit-blocks), b: runs + sparse bits (GAP blocks) */
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        REQUIRE(BM_bvector_set_bit(a, x % 6000000u, BM_TRUE) == BM_OK);
        if ((i & 7u) == 0) REQUIRE(BM_bvector_set_bit(b, (x >> 3) % 9000000u, BM_TRUE) == BM_OK);
    }
    for (i = 1000000; i < 1300000; ++i) REQUIRE(BM_bvector_set_bit(b, i, BM_TRUE) == BM_OK);
    REQUIRE(BM_bvector_optimize(b, 3, 0) == BM_OK);
    REQUIRE(BM_bvector_count(a, &r0) == BM_OK && BMX_bvector_count(a, &r1) == BM_OK && r0 == r1 && r0 > 300000u);
    REQUIRE(BM_bvector_count_AND(a, b, &r0) == BM_OK && BMX_bvector_count_AND(a, b, &r1) == BM_OK && r0 == r1);
    REQUIRE(BM_bvector_count_OR(a, b, &r0) == BM_OK && BMX_bvector_count_OR(a, b, &r1) == BM_OK && r0 == r1);
    REQUIRE(BM_bvector_count_XOR(a, b, &r0) == BM_OK && BMX_bvector_count_XOR(a, b, &r1) == BM_OK && r0 == r1);
    REQUIRE(BM_bvector_count_SUB(a, b, &r0) == BM_OK && BMX_bvector_count_SUB(a, b, &r1) == BM_OK && r0 == r1);
    /* dst OP= src on copies: CPU edition on c, GPU edition on d */
    {
        int (*cpu[4])(BM_BVHANDLE, BM_BVHANDLE) = {BM_bvector_combine_AND, BM_bvector_combine_OR, BM_bvector_combine_XOR, BM_bvector_combine_SUB};
        int (*gpu[4])(BM_BVHANDLE, BM_BVHANDLE) = {BMX_bvector_combine_AND, BMX_bvector_combine_OR, BMX_bvector_combine_XOR, BMX_bvector_combine_SUB};
        int op;
        for (op = 0; op < 4; ++op) {
            REQUIRE(BM_bvector_construct_copy(&c, a) == BM_OK && BM_bvector_construct_copy(&d, a) == BM_OK);
            REQUIRE(cpu[op](c, b) == BM_OK && gpu[op](d, b) == BM_OK);
            REQUIRE(BM_bvector_compare(c, d, &cmp) == BM_OK && cmp == 0);
            REQUIRE(BM_bvector_count(c, &r0) == BM_OK && BMX_bvector_count(d, &r1) == BM_OK && r0 == r1);
            BMX_bvector_release(c); BMX_bvector_release(d);
            BM_bvector_free(c); BM_bvector_free(d); c = d = 0;
        }
    }
    /* mutate through the CPU API, invalidate, ask the GPU again */
    REQUIRE(BM_bvector_set_bit(a, 7777777u, BM_TRUE) == BM_OK && BMX_bvector_invalidate(a) == BM_OK);
    REQUIRE(BM_bvector_count(a, &r0) == BM_OK && BMX_bvector_count(a, &r1) == BM_OK && r0 == r1);
    REQUIRE(BMX_bvector_count_AND(0, b, &r1) == BM_ERR_BADARG);
    BMX_bvector_release(a); BMX_bvector_release(b);
    BM_bvector_free(a); BM_bvector_free(b);
    printf("test_libbm_gpu ok\n");
    return 0;
}
