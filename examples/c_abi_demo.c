/* Plain C99 client of the C-ABI (include/bmx.h -> libbmx.so): what a C / cgo / JNI binding sees.
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lbitmagic_amd/lib -lbmx -Wl,-rpath,$PWD/bitmagic_amd/lib -o /tmp/c_abi_demo
 * Two one-block vectors are handed over as flat block tables (kinds / offsets / bit slab), AND-ed, counted,
 * ranked; every call returns an int status like lang-maps/libbm (BM_OK = 0). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bmx.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != BMX_OK) { \
    fprintf(stderr, "%s -> %s [%s]\n", #call, bmx_error_msg(rc_), bmx_last_error()); return 1; } } while (0)

int main(void)
{
    bmx_ctx* ctx = NULL;
    CHECK(bmx_ctx_create(0, NULL, &ctx));
    uint32_t* a = (uint32_t*)calloc(BMX_BLOCK_WORDS, 4);
    uint32_t* b = (uint32_t*)calloc(BMX_BLOCK_WORDS, 4);
    for (unsigned i = 0; i < BMX_BLOCK_WORDS; ++i) { a[i] = 0x0F0F0F0Fu; b[i] = (i & 1u) ? 0xFFFFFFFFu : 0u; }
    uint8_t kind = BMX_BIT; uint32_t off = 0;
    bmx_vec *va = NULL, *vb = NULL, *vt = NULL;
    CHECK(bmx_vec_upload(ctx, BMX_BLOCK_BITS, 1, &kind, &off, a, 1, NULL, 0, &va));
    CHECK(bmx_vec_upload(ctx, BMX_BLOCK_BITS, 1, &kind, &off, b, 1, NULL, 0, &vb));
    uint64_t ca = 0, cand = 0, ct = 0;
    CHECK(bmx_count(ctx, va, &ca));
    CHECK(bmx_count_op2(ctx, BMX_AND, va, vb, &cand));
    CHECK(bmx_op2(ctx, BMX_AND, va, vb, 1, &vt));
    CHECK(bmx_count(ctx, vt, &ct));
    bmx_rs* rs = NULL;
    CHECK(bmx_rs_build(ctx, vt, &rs));
    uint64_t q = BMX_BLOCK_BITS - 1, r = 0;
    CHECK(bmx_rank_batch(ctx, vt, rs, &q, 1, &r));
    printf("count(a)=%llu count_and=%llu count(a&b)=%llu rank(last)=%llu\n",
           (unsigned long long)ca, (unsigned long long)cand, (unsigned long long)ct, (unsigned long long)r);
    int ok = ca == 32768u && cand == 16384u && ct == cand && r == ct;
    {   /* the same through a device GROUP (here: device 0 listed twice = two shards on one GPU; {0,1,...,7} on a node):
           vectors are sharded by block range, the only exchange is the sum of the counts */
        int devs[2] = {0, 0};
        bmx_group* grp = NULL;
        bmx_gvec *ga = NULL, *gb = NULL;
        uint8_t kinds[2] = {BMX_BIT, BMX_BIT}; uint32_t offs[2] = {0, 0};          /* two blocks, both = the same 8 KiB */
        uint64_t gc = 0, gand = 0, sel = 0; uint8_t found = 0; uint64_t one = 1;
        bmx_grs* grs = NULL;
        CHECK(bmx_group_create(devs, 2, BMX_GROUP_HOST_SUM, &grp));
        CHECK(bmx_gvec_upload(grp, 2 * BMX_BLOCK_BITS, 2, kinds, offs, a, 1, NULL, 0, &ga));
        CHECK(bmx_gvec_upload(grp, 2 * BMX_BLOCK_BITS, 2, kinds, offs, b, 1, NULL, 0, &gb));
        CHECK(bmx_gvec_count(grp, ga, &gc));
        CHECK(bmx_gvec_count_op2(grp, BMX_AND, ga, gb, &gand));
        CHECK(bmx_grs_build(grp, gb, &grs));
        CHECK(bmx_gselect_batch(grp, gb, grs, &one, 1, &sel, &found));               /* first set bit of b: bit 32 */
        printf("group: count(a)=%llu count_and=%llu select(1)=%llu\n", (unsigned long long)gc, (unsigned long long)gand, (unsigned long long)sel);
        ok = ok && gc == 2 * ca && gand == 2 * cand && found && sel == 32u;
        bmx_grs_free(grp, grs);
        bmx_gvec_free(grp, ga); bmx_gvec_free(grp, gb);
        bmx_group_destroy(grp);
    }
    bmx_rs_free(ctx, rs);
    bmx_vec_free(ctx, vt); bmx_vec_free(ctx, va); bmx_vec_free(ctx, vb);
    bmx_ctx_destroy(ctx);
    free(a); free(b);
    puts(ok ? "c_abi_demo ok" : "c_abi_demo MISMATCH");
    return ok ? 0 : 1;
}
