/* bmx_diag.h -- diagnostics exported ONLY by the tuning build of the library
 * (make -C bitmagic_amd/csrc tune -> lib/libbmx_tune.so, -DBMX_TUNE -DBMX_DIAG); not part of the product ABI. */
#pragma once
#include "../../include/bmx.h"
#ifdef __cplusplus
extern "C" {
#endif
/* plain streaming read of a scratch buffer: the practical HBM read ceiling of this box,
 * to put next to the product kernels (tools/tune_pipe.py).  ms_per_pass = avg of iters passes. */
int bmx_diag_stream_read(bmx_ctx* ctx, uint64_t bytes, int nontemporal, uint32_t blocks_per_wave,
                         int pattern /* 0 contiguous per wave, 1 strided like an N-way aggregation */,
                         int iters, float* ms_per_pass);
#ifdef __cplusplus
}
#endif
