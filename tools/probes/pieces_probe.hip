// pieces_probe.hip -- what bounds "N streams read in P-byte pieces" on MI355X?  (round 4)
//
// The descriptor-table combine_or kernel (k_agg_or_gap_tiled) visits each of 4096 operand slabs in ~1-KiB pieces
// (16 block columns x ~56 B of GAP data) and was measured at 4.4 TB/s with the run application compiled out.  This
// stand-alone probe reproduces ONLY the access pattern, so that the candidates can be told apart:
//   alloc  = sep   : every stream its own hipMalloc (what bmx_vec_upload / generate did through round 3)
//            arena : all streams carved from ONE hipMalloc (stride = stream bytes rounded to 256 B)
//            pow2  : one hipMalloc, stride exactly 4 MiB (channel / bank aliasing test)
//   piece  = bytes one workgroup reads from one stream per visit (1 KiB = one dwordx4 wave load)
//   wg     = threads per workgroup; lds = dynamic LDS bytes requested (131072 -> one workgroup per CU, as the tiled kernel)
// Work item = one piece index; a workgroup walks ALL streams for its piece index (wave w takes streams w, w + W, ...),
// DEPTH wave loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 pieces_probe.hip -o ../bin/pieces_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr;

template <int DEPTH, int LPP, int WORK = 0>   // WORK: 0 loads only, 1 + the row arithmetic of k_agg_or_rows, 2 + its LDS atomics; LPP = 16-byte loads per lane per piece (piece = LPP KiB)
__global__ void k_pieces(const u64* __restrict__ bases, u32 nstreams, u32 npieces, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u32 piece = blockIdx.x;
    if (piece >= npieces) return;
    const u64 off = (u64)piece * (LPP * 1024u) + lane * 16u;
    u32x4 acc = (u32x4)(0u);
    u64 work_mask = 0x1111111111111111ull;
    u32 s = wave;
    for (; s + (DEPTH - 1) * W < nstreams; s += DEPTH * W) {
        u32x4 v[DEPTH][LPP];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const u64 bv = bases[s + d * W];
            u64 b = (u64)(u32)__builtin_amdgcn_readfirstlane((u32)bv) | ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(bv >> 32)) << 32);
            gptr p = (gptr)(b + off);
#pragma unroll
            for (int l = 0; l < LPP; ++l) v[d][l] = __builtin_nontemporal_load(p + l * 64);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int l = 0; l < LPP; ++l) acc ^= v[d][l];
            if (WORK) {                                  // what k_agg_or_rows does with a row, without its records: pairs -> LDS atomics (2) or VALU only (1)
                const u32 nx = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v[d][0].x, 0x130, 0xf, 0xf, false);
                const u32 x[5] = {v[d][0].x, v[d][0].y, v[d][0].z, v[d][0].w, nx};
                const u32 col = (u32)__popcll(work_mask & ((2ull << lane) - 1ull)) % 14u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 y = __builtin_amdgcn_alignbit(x[i + 1], x[i], 16);
                    const u32 e = y >> 16;
                    const u32 bit = (y & 0xFFFFu) != 0xFFFFu ? 1u << (e & 31u) : 0u;
                    if (WORK == 2) atomicOr(&lds[col * 2048u + (e >> 5)], bit);
                    else acc.x += bit * (e >> 5);
                }
                work_mask = work_mask * 6364136223846793005ull + 1442695040888963407ull + v[d][0].x * 0;   // (uniform: a new start mask per row)
            }
        }
    }
    for (; s < nstreams; s += W) {
        gptr p = (gptr)(bases[s] + off);
#pragma unroll
        for (int l = 0; l < LPP; ++l) acc ^= __builtin_nontemporal_load(p + l * 64);
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

// Round 6: would GAP slabs on a 4-byte grid pay?  A configs[4] block is 2 (27 + 1) = 56 B; on the 16-byte grid of the
// slabs it occupies 64 B, so a tile's row (14 blocks) is 896 B of which 112 B are padding.  This kernel reads rows of P bytes
// packed back to back (P = 784: no padding; P = 896: today's bytes) that start on an arbitrary 4-byte boundary (jitter: the real
// blocks have random lengths), either as 16-byte chunks from the aligned-down start (mode 0: what a kernel would do, shifting by
// the row's phase afterwards) or as unaligned 16-byte loads from the row's own start (mode 1).  Same walk as k_pieces otherwise.
template <int DEPTH>
__global__ void k_rows_var(const u64* __restrict__ bases, u32 nstreams, u32 npieces, u32 P, u32 mode, u32 jitter, u32 S, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = (u32)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;     // (uniform: the bases come by scalar loads)
    u32 piece = blockIdx.x;
    if (mode & 256u) {                                   // xcd_remap of the library: every XCD takes one contiguous slice of the rows
        const u32 q = npieces >> 3, rem = npieces & 7u, x = piece & 7u, i = piece >> 3;
        piece = x * q + (x < rem ? x : rem) + i;
        mode &= 255u;
    }
    if (piece >= npieces) return;
    u32x4 acc = (u32x4)(0u);
    u32 s = wave;
    for (; s + (DEPTH - 1) * W < nstreams; s += DEPTH * W) {
        u32x4 v[DEPTH]; u64 bvs[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) bvs[d] = bases[s + d * W];            // (all bases first: one wait, then DEPTH rows in flight)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const u64 bv = bvs[d];
            const u64 b = (u64)(u32)__builtin_amdgcn_readfirstlane((u32)bv) | ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(bv >> 32)) << 32);
            const u32 h = ((s + d * W) * 2654435761u) ^ (piece * 40503u);
            // byte offset of the row inside the stream.  jitter 1: a 4-byte phase (0 .. 12); 2: a 16-byte phase (0 .. 112: today's rows start
            // on any 16-byte boundary); 3: 2 + the row's length varies by -32 .. +32 bytes (blocks of 48 / 64 / 80 bytes)
            // jitter 5: TILE-ALIGNED rows -- every row starts on a 128-byte line, its length varies by -32 .. +32 bytes and two rows in five
            // take another line (strides of 896 and 1024 bytes alternate): what padding the slab at tile boundaries would give
            const u64 start = jitter == 5u ? (u64)piece * S + 128ull * ((2u * piece) / 5u)
                                           : (u64)piece * S + (jitter == 1u ? ((h >> 7) & 3u) * 4u : jitter >= 2u ? ((h >> 7) & 7u) * 16u : 0u);
            const u32 Pv = (jitter == 3u || jitter == 5u) ? P + ((h >> 12) % 5u) * 16u - 32u : P;
            const u64 first = mode == 0 ? (start & ~15ull) : start;
            const u64 end = start + Pv;
            const u64 a = first + lane * 16u;
            v[d] = (u32x4)(0u);
            if (jitter == 4u) v[d] = __builtin_nontemporal_load((gptr)(b + (a < end ? a : end - 16u)));
            else if (a < end) v[d] = __builtin_nontemporal_load((gptr)(b + a));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

typedef void (*kfn)(const u64*, u32, u32, u32*);
static kfn pick(int depth, int lpp, int work)
{
    if (work == 1 && lpp == 1) return depth == 8 ? k_pieces<8, 1, 1> : k_pieces<4, 1, 1>;
    if (work == 2 && lpp == 1) return depth == 8 ? k_pieces<8, 1, 2> : k_pieces<4, 1, 2>;
    if (work) return nullptr;
#define K(D, L) if (depth == D && lpp == L) return k_pieces<D, L>;
    K(1, 1) K(2, 1) K(4, 1) K(8, 1) K(1, 2) K(2, 2) K(4, 2) K(1, 4) K(2, 4) K(4, 4) K(1, 8) K(2, 8)
#undef K
    return nullptr;
}

__global__ void k_fill_random(u64* p, u64 n, u64 seed)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 z = (i + seed) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; p[i] = z ^ (z >> 31);
    }
}

int main(int argc, char** argv)
{
    u32 nstreams = 4096; u64 stream_bytes = 3418016;    // configs[4]: 61,036 GAP blocks x ~56 B
    if (argc > 1) nstreams = (u32)atoi(argv[1]);
    if (argc > 2) stream_bytes = strtoull(argv[2], nullptr, 10);
    CHK(hipSetDevice(0));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    u32* sink; CHK(hipMalloc(&sink, 64));
    u64* d_bases; CHK(hipMalloc(&d_bases, (size_t)nstreams * 8));
    if (argc > 6) {                                      // argv[6] = "rows": the round-6 question (see k_rows_var)
        std::vector<void*> owned; std::vector<u64> bases(nstreams);
        const u64 alloc_bytes = ((stream_bytes / 784u + 2u) * 1100u + (2u << 20)) / (2u << 20) * (2u << 20);     // (rows of up to 1 KiB)
        for (u32 s = 0; s < nstreams; ++s) { void* p; CHK(hipMalloc(&p, alloc_bytes)); owned.push_back(p); bases[s] = (u64)(uintptr_t)p;
            hipLaunchKernelGGL(k_fill_random, dim3(256), dim3(256), 0, st, (u64*)p, alloc_bytes / 8, (u64)(uintptr_t)p); }
        CHK(hipMemcpy(d_bases, bases.data(), (size_t)nstreams * 8, hipMemcpyHostToDevice));
        CHK(hipStreamSynchronize(st));
        const u32 nrows = (u32)(stream_bytes / 784u);                             // rows of 14 blocks: the same number either way
        struct V { u32 P, mode, jitter; const char* what; u32 S = 0, load = 0; };
        const V vs[] = {{896, 0, 0, "16-byte grid, every row 896 B and line-aligned"}, {896, 0, 2, "16-byte grid (today): rows of 896 B starting on any 16-byte boundary"},
                        {896, 0, 3, "16-byte grid (today): rows of 864 .. 928 B starting on any 16-byte boundary"},
                        {784, 0, 1, "4-byte grid: rows of 784 B, chunks from the aligned-down start"},
                        {784, 1, 1, "4-byte grid: rows of 784 B, unaligned 16-byte loads"}, {784, 0, 0, "rows of 784 B that happen to start 16-byte aligned"},
                        {800, 0, 2, "blocks packed on a 4-byte grid inside a tile, tiles on the 16-byte grid: rows of 800 B on any 16-byte boundary"},
                        {896, 0, 0, "stride 896, 1024 B loaded (all 64 lanes: the row and 128 B of the next)", 896, 1024}, {896, 0, 0, "stride 896, 880 B loaded (55 lanes)", 896, 880},
                        {896, 0, 0, "stride 896, 912 B loaded (57 lanes)", 896, 912}, {896, 0, 0, "stride 960, 896 B loaded (56 lanes)", 960, 896}, {896, 0, 0, "stride 1024, 896 B loaded (56 lanes)", 1024, 896},
                        {896, 0, 0, "stride 832, 896 B loaded (56 lanes, rows overlap)", 832, 896}, {896, 0, 4, "stride 896, 896 B loaded by 64 lanes (lanes past the row repeat its last chunk: what k_agg_or_rows does)", 896, 896},
                        {896, 256, 0, "XCD-contiguous rows (xcd_remap): stride 896, 896 B loaded", 896, 896}, {896, 256, 3, "XCD-contiguous rows: stride 896, 864 .. 928 B loaded on any 16-byte boundary", 896, 896},
                        {896, 256, 0, "XCD-contiguous rows: stride 1024, 896 B loaded", 1024, 896}, {896, 256, 0, "XCD-contiguous rows: stride 960, 896 B loaded", 960, 896},
                        {832, 256, 0, "XCD-contiguous rows: stride 832, 832 B loaded (13 columns)", 832, 832}, {960, 256, 0, "XCD-contiguous rows: stride 960, 960 B loaded (15 columns)", 960, 960},
                        {1024, 256, 0, "XCD-contiguous rows: stride 1024, 1024 B loaded (16 columns)", 1024, 1024}, {784, 256, 1, "XCD-contiguous rows: 4-byte grid, rows of 784 B", 784, 784},
                        {896, 256, 5, "XCD-contiguous rows: TILE-ALIGNED slab -- rows of 864 .. 928 B, each starting on a 128-byte line (strides 896 / 1024)", 896, 896},
                        {896, 256, 3, "XCD-contiguous rows: today (again, next to the line above)", 896, 896},
                        {896, 256, 5, "XCD-contiguous rows: TILE-ALIGNED slab (again)", 896, 896},
                        {640, 0, 0, "sweep"}, {704, 0, 0, "sweep"}, {768, 0, 0, "sweep"}, {832, 0, 0, "sweep"}, {960, 0, 0, "sweep"},
                        {1024, 0, 0, "1-KiB pieces (the round-4 probe's shape)"}};
        for (int depth : {4}) for (const V& v : vs) for (int rep2 = 0; rep2 < 2; ++rep2) {
            auto k = depth == 4 ? k_rows_var<4> : k_rows_var<8>;
            CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k, dim3(nrows), dim3(1024), 131072, st, (const u64*)d_bases, nstreams, nrows, v.load ? v.load : v.P, v.mode, v.jitter, v.S ? v.S : v.P, sink);
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double bytes = (double)nrows * v.P * nstreams;
            printf("{\"rows\": %u, \"row_bytes\": %u, \"mode\": %u, \"jitter\": %u, \"depth\": %d, \"ms\": %.4f, \"GBps\": %.1f, \"what\": \"%s\"}\n", nrows, v.P, v.mode, v.jitter, depth, best, bytes / best / 1e6, v.what);
            fflush(stdout);
        }
        for (void* p : owned) CHK(hipFree(p));
        return 0;
    }
    const char* allocs[3] = {"sep", "arena", "pow2"};
    for (int am = 0; am < 3; ++am) {
        std::vector<void*> owned; std::vector<u64> bases(nstreams);
        u64 stride = 0;
        if (am == 0) {
            for (u32 s = 0; s < nstreams; ++s) { void* p; CHK(hipMalloc(&p, (stream_bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20))); owned.push_back(p); bases[s] = (u64)(uintptr_t)p; }
        } else {
            stride = am == 1 ? (stream_bytes + 255) / 256 * 256 : (4ull << 20);
            void* p; CHK(hipMalloc(&p, stride * nstreams + (1u << 20))); owned.push_back(p);
            for (u32 s = 0; s < nstreams; ++s) bases[s] = (u64)(uintptr_t)p + stride * s;
        }
        const bool rnd = argc > 3 && atoi(argv[3]) != 0;          // argv[3] = 1: random data instead of a constant byte (data-dependent power / clocks)
        for (void* p : owned) {
            const u64 bytes = am == 0 ? stream_bytes : stride * nstreams;
            if (rnd) hipLaunchKernelGGL(k_fill_random, dim3(1024), dim3(256), 0, st, (u64*)p, bytes / 8, (u64)(uintptr_t)p);
            else CHK(hipMemsetAsync(p, 0x5a, bytes, st));
        }
        CHK(hipMemcpy(d_bases, bases.data(), (size_t)nstreams * 8, hipMemcpyHostToDevice));
        CHK(hipStreamSynchronize(st));
        struct Cfg { int lpp, depth, wg, lds; };
        const Cfg cfgs[] = {
            {1, 4, 1024, 131072}, {1, 8, 1024, 131072}, {1, 2, 1024, 131072},     // the tiled kernel's residency: 16 waves per CU
            {1, 4, 256, 0}, {1, 8, 256, 0},                                      // 8 small workgroups per CU
            {2, 4, 1024, 131072}, {2, 4, 256, 0},
            {4, 2, 1024, 131072}, {4, 4, 256, 0}, {4, 2, 256, 0},
            {8, 2, 256, 0}, {8, 1, 1024, 131072},
        };
        const bool quick = argc > 4 && atoi(argv[4]) != 0;
        const int work = argc > 5 ? atoi(argv[5]) : 0;                     // argv[5]: 1 = + row arithmetic, 2 = + LDS atomics (lpp 1 configurations only)
        int ci = 0;
        for (const Cfg& c : cfgs) {
            if (quick && ci++ >= 2) break;
            kfn k = pick(c.depth, c.lpp, work);
            if (!k) continue;
            CHK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            u32 npieces = (u32)(stream_bytes / (c.lpp * 1024u));
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k, dim3(npieces), dim3(c.wg), c.lds, st, (const u64*)d_bases, nstreams, npieces, sink);
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            double bytes = (double)npieces * c.lpp * 1024.0 * nstreams;
            printf("{\"alloc\": \"%s\", \"piece\": %d, \"depth\": %d, \"wg\": %d, \"lds\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n",
                   allocs[am], c.lpp * 1024, c.depth, c.wg, c.lds, best, bytes / best / 1e6);
            fflush(stdout);
        }
        for (void* p : owned) CHK(hipFree(p));
    }
    return 0;
}
