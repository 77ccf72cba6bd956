// tiled_probe.hip -- which part of k_agg_or_gap_tiled's load side costs the bandwidth?  (round 4)
//
// pieces_probe showed that 4096 streams read in 1-KiB pieces stream at 6.5 TB/s: the access PATTERN is not the floor the
// round-2 analysis took it for.  This probe rebuilds the tiled kernel's load side step by step on data shaped like
// configs[4] (4096 vectors x 61,036 GAP blocks in 64-byte slots + a u64 descriptor per block, every vector its own
// allocations) so that the step that loses the bandwidth shows:
//   V0  one coalesced 1-KiB wave load per (operand, tile of 16 columns), addresses computed, DEPTH loads in flight
//   V1  lane per block: 16 lanes x 4 operands per wave, each lane 4 x 16 B at its block (stride 64 B), addresses computed,
//       next step's loads issued before the current step's data is consumed
//   V2  V1 + the descriptor chain of the real kernel: table pointer -> descriptor -> block (three dependent stages,
//       each one step ahead of the next)
//   V3  V1 with non-temporal loads;  V4  V0 with plain loads
//   V5  V2 but descriptors 2 steps ahead and blocks double-buffered (two block batches in flight)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tiled_probe.hip -o ../bin/tiled_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4* gptr;
typedef const __attribute__((address_space(1))) u64* gptr64;

__device__ __forceinline__ u64 uni64(u64 v)
{
    return (u64)(u32)__builtin_amdgcn_readfirstlane((u32)v) | ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32);
}

template <bool NT> __device__ __forceinline__ u32x4 ld(gptr p) { if (NT) return __builtin_nontemporal_load(p); else return *p; }

// V0 / V4
template <int DEPTH, bool NT>
__global__ __launch_bounds__(1024) void k_rows(const u64* __restrict__ slabs, u32 n, u32 ntiles, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u64 off = (u64)blockIdx.x * 1024u + lane * 16u;
    u32x4 acc = (u32x4)(0u);
    u32 s = wave;
    for (; s + (DEPTH - 1) * W < n; s += DEPTH * W) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = ld<NT>((gptr)(uni64(slabs[s + d * W]) + off));
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

// V1 / V3: lane per block, computed addresses
template <bool NT>
__global__ __launch_bounds__(1024) void k_lanes(const u64* __restrict__ slabs, u32 n, u32 ntiles, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u32 t = lane & 15u, grp = lane >> 4;
    const u64 off = ((u64)blockIdx.x * 16u + t) * 64u;
    const u32 S = W * 4u;
    u32x4 acc = (u32x4)(0u);
    u32 op = wave * 4u + grp;
    u32x4 a[4], b[4];
    { gptr p = (gptr)(slabs[op < n ? op : n - 1u] + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = ld<NT>(p + j); }
    for (; op < n; op += S) {
        u32 o2 = op + S < n ? op + S : n - 1u;
        gptr p = (gptr)(slabs[o2] + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ld<NT>(p + j);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= a[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = b[j];
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

// V2: + descriptor chain (table pointer -> descriptor -> block), each stage one step ahead of the next
__global__ __launch_bounds__(1024) void k_chain(const u64* __restrict__ descs, u32 n, u32 ncols, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u32 t = lane & 15u, grp = lane >> 4;
    const u32 col = blockIdx.x * 16u + t;
    const u32 cc = col < ncols ? col : ncols - 1u;
    const u32 S = W * 4u;
    const u32 nm1 = n - 1u;
    gptr64 g_descs = (gptr64)(uintptr_t)descs;
    u32x4 acc = (u32x4)(0u);
    u32 op = wave * 4u + grp;
#define STAGE_A(OP) (g_descs[(OP) < n ? (OP) : nm1])
#define STAGE_B(PA) (((gptr64)(uintptr_t)(PA))[cc])
    u64 pa0 = STAGE_A(op), pa1 = STAGE_A(op + S), pa2 = STAGE_A(op + 2u * S), pa3;
    u64 d0 = STAGE_B(pa0), d1 = STAGE_B(pa1), d2;
    u32x4 h0[4], h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h0[j] = ((gptr)(uintptr_t)d0)[j];
    for (; op < n; op += S) {
        pa3 = STAGE_A(op + 3u * S);
        d2 = STAGE_B(pa2);
#pragma unroll
        for (int j = 0; j < 4; ++j) h1[j] = ((gptr)(uintptr_t)d1)[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= h0[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) h0[j] = h1[j];
        d1 = d2; pa2 = pa3;
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

// V5: descriptors 3 steps ahead, two block batches in flight
__global__ __launch_bounds__(1024) void k_chain2(const u64* __restrict__ descs, u32 n, u32 ncols, u32* __restrict__ sink)
{
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const u32 t = lane & 15u, grp = lane >> 4;
    const u32 col = blockIdx.x * 16u + t;
    const u32 cc = col < ncols ? col : ncols - 1u;
    const u32 S = W * 4u;
    const u32 nm1 = n - 1u;
    gptr64 g_descs = (gptr64)(uintptr_t)descs;
    u32x4 acc = (u32x4)(0u);
    u32 op = wave * 4u + grp;
    u64 pa[5], d[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) pa[k] = STAGE_A(op + (u32)k * S);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = STAGE_B(pa[k]);
    u32x4 h0[4], h1[4], h2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h0[j] = ((gptr)(uintptr_t)d[0])[j]; h1[j] = ((gptr)(uintptr_t)d[1])[j]; }
    for (; op < n; op += S) {
        u64 pan = STAGE_A(op + 5u * S);
        u64 dn = STAGE_B(pa[4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) h2[j] = ((gptr)(uintptr_t)d[2])[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= h0[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h0[j] = h1[j]; h1[j] = h2[j]; }
        d[2] = d[3]; d[3] = dn; pa[4] = pan;
    }
    u32 x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x12345679u) { lds[threadIdx.x] = x; sink[0] = lds[(threadIdx.x + 1) % blockDim.x]; }
}

int main(int argc, char** argv)
{
    u32 n = 4096, ncols = 61036;
    if (argc > 1) n = (u32)atoi(argv[1]);
    const u32 ntiles = (ncols + 15) / 16;
    const size_t slab_bytes = (size_t)ntiles * 1024, desc_bytes = (size_t)ncols * 8;
    CHK(hipSetDevice(0));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    u32* sink; CHK(hipMalloc(&sink, 64));
    std::vector<u64> slabs(n), descs(n);
    std::vector<u64> hd(ncols);
    for (u32 s = 0; s < n; ++s) {
        void* p; CHK(hipMalloc(&p, (slab_bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20))); slabs[s] = (u64)(uintptr_t)p;
        CHK(hipMemsetAsync(p, 0x5a, slab_bytes, st));
        void* q; CHK(hipMalloc(&q, (desc_bytes + 65535) / 65536 * 65536)); descs[s] = (u64)(uintptr_t)q;
        for (u32 c = 0; c < ncols; ++c) hd[c] = slabs[s] + (u64)c * 64u;
        CHK(hipMemcpy(q, hd.data(), desc_bytes, hipMemcpyHostToDevice));
    }
    u64 *d_slabs, *d_descs;
    CHK(hipMalloc(&d_slabs, n * 8)); CHK(hipMalloc(&d_descs, n * 8));
    CHK(hipMemcpy(d_slabs, slabs.data(), n * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_descs, descs.data(), n * 8, hipMemcpyHostToDevice));
    CHK(hipStreamSynchronize(st));
    const double slab_total = (double)slab_bytes * n, desc_total = (double)desc_bytes * n;
    auto run = [&](const char* name, auto launch, double bytes) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0, st));
            launch();
            CHK(hipEventRecord(e1, st));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        CHK(hipGetLastError());
        printf("{\"variant\": \"%s\", \"ms\": %.4f, \"GBps\": %.1f, \"bytes\": %.0f}\n", name, best, bytes / best / 1e6, bytes);
        fflush(stdout);
    };
    const int LDS = 131072;
#define ATTR(K) CHK(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    ATTR((k_rows<4, true>)); ATTR((k_rows<4, false>)); ATTR((k_rows<2, true>)); ATTR(k_lanes<false>); ATTR(k_lanes<true>); ATTR(k_chain); ATTR(k_chain2);
    run("V0 rows depth4 nt", [&] { hipLaunchKernelGGL((k_rows<4, true>), dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_slabs, n, ntiles, sink); }, slab_total);
    run("V4 rows depth4 plain", [&] { hipLaunchKernelGGL((k_rows<4, false>), dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_slabs, n, ntiles, sink); }, slab_total);
    run("V0 rows depth2 nt", [&] { hipLaunchKernelGGL((k_rows<2, true>), dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_slabs, n, ntiles, sink); }, slab_total);
    run("V1 lane-per-block plain", [&] { hipLaunchKernelGGL(k_lanes<false>, dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_slabs, n, ntiles, sink); }, slab_total);
    run("V3 lane-per-block nt", [&] { hipLaunchKernelGGL(k_lanes<true>, dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_slabs, n, ntiles, sink); }, slab_total);
    run("V2 lane-per-block + descriptor chain", [&] { hipLaunchKernelGGL(k_chain, dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_descs, n, ncols, sink); }, slab_total + desc_total);
    run("V5 chain, two block batches in flight", [&] { hipLaunchKernelGGL(k_chain2, dim3(ntiles), dim3(1024), LDS, st, (const u64*)d_descs, n, ncols, sink); }, slab_total + desc_total);
    return 0;
}
