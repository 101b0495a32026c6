/*
 * ubench.hip — developer microbenchmarks for the design constants of the workgroup-per-chunk parse (qzk_deflate_wide.h):
 * what a workgroup barrier, a dependent ds_bpermute / LDS-read chain, an 8-bit match-any, an L2-served 16-bit gather and
 * scatter over a 128 KiB per-workgroup table, and an unaligned 16-byte LDS compare cost with SIXTEEN waves per CU,
 * one workgroup per CU.  Not part of the product; prints cycles (s_memtime) per operation.
 *   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define T 1024
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static __device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

struct Res { uint64_t c[16]; };

__global__ void __launch_bounds__(T) k_bench(Res *res, uint16_t *tab /* 65536 u16 per workgroup */, uint32_t iters, uint32_t sink_mask)
{
    __shared__ uint32_t lds[16384];             /* 64 KiB */
    __shared__ uint32_t small[2048];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint16_t *mytab = tab + (size_t)blockIdx.x * 65536;
    uint32_t acc = tid * 2654435761u;
    for (uint32_t i = tid; i < 16384; i += T) lds[i] = (i * 2654435761u) >> 7;
    for (uint32_t i = tid; i < 2048; i += T) small[i] = (i * 40503u) & 1023;
    for (uint32_t i = tid; i < 65536; i += T) mytab[i] = (uint16_t)(i * 31);
    __syncthreads();
    uint64_t t0, t1;
    Res r = {};

    /* 0: barrier with one LDS write + read in each interval */
    t0 = now();
    for (uint32_t i = 0; i < iters; i++) {
        small[tid] = acc;
        __syncthreads();
        acc += small[(tid + 65 * (i + 1)) & 1023];
        __syncthreads();
    }
    t1 = now(); r.c[0] = (t1 - t0) / (2 * iters);

    /* 1: dependent ds_bpermute chain */
    {
        uint32_t v = (lane * 7 + 3) & 63;
        t0 = now();
        for (uint32_t i = 0; i < 64 * iters; i++) v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(v << 2), (int)(v + i)) & 63;
        t1 = now(); r.c[1] = (t1 - t0) / (64 * iters); acc += v;
    }
    /* 2: dependent LDS read chain (uniform address within the wave: the cross-wave exit chain) */
    {
        uint32_t v = wv;
        t0 = now();
        for (uint32_t i = 0; i < 64 * iters; i++) v = small[(v + i) & 2047];
        t1 = now(); r.c[2] = (t1 - t0) / (64 * iters); acc += v;
    }
    /* 3: dependent LDS read chain, per-lane addresses (the prev[] walk) */
    {
        uint32_t v = tid;
        t0 = now();
        for (uint32_t i = 0; i < 64 * iters; i++) v = lds[(v + i) & 16383] & 16383;
        t1 = now(); r.c[3] = (t1 - t0) / (64 * iters); acc += v;
    }
    /* 4: 8-bit match-any (8 ballots) + rank */
    {
        uint32_t d = (acc >> 5) & 255;
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            uint64_t m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint32_t bit = (d >> b) & 1;
                const uint64_t bal = __ballot(bit);
                m &= bal ^ ((uint64_t)bit - 1);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            d = (d + rank + (uint32_t)__popcll(m) + i) & 255;
        }
        t1 = now(); r.c[4] = (t1 - t0) / (16 * iters); acc += d;
    }
    __syncthreads();
    /* 5: L2-served (sc1) 16-bit gather from the workgroup's 128 KiB table: 1024 random lanes, waited for */
    {
        uint32_t v = acc;
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            const uint32_t idx = (v * 2654435761u + i * 97u) >> 16;
            v += __hip_atomic_load(&mytab[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t1 = now(); r.c[5] = (t1 - t0) / (16 * iters); acc += v;
    }
    __syncthreads();
    /* 6: the same, four independent gathers per trip (throughput) */
    {
        uint32_t v = acc;
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            const uint32_t a = (v * 2654435761u + i * 97u), b = a * 40503u + 1, c = b * 40503u + 7, d = c * 40503u + 9;
            const uint32_t x0 = __hip_atomic_load(&mytab[a >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t x1 = __hip_atomic_load(&mytab[b >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t x2 = __hip_atomic_load(&mytab[c >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t x3 = __hip_atomic_load(&mytab[d >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += x0 + x1 + x2 + x3;
        }
        t1 = now(); r.c[6] = (t1 - t0) / (16 * iters); acc += v;
    }
    __syncthreads();
    /* 7: scattered 16-bit stores (40 % of the lanes), then vmcnt(0) */
    {
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            const uint32_t a = (acc * 2654435761u + i * 97u + tid * 7919u);
            if ((a & 7) < 3) mytab[a >> 16] = (uint16_t)a;
            __builtin_amdgcn_s_waitcnt(0);
            acc += i;
        }
        t1 = now(); r.c[7] = (t1 - t0) / (16 * iters);
    }
    __syncthreads();
    /* 8: unaligned 16-byte LDS fetch + compare against own bytes (5 aligned dwords + 4 alignbytes + xor/ctz), 4 candidates */
    {
        uint32_t v = acc;
        const uint32_t w0 = lds[tid], w1 = lds[tid + 1], w2 = lds[tid + 2], w3 = lds[tid + 3];
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            uint32_t best = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q = (v * 2654435761u + (uint32_t)k * 977u + i) & 0xffff;
                const uint32_t j = q >> 2, s = q & 3;
                const uint32_t d0 = lds[j & 16383], d1 = lds[(j + 1) & 16383], d2 = lds[(j + 2) & 16383], d3 = lds[(j + 3) & 16383], d4 = lds[(j + 4) & 16383];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(d1, d0, s) ^ w0, x1 = __builtin_amdgcn_alignbyte(d2, d1, s) ^ w1,
                               x2 = __builtin_amdgcn_alignbyte(d3, d2, s) ^ w2, x3 = __builtin_amdgcn_alignbyte(d4, d3, s) ^ w3;
                uint32_t len = 16;
                if (x3) len = 12 + (__builtin_ctz(x3) >> 3);
                if (x2) len = 8 + (__builtin_ctz(x2) >> 3);
                if (x1) len = 4 + (__builtin_ctz(x1) >> 3);
                if (x0) len = (__builtin_ctz(x0) >> 3);
                best = len > best ? len : best;
            }
            v += best;
        }
        t1 = now(); r.c[8] = (t1 - t0) / (16 * iters); acc += v;
    }
    /* 9: LDS atomic add histogram (286 bins), one per lane */
    {
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) { atomicAdd(&small[(acc + i * 13) % 286], 1u); acc = acc * 1664525u + 1013904223u; }
        __builtin_amdgcn_s_waitcnt(0);
        t1 = now(); r.c[9] = (t1 - t0) / (16 * iters);
    }
    /* 10: wave inclusive scan via shfl (6 steps) */
    {
        uint32_t v = acc & 15;
        t0 = now();
        for (uint32_t i = 0; i < 16 * iters; i++) {
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl((int)v, (int)lane - d, 64); if ((int)lane >= d) v += o; }
            v &= 15;
        }
        t1 = now(); r.c[10] = (t1 - t0) / (16 * iters); acc += v;
    }
    /* 11: bare barrier */
    t0 = now();
    for (uint32_t i = 0; i < 4 * iters; i++) __syncthreads();
    t1 = now(); r.c[11] = (t1 - t0) / (4 * iters);

    if ((acc & sink_mask) == 0x12345) small[0] = acc;
    if (tid == 0) res[blockIdx.x] = r;
    if (tid == 1 && (acc & sink_mask) == 0x54321) res[blockIdx.x].c[15] = acc;
}

int main(int argc, char **argv)
{
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 64;
    static const char *names[12] = {"barrier + LDS write/read interval", "dependent ds_bpermute", "dependent LDS read, uniform address",
        "dependent LDS read, per-lane address", "8-bit match-any + rank", "sc1 u16 gather, 1 per trip (latency)", "sc1 u16 gather, 4 per trip",
        "scattered u16 stores (3/8 lanes) + vmcnt(0)", "4 x unaligned 16-B LDS compare", "LDS atomicAdd, 286 bins", "wave inclusive scan (6 shfl)", "bare barrier"};
    for (int grid : {1, 256}) {
        Res *d_res; uint16_t *d_tab;
        CHK(hipMalloc(&d_res, sizeof(Res) * grid));
        CHK(hipMalloc(&d_tab, (size_t)grid * 65536 * 2));
        hipLaunchKernelGGL(k_bench, dim3(grid), dim3(T), 0, 0, d_res, d_tab, iters, 0xffffffffu);
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_bench, dim3(grid), dim3(T), 0, 0, d_res, d_tab, iters, 0xffffffffu);
        CHK(hipDeviceSynchronize());
        std::vector<Res> h(grid);
        CHK(hipMemcpy(h.data(), d_res, sizeof(Res) * grid, hipMemcpyDeviceToHost));
        printf("== %d workgroup(s) of %d threads, cycles per operation (median / max over workgroups)\n", grid, T);
        for (int k = 0; k < 12; k++) {
            std::vector<uint64_t> v;
            for (auto &x : h) v.push_back(x.c[k]);
            std::sort(v.begin(), v.end());
            printf("  %-48s %8llu %8llu\n", names[k], (unsigned long long)v[v.size() / 2], (unsigned long long)v.back());
        }
        hipFree(d_res); hipFree(d_tab);
    }
    return 0;
}
