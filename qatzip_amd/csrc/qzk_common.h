/*
 * qzk_common.h — device-side vocabulary shared by the gfx950 kernels.
 *
 * The kernels are written for CDNA4 wave64 and only ever built for the GPU by
 * hipcc --offload-arch=gfx950.  The single `QZ_SIM` switch below exists for the
 * test suite: tests/sim/ compiles the same kernel bodies with g++ on top of a
 * fiber-based SIMT emulator (tests/sim/hipsim.h) so that bit-exactness can be
 * fuzzed in a container without a GPU.  It is not a second backend.
 */
#ifndef QZK_COMMON_H
#define QZK_COMMON_H
#include <stdint.h>

#ifdef QZ_SIM
#include "hipsim.h"
#define QZ_DEV static inline
#define QZ_KERNEL static void
#define QZ_KERNEL_MAX(n) static void
#define QZ_KERNEL_OCC(n, w) static void
#define QZ_LDS static
#define QZ_CONST static const
#else
#include <hip/hip_runtime.h>
#define QZ_DEV static __device__ __forceinline__
#define QZ_KERNEL static __global__ void   /* internal linkage: kernels live in headers shared by several .hip units */
/* a kernel that is only ever launched with <= n threads per workgroup: without the bound the compiler budgets VGPRs for
 * 1024-thread workgroups (128 per lane) and spills the rest to scratch */
#define QZ_KERNEL_MAX(n) static __global__ void __launch_bounds__(n)
/* ... and that should keep at least w waves per SIMD resident (the compiler caps the VGPRs accordingly) */
#define QZ_KERNEL_OCC(n, w) static __global__ void __launch_bounds__(n, w)
#define QZ_LDS __shared__
#define QZ_CONST static __device__ const

QZ_DEV uint64_t qz_ballot(bool p) { return __ballot(p); }
QZ_DEV uint32_t qz_shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
QZ_DEV uint32_t qz_readlane(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
QZ_DEV uint32_t qz_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
/* a wave-uniform value into ONE lane of a vector register (this clang has no writelane builtin: a compare the fields of one
 * record share, and a v_cndmask per field) */
QZ_DEV uint32_t qz_writelane(uint32_t old, uint32_t val, int sel) { return (int)(threadIdx.x & 63) == sel ? val : old; }
/* orders LDS traffic between the lanes of ONE wave (single-wave workgroups) */
QZ_DEV void qz_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
/* orders only LDS traffic between the lanes of one wave: DS instructions of a wave execute in order, so all
 * that is needed is to stop the compiler from moving them; outstanding global loads stay in flight */
QZ_DEV void qz_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
QZ_DEV void qz_block_sync() { __syncthreads(); }
QZ_DEV int qz_lane() { return (int)(threadIdx.x & 63); }
/* byte load served by the L2 (sc1, bypasses this CU's vector L1): used to read back bytes this same wave
 * stored a moment ago.  Vector memory requests of one wave reach the L2 channel of an address in issue order
 * and the L1 is write-through, so the load observes the earlier store without waiting for its completion. */
QZ_DEV uint8_t qz_ld8_l2(const uint8_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
#ifdef QZ_SIM
static inline void qz_lds_sync() { qz_wave_sync(); }
static inline uint32_t qz_writelane(uint32_t old, uint32_t val, int sel) { return qz_lane() == sel ? val : old; }
static inline uint8_t qz_ld8_l2(const uint8_t *p) { return *p; }
#endif

/* inclusive prefix sum over the 64 lanes of a wave (every lane active).  On the GPU: six DPP adds - shifts by 1, 2, 4, 8 inside
 * the rows of sixteen, then the last lane of row 0 / 2 into the row behind it and of the lower half into the upper one - with no trip
 * through the LDS crossbar (six ds_bpermute round trips the other way). */
#ifndef QZ_SIM
QZ_DEV uint32_t qz_wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      /* row_bcast:15 -> rows 1 and 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      /* row_bcast:31 -> rows 2 and 3 */
    return v;
}
#else
static inline uint32_t qz_wave_incl_scan(uint32_t v)
{
    const int lane = qz_lane();
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = qz_shfl(v, lane - d);
        if (lane >= d) v += o;
    }
    return v;
}
#endif

QZ_DEV int qz_popc64(uint64_t v) { return __builtin_popcountll(v); }
QZ_DEV int qz_ctz64(uint64_t v) { return __builtin_ctzll(v); }      /* v != 0 */
QZ_DEV int qz_ctz32(uint32_t v) { return __builtin_ctz(v); }        /* v != 0 */
QZ_DEV int qz_msb64(uint64_t v) { return 63 - __builtin_clzll(v); } /* v != 0 */
QZ_DEV uint64_t qz_below(int l) { return l >= 64 ? ~0ull : ((1ull << l) - 1); } /* bits < l */

/* little-endian loads at arbitrary byte alignment (gfx950 global memory takes
 * unaligned dword accesses; the packed type lets the compiler choose) */
typedef struct __attribute__((packed, aligned(1))) { uint32_t v; } qz_u32u;
typedef struct __attribute__((packed, aligned(1))) { uint16_t v; } qz_u16u;
QZ_DEV uint32_t qz_ld32(const uint8_t *p) { return ((const qz_u32u *)p)->v; }
QZ_DEV uint32_t qz_ld16(const uint8_t *p) { return ((const qz_u16u *)p)->v; }

/* wave-uniform value forwarded through an SGPR (helps hipcc scalarise loops) */
QZ_DEV uint32_t qz_uniform(uint32_t v) { return qz_readfirstlane(v); }

#endif
