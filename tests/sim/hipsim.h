/*
 * hipsim.h — a tiny CPU SIMT emulator for unit-testing the HIP kernels of this
 * repo in a container without a GPU.  TEST INFRASTRUCTURE ONLY: the product is
 * the hipcc build of qatzip_amd/csrc; nothing here ships or is benchmarked.
 *
 * Model: one workgroup at a time; every work-item is a fiber (hand-rolled
 * x86-64 context switch) that runs until it reaches a cross-lane operation
 * (ballot / shfl / readlane / wave_sync) or a workgroup barrier, where it
 * parks until every live lane of its wave (or workgroup) has arrived.  The
 * emulator is stricter than the hardware on purpose: cross-lane ops must be
 * reached by all live lanes of a wave with the same op tag (wave-uniform
 * control flow), and LDS traffic between lanes is only ordered by an explicit
 * qz_wave_sync()/qz_block_sync() — exactly the discipline the kernels follow.
 */
#ifndef HIPSIM_H
#define HIPSIM_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <functional>
#include <vector>

struct sim_dim3 { unsigned x, y, z; };
static sim_dim3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void sim_switch(void **save_sp, void *load_sp);
asm(".text\n.globl sim_switch\n.type sim_switch,@function\nsim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");

namespace sim {
enum { STACK = 256 * 1024, MAXT = 1024 };
struct Fiber { void *sp; char *stack; bool done; int wait_kind; unsigned wait_gen; };
struct Wave {
    unsigned gen, arrived, live; int tag;
    uint64_t a[2][64], b[2][64];      /* double-buffered exchange slots */
    uint64_t part[2];                 /* lanes that took part in the rendezvous (snapshot when it completed) */
};
static Fiber fib[MAXT];
static Wave waves[MAXT / 64];
static void *sched_sp;
static int cur, nthreads;
static unsigned blk_gen, blk_arrived, blk_live;
static std::function<void()> *body;
static char *dyn_lds;

static inline uint64_t live_mask(int wave)
{
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) if (wave * 64 + i < nthreads && !fib[wave * 64 + i].done) m |= 1ull << i;
    return m;
}

static void fiber_main()
{
    (*body)();
    Fiber &f = fib[cur];
    f.done = true;
    waves[cur / 64].live--;
    blk_live--;
    /* a lane leaving may complete a pending rendezvous */
    Wave &w = waves[cur / 64];
    if (w.live && w.arrived == w.live) { w.part[w.gen & 1] = live_mask(cur / 64); w.arrived = 0; w.gen++; }
    if (blk_live && blk_arrived == blk_live) { blk_arrived = 0; blk_gen++; }
    sim_switch(&f.sp, sched_sp);
    abort();
}

static inline void yield() { sim_switch(&fib[cur].sp, sched_sp); }

static inline int lane() { return cur & 63; }

/* wave rendezvous; returns buffer parity used for this op */
static inline unsigned wave_arrive(int tag, uint64_t va, uint64_t vb)
{
    Wave &w = waves[cur / 64];
    unsigned par = w.gen & 1, g = w.gen;
    if (w.arrived == 0) w.tag = tag;
    else if (w.tag != tag) { fprintf(stderr, "hipsim: divergent cross-lane op (tag %d vs %d) lane %d\n", w.tag, tag, lane()); abort(); }
    w.a[par][lane()] = va; w.b[par][lane()] = vb;
    if (++w.arrived == w.live) { w.part[par] = live_mask(cur / 64); w.arrived = 0; w.gen++; }
    else { fib[cur].wait_kind = 1; fib[cur].wait_gen = g; while (waves[cur / 64].gen == g) yield(); }
    return par;
}

static inline void block_arrive()
{
    unsigned g = blk_gen;
    if (++blk_arrived == blk_live) { blk_arrived = 0; blk_gen++; }
    else { while (blk_gen == g) yield(); }
}

static void launch(unsigned grid, unsigned block, size_t shmem, std::function<void()> fn)
{
    static bool init;
    if (!init) {
        for (int i = 0; i < MAXT; i++) {
            fib[i].stack = (char *)mmap(NULL, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (fib[i].stack == MAP_FAILED) abort();
        }
        init = true;
    }
    if (block > MAXT || block % 64) { fprintf(stderr, "hipsim: bad block size %u\n", block); abort(); }
    body = &fn;
    nthreads = (int)block;
    gridDim = {grid, 1, 1}; blockDim = {block, 1, 1};
    dyn_lds = (char *)realloc(dyn_lds, shmem + 64);
    for (unsigned b = 0; b < grid; b++) {
        blockIdx = {b, 0, 0};
        blk_gen = blk_arrived = 0; blk_live = block;
        for (unsigned wv = 0; wv < block / 64; wv++) { waves[wv].gen = 0; waves[wv].arrived = 0; waves[wv].live = 64; }
        for (unsigned t = 0; t < block; t++) {
            Fiber &f = fib[t];
            uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
            void **sp = (void **)(top - 16);
            *sp = (void *)fiber_main;            /* return address */
            sp -= 6;                              /* r15..rbp */
            memset(sp, 0, 6 * sizeof(void *));
            f.sp = sp; f.done = false;
        }
        unsigned live = block;
        while (live) {
            live = 0;
            for (unsigned t = 0; t < block; t++) {
                if (fib[t].done) continue;
                live++;
                cur = (int)t; threadIdx = {t, 0, 0};
                sim_switch(&sched_sp, fib[t].sp);
            }
        }
    }
}
} // namespace sim

/* ---- the cross-lane vocabulary the kernels use (see qzk_common.h) ---- */
static inline uint64_t qz_ballot(bool p)
{
    unsigned par = sim::wave_arrive(1, p, 0);
    sim::Wave &w = sim::waves[sim::cur / 64];
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) if (((w.part[par] >> i) & 1) && w.a[par][i]) m |= 1ull << i;
    return m;
}
static inline uint32_t qz_shfl(uint32_t v, int src)
{
    unsigned par = sim::wave_arrive(2, v, 0);
    return (uint32_t)sim::waves[sim::cur / 64].a[par][src & 63];
}
static inline uint32_t qz_readlane(uint32_t v, int src)
{
    unsigned par = sim::wave_arrive(3, v, (uint64_t)src);
    sim::Wave &w = sim::waves[sim::cur / 64];
    for (int i = 0; i < 64; i++)
        if (((w.part[par] >> i) & 1) && (int)w.b[par][i] != src) {
            fprintf(stderr, "hipsim: readlane with non-uniform lane index\n"); abort();
        }
    return (uint32_t)w.a[par][src & 63];
}
static inline uint32_t qz_readfirstlane(uint32_t v)
{
    unsigned par = sim::wave_arrive(4, v, 0);
    sim::Wave &w = sim::waves[sim::cur / 64];
    for (int i = 0; i < 64; i++) if ((w.part[par] >> i) & 1) return (uint32_t)w.a[par][i];
    return v;
}
static inline void qz_wave_sync() { sim::wave_arrive(5, 0, 0); }
static inline void qz_block_sync() { sim::block_arrive(); }
static inline int qz_lane() { return sim::cur & 63; }

template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }

#endif
