/*
 * TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the device header
 * ms-eetc_amd/csrc/msd_kernel.hpp and run one workgroup as NT host threads, so that the kernel's
 * control flow can be debugged and run under AddressSanitizer/UBSan in a container without a GPU
 * (GPU sanitizers are not available on the pool).  It is never part of the product: the shipped
 * library is built by hipcc from the same header with the real HIP runtime, and nothing in
 * ms-eetc_amd/ references this directory.  Only what the kernel uses is provided.
 */
#pragma once

#include <pthread.h>

#include <cmath>
#include <cstddef>
#include <cstring>

#define MSD_HOST_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))

struct emu_dim3 { unsigned x, y, z; };

struct emu_block {
    pthread_barrier_t bar;
    unsigned nthreads;
    double *shfl;          /* nthreads doubles */
    double *xch;           /* nthreads * EMU_XCH doubles: wave_fetch exchange */
    void *lds;
};

extern thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
extern thread_local emu_block *emu_blk;

#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)emu_blk->lds;

static inline void __syncthreads() { pthread_barrier_wait(&emu_blk->bar); }

/* wave64 butterfly: every thread of the block calls it (the kernel only shuffles in block-wide reductions) */
static inline double __shfl_xor(double v, int mask)
{
    emu_blk->shfl[threadIdx.x] = v;
    pthread_barrier_wait(&emu_blk->bar);
    double r = emu_blk->shfl[threadIdx.x ^ (unsigned)mask];
    pthread_barrier_wait(&emu_blk->bar);
    return r;
}

/* every thread of the block calls it (uniform control flow around the wave exchanges of the kernel) */
#define EMU_XCH 32
static inline void emu_wave_fetch(const double *x, double *y, int K, int src)
{
    double *mine = emu_blk->xch + (size_t)threadIdx.x*EMU_XCH;
    for (int k = 0; k < K; k++) mine[k] = x[k];
    pthread_barrier_wait(&emu_blk->bar);
    const double *from = emu_blk->xch + (size_t)((threadIdx.x & ~63u) | ((unsigned)src & 63u))*EMU_XCH;
    for (int k = 0; k < K; k++) y[k] = from[k];
    pthread_barrier_wait(&emu_blk->bar);
}

/* the blocks of an emulated launch run one after the other: a plain read-modify-write is atomic enough */
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

using std::isfinite;

static inline void __builtin_amdgcn_sched_barrier(int) {}

/* scalarisation intrinsics: identity on the host */
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline int __double2loint(double d) { unsigned long long u; std::memcpy(&u, &d, 8); return (int)(u & 0xffffffffu); }
static inline int __double2hiint(double d) { unsigned long long u; std::memcpy(&u, &d, 8); return (int)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) { unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double d; std::memcpy(&d, &u, 8); return d; }

#include <x86intrin.h>
#define __builtin_readcyclecounter() ((unsigned long long)__rdtsc())
