// Host emulation of the small HIP subset the engine uses -- TEST INFRASTRUCTURE ONLY.
//
// tests/hostemu/build_emu.py compiles traceweaver_amd/csrc/tw_engine.hip *unchanged* with g++ and
// this directory first on the include path, producing libtwgpu_emu.so.  Kernels run as plain
// functions, one workgroup after another, one thread after another; barriers are no-ops, so the
// emulated library is driven with TW_TILE=1 TW_COOP_THREADS=1 (one thread per workgroup), which
// every kernel is written to tolerate.  It exists so that the CPU-only test tier can check the
// engine's *logic* (windows, enumeration order, tie handling, repair walk) against the oracle; it
// is never loaded by the product package and says nothing about GPU behaviour or speed.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorEmu = 1 };
typedef void* hipStream_t;
struct emu_event { std::chrono::steady_clock::time_point t; };
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

inline const char* hipGetErrorString(hipError_t) { return "host emulation error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmu; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

template <class K, class... A>
inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid;
    blockDim = block;
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned b = 0; b < grid.x; b++)
            for (unsigned t = 0; t < block.x; t++) {
                blockIdx = dim3(b, by, 0);
                threadIdx = dim3(t, 0, 0);
                kernel(args...);
            }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() {}
inline void __threadfence() {}
inline void __threadfence_block() {}
inline long long __double_as_longlong(double x) { long long r; memcpy(&r, &x, 8); return r; }
inline double __longlong_as_double(long long x) { double r; memcpy(&r, &x, 8); return r; }
template <class T> inline T __shfl_down(T v, int) { return v; }  // only reachable with one lane per wavefront
// wave intrinsics for a wavefront of one lane (kernels that use them run with one thread per workgroup here)
inline unsigned long long __ballot(int pred) { return pred ? 1ull : 0ull; }
template <class T> inline T __shfl(T v, int) { return v; }
template <class T> inline T __shfl_xor(T v, int) { return v; }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_wave_barrier() {}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
using std::max;
using std::min;

template <class T> inline T atomicCAS(T* p, T cmp, T val) { T old = *p; if (old == cmp) *p = val; return old; }
template <class T> inline T atomicMin(T* p, T v) { T old = *p; if (v < old) *p = v; return old; }
template <class T> inline T atomicMax(T* p, T v) { T old = *p; if (v > old) *p = v; return old; }
template <class T> inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }
template <class T> inline T atomicOr(T* p, T v) { T old = *p; *p = old | v; return old; }
template <class T> inline T atomicExch(T* p, T v) { T old = *p; *p = v; return old; }
