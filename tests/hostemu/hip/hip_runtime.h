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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define TW_HOST_EMULATION 1   // tw_kernels.h: lane arrays (LaneArr) are plain arrays here
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) static type var[8192];   // (dynamic LDS of a launch: the tests' pools stay below this)
#define __launch_bounds__(...)

struct uint4 { unsigned x, y, z, w; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

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
enum { hipStreamDefault = 0 };
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event(); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmu; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorEmu; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

// ---- lane-threaded mode ---------------------------------------------------------------------------------------
struct EmuBarrier {   // barrier whose participants may leave for good
    std::mutex m;
    std::condition_variable cv;
    int expected = 0, arrived = 0;
    unsigned long long gen = 0;
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long long g = gen;
        if (++arrived >= expected) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
    void drop() {
        std::unique_lock<std::mutex> lk(m);
        expected--;
        if (expected > 0 && arrived >= expected) { arrived = 0; gen++; cv.notify_all(); }
    }
};
struct EmuWave {
    EmuBarrier bar;
    unsigned long long slot[64];
    std::atomic<unsigned long long> live{0};
};
struct EmuBlock {
    EmuBarrier bar;
    std::vector<std::unique_ptr<EmuWave>> waves;
};
inline EmuBlock* emu_block = nullptr;            // the workgroup that is running (workgroups run one after another)
inline thread_local EmuWave* emu_wave = nullptr;  // null in the sequential mode
inline thread_local int emu_lane = 0;
inline void emu_segv(int) { void* bt[48]; const int n = backtrace(bt, 48); backtrace_symbols_fd(bt, n, 2); _exit(139); }
inline bool emu_lanes_enabled() { const char* v = getenv("TW_EMU_LANES"); return v && *v && *v != '0'; }

template <class K, class... A>
inline void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
    gridDim = grid;
    blockDim = block;
    const bool threaded = emu_lanes_enabled() && block.x > 1;
    if (threaded && getenv("TW_EMU_BACKTRACE")) signal(SIGSEGV, emu_segv);
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned b = 0; b < grid.x; b++) {
            if (!threaded) {
                for (unsigned t = 0; t < block.x; t++) {
                    blockIdx = dim3(b, by, 0);
                    threadIdx = dim3(t, 0, 0);
                    kernel(args...);
                }
                continue;
            }
            EmuBlock blk;
            const unsigned nwave = (block.x + 63) / 64;
            for (unsigned w = 0; w < nwave; w++) {
                blk.waves.emplace_back(new EmuWave());
                const unsigned lanes = std::min(64u, block.x - 64 * w);
                blk.waves.back()->bar.expected = (int)lanes;
                blk.waves.back()->live = lanes == 64 ? ~0ull : ((1ull << lanes) - 1ull);
            }
            blk.bar.expected = (int)block.x;
            emu_block = &blk;
            std::vector<std::thread> lanes;
            for (unsigned t = 0; t < block.x; t++)
                lanes.emplace_back([&, t]() {
                    blockIdx = dim3(b, by, 0);
                    threadIdx = dim3(t, 0, 0);
                    emu_wave = blk.waves[t / 64].get();
                    emu_lane = (int)(t % 64);
                    kernel(args...);
                    emu_wave->live.fetch_and(~(1ull << emu_lane));   // this lane takes part in nothing any more
                    emu_wave->bar.drop();
                    blk.bar.drop();
                    emu_wave = nullptr;
                });
            for (auto& th : lanes) th.join();
            emu_block = nullptr;
        }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { if (emu_wave) emu_block->bar.arrive_and_wait(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline long long __double_as_longlong(double x) { long long r; memcpy(&r, &x, 8); return r; }
inline double __longlong_as_double(long long x) { double r; memcpy(&r, &x, 8); return r; }
inline int __builtin_amdgcn_readfirstlane(int v);
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
inline void __builtin_amdgcn_wave_barrier() { if (emu_wave) emu_wave->bar.arrive_and_wait(); }

// every lane deposits a value, all live lanes meet, every lane reads what it needs, all meet again
template <class T, class F>
inline T emu_exchange(T v, F pick) {
    static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits");
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    emu_wave->slot[emu_lane] = raw;
    emu_wave->bar.arrive_and_wait();
    const unsigned long long live = emu_wave->live.load();
    const int src = pick(live);
    unsigned long long got = (src >= 0 && src < 64 && ((live >> src) & 1ull)) ? emu_wave->slot[src] : raw;
    emu_wave->bar.arrive_and_wait();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
inline unsigned long long __ballot(int pred) {
    if (!emu_wave) return pred ? 1ull : 0ull;
    emu_wave->slot[emu_lane] = pred ? 1ull : 0ull;
    emu_wave->bar.arrive_and_wait();
    const unsigned long long live = emu_wave->live.load();
    unsigned long long mask = 0;
    for (int l = 0; l < 64; l++) if (((live >> l) & 1ull) && emu_wave->slot[l]) mask |= 1ull << l;
    emu_wave->bar.arrive_and_wait();
    return mask;
}
template <class T> inline T __shfl(T v, int src) { return emu_wave ? emu_exchange(v, [&](unsigned long long) { return src; }) : v; }
template <class T> inline T __shfl_down(T v, int off) { return emu_wave ? emu_exchange(v, [&](unsigned long long) { return emu_lane + off; }) : v; }
template <class T> inline T __shfl_up(T v, int off) { return emu_wave ? emu_exchange(v, [&](unsigned long long) { return emu_lane - off; }) : v; }
template <class T> inline T __shfl_xor(T v, int m) { return emu_wave ? emu_exchange(v, [&](unsigned long long) { return emu_lane ^ m; }) : v; }
inline int __builtin_amdgcn_readfirstlane(int v) {
    return emu_wave ? emu_exchange(v, [&](unsigned long long live) { return __builtin_ffsll((long long)live) - 1; }) : v;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
using std::max;
using std::min;

template <class T> inline T atomicCAS(T* p, T cmp, T val) { __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <class T> inline T atomicMin(T* p, T v) { T old = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
template <class T> inline T atomicMax(T* p, T v) { T old = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return old; }
template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
