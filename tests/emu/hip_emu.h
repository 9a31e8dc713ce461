// TEST INFRASTRUCTURE ONLY -- never part of the product build.
//
// Minimal CPU emulation of the HIP constructs the kernels in scpp_amd/csrc use, so that the SAME
// kernel sources can be compiled with g++ and executed lane-accurately in this GPU-less dev
// container (and under pytest -m "not gpu") to debug kernel logic before spending GPU minutes:
//   * each workgroup runs as `blockDim.x` cooperative fibers (hand-rolled x86-64 context switch);
//   * __syncthreads() = round-robin yield; blocks execute sequentially;
//   * wave shuffles / ballot-free reductions and the f64 MFMA builtin are emulated through an
//     exchange buffer with the gfx950 lane->element maps of cdna_hip_programming.md §3;
//   * a tiny hipMalloc/hipMemcpy/hipStream/hipEvent shim lets the C-ABI layer run unchanged.
// The product library (libscpp_hip.so) is built by hipcc from the same sources WITHOUT this header.
#pragma once
#ifndef SCPP_HIP_EMU
#error "hip_emu.h is only for the emulation build (-DSCPP_HIP_EMU)"
#endif
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>
#include <mutex>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu
{
struct Fiber
{
    void *sp = nullptr;
    void *stack = nullptr;
    bool done = false;
    dim3 tid;
};
struct State
{
    std::vector<Fiber> fibers;
    void *main_sp = nullptr;
    int cur = -1;
    dim3 bid, bdim, gdim;
    std::function<void()> body;
    double xchg[1024][4];
    long xchg_i[1024];
};
inline State &st()
{
    static State s;
    return s;
}
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
__asm__(".text\n"
        ".globl hipemu_switch\n"
        ".type hipemu_switch,@function\n"
        "hipemu_switch:\n"
        "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
        "  movq %rsp, (%rdi)\n"
        "  movq %rsi, %rsp\n"
        "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
        "  ret\n"
        ".size hipemu_switch,.-hipemu_switch\n");

inline void yield_to_main()
{
    State &s = st();
    Fiber &f = s.fibers[s.cur];
    hipemu_switch(&f.sp, s.main_sp);
}
inline void fiber_entry()
{
    State &s = st();
    s.body();
    s.fibers[s.cur].done = true;
    for (;;)
        yield_to_main();
}
constexpr size_t STACK_BYTES = 512 * 1024;
inline void run_block(unsigned nthreads)
{
    State &s = st();
    if (s.fibers.size() < nthreads)
    {
        size_t old = s.fibers.size();
        s.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; i++)
            s.fibers[i].stack = std::aligned_alloc(64, STACK_BYTES);
    }
    for (unsigned i = 0; i < nthreads; i++)
    {
        Fiber &f = s.fibers[i];
        f.done = false;
        f.tid = dim3(i, 0, 0);
        uintptr_t top = (uintptr_t(f.stack) + STACK_BYTES) & ~uintptr_t(15);
        void **p = reinterpret_cast<void **>(top);
        *--p = nullptr;                                     // fake return address of fiber_entry
        *--p = reinterpret_cast<void *>(&fiber_entry);      // popped by `ret`
        for (int r = 0; r < 6; r++)
            *--p = nullptr;                                 // r15..rbp
        f.sp = p;
    }
    for (;;)
    {
        bool any = false;
        for (unsigned i = 0; i < nthreads; i++)
        {
            if (s.fibers[i].done)
                continue;
            any = true;
            s.cur = int(i);
            hipemu_switch(&s.main_sp, s.fibers[i].sp);
        }
        if (!any)
            break;
    }
    s.cur = -1;
}
inline std::mutex &launch_mutex()
{
    static std::mutex m;
    return m;
}
template <class K, class... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args)
{
    // one emulated kernel at a time: the fiber state and the `static` stand-ins for __shared__ are process-wide (host
    // front ends that drive one context per thread, sc_oneshot --gpus N, are tested against this build)
    std::lock_guard<std::mutex> lock(launch_mutex());
    State &s = st();
    s.gdim = grid;
    s.bdim = block;
    for (unsigned b = 0; b < grid.x; b++)
    {
        s.bid = dim3(b, 0, 0);
        s.body = [=]() { kernel(args...); };
        run_block(block.x);
    }
}
struct TidProxy
{
    operator dim3() const { return st().fibers[st().cur].tid; }
    unsigned get_x() const { return st().fibers[st().cur].tid.x; }
};
} // namespace hipemu

struct hipemu_tid_t
{
    struct X
    {
        operator unsigned() const { return hipemu::st().fibers[hipemu::st().cur].tid.x; }
    } x;
};
struct hipemu_bid_t
{
    struct X
    {
        operator unsigned() const { return hipemu::st().bid.x; }
    } x;
};
struct hipemu_bdim_t
{
    struct X
    {
        operator unsigned() const { return hipemu::st().bdim.x; }
    } x;
};
struct hipemu_gdim_t
{
    struct X
    {
        operator unsigned() const { return hipemu::st().gdim.x; }
    } x;
};
static hipemu_tid_t threadIdx;
static hipemu_bid_t blockIdx;
static hipemu_bdim_t blockDim;
static hipemu_gdim_t gridDim;

inline void __syncthreads() { hipemu::yield_to_main(); }

// ---- wave-level exchange (64-lane waves) ----
inline double __shfl(double v, int srcLane)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x;
    s.xchg[t][0] = v;
    __syncthreads();
    const double r = s.xchg[(t & ~63u) | (unsigned(srcLane) & 63u)][0];
    __syncthreads();
    return r;
}
inline double __shfl_xor(double v, int mask)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x;
    s.xchg[t][0] = v;
    __syncthreads();
    const double r = s.xchg[t ^ (unsigned(mask) & 63u)][0];
    __syncthreads();
    return r;
}
inline int __shfl_xor(int v, int mask)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x;
    s.xchg_i[t] = v;
    __syncthreads();
    const int r = int(s.xchg_i[t ^ (unsigned(mask) & 63u)]);
    __syncthreads();
    return r;
}
inline int __shfl(int v, int srcLane)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x;
    s.xchg_i[t] = v;
    __syncthreads();
    const int r = int(s.xchg_i[(t & ~63u) | (unsigned(srcLane) & 63u)]);
    __syncthreads();
    return r;
}

typedef double d4_t __attribute__((vector_size(32)));
// v_mfma_f64_16x16x4_f64:  D(16x16) = A(16x4) B(4x16) + C
//   A operand lane l: A[i = l&15][k = l>>4] ; B operand lane l: B[k = l>>4][j = l&15]
//   C/D lane l, reg r: row = (l>>4) + 4r, col = l&15        (cdna_hip_programming.md §3)
inline d4_t __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, d4_t c, int, int, int)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    s.xchg[t][0] = a;
    s.xchg[t][1] = b;
    __syncthreads();
    d4_t d = c;
    const unsigned col = l & 15u;
    for (int r = 0; r < 4; r++)
    {
        const unsigned row = (l >> 4) + 4u * unsigned(r);
        double acc = d[r];
        for (unsigned k = 0; k < 4; k++)
            acc += s.xchg[base + k * 16u + row][0] * s.xchg[base + k * 16u + col][1];
        d[r] = acc;
    }
    __syncthreads();
    return d;
}

// v_mfma_f64_4x4x4_4b_f64: four independent blocks  D_b(4x4) = A_b(4x4) B_b(4x4) + C_b,  one double per lane and operand.
//   lane l: block b = (l >> 2) & 3 ; A: A_b[i = l & 3][k = l >> 4] ; B: B_b[k = l >> 4][j = l & 3] ; C/D: D_b[i = l >> 4][j = l & 3]
//   (the 16x16x4 layout with the 16-index split into block and 4-index; measured on gfx950: tests/tools/mfma4x4_probe.hip).
//   cbsz / abid: the A block `abid` is broadcast to groups of 2^cbsz blocks (not used by the kernels: 0, 0).
inline double __builtin_amdgcn_mfma_f64_4x4x4f64(double a, double b, double c, int, int, int)
{
    hipemu::State &s = hipemu::st();
    const unsigned t = threadIdx.x, base = t & ~63u, l = t & 63u;
    s.xchg[t][0] = a;
    s.xchg[t][1] = b;
    __syncthreads();
    const unsigned blk = (l >> 2) & 3u, j = l & 3u, i = l >> 4;
    double acc = c;
    for (unsigned k = 0; k < 4; k++)
        acc += s.xchg[base + k * 16u + 4u * blk + i][0] * s.xchg[base + k * 16u + 4u * blk + j][1];
    __syncthreads();
    return acc;
}

// ---- buffer resources (raw, stride 0): base + voffset + soffset, out-of-range loads return 0, stores are dropped ----
struct hipemu_rsrc
{
    char *base;
    unsigned nbytes;
};
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned int hipemu_u32x2 __attribute__((vector_size(8)));
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short, int num_records, int)
{
    return hipemu_rsrc{reinterpret_cast<char *>(p), unsigned(num_records)};
}
// ---- traffic tracer (round 6; tools/traffic_table.py): with SCPP_EMU_TRAFFIC=<file> every in-range buffer access of the kernels is counted per
// (phase, record block, field, load / store) -- the per-field traffic table of DESIGN.md 4.2c is this count, not an estimate.  The kernels announce
// their record blocks (EMU_TRAFFIC_REGION) and the phase they are in (EMU_PHASE); accesses outside every announced block are counted as "other".
namespace hipemu
{
struct TrafficRegion
{
    std::string name;
    const char *base;
    size_t bytes;
    size_t field_bytes; // field-major block: bytes of one field row (pitch x 8) ; stage-major block: 8
    size_t rec_bytes;   // stage-major block: bytes of one stage's record (field = (offset % rec) / 8) ; field-major block: 0
};
struct Traffic
{
    bool on = false;
    std::string file;
    std::vector<TrafficRegion> regions;
    const char *phase = "outside";
    std::map<std::string, long long> cnt; // "phase|region|field|L" or "...|S" -> 8-byte lane accesses
    // distinct 128-byte lines touched by the loads / stores of ONE invocation of a phase (what reaches the L2 / the fabric once the lanes of an
    // instruction, and the instructions of a phase that re-touch a line, have coalesced): "U|phase|region|L" -> lines, summed over invocations
    std::map<std::pair<int, int>, std::set<uintptr_t>> lines; // (region index, store) -> lines of the current invocation
    void flush()
    {
        for (auto &kv : lines)
        {
            const std::string region = kv.first.first >= 0 ? regions[size_t(kv.first.first)].name : "other";
            cnt["U|" + std::string(phase) + "|" + region + (kv.first.second ? "|S" : "|L")] += (long long)kv.second.size();
        }
        lines.clear();
    }
    Traffic()
    {
        if (const char *e = std::getenv("SCPP_EMU_TRAFFIC"))
        {
            on = true;
            file = e;
        }
    }
    ~Traffic()
    {
        if (!on)
            return;
        flush();
        if (FILE *f = std::fopen(file.c_str(), "w"))
        {
            std::fprintf(f, "{\n");
            bool first = true;
            for (const auto &kv : cnt)
            {
                std::fprintf(f, "%s \"%s\": %lld", first ? "" : ",\n", kv.first.c_str(), kv.second);
                first = false;
            }
            std::fprintf(f, "\n}\n");
            std::fclose(f);
        }
    }
};
inline Traffic &traffic()
{
    static Traffic t;
    return t;
}
inline void traffic_region(const char *name, const void *base, size_t bytes, size_t field_bytes, size_t rec_bytes)
{
    Traffic &t = traffic();
    if (!t.on)
        return;
    for (auto &r : t.regions)
        if (r.name == name)
        {
            r = TrafficRegion{name, reinterpret_cast<const char *>(base), bytes, field_bytes, rec_bytes};
            return;
        }
    t.regions.push_back(TrafficRegion{name, reinterpret_cast<const char *>(base), bytes, field_bytes, rec_bytes});
}
inline void traffic_access(const char *addr, bool store)
{
    Traffic &t = traffic();
    if (!t.on)
        return;
    std::string key = std::string(t.phase) + "|";
    bool found = false;
    int ri = -1;
    for (size_t q = 0; q < t.regions.size(); q++)
    {
        const auto &r = t.regions[q];
        if (addr >= r.base && addr < r.base + r.bytes)
        {
            const size_t o = size_t(addr - r.base);
            const size_t field = r.rec_bytes ? (o % r.rec_bytes) / 8 : o / r.field_bytes;
            key += r.name + "|" + std::to_string(field);
            found = true;
            ri = int(q);
            break;
        }
    }
    t.lines[{ri, store ? 1 : 0}].insert(reinterpret_cast<uintptr_t>(addr) >> 7);
    if (!found)
        key += "other|0";
    key += store ? "|S" : "|L";
    t.cnt[key] += 1;
}
inline void traffic_manual(const char *what, long long n, bool store) // accesses through plain pointers (not buffer resources), counted by hand
{
    Traffic &t = traffic();
    if (t.on)
        t.cnt[std::string(t.phase) + "|" + what + "|0|" + (store ? "S" : "L")] += n;
}
} // namespace hipemu
#define EMU_TRAFFIC_MANUAL(what, n, store) hipemu::traffic_manual(what, n, store)
// (lane 0 enters a phase first -- the lanes are fibers run in order between synchronisation points, and every phase ends on one: the lines of the
//  invocation that just ended are complete when lane 0 announces the next)
#define EMU_PHASE(name)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        if (hipemu::traffic().on && threadIdx.x == 0)                                                                  \
            hipemu::traffic().flush();                                                                                 \
        hipemu::traffic().phase = (name);                                                                              \
    } while (0)
#define EMU_TRAFFIC_REGION(name, base, bytes, field_bytes, rec_bytes) hipemu::traffic_region(name, base, bytes, field_bytes, rec_bytes)
inline hipemu_u32x2 __builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int)
{
    hipemu_u32x2 v = {0u, 0u};
    const unsigned o = unsigned(voffset) + unsigned(soffset);
    if (o + 8u <= r.nbytes)
    {
        std::memcpy(&v, r.base + o, 8);
        hipemu::traffic_access(r.base + o, false);
    }
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b64(hipemu_u32x2 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset, int)
{
    const unsigned o = unsigned(voffset) + unsigned(soffset);
    if (o + 8u <= r.nbytes)
    {
        std::memcpy(r.base + o, &v, 8);
        hipemu::traffic_access(r.base + o, true);
    }
}

// ---- host runtime shim ----
typedef int hipError_t;
typedef void *hipStream_t;
struct hipemu_event
{
    std::chrono::steady_clock::time_point t;
};
typedef hipemu_event *hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4
inline hipError_t hipMalloc(void **p, size_t n)
{
    *p = std::aligned_alloc(256, ((n ? n : 1) + 255) & ~size_t(255)); // 256-byte aligned like the device allocator (the traffic tracer counts 128-byte lines)
    if (*p)
        std::memset(*p, 0xFF, n); // poison: any read-before-write of device memory shows up as NaN
    return *p ? 0 : 2;
}
template <class T>
hipError_t hipMalloc(T **p, size_t n)
{
    return hipMalloc(reinterpret_cast<void **>(p), n);
}
inline hipError_t hipFree(void *p)
{
    std::free(p);
    return 0;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int)
{
    std::memcpy(d, s, n);
    return 0;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { return hipMemcpy(d, s, n, 0); }
inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, int)
{
    for (size_t r = 0; r < height; r++)
        std::memcpy(static_cast<char *>(d) + r * dpitch, static_cast<const char *>(s) + r * spitch, width);
    return 0;
}
// lanes run as fibers of one host thread: a plain read-modify-write is atomic
inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, int k, hipStream_t)
{
    return hipMemcpy2D(d, dpitch, s, spitch, width, height, k);
}
inline int atomicAdd(int *p, int v)
{
    const int o = *p;
    *p = o + v;
    return o;
}
inline hipError_t hipMemset(void *d, int v, size_t n)
{
    std::memset(d, v, n);
    return 0;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t *s)
{
    *s = nullptr;
    return 0;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; } // emulated streams run in issue order
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return 0;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { return hipFree(p); }
inline void __threadfence() {}
inline hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return 0;
}
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = new hipemu_event;
    return 0;
}
inline hipError_t hipEventDestroy(hipEvent_t e)
{
    delete e;
    return 0;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    e->t = std::chrono::steady_clock::now();
    return 0;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
inline long long clock64() { return 0; }
