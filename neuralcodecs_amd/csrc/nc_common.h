// Internal helpers shared by the engine's translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nc_mi355x.h"

namespace nc {

struct Error : std::runtime_error {
    nc_status code;
    Error(nc_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(nc_status c, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(c, buf);
}

#define NC_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            ::nc::fail(e__ == hipErrorOutOfMemory ? NC_ENOMEM : NC_EDEVICE, "%s failed: %s (%s:%d)", #expr, \
                       hipGetErrorString(e__), __FILE__, __LINE__);                                    \
    } while (0)

void set_last_error(const char* msg);

// Diagnostic switches (environment): ONE table in nc_util.hip holds every name the engine reads, its kind and what it
// switches.  The accessors fail on a name the table does not hold, so a switch cannot exist without its row; nc_debug_switches()
// (C ABI) lists the table, which is what tools/probe/envmatrix.sh enumerates and tests/test_abi_cpu.py holds DESIGN.md to.
// Call sites keep their `static const` caches: a switch is read once per process.
bool env_flag(const char* name);                  // kind 'b': set to 1
bool env_present(const char* name);               // kind 'p': set at all
long env_int(const char* name, long dflt);        // kind 'i'
const char* env_str(const char* name);            // kind 's': nullptr when unset
const char* env_switch_table();                   // "NAME\tkind\twhat it does\n" per switch

// Opt a kernel into > 64 KB of dynamic LDS on the CURRENT device.  The attribute is per (device, function): the cache is keyed on
// both and guarded, so models on different GPUs (and host threads calling in concurrently) each get their opt-in.
void ensure_dynamic_lds(const void* fn, size_t bytes);

// Grow-only device arena: activations for the largest (B,T) seen so far stay resident
// (288 GB of HBM per GPU: nothing is ever freed or re-allocated inside a timed region).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }   // a model's packed weights, codebooks and workspaces go back to the device with the handle (Dispose)
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) NC_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = (bytes + (1 << 20) - 1) & ~size_t((1 << 20) - 1);
        NC_HIP(hipMalloc(&p, want));
        cap = want;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// ---- weight blob (NCWB0001) ----------------------------------------------------------------
struct BlobTensor {
    std::string name;
    int dtype = 0;
    std::vector<int64_t> dims;
    const void* data = nullptr;
    int64_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct Blob {
    std::vector<uint8_t> storage;
    std::map<std::string, BlobTensor> tensors;
    void parse(const void* data, size_t nbytes);  // copies
    const BlobTensor& get(const std::string& name) const;
    const BlobTensor* find(const std::string& name) const;
};

// ---- profiling -----------------------------------------------------------------------------
struct Profiler {
    bool on = false;
    struct Pending {
        hipEvent_t a, b;
        int cls;
        double flops, bytes;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    nc_profile_entry acc[NC_KC_COUNT] = {};
    hipEvent_t get_event();
    void begin(hipStream_t s, int cls, double flops, double bytes);
    void end(hipStream_t s);
    void resolve();
    void reset();
    ~Profiler();
};

// begin/end of one profiled region around a launch (no-op unless the handle's profiler is on)
struct ProfScope {
    Profiler* p;
    hipStream_t s;
    ProfScope(Profiler* prof, hipStream_t st, int cls, double flops, double bytes) : p(prof && prof->on ? prof : nullptr), s(st) {
        if (p) p->begin(s, cls, flops, bytes);
    }
    ~ProfScope() {
        if (p) (void)hipEventRecord(p->pending.back().b, s);
    }
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

}  // namespace nc
