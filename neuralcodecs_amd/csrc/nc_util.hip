// Blob parser, profiler, thread-local error string, host-side weight-norm fold.
#include <cmath>
#include <mutex>
#include <set>
#include <tuple>

#include "nc_common.h"
#include "nc_model.h"

namespace nc {

static thread_local std::string g_last_error;
void set_last_error(const char* msg) { g_last_error = msg ? msg : ""; }
const char* get_last_error() { return g_last_error.c_str(); }

void ensure_dynamic_lds(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::set<std::tuple<int, const void*, size_t>> done;
    int dev = 0;
    NC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_tuple(dev, fn, bytes);
    if (done.count(key)) return;
    NC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.insert(key);
}

void Blob::parse(const void* data, size_t nbytes) {
    const uint8_t* buf = static_cast<const uint8_t*>(data);
    if (!buf || nbytes < 24 || std::memcmp(buf, "NCWB0001", 8) != 0) fail(NC_EINVAL, "not an NCWB0001 weight blob");
    storage.assign(buf, buf + nbytes);
    buf = storage.data();
    uint64_t n, idx_len;
    std::memcpy(&n, buf + 8, 8);
    std::memcpy(&idx_len, buf + 16, 8);
    // every index read is checked against the data origin and every (offset, size) against the image with subtraction-only
    // comparisons, so a truncated or crafted file (nc_codec_load_weights takes any path) yields NC_EINVAL, never an out-of-bounds read
    if (idx_len > nbytes - 24) fail(NC_EINVAL, "weight blob index exceeds the image");
    const uint64_t data0 = (24 + idx_len + 63) & ~uint64_t(63);
    if (data0 > nbytes) fail(NC_EINVAL, "weight blob index exceeds the image");
    const uint64_t idx_end = 24 + idx_len;
    uint64_t p = 24;
    tensors.clear();
    auto need = [&](uint64_t k) {
        if (k > idx_end - p) fail(NC_EINVAL, "truncated weight blob index");
    };
    for (uint64_t i = 0; i < n; ++i) {
        need(2);
        uint16_t ln;
        std::memcpy(&ln, buf + p, 2);
        p += 2;
        need((uint64_t)ln + 2);
        BlobTensor t;
        t.name.assign(reinterpret_cast<const char*>(buf + p), ln);
        p += ln;
        t.dtype = buf[p];
        const int nd = buf[p + 1];
        p += 2;
        if (t.dtype != 0 && t.dtype != 1) fail(NC_EINVAL, "tensor %s: unknown dtype %d (0 = float32, 1 = int64)", t.name.c_str(), t.dtype);
        if (nd > 8) fail(NC_EINVAL, "tensor %s has %d dimensions (at most 8)", t.name.c_str(), nd);
        need((uint64_t)nd * 8 + 16);
        uint64_t numel = 1;
        for (int d = 0; d < nd; ++d) {
            uint64_t v;
            std::memcpy(&v, buf + p, 8);
            p += 8;
            if (v > (uint64_t)1 << 40 || (v != 0 && numel > ((uint64_t)1 << 40) / v)) fail(NC_EINVAL, "tensor %s: dimensions overflow", t.name.c_str());
            numel *= v;
            t.dims.push_back((int64_t)v);
        }
        uint64_t off, nb;
        std::memcpy(&off, buf + p, 8);
        std::memcpy(&nb, buf + p + 8, 8);
        p += 16;
        const uint64_t room = nbytes - data0;
        if (off > room || nb > room - off) fail(NC_EINVAL, "tensor %s exceeds the weight blob", t.name.c_str());
        if (nb != numel * (t.dtype == 0 ? 4 : 8))
            fail(NC_EINVAL, "tensor %s: %llu bytes for %llu elements of dtype %d", t.name.c_str(), (unsigned long long)nb, (unsigned long long)numel, t.dtype);
        t.data = buf + data0 + off;
        t.nbytes = (int64_t)nb;
        tensors[t.name] = t;
    }
}

const BlobTensor* Blob::find(const std::string& name) const {
    auto it = tensors.find(name);
    if (it == tensors.end()) return nullptr;
    // every consumer of the engine reads float32 (numel floats): anything else must not reach a memcpy
    if (it->second.dtype != 0) fail(NC_EINVAL, "weight tensor '%s' is not float32", name.c_str());
    return &it->second;
}

const BlobTensor& Blob::get(const std::string& name) const {
    const BlobTensor* t = find(name);
    if (!t) fail(NC_ENOTFOUND, "weight tensor '%s' not found in blob", name.c_str());
    return *t;
}

// w = (v / (fl(sqrt(fl(sum_f64 fl(v*v)))) + 1e-7f)) * g   per dim-0 slice
// (WNConv1d.cs:145-150 / WNConvTranspose1d.cs:146-150, deviation D2; canonical fold of DESIGN.md)
void fold_weight_norm_dac(const float* v, const float* g, int64_t d0, int64_t inner, float* w) {
    for (int64_t i = 0; i < d0; ++i) {
        double ss = 0.0;
        for (int64_t j = 0; j < inner; ++j) {
            const float q = v[i * inner + j] * v[i * inner + j];
            ss += (double)q;
        }
        const float denom = std::sqrt((float)ss) + 1e-7f;
        for (int64_t j = 0; j < inner; ++j) w[i * inner + j] = (v[i * inner + j] / denom) * g[i];
    }
}

// ---- profiler ----------------------------------------------------------------------------
hipEvent_t Profiler::get_event() {
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e;
    NC_HIP(hipEventCreate(&e));
    return e;
}

void Profiler::begin(hipStream_t s, int cls, double flops, double bytes) {
    Pending pd{get_event(), get_event(), cls, flops, bytes};
    NC_HIP(hipEventRecord(pd.a, s));
    pending.push_back(pd);
}

void Profiler::end(hipStream_t s) { NC_HIP(hipEventRecord(pending.back().b, s)); }

void Profiler::resolve() {
    for (auto& pd : pending) {
        NC_HIP(hipEventSynchronize(pd.b));
        float ms = 0.f;
        NC_HIP(hipEventElapsedTime(&ms, pd.a, pd.b));
        acc[pd.cls].launches += 1;
        acc[pd.cls].ms += ms;
        acc[pd.cls].flops += pd.flops;
        acc[pd.cls].bytes += pd.bytes;
        pool.push_back(pd.a);
        pool.push_back(pd.b);
    }
    pending.clear();
}

void Profiler::reset() {
    resolve();
    for (auto& e : acc) e = nc_profile_entry{};
}

Profiler::~Profiler() {
    for (auto& pd : pending) {
        (void)hipEventDestroy(pd.a);
        (void)hipEventDestroy(pd.b);
    }
    for (auto e : pool) (void)hipEventDestroy(e);
}

}  // namespace nc
