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

// ---- diagnostic switches: the one table (see nc_common.h) --------------------------------------------------------------------------
namespace {
struct EnvRow { const char* name; char kind; const char* doc; };
const EnvRow kEnv[] = {
    {"NC_NO_FLAT", 'b', "one-clip column tiles instead of the flattened (clip, column) axis"},
    {"NC_NO_FLAT_GN", 'b', "one-clip tiles wherever GroupNorm sums are emitted"},
    {"NC_CO_GROUP", 'i', "cap of the row-tile groups (1 = one weight panel per XCD)"},
    {"NC_NO_WIDE_FUSE", 'b', "no whole-channel fused residual units (C = 256)"},
    {"NC_WIDE_FUSE_192", 'b', "whole-channel fused unit at C = 192 as well"},
    {"NC_NO_FUSE", 'b', "every residual unit in two launches"},
    {"NC_NO_TILE_ALTS", 'b', "primary row-tile height only"},
    {"NC_TM_PICK", 'i', "force a packed row-tile variant"},
    {"NC_TM_FORCE", 'i', "force the row-tile height where it divides Cout"},
    {"NC_TN_THRESH", 'i', "column-tile width threshold"},
    {"NC_NO_TN_ROUNDS", 'b', "no round-count rule for the column-tile width"},
    {"NC_NO_XR", 'b', "generic B-fragment addressing in the conv template (no constant-pitch immediate offsets)"},
    {"NC_NO_XV", 'b', "legacy instances with item-wise window staging everywhere (no XV-only instances)"},
    {"NC_DUO", 'b', "EXPERIMENTS=1 builds: k = 7 residual-unit convolutions as DUO instances (two tiles per 8-wavefront workgroup, the second half a block behind; measured slower)"},
    {"NC_XV_K7_MIN_COLS", 'i', "columns per clip from which the k = 7 convolutions take the XV-only instances (0: wherever the rows are 64-byte aligned)"},
    {"NC_NO_XV_K7", 'b', "legacy instances for the k = 7 residual-unit convolutions (no XV-only instances on the long rows)"},
    {"NC_NO_NARROW", 'b', "no 3-wave narrow variants"},
    {"NC_NO_SLIM", 'b', "no half-size reduction blocks for narrow long rows"},
    {"NC_NO_SUBPIXEL", 'b', "per-phase launches for power-of-two strided transposed convolutions"},
    {"NC_NO_SUBPIXEL_ANY", 'b', "per-phase launches for the other strides (3, 5)"},
    {"NC_NO_CONV1X1", 'b', "pointwise layers through the windowed template"},
    {"NC_PW_STREAM", 'b', "EXPERIMENTS=1 builds: streaming pointwise variant for the narrow long rows (overtaken in round 4)"},
    {"NC_NO_SKINNY", 'b', "no skinny projection kernel (Cout <= 16)"},
    {"NC_NO_THIN", 'b', "PCM heads through the matrix-core template"},
    {"NC_THIN_NO_VEC", 'p', "scalar window loads in the thin-output kernel"},
    {"NC_NO_THIN_INM", 'b', "summed copy + plain head instead of the input-mode head"},
    {"NC_NO_STEM", 'b', "stems through the matrix-core template"},
    {"NC_NO_TINY_TILES", 'b', "no smallest-tile rule for tiny grids"},
    {"NC_TINY_BLOCKS", 'i', "grid size below which the tiny-grid rule applies (256)"},
    {"NC_NO_DIST_SMALL", 'b', "segmented staging on small grids"},
    {"NC_DIST_MAX_GRID", 'i', "largest grid that takes distributed staging (768)"},
    {"NC_LDS_MIN", 'i', "minimum dynamic LDS per workgroup (placement experiments)"},
    {"NC_NO_CONV_SMALL", 'b', "no short-row 16x16x4 kernel"},
    {"NC_SMALL_MAX_GRID", 'i', "largest short-row grid (2048)"},
    {"NC_SMALL_WIDE_BELOW", 'i', "template grid below which the 32-column short-row form is taken (512)"},
    {"NC_SMALL_K1_COLS", 'i', "columns up to which k = 1 layers take the short-row kernel (4096; 0 = off)"},
    {"NC_SMALL_TN", 'i', "force the short-row column tiles (1 | 2 | 4)"},
    {"NC_SMALL_ROLLED", 'b', "rolled short-row loop"},
    {"NC_DAC_RVQ_STAGEWISE", 'p', "DAC quantizer stage by stage"},
    {"NC_RVQ_8WAVES", 'b', "EXPERIMENTS=1 builds: 8-wavefront Euclidean RVQ workgroups (measured equal)"},
    {"NC_EUCLID_NO_MFMA", 'p', "vector Euclidean codebook search"},
    {"NC_ENCODEC_NO_FUSE", 'b', "padded / activated copies instead of the fused SConv1d input mode"},
    {"NC_ENCODEC_NO_OVERLAP", 'b', "segment groups one after the other"},
    {"NC_NO_GN_FUSE", 'b', "stand-alone GroupNorm sums"},
    {"NC_NO_GN_FINISH", 'b', "separate GroupNorm final launch"},
    {"NC_NO_IN2", 'b', "summed copies instead of the two-input staging mode"},
    {"NC_NO_CONV3S", 'b', "windowed k = 3 instead of the streaming kernel"},
    {"NC_DAC_PITCH", 'b', "DAC activation rows at pitch L rounded up to 16 samples (64-byte aligned rows: the 696-step layers take the XV-only instances; measured slower)"},
    {"NC_NO_DOWN2", 'b', "Encodec 48 kHz encoder: the windowed two-input instance for the stride-2 down-convolution instead of the streaming kernel"},
    {"NC_NO_DOWN4", 'b', "Encodec 48 kHz encoder: summed copy + windowed instance for the stride-4 down-convolution instead of the streaming kernel"},
    {"NC_NO_DOWN5", 'b', "Encodec 48 kHz encoder: summed copy + windowed instance for the stride-5 down-convolution instead of the streaming kernel"},
    {"NC_NO_UP_PITCH", 'b', "Encodec transposed convolutions write dense rows (the trimmed view of the stride-5 layer then starts on an odd sample)"},
    {"NC_NO_UP4", 'b', "Encodec 48 kHz decoder: summed copy + windowed instance for the stride-4 up-convolution instead of the streaming kernel"},
    {"NC_NO_UP2", 'b', "Encodec 48 kHz decoder: the windowed two-input instance for the stride-2 up-convolution instead of the streaming kernel"},
    {"NC_RMS_TWO_PASS", 'b', "Encodec RMS scale as two launches (chunk sums, final) instead of the one-launch form"},
    {"NC_NO_RES_A", 'b', "residual blocks of the Encodec 48 kHz outer stages: shortcut and k = 3 branch as two launches instead of the fused first pass"},
    {"NC_SYNC_ACQUIRE", 'b', "agent-scope acquire fence behind the persistent LSTM's flag poll and in front of the in-launch GroupNorm finish"},
    {"NC_LSTM_SPLIT", 'b', "EXPERIMENTS=1 builds: per-layer persistent LSTM with load-only / store-only wave roles and value-validated exchange regions (lstm1_kernel; measured slower)"},
    {"NC_LSTM_STEPWISE", 'b', "one LSTM launch per step"},
    {"NC_LSTM_CHUNKS", 'i', "layer-pipeline chunks of the per-layer persistent LSTM (6; 1 = layers in sequence)"},
    {"NC_LSTM_EVEN_CHUNKS", 'b', "equal LSTM chunks"},
    {"NC_LSTM_UB", 'i', "hidden-unit blocks per LSTM workgroup (2 | 4)"},
    {"NC_LSTM_NO_HTILE", 'b', "persistent LSTM: every wavefront fetches its own h operands, W_hh in LDS (the form before the LDS h tile)"},
    {"NC_LSTM_NO_ELU", 'b', "the consumer applies the ELU behind an SLSTM"},
    {"NC_LSTM_FUSED", 'b', "EXPERIMENTS=1 builds: fused two-layer persistent LSTM (nc_lstm.hip; measured slower)"},
    {"NC_LSTM2_TRACE", 's', "EXPERIMENTS=1 builds: file for the in-kernel stamps of the fused LSTM (tools/probe/lstm2_trace.py)"},
    {"NC_LSTM_FAKE_TIMEOUT", 'b', "tests: report the first persistent LSTM launch as timed out"},
    {"NC_SNAC_NO_FUSE", 'b', "SNAC residual units in two launches (depthwise, pointwise)"},
    {"NC_SNAC_FUSE_MIN_COLS", 'i', "columns (clips x steps) from which the one-launch SNAC residual unit is taken (65536)"},
    {"NC_SNAC_FUSE_MAX_ELEMS", 'i', "channels x steps of ONE clip up to which the one-launch SNAC residual unit is taken (2^30: its 32-bit lane offsets; tests lower it)"},
    {"NC_SNAC_UNIT_TRACE", 's', "file for the in-kernel stamps of the first one-launch SNAC residual unit at C = 96"},
    {"NC_DW_NO_VEC", 'b', "scalar depthwise kernel"},
    {"NC_LN_TILE", 'i', "LayerNorm tile width (8 | 16; 0 = one thread per column)"},
    {"NC_ATTN_NO_MFMA", 'b', "vector local-attention kernel"},
    {"NC_RCCL_LIB", 's', "the one RCCL library to open"},
    {"NC_LAUNCH_LOG", 's', "launch log for the per-class PMC attribution"},
    {"NC_CONV_TRACE_FILE", 's', "-DNC_CONV_TRACE builds: file for the in-kernel phase trace of the convolution template (tools/probe/conv_trace.py)"},
    {"NC_CONV_TRACE_SEL", 's', "-DNC_CONV_TRACE builds: \"K,Cin,dilation\" of the launch to trace (default 7,384,1)"},
    {"NC_LIGHT", 'i', "EXPERIMENTS=1 builds: light k = 7 variant"},
    {"NC_WIDE", 'i', "EXPERIMENTS=1 builds: 8-wave k = 7 variant"},
    {"NC_SPEC", 'i', "EXPERIMENTS=1 builds: producer / consumer k = 7 variant"},
    {"NC_DIST", 'i', "EXPERIMENTS=1 builds: distributed-staging k = 7 variant"},
};
const EnvRow& env_row(const char* name, char kind) {
    for (const EnvRow& r : kEnv)
        if (std::strcmp(r.name, name) == 0) {
            if (r.kind != kind) fail(NC_ESTATE, "switch %s is registered as kind '%c', read as '%c'", name, r.kind, kind);
            return r;
        }
    fail(NC_ESTATE, "switch %s is not in the table of nc_util.hip", name);
}
}  // namespace
bool env_flag(const char* name) { const char* v = std::getenv(env_row(name, 'b').name); return v && v[0] == '1'; }
bool env_present(const char* name) { return std::getenv(env_row(name, 'p').name) != nullptr; }
long env_int(const char* name, long dflt) { const char* v = std::getenv(env_row(name, 'i').name); return v ? atol(v) : dflt; }
const char* env_str(const char* name) { const char* v = std::getenv(env_row(name, 's').name); return v && v[0] ? v : nullptr; }
const char* env_switch_table() {
    static const std::string text = [] {
        std::string t;
        for (const EnvRow& r : kEnv) { t += r.name; t += '\t'; t += r.kind; t += '\t'; t += r.doc; t += '\n'; }
        return t;
    }();
    return text.c_str();
}

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
