// Multi-GPU groups behind the C ABI: batch sharding over the GPUs of one node + the RCCL all-gather of the emitted codes.
//
// The Encode/Decode path has no cross-clip dependency (no BatchNorm; GroupNorm / LayerNorm / RMS scale are per sample, RVQ per frame),
// so every device runs the single-GPU engine on its contiguous block of clips with replicated weights; the ONE collective is the
// all-gather of the int64 codes (DAC [B,9,87]: 200 KB per rank at BASELINE config C4; SNAC: the levels of a clip side by side, one
// collective for all levels).  It is issued on a side stream per device, ordered after the encode by an event, so whatever the caller
// queues next on the codec's stream (the local decode needs only local latents) overlaps it.
//
// librccl is opened at run time by these entry points only (dlopen, preferring a copy already mapped into the process): the engine
// library itself keeps libamdhip64 as its only dependency.  Two modes:
//   rank  mode: one process per GPU (the torch.distributed / MPI / C# multi-process layout); ranks share a 128-byte unique id.
//   local mode: one process drives ndev GPUs (the natural layout of a C# host): ncclCommInitAll + grouped collectives.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>

#include "nc_model.h"

using namespace nc;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    static std::string err;
    std::call_once(once, [] {
        // NC_RCCL_LIB=<path> names the one library to open (deployments with a private RCCL; tests force the load failure with it)
        const char* forced = env_str("NC_RCCL_LIB");
        const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::vector<const char*> names;
        if (forced && forced[0]) names.push_back(forced);
        else names.assign(std::begin(defaults), std::end(defaults));
        if (!(forced && forced[0]))
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // a copy already in the process (e.g. PyTorch's) wins
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) {
            const char* e = dlerror();   // (one call: dlerror() clears the message it returns)
            err = e ? e : "librccl not found";
            return;
        }
        auto sym = [&](const char* s) { void* p = dlsym(r.lib, s); if (!p) err = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (!r.lib || !err.empty()) fail(NC_EDEVICE, "RCCL is not available: %s", err.c_str());
    return r;
}

#define NC_RCCL(expr)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r__ = (expr);                                                                        \
        if (r__ != ncclSuccess) fail(NC_EDEVICE, "%s failed: %s", #expr, rccl().GetErrorString(r__));      \
    } while (0)

// The local-mode entry points walk the members' devices with hipSetDevice: the caller's current device is put back on the way out
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

template <class F>
nc_status guard(F&& f) {
    try {
        f();
        return NC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return NC_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return NC_ESTATE;
    }
}

}  // namespace

// pinned host staging (grow-only): the host-pointer entry points take pageable memory, and a hipMemcpyAsync from pageable memory is
// synchronous -- staged through pinned buffers by one host thread per device, the uploads of all devices overlap
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        NC_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct nc_group {
    int world = 1, rank = -1;        // rank == -1: local mode (this process drives all `world` devices)
    int code_bits = 0;               // > 0: the all-gather moves bit-packed codes (nc_group_set_code_bits)
    bool peer_copy = false;          // local mode, NC_GROUP_PEER_COPY: the gather pulls the slots with peer copies instead of RCCL
    struct Member {
        nc_codec* h = nullptr;       // borrowed: the codec must outlive the group (nc_group_destroy before nc_codec_destroy)
        int device = 0;              // (kept here so that the destructor never reads through the borrowed handle)
        ncclComm_t comm = nullptr;
        hipStream_t side = nullptr;
        hipEvent_t ev_enc = nullptr, ev_gather = nullptr;
        hipEvent_t ev_slot = nullptr, ev_pulled = nullptr;   // peer-copy transport: "my slot is final" / "I hold every slot"
        DevBuf pcm, codes_all, z;    // local mode staging
        DevBuf packed_all;           // [world][slot bytes]: the packed payload of the collective (code_bits > 0)
        PinBuf pin_in, pin_out, pin_z;
    };
    std::vector<Member> m;
    ~nc_group() {
        for (auto& x : m) {
            (void)hipSetDevice(x.device);
            if (x.comm) (void)rccl().CommDestroy(x.comm);
            if (x.side) (void)hipStreamDestroy(x.side);
            if (x.ev_enc) (void)hipEventDestroy(x.ev_enc);
            if (x.ev_gather) (void)hipEventDestroy(x.ev_gather);
            if (x.ev_slot) (void)hipEventDestroy(x.ev_slot);
            if (x.ev_pulled) (void)hipEventDestroy(x.ev_pulled);
            x.pcm.release(); x.codes_all.release(); x.z.release(); x.packed_all.release();
            x.pin_in.release(); x.pin_out.release(); x.pin_z.release();
        }
    }
};

namespace {

void init_member(nc_group::Member& x, nc_codec* h) {
    if (!h || !h->impl) fail(NC_EINVAL, "null codec handle");
    x.h = h;
    x.device = h->impl->device;
    h->impl->use_device();
    NC_HIP(hipStreamCreateWithFlags(&x.side, hipStreamNonBlocking));
    NC_HIP(hipEventCreateWithFlags(&x.ev_enc, hipEventDisableTiming));
    NC_HIP(hipEventCreateWithFlags(&x.ev_gather, hipEventDisableTiming));
    NC_HIP(hipEventCreateWithFlags(&x.ev_slot, hipEventDisableTiming));
    NC_HIP(hipEventCreateWithFlags(&x.ev_pulled, hipEventDisableTiming));
}

// codes of this member -> its slot of codes_all; returns elements per rank.  The encode writes straight into the slot, so the
// all-gather runs in place (sendbuf == recvbuf + rank * count).
int64_t encode_into_slot(nc_group::Member& x, int kind, int slot, const float* pcm, int B, int64_t T, int sample_rate, int n_q, int64_t* codes_all,
                         float* z, float* lat) {
    Codec& c = *x.h->impl;
    c.use_device();
    int64_t per_rank = 0;
    if (kind == 0) {
        if (x.h->kind != 0) fail(NC_EINVAL, "handle is not a DAC codec");
        DacModel& m = static_cast<DacModel&>(c);
        const int nq = (n_q <= 0 || n_q > m.cfg.n_codebooks) ? m.cfg.n_codebooks : n_q;
        per_rank = (int64_t)B * nq * m.frames(T);
        m.encode_dev(pcm, B, T, sample_rate, n_q, codes_all + (int64_t)slot * per_rank, z, lat);
    } else {
        if (x.h->kind != 1) fail(NC_EINVAL, "handle is not a SNAC codec");
        SnacModel& m = static_cast<SnacModel&>(c);
        per_rank = (int64_t)B * m.codes_per_clip(m.padded_len(T) / m.hop);
        m.encode_dev(pcm, B, T, codes_all + (int64_t)slot * per_rank, nullptr, nullptr, true);
    }
    NC_HIP(hipEventRecord(x.ev_enc, c.stream));
    NC_HIP(hipStreamWaitEvent(x.side, x.ev_enc, 0));
    return per_rank;
}

// The all-gather of one member on its side stream, in two halves so that local mode keeps ONLY the collectives between ncclGroupStart and
// ncclGroupEnd (no allocation, kernel launch or throwing call inside an open RCCL group).  code_bits == 0: the int64 slots in place.
// code_bits > 0 (nc_group_set_code_bits): the member packs its own slot into the wire layout of the reference's BitPacker
// (Modules/Encodec/BitPacker.cs: `bits` per value, LSB first; one packed row per clip, csrc/nc_pack.hip), the collective moves the packed
// rows -- 64 / bits times fewer bytes -- and every slot is unpacked back into the int64 tensor the caller sees.  rows = clips per slot,
// per_clip = values per clip.
void prepare_slot(nc_group* g, nc_group::Member& x, int slot, int64_t* codes_all, int rows, int64_t per_clip) {
    if (g->code_bits <= 0) return;
    const int64_t per_rank = (int64_t)rows * per_clip;
    const int64_t row_bytes = nc_packed_bytes(per_clip, g->code_bits), slot_bytes = (int64_t)rows * row_bytes;
    x.packed_all.reserve((size_t)slot_bytes * g->world);
    uint8_t* pk = x.packed_all.as<uint8_t>();
    nc_status st = nc_pack_codes_dev(x.device, codes_all + (int64_t)slot * per_rank, rows, 1, per_clip, g->code_bits, pk + (int64_t)slot * slot_bytes, x.side);
    if (st != NC_OK) fail(st, "%s", get_last_error());
}
ncclResult_t issue_gather(nc_group* g, nc_group::Member& x, int slot, int64_t* codes_all, int rows, int64_t per_clip) {
    const int64_t per_rank = (int64_t)rows * per_clip;
    if (g->code_bits <= 0) return rccl().AllGather(codes_all + (int64_t)slot * per_rank, codes_all, (size_t)per_rank, ncclInt64, x.comm, x.side);
    const int64_t slot_bytes = (int64_t)rows * nc_packed_bytes(per_clip, g->code_bits);
    uint8_t* pk = x.packed_all.as<uint8_t>();
    return rccl().AllGather(pk + (int64_t)slot * slot_bytes, pk, (size_t)slot_bytes, ncclUint8, x.comm, x.side);
}
void gather_slots(nc_group* g, nc_group::Member& x, int slot, int64_t* codes_all, int rows, int64_t per_clip) {
    prepare_slot(g, x, slot, codes_all, rows, per_clip);
    NC_RCCL(issue_gather(g, x, slot, codes_all, rows, per_clip));
}
// Peer-copy transport (local mode created with NC_GROUP_PEER_COPY): the same in-place all-gather -- slot d of member d's buffer ends up in
// slot d of EVERY member's buffer -- moved by hipMemcpyPeerAsync on the members' side streams instead of RCCL: member e waits for the
// "slot final" event of member d and pulls slot d.  No librccl, and members may share a device (RCCL refuses a communicator with two ranks
// on one GPU): a W-way group over fewer GPUs, which is how a one-GPU box runs BASELINE configs C4 / C5 through all eight slots.
// bufs[d] = member d's buffer of W slots of slot_bytes each; everything queued on a side stream so far belongs to the slot.
void copy_allgather(nc_group* g, const std::vector<uint8_t*>& bufs, int64_t slot_bytes) {
    const int W = g->world;
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        NC_HIP(hipSetDevice(x.device));
        NC_HIP(hipEventRecord(x.ev_slot, x.side));
    }
    for (int e = 0; e < W; ++e) {
        nc_group::Member& x = g->m[(size_t)e];
        NC_HIP(hipSetDevice(x.device));
        for (int k = 1; k < W; ++k) {                     // (staggered: at any moment the members read from W different sources)
            const int d = (e + k) % W;
            nc_group::Member& src = g->m[(size_t)d];
            NC_HIP(hipStreamWaitEvent(x.side, src.ev_slot, 0));
            NC_HIP(hipMemcpyPeerAsync(bufs[(size_t)e] + (int64_t)d * slot_bytes, x.device, bufs[(size_t)d] + (int64_t)d * slot_bytes, src.device,
                                      (size_t)slot_bytes, x.side));
        }
        NC_HIP(hipEventRecord(x.ev_pulled, x.side));
    }
    // a member's buffer may be rewritten (its next encode) only when every reader is done with it: each side stream meets all pulls
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        NC_HIP(hipSetDevice(x.device));
        for (int e = 0; e < W; ++e)
            if (e != d) NC_HIP(hipStreamWaitEvent(x.side, g->m[(size_t)e].ev_pulled, 0));
    }
}

// local mode: every member's collective inside ONE RCCL group; the group is closed on the error path too
void grouped_gather(nc_group* g, const std::vector<int64_t*>& codes_all, int rows, int64_t per_clip) {
    const int W = g->world;
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        NC_HIP(hipSetDevice(x.device));
        prepare_slot(g, x, d, codes_all[(size_t)d], rows, per_clip);
    }
    if (g->peer_copy) {
        std::vector<uint8_t*> bufs((size_t)W);
        const bool packed = g->code_bits > 0;
        for (int d = 0; d < W; ++d)
            bufs[(size_t)d] = packed ? g->m[(size_t)d].packed_all.as<uint8_t>() : reinterpret_cast<uint8_t*>(codes_all[(size_t)d]);
        copy_allgather(g, bufs, packed ? (int64_t)rows * nc_packed_bytes(per_clip, g->code_bits) : (int64_t)rows * per_clip * 8);
        return;
    }
    NC_RCCL(rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int d = 0; d < W && bad == ncclSuccess; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        if (hipSetDevice(x.device) != hipSuccess) { bad = ncclUnhandledCudaError; break; }
        bad = issue_gather(g, x, d, codes_all[(size_t)d], rows, per_clip);
    }
    const ncclResult_t end = rccl().GroupEnd();
    if (bad != ncclSuccess) fail(NC_EDEVICE, "ncclAllGather failed inside the group: %s", rccl().GetErrorString(bad));
    if (end != ncclSuccess) fail(NC_EDEVICE, "ncclGroupEnd failed: %s", rccl().GetErrorString(end));
}
void unpack_slots(nc_group* g, nc_group::Member& x, int64_t* codes_all, int rows, int64_t per_clip) {
    if (g->code_bits <= 0) return;
    const nc_status st = nc_unpack_codes_dev(x.device, x.packed_all.as<uint8_t>(), rows * g->world, 1, per_clip, g->code_bits, codes_all, x.side);
    if (st != NC_OK) fail(st, "%s", get_last_error());
}

void rank_encode_allgather(nc_group* g, int kind, const float* pcm, int B, int64_t T, int sample_rate, int n_q, int64_t* codes_all, float* z,
                           float* lat) {
    if (!g || g->rank < 0) fail(NC_EINVAL, "group was not created with nc_group_create_rank");
    if (!pcm || !codes_all || B <= 0 || T <= 0) fail(NC_EINVAL, "bad arguments");
    nc_group::Member& x = g->m[0];
    const int64_t per_rank = encode_into_slot(x, kind, g->rank, pcm, B, T, sample_rate, n_q, codes_all, z, lat);
    gather_slots(g, x, g->rank, codes_all, B, per_rank / B);
    unpack_slots(g, x, codes_all, B, per_rank / B);
    NC_HIP(hipEventRecord(x.ev_gather, x.side));
}

// Local mode: ONE process drives all devices.  Clips are split into contiguous blocks (the first B_total % W devices hold one clip more:
// neuralcodecs_amd/parallel.py shard_bounds); every device's slot of the gathered buffer is sized for the largest block, so ragged
// batches gather with one collective and the padding rows are dropped on the way back to the host.  One host thread per device stages
// its block through pinned memory, uploads, and queues the encode; the grouped all-gather is issued by the calling thread.
void local_encode_allgather(nc_group* g, int kind, const float* pcm, int B_total, int64_t T, int sample_rate, int n_q, int64_t* codes, float* z) {
    DeviceRestore restore_device;
    if (!g || g->rank >= 0) fail(NC_EINVAL, "group was not created with nc_group_create_local");
    if (!pcm || !codes || B_total <= 0 || T <= 0) fail(NC_EINVAL, "bad arguments");
    const int W = g->world;
    for (int d = 0; d < W; ++d)   // before any cast below
        if (g->m[(size_t)d].h->kind != kind) fail(NC_EINVAL, kind == 0 ? "handle is not a DAC codec" : "handle is not a SNAC codec");
    const int q = B_total / W, r = B_total % W, B_max = q + (r ? 1 : 0);
    auto lo_of = [&](int d) { return d * q + std::min(d, r); };
    auto n_of = [&](int d) { return q + (d < r ? 1 : 0); };
    int64_t per_clip = 0, z_clip = 0;
    int channels = 1;
    {
        Codec& c0 = *g->m[0].h->impl;
        if (kind == 0) {
            DacModel& m = static_cast<DacModel&>(c0);
            const int nq = (n_q <= 0 || n_q > m.cfg.n_codebooks) ? m.cfg.n_codebooks : n_q;
            per_clip = (int64_t)nq * m.frames(T);
            z_clip = (int64_t)m.latent * m.frames(T);
        } else {
            SnacModel& m = static_cast<SnacModel&>(c0);
            per_clip = m.codes_per_clip(m.padded_len(T) / m.hop);
        }
    }
    const int64_t per_rank = (int64_t)B_max * per_clip;
    const bool want_z = z && kind == 0;
    std::vector<std::exception_ptr> errs((size_t)W);
    auto phase1 = [&](int d) {
        try {
            nc_group::Member& x = g->m[(size_t)d];
            Codec& c = *x.h->impl;
            c.use_device();
            const int B = n_of(d);
            if (g->peer_copy) NC_HIP(hipStreamSynchronize(x.side));   // (host mode: before a grow-only buffer may move)
            x.codes_all.reserve((size_t)per_rank * W * 8);
            if (B > 0) {
                const size_t in_bytes = (size_t)B * channels * T * 4;
                x.pcm.reserve(in_bytes);
                x.pin_in.reserve(in_bytes);
                std::memcpy(x.pin_in.p, pcm + (int64_t)lo_of(d) * channels * T, in_bytes);
                NC_HIP(hipMemcpyAsync(x.pcm.p, x.pin_in.p, in_bytes, hipMemcpyHostToDevice, c.stream));
                if (want_z) { x.z.reserve((size_t)B * z_clip * 4); x.pin_z.reserve((size_t)B * z_clip * 4); }
                // (the slot is sized for B_max clips; this device fills the first B of them)
                int64_t* slot = x.codes_all.as<int64_t>() + (int64_t)d * per_rank;
                if (kind == 0) static_cast<DacModel&>(c).encode_dev(x.pcm.as<float>(), B, T, sample_rate, n_q, slot, want_z ? x.z.as<float>() : nullptr, nullptr);
                else static_cast<SnacModel&>(c).encode_dev(x.pcm.as<float>(), B, T, slot, nullptr, nullptr, true);
                if (want_z) NC_HIP(hipMemcpyAsync(x.pin_z.p, x.z.p, (size_t)B * z_clip * 4, hipMemcpyDeviceToHost, c.stream));
            }
            NC_HIP(hipEventRecord(x.ev_enc, c.stream));
            NC_HIP(hipStreamWaitEvent(x.side, x.ev_enc, 0));
        } catch (...) {
            errs[(size_t)d] = std::current_exception();
        }
    };
    if (W == 1) {
        phase1(0);
    } else {
        std::vector<std::thread> th;
        for (int d = 0; d < W; ++d) th.emplace_back(phase1, d);
        for (auto& t : th) t.join();
    }
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
    if (g->code_bits > 0)   // (a device without clips still owns a slot: zero it so that its packed rows are defined)
        for (int d = 0; d < W; ++d)
            if (n_of(d) < B_max) {
                nc_group::Member& x = g->m[(size_t)d];
                (void)hipSetDevice(x.device);
                NC_HIP(hipMemsetAsync(x.codes_all.as<int64_t>() + (int64_t)d * per_rank + (int64_t)n_of(d) * per_clip, 0,
                                      (size_t)(B_max - n_of(d)) * per_clip * 8, x.side));
            }
    {
        std::vector<int64_t*> slots((size_t)W);
        for (int d = 0; d < W; ++d) slots[(size_t)d] = g->m[(size_t)d].codes_all.as<int64_t>();
        grouped_gather(g, slots, B_max, per_clip);
    }
    {   // only device 0's copy goes back to the host: it alone unpacks
        nc_group::Member& x = g->m[0];
        (void)hipSetDevice(x.device);
        unpack_slots(g, x, x.codes_all.as<int64_t>(), B_max, per_clip);
    }
    {   // every device now holds all codes; device 0's copy goes back to the host (pinned, then the valid rows of every slot)
        nc_group::Member& x = g->m[0];
        (void)hipSetDevice(x.device);
        x.pin_out.reserve((size_t)per_rank * W * 8);
        NC_HIP(hipMemcpyAsync(x.pin_out.p, x.codes_all.p, (size_t)per_rank * W * 8, hipMemcpyDeviceToHost, x.side));
    }
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        (void)hipSetDevice(x.device);
        NC_HIP(hipStreamSynchronize(x.side));
        NC_HIP(hipStreamSynchronize(x.h->impl->stream));
        x.h->impl->check_async_errors();
        if (want_z && n_of(d) > 0) std::memcpy(z + (int64_t)lo_of(d) * z_clip, x.pin_z.p, (size_t)n_of(d) * z_clip * 4);
    }
    const int64_t* all = static_cast<const int64_t*>(g->m[0].pin_out.p);
    for (int d = 0; d < W; ++d)
        if (n_of(d) > 0) std::memcpy(codes + (int64_t)lo_of(d) * per_clip, all + (int64_t)d * per_rank, (size_t)n_of(d) * per_clip * 8);
}

// Local mode with DEVICE-resident clips (the single-host layout of Examples/Program.cs:228-322 with the batch already split and uploaded):
// member d encodes its own block pcm[d] [B_local[d], 1, T] on its codec's stream with the codes written straight into slot d of ITS copy of
// the gathered tensor codes_all[d] [W * B_max, ...]; the grouped in-place all-gather then runs on the members' side streams, so whatever the
// caller queues next on a codec's stream (the local decode) overlaps it.  Nothing here waits on the host.
void local_encode_allgather_dev(nc_group* g, int kind, const float* const* pcm, const int32_t* B_local, int64_t T, int sample_rate, int n_q,
                                int64_t* const* codes_all, float* const* z, float* const* lat) {
    DeviceRestore restore_device;
    if (!g || g->rank >= 0) fail(NC_EINVAL, "group was not created with nc_group_create_local");
    if (!pcm || !B_local || !codes_all || T <= 0) fail(NC_EINVAL, "bad arguments");
    const int W = g->world;
    int B_max = 0;
    for (int d = 0; d < W; ++d) {
        if (g->m[(size_t)d].h->kind != kind) fail(NC_EINVAL, kind == 0 ? "handle is not a DAC codec" : "handle is not a SNAC codec");
        if (B_local[d] < 0 || !codes_all[d] || (B_local[d] > 0 && !pcm[d])) fail(NC_EINVAL, "bad block for device %d", d);
        B_max = std::max(B_max, (int)B_local[d]);
    }
    if (B_max <= 0) fail(NC_EINVAL, "no clips");
    int64_t per_clip = 0;
    {
        Codec& c0 = *g->m[0].h->impl;
        if (kind == 0) {
            DacModel& m = static_cast<DacModel&>(c0);
            const int nq = (n_q <= 0 || n_q > m.cfg.n_codebooks) ? m.cfg.n_codebooks : n_q;
            per_clip = (int64_t)nq * m.frames(T);
        } else {
            SnacModel& m = static_cast<SnacModel&>(c0);
            per_clip = m.codes_per_clip(m.padded_len(T) / m.hop);
        }
    }
    const int64_t per_rank = (int64_t)B_max * per_clip;
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        Codec& c = *x.h->impl;
        c.use_device();
        if (g->peer_copy) NC_HIP(hipStreamWaitEvent(c.stream, x.ev_gather, 0));   // (the previous gather's readers of this member's buffer)
        const int B = B_local[d];
        int64_t* slot = codes_all[d] + (int64_t)d * per_rank;
        if (B > 0) {
            if (kind == 0) static_cast<DacModel&>(c).encode_dev(pcm[d], B, T, sample_rate, n_q, slot, z ? z[d] : nullptr, lat ? lat[d] : nullptr);
            else static_cast<SnacModel&>(c).encode_dev(pcm[d], B, T, slot, nullptr, nullptr, true);
        }
        NC_HIP(hipEventRecord(x.ev_enc, c.stream));
        NC_HIP(hipStreamWaitEvent(x.side, x.ev_enc, 0));
        if (B < B_max)   // (the rest of a short block's slot is padding: defined values for the packed payload and for the caller)
            NC_HIP(hipMemsetAsync(slot + (int64_t)B * per_clip, 0, (size_t)(B_max - B) * per_clip * 8, x.side));
    }
    std::vector<int64_t*> slots(codes_all, codes_all + W);
    grouped_gather(g, slots, B_max, per_clip);
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        NC_HIP(hipSetDevice(x.device));
        unpack_slots(g, x, codes_all[d], B_max, per_clip);
        NC_HIP(hipEventRecord(x.ev_gather, x.side));
    }
}

// Encodec (Models/Encodec.cs:259-285): a rank's Encode emits, per segment, codes [B, n_q, T'_f] (laid end to end in segment order: the
// nc_encodec_encode_dev layout) and a scale [B] per segment.  Both gather rank by rank: codes_all = [world][per-rank code block], scales_all
// = [world][n_frames][B] -- two collectives on the side stream (the scales are 4 * n_frames * B bytes: nothing to pack).
int64_t encodec_codes_per_clip(EncodecModel& m, int64_t T, int* n_frames) {
    const auto segs = m.segments(T);
    int64_t fr = 0;
    for (auto& sg : segs) fr += sg.frames;
    if (n_frames) *n_frames = (int)segs.size();
    return (int64_t)m.n_q * fr;
}

void rank_encodec_allgather(nc_group* g, const float* pcm, int B, int64_t T, int64_t* codes_all, float* scales_all) {
    if (!g || g->rank < 0) fail(NC_EINVAL, "group was not created with nc_group_create_rank");
    if (!pcm || !codes_all || B <= 0 || T <= 0) fail(NC_EINVAL, "bad arguments");
    nc_group::Member& x = g->m[0];
    if (x.h->kind != 2) fail(NC_EINVAL, "handle is not an Encodec codec");
    EncodecModel& m = static_cast<EncodecModel&>(*x.h->impl);
    m.use_device();
    int nf = 0;
    const int64_t per_clip = encodec_codes_per_clip(m, T, &nf), per_rank = (int64_t)B * per_clip;
    if (m.cfg.normalize && !scales_all) fail(NC_EINVAL, "this model normalises frames: scales_all must be given");
    float* my_scales = scales_all ? scales_all + (int64_t)g->rank * nf * B : nullptr;
    m.encode_dev(pcm, B, T, codes_all + (int64_t)g->rank * per_rank, my_scales, nullptr);
    NC_HIP(hipEventRecord(x.ev_enc, m.stream));
    NC_HIP(hipStreamWaitEvent(x.side, x.ev_enc, 0));
    // (the per-rank block is segment-major, not one row per clip: it travels as B "rows" of per_clip values only for the packed payload's
    //  row arithmetic -- any split into equal rows packs the same value sequence)
    gather_slots(g, x, g->rank, codes_all, B, per_clip);
    unpack_slots(g, x, codes_all, B, per_clip);
    if (my_scales && m.cfg.normalize)
        NC_RCCL(rccl().AllGather(my_scales, scales_all, (size_t)nf * B, ncclFloat, x.comm, x.side));
    NC_HIP(hipEventRecord(x.ev_gather, x.side));
}

void local_encodec_allgather_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T, int64_t* const* codes_all,
                                 float* const* scales_all) {
    DeviceRestore restore_device;
    if (!g || g->rank >= 0) fail(NC_EINVAL, "group was not created with nc_group_create_local");
    if (!pcm || !B_local || !codes_all || T <= 0) fail(NC_EINVAL, "bad arguments");
    const int W = g->world;
    int B = 0;
    for (int d = 0; d < W; ++d) {
        if (g->m[(size_t)d].h->kind != 2) fail(NC_EINVAL, "handle is not an Encodec codec");
        if (!pcm[d] || !codes_all[d] || B_local[d] <= 0 || (d && B_local[d] != B_local[0])) fail(NC_EINVAL, "Encodec groups take equal, non-empty blocks (block %d)", d);
    }
    B = B_local[0];
    EncodecModel& m0 = static_cast<EncodecModel&>(*g->m[0].h->impl);
    int nf = 0;
    const int64_t per_clip = encodec_codes_per_clip(m0, T, &nf), per_rank = (int64_t)B * per_clip;
    const bool sc = m0.cfg.normalize;
    if (sc && !scales_all) fail(NC_EINVAL, "this model normalises frames: scales_all must be given");
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        EncodecModel& m = static_cast<EncodecModel&>(*x.h->impl);
        m.use_device();
        if (g->peer_copy) NC_HIP(hipStreamWaitEvent(m.stream, x.ev_gather, 0));
        m.encode_dev(pcm[d], B, T, codes_all[d] + (int64_t)d * per_rank, sc ? scales_all[d] + (int64_t)d * nf * B : nullptr, nullptr);
        NC_HIP(hipEventRecord(x.ev_enc, m.stream));
        NC_HIP(hipStreamWaitEvent(x.side, x.ev_enc, 0));
    }
    std::vector<int64_t*> slots(codes_all, codes_all + W);
    grouped_gather(g, slots, B, per_clip);
    if (sc && g->peer_copy) {
        std::vector<uint8_t*> bufs((size_t)W);
        for (int d = 0; d < W; ++d) bufs[(size_t)d] = reinterpret_cast<uint8_t*>(scales_all[d]);
        copy_allgather(g, bufs, (int64_t)nf * B * 4);
    } else if (sc) {
        NC_RCCL(rccl().GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int d = 0; d < W && bad == ncclSuccess; ++d) {
            nc_group::Member& x = g->m[(size_t)d];
            if (hipSetDevice(x.device) != hipSuccess) { bad = ncclUnhandledCudaError; break; }
            bad = rccl().AllGather(scales_all[d] + (int64_t)d * nf * B, scales_all[d], (size_t)nf * B, ncclFloat, x.comm, x.side);
        }
        const ncclResult_t end = rccl().GroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess) fail(NC_EDEVICE, "ncclAllGather of the frame scales failed: %s", rccl().GetErrorString(bad != ncclSuccess ? bad : end));
    }
    for (int d = 0; d < W; ++d) {
        nc_group::Member& x = g->m[(size_t)d];
        NC_HIP(hipSetDevice(x.device));
        unpack_slots(g, x, codes_all[d], B, per_clip);
        NC_HIP(hipEventRecord(x.ev_gather, x.side));
    }
}

}  // namespace

extern "C" {

nc_status nc_group_unique_id(void* uid) {
    return guard([&] {
        if (!uid) fail(NC_EINVAL, "uid must not be null");
        static_assert(NC_GROUP_UID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId id;
        NC_RCCL(rccl().GetUniqueId(&id));
        std::memcpy(uid, id.internal, NCCL_UNIQUE_ID_BYTES);
    });
}

nc_status nc_group_create_rank(int32_t world, int32_t rank, const void* uid, nc_codec* local, nc_group** out) {
    return guard([&] {
        if (!out || !uid) fail(NC_EINVAL, "uid and out must not be null");
        *out = nullptr;
        if (world <= 0 || rank < 0 || rank >= world) fail(NC_EINVAL, "rank %d outside a group of %d", rank, world);
        std::unique_ptr<nc_group> g(new nc_group());
        g->world = world; g->rank = rank;
        g->m.resize(1);
        init_member(g->m[0], local);
        ncclUniqueId id;
        std::memcpy(id.internal, uid, NCCL_UNIQUE_ID_BYTES);
        NC_RCCL(rccl().CommInitRank(&g->m[0].comm, world, id, rank));
        *out = g.release();
    });
}

namespace {
void create_local(int32_t ndev, nc_codec* const* handles, uint32_t flags, nc_group** out) {
    if (!out || !handles) fail(NC_EINVAL, "handles and out must not be null");
    *out = nullptr;
    if (ndev <= 0 || ndev > 64) fail(NC_EINVAL, "bad device count %d", ndev);
    if (flags & ~(uint32_t)NC_GROUP_PEER_COPY) fail(NC_EINVAL, "unknown group flags 0x%x", flags);
    DeviceRestore restore_device;
    std::unique_ptr<nc_group> g(new nc_group());
    g->world = ndev; g->rank = -1;
    g->peer_copy = (flags & NC_GROUP_PEER_COPY) != 0;
    g->m.resize((size_t)ndev);
    std::vector<int> devs((size_t)ndev);
    for (int d = 0; d < ndev; ++d) {
        if (!handles[d]) fail(NC_EINVAL, "null codec handle");
        for (int e = 0; e < d; ++e)
            if (handles[e] == handles[d]) fail(NC_EINVAL, "handle %d is handle %d again: one codec (one stream, one workspace) per member", d, e);
        init_member(g->m[(size_t)d], handles[d]);
        devs[(size_t)d] = handles[d]->impl->device;
        if (!g->peer_copy)   // (an RCCL communicator cannot hold two ranks of one GPU; the peer-copy transport can)
            for (int e = 0; e < d; ++e)
                if (devs[(size_t)e] == devs[(size_t)d]) fail(NC_EINVAL, "handles %d and %d live on the same device %d", e, d, devs[(size_t)d]);
        if (handles[d]->kind != handles[0]->kind) fail(NC_EINVAL, "the handles of a group must be of one codec kind");
    }
    if (g->peer_copy) {
        for (int d = 0; d < ndev; ++d)      // direct peer access where the hardware offers it (otherwise the runtime stages the copies)
            for (int e = 0; e < ndev; ++e) {
                int can = 0;
                if (devs[(size_t)d] == devs[(size_t)e] || hipDeviceCanAccessPeer(&can, devs[(size_t)d], devs[(size_t)e]) != hipSuccess || !can) continue;
                NC_HIP(hipSetDevice(devs[(size_t)d]));
                const hipError_t pe = hipDeviceEnablePeerAccess(devs[(size_t)e], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) NC_HIP(pe);
                (void)hipGetLastError();
            }
    } else {
        std::vector<ncclComm_t> comms((size_t)ndev);
        NC_RCCL(rccl().CommInitAll(comms.data(), ndev, devs.data()));
        for (int d = 0; d < ndev; ++d) g->m[(size_t)d].comm = comms[(size_t)d];
    }
    *out = g.release();
}
}  // namespace

nc_status nc_group_create_local(int32_t ndev, nc_codec* const* handles, nc_group** out) {
    return guard([&] { create_local(ndev, handles, 0, out); });
}

nc_status nc_group_create_local_ex(int32_t ndev, nc_codec* const* handles, uint32_t flags, nc_group** out) {
    return guard([&] { create_local(ndev, handles, flags, out); });
}

nc_status nc_group_destroy(nc_group* g) {
    return guard([&] { delete g; });
}

nc_status nc_group_set_code_bits(nc_group* g, int32_t bits) {
    return guard([&] {
        if (!g) fail(NC_EINVAL, "null group");
        if (bits < 0 || bits > 24) fail(NC_EINVAL, "Bits must be between 1 and 24 (0 = unpacked int64 payload)");   // BitPacker.cs MaxBits
        g->code_bits = bits;
    });
}

nc_status nc_group_info(const nc_group* g, int32_t* world, int32_t* rank) {
    return guard([&] {
        if (!g) fail(NC_EINVAL, "null group");
        if (world) *world = g->world;
        if (rank) *rank = g->rank;
    });
}

nc_status nc_group_dac_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int32_t sample_rate, int32_t n_q,
                                            int64_t* codes_all, float* z_local, float* latents_local) {
    return guard([&] { rank_encode_allgather(g, 0, pcm, B_local, T, sample_rate, n_q, codes_all, z_local, latents_local); });
}

nc_status nc_group_snac_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int64_t* codes_all) {
    return guard([&] { rank_encode_allgather(g, 1, pcm, B_local, T, 0, 0, codes_all, nullptr, nullptr); });
}

nc_status nc_group_wait(nc_group* g) {
    return guard([&] {
        if (!g || g->m.empty()) fail(NC_EINVAL, "null group");
        DeviceRestore restore_device;
        for (auto& x : g->m) {
            (void)hipSetDevice(x.device);
            NC_HIP(hipStreamWaitEvent(x.h->impl->stream, x.ev_gather, 0));
        }
    });
}

nc_status nc_group_dac_encode_allgather(nc_group* g, const float* pcm, int32_t B_total, int64_t T, int32_t sample_rate, int32_t n_q, int64_t* codes,
                                        float* z) {
    return guard([&] { local_encode_allgather(g, 0, pcm, B_total, T, sample_rate, n_q, codes, z); });
}

nc_status nc_group_snac_encode_allgather(nc_group* g, const float* pcm, int32_t B_total, int64_t T, int64_t* codes) {
    return guard([&] { local_encode_allgather(g, 1, pcm, B_total, T, 0, 0, codes, nullptr); });
}

nc_status nc_group_encodec_encode_allgather_dev(nc_group* g, const float* pcm, int32_t B_local, int64_t T, int64_t* codes_all, float* scales_all) {
    return guard([&] { rank_encodec_allgather(g, pcm, B_local, T, codes_all, scales_all); });
}

nc_status nc_group_encodec_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T, int64_t* const* codes_all,
                                                      float* const* scales_all) {
    return guard([&] { local_encodec_allgather_dev(g, pcm, B_local, T, codes_all, scales_all); });
}

nc_status nc_group_dac_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T, int32_t sample_rate,
                                                  int32_t n_q, int64_t* const* codes_all, float* const* z_local, float* const* latents_local) {
    return guard([&] { local_encode_allgather_dev(g, 0, pcm, B_local, T, sample_rate, n_q, codes_all, z_local, latents_local); });
}

nc_status nc_group_snac_encode_allgather_local_dev(nc_group* g, const float* const* pcm, const int32_t* B_local, int64_t T, int64_t* const* codes_all) {
    return guard([&] { local_encode_allgather_dev(g, 1, pcm, B_local, T, 0, 0, codes_all, nullptr, nullptr); });
}

}  // extern "C"
