// Bit packing of code tensors on the device.
//
// Wire layout = the reference's BitPacker (Modules/Encodec/BitPacker.cs:66-90): value i occupies bits [i*bits, (i+1)*bits) of a
// little-endian bit stream (LSB first), the stream is flushed to a whole byte per frame (:48-62); values are written t outer,
// codebook inner (EncodecCompressor.cs:170-181).  Pure integer, HBM-bound work: one thread per output byte (pack) / per value
// (unpack), coalesced byte stores and int64 stores.
#include "nc_common.h"

namespace nc {

// value index v = t*K + k of clip b  <-  codes[b][k][t]
__global__ void pack_codes_kernel(const int64_t* __restrict__ codes, int B, int K, int64_t T, int bits, uint8_t* __restrict__ out,
                                  int64_t nbytes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * nbytes) return;
    const int64_t b = i / nbytes, byte = i - b * nbytes;
    const int64_t nval = (int64_t)K * T;
    const int64_t bit0 = byte * 8;
    int64_t v = bit0 / bits;                    // first value overlapping this byte
    unsigned acc = 0;
    for (; v < nval && v * bits < bit0 + 8; ++v) {
        const int64_t t = v / K;
        const int k = (int)(v - t * K);
        const uint64_t val = (uint64_t)codes[(b * K + k) * T + t] & ((1ull << bits) - 1);
        const int64_t sh = v * bits - bit0;     // position of the value's bit 0 relative to this byte
        acc |= sh >= 0 ? (unsigned)((val << sh) & 0xFF) : (unsigned)((val >> (-sh)) & 0xFF);
    }
    out[i] = (uint8_t)acc;
}

__global__ void unpack_codes_kernel(const uint8_t* __restrict__ in, int B, int K, int64_t T, int bits, int64_t* __restrict__ codes,
                                    int64_t nbytes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nval = (int64_t)K * T;
    if (i >= (int64_t)B * nval) return;
    const int64_t b = i / nval, r = i - b * nval;   // r = k*T + t in the output tensor
    const int k = (int)(r / T);
    const int64_t t = r - (int64_t)k * T;
    const int64_t v = t * K + k;
    const int64_t bit0 = v * bits;
    const uint8_t* p = in + b * nbytes;
    uint64_t acc = 0;
    int got = 0;
    for (int64_t byte = bit0 / 8; got < bits + 8 && byte < nbytes && byte * 8 < bit0 + bits; ++byte, got += 8)
        acc |= (uint64_t)p[byte] << got;
    codes[i] = (int64_t)((acc >> (bit0 & 7)) & ((1ull << bits) - 1));
}

}  // namespace nc

using namespace nc;

static void check_pack_args(const void* a, const void* b, int B, int K, int64_t T, int bits) {
    if (!a || !b) fail(NC_EINVAL, "null pointer");
    if (B <= 0 || K <= 0 || T <= 0) fail(NC_EINVAL, "B, K, T must be positive");
    if (bits <= 0 || bits > 24) fail(NC_EINVAL, "Bits must be between 1 and 24");   // BitPacker.cs MaxBits
}

extern "C" {

int64_t nc_packed_bytes(int64_t n_values, int32_t bits) { return (n_values * bits + 7) / 8; }

static nc_status pack_impl(int device_index, const int64_t* codes, int B, int K, int64_t T, int bits, uint8_t* packed, hipStream_t s) {
    try {
        check_pack_args(codes, packed, B, K, T, bits);
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
        if (device_index < 0 || device_index >= n) fail(NC_EINVAL, "device index out of range");
        NC_HIP(hipSetDevice(device_index));
        const int64_t nbytes = nc_packed_bytes((int64_t)K * T, bits), tot = (int64_t)B * nbytes;
        hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, codes, B, K, T, bits, packed, nbytes);
        NC_HIP(hipGetLastError());
        return NC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}
static nc_status unpack_impl(int device_index, const uint8_t* packed, int B, int K, int64_t T, int bits, int64_t* codes, hipStream_t s) {
    try {
        check_pack_args(codes, packed, B, K, T, bits);
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
        if (device_index < 0 || device_index >= n) fail(NC_EINVAL, "device index out of range");
        NC_HIP(hipSetDevice(device_index));
        const int64_t nbytes = nc_packed_bytes((int64_t)K * T, bits), tot = (int64_t)B * K * T;
        hipLaunchKernelGGL(unpack_codes_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, packed, B, K, T, bits, codes, nbytes);
        NC_HIP(hipGetLastError());
        return NC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}

nc_status nc_pack_codes_dev(int device_index, const int64_t* codes, int32_t B, int32_t K, int64_t T, int32_t bits, uint8_t* packed,
                            void* hip_stream) {
    return pack_impl(device_index, codes, B, K, T, bits, packed, static_cast<hipStream_t>(hip_stream));
}
nc_status nc_unpack_codes_dev(int device_index, const uint8_t* packed, int32_t B, int32_t K, int64_t T, int32_t bits, int64_t* codes,
                              void* hip_stream) {
    return unpack_impl(device_index, packed, B, K, T, bits, codes, static_cast<hipStream_t>(hip_stream));
}

nc_status nc_pack_codes(int device_index, const int64_t* codes, int32_t B, int32_t K, int64_t T, int32_t bits, uint8_t* packed) {
    try {
        check_pack_args(codes, packed, B, K, T, bits);
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
        NC_HIP(hipSetDevice(device_index));
        DevBuf dc, dp;
        const size_t nc_ = (size_t)B * K * T * 8, np = (size_t)B * nc_packed_bytes((int64_t)K * T, bits);
        dc.reserve(nc_); dp.reserve(np);
        NC_HIP(hipMemcpy(dc.p, codes, nc_, hipMemcpyHostToDevice));
        nc_status st = pack_impl(device_index, dc.as<int64_t>(), B, K, T, bits, dp.as<uint8_t>(), nullptr);
        if (st == NC_OK) {
            NC_HIP(hipDeviceSynchronize());
            NC_HIP(hipMemcpy(packed, dp.p, np, hipMemcpyDeviceToHost));
        }
        dc.release(); dp.release();
        return st;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}
nc_status nc_unpack_codes(int device_index, const uint8_t* packed, int32_t B, int32_t K, int64_t T, int32_t bits, int64_t* codes) {
    try {
        check_pack_args(codes, packed, B, K, T, bits);
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(NC_EDEVICE, "no HIP device available (the engine has no CPU fallback)");
        NC_HIP(hipSetDevice(device_index));
        DevBuf dc, dp;
        const size_t nc_ = (size_t)B * K * T * 8, np = (size_t)B * nc_packed_bytes((int64_t)K * T, bits);
        dc.reserve(nc_); dp.reserve(np);
        NC_HIP(hipMemcpy(dp.p, packed, np, hipMemcpyHostToDevice));
        nc_status st = unpack_impl(device_index, dp.as<uint8_t>(), B, K, T, bits, dc.as<int64_t>(), nullptr);
        if (st == NC_OK) {
            NC_HIP(hipDeviceSynchronize());
            NC_HIP(hipMemcpy(codes, dc.p, nc_, hipMemcpyDeviceToHost));
        }
        dc.release(); dp.release();
        return st;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    }
}

}  // extern "C"
