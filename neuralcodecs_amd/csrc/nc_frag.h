// Matrix-core fragment helpers shared by the convolution kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace nc {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// A fragments of one lane for one kk row (layout a_tile_pos).  lane_p = start of the kk row + nc_a_lane_off<TM>(r), r = lane & 31;
// every address is lane_p + a compile-time constant, so the reads take immediate offsets.
template <int TM>
__device__ __forceinline__ int nc_a_lane_off(int r) { return TM == 1 ? r : (TM == 4 || TM == 8) ? 4 * r : 2 * r; }

template <int TM>
__device__ __forceinline__ void nc_load_a_frag(const float* lane_p, int r, float (&a)[TM]) {
    if constexpr (TM == 1) {
        a[0] = lane_p[0];
    } else if constexpr (TM == 2) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(lane_p);
        a[0] = v[0]; a[1] = v[1];
    } else if constexpr (TM == 3) {
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(lane_p);
        a[0] = v[0]; a[1] = v[1]; a[2] = (lane_p - r)[64];   // row + 64 + r
    } else if constexpr (TM == 6) {   // two TM = 3 images, 96 floats apart
        const f32x2_t v = *reinterpret_cast<const f32x2_t*>(lane_p), w = *reinterpret_cast<const f32x2_t*>(lane_p + 96);
        a[0] = v[0]; a[1] = v[1]; a[2] = (lane_p - r)[64];
        a[3] = w[0]; a[4] = w[1]; a[5] = (lane_p - r)[96 + 64];
    } else if constexpr (TM == 8) {   // two TM = 4 images, 128 floats apart
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(lane_p), w = *reinterpret_cast<const f32x4_t*>(lane_p + 128);
        a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
        a[4] = w[0]; a[5] = w[1]; a[6] = w[2]; a[7] = w[3];
    } else {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(lane_p);
        a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
    }
}

}  // namespace nc
