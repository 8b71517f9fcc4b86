// device_math.hpp — device-side f32 helpers of the render path (included by kernels.hip only).
//
// Numerics contract: the translation unit is compiled with -ffp-contract=off and
// -fhip-fp32-correctly-rounded-divide-sqrt; everything here is IEEE f32 in the reference's
// operation order (raytracer_lib/src/vecmath.rs), so values are bit-comparable with unfused CPU code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355rt {

struct f3 { float x, y, z; };

// IEEE correctly rounded divide / sqrt: plain `/` and sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt.  (HIP's __fdiv_rn/__fsqrt_rn are NOT: __fsqrt_rn is the
// native approximation.)  tests/test_gpu_parity.py::test_device_arithmetic_is_ieee checks both.
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 vscale(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }    // Vec3 * f32
__device__ __forceinline__ f3 sscale(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }    // f32 * Vec3
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vecmath.rs:74-76
__device__ __forceinline__ f3 cross3(f3 a, f3 b)                                                  // vecmath.rs:79-85
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalized3(f3 a)                                                   // vecmath.rs:23-26
{
    const float len = sqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk3(div_rn(a.x, len), div_rn(a.y, len), div_rn(a.z, len));
}

// counter RNG (replaces the reference's OS-entropy StdRng): pcg4d, Jarzynski & Olano, JCGT 2020
__device__ __forceinline__ void pcg4d(uint32_t& x, uint32_t& y, uint32_t& z, uint32_t& w)
{
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
}
// rand 0.9.1 UniformFloat<f32> for 0.0..1.0: 23 random mantissa bits
__device__ __forceinline__ float u01(uint32_t bits) { return (float)(bits >> 9) * (1.0f / 8388608.0f); }

// x.powf(32.0), raytracer/mod.rs:255: five squarings in f64, rounded once
__device__ __forceinline__ float pow32(float x)
{
    double d = (double)x;
    d = d * d; d = d * d; d = d * d; d = d * d; d = d * d;
    return (float)d;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint32_t bcast_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

}  // namespace mi355rt
