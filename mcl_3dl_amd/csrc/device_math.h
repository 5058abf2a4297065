// device_math.h — float/double helpers shared by the gfx950 kernels.
//
// Every expression here is written in the reference's operation ORDER and compiled with
// -ffp-contract=off: the reference is built for baseline x86-64 (no FMA), so each a*b+c is two
// roundings.  Keeping the same order makes the transformed scan points, the squared distances and
// every DDA decision bit-identical to the CPU path; only the final reductions differ (fp64 tree
// instead of float sequential), see DESIGN.md "Numerics".
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace mcl3dl
{
struct Quat
{
  float x, y, z, w;
};

struct Vec3f
{
  float x, y, z;
};

__host__ __device__ inline Vec3f vadd(Vec3f a, Vec3f b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__host__ __device__ inline Vec3f vsub(Vec3f a, Vec3f b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__host__ __device__ inline Vec3f vscale(Vec3f a, float s) { return { a.x * s, a.y * s, a.z * s }; }
// Vec3::dot, include/mcl_3dl/vec3.h:141-144
__host__ __device__ inline float vdot(Vec3f a, Vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Quat::operator*(Quat), include/mcl_3dl/quat.h:131-138 (term order kept).
__host__ __device__ inline Quat qmul(Quat a, Quat q)
{
  Quat r;
  r.x = a.w * q.x + a.x * q.w + a.y * q.z - a.z * q.y;
  r.y = a.w * q.y + a.y * q.w + a.z * q.x - a.x * q.z;
  r.z = a.w * q.z + a.z * q.w + a.x * q.y - a.y * q.x;
  r.w = a.w * q.w - a.x * q.x - a.y * q.y - a.z * q.z;
  return r;
}

// Quat::operator*(Vec3), include/mcl_3dl/quat.h:139-143: (q (x) (v,0)) (x) conj(q) — two Hamilton products.
__host__ __device__ inline Vec3f qrot(Quat q, Vec3f v)
{
  const Quat qv = { v.x, v.y, v.z, 0.0f };
  const Quat c = { -q.x, -q.y, -q.z, q.w };
  const Quat r = qmul(qmul(q, qv), c);
  return { r.x, r.y, r.z };
}

// Quat::normalized, include/mcl_3dl/quat.h:175-178 via operator/(float) :148-151 = operator*(1.0 / s):
// the reciprocal is formed in double and narrowed to float.
__host__ __device__ inline Quat qnormalized(Quat q)
{
  const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  const float s = static_cast<float>(1.0 / static_cast<double>(n));
  return { q.x * s, q.y * s, q.z * s, q.w * s };
}

// 64-lane wavefront reductions (gfx950: wave64).
__device__ inline double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}
__device__ inline unsigned wave_sum(unsigned v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}
__device__ inline double wave_max(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    const double o = __shfl_down(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
}  // namespace mcl3dl
