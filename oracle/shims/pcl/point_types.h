// Stand-in for <pcl/point_types.h> — TEST INFRASTRUCTURE ONLY (oracle build).
// Written for this repo; provides just what the reference's hot-path headers use so that the
// reference sources under /root/reference compile UNMODIFIED without PCL installed.
#ifndef ORACLE_SHIM_PCL_POINT_TYPES_H
#define ORACLE_SHIM_PCL_POINT_TYPES_H
#include <cstdint>
#include <cstddef>

#define EIGEN_ALIGN16 alignas(16)
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define PCL_ADD_POINT4D \
  union             \
  {                 \
    float data[4];  \
    struct          \
    {               \
      float x;      \
      float y;      \
      float z;      \
    };              \
  }
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, fields)

namespace pcl
{
struct EIGEN_ALIGN16 PointXYZ
{
  PCL_ADD_POINT4D;
  inline PointXYZ()
  {
    x = y = z = 0.0f;
    data[3] = 1.0f;
  }
  inline PointXYZ(float x_, float y_, float z_)
  {
    x = x_;
    y = y_;
    z = z_;
    data[3] = 1.0f;
  }
};
}  // namespace pcl
#endif
