// Stand-in for <pcl/kdtree/kdtree.h> — TEST INFRASTRUCTURE ONLY (oracle build).
// pcl::PointRepresentation as the reference uses it (src/mcl_3dl.cpp:109-126,1270 and
// include/mcl_3dl/chunked_kdtree.h:108-118): copyToFloatArray() then per-dimension rescale (alpha).
#ifndef ORACLE_SHIM_PCL_KDTREE_KDTREE_H
#define ORACLE_SHIM_PCL_KDTREE_KDTREE_H
#include <memory>
#include <vector>
#include <pcl/point_cloud.h>

namespace pcl
{
template <typename PointT>
class PointRepresentation
{
protected:
  int nr_dimensions_ = 0;
  std::vector<float> alpha_;
  bool trivial_ = false;

public:
  using Ptr = std::shared_ptr<PointRepresentation<PointT>>;
  using ConstPtr = std::shared_ptr<const PointRepresentation<PointT>>;
  virtual ~PointRepresentation() = default;
  virtual void copyToFloatArray(const PointT& p, float* out) const = 0;
  inline int getNumberOfDimensions() const { return nr_dimensions_; }
  void setRescaleValues(const float* rescale_array)
  {
    alpha_.resize(nr_dimensions_);
    for (int i = 0; i < nr_dimensions_; ++i)
      alpha_[i] = rescale_array[i];
  }
  // out[i] = in[i] * alpha[i]  (one float rounding per coordinate), as PCL's vectorize().
  template <typename OutputType>
  void vectorize(const PointT& p, OutputType& out) const
  {
    float temp[16];
    copyToFloatArray(p, temp);
    if (alpha_.empty())
    {
      for (int i = 0; i < nr_dimensions_; ++i)
        out[i] = temp[i];
    }
    else
    {
      for (int i = 0; i < nr_dimensions_; ++i)
        out[i] = temp[i] * alpha_[i];
    }
  }
};

template <typename PointT>
class DefaultPointRepresentation : public PointRepresentation<PointT>
{
  using PointRepresentation<PointT>::nr_dimensions_;
  using PointRepresentation<PointT>::trivial_;

public:
  DefaultPointRepresentation()
  {
    nr_dimensions_ = 3;
    trivial_ = true;
  }
  void copyToFloatArray(const PointT& p, float* out) const override
  {
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
  }
};
}  // namespace pcl
#endif
