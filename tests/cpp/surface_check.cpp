// tests/cpp/surface_check.cpp — TEST INFRASTRUCTURE: the parts of the LidarMeasurementModelBase plugin surface that are
// not measure(): getMaxSearchRange, refreshParameters, setGlobalLocalizationStatus, filter,
// getSinTotalRef, getFilterLabelMax, and getBeamStatus with its CastResult (status, pos_ = the collided voxel's centre,
// point_ = the collided map point: what the node's collision markers are drawn from, src/mcl_3dl.cpp:471-500).
// ONE source, built TWICE:
//   tests/cpp/surface_gpu.bin     against the drop-in headers of mcl_3dl_amd/cpp/include (classes backed by the HIP engine)
//   oracle/_ref/surface_ref.bin   against the reference's own headers and sources (oracle/Makefile, target `ref`)
// Both write the same record stream; tests/test_gpu_adapter.py demands byte equality (and keeps the reference's output
// as tests/golden/surface_ref.dat for boxes without oracle/_ref).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <random>
#include <vector>

#include <mcl_3dl/chunked_kdtree.h>
#include <mcl_3dl/lidar_measurement_model_base.h>
#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_beam.h>
#include <mcl_3dl/lidar_measurement_models/lidar_measurement_model_likelihood.h>
#include <mcl_3dl/point_cloud_random_sampler.h>
#include <mcl_3dl/point_types.h>

using PointType = mcl_3dl::LidarMeasurementModelBase::PointType;
using Cloud = pcl::PointCloud<PointType>;

// deterministic stand-in for PointCloudUniformSampler (which seeds itself from std::random_device): same contract —
// empty input gives an empty cloud, otherwise exactly `num` points drawn with replacement
class StrideSampler : public mcl_3dl::PointCloudRandomSampler<PointType>
{
public:
  Cloud::Ptr sample(const Cloud::ConstPtr& pc, const size_t num) const final
  {
    Cloud::Ptr out(new Cloud);
    out->header = pc->header;
    if (pc->points.size() == 0)
      return out;
    for (size_t i = 0; i < num; ++i)
      out->push_back(pc->points[(i * 7919 + 13) % pc->points.size()]);
    return out;
  }
};

static FILE* g_out;
static void put(const void* p, size_t n)
{
  fwrite(p, 1, n, g_out);
}
static void putCloud(const Cloud::Ptr& pc)
{
  const uint64_t n = pc->points.size();
  put(&n, 8);
  for (const auto& p : pc->points)
  {
    const float f[3] = { p.x, p.y, p.z };
    const uint32_t l = p.label;
    put(f, 12);
    put(&l, 4);
  }
}

int main(int argc, char** argv)
{
  if (argc != 2)
    return 2;
  g_out = fopen(argv[1], "wb");
  if (!g_out)
    return 2;
  // a raw scan that straddles every clip boundary: ranges 0..14 m, heights -3..3 m
  std::mt19937 rng(20240917);
  std::uniform_real_distribution<float> ux(-10.f, 10.f), uz(-3.f, 3.f);
  Cloud::Ptr raw(new Cloud);
  for (int i = 0; i < 5000; ++i)
  {
    PointType p;
    p.x = ux(rng);
    p.y = ux(rng);
    p.z = uz(rng);
    p.label = static_cast<uint32_t>(i % 5);
    raw->push_back(p);
  }
  // points exactly on the boundaries (the predicates are strict: likelihood.cpp:84-93, beam.cpp:103-112)
  const float edge[][3] = { { 0.5f, 0.f, 0.f }, { 10.f, 0.f, 0.f }, { 4.f, 0.f, 0.f },  { 0.f, 0.3f, 0.4f },
                            { 1.f, 1.f, 2.f },  { 1.f, 1.f, -2.f }, { 6.f, 8.f, 1.9f }, { 2.4f, 3.2f, 0.f } };
  for (const auto& e : edge)
  {
    PointType p;
    p.x = e[0];
    p.y = e[1];
    p.z = e[2];
    p.label = 9;
    raw->push_back(p);
  }
  Cloud::Ptr empty(new Cloud);

  auto lik_params = std::make_shared<mcl_3dl::LidarMeasurementModelLikelihoodParameters>();
  auto beam_params = std::make_shared<mcl_3dl::LidarMeasurementModelBeamParameters>();
  beam_params->use_raycast_using_dda_ = true;
  mcl_3dl::LidarMeasurementModelBase::Ptr models[2] = {
    mcl_3dl::LidarMeasurementModelBase::Ptr(new mcl_3dl::LidarMeasurementModelLikelihood(lik_params)),
    mcl_3dl::LidarMeasurementModelBase::Ptr(new mcl_3dl::LidarMeasurementModelBeam(beam_params)),
  };
  const StrideSampler sampler;
  // (num_particles, current_num_particles) as MCL3dlNode passes them (src/mcl_3dl.cpp:378-381): tracking, global
  // localisation at several stages, and the degenerate ratios
  const size_t status[][2] = { { 64, 64 }, { 64, 6400 }, { 64, 640 }, { 64, 65 }, { 100, 33 }, { 1, 1000000 }, { 500, 501 } };

  for (int round = 0; round < 3; ++round)
  {
    if (round == 1)
    {
      // parameters mutated behind the models' backs + refreshParameters (the dynamic_reconfigure path)
      lik_params->num_points_default_ = 300;
      lik_params->num_points_global_ = 17;
      lik_params->clip_near_ = 1.5;
      lik_params->clip_far_ = 6.0;
      lik_params->clip_z_min_ = -0.5;
      lik_params->clip_z_max_ = 1.0;
      lik_params->match_dist_min_ = 0.45;
      beam_params->num_points_default_ = 11;
      beam_params->num_points_global_ = 2;
      beam_params->clip_near_ = 0.1;
      beam_params->clip_far_ = 9.0;
      beam_params->clip_z_min_ = -2.5;
      beam_params->clip_z_max_ = 0.2;
      beam_params->map_grid_x_ = 0.05;
      beam_params->map_grid_y_ = 0.2;
      beam_params->map_grid_z_ = 0.15;
      beam_params->ang_total_ref_ = 0.3;
      beam_params->filter_label_max_ = 3;
      for (auto& m : models)
        m->refreshParameters();
    }
    if (round == 2)
    {
      lik_params->num_points_global_ = 0;  // global localisation with no points: filter() returns an empty cloud
      beam_params->num_points_default_ = 1;
      for (auto& m : models)
        m->refreshParameters();
    }
    for (auto& m : models)
    {
      const float r = m->getMaxSearchRange();
      put(&r, 4);
      for (const auto& st : status)
      {
        m->setGlobalLocalizationStatus(st[0], st[1]);
        putCloud(m->filter(raw, sampler));
        putCloud(m->filter(empty, sampler));
      }
    }
    auto beam = std::dynamic_pointer_cast<mcl_3dl::LidarMeasurementModelBeam>(models[1]);
    const float s = beam->getSinTotalRef();
    const uint32_t fl = beam->getFilterLabelMax();
    put(&s, 4);
    put(&fl, 4);
  }
  // ---- getBeamStatus: rays from inside a box of map points (a wall at x = 2 with a hole, a labelled wall at y = -1.5, a
  // floor), towards the walls, through the hole, short of the wall, and from outside the map
  {
    auto bp = std::make_shared<mcl_3dl::LidarMeasurementModelBeamParameters>();
    bp->use_raycast_using_dda_ = true;
    bp->filter_label_max_ = 1;
    mcl_3dl::LidarMeasurementModelBeam beam(bp);
    Cloud::Ptr map(new Cloud);
    const auto add = [&](float x, float y, float z, uint32_t label)
    {
      PointType p;
      p.x = x;
      p.y = y;
      p.z = z;
      p.label = label;
      map->push_back(p);
    };
    for (int i = 0; i < 40; ++i)
      for (int j = 0; j < 30; ++j)
      {
        const float u = -2.f + 0.1f * i + 0.013f * ((i * 7 + j * 3) % 5), v = -0.5f + 0.1f * j + 0.011f * ((i + j * 5) % 7);
        if (!(i > 17 && i < 23 && j > 12 && j < 18))
          add(2.0f + 0.004f * ((i + j) % 3), u, v, 0);  // wall with a hole
        add(u, -1.5f, v, 3);                               // wall whose label is above filter_label_max
        add(u, -2.2f + 0.1f * j, -0.5f, 0);                // floor
      }
    map->header.stamp = 7;
    mcl_3dl::ChunkedKdtree<PointType>::Ptr kdtree(new mcl_3dl::ChunkedKdtree<PointType>(20.0, beam.getMaxSearchRange()));
    kdtree->setInputCloud(map);
    std::mt19937 rr(99);
    std::uniform_real_distribution<float> ang(-1.2f, 1.2f), el(-0.5f, 0.4f), len(0.3f, 4.5f);
    for (int k = 0; k < 400; ++k)
    {
      const mcl_3dl::Vec3 from = (k % 50 == 49) ? mcl_3dl::Vec3(9.f, 9.f, 9.f) : mcl_3dl::Vec3(0.1f * (k % 7), -0.2f, 0.3f);
      const float a = ang(rr), e = el(rr), l = len(rr);
      const mcl_3dl::Vec3 to = from + mcl_3dl::Vec3(l * std::cos(e) * std::cos(a), l * std::cos(e) * std::sin(a), l * std::sin(e));
      mcl_3dl::Raycast<PointType>::CastResult cr;
      const int32_t st = static_cast<int32_t>(beam.getBeamStatus(kdtree, from, to, cr));
      put(&st, 4);
      if (st != static_cast<int32_t>(mcl_3dl::LidarMeasurementModelBeam::BeamStatus::LONG))
      {
        const float pos[3] = { cr.pos_.x_, cr.pos_.y_, cr.pos_.z_ };
        const int64_t index = cr.point_ ? static_cast<int64_t>(cr.point_ - &map->points[0]) : -1;
        put(pos, 12);
        put(&index, 8);
      }
    }
  }
  fclose(g_out);
  return 0;
}
