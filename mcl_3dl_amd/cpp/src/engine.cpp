// The process-wide engine shared by the drop-in LiDAR models: one map, one device group.
//   MCL3DL_HIP_DEVICES=0,1,2,3   GPUs the particles are sharded over (default: MCL3DL_HIP_DEVICE, or device 0)
//   MCL3DL_HIP_COLLECTIVE=host   combine the per-device sums on the host instead of an RCCL all-reduce
//   MCL3DL_HIP_BATCH_SLICE=n      particles per slice of the batch behind pf::measure (option "batch_slice"; default: automatic)
//   MCL3DL_HIP_ENGINE_ORDER=1     the likelihood model's filter() returns its sampled cloud in the engine's scan order
//                                 (mcl3dl_hip_scan_order_host) and the likelihoods are the reference's float recurrence over that
//                                 cloud, computed inside the likelihood kernel (option "strict_order" = 3): bit-identical to
//                                 what the reference's own class returns for the cloud the node holds, at ~1.05x the default time
#include <cstdlib>
#include <string>
#include <vector>

#include <mcl_3dl_hip/engine.hpp>

namespace mcl_3dl
{
namespace hip
{
Engine& Engine::shared()
{
  static Engine engine([]
                       {
                         std::vector<int> devices;
                         if (const char* list = std::getenv("MCL3DL_HIP_DEVICES"))
                         {
                           const std::string s(list);
                           std::size_t pos = 0;
                           while (pos < s.size())
                           {
                             const std::size_t comma = s.find(',', pos);
                             const std::string item = s.substr(pos, comma == std::string::npos ? comma : comma - pos);
                             if (!item.empty())
                               devices.push_back(std::atoi(item.c_str()));
                             if (comma == std::string::npos)
                               break;
                             pos = comma + 1;
                           }
                         }
                         if (devices.empty())
                         {
                           const char* one = std::getenv("MCL3DL_HIP_DEVICE");
                           devices.push_back(one ? std::atoi(one) : 0);
                         }
                         return devices;
                       }());
  static const bool configured = []
  {
    if (const char* slice = std::getenv("MCL3DL_HIP_BATCH_SLICE"))
      engine.check(mcl3dl_hip_group_set_option(engine.group(), "batch_slice", std::atof(slice)));
    if (const char* eo = std::getenv("MCL3DL_HIP_ENGINE_ORDER"))
      if (std::atoi(eo) != 0)
      {
        engine.check(mcl3dl_hip_group_set_option(engine.group(), "strict_order", 3.0));
        engine.check(mcl3dl_hip_group_set_option(engine.group(), "scan_presorted", 1.0));  // filter() has ordered the cloud
        engine.engine_order = true;
      }
    return true;
  }();
  (void)configured;
  live() = &engine;
  return engine;
}
}  // namespace hip
}  // namespace mcl_3dl
