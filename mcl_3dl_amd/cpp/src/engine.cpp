// The process-wide engine shared by the drop-in LiDAR models (one map, one GPU; MCL3DL_HIP_DEVICE selects the device).
#include <cstdlib>

#include <mcl_3dl_hip/engine.hpp>

namespace mcl_3dl
{
namespace hip
{
Engine& Engine::shared()
{
  static Engine engine([]
                       {
                         const char* env = std::getenv("MCL3DL_HIP_DEVICE");
                         return env ? std::atoi(env) : 0;
                       }());
  return engine;
}
}  // namespace hip
}  // namespace mcl_3dl
