// Stand-in for the catkin-generated <mcl_3dl/MCL3DLParamsConfig.h> — TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_SHIM_MCL3DLPARAMSCONFIG_H
#define ORACLE_SHIM_MCL3DLPARAMSCONFIG_H
namespace mcl_3dl
{
struct MCL3DLParamsConfig
{
  double match_ratio_thresh;
  double match_output_dist;
  double unmatch_output_dist;
};
}  // namespace mcl_3dl
#endif
