// Stand-in for <ros/ros.h> — TEST INFRASTRUCTURE ONLY (oracle build).
#ifndef ORACLE_SHIM_ROS_ROS_H
#define ORACLE_SHIM_ROS_ROS_H
#include <array>
#include <cstdio>
#define ROS_ERROR(...) do { std::fprintf(stderr, "[ROS_ERROR] " __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { } while (0)
#define ROS_INFO(...) do { } while (0)
#define ROS_DEBUG(...) do { } while (0)
namespace ros
{
class NodeHandle
{
};
class Duration
{
public:
  double sec_ = 0;
};
class Time
{
};
}  // namespace ros
#endif
