// Stand-in for <pcl_ros/point_cloud.h> — TEST INFRASTRUCTURE ONLY (oracle build).
#ifndef ORACLE_SHIM_PCL_ROS_POINT_CLOUD_H
#define ORACLE_SHIM_PCL_ROS_POINT_CLOUD_H
#include <pcl/point_cloud.h>
#endif
