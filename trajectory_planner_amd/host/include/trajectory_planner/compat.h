/*
 * compat.h — the Eigen / ROS message / map_manager types the planner facades' signatures name.
 *
 * -DVIGO_WITH_ROS (a catkin workspace): the real headers.  The facade sources touch a map only through
 * getRes / isInflatedOccupied / isInflatedOccupiedLine / isUnknown and the one adapter in mapAdapter.{h,cpp}, and use
 * only real-Eigen members; `make strict` (host/Makefile) proves it by compiling them against host/test/strict_api/,
 * a header set that declares nothing beyond that surface.  What still differs from the reference's classes is
 * listed in INTEGRATION.md §3.
 *
 * Otherwise (this image, the GPU box: neither Eigen nor ROS): the stand-ins of standin/ — same member names, only
 * what the facades touch; the map is a dense byte grid.  They are NOT used to build any reference source.
 */
#ifndef TRAJECTORY_PLANNER_COMPAT_H
#define TRAJECTORY_PLANNER_COMPAT_H

#ifdef VIGO_WITH_ROS
#include <ros/ros.h>
#include <Eigen/Eigen>
#include <nav_msgs/Path.h>
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/Twist.h>
#include <map_manager/occupancyMap.h>
#else
#include <trajectory_planner/standin/mini_eigen.h>
#include <trajectory_planner/standin/mini_ros.h>
#include <trajectory_planner/standin/dense_occmap.h>
#endif
#endif
