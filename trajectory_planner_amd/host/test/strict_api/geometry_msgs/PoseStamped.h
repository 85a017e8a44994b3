// strict_api/geometry_msgs/PoseStamped.h — ros::NodeHandle::getParam/setParam, ros::Time::now() and the message structs of the
// facades' signatures (public data members named like the real messages).
#include <trajectory_planner/standin/mini_ros.h>
