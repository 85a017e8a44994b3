// strict_api/geometry_msgs/Twist.h — the message struct of polyTrajOctomap::updateInitVel / updateInitAcc (PO.h:82-85)
#include <trajectory_planner/standin/mini_ros.h>
