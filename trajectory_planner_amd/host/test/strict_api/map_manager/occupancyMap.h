// strict_api/map_manager/occupancyMap.h — mapManager::occMap with ONLY what the reference calls on it
// (BT.h:197,199,312,319,332; BT.cpp:191-193,292,412,435,455,470,736,743,775,783,841,1434; AS.h:58): four methods,
// declared non-const (the stricter case for a caller), no data, no construction.  The facade sources compiled
// against this header (`make strict`) can therefore not depend on anything else of the map.
#ifndef STRICT_MAP_MANAGER_OCCUPANCY_MAP_H
#define STRICT_MAP_MANAGER_OCCUPANCY_MAP_H
#include <Eigen/Eigen>

namespace mapManager {
class occMap {
public:
    double getRes();
    bool isInflatedOccupied(const Eigen::Vector3d& pos);
    bool isInflatedOccupiedLine(const Eigen::Vector3d& pos1, const Eigen::Vector3d& pos2);
    bool isUnknown(const Eigen::Vector3d& pos);

private:
    occMap();
};
}  // namespace mapManager
#endif
