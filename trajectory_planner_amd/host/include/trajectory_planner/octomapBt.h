/*
 * octomapBt.h — OctoMap binary tree (.bt) -> dense mapManager::occMap.
 *
 * The on-disk format either side of the corridor checker (SURVEY.md §8f "next" #4; the reference
 * receives the same tree over the /octomap_binary service, polyTrajOctomap.cpp:133-145).  Format
 * (octomap OcTree::readBinary, third-party, not vendored by the reference): text header lines
 * "id", "size <nodes>", "res <m>", "data", then a pre-order stream of 2 bytes per inner node,
 * 2 bits per child: 10b (bit0=1,bit1=0) free leaf, 01b occupied leaf, 11b inner node, 00b absent
 * (unknown); depth 16, key origin 32768.  Leaves above depth 16 (pruned) are expanded.
 */
#ifndef OCTOMAP_BT_H
#define OCTOMAP_BT_H
#include <trajectory_planner/compat.h>

#include <memory>
#include <string>

namespace trajPlanner {
struct BtInfo {
    long long nodes_header = 0;   // "size" line
    long long nodes_parsed = 0;   // nodes visited by the parser (must equal nodes_header)
    long long bytes_consumed = 0;
    double res = 0.0;
    int key_min[3] = {0, 0, 0};   // inclusive voxel-key bounds of all leaves (key - 32768)
    int key_max[3] = {0, 0, 0};
    long long occupied = 0, free_ = 0;
};

/* Dense map over the leaves' bounding box (+ `margin` voxels): bit2 = occupied, bit1 = unknown
 * (never observed), bit0 = occupied inflated by `inflate` metres per axis.  The map origin is a
 * multiple of res (octomap key lattice), as the corridor checker requires.  nullptr on a
 * malformed file. */
std::shared_ptr<mapManager::occMap> loadOctomapBt(const std::string& path, const double inflate[3], int margin,
                                                  BtInfo* info = nullptr);

/* ASCII .pcd point cloud (the reference's map/square_static_map.pcd, fed to map_manager's static-map
 * loader there) -> dense map at resolution `res`: a voxel holding a point is occupied, all others are
 * free and known; same inflation and lattice-aligned origin as above.  nullptr on a malformed file. */
std::shared_ptr<mapManager::occMap> loadPcdAscii(const std::string& path, double res, const double inflate[3], int margin,
                                                 long long* pointsRead = nullptr);
}  // namespace trajPlanner
#endif
