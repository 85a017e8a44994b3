/*
 * mapAdapter.h — the ONE place where the planner facades go beyond the four map methods the reference calls
 * (getRes, isInflatedOccupied, isInflatedOccupiedLine, isUnknown).
 *
 * The device works on a dense snapshot of the voxel map (include/vigo.h, "voxel map").  An arbitrary
 * mapManager::occMap offers no bulk access, so the adapter RASTERISES it: every voxel centre of a caller-given box
 * is asked isInflatedOccupied / isUnknown and the answers become the byte grid vigo_set_grid_host() uploads
 * (bit0 inflated-occupied, bit1 unknown, bit2 := bit0 — the un-inflated occupancy is not observable through the four
 * methods; only polyTrajOctomap's octree-style checks read bit2).  The box comes from setMapRegion() of the planner
 * classes; a live map is re-rasterised when the owner calls refreshMap() (the reference's planners see map updates
 * through the shared pointer, a device snapshot cannot).
 *
 * Builds without map_manager (this tree's dense stand-in, standin/dense_occmap.h) take a fast path: the stand-in's
 * own bytes are uploaded as they are and its `version` counter tells when they changed — no region, no refresh call.
 */
#ifndef TRAJECTORY_PLANNER_MAP_ADAPTER_H
#define TRAJECTORY_PLANNER_MAP_ADAPTER_H
#include <trajectory_planner/compat.h>

#include <cstdint>
#include <memory>

struct vigo_context;

namespace trajPlanner {

struct mapRegion {
    bool set = false;
    Eigen::Vector3d boxMin, boxMax;   // metric box the planner works in
};
// two planners may share one device snapshot only when they would rasterise the same box
inline bool sameRegion(const mapRegion& a, const mapRegion& b) {
    if (a.set != b.set) return false;
    if (!a.set) return true;
    for (int k = 0; k < 3; ++k)
        if (a.boxMin(k) != b.boxMin(k) || a.boxMax(k) != b.boxMax(k)) return false;
    return true;
}

class mapAdapter {
public:
    /* Make `dev` hold a current snapshot of `map`.  `stamp` is the caller's memo of what it last uploaded (0 = nothing
     * yet / refresh requested); it is updated.  false: no region for a map that needs one, or a device failure. */
    static bool uploadSnapshot(vigo_context* dev, const std::shared_ptr<mapManager::occMap>& map, const mapRegion& region,
                               uint64_t& stamp);
    /* A live map (one the adapter can only rasterise) changed: EVERY handle that snapshotted it is stale, whichever
     * planner's refreshMap() / updateMap() said so — the generation is per map object, process-wide, and a planner's
     * `stamp` records the generation its own handle holds (so a batch whose lead never asked for the refresh itself
     * still re-rasterises).  Generations start at 1; a stamp of 0 never matches. */
    static uint64_t generation(const mapManager::occMap* map);
    static void bumpGeneration(const mapManager::occMap* map);
    /* octomap-style node lookup for polyTrajOctomap::checkCollisionPoint (PO.cpp:571-589): kOutside beyond the metric
     * bounds, else bit1 = no node (unknown), bit2 = occupied */
    enum : unsigned { kUnknown = 2u, kOccupied = 4u, kOutside = 0x80u };
    static unsigned nodeBits(const std::shared_ptr<mapManager::occMap>& map, const mapRegion& region, float x, float y, float z);
    /* rasterisation of any occMap through its four public methods over `region` (what uploadSnapshot uploads for a
     * map it cannot read in bulk); dims and origin of the grid are returned */
    static bool rasterise(mapManager::occMap& map, const mapRegion& region, std::vector<uint8_t>& voxels, int dims[3], double origin[3]);
};

}  // namespace trajPlanner
#endif
