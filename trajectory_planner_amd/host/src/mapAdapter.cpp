// mapAdapter.cpp — see mapAdapter.h.  The generic path uses getRes / isInflatedOccupied / isUnknown only; the dense
// fast path exists only in builds with this tree's stand-in map (no VIGO_WITH_ROS).
#include <trajectory_planner/mapAdapter.h>

#include <cmath>
#include <iostream>
#include <map>
#include <mutex>
#include <vector>

#include "../../../include/vigo.h"

namespace trajPlanner {

bool mapAdapter::rasterise(mapManager::occMap& map, const mapRegion& region, std::vector<uint8_t>& voxels, int dims[3], double origin[3]) {
    if (!region.set) return false;
    const double res = map.getRes();
    if (!(res > 0)) return false;
    for (int a = 0; a < 3; ++a) {
        // the grid is aligned to multiples of res (the corridor checker's octomap keys need that)
        const double lo = std::floor(region.boxMin(a) / res), hi = std::ceil(region.boxMax(a) / res);
        if (!(hi > lo) || hi - lo > 4096) return false;
        origin[a] = lo * res;
        dims[a] = (int)(hi - lo);
    }
    voxels.assign((size_t)dims[0] * dims[1] * dims[2], 0);
    for (int ix = 0; ix < dims[0]; ++ix)
        for (int iy = 0; iy < dims[1]; ++iy)
            for (int iz = 0; iz < dims[2]; ++iz) {
                const Eigen::Vector3d c(origin[0] + (ix + 0.5) * res, origin[1] + (iy + 0.5) * res, origin[2] + (iz + 0.5) * res);
                uint8_t v = 0;
                if (map.isInflatedOccupied(c)) v |= 1u | 4u;
                if (map.isUnknown(c)) v |= 2u;
                voxels[((size_t)ix * dims[1] + iy) * dims[2] + iz] = v;
            }
    return true;
}

namespace {
std::mutex g_genMutex;
std::map<const mapManager::occMap*, uint64_t> g_generation;
}  // namespace

uint64_t mapAdapter::generation(const mapManager::occMap* map) {
    std::lock_guard<std::mutex> lk(g_genMutex);
    auto it = g_generation.find(map);
    return it == g_generation.end() ? 1 : it->second;
}

void mapAdapter::bumpGeneration(const mapManager::occMap* map) {
    if (!map) return;
    std::lock_guard<std::mutex> lk(g_genMutex);
    auto it = g_generation.find(map);
    if (it == g_generation.end()) g_generation[map] = 2;
    else ++it->second;
}

bool mapAdapter::uploadSnapshot(vigo_context* dev, const std::shared_ptr<mapManager::occMap>& map, const mapRegion& region,
                                uint64_t& stamp) {
    if (!dev || !map) return false;
#ifndef VIGO_WITH_ROS
    (void)region;
    if (stamp == map->version) return true;
    const double o[3] = {map->origin()(0), map->origin()(1), map->origin()(2)};
    if (vigo_set_grid_host(dev, map->nx(), map->ny(), map->nz(), o, map->getRes(), map->voxels().data()) != VIGO_OK) return false;
    stamp = map->version;
    return true;
#else
    const uint64_t gen = generation(map.get());
    if (stamp == gen) return true;        // current until an owner of this map asks for a refresh
    std::vector<uint8_t> vox;
    int dims[3];
    double origin[3];
    if (!rasterise(*map, region, vox, dims, origin)) {
        std::cout << "[mapAdapter]: setMapRegion() must give the box to snapshot before planning on this map type." << std::endl;
        return false;
    }
    if (vigo_set_grid_host(dev, dims[0], dims[1], dims[2], origin, map->getRes(), vox.data()) != VIGO_OK) return false;
    stamp = gen;
    return true;
#endif
}

unsigned mapAdapter::nodeBits(const std::shared_ptr<mapManager::occMap>& map, const mapRegion& region, float x, float y, float z) {
    if (!map) return kOutside;
#ifndef VIGO_WITH_ROS
    (void)region;
    const double res = map->getRes();
    const Eigen::Vector3d o = map->origin();
    // octomap getMetricMin/Max (PO.cpp:572-577), then coordToKey: floor(coord * resolution_factor)
    if (x < o(0) || x > o(0) + map->nx() * res || y < o(1) || y > o(1) + map->ny() * res || z < o(2) || z > o(2) + map->nz() * res) return kOutside;
    const double rf = 1.0 / res;
    const int kx = (int)std::floor(rf * (double)x) - (int)std::floor(o(0) / res + 0.5);
    const int ky = (int)std::floor(rf * (double)y) - (int)std::floor(o(1) / res + 0.5);
    const int kz = (int)std::floor(rf * (double)z) - (int)std::floor(o(2) / res + 0.5);
    if (kx < 0 || ky < 0 || kz < 0 || kx >= map->nx() || ky >= map->ny() || kz >= map->nz()) return kUnknown;
    const unsigned v = map->voxels()[((size_t)kx * map->ny() + ky) * map->nz() + kz];
    return v & (kUnknown | kOccupied);
#else
    if (!region.set) return kOutside;
    if (x < region.boxMin(0) || x > region.boxMax(0) || y < region.boxMin(1) || y > region.boxMax(1) || z < region.boxMin(2) || z > region.boxMax(2))
        return kOutside;
    const Eigen::Vector3d p((double)x, (double)y, (double)z);
    unsigned v = 0;
    if (map->isUnknown(p)) v |= kUnknown;
    if (map->isInflatedOccupied(p)) v |= kOccupied;
    return v;
#endif
}

}  // namespace trajPlanner
