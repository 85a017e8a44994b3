/*
 * dense_occmap.h — the in-tree stand-in for mapManager::occMap (external package map_manager, not vendored by the
 * reference): a dense byte grid with the four methods the reference calls (getRes, isInflatedOccupied,
 * isInflatedOccupiedLine, isUnknown; call sites BT.h:197,199,312,319,332, BT.cpp:292,412,435,841) plus direct access
 * to the bytes for the tools that BUILD maps (the .bt / .pcd readers, the tests) and for the adapter's fast path
 * (mapAdapter.cpp).  The planner sources themselves use the four methods only — `make strict` compiles them against
 * a header that declares nothing else.
 */
#ifndef TRAJECTORY_PLANNER_DENSE_OCCMAP_H
#define TRAJECTORY_PLANNER_DENSE_OCCMAP_H
#include <trajectory_planner/standin/mini_eigen.h>

#include <cmath>
#include <cstdint>
#include <vector>

namespace mapManager {
/*
 * Dense stand-in for mapManager::occMap (external package map_manager, not vendored by the
 * reference).  Contract = include/vigo.h "voxel map": byte per voxel, bit0 inflated-occupied,
 * bit1 unknown, bit2 occupied; index = floor((p - origin)/res); outside => occupied and unknown.
 */
class occMap {
public:
    occMap(int nx, int ny, int nz, const Eigen::Vector3d& origin, double res)
        : nx_(nx), ny_(ny), nz_(nz), origin_(origin), res_(res), vox_((size_t)nx * ny * nz, 0) {}
    double getRes() const { return res_; }
    int nx() const { return nx_; }
    int ny() const { return ny_; }
    int nz() const { return nz_; }
    const Eigen::Vector3d& origin() const { return origin_; }
    std::vector<uint8_t>& voxels() { return vox_; }
    const std::vector<uint8_t>& voxels() const { return vox_; }
    uint8_t& at(int ix, int iy, int iz) { return vox_[((size_t)ix * ny_ + iy) * nz_ + iz]; }
    unsigned byteAt(const Eigen::Vector3d& p) const {
        int ix = (int)std::floor((p(0) - origin_(0)) / res_);
        int iy = (int)std::floor((p(1) - origin_(1)) / res_);
        int iz = (int)std::floor((p(2) - origin_(2)) / res_);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= nx_ || iy >= ny_ || iz >= nz_) return 0xFFu;
        return vox_[((size_t)ix * ny_ + iy) * nz_ + iz];
    }
    bool isInflatedOccupied(const Eigen::Vector3d& p) const { return byteAt(p) & 1u; }
    bool isUnknown(const Eigen::Vector3d& p) const { return (byteAt(p) >> 1) & 1u; }
    bool isInflatedOccupiedLine(const Eigen::Vector3d& p1, const Eigen::Vector3d& p2) const {
        if (isInflatedOccupied(p1) || isInflatedOccupied(p2)) return true;
        Eigen::Vector3d diff = p2 - p1;
        double dist = diff.norm();
        Eigen::Vector3d inc(diff(0) / dist * res_, diff(1) / dist * res_, diff(2) / dist * res_);
        int steps = (int)(dist / res_);
        for (int i = 1; i < steps; ++i) {
            Eigen::Vector3d q(p1(0) + i * inc(0), p1(1) + i * inc(1), p1(2) + i * inc(2));
            if (isInflatedOccupied(q)) return true;
        }
        return false;
    }
    /* bumped by the owner whenever voxels() changes, so planners re-snapshot */
    uint64_t version = 1;
private:
    int nx_, ny_, nz_;
    Eigen::Vector3d origin_;
    double res_;
    std::vector<uint8_t> vox_;
};
}  // namespace mapManager
#endif
