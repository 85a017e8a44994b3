/*
 * astarOcc.h — 26-connected A* on the occupancy map, feeding guide points to the B-spline
 * optimizer.  Public interface of the reference's AStar (path_search/astarOcc.h:41-85:
 * initGridMap / AstarSearch / getPath); own implementation on flat arrays.  Stays on the host by
 * design: an irregular, serial search per collision segment.
 */
#ifndef ASTAROCC_H
#define ASTAROCC_H
#include <trajectory_planner/compat.h>

#include <memory>
#include <vector>

class AStar {
public:
    typedef std::shared_ptr<AStar> Ptr;
    AStar() {}
    void initGridMap(std::shared_ptr<mapManager::occMap> occ_map, const Eigen::Vector3i pool_size,
                     double minHeight = 0.0, double maxHeight = 3.0);
    bool AstarSearch(const double step_size, Eigen::Vector3d start_pt, Eigen::Vector3d end_pt);
    std::vector<Eigen::Vector3d> getPath();
    double timeLimit = 0.2;  /* seconds, astarOcc.cpp:231 */

private:
    struct Node {
        int round = 0;
        uint8_t state = 0;  // 1 open, 2 closed
        double g = 0, f = 0;
        int parent = -1;
    };
    std::shared_ptr<mapManager::occMap> map_;
    Eigen::Vector3i pool_, centerIdx_;
    double minHeight_ = 0.0, maxHeight_ = 3.0;
    double step_ = 0.1, invStep_ = 10.0;
    Eigen::Vector3d center_;
    std::vector<Node> nodes_;
    std::vector<int> pathIdx_;
    int round_ = 0;

    int flat(int x, int y, int z) const { return (x * pool_(1) + y) * pool_(2) + z; }
    Eigen::Vector3d idx2coord(int x, int y, int z) const {
        return Eigen::Vector3d((x - centerIdx_(0)) * step_ + center_(0), (y - centerIdx_(1)) * step_ + center_(1),
                               (z - centerIdx_(2)) * step_ + center_(2));
    }
    bool coord2idx(const Eigen::Vector3d& p, int& x, int& y, int& z) const;
    bool adjustEnds(Eigen::Vector3d s, Eigen::Vector3d e, int (&si)[3], int (&ei)[3]);
    double heuristic(const int (&a)[3], const int (&b)[3]) const;
};
#endif
