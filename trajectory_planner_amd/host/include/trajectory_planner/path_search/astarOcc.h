/*
 * astarOcc.h — 26-connected A* on the occupancy map, feeding guide points to the B-spline
 * optimizer.  Public interface of the reference's AStar (path_search/astarOcc.h:41-85:
 * initGridMap / AstarSearch / getPath) and its algorithm statement by statement — the std::priority_queue discipline
 * with its in-place score rewrites included (astarOcc.cpp:223-228) — on one flat node pool.  Stays on the host by
 * design: an irregular, serial search per collision segment.  Pinned by
 * tests/test_astar_restatement.py.
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
    struct Node {           // 24 bytes: the pool is walked at random, its size is what the search costs
        double g = 0, f = 0;    // astarOcc.h:13,29: the reference's `inf` is 1 >> 20 == 0
        int parent = -1;
        uint16_t round = 0; // the search that reached the node last (the pool is cleared when the counter wraps)
        uint8_t state = 0;  // 1 open, 2 closed; like the reference's, NOT reset between searches
        uint8_t occ = 0;    // this search's verdict of the height band + the map at the node: 0 not asked yet, 1 blocked, 2 free
    };
    // the reference's open set, astarOcc.h:33-38, :70: a std::priority_queue ordered by the nodes' CURRENT fScore — a node
    // is pushed once, when it is discovered, and a better path found later rewrites its fScore in place
    // (astarOcc.cpp:223-228) without re-establishing the heap; which of several equal or stale entries comes out next
    // is decided by libstdc++'s heap algorithms, reproduced here by using the same container with the same comparator
    struct ByF {
        const Node* nodes;
        bool operator()(int a, int b) const { return nodes[a].f > nodes[b].f; }
    };
    std::shared_ptr<mapManager::occMap> map_;
    Eigen::Vector3i pool_, centerIdx_;
    double minHeight_ = 0.0, maxHeight_ = 3.0;
    double step_ = 0.1, invStep_ = 10.0;
    Eigen::Vector3d center_;
    std::vector<Node> nodes_;
    std::vector<int> pathIdx_;
    uint16_t round_ = 0;

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
