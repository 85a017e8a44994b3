// astarOcc.cpp — host A* (interface and behaviour of path_search/astarOcc.{h,cpp}: grid centred
// on the start/end midpoint, +0.5 index rounding, diagonal heuristic with the 1+1e-4 tie breaker,
// height band, start/end pushed out of obstacles, 0.2 s budget).
#include <trajectory_planner/path_search/astarOcc.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <queue>

void AStar::initGridMap(std::shared_ptr<mapManager::occMap> occ_map, const Eigen::Vector3i pool_size, double minHeight,
                        double maxHeight) {
    pool_ = pool_size;
    centerIdx_ = Eigen::Vector3i(pool_size(0) / 2, pool_size(1) / 2, pool_size(2) / 2);
    minHeight_ = minHeight;
    maxHeight_ = maxHeight;
    nodes_.assign((size_t)pool_(0) * pool_(1) * pool_(2), Node());
    map_ = occ_map;
}

bool AStar::coord2idx(const Eigen::Vector3d& p, int& x, int& y, int& z) const {
    x = (int)((p(0) - center_(0)) * invStep_ + 0.5) + centerIdx_(0);
    y = (int)((p(1) - center_(1)) * invStep_ + 0.5) + centerIdx_(1);
    z = (int)((p(2) - center_(2)) * invStep_ + 0.5) + centerIdx_(2);
    return !(x < 0 || x >= pool_(0) || y < 0 || y >= pool_(1) || z < 0 || z >= pool_(2));
}

bool AStar::adjustEnds(Eigen::Vector3d s, Eigen::Vector3d e, int (&si)[3], int (&ei)[3]) {
    if (!coord2idx(s, si[0], si[1], si[2]) || !coord2idx(e, ei[0], ei[1], ei[2])) return false;
    if (map_->isInflatedOccupied(idx2coord(si[0], si[1], si[2]))) {
        do {
            s = (s - e).normalized() * step_ + s;
            if (!coord2idx(s, si[0], si[1], si[2])) return false;
        } while (map_->isInflatedOccupied(idx2coord(si[0], si[1], si[2])));
    }
    if (map_->isInflatedOccupied(idx2coord(ei[0], ei[1], ei[2]))) {
        do {
            e = (e - s).normalized() * step_ + e;
            if (!coord2idx(e, ei[0], ei[1], ei[2])) return false;
        } while (map_->isInflatedOccupied(idx2coord(ei[0], ei[1], ei[2])));
    }
    return true;
}

double AStar::heuristic(const int (&a)[3], const int (&b)[3]) const {
    double dx = std::abs(a[0] - b[0]), dy = std::abs(a[1] - b[1]), dz = std::abs(a[2] - b[2]);
    const int diag = (int)std::min(std::min(dx, dy), dz);
    dx -= diag; dy -= diag; dz -= diag;
    double h = 0.0;
    if (dx == 0) h = std::sqrt(3.0) * diag + std::sqrt(2.0) * std::min(dy, dz) + std::abs(dy - dz);
    if (dy == 0) h = std::sqrt(3.0) * diag + std::sqrt(2.0) * std::min(dx, dz) + std::abs(dx - dz);
    if (dz == 0) h = std::sqrt(3.0) * diag + std::sqrt(2.0) * std::min(dx, dy) + std::abs(dx - dy);
    return (1.0 + 1.0 / 10000) * h;
}

// astarOcc.cpp:120-244, statement by statement on the flat pool — including what a textbook A* would do differently:
// a node's `round` stamp is written before the height and occupancy tests (:198), its state and scores are never reset
// between searches, a better path to an open node rewrites its scores in place and leaves the heap as it is (:223-228),
// and the goal test is made when a node is POPPED (:165).  The height band and the map are asked once per node and search
// (they answer the same every time; a node is reached from up to 26 parents), sqrt(dx^2 + dy^2 + dz^2) comes from a table
// of the same three doubles, and the 0.2 s budget is read every 256 expansions instead of every one.
bool AStar::AstarSearch(const double step_size, Eigen::Vector3d start_pt, Eigen::Vector3d end_pt) {
    const auto t0 = std::chrono::steady_clock::now();
    if (++round_ == 0) {   // 65 536 searches on this object: the stamps of the first one would look current again
        for (Node& n : nodes_) n.round = 0;
        round_ = 1;
    }
    step_ = step_size;
    invStep_ = 1 / step_size;
    center_ = (start_pt + end_pt) / 2;
    int si[3], ei[3];
    if (!adjustEnds(start_pt, end_pt, si, ei)) return false;

    std::priority_queue<int, std::vector<int>, ByF> open(ByF{nodes_.data()});
    const int s = flat(si[0], si[1], si[2]);
    const int goal = flat(ei[0], ei[1], ei[2]);
    nodes_[s].round = round_;
    nodes_[s].occ = 0;
    nodes_[s].g = 0;
    nodes_[s].f = heuristic(si, ei);
    nodes_[s].state = 1;
    nodes_[s].parent = -1;
    open.push(s);
    const int py = pool_(1), pz = pool_(2);
    const double stepLen[4] = {0.0, 1.0, std::sqrt(2.0), std::sqrt(3.0)};
    int iter = 0;
    while (!open.empty()) {
        const int cur = open.top();
        open.pop();
        if (cur == goal) {
            pathIdx_.clear();
            for (int n = cur; n >= 0; n = nodes_[n].parent) pathIdx_.push_back(n);
            return true;
        }
        nodes_[cur].state = 2;
        const double gCur = nodes_[cur].g;
        const int cx = cur / (py * pz), cy = (cur / pz) % py, cz = cur % pz;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    if (!dx && !dy && !dz) continue;
                    const int nx = cx + dx, ny = cy + dy, nz = cz + dz;
                    if (nx < 1 || nx >= pool_(0) - 1 || ny < 1 || ny >= py - 1 || nz < 1 || nz >= pz - 1) continue;
                    const int nb = flat(nx, ny, nz);
                    Node& N = nodes_[nb];
                    const bool explored = N.round == round_;
                    if (explored && N.state == 2) continue;   // (a stale CLOSED of an earlier search only ever hides a blocked node)
                    if (!explored) { N.round = round_; N.occ = 0; }
                    if (N.occ == 0) {
                        const Eigen::Vector3d pos = idx2coord(nx, ny, nz);
                        const bool blocked = pos(2) > maxHeight_ || pos(2) < minHeight_ || map_->isInflatedOccupied(pos);
                        N.occ = blocked ? 1 : 2;
                    }
                    if (N.occ == 1) continue;
                    const double g = gCur + stepLen[dx * dx + dy * dy + dz * dz];
                    const int nidx[3] = {nx, ny, nz};
                    if (!explored) {                      // discovered: the only push of this node
                        N.state = 1;
                        N.parent = cur;
                        N.g = g;
                        N.f = g + heuristic(nidx, ei);
                        open.push(nb);
                    } else if (g < N.g) {                 // open, better path: scores rewritten where the node sits in the heap
                        N.parent = cur;
                        N.g = g;
                        N.f = g + heuristic(nidx, ei);
                    }
                }
        if ((++iter & 255) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeLimit)
            return false;
    }
    return false;
}

std::vector<Eigen::Vector3d> AStar::getPath() {
    std::vector<Eigen::Vector3d> path;
    const int py = pool_(1), pz = pool_(2);
    for (auto it = pathIdx_.rbegin(); it != pathIdx_.rend(); ++it)
        path.push_back(idx2coord(*it / (py * pz), (*it / pz) % py, *it % pz));
    return path;
}
