// octomapBt.cpp — .bt reader (see the header for the format).  Two passes over the byte stream:
// bounds, then fill.  Own implementation of the published octomap binary format.
#ifdef VIGO_WITH_ROS
#error "tools of the in-tree dense map (standin/dense_occmap.h): not part of a build against map_manager"
#endif
#include <trajectory_planner/octomapBt.h>
#include <climits>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

namespace trajPlanner {
namespace {

struct Walker {
    const unsigned char* p;
    const unsigned char* end;
    long long nodes = 0;
    bool ok = true;
    // pass 1: bounds; pass 2: fill
    int kmin[3] = {1 << 30, 1 << 30, 1 << 30}, kmax[3] = {-(1 << 30), -(1 << 30), -(1 << 30)};
    long long occ = 0, fre = 0;
    mapManager::occMap* map = nullptr;
    int o[3] = {0, 0, 0};  // key of map voxel (0,0,0)

    void leaf(int x0, int y0, int z0, int size, bool occupied) {
        if (occupied) occ += (long long)size * size * size; else fre += (long long)size * size * size;
        const int lo[3] = {x0, y0, z0};
        for (int a = 0; a < 3; ++a) {
            kmin[a] = std::min(kmin[a], lo[a]);
            kmax[a] = std::max(kmax[a], lo[a] + size - 1);
        }
        if (!map) return;
        // (the leaf clipped to the map first: a pruned leaf high in the tree is up to 32768 keys wide)
        const int b0[3] = {std::max(x0 - o[0], 0), std::max(y0 - o[1], 0), std::max(z0 - o[2], 0)};
        const int b1[3] = {std::min(x0 + size - o[0], map->nx()), std::min(y0 + size - o[1], map->ny()), std::min(z0 + size - o[2], map->nz())};
        for (int ix = b0[0]; ix < b1[0]; ++ix)
            for (int iy = b0[1]; iy < b1[1]; ++iy)
                for (int iz = b0[2]; iz < b1[2]; ++iz) {
                    uint8_t& v = map->at(ix, iy, iz);
                    v = (uint8_t)((v & ~2u) | (occupied ? 4u : 0u));  // observed; occupied or free
                }
    }

    // one inner node: 2 bytes, then its inner children in index order (pre-order)
    void inner(int x0, int y0, int z0, int size, int depth) {
        if (!ok) return;
        if (end - p < 2 || depth >= 16) { ok = false; return; }
        const unsigned bits = (unsigned)p[0] | ((unsigned)p[1] << 8);
        p += 2;
        ++nodes;
        const int h = size / 2;
        unsigned kind[8];
        for (int i = 0; i < 8; ++i) kind[i] = (bits >> (2 * i)) & 3u;  // bit0 | bit1 << 1
        for (int i = 0; i < 8; ++i) {
            const int cx = x0 + ((i & 1) ? h : 0), cy = y0 + ((i & 2) ? h : 0), cz = z0 + ((i & 4) ? h : 0);
            if (kind[i] == 1u) { ++nodes; leaf(cx, cy, cz, h, false); }       // bit0=1, bit1=0: free
            else if (kind[i] == 2u) { ++nodes; leaf(cx, cy, cz, h, true); }   // bit0=0, bit1=1: occupied
        }
        for (int i = 0; i < 8; ++i) {
            if (kind[i] != 3u) continue;
            const int cx = x0 + ((i & 1) ? h : 0), cy = y0 + ((i & 2) ? h : 0), cz = z0 + ((i & 4) ? h : 0);
            inner(cx, cy, cz, h, depth + 1);
        }
    }
};

}  // namespace

// sanity limits of the readers: a file outside them is refused, not allocated
static const double kMinRes = 1e-4, kMaxRes = 1e4;           // metres per voxel
static const long long kMaxVoxels = 1LL << 31;               // bytes of the dense grid (2 GiB)

// bit0 = occupied (bit2) dilated by `inflate` metres per axis (separable box dilation)
static void inflateOccupied(mapManager::occMap& m, const double inflate[3]) {
    const double res = m.getRes();
    const int n[3] = {m.nx(), m.ny(), m.nz()};
    mapManager::occMap* map = &m;
    int r[3];
    for (int a = 0; a < 3; ++a) {   // a radius beyond the map's extent changes nothing: clamped (and never out of int range)
        const double ra = std::ceil(inflate[a] / res - 1e-9);
        r[a] = !(ra > 0) ? 0 : (ra >= n[a] ? n[a] : (int)ra);
    }
    std::vector<uint8_t> cur((size_t)n[0] * n[1] * n[2]), nxt(cur.size());
    for (size_t i = 0; i < cur.size(); ++i) cur[i] = (map->voxels()[i] & 4u) ? 1 : 0;
    auto idx = [&](int x, int y, int z) { return ((size_t)x * n[1] + y) * n[2] + z; };
    for (int axis = 0; axis < 3; ++axis) {
        std::fill(nxt.begin(), nxt.end(), 0);
        for (int x = 0; x < n[0]; ++x) for (int y = 0; y < n[1]; ++y) for (int z = 0; z < n[2]; ++z) {
            if (!cur[idx(x, y, z)]) continue;
            for (int d = -r[axis]; d <= r[axis]; ++d) {
                int q[3] = {x, y, z};
                q[axis] += d;
                if (q[axis] < 0 || q[axis] >= n[axis]) continue;
                nxt[idx(q[0], q[1], q[2])] = 1;
            }
        }
        cur.swap(nxt);
    }
    for (size_t i = 0; i < cur.size(); ++i) if (cur[i]) map->voxels()[i] |= 1u;
}

std::shared_ptr<mapManager::occMap> loadOctomapBt(const std::string& path, const double inflate[3], int margin, BtInfo* info) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return nullptr;
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    // header
    size_t pos = 0;
    long long size = -1;
    double res = 0.0;
    bool data = false;
    while (pos < buf.size()) {
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n') ++e;
        std::string line(buf.begin() + pos, buf.begin() + e);
        pos = e + 1;
        if (line.rfind("size", 0) == 0) size = std::atoll(line.c_str() + 4);
        else if (line.rfind("res", 0) == 0) res = std::atof(line.c_str() + 3);
        else if (line.rfind("data", 0) == 0) { data = true; break; }
    }
    if (!data || size < 0 || !(res >= kMinRes && res <= kMaxRes)) return nullptr;

    Walker w1;
    w1.p = buf.data() + pos;
    w1.end = buf.data() + buf.size();
    if (size > 0) w1.inner(-32768, -32768, -32768, 65536, 0);   // keys relative to 32768
    if (!w1.ok) return nullptr;
    const long long parsed = w1.nodes;  // inner nodes (root included) + leaves
    BtInfo bi;
    bi.nodes_header = size;
    bi.nodes_parsed = parsed;
    bi.bytes_consumed = (long long)(w1.p - (buf.data() + pos));
    bi.res = res;
    bi.occupied = w1.occ;
    bi.free_ = w1.fre;
    if (w1.kmax[0] < w1.kmin[0]) { if (info) *info = bi; return nullptr; }
    for (int a = 0; a < 3; ++a) { bi.key_min[a] = w1.kmin[a]; bi.key_max[a] = w1.kmax[a]; }
    if (info) *info = bi;

    int n[3], o[3];
    for (int a = 0; a < 3; ++a) {
        o[a] = w1.kmin[a] - margin;
        n[a] = w1.kmax[a] - w1.kmin[a] + 1 + 2 * margin;
    }
    if (margin < 0 || margin > 4096 || (long long)n[0] * n[1] * n[2] > kMaxVoxels) return nullptr;   // a dense byte grid of that extent is not a map
    auto map = std::make_shared<mapManager::occMap>(n[0], n[1], n[2], Eigen::Vector3d(o[0] * res, o[1] * res, o[2] * res), res);
    std::fill(map->voxels().begin(), map->voxels().end(), (uint8_t)2);  // everything unknown until observed
    Walker w2;
    w2.p = buf.data() + pos;
    w2.end = buf.data() + buf.size();
    w2.map = map.get();
    for (int a = 0; a < 3; ++a) w2.o[a] = o[a];
    w2.inner(-32768, -32768, -32768, 65536, 0);
    inflateOccupied(*map, inflate);
    ++map->version;
    return map;
}

// ASCII .pcd (PCL "DATA ascii", FIELDS x y z ...): every point marks its voxel occupied; voxels without a
// point are free and known (a point cloud map carries no unknown space).  The grid covers the points'
// bounding box plus `margin` voxels, origin on the res lattice.
std::shared_ptr<mapManager::occMap> loadPcdAscii(const std::string& path, double res, const double inflate[3], int margin,
                                                 long long* pointsRead) {
    std::ifstream f(path);
    if (!f || !(res >= kMinRes && res <= kMaxRes) || margin < 0 || margin > 4096) return nullptr;
    std::string line;
    long long declared = -1;
    bool ascii = false, xyzFirst = false;
    while (std::getline(f, line)) {
        if (line.rfind("FIELDS", 0) == 0) xyzFirst = line.find("x y z") != std::string::npos;
        if (line.rfind("POINTS", 0) == 0) declared = std::atoll(line.c_str() + 6);
        if (line.rfind("DATA", 0) == 0) { ascii = line.find("ascii") != std::string::npos; break; }
    }
    if (!ascii || !xyzFirst) return nullptr;
    std::vector<double> pts;
    double x, y, z;
    while (std::getline(f, line)) {
        if (std::sscanf(line.c_str(), "%lf %lf %lf", &x, &y, &z) == 3 && std::fabs(x) / res < 1e9 && std::fabs(y) / res < 1e9 && std::fabs(z) / res < 1e9) {
            pts.push_back(x); pts.push_back(y); pts.push_back(z);
        }
    }
    const long long np = (long long)pts.size() / 3;
    if (pointsRead) *pointsRead = np;
    if (np == 0 || (declared >= 0 && np != declared)) return nullptr;
    long long lo[3] = {LLONG_MAX, LLONG_MAX, LLONG_MAX}, hi[3] = {LLONG_MIN, LLONG_MIN, LLONG_MIN};
    for (long long i = 0; i < np; ++i)
        for (int a = 0; a < 3; ++a) {
            const long long k = (long long)std::floor(pts[3 * i + a] / res);
            lo[a] = std::min(lo[a], k);
            hi[a] = std::max(hi[a], k);
        }
    long long n[3];
    for (int a = 0; a < 3; ++a) { lo[a] -= margin; n[a] = hi[a] - lo[a] + 1 + margin; if (n[a] > 4096) return nullptr; }
    if (n[0] * n[1] * n[2] > kMaxVoxels) return nullptr;
    auto map = std::make_shared<mapManager::occMap>((int)n[0], (int)n[1], (int)n[2], Eigen::Vector3d(lo[0] * res, lo[1] * res, lo[2] * res), res);
    for (long long i = 0; i < np; ++i) {
        const int ix = (int)((long long)std::floor(pts[3 * i] / res) - lo[0]), iy = (int)((long long)std::floor(pts[3 * i + 1] / res) - lo[1]),
                  iz = (int)((long long)std::floor(pts[3 * i + 2] / res) - lo[2]);
        map->at(ix, iy, iz) |= 4u;
    }
    inflateOccupied(*map, inflate);
    ++map->version;
    return map;
}

}  // namespace trajPlanner
