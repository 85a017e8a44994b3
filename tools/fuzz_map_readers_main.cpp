// driver of tools/fuzz_map_readers.sh: loads every file of a directory through the map readers; the sanitizers
// (and the per-file wall-clock check) are the oracle, a refusal is as good as a load
#include <trajectory_planner/octomapBt.h>

#include <chrono>
#include <cstdio>
#include <dirent.h>
#include <string>

int main(int argc, char** argv) {
    using namespace trajPlanner;
    if (argc < 2) return 2;
    DIR* d = opendir(argv[1]);
    if (!d) return 2;
    const double inflate[3] = {0.1, 0.1, 0.0};
    long files = 0, loaded = 0, refused = 0, slow = 0;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() < 5 || name[0] != 'm') continue;
        const std::string path = std::string(argv[1]) + "/" + name;
        const auto t0 = std::chrono::steady_clock::now();
        bool ok;
        if (name.substr(name.size() - 4) == ".pcd") {
            long long n = 0;
            ok = (bool)loadPcdAscii(path, 0.1, inflate, 1, &n);
        } else {
            BtInfo bi;
            ok = (bool)loadOctomapBt(path, inflate, 2, &bi);
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (s > 5.0) { ++slow; std::printf("SLOW %s %.1f s\n", name.c_str(), s); }
        ++files; loaded += ok; refused += !ok;
    }
    closedir(d);
    std::printf("%ld files: %ld loaded, %ld refused, %ld slower than 5 s\n", files, loaded, refused, slow);
    return slow ? 1 : 0;
}
