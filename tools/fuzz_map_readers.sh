#!/bin/bash
# Mutation fuzz of the on-disk map readers (octomap .bt, ASCII .pcd) under AddressSanitizer + UBSan, host only:
# valid files (a generated octree; the reference's map/*.bt when /root/reference is mounted) are truncated,
# bit-flipped and given hostile headers (huge / negative node counts, zero or NaN resolution, missing "data");
# every mutant must come back as a clean refusal or a clean load — never a crash, a hang or a sanitizer report.
#   bash tools/fuzz_map_readers.sh [mutants-per-seed]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
N="${1:-300}"
OUT=/tmp/vigo_mapfuzz
rm -rf "$OUT" && mkdir -p "$OUT"
cd "$ROOT/trajectory_planner_amd/host"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Iinclude -o /tmp/vigo_mapfuzz_drv \
    "$ROOT/tools/fuzz_map_readers_main.cpp" src/octomapBt.cpp
cd "$ROOT"
python3 - "$OUT" "$N" <<'PY'
import glob, os, sys
import numpy as np
sys.path.insert(0, "tests")
out, N = sys.argv[1], int(sys.argv[2])
from test_host_plumbing import write_bt
rng = np.random.default_rng(12)
seeds = []
occ = rng.choice([-1, 0, 1], size=(16, 16, 16), p=[0.3, 0.5, 0.2]).astype(np.int8)
write_bt(os.path.join(out, "seed_gen.bt"), occ, 0.1)
seeds.append(open(os.path.join(out, "seed_gen.bt"), "rb").read())
for p in sorted(glob.glob("/root/reference/map/*.bt"))[:3]:      # authoring container only: read as data
    b = open(p, "rb").read()
    if len(b) < 4_000_000:
        seeds.append(b)
pcd = b"# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 5\nDATA ascii\n" + \
      b"".join(b"%g %g %g\n" % tuple(rng.uniform(-2, 2, 3)) for _ in range(5))
k = 0
def emit(data, ext):
    global k
    open(os.path.join(out, f"m{k:05d}{ext}"), "wb").write(data); k += 1
for s in seeds:
    hdr_end = s.index(b"data\n") + 5
    for i in range(N):
        b = bytearray(s)
        mode = i % 6
        if mode == 0:   b = b[:rng.integers(0, len(b))]                                   # truncation
        elif mode == 1:
            for _ in range(int(rng.integers(1, 20))): b[rng.integers(hdr_end, len(b))] ^= 1 << int(rng.integers(0, 8))   # payload bit flips
        elif mode == 2:
            for _ in range(int(rng.integers(1, 6))): b[rng.integers(0, hdr_end)] = int(rng.integers(0, 256))            # header corruption
        elif mode == 3:
            size = rng.choice([b"-1", b"0", b"99999999999999999999", b"2147483648", b"abc", b"1"])
            b = bytearray(s[:s.index(b"size ")] + b"size " + size + s[s.index(b"\nres "):])
        elif mode == 4:
            res = rng.choice([b"0", b"-0.1", b"nan", b"inf", b"1e-320", b"1e308", b""])
            i0 = s.index(b"\nres ") + 5; i1 = s.index(b"\n", i0)
            b = bytearray(s[:i0] + res + s[i1:])
        else:
            b = bytearray(s[:hdr_end - 5] + s[hdr_end:]) if i % 12 == 5 else bytearray(s[:hdr_end] + bytes(rng.integers(0, 256, size=int(rng.integers(0, 4096)), dtype=np.uint8)))
        emit(bytes(b), ".bt")
for i in range(N):
    b = bytearray(pcd)
    mode = i % 4
    if mode == 0:   b = b[:rng.integers(0, len(b))]
    elif mode == 1:
        for _ in range(int(rng.integers(1, 8))): b[rng.integers(0, len(b))] = int(rng.integers(0, 256))
    elif mode == 2: b = bytearray(pcd.replace(b"POINTS 5", rng.choice([b"POINTS -3", b"POINTS 999999999999", b"POINTS x"])))
    else:           b = bytearray(pcd.replace(b"0.7\nFIELDS", b"0.7\nFIELDS q\nFIELDS")) + b"1e999 nan -inf\n" * int(rng.integers(1, 4))
    emit(bytes(b), ".pcd")
print(f"{k} mutants from {len(seeds)} .bt seeds + 1 .pcd seed in {out}")
PY
timeout 900 /tmp/vigo_mapfuzz_drv "$OUT"
