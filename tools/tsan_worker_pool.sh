#!/bin/bash
# host/src/workerPool.h under ThreadSanitizer (CPU only):  bash tools/tsan_worker_pool.sh
set -e
cd "$(dirname "$0")"
g++ -std=c++17 -g -O1 -fsanitize=thread -pthread -o /tmp/vigo_tsan_pool tsan_worker_pool_main.cpp
/tmp/vigo_tsan_pool
