// vigo_corridor.hpp — launcher of the corridor checker's two passes (vigo_corridor.hip, k_corridor).
// Supersedes launch_corridor_check() as declared in vigo_internal.hpp: that declaration has no work list and is left
// there only because vigo_internal.hpp is part of the solve kernels' build id (profiles/pmc_*.json are tied to it).
// For the same reason this header is not in the Makefile's HDRS list (the Makefile is part of that id too): the two files
// that include it are vigo_corridor.hip and vigo_api.cpp — touch them, or `make clean`, after editing it.
#pragma once

#include "vigo_internal.hpp"

namespace vigo {

// todo: S ints of device scratch (which segments the first pass left to the second); clock_ws: corridor_clock_ws_bytes(S)
// bytes of device scratch, 8-byte aligned, for the segments' sample-clock tables, or NULL (every workgroup then builds its own)
size_t corridor_clock_ws_bytes(int S);
int launch_corridor_check2(hipStream_t s, const GridView& g, int S, int deg, const double* coeffs, const int32_t* n_samp,
                           const double* delT, const double box[3], double map_res, uint8_t* out_flag, int32_t* out_first,
                           int32_t* out_count, int* todo, void* clock_ws);

}  // namespace vigo
