// TEST INFRASTRUCTURE ONLY (like oracle/): lets dcreg_amd/csrc/device/search.hpp -- the per-thread device functions of the
// HIP path -- compile for the host, so tests can replay the device ALGORITHM on the CPU (exactness of the grid search on
// ties / borders / empty space, visit counts for the wave cost model, the reduced-instruction plane fit) without a GPU.
// Nothing under dcreg_amd/ includes, links or loads this; libdcreg_hip.so has no host path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

struct float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
template <class T> inline T min(T a, T b) { return b < a ? b : a; }
template <class T> inline T max(T a, T b) { return a < b ? b : a; }
struct EmuThreadIdx { unsigned x = 0, y = 0, z = 0; };
static thread_local EmuThreadIdx threadIdx;
inline unsigned long long clock64() { return 0ull; }

// visit counters of the query being replayed (DCREG_STAT in search.hpp)
struct EmuStats {
    uint32_t table_loads;     // cell_start / gap loads
    uint32_t rows;            // (y,z) rows tested in the shell walk
    uint32_t runs;            // x-runs scanned (phase B + shells)
    uint32_t trips;           // 4-candidate trips
    uint32_t faces;           // shell faces entered
    uint32_t face_skips;      // shell faces skipped by the empty-space field
};
static thread_local EmuStats emu_stats;
// optional trace of the ring walk: (ring << 20 | (dz + 512) << 10 | (dy + 512)) * 2 + which, trips
struct EmuTrace { uint32_t *buf = nullptr; uint32_t cap = 0, n = 0; };
static thread_local EmuTrace emu_trace;
inline void emu_trace_push(uint32_t key, uint32_t trips) {
    if (emu_trace.buf && emu_trace.n + 2 <= emu_trace.cap) { emu_trace.buf[emu_trace.n++] = key; emu_trace.buf[emu_trace.n++] = trips; }
}
