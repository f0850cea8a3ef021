// Single-thread host build of the level-synchronous quadtree (octree_algo.h) for CPU unit tests:
// the HIP kernel runs the same source with one workgroup per (frame, level).
// table_cap: capacity of the count tables (0: every sweep walks the keys; -1: what the library would give this level).
#define ORBX_OCTREE_HOST_SIM 1
#include "../../visual-slam_amd/csrc/octree_algo.h"
#include <vector>
#include <cmath>
using namespace orbx;

extern "C" int otsim_distribute(const uint32_t* keys, int nk, int minX, int maxX, int minY, int maxY, int N,
                                uint32_t* out, int cap, int table_cap)
{
    const int W = maxX - minX, H = maxY - minY;
    const int nIni = (int)std::round((float)W / H);
    if (nIni <= 0) return -2;
    const float hX = (float)W / nIni;
    int NC = std::max(N + 3, 4 * nIni) + 1;
    std::vector<OtNode> a(NC), b(NC);
    std::vector<int32_t> cc(4 * NC), candPos(NC), pi(NC), cByPi(NC), keep(NC), posToCand(NC), cc2(2 * NC + 2), scal(16), tmp(8);
    std::vector<uint16_t> nodeOf(nk > 0 ? nk : 1);
    const int TC = table_cap < 0 ? ot_table_size(nIni, ot_table_depth(N, nIni, 4096)) : table_cap;
    std::vector<int32_t> tcnt(TC + 1);
    std::vector<uint16_t> tpos(TC + 1), ent0(NC), ent1(NC), xy(W + H + 1);
    OtShared s{a.data(), b.data(), cc.data(), candPos.data(), pi.data(), cByPi.data(), keep.data(), posToCand.data(), reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(cc2.data()) + 7) & ~(uintptr_t)7), scal.data(), tmp.data(),
               tcnt.data(), tpos.data(), ent0.data(), ent1.data(), TC, xy.data(), xy.data(), W + H};
    OtMemKeys ks{keys, nodeOf.data(), nk};
    return ot_distribute(s, ks, keys, W, H, nIni, hX, N, out, cap);
}

// The direct form the kernel takes for levels whose keys stay in device memory (octree_level_direct): keys counted into the
// tables' deepest level, ot_distribute_tables, every key competing for its node through the leaf table, ranks
// (cell << 10 | index) with `cell_keys` consecutive keys per stand-in FAST cell.  Returns -1 where the kernel would fall
// back to the dense form (steps below the tables' depth) and -3 when the tables have no depth for this set.
extern "C" int otsim_distribute_direct(const uint32_t* keys, int nk, int minX, int maxX, int minY, int maxY, int N,
                                       uint32_t* out, int cap, int table_cap, int cell_keys)
{
    const int W = maxX - minX, H = maxY - minY;
    const int nIni = (int)std::round((float)W / H);
    if (nIni <= 0) return -2;
    const float hX = (float)W / nIni;
    int NC = std::max(N + 3, 4 * nIni) + 1;
    std::vector<OtNode> a(NC), b(NC);
    std::vector<int32_t> cc(4 * NC), candPos(NC), pi(NC), cByPi(NC), keep(NC), posToCand(NC), cc2(2 * NC + 2), scal(16), tmp(8);
    const int TC = table_cap < 0 ? ot_table_size(nIni, ot_table_depth(N, nIni, 4096)) : table_cap;
    const int D = ot_table_depth(N, nIni, TC);
    if (D <= 0) return -3;
    std::vector<int32_t> tcnt(TC + 1, 0);
    std::vector<uint16_t> tpos(TC + 1), ent0(NC), ent1(NC), xy(W + H + 1);
    OtShared s{a.data(), b.data(), cc.data(), candPos.data(), pi.data(), cByPi.data(), keep.data(), posToCand.data(), reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(cc2.data()) + 7) & ~(uintptr_t)7), scal.data(), tmp.data(),
               tcnt.data(), tpos.data(), ent0.data(), ent1.data(), TC, xy.data(), xy.data() + W, W + H};
    ot_build_axis_tables(s, W, H, nIni, hX, D);
    const int offD = ot_table_off(nIni, D), cellsD = ot_table_size(nIni, D) - offD;
    for (int k = 0; k < nk; ++k) ++tcnt[offD + ot_key_cell(s, keys[k])];
    std::vector<uint16_t> leaf(cellsD + 1);
    const int n = ot_distribute_tables(s, H, nIni, hX, N, D, leaf.data());
    if (n < 0) return -1;
    std::vector<unsigned> best(n + 1, 0u);
    for (int k = 0; k < nk; ++k) {
        const unsigned rank = ((unsigned)(k / cell_keys) << 10) | (unsigned)(k % cell_keys);
        const unsigned v = ((keys[k] & 0xFFu) << 24) | (0xFFFFFFu - rank);
        unsigned& bst = best[leaf[ot_key_cell(s, keys[k])]];
        if (v > bst) bst = v;
    }
    for (int p = 0; p < n && p < cap; ++p) {
        const unsigned c = 0xFFFFFFu - (best[p] & 0xFFFFFFu);
        out[p] = keys[(c >> 10) * cell_keys + (c & 1023u)];
    }
    return n;
}

extern "C" void otsim_sweeps(int* on_tables, int* on_keys, int reset)
{
    *on_tables = ot_sim_sweeps[0]; *on_keys = ot_sim_sweeps[1];
    if (reset) ot_sim_sweeps[0] = ot_sim_sweeps[1] = 0;
}
