// Host-side handshake between the C-ABI translation unit (ppcr_hip.hip) and the K1 translation units
// (ppcr_nn_tile.hip, compiled once per list width M): everything one K1 launch needs, by value.
#pragma once
#include "ppcr_device.hip.h"

namespace ppcr {

// Halo capacity of the steady-state (16-slot) variant: 31.3 KB of LDS, five workgroups per CU (1792: four).
constexpr int kCapSteady = 1728;
// ... of the Verlet variant (four workgroups per CU either way: its list path holds 48 coordinates in registers), whose
// lists reach further than the plain search's cut-off: larger cells, larger halos
constexpr int kCapVerlet = 1920;
// ... and of that variant where the cloud is denser (a block's halo outgrows 1920 candidates for more blocks than can be split):
// the first association's 2240-candidate tile, three workgroups per CU (TileLaunch::verlet_big_tile)
constexpr int kCapVerletBig = 2240;
// ... and of the Verlet variant of the mid-width lists (the command line's 20 neighbours: 32-slot Verlet lists, 36-slot scan
// lists — 18.9 KB — beside the halo: 1600 candidates keep the kernel at four workgroups per CU)
constexpr int kCapVerletMid = 1600;
// workgroups (= partial-sum slots when K23 is folded in) of a steady-state K1 launch over nb blocks of 256 queries
inline int steady_grid(int nb, bool all_halves) { return all_halves ? 2 * ((nb + 7) & ~7) : nb + dev::kMaxSplit; }

struct TileLaunch {
    hipStream_t stream;
    float4 *src;                  // the (sorted) source; moved in place when pm.enabled
    int ns;
    const float4 *tgt;            // the cell-sorted target
    const int *cell_start;
    dev::GridDesc grid;
    float r2;                     // search radius^2 of THIS pass (the first pass of a two-pass search: (radius / reach)^2)
    int reach;                    // > 1: two-pass search — rows that come back short are searched again with r2_full over a
    float r2_full;                //      stencil `reach` cells wide (nn_wide_kernel) ...
    int *short_list;              // when short_count is given: the rows of handed-over workgroups (and the short rows of a two-pass
    unsigned *short_count;        //      search) are listed here [ns] + counted and nn_wide_kernel searches them; there is no cleanup
                                  //      launch then
    unsigned *short_next;         // the other counter of that list's ping-pong pair (nullable): every launch leaves it at zero
    unsigned *short_seen;         // diagnostic: the count nn_wide_kernel saw
    const dev::GridLevel *levels; // multi-level search (n_levels > 1; needs short_count): the level table in device memory,
    int n_levels, base_level;     //      ascending r2_cap; tgt / cell_start / grid above are the base level's
    float r2_cap[dev::kMaxLevels]; //     the levels' r2_cap (host copy of the table's column)
    const unsigned short *level_in;   //  per block and half: the feedback words the previous launch left (UnansweredRows::level_in) ...
    unsigned short *level_out;        //  ... and where this launch leaves its own [2 x blocks of 256 queries each]
    unsigned *level_dbg;          //      diagnostic counters (nullable)
    int m;                        // max_neighbours (<= the M of the variant that is called)
    int *nbr, *cnt;               // the ELL association [m][ns], [ns]
    unsigned *dm2;                // per query: float bits of its m-th neighbour's d2 (the temporal cut-off)
    int dm2_in;                   // dm2 of the previous association is usable
    int short_lists;              // option short_lists
    int list_all;                 // with short_count: try no tile, list every row (UnansweredRows::list_all)
    int all_halves;               // steady-state variant: scan every block as two half-blocks (small clouds; see SplitTable)
    unsigned long long *stamps;   // diagnostic build only (nullptr otherwise)
    int *ovf_list;                // workgroups handed to the cleanup kernel by this launch ...
    unsigned *ovf_now, *ovf_next; // ... counted here; the other counter of the ping-pong pair
    bool quiet;                   // the last association this handle heard from handed nothing over
    unsigned handed_last;         // ... how many workgroups it handed over (~0u: not known)
    unsigned char *split_flag;    // the split table (see SplitTable)
    int *split_list;
    unsigned *split_state;        // {registrations, registrations visible to extra workgroups}
    dev::PendingMove pm;
    dev::VerletLists verlet;        // steady-state Verlet lists (buffers of [16][ns] + per-row state), used when verlet_mode != 0:
    int verlet_mode;                //   0 off; 1 this association builds every row's list; 2 workgroups whose rows' lists still
                                    //   hold answer from them, the others search and rebuild
    int verlet_rows;                //   widths up to 10: the kernel variant that rebuilds a few failing rows inside the workgroup
    int verlet_big_tile;            //   widths up to 10: the 2240-candidate tile (three workgroups per CU) — denser clouds
    const dev::FusedMoments *fuse;  // fold K23 into K1 at this pose / model when the steady-state variant runs
    int fuse_tm;                    // ... in this compiled form: 0 Gaussian, 8 t with v + dim = 8, -3 t with another integer v + dim
    const dev::FoldSolve *fold;     // ... and the fold-and-solve step into the cleanup launch
    dev::LoopState *loop_st;        // device-paced loop: every launch steps aside while its abort flag is up (nullable)
    void (*between)(void *);        // called between the two launches (profiling scopes), may be null
    void (*between2)(void *);       // ... and before the second pass of a two-pass search
    void *between_arg;
    // out
    bool fused, merged;
    bool verlet_built;              // the launch wrote / kept Verlet lists (the steady-state variant ran in Verlet mode)
};

// One entry per compiled-in list width (register list of the selection); defined in ppcr_nn_tile.hip.
__attribute__((visibility("hidden"))) void launch_tile_m4(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m5(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m8(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m10(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m16(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m20(TileLaunch &t);
__attribute__((visibility("hidden"))) void launch_tile_m32(TileLaunch &t);

}  // namespace ppcr
