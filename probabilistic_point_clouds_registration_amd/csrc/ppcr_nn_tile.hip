// K1 launches for ONE compiled-in list width M (-DPPCR_TILE_M=<4|5|8|10|16|20|32>): the build compiles this file once
// per width, in parallel — the instantiations of nn_fast_kernel / nn_tile_cleanup_kernel are most of the library's
// compile time.  Host code here only picks the variant and launches; the C-ABI translation unit (ppcr_hip.hip) fills
// the TileLaunch and owns every buffer it names.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ppcr_nn_tile.hip.h"
#include "ppcr_nn_tile_launch.hip.h"

#ifndef PPCR_TILE_M
#error "compile with -DPPCR_TILE_M=<list width>"
#endif

namespace ppcr {
namespace {

using namespace ppcr::dev;

// nn_fast_kernel over every block, then nn_tile_cleanup_kernel over the blocks the fast flavour handed over (usually
// none: the cleanup launch then costs a few microseconds of an empty grid).
// LDS budget per 256-query block: halo CAP*13 B (x, y, z + a row byte) + list C*512 B (+0.4 KB tables).  With the
// column order of the source typical halos are ~1000-1400 candidates at the benchmark density; the margin keeps
// denser clouds and drifted sources in the fast flavour (measured when the source was ordered in 4x4x4 bricks, fresh /
// drifted: CAP 2048: 249/341 us, 2176: 231/246, 2240: 230/246, 2272: 302/318).
//
// t.fuse: when given (and the steady-state variant runs) K23 is folded into K1 for that pose/model; t.fused tells
// whether it was — the partials then have one slot per fast-kernel workgroup (nb + kMaxSplit).
// t.fold (with fuse, steady-state variant only): when given, the fold-and-solve step rides in the cleanup launch and
// t.merged tells whether it did.
// PPCR_LIST_NOCLAMP: every instantiation of nn_fast_kernel this unit can launch must own exactly FastLds::kBytes of LDS
// (see FastLds).  Checked once per process and list width; a mismatch is a build defect, not a run-time condition: abort.
template <int M, int C, int CAP, bool STAMPS, int FTM, bool MULTI, int VERLET = 0>
bool fast_kernel_lds_ok()
{
#if PPCR_LIST_NOCLAMP
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&nn_fast_kernel<M, C, CAP, STAMPS, FTM, MULTI, VERLET>)) != hipSuccess) return false;
    return (int)attr.sharedSizeBytes == FastLds<C, CAP, MULTI, ((VERLET != 0 && FTM != -2) ? M : 0)>::kAllocBytes;
#else
    return true;
#endif
}
template <int M>
void check_fast_kernel_lds()
{
    static const bool ok = [] {
        constexpr int C = (M <= 24) ? 32 : 48;
        constexpr int CAP = (M <= 24) ? 2240 : 2048;
        bool good = fast_kernel_lds_ok<M, C, CAP, false, -2, false>() && fast_kernel_lds_ok<M, C, CAP, false, -2, true>();
        if constexpr (M <= 24) good = good && fast_kernel_lds_ok<M, verlet_scan_slots(M), (M <= 12 ? kCapVerlet : CAP), false, -2, true, 1>();
        if constexpr (M > 12 && M <= 24)
            good = good && fast_kernel_lds_ok<M, (M <= 16 ? 24 : 28), (M <= 16 ? 2048 : 1920), false, -2, false>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerletMid, false, -2, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), CAP, false, -2, false, 1>();
        if constexpr (M <= 12)
            good = good && fast_kernel_lds_ok<M, 16, kCapSteady, false, -2, false>() && fast_kernel_lds_ok<M, 16, kCapSteady, false, 8, false>() &&
                   fast_kernel_lds_ok<M, 16, kCapSteady, false, 0, false>() && fast_kernel_lds_ok<M, 16, kCapSteady, false, -3, false>() &&
                   fast_kernel_lds_ok<M, 16, kCapSteady, false, -2, true>() &&
                   // (the Verlet variants: their allocation is the larger of the search's and the list path's)
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, -2, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, 8, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, 0, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, -3, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, -2, false, 2>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, 8, false, 2>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, 0, false, 2>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), kCapVerlet, false, -3, false, 2>() &&
                   // (... in the large tile: TileLaunch::verlet_big_tile)
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), CAP, false, -2, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), CAP, false, 8, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), CAP, false, 0, false, 1>() &&
                   fast_kernel_lds_ok<M, verlet_scan_slots(M), CAP, false, -3, false, 1>();
        // the diagnostic (option "stamps") instantiations launch_tile can reach
        if constexpr (M == 10)
            good = good && fast_kernel_lds_ok<M, 16, kCapSteady, true, -2, true>() && fast_kernel_lds_ok<M, 16, kCapSteady, true, -2, false>() &&
                   fast_kernel_lds_ok<M, C, CAP, true, -2, false>();
        if constexpr (M == 10 || M == 20) good = good && fast_kernel_lds_ok<M, C, CAP, true, -2, true>();
        return good;
    }();
    if (!ok) {
        std::fprintf(stderr, "libppcr_hip: nn_fast_kernel<%d, ...> owns LDS besides its FastLds allocation: the unclamped list stores "
                             "would overwrite it (build defect)\n", M);
        std::abort();
    }
}

template <int M>
void launch_tile(TileLaunch &t)
{
    check_fast_kernel_lds<M>();
    unsigned long long *const st = t.stamps;
    constexpr int C = (M <= 24) ? 32 : 48;
    constexpr int CAP = (M <= 24) ? 2240 : 2048;
    // Mid-width lists (12 < M <= 24: the command line's 20 neighbours) have a steady-state variant of their own: with the
    // temporal cut-off the lists end near m entries, so 24 / 28 slots do (an overflowing lane tightens its threshold and
    // scans again), and with a 2048- / 1920-candidate halo the kernel takes ~40 KB of LDS: FOUR workgroups per CU instead
    // of three.  200k points are 782 blocks: three per CU run them in two residency rounds (768 slots), four in one.
    constexpr bool kMid = (M > 12 && M <= 24);
    constexpr int C2 = (M <= 16) ? 24 : 28, CAP2 = (M <= 16) ? 2048 : 1920;
    // the cleanup flavour and the second pass of a two-pass search meet neighbourhoods of 50+ in-radius candidates without
    // a cut-off: lists of 64 (one scan and one selection where 32 slots need two or three rescans); 62 KB of LDS: two
    // workgroups per CU for kernels that run on a few workgroups
    constexpr int CC = 64;
    const int nb = (t.ns + 255) / 256;
    // the steady-state variant acts on the split table (extra workgroups) and extends it; the first association only
    // extends it (blocks whose fresh halo is already close to the steady-state capacity)
    // (small clouds, steady state: every block as two half-blocks, see SplitTable::all_halves)
    const bool halves = t.all_halves != 0;
    const SplitTable split_on = halves ? SplitTable{nullptr, nullptr, nullptr, nullptr, 0, INT_MAX, 1}
                                       : SplitTable{t.split_flag, t.split_list, t.split_state, t.split_state + 1, kMaxSplit, kCapSteady * 15 / 16, 0};
    const SplitTable split_off = ((M <= 12 || kMid) && !halves)
                                     ? SplitTable{t.split_flag, t.split_list, t.split_state, t.split_state + 1, 0, (kMid ? CAP2 : kCapSteady) * 15 / 16, 0}
                                     : SplitTable{nullptr, nullptr, nullptr, nullptr, 0, INT_MAX, 0};  // no steady-state variant to split for
    const int grid_steady = steady_grid(nb, halves);
    FusedMoments fm_none;
    std::memset(&fm_none, 0, sizeof(fm_none));
    VerletLists vv_none;
    std::memset(&vv_none, 0, sizeof(vv_none));
    const LoopReset lr{t.loop_st};
    // (one-pass search: only rows marked unsearched are listed, its short rows are final)
    const bool multi = t.n_levels > 1 && t.levels != nullptr && t.short_count != nullptr;
    UnansweredRows un{(t.short_count ? t.short_list : nullptr), t.short_count, t.short_next, ((t.reach > 1 || multi) ? t.m : 0),
                      (multi ? t.levels : nullptr), (multi ? t.n_levels : 0), t.base_level, t.r2_full, {},
                      (multi ? t.level_in : nullptr), (multi ? t.level_out : nullptr), (multi ? t.level_dbg : nullptr),
                      ((t.list_all && t.short_count != nullptr) ? 1 : 0)};
    for (int l = 0; l < kMaxLevels; l++) un.r2_cap[l] = (multi && l < t.n_levels) ? t.r2_cap[l] : 0.f;
    // the steady-state (16-slot) variant runs with a halo capacity that gives FIVE workgroups per CU (31.3 KB of LDS)
    // and splits the few blocks that outgrow it; the first association (32 slots, three per CU) keeps the large one
#define PPCR_FAST(Cc, STAMPc, FTMc, FMc)                                                                               \
    nn_fast_kernel<M, Cc, (Cc <= 16 ? kCapSteady : CAP), STAMPc, FTMc><<<(Cc <= 16 ? grid_steady : nb), 256, 0, t.stream>>>( \
        t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now,  \
        t.ovf_next, (Cc <= 16 ? split_on : split_off), st, FMc, lr, un, vv_none)
    // steady state with Verlet lists: nn_fast_kernel<..., VERLET> answers from the lists where they still hold
#define PPCR_FAST_V1(FTMc, FMc, Kc, CAPc)                                                                               \
    nn_fast_kernel<M, verlet_scan_slots(M), CAPc, false, FTMc, false, Kc><<<grid_steady, 256, 0, t.stream>>>(               \
        t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now,    \
        t.ovf_next, split_on_v, nullptr, FMc, lr, un, vv)
    // (t.verlet_rows: the variant that rebuilds a few failing rows inside the workgroup — grids resident all at once;
    //  t.verlet_big_tile: denser clouds, whose blocks' halos outgrow the 1920-candidate tile but not the 2240-candidate
    //  one — three workgroups per CU instead of four, which costs the timed windows 2-3 % where the small tile does:
    //  kCapVerlet stays the default)
#define PPCR_FAST_V(FTMc, FMc)                                                                                          \
    do {                                                                                                                \
        if (t.verlet_big_tile) PPCR_FAST_V1(FTMc, FMc, 1, CAP);                                                         \
        else if (t.verlet_rows) PPCR_FAST_V1(FTMc, FMc, 2, kCapVerlet);                                                 \
        else PPCR_FAST_V1(FTMc, FMc, 1, kCapVerlet);                                                                    \
    } while (0)
    t.fused = false;
    int ftm = -2;  // model folded into this launch (-2: none)
    bool steady = false;
    if (multi) {
        // multi-level search: every block picks its level of the grid in the kernel; no split table (a block whose halo
        // does not fit at its level leaves its rows to nn_wide_kernel), nothing folded in
        // (the two-workgroups-per-block grid: the second workgroup only works for blocks marked split, see the kernel)
        const SplitTable two_per_block{nullptr, nullptr, nullptr, nullptr, 0, INT_MAX, 1};
        const int grid_multi = steady_grid(nb, true);
        bool done = false;
        if constexpr (M == 10) {  // diagnostic builds (option stamps): per-phase cycle counts of the multi-level kernel
            if (st && t.dm2_in && t.short_lists) {
                nn_fast_kernel<M, 16, kCapSteady, true, -2, true><<<grid_multi, 256, 0, t.stream>>>(
                    t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                    two_per_block, st, fm_none, lr, un, vv_none);
                done = true;
            }
        }
        if constexpr (M == 10 || M == 20) {
            if (st && !done) {
                nn_fast_kernel<M, C, CAP, true, -2, true><<<grid_multi, 256, 0, t.stream>>>(
                    t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                    two_per_block, st, fm_none, lr, un, vv_none);
                done = true;
            }
        }
        // (the mid-width steady-state variant — 28 slots, four workgroups per CU — was measured here as well: rows that search
        //  a level's whole radius accept more than 28 candidates more often, overflow twice and go to nn_wide_kernel: 3.92 k
        //  against 4.07 k it/s on the LiDAR-like scene, 4.13 k against 4.44 k on the slab.  Multi-level searches keep 32 slots.)
        // Verlet lists in a multi-level search: every level's scan builds the lists of the rows whose reach its stencil covers
        // (in base positions: to_base), nn_wide_kernel those of the rows it searches; a block whose rows' lists all hold —
        // or all but a few: see the kernel — picks no level at all.  (No dispatch order: the two-workgroups-per-block grid
        // is not the one the order's buffers were sized for.)
        if constexpr (M <= 24) {
            if (!done && !st && t.verlet_mode != 0 && t.dm2_in && t.short_lists) {
                VerletLists vv = t.verlet;
                vv.build_all = t.verlet_mode == 2 ? 0 : 1;
                vv.order_now = nullptr, vv.order_next = nullptr, vv.count_now = nullptr, vv.count_next = nullptr, vv.count_clear = nullptr;
                constexpr int CAPV = M <= 12 ? kCapVerlet : CAP;
                nn_fast_kernel<M, verlet_scan_slots(M), CAPV, false, -2, true, 1><<<grid_multi, 256, 0, t.stream>>>(
                    t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                    two_per_block, nullptr, fm_none, lr, un, vv);
                t.verlet_built = true;
                done = true;
            }
        }
        if constexpr (M <= 12) {
            if (!done && t.dm2_in && t.short_lists) {
                nn_fast_kernel<M, 16, kCapSteady, false, -2, true><<<grid_multi, 256, 0, t.stream>>>(
                    t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                    two_per_block, nullptr, fm_none, lr, un, vv_none);
                done = true;
            }
        }
        // (32-slot lists with the steady-state halo capacity — 39.4 KB, four workgroups per CU, one residency round for a
        //  200k-point cloud instead of two — were measured: K1 145 -> 128 us on the LiDAR-like scene, 118 -> 92 us on the
        //  slab, but 30-80 % more rows in nn_wide_kernel (more blocks outgrow the smaller tile and cascade): 4.09 k against
        //  3.99 k and 4.17 k against 4.47 k it/s.  The large tile stays.)
        constexpr int CAPM = CAP;
        if (!done)
            nn_fast_kernel<M, C, CAPM, false, -2, true><<<grid_multi, 256, 0, t.stream>>>(t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr,
                                                                                         t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now,
                                                                                         t.ovf_next, two_per_block, nullptr, fm_none, lr, un, vv_none);
        steady = true;  // (skips the single-level launches below)
    } else
    if constexpr (M <= 12) {
        // steady state: the temporal cut-off keeps every list near m entries, so half the list capacity does (an
        // overflowing lane tightens its threshold and scans again)
        if (t.dm2_in && t.short_lists) {
            steady = true;
            if (t.fuse && !st) ftm = t.fuse_tm;
            // Verlet lists (t.verlet_mode: 1 build them in this launch, 2 verify first and rebuild where needed)
            if (t.verlet_mode != 0 && !st && !halves) {
                VerletLists vv = t.verlet;
                vv.build_all = t.verlet_mode == 2 ? 0 : 1;
                const SplitTable split_on_v{t.split_flag, t.split_list, t.split_state, t.split_state + 1, kMaxSplit, (t.verlet_big_tile ? CAP : kCapVerlet) * 15 / 16, 0};
                if (ftm == 0) PPCR_FAST_V(0, *t.fuse);
                else if (ftm == 8) PPCR_FAST_V(8, *t.fuse);
                else if (ftm == -3) PPCR_FAST_V(-3, *t.fuse);
                else PPCR_FAST_V(-2, fm_none);
                t.verlet_built = true;
            } else
            if constexpr (M == 10) {
                if (st) PPCR_FAST(16, true, -2, fm_none);
                else if (ftm == 0) PPCR_FAST(16, false, 0, *t.fuse);
                else if (ftm == 8) PPCR_FAST(16, false, 8, *t.fuse);
                else if (ftm == -3) PPCR_FAST(16, false, -3, *t.fuse);
                else PPCR_FAST(16, false, -2, fm_none);
            } else {
                if (ftm == 0) PPCR_FAST(16, false, 0, *t.fuse);
                else if (ftm == 8) PPCR_FAST(16, false, 8, *t.fuse);
                else if (ftm == -3) PPCR_FAST(16, false, -3, *t.fuse);
                else PPCR_FAST(16, false, -2, fm_none);
            }
            t.fused = ftm != -2;
        }
    } else if constexpr (kMid) {
        if (t.dm2_in && t.short_lists && !halves && t.short_count != nullptr) {
            steady = true;
            if (t.verlet_mode != 0 && !st) {
                // Verlet lists (32 slots): rows are answered from their lists where those still hold; the rows this launch
                // leaves unanswered — short rows of a two-pass search, a few failing rows of an answering workgroup — are
                // searched AND given fresh lists by nn_wide_kernel below
                VerletLists vv = t.verlet;
                vv.build_all = t.verlet_mode == 2 ? 0 : 1;
                const SplitTable split_mid_v{t.split_flag, t.split_list, t.split_state, t.split_state + 1, kMaxSplit, kCapVerletMid * 15 / 16, 0};
                if (t.verlet_mode == 1) {
                    // the launch that BUILDS every list scans with the lists' reach in place of the rows' cut-offs: wider x
                    // windows, larger halos — in the 1600-candidate tile a fifth of a 200k cloud's blocks were handed over
                    // there (43 k rows for nn_wide_kernel, 120-200 us).  It runs once per registration: the 2240-candidate
                    // tile, three workgroups per CU.
                    nn_fast_kernel<M, verlet_scan_slots(M), CAP, false, -2, false, 1><<<nb + kMaxSplit, 256, 0, t.stream>>>(
                        t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                        split_mid_v, nullptr, fm_none, lr, un, vv);
                } else {
                    nn_fast_kernel<M, verlet_scan_slots(M), kCapVerletMid, false, -2, false, 1><<<nb + kMaxSplit, 256, 0, t.stream>>>(
                        t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                        split_mid_v, nullptr, fm_none, lr, un, vv);
                }
                t.verlet_built = true;
            } else {
                const SplitTable split_mid{t.split_flag, t.split_list, t.split_state, t.split_state + 1, kMaxSplit, CAP2 * 15 / 16, 0};
                nn_fast_kernel<M, C2, CAP2, false, -2, false><<<nb + kMaxSplit, 256, 0, t.stream>>>(
                    t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.pm, t.dm2, t.dm2_in, t.ovf_list, t.ovf_now, t.ovf_next,
                    split_mid, nullptr, fm_none, lr, un, vv_none);
            }
        }
    }
    if (!steady && !multi) {
        if constexpr (M == 10) {
            if (st) PPCR_FAST(C, true, -2, fm_none);
            else PPCR_FAST(C, false, -2, fm_none);
        } else {
            PPCR_FAST(C, false, -2, fm_none);
        }
    }
#undef PPCR_FAST
#undef PPCR_FAST_V
#undef PPCR_FAST_V1
    if (t.short_count == nullptr && t.between) t.between(t.between_arg);
    if (t.short_count != nullptr) {
        if (t.between2) t.between2(t.between_arg);
        // the rows the tiled kernel did not answer — those of handed-over workgroups (marked unsearched) and, in a two-pass
        // search, those that came back short — were listed by the kernel itself; they are searched one row per wave, in
        // place of the cleanup launch (nothing is folded in this mode).  A fixed grid of waves strides over the list: its
        // length is only known on the device.
        t.merged = false;
#ifndef PPCR_WIDE_PER_CU
#define PPCR_WIDE_PER_CU 6
#endif
        constexpr int kWideGrid = 256 * PPCR_WIDE_PER_CU;  // six workgroups per CU are resident (74 VGPRs, 22 KB of LDS): 6144 waves, a row each
        // (with Verlet lists in use the rows searched here get theirs; otherwise vl is null and the kernel is the plain search)
        VerletLists vv_wide = t.verlet_built ? t.verlet : vv_none;
        nn_wide_kernel<M><<<kWideGrid, 256, 0, t.stream>>>(t.src, t.ns, t.tgt, t.cell_start, t.grid, t.reach, t.r2, t.r2_full, t.m, t.nbr,
                                                            t.cnt, t.dm2, t.short_list, t.short_count, t.short_seen, t.loop_st, vv_wide);
        return;
    }
    // persistent workgroups over the list: few when the last association this handle heard from handed nothing over
    // (a handful of hand-overs — a registration that still moves a few per cent of the radius per iteration leaves one to three
    //  per launch — do not need 512 workgroups that read a counter and leave: four per workgroup the last launch handed over)
    const int cleanup_grid = t.quiet ? std::min(nb, 32)
                                     : std::min(nb, (t.handed_last != ~0u && t.handed_last < 128u) ? std::max(32, 4 * (int)t.handed_last) : 512);
    const int n_extra = steady ? (halves ? -1 : kMaxSplit) : 0;
    FoldSolve fs_none;
    std::memset(&fs_none, 0, sizeof(fs_none));
    fs_none.loop.st = t.loop_st;  // the cleanup role steps aside with everybody else
    const bool merge = ftm != -2 && t.fold != nullptr;
    t.merged = merge;
    FoldSolve fold_now = t.fold ? *t.fold : fs_none;
    fold_now.handed_over = t.ovf_now;  // this launch's counter (the caller toggled the pair after the fold was prepared)
#define PPCR_CLEANUP(FTMc, FMc, MERGEc, FSc)                                                                           \
    nn_tile_cleanup_kernel<M, CC, 256, CAP, FTMc, MERGEc><<<cleanup_grid + (MERGEc ? kNSums : 0), 256, 0, t.stream>>>(  \
        t.src, t.ns, t.tgt, t.cell_start, t.grid, t.r2, t.m, t.nbr, t.cnt, t.dm2, t.ovf_list, t.ovf_now, t.split_list,  \
        n_extra, FMc, FSc)
    if (ftm == 0 && merge) PPCR_CLEANUP(0, *t.fuse, true, fold_now);
    else if (ftm == 8 && merge) PPCR_CLEANUP(8, *t.fuse, true, fold_now);
    else if (ftm == -3 && merge) PPCR_CLEANUP(-3, *t.fuse, true, fold_now);
    else if (ftm == 0) PPCR_CLEANUP(0, *t.fuse, false, fs_none);
    else if (ftm == 8) PPCR_CLEANUP(8, *t.fuse, false, fs_none);
    else if (ftm == -3) PPCR_CLEANUP(-3, *t.fuse, false, fs_none);
    else PPCR_CLEANUP(-2, fm_none, false, fs_none);
#undef PPCR_CLEANUP
}

}  // namespace

#define PPCR_CAT2(a, b) a##b
#define PPCR_CAT(a, b) PPCR_CAT2(a, b)
void PPCR_CAT(launch_tile_m, PPCR_TILE_M)(TileLaunch &t) { launch_tile<PPCR_TILE_M>(t); }

}  // namespace ppcr
