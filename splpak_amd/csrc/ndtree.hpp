// Nested-dissection elimination tree of the node grid (host side; built once per plan).
//
// The normal equations N of the fit couple two nodes iff their grid indices differ by at most 3 in every
// dimension (7^d stencil: windows of 4 basis functions per dimension, src/splpak.F90:821-827 and the column
// order of :657-666).  The grid is structured, so no symbolic phase is needed: a box of nodes is bisected
// along its longest dimension by a SEPARATOR SLAB three nodes thick -- the two halves then share no entry of
// N -- and the halves are bisected again until a box is small.  Eliminating the boxes bottom-up (leaves, then
// the separators that join them) is a Cholesky factorisation of N in the nested-dissection order; its fill
// stays inside "fronts":
//
//   front of a tree node = [ own variables | border ],   own    = the separator slab (the whole box of a leaf),
//                                                        border = the nodes within 3 of the node's REGION (the box
//                                                                 its subtree covers) that lie outside it
//
// Every border node belongs to the separator of an ancestor, and the border of a child is contained in
// own + border of its parent, so the Schur complement a front leaves behind is added into its parent's front
// (multifrontal method).  All index sets are ordered by elimination position, which makes every child -> parent
// map monotone: lower triangles map to lower triangles.
//
// For 64^3 nodes this needs ~1e13 flop and ~10 GB of factor instead of the band's 4.1e13 / 26.9 GB; for the
// 4-D grids the gap is an order of magnitude.
#pragma once
#include "common.hpp"
#include <cstdlib>
#include <string>
#include <vector>

namespace splpak {

struct NdFront {
    int lo[MAXD], hi[MAXD];        // region: the box of nodes the subtree covers, [lo, hi) per internal dimension
    int olo[MAXD], ohi[MAXD];      // own variables: the separator slab, or the whole box of a leaf
    int parent = -1, child[2] = {-1, -1};
    int slot = 0;                  // 0 / 1: first or second child of its parent
    int depth = 0;                 // root = 0
    int w = 0, wp = 0;             // own variables; padded to a multiple of 256 (identity on the padding)
    int h = 0, hp = 0;             // border variables; padded to a multiple of 64 (zero rows)
    int fp = 0;                    // rows of the front = wp + hp
    int own0 = 0;                  // elimination position of the first own variable
    long long ld = 0, lds = 0;     // leading dimensions of the panel (fp x wp) and of the Schur buffer (hp x hp)
    long long panel_off = 0;       // doubles into the factor arena
    long long s_off = 0;           // doubles into the Schur arena of the front's depth parity
    long long rofs = 0;            // offset into ownvar (wp entries)
    long long bofs = 0;            // offset into bvar / bpos / pmap (hp entries)
    long long vofs = 0;            // offset of the front's local vector (fp entries)
    int blk0 = 0;                  // index of its first 256x256 diagonal block (inverse storage)
    int nsteps = 0;                // wp / 256
};

struct NdTree {
    Grid g{};
    std::vector<NdFront> fr;       // postorder: children before parents, the root last
    int root = -1, maxdepth = 0;
    std::vector<int> pos;          // [ncol] elimination position of node (natural internal column index)
    std::vector<int> front_of;     // [ncol] front that owns the node
    std::vector<int> ownvar;       // [sum wp] node of every own row, -1 on the padding
    std::vector<int> bvar;         // [sum hp] node of every border row (-1 padding)
    std::vector<int> bpos;         // [sum hp] its elimination position, ascending (INT_MAX on the padding)
    std::vector<int> pmap;         // [sum hp] row of the PARENT's front the border row maps to (-1 padding)
    std::vector<std::vector<int>> by_depth;
    long long factor_doubles = 0;  // sum ld * wp
    long long s_doubles[2] = {0, 0};   // Schur arenas by depth parity (max over the depths of that parity)
    long long s_total = 0;         // sum over all fronts (what one fit zeroes and streams)
    long long vec_doubles = 0;     // sum fp
    long long own_rows = 0, border_rows = 0;
    int nblocks = 0;               // sum nsteps
    double flop = 0.0;             // 2 * 64^2 * 256 per trailing-update item, all launches
    double flop_exact = 0.0;       // without the padding: w^3/3 + w^2 h + w h^2 per front
};

// ELIMINATION SCHEDULE and the Schur-buffer arena (round 5).
//
// A STAGE is a set of fronts of one tree depth that are eliminated together (every launch of the factorisation is a batch over
// the fronts of a stage).  Rounds 3-4 knew one schedule: a stage per tree depth, deepest first, and two Schur arenas by depth
// parity, each as large as the largest depth of its parity -- at a 4-D grid every depth of the tree holds 50-70 % of all the
// border x border buffers of the top of the tree at once (24^4: 132 GB of arenas for 76 GB of factor; 32^4: 714 GB).
// Now the fronts of depth < cut are stages of their own, visited in POSTORDER, and the subtrees below depth cut are
// eliminated one after the other, each level by level as before: at any time only the buffers of ONE root-to-leaf path of the
// top of the tree plus two levels of one subtree are alive.  cut = 0 is the old order.
//
// A front's Schur buffer is alive from the start of the first stage that adds into it (its first child's stage: the
// extend-add is fused into the child's last pass) to the end of its own stage; the arena offsets come from a first-fit
// interval allocation over the stage sequence, done once per plan on the host.  All accesses to the arena are ordered by the
// update stream (Schur passes, fused extend-adds, zeroing), so reuse needs no further synchronisation.
//
// `packed`: a Schur buffer holds only the 64-column tile columns of its LOWER TRIANGLE, each from its diagonal tile down
// (tile column c: 64 columns of leading dimension L - 64 c, L = hp + 16) -- half the bytes of the square buffer.
struct NdStage {
    std::vector<int> ids;          // its fronts (ascending id)
    int depth = 0;
    int dep = -1;                  // last stage whose fronts add into this stage's fronts (-1: none)
    int first = 0;                 // stage at whose start this stage's buffers are allocated and zeroed (its first child stage, or itself)
    long long off = 0, doubles = 0;    // its block of the arena
    bool keep = false;             // never freed (subtree roots of a multi-GPU fit: other GPUs pull them after the stage)
};
struct NdSchedule {
    int cut = 0;
    bool packed = false;
    std::vector<NdStage> st;       // in execution order
    std::vector<int> stage_of;     // [front] (-1: not eliminated by this schedule)
    std::vector<long long> soff;   // [front] doubles into the arena (-1: the front has no buffer)
    long long arena = 0;           // doubles: peak of the allocation
    long long total = 0;           // doubles of all buffers (what one fit zeroes)
};
// doubles of the Schur buffer of a front; leading dimension argument of the kernels: lds, or -(hp + 16) for the packed form
inline long long nd_schur_doubles(const NdFront &f, bool packed)
{
    if (f.hp <= 0) return 0;
    const long long nt = f.hp / 64, L = f.hp + 16;
    return packed ? 64 * L * nt - 2048 * nt * (nt - 1) : f.lds * (long long)f.hp;
}
inline long long nd_schur_ld(const NdFront &f, bool packed) { return packed ? -(long long)(f.hp + 16) : f.lds; }
// mine: fronts this schedule eliminates (NULL: all); needs: fronts whose buffer is materialised (NULL: every front with a
// border); dlow: fronts above this depth are left out (multi-GPU: the top phase has them), and those AT dlow > 0 are kept
// halves > 0 (cut = 0 only): the depths 1 .. halves as two half-stages each, by the root's subtrees, interleaved (see nd_schedule)
void nd_schedule(const NdTree &t, int cut, bool packed, const std::vector<char> *mine, const std::vector<char> *needs, int dlow, NdSchedule &sc, int halves = 0);

// Distribution of the tree over the R GPUs of a one-process multi-GPU fit (round 4; ndchol.hip "top phase"):
//   * the subtrees below tree depth dcut = ceil(log2 R) are dealt to the ranks (subtree i of the depth-dcut fronts -> rank
//     i mod R): a rank stores and eliminates ITS subtrees only (panels, Schur buffers, block inverses);
//   * the fronts above (depth < dcut: "top fronts") are DISTRIBUTED BY BLOCK COLUMNS: a top front is one square lower
//     triangular matrix of fp = wp + hp rows (own | border), cut into block columns of 256; block column J belongs to rank
//     (J / chunk) mod R and is stored from its diagonal block down (rows J*256 .. fp-1, leading dimension fp - J*256 + 16).
//     The first wp / 256 block columns are eliminated, the others receive the Schur complement.
struct NdPartition {
    int R = 1, chunk = 1, dcut = 0;
    std::vector<int> owner;            // [front] rank of its subtree (depth >= dcut); -1: top front
    std::vector<int> top;              // top fronts in elimination order (depth dcut-1 .. 0, ascending id inside a depth)
    std::vector<int> top_index;        // [front] index into top, or -1
    std::vector<int> seq0;             // [top index] first step of the front in the global sequence of top block steps
    int nseq = 0;                      // steps of all top fronts
    long long max_panel = 0;           // doubles of the largest solved panel that travels (receive buffers)
    // bytes per rank
    std::vector<double> b_panels, b_schur, b_top, b_inv, b_recv, b_vec, b_total;
    std::vector<double> flop_sub, flop_top;      // padded flop of the rank's subtrees / of its share of the top fronts
};
inline int top_nblocks(const NdFront &f) { return (f.fp + 255) / 256; }
inline int top_block_cols(const NdFront &f, int J) { const int c = f.fp - J * 256; return c < 256 ? c : 256; }
inline long long top_block_ld(const NdFront &f, int J) { return (long long)(f.fp - J * 256) + 16; }
inline int top_owner(const NdPartition &pt, int J) { return (J / pt.chunk) % pt.R; }
// R >= 1; chunk < 1: 1.  dcut is clamped to the tree's depth (more ranks than subtrees: some own no subtree)
void nd_partition(const NdTree &t, int R, int chunk, NdPartition &pt);

// boxes whose largest extent reaches this are bisected (SPLPAK_ND_SPLIT overrides).  3-D / 4-D: 8 (leaves of 5 .. 7 nodes
// per dimension: deeper trees cost flops nowhere, shallower ones +25 % at 64^3).  2-D grids are launch bound, not flop
// bound: leaves of up to 15 x 15 nodes (one 256-column block) save two tree levels -- 64^2 / 1e6 points (BASELINE config 2)
// 2.86 -> 2.46 ms per fit.
inline int nd_default_split_min(int ndim = 3)
{
    if (const char *e = splpak::opt_get("SPLPAK_ND_SPLIT")) return atoi(e);
    return ndim <= 2 ? 16 : 8;
}
// split_min: a box whose largest extent is at least this is bisected (>= 5); returns false on inconsistency
bool nd_build(const Grid &g, NdTree &t, int split_min);
// invariants of the tree (every node owned once, borders inside the parent's rows, monotone maps); "" = fine
std::string nd_check(const NdTree &t);

}  // namespace splpak
