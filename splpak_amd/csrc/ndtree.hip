// Host side of the nested-dissection factorisation: the elimination tree of the node grid (ndtree.hpp).
// Pure host code (no device calls): it is also what the CPU test tier exercises through
// splpak_debug_nd_tree (include/splpak_hip.h).
#include "ndtree.hpp"
#include "plan.hpp"
#include <algorithm>
#include <climits>
#include <cstring>
#include <functional>

namespace splpak {

namespace {

inline int round_up(int v, int m) { return ((v + m - 1) / m) * m; }

struct Builder {
    NdTree &t;
    int split_min;
    int counter = 0;

    // visits the nodes of the box [lo, hi) in natural order (dimension 0 fastest)
    template <typename F>
    void for_box(const int *lo, const int *hi, F &&f) const
    {
        const Grid &g = t.g;
        int i[MAXD];
        for (int d = 0; d < MAXD; ++d) i[d] = d < g.ndim ? lo[d] : 0;
        for (int d = 0; d < g.ndim; ++d)
            if (hi[d] <= lo[d]) return;
        for (;;) {
            int col = 0;
            for (int d = 0; d < g.ndim; ++d) col += i[d] * g.colstride[d];
            f(col, i);
            int d = 0;
            while (d < g.ndim) {
                if (++i[d] < hi[d]) break;
                i[d] = lo[d];
                ++d;
            }
            if (d == g.ndim) break;
        }
    }

    int build(const int *lo, const int *hi, int depth)
    {
        const Grid &g = t.g;
        int k = 0, e = 0;
        for (int d = 0; d < g.ndim; ++d)
            if (hi[d] - lo[d] > e) { e = hi[d] - lo[d]; k = d; }
        NdFront f;
        for (int d = 0; d < MAXD; ++d) {
            f.lo[d] = d < g.ndim ? lo[d] : 0;
            f.hi[d] = d < g.ndim ? hi[d] : 1;
            f.olo[d] = f.lo[d];
            f.ohi[d] = f.hi[d];
        }
        f.depth = depth;
        int c0 = -1, c1 = -1;
        if (e >= split_min) {
            const int s = lo[k] + (e - 3) / 2;            // separator = [s, s+3) along dimension k
            int lhi[MAXD], rlo[MAXD];
            for (int d = 0; d < g.ndim; ++d) { lhi[d] = hi[d]; rlo[d] = lo[d]; }
            lhi[k] = s;
            rlo[k] = s + 3;
            c0 = build(lo, lhi, depth + 1);
            c1 = build(rlo, hi, depth + 1);
            f.olo[k] = s;
            f.ohi[k] = s + 3;
        }
        const int id = (int)t.fr.size();
        f.child[0] = c0;
        f.child[1] = c1;
        if (c0 >= 0) { t.fr[(size_t)c0].parent = id; t.fr[(size_t)c0].slot = 0; }
        if (c1 >= 0) { t.fr[(size_t)c1].parent = id; t.fr[(size_t)c1].slot = 1; }
        f.own0 = counter;
        for_box(f.olo, f.ohi, [&](int col, const int *) {
            t.pos[(size_t)col] = counter++;
            t.front_of[(size_t)col] = id;
        });
        f.w = counter - f.own0;
        t.fr.push_back(f);
        if (depth > t.maxdepth) t.maxdepth = depth;
        return id;
    }
};

}  // namespace

bool nd_build(const Grid &g, NdTree &t, int split_min)
{
    t = NdTree();
    t.g = g;
    if (split_min < 5) split_min = 5;
    t.pos.assign((size_t)g.ncol, -1);
    t.front_of.assign((size_t)g.ncol, -1);
    Builder b{t, split_min};
    int lo[MAXD] = {0, 0, 0, 0}, hi[MAXD] = {1, 1, 1, 1};
    for (int d = 0; d < g.ndim; ++d) hi[d] = g.nodes[d];
    t.root = b.build(lo, hi, 0);
    if (b.counter != g.ncol) return false;
    std::vector<int> var_of_pos((size_t)g.ncol);
    for (int c = 0; c < g.ncol; ++c) {
        if (t.pos[(size_t)c] < 0) return false;
        var_of_pos[(size_t)t.pos[(size_t)c]] = c;
    }
    // sizes, offsets, own rows
    t.by_depth.assign((size_t)t.maxdepth + 1, {});
    std::vector<long long> s_depth((size_t)t.maxdepth + 1, 0);
    std::vector<int> tmp;
    for (size_t id = 0; id < t.fr.size(); ++id) {
        NdFront &f = t.fr[id];
        // border: nodes within 3 of the region, outside it; all of them are eliminated later than the region
        int elo[MAXD], ehi[MAXD];
        for (int d = 0; d < MAXD; ++d) {
            elo[d] = d < g.ndim ? std::max(0, f.lo[d] - 3) : 0;
            ehi[d] = d < g.ndim ? std::min(g.nodes[d], f.hi[d] + 3) : 1;
        }
        tmp.clear();
        b.for_box(elo, ehi, [&](int col, const int *i) {
            bool inside = true;
            for (int d = 0; d < g.ndim; ++d) inside = inside && i[d] >= f.lo[d] && i[d] < f.hi[d];
            if (!inside) tmp.push_back(t.pos[(size_t)col]);
        });
        std::sort(tmp.begin(), tmp.end());
        f.h = (int)tmp.size();
        f.wp = round_up(f.w, 256);
        f.hp = round_up(f.h, 64);
        f.fp = f.wp + f.hp;
        f.ld = f.fp + 16;                       // multiple of 16 doubles, odd multiple of 128 B (fp is a multiple of 64)
        f.lds = f.hp > 0 ? f.hp + 16 : 0;
        f.nsteps = f.wp / 256;
        f.panel_off = t.factor_doubles;
        t.factor_doubles += f.ld * (long long)f.wp;
        f.rofs = (long long)t.ownvar.size();
        f.bofs = (long long)t.bpos.size();
        f.vofs = t.vec_doubles;
        t.vec_doubles += f.fp;
        f.blk0 = t.nblocks;
        t.nblocks += f.nsteps;
        t.own_rows += f.wp;
        t.border_rows += f.hp;
        for (int r = 0; r < f.wp; ++r) t.ownvar.push_back(r < f.w ? var_of_pos[(size_t)(f.own0 + r)] : -1);
        for (int i = 0; i < f.hp; ++i) {
            t.bpos.push_back(i < f.h ? tmp[(size_t)i] : INT_MAX);
            t.bvar.push_back(i < f.h ? var_of_pos[(size_t)tmp[(size_t)i]] : -1);
        }
        f.s_off = s_depth[(size_t)f.depth];
        s_depth[(size_t)f.depth] += f.lds * (long long)f.hp;
        t.s_total += f.lds * (long long)f.hp;
        t.by_depth[(size_t)f.depth].push_back((int)id);
        const double w = f.w, h = f.h;
        t.flop_exact += w * w * w / 3.0 + w * w * h + w * h * h;
        // trailing-update items of the blocked factorisation (64x64x256 each): panel columns right of step k and the Schur buffer
        for (int k = 0; k < f.nsteps; ++k) {
            const long long nc = (f.wp - (k + 1) * 256) / 64, nr = (f.fp - (k + 1) * 256) / 64, ns = f.hp / 64;
            t.flop += 2.0 * 64 * 64 * 256 * (double)(nc * nr - nc * (nc - 1) / 2 + ns * (ns + 1) / 2);
        }
    }
    for (int d = 0; d <= t.maxdepth; ++d) t.s_doubles[d & 1] = std::max(t.s_doubles[d & 1], s_depth[(size_t)d]);
    // child border row -> parent row
    t.pmap.assign(t.bpos.size(), -1);
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        if (f.parent < 0) continue;
        const NdFront &p = t.fr[(size_t)f.parent];
        const int *pb = t.bpos.data() + p.bofs;
        for (int i = 0; i < f.h; ++i) {
            const int ps = t.bpos[(size_t)(f.bofs + i)];
            int row = -1;
            if (ps >= p.own0 && ps < p.own0 + p.w) row = ps - p.own0;
            else {
                const int *it = std::lower_bound(pb, pb + p.h, ps);
                if (it == pb + p.h || *it != ps) return false;
                row = p.wp + (int)(it - pb);
            }
            t.pmap[(size_t)(f.bofs + i)] = row;
        }
    }
    return true;
}

void nd_schedule(const NdTree &t, int cut, bool packed, const std::vector<char> *mine, const std::vector<char> *needs, int dlow, NdSchedule &sc, int halves)
{
    sc = NdSchedule();
    sc.cut = cut < 0 ? 0 : cut;
    sc.packed = packed;
    const size_t nf = t.fr.size();
    sc.stage_of.assign(nf, -1);
    sc.soff.assign(nf, -1);
    auto is_mine = [&](int id) { return t.fr[(size_t)id].depth >= dlow && (!mine || (*mine)[(size_t)id]); };
    auto add_stage = [&](std::vector<int> &ids, int depth) {
        if (ids.empty()) return;
        std::sort(ids.begin(), ids.end());
        NdStage st;
        st.depth = depth;
        st.ids.swap(ids);
        for (int id : st.ids) sc.stage_of[(size_t)id] = (int)sc.st.size();
        sc.st.push_back(std::move(st));
    };
    // level by level through the subtree of `r` (children before parents; ids ascend in postorder, so a front's descendants
    // are the contiguous range of ids before it -- walked explicitly here to stay independent of that)
    auto emit_subtree = [&](int r) {
        std::vector<std::vector<int>> lev((size_t)t.maxdepth + 1);
        std::vector<int> stack{r};
        while (!stack.empty()) {
            const int id = stack.back();
            stack.pop_back();
            const NdFront &f = t.fr[(size_t)id];
            if (is_mine(id)) lev[(size_t)f.depth].push_back(id);
            for (int c : f.child) if (c >= 0) stack.push_back(c);
        }
        // `halves` (level-by-level order of the whole tree only): the depths 1 .. halves are split by the root's two subtrees and
        // the halves interleaved -- left d, right d, left d - 1, right d - 1, ... -- so that the chain of a half-stage, which waits
        // for the last Schur passes of ITS children only, starts beside the passes of the other half of the depth below instead of
        // behind them (a whole stage ends in one batched last pass, and the chip then waited for the first block steps of the
        // next stage's chain: 25 of 226 ms at 64^3).  A front's two children stay in one half, in their order: same bits.
        if (halves > 0 && r == t.root && sc.cut == 0 && t.fr[(size_t)r].child[0] >= 0 && t.fr[(size_t)r].child[1] >= 0) {
            std::vector<char> right(t.fr.size(), 0);
            std::vector<int> st2{t.fr[(size_t)r].child[1]};
            while (!st2.empty()) {
                const int id = st2.back();
                st2.pop_back();
                right[(size_t)id] = 1;
                for (int c : t.fr[(size_t)id].child) if (c >= 0) st2.push_back(c);
            }
            for (int d = t.maxdepth; d >= 0; --d) {
                if (d < 1 || d > halves) { add_stage(lev[(size_t)d], d); continue; }
                std::vector<int> a, b;
                for (int id : lev[(size_t)d]) (right[(size_t)id] ? b : a).push_back(id);
                add_stage(a, d);
                add_stage(b, d);
            }
            return;
        }
        for (int d = t.maxdepth; d >= 0; --d) add_stage(lev[(size_t)d], d);
    };
    struct Rec { static void go(const NdTree &t, int id, int cut, const std::function<void(int)> &sub, const std::function<void(int)> &one) {
        const NdFront &f = t.fr[(size_t)id];
        if (f.depth >= cut) { sub(id); return; }
        for (int c : f.child) if (c >= 0) go(t, c, cut, sub, one);
        one(id);
    } };
    if (t.root >= 0)
        Rec::go(t, t.root, sc.cut, emit_subtree, [&](int id) { std::vector<int> one; if (is_mine(id)) one.push_back(id); add_stage(one, t.fr[(size_t)id].depth); });
    const int ns = (int)sc.st.size();
    for (int i = 0; i < ns; ++i) {
        NdStage &st = sc.st[(size_t)i];
        st.first = i;
        for (int id : st.ids)
            for (int c : t.fr[(size_t)id].child) {
                const int cs = c >= 0 ? sc.stage_of[(size_t)c] : -1;
                if (cs < 0) continue;
                st.dep = std::max(st.dep, cs);
                st.first = std::min(st.first, cs);
            }
        st.keep = dlow > 0 && st.depth == dlow;
        for (int id : st.ids) {
            const NdFront &f = t.fr[(size_t)id];
            if (f.hp <= 0 || (needs && !(*needs)[(size_t)id])) continue;
            sc.soff[(size_t)id] = st.doubles;               // relative to the block, for now
            st.doubles += (nd_schur_doubles(f, packed) + 63) / 64 * 64;
        }
        sc.total += st.doubles;
    }
    // Interval allocation over the stage sequence (offline: the lifetimes are known).  Three placements are tried and the
    // smallest arena kept: first fit from the bottom; best fit; and "two-ended" -- blocks of even tree depth from the bottom,
    // of odd depth from the top of a trial arena -- which is exact for the level-by-level order (the two arenas of rounds 3-4).
    std::vector<std::vector<int>> starts((size_t)ns);
    for (int i = 0; i < ns; ++i) starts[(size_t)sc.st[(size_t)i].first].push_back(i);
    for (int i = 0; i < ns; ++i) std::stable_sort(starts[(size_t)i].begin(), starts[(size_t)i].end(), [&](int a, int b) { return a > b; });   // (longest-lived first)
    auto place = [&](int mode, long long cap, std::vector<long long> &off) -> long long {
        std::vector<std::pair<long long, long long>> freel{{0, cap}};      // (offset, length), ascending offsets
        auto take = [&](long long n, bool high) {
            long long best = -1;
            size_t bk = 0;
            for (size_t k = 0; k < freel.size(); ++k) {
                if (freel[k].second < n) continue;
                if (mode == 1) { if (best < 0 || freel[k].second < freel[bk].second) { best = 0; bk = k; } continue; }
                best = 0; bk = k;
                if (!high) break;                                       // first fit from the bottom; `high`: the last fit
            }
            if (best < 0) return (long long)-1;
            long long o;
            if (high) { o = freel[bk].first + freel[bk].second - n; freel[bk].second -= n; }
            else { o = freel[bk].first; freel[bk].first += n; freel[bk].second -= n; }
            if (freel[bk].second == 0) freel.erase(freel.begin() + (long)bk);
            return o;
        };
        auto give = [&](long long o, long long n) {
            size_t k = 0;
            while (k < freel.size() && freel[k].first < o) ++k;
            freel.insert(freel.begin() + (long)k, {o, n});
            if (k + 1 < freel.size() && freel[k].first + freel[k].second == freel[k + 1].first) { freel[k].second += freel[k + 1].second; freel.erase(freel.begin() + (long)k + 1); }
            if (k > 0 && freel[k - 1].first + freel[k - 1].second == freel[k].first) { freel[k - 1].second += freel[k].second; freel.erase(freel.begin() + (long)k); }
        };
        long long lo_end = 0, hi_beg = cap;
        for (int i = 0; i < ns; ++i) {
            for (int x : starts[(size_t)i]) {
                const NdStage &st = sc.st[(size_t)x];
                if (st.doubles <= 0) continue;
                const bool high = mode == 2 && (st.depth & 1) != 0;
                const long long o = take(st.doubles, high);
                if (o < 0) return -1;
                off[(size_t)x] = o;
                if (high) hi_beg = std::min(hi_beg, o); else lo_end = std::max(lo_end, o + st.doubles);
            }
            const NdStage &me = sc.st[(size_t)i];
            if (me.doubles > 0 && !me.keep) give(off[(size_t)i], me.doubles);
        }
        if (mode != 2) return lo_end;
        return hi_beg >= lo_end ? lo_end + (cap - hi_beg) : -1;        // (the two ends must not have met)
    };
    std::vector<long long> off((size_t)ns, 0), best_off;
    long long best = -1;
    for (int mode = 0; mode < 2; ++mode) {
        const long long a = place(mode, (long long)1 << 60, off);
        if (a >= 0 && (best < 0 || a < best)) { best = a; best_off = off; }
    }
    if (best > 0) {
        // two-ended: bisect the smallest trial arena in which the two ends do not collide
        long long lo = 0, hi = best;
        std::vector<long long> o2((size_t)ns, 0);
        if (place(2, hi, o2) >= 0) {
            while (hi - lo > 4096) {
                const long long mid = (lo + (hi - lo) / 2 + 63) / 64 * 64;
                if (place(2, mid, off) >= 0) { hi = mid; o2 = off; }
                else lo = mid;
            }
            if (hi < best) { best = hi; best_off = o2; }
        }
    }
    sc.arena = best < 0 ? 0 : best;
    for (int i = 0; i < ns; ++i) sc.st[(size_t)i].off = best_off.empty() ? 0 : best_off[(size_t)i];
    for (int i = 0; i < ns; ++i)
        for (int id : sc.st[(size_t)i].ids)
            if (sc.soff[(size_t)id] >= 0) sc.soff[(size_t)id] += sc.st[(size_t)i].off;
}

void nd_partition(const NdTree &t, int R, int chunk, NdPartition &pt)
{
    pt = NdPartition();
    pt.R = R < 1 ? 1 : R;
    pt.chunk = chunk < 1 ? 1 : chunk;
    int dcut = 0;
    while ((1 << dcut) < pt.R) ++dcut;
    if (dcut > t.maxdepth) dcut = t.maxdepth;
    if (pt.R == 1) dcut = 0;
    pt.dcut = dcut;
    const size_t nf = t.fr.size();
    pt.owner.assign(nf, -1);
    pt.top_index.assign(nf, -1);
    // subtree i of the depth-dcut fronts -> rank i mod R.  A front above dcut that has no children (an unbalanced tree:
    // a branch that stops short) is a top front like the others: it is simply eliminated in the top phase.
    std::vector<int> slot(nf, -1);
    const std::vector<int> &cut = t.by_depth[(size_t)dcut];
    for (size_t i = 0; i < cut.size(); ++i) slot[(size_t)cut[i]] = (int)i;
    for (int id = (int)nf - 1; id >= 0; --id) {            // parents have larger ids than their children (postorder)
        const NdFront &f = t.fr[(size_t)id];
        if (f.depth > dcut) slot[(size_t)id] = slot[(size_t)f.parent];
        if (f.depth >= dcut) pt.owner[(size_t)id] = slot[(size_t)id] % pt.R;
    }
    for (int d = dcut - 1; d >= 0; --d)
        for (int id : t.by_depth[(size_t)d]) {
            pt.top_index[(size_t)id] = (int)pt.top.size();
            pt.top.push_back(id);
        }
    pt.seq0.assign(pt.top.size(), 0);
    for (size_t i = 0; i < pt.top.size(); ++i) {
        pt.seq0[i] = pt.nseq;
        pt.nseq += t.fr[(size_t)pt.top[i]].nsteps;
    }
    // bytes per rank
    const size_t Rn = (size_t)pt.R;
    for (auto *v : {&pt.b_panels, &pt.b_schur, &pt.b_top, &pt.b_inv, &pt.b_recv, &pt.b_vec, &pt.b_total, &pt.flop_sub, &pt.flop_top}) v->assign(Rn, 0.0);
    std::vector<std::vector<double>> sdepth(Rn, std::vector<double>((size_t)t.maxdepth + 1, 0.0));
    for (size_t id = 0; id < nf; ++id) {
        const NdFront &f = t.fr[id];
        const int o = pt.owner[id];
        if (o >= 0) {
            pt.b_panels[(size_t)o] += 8.0 * (double)f.ld * f.wp;
            sdepth[(size_t)o][(size_t)f.depth] += 8.0 * (double)f.lds * f.hp;
            pt.b_inv[(size_t)o] += (2.0 * 65536.0 + 4096.0) * 8.0 * f.nsteps;
            for (int k = 0; k < f.nsteps; ++k) {
                const long long nc = (f.wp - (k + 1) * 256) / 64, nr = (f.fp - (k + 1) * 256) / 64, ns = f.hp / 64;
                pt.flop_sub[(size_t)o] += 2.0 * 64 * 64 * 256 * (double)(nc * nr - nc * (nc - 1) / 2 + ns * (ns + 1) / 2);
            }
        } else {
            const int nb = top_nblocks(f);
            for (int J = 0; J < nb; ++J) {
                const int q = top_owner(pt, J);
                pt.b_top[(size_t)q] += 8.0 * (double)top_block_ld(f, J) * top_block_cols(f, J);
                if (J < f.nsteps) pt.b_inv[(size_t)q] += (2.0 * 65536.0 + 4096.0) * 8.0;
                // updates this block column receives: one 64-row-tile trapezoid per eliminated block column left of it
                const long long nr = (f.fp - J * 256) / 64, nc = (top_block_cols(f, J) + 63) / 64;
                const long long items = nc * nr - nc * (nc - 1) / 2;
                pt.flop_top[(size_t)q] += 2.0 * 64 * 64 * 256 * (double)items * (double)std::min(J, f.nsteps);
            }
            pt.max_panel = std::max(pt.max_panel, top_block_ld(f, 0) * 256);
        }
    }
    for (size_t r = 0; r < Rn; ++r) {
        double par[2] = {0.0, 0.0};
        for (int d = 0; d <= t.maxdepth; ++d) par[d & 1] = std::max(par[d & 1], sdepth[r][(size_t)d]);
        pt.b_schur[r] = par[0] + par[1];
        pt.b_recv[r] = pt.top.empty() ? 0.0 : 3.0 * 8.0 * (double)pt.max_panel;
        pt.b_vec[r] = 2.0 * 8.0 * (double)t.vec_doubles + 4.0 * ((double)t.bpos.size() * 2 + (double)t.vec_doubles * 2 + 2.0 * t.g.ncol);
        pt.b_total[r] = pt.b_panels[r] + pt.b_schur[r] + pt.b_top[r] + pt.b_inv[r] + pt.b_recv[r] + pt.b_vec[r];
    }
}

std::string nd_check(const NdTree &t)
{
    const Grid &g = t.g;
    std::vector<char> seen((size_t)g.ncol, 0);
    for (size_t id = 0; id < t.fr.size(); ++id) {
        const NdFront &f = t.fr[id];
        for (int r = 0; r < f.w; ++r) {
            const int v = t.ownvar[(size_t)(f.rofs + r)];
            if (v < 0 || v >= g.ncol || seen[(size_t)v]) return "a node is owned twice or not at all";
            if (t.pos[(size_t)v] != f.own0 + r || t.front_of[(size_t)v] != (int)id) return "own rows out of order";
            seen[(size_t)v] = 1;
        }
        int last = f.own0 + f.w - 1;                 // the border is eliminated after the own variables, in ascending order
        for (int i = 0; i < f.h; ++i) {
            const int ps = t.bpos[(size_t)(f.bofs + i)];
            if (ps <= last) return "border not ascending / not after the own variables";
            last = ps;
        }
        if (f.parent >= 0) {
            const NdFront &p = t.fr[(size_t)f.parent];
            if (p.depth != f.depth - 1) return "depth mismatch";
            int lastrow = -1;
            for (int i = 0; i < f.h; ++i) {
                const int row = t.pmap[(size_t)(f.bofs + i)];
                if (row <= lastrow || row >= p.fp) return "child -> parent map not monotone";
                if (row >= p.w && row < p.wp) return "child row mapped onto the parent's padding";
                const int pv = row < p.wp ? t.ownvar[(size_t)(p.rofs + row)] : t.bvar[(size_t)(p.bofs + row - p.wp)];
                if (pv != t.bvar[(size_t)(f.bofs + i)]) return "child -> parent map hits another node";
                lastrow = row;
            }
        } else if (f.h != 0) return "the root has a border";
    }
    for (int c = 0; c < g.ncol; ++c)
        if (!seen[(size_t)c]) return "a node is not owned";
    // every entry of N (7^d stencil) has a place: the column's front holds the row
    // (checked on a sample of nodes to keep the test fast on big grids)
    const int step = g.ncol > 200000 ? 97 : 1;
    for (int c = 0; c < g.ncol; c += step) {
        int ic[MAXD];
        for (int d = 0; d < g.ndim; ++d) ic[d] = (c / g.colstride[d]) % g.nodes[d];
        int o[MAXD] = {-3, -3, -3, -3};
        for (int d = g.ndim; d < MAXD; ++d) o[d] = 0;
        for (;;) {
            bool ok = true;
            int r = 0;
            for (int d = 0; d < g.ndim; ++d) {
                const int j = ic[d] + o[d];
                if (j < 0 || j >= g.nodes[d]) ok = false;
                r += j * g.colstride[d];
            }
            if (ok && t.pos[(size_t)r] > t.pos[(size_t)c]) {
                const NdFront &f = t.fr[(size_t)t.front_of[(size_t)c]];
                const int pr = t.pos[(size_t)r];
                bool found = pr >= f.own0 && pr < f.own0 + f.w;
                if (!found) {
                    const int *pb = t.bpos.data() + f.bofs;
                    found = std::binary_search(pb, pb + f.h, pr);
                }
                if (!found) return "an entry of the normal equations has no row in its column's front";
            }
            int d = 0;
            while (d < g.ndim) {
                if (++o[d] <= 3) break;
                o[d] = -3;
                ++d;
            }
            if (d == g.ndim) break;
        }
    }
    return "";
}

}  // namespace splpak

extern "C" int32_t splpak_debug_nd_partition(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t ngpus, int32_t chunk,
                                             double *out_per_rank8, double *out8)
{
    using namespace splpak;
    double xmin[MAXD] = {0, 0, 0, 0}, xmax[MAXD] = {1, 1, 1, 1};
    Grid g;
    if (!nodes || !out_per_rank8 || ngpus < 1 || ngpus > 64) return SPLPAK_E_BADARG;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, nullptr, true);
    if (v != 0) return v;
    NdTree t;
    if (!nd_build(g, t, split_min > 0 ? split_min : nd_default_split_min(ndim))) { set_error("nested dissection: inconsistent tree"); return SPLPAK_E_BADARG; }
    NdPartition pt;
    nd_partition(t, ngpus, chunk, pt);
    for (int r = 0; r < pt.R; ++r) {
        double *o = out_per_rank8 + 8 * r;
        o[0] = pt.b_total[(size_t)r];
        o[1] = pt.b_panels[(size_t)r];
        o[2] = pt.b_schur[(size_t)r];
        o[3] = pt.b_top[(size_t)r];
        o[4] = pt.b_inv[(size_t)r];
        o[5] = pt.b_recv[(size_t)r] + pt.b_vec[(size_t)r];
        o[6] = pt.flop_sub[(size_t)r];
        o[7] = pt.flop_top[(size_t)r];
    }
    if (out8) {
        out8[0] = pt.dcut;
        out8[1] = (double)pt.top.size();
        out8[2] = pt.nseq;
        out8[3] = 8.0 * (double)pt.max_panel;
        // what every rank holds besides the factorisation, for the caller's bookkeeping (doubles -> bytes): the all-reduced
        // normal equations (half stencil + right-hand side + histogram + residual)
        out8[4] = 8.0 * ((double)g.ncol * g.hstencil + 3.0 * g.ncol);
        out8[5] = (double)t.fr.size();
        out8[6] = t.maxdepth;
        out8[7] = t.flop;
    }
    return 0;
}

extern "C" int32_t splpak_debug_nd_tree(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t check, double *out16)
{
    using namespace splpak;
    double xmin[MAXD] = {0, 0, 0, 0}, xmax[MAXD] = {1, 1, 1, 1};
    Grid g;
    if (!nodes || !out16) return SPLPAK_E_BADARG;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, nullptr, true);
    if (v != 0) return v;
    NdTree t;
    if (!nd_build(g, t, split_min > 0 ? split_min : nd_default_split_min(ndim))) { set_error("nested dissection: inconsistent tree"); return SPLPAK_E_BADARG; }
    if (check) {
        const std::string msg = nd_check(t);
        if (!msg.empty()) { set_error("nested dissection: " + msg); return SPLPAK_E_BADARG; }
    }
    if (splpak::opt_get("SPLPAK_DEBUG")) {
        for (int d = 0; d <= t.maxdepth; ++d) {
            double sb = 0, fb = 0, fl = 0, flp = 0;
            int steps = 0, mw = 0, mh = 0;
            for (int id : t.by_depth[(size_t)d]) {
                const NdFront &f = t.fr[(size_t)id];
                sb += 8.0 * f.lds * f.hp;
                fb += 8.0 * f.ld * f.wp;
                const double w = f.w, h = f.h;
                fl += w * w * w / 3.0 + w * w * h + w * h * h;
                for (int k = 0; k < f.nsteps; ++k) {
                    const long long nc = (f.wp - (k + 1) * 256) / 64, nr = (f.fp - (k + 1) * 256) / 64, ns = f.hp / 64;
                    flp += 2.0 * 64 * 64 * 256 * (double)(nc * nr - nc * (nc - 1) / 2 + ns * (ns + 1) / 2);
                }
                steps = std::max(steps, f.nsteps);
                mw = std::max(mw, f.w);
                mh = std::max(mh, f.h);
            }
            fprintf(stderr, "[nd] depth %2d: %5zu fronts, steps %3d, max w %6d h %6d, S %.2f GB, panels %.2f GB, flop exact %.3e padded %.3e\n", d,
                    t.by_depth[(size_t)d].size(), steps, mw, mh, sb / 1e9, fb / 1e9, fl, flp);
        }
    }
    int maxw = 0, maxh = 0;
    for (const NdFront &f : t.fr) { maxw = std::max(maxw, f.w); maxh = std::max(maxh, f.h); }
    out16[0] = (double)t.fr.size();
    out16[1] = t.maxdepth;
    out16[2] = 8.0 * (double)t.factor_doubles;                         // bytes of the factor panels
    out16[3] = 8.0 * (double)(t.s_doubles[0] + t.s_doubles[1]);       // bytes of the two Schur arenas
    out16[4] = 8.0 * (double)t.s_total;
    out16[5] = t.flop;
    out16[6] = t.flop_exact;
    out16[7] = maxw;
    out16[8] = maxh;
    out16[9] = t.nblocks;
    out16[10] = (double)t.vec_doubles;
    out16[11] = (double)t.own_rows;
    out16[12] = (double)t.border_rows;
    out16[13] = 2.0 * 8.0 * 65536.0 * t.nblocks + 8.0 * 4096.0 * t.nblocks;   // bytes of the block inverses
    out16[14] = out16[15] = 0;
    return 0;
}

extern "C" int32_t splpak_debug_nd_schedule(int32_t ndim, const int32_t *nodes, int32_t split_min, int32_t cut, int32_t packed, double *out8)
{
    using namespace splpak;
    double xmin[MAXD] = {0, 0, 0, 0}, xmax[MAXD] = {1, 1, 1, 1};
    Grid g;
    if (!nodes || !out8) return SPLPAK_E_BADARG;
    const int v = build_grid(ndim, nodes, xmin, xmax, g, nullptr, true);
    if (v != 0) return v;
    NdTree t;
    if (!nd_build(g, t, split_min > 0 ? split_min : nd_default_split_min(ndim))) { set_error("nested dissection: inconsistent tree"); return SPLPAK_E_BADARG; }
    NdSchedule sc;
    nd_schedule(t, cut, packed != 0, nullptr, nullptr, 0, sc);
    // every front in exactly one stage, children in earlier stages, no two live blocks overlap
    std::string bad;
    for (size_t id = 0; id < t.fr.size() && bad.empty(); ++id) {
        const int st = sc.stage_of[id];
        if (st < 0) { bad = "a front is in no stage"; break; }
        for (int c : t.fr[id].child)
            if (c >= 0 && sc.stage_of[(size_t)c] >= st) bad = "a child is not eliminated before its parent";
    }
    const int ns = (int)sc.st.size();
    for (int a = 0; a < ns && bad.empty(); ++a)
        for (int b = a + 1; b < ns && bad.empty(); ++b) {
            const NdStage &A = sc.st[(size_t)a], &B = sc.st[(size_t)b];
            if (A.doubles <= 0 || B.doubles <= 0) continue;
            const bool time_overlap = B.first <= a;                      // A lives over stages [A.first, a], B over [B.first, b], a < b
            const bool mem_overlap = A.off < B.off + B.doubles && B.off < A.off + A.doubles;
            if (time_overlap && mem_overlap) bad = "two live Schur blocks overlap";
        }
    if (!bad.empty()) { set_error("nested dissection schedule: " + bad); return SPLPAK_E_BADARG; }
    out8[0] = ns;
    out8[1] = 8.0 * (double)sc.arena;
    out8[2] = 8.0 * (double)sc.total;
    out8[3] = 8.0 * (double)t.factor_doubles;
    out8[4] = sc.cut;
    out8[5] = t.maxdepth;
    out8[6] = out8[7] = 0;
    return 0;
}
