// srcnn_plan.cpp -- launch geometry of the strip kernels: the regular strip x segment x frame grid (make_plan), the balanced
// work items of a plane launched alone and their seams (plan_items*), the per-context cache of device item tables, seam
// scratch per stream, frames per launch, and srcnn_query_plan.  Nothing here changes a pixel: any plan computes the same plane.
#include "srcnn_ctx.h"

using namespace srcnn;
using namespace srcnn::host;

namespace srcnn {
namespace host {

// Choose the row-segment height: taller segments waste fewer halo rows (2*halo recomputed feature
// rows per segment) and fewer workgroup start-ups (weight fragments, 9-row Y prologue: worth about
// STARTUP_ROWS rows), more segments fill the 2-workgroups-per-CU slots more evenly.  Everything is
// regular, so scan.
Plan make_plan(const srcnn_ctx *c, int width, int rows, int n_frames, int halo, int wgs_per_cu, int col_halo)
{
    const int ow = FW - 2 * (col_halo < 0 ? halo : col_halo);
    Plan best{rows, (width + ow - 1) / ow, 1};
    const long slots = (long)wgs_per_cu * c->n_cu;
    double best_eff = -1.0;
    const int max_segs = std::min(rows, 4096);
    for (int ns = 1; ns <= max_segs; ++ns) {
        const int seg = (rows + ns - 1) / ns;
        const int real_ns = (rows + seg - 1) / seg;
        if (real_ns != ns) continue;
        const long wgs = (long)best.n_strips * ns * n_frames;
        const long rounds = (wgs + slots - 1) / slots;
        const double fill = (double)wgs / (double)(rounds * slots);
        constexpr int STARTUP_ROWS = 3;
        const double useful = (double)rows / ((double)ns * (seg + 2 * halo + STARTUP_ROWS));
        const double eff = fill * useful;
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best.seg_rows = seg;
            best.n_segs = ns;
        }
    }
    return best;
}

// Balanced plan for two workgroups per CU with seams (the float32 fused kernel on a plane launched alone).
//
// Measured (tools/diag_light.py, profiles/r02/diag_light_*.txt): with two workgroups on a CU the first-dispatched
// one (wave slot 0 wins the age-based MFMA arbitration) takes kPairFast us per row, the one that joins it kPairSlow;
// a workgroup left alone on its CU takes kAlone -- less per row than either, but more than half of both together,
// so a CU is fastest when its two items end together, and the launch ends with the slowest CU.  The round-1 planner
// cut every strip into k or k+1 items of two heights and paired tall with short: CUs carried 247..255 rows and
// the median CU idled for the last 1.5-3.4 % of the launch.
//
// Here: same item counts per strip (every strip is tiled exactly by ITS items, whatever their order), same pairing
// to start from, then a local search moves single rows between two items of the same strip while that lowers the
// estimated finish time of the slower of the two CUs involved -- until the slowest CU cannot be improved.
// `cu_speed` (optional, one factor per CU) scales the estimate per CU.  It is not used in production: feeding back the
// per-XCD finish times of earlier launches was tried and made things worse -- which XCD runs 1-2 % slow changes from
// launch to launch (profiles/r02/ablation.txt).  Placement only affects speed: any plan computes the same plane.
// (round 3: least-squares fit of this model to the finish times of 4,096 CUs over 16 stamped launches with different row splits,
// profiles/r03/planner_fit.txt -- rms 6.7 us, of which launch-to-launch and per-XCD noise is most; round 2's constants were
// 6.85 / 8.35 / 4.3 / 3.0 / 7.2 and left 247..254 rows per CU where these leave 251..254)
double kPairFast = 6.40, kPairSlow = 8.40, kAlone = 3.76, kStartFast = 3.63, kStartSlow = 5.44;   // us

double cu_finish_estimate(int fast_rows, int slow_rows, double speed)
{
    static const bool once = [] {
        if (const char *e = SRCNN_DEBUG_ENV("SRCNN_DEBUG_RATES"))      // experiment knob: "fast,slow,alone[,start_fast,start_slow]"
            std::sscanf(e, "%lf,%lf,%lf,%lf,%lf", &kPairFast, &kPairSlow, &kAlone, &kStartFast, &kStartSlow);
        return true;
    }();
    (void)once;
    const double tf = kStartFast + fast_rows * kPairFast, ts = kStartSlow + slow_rows * kPairSlow;
    double t;
    if (tf <= ts) t = tf + std::max(0.0, slow_rows - (tf - kStartSlow) / kPairSlow) * kAlone;     // the slow one is left alone
    else t = ts + std::max(0.0, fast_rows - (ts - kStartFast) / kPairFast) * kAlone;
    return t / speed;
}

// extra_top / extra_bot: feature rows the first / last item of every strip computes beyond its own output rows -- the two rows of
// layer-3 radius when the launch covers rows [row_begin, row_end) of a taller plane (a rank's stripe: srcnn_forward_y_rows*_dev);
// at the image's own top and bottom those rows replicate and cost nothing.  They weigh like rows of the item in the balance
// (a middle rank's 540-row stripe of a 7680-wide plane otherwise finishes one row pair late on the 60 CUs that hold such items).
ItemPlan plan_items_balanced(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int extra_top, int extra_bot,
                             const double *cu_speed = nullptr)
{
    const ItemPlan none;
    constexpr int kMinRows = 10;
    // Items at least kSepMinRows tall: worth trying to keep the seam windows of neighbouring strips apart; kLaunchSaved = us such a
    // plan may cost in balance.  One seam launch instead of two saves ~2 us, but in a stream of launches with seam deferral only a
    // separated plan lets the seam blocks ride behind the next launch (no seam launch at all).  Measured A B B A
    // (profiles/r05/plan_separation_ab.txt), (12, 8) against round 4's (24, 2): 1280x720 0.743 -> 0.789 of the MFMA peak with
    // deferral, 0.744 -> 0.753 without; 1366x768 and 1280x1024 the same picture; unchanged plans from 1920x1080 up.
    constexpr int kSepMinRows = 12;
    constexpr double kLaunchSaved = 8.0;
    const int hs = row_end - row_begin;
    const int slots = 2 * n_cu;
    if (n_cu <= 0 || n_strips <= 0 || n_strips > n_cu || hs <= 0 || slots / n_strips < 2 || hs / (slots / n_strips + 1) < kMinRows + 2)
        return none;
    // items per strip and how many of them go to first-dispatched blocks (as many fast as slow items overall)
    const int kbase = slots / n_strips, kextra = slots % n_strips;
    std::vector<int> k((size_t)n_strips), a((size_t)n_strips);
    int fast_total = 0;
    for (int s = 0; s < n_strips; ++s) {
        k[(size_t)s] = kbase + (s < kextra ? 1 : 0);
        a[(size_t)s] = k[(size_t)s] / 2;
        fast_total += a[(size_t)s];
    }
    for (int pass = 0; pass < 2 && fast_total < n_cu; ++pass)
        for (int s = 0; s < n_strips && fast_total < n_cu; ++s)
            if ((pass == 1 || k[(size_t)s] % 2 == 1) && a[(size_t)s] < k[(size_t)s] - 1) { ++a[(size_t)s]; ++fast_total; }
    if (fast_total != n_cu) return none;
    struct Item { int strip, rows, cu; bool fast; int extra = 0; };      // extra: feature rows computed beyond `rows` (see above)
    std::vector<Item> items;
    const double d = std::min(std::max(skew_pct, 0), 60) / 100.0;
    for (int s = 0; s < n_strips; ++s) {
        const int na = a[(size_t)s], nb = k[(size_t)s] - na;
        const double u = hs / (na * (1.0 + d) + nb * (1.0 - d));
        double acc = 0.0;
        int used = 0;
        for (int j = 0; j < k[(size_t)s]; ++j) {               // alternate tall / short down the strip
            const bool fast = (j % 2 == 0) ? (j / 2 < na) : !((j / 2) < nb);
            acc += fast ? (1.0 + d) * u : (1.0 - d) * u;
            const int upto = (j == k[(size_t)s] - 1) ? hs : (int)std::lround(acc);
            items.push_back({s, upto - used, -1, fast, (j == 0 ? extra_top : 0) + (j == k[(size_t)s] - 1 ? extra_bot : 0)});
            used = upto;
        }
        // Strips get the same item heights, so their boundaries would line up from strip to strip.  Every second strip
        // is shifted up by kStagger rows (its first item shorter, its last one taller): the seam windows of neighbouring
        // strips then lie well apart (ItemPlan::separated: one seam launch instead of two) and the search below can
        // still move boundaries by a few rows.
        constexpr int kStagger = 16;
        if ((s & 1) && k[(size_t)s] >= 3) {
            Item &first = items[items.size() - (size_t)k[(size_t)s]], &last = items.back();
            const int x = std::min(kStagger, first.rows - kMinRows);
            if (x > 0) {
                first.rows -= x;
                last.rows += x;
            }
        }
    }
    // the alternation above may not hand out exactly na fast items per strip when na != nb: recount and fix the flags
    for (int s = 0; s < n_strips; ++s) {
        int have = 0;
        for (auto &it : items) if (it.strip == s && it.fast) ++have;
        for (auto &it : items) if (it.strip == s && have > a[(size_t)s] && it.fast) { it.fast = false; --have; }
        for (auto &it : items) if (it.strip == s && have < a[(size_t)s] && !it.fast) { it.fast = true; ++have; }
    }
    // pair the tallest fast item with the shortest slow one
    std::vector<int> fi, si;
    for (int i = 0; i < (int)items.size(); ++i) (items[(size_t)i].fast ? fi : si).push_back(i);
    if ((int)fi.size() != n_cu || (int)si.size() != n_cu) return none;
    std::stable_sort(fi.begin(), fi.end(), [&](int x, int y) { return items[(size_t)x].rows + items[(size_t)x].extra > items[(size_t)y].rows + items[(size_t)y].extra; });
    std::stable_sort(si.begin(), si.end(), [&](int x, int y) { return items[(size_t)x].rows + items[(size_t)x].extra < items[(size_t)y].rows + items[(size_t)y].extra; });
    for (int c = 0; c < n_cu; ++c) items[(size_t)fi[(size_t)c]].cu = items[(size_t)si[(size_t)c]].cu = c;
    auto speed = [&](int c) { return cu_speed && cu_speed[c] > 0.5 && cu_speed[c] < 2.0 ? cu_speed[c] : 1.0; };
    auto work = [&](int i) { return items[(size_t)i].rows + items[(size_t)i].extra; };
    auto finish = [&](int c) { return cu_finish_estimate(work(fi[(size_t)c]), work(si[(size_t)c]), speed(c)); };
    // local search: take one row from an item of the slowest improvable CU, give it to the item of the same strip
    // whose CU stays fastest; stop when no such move lowers the pair's maximum
    std::vector<std::vector<int>> in_strip((size_t)n_strips);
    for (int i = 0; i < (int)items.size(); ++i) in_strip[(size_t)items[(size_t)i].strip].push_back(i);
    std::vector<double> fin((size_t)n_cu);
    for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    std::vector<int> by_time((size_t)n_cu);
    // Seam windows: the boundary rows of strip s (relative to row_begin) and whether they keep SEAM_ROWS rows away from
    // every boundary of the strips next to it (ItemPlan::separated: one seam launch instead of two).
    auto bounds = [&](int s_) {
        std::vector<int> b;
        int y = 0;
        for (size_t q = 0; q + 1 < in_strip[(size_t)s_].size(); ++q) b.push_back(y += items[(size_t)in_strip[(size_t)s_][q]].rows);
        return b;
    };
    auto apart = [&](const std::vector<int> &x, const std::vector<int> &y) {
        for (int u : x)
            for (int v : y)
                if (std::abs(u - v) < SEAM_ROWS) return false;
        return true;
    };
    auto strip_apart = [&](int s_) {
        const std::vector<int> b = bounds(s_);
        return (s_ == 0 || apart(b, bounds(s_ - 1))) && (s_ + 1 >= n_strips || apart(b, bounds(s_ + 1)));
    };
    // local search: take one row from an item of the slowest improvable CU, give it to the item of the same strip
    // whose CU stays fastest; stop when no such move lowers the pair's maximum.  `keep_apart`: only moves that leave the
    // strip's seam windows clear of its neighbours'.
    // the same test for ONE candidate move of the search, incrementally: a row from item `from` to item `to` of a strip
    // shifts the boundaries between them by one row (cur[s] = the strip's boundaries, kept in step by move_apply())
    std::vector<std::vector<int>> cur((size_t)n_strips);
    std::vector<int> pos(items.size());
    for (int s_ = 0; s_ < n_strips; ++s_)
        for (size_t q = 0; q < in_strip[(size_t)s_].size(); ++q) pos[(size_t)in_strip[(size_t)s_][q]] = (int)q;
    auto clear_of = [&](const std::vector<int> &nbr, int b) {       // b keeps SEAM_ROWS rows away from every entry of nbr (sorted)
        const auto it = std::lower_bound(nbr.begin(), nbr.end(), b);
        return (it == nbr.end() || *it - b >= SEAM_ROWS) && (it == nbr.begin() || b - *(it - 1) >= SEAM_ROWS);
    };
    auto move_ok = [&](int from, int to) {
        const int s_ = items[(size_t)from].strip, pa = pos[(size_t)from], pb = pos[(size_t)to];
        const int lo = std::min(pa, pb), hi = std::max(pa, pb) - 1, delta = pa < pb ? -1 : 1;
        for (int q = lo; q <= hi; ++q) {
            const int b = cur[(size_t)s_][(size_t)q] + delta;
            if (s_ > 0 && !clear_of(cur[(size_t)s_ - 1], b)) return false;
            if (s_ + 1 < n_strips && !clear_of(cur[(size_t)s_ + 1], b)) return false;
        }
        return true;
    };
    auto move_apply = [&](int from, int to) {
        const int s_ = items[(size_t)from].strip, pa = pos[(size_t)from], pb = pos[(size_t)to];
        const int lo = std::min(pa, pb), hi = std::max(pa, pb) - 1, delta = pa < pb ? -1 : 1;
        for (int q = lo; q <= hi; ++q) cur[(size_t)s_][(size_t)q] += delta;
    };
    auto search = [&](bool keep_apart) {
        for (int iter = 0; iter < 40 * n_cu; ++iter) {
            for (int c = 0; c < n_cu; ++c) by_time[(size_t)c] = c;
            std::sort(by_time.begin(), by_time.end(), [&](int x, int y) { return fin[(size_t)x] > fin[(size_t)y]; });
            bool moved = false;
            for (int rank = 0; rank < n_cu && !moved; ++rank) {
                const int c = by_time[(size_t)rank];
                double best_gain = 1e-6;
                int best_from = -1, best_to = -1;
                for (int from : {fi[(size_t)c], si[(size_t)c]}) {
                    if (items[(size_t)from].rows <= kMinRows) continue;
                    items[(size_t)from].rows -= 1;
                    const double mine = finish(c);
                    for (int to : in_strip[(size_t)items[(size_t)from].strip]) {
                        const int c2 = items[(size_t)to].cu;
                        if (c2 == c) continue;
                        items[(size_t)to].rows += 1;
                        const double theirs = finish(c2);
                        const double gain = fin[(size_t)c] - std::max(mine, theirs);
                        if (theirs < fin[(size_t)c] && gain > best_gain && (!keep_apart || move_ok(from, to))) {
                            best_gain = gain;
                            best_from = from;
                            best_to = to;
                        }
                        items[(size_t)to].rows -= 1;
                    }
                    items[(size_t)from].rows += 1;
                }
                if (best_from >= 0) {
                    if (keep_apart) move_apply(best_from, best_to);
                    items[(size_t)best_from].rows -= 1;
                    items[(size_t)best_to].rows += 1;
                    fin[(size_t)c] = finish(c);
                    fin[(size_t)items[(size_t)best_to].cu] = finish(items[(size_t)best_to].cu);
                    moved = true;
                }
            }
            if (!moved) break;
        }
    };
    // move colliding boundaries apart first (left to right: the shift that clears the left neighbour's windows with some
    // slack and keeps the two CUs involved fastest), then balance under that constraint; if that fails, balance freely
    // (two seam launches then)
    const std::vector<Item> start = items;
    static const char *env_sep = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEPARATE");     // experiment knob: 0 = never keep the seam windows apart
    // (worth trying only with neighbours to keep apart from and items tall enough to give up a few rows)
    static const char *env_sep_rows = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEP_MINROWS"), *env_sep_saved = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEP_SAVED");   // experiment knobs
    const int sep_min_rows = env_sep_rows ? std::atoi(env_sep_rows) : kSepMinRows;
    bool separated = !(env_sep && std::atoi(env_sep) == 0) && n_strips >= 2 && hs / (kbase + 1) >= sep_min_rows;
    constexpr int kSlack = 4;        // preferred extra distance: the search needs room to move boundaries
    for (int s_ = 1; s_ < n_strips && separated; ++s_) {
        const std::vector<int> left = bounds(s_ - 1);
        const std::vector<int> &mine = in_strip[(size_t)s_];
        for (size_t q = 0; q + 1 < mine.size() && separated; ++q) {
            int y = 0;
            for (size_t r = 0; r <= q; ++r) y += items[(size_t)mine[r]].rows;
            auto dist = [&](int b) {
                int dmin = 1 << 30;
                for (int v : left) dmin = std::min(dmin, std::abs(v - b));
                return dmin;
            };
            if (dist(y) >= SEAM_ROWS + kSlack) continue;
            Item &up = items[(size_t)mine[q]], &dn = items[(size_t)mine[q + 1]];
            int best_d = 0;
            double best_t = 1e30;
            for (int d = -3 * SEAM_ROWS; d <= 3 * SEAM_ROWS; ++d) {
                if (dist(y + d) < SEAM_ROWS || up.rows + d < kMinRows || dn.rows - d < kMinRows) continue;
                up.rows += d;
                dn.rows -= d;
                const double t = std::max(finish(up.cu), finish(dn.cu)) + 0.5 * std::abs(d) +
                                 4.0 * std::max(0, SEAM_ROWS + kSlack - dist(y + d));
                up.rows -= d;
                dn.rows += d;
                if (t < best_t) { best_t = t; best_d = d; }
            }
            if (best_t >= 1e30) { separated = false; break; }
            up.rows += best_d;
            dn.rows -= best_d;
        }
    }
    for (int s_ = 0; s_ < n_strips && separated; ++s_) separated = strip_apart(s_);
    auto slowest = [&] {
        double mx = 0.0;
        for (int c = 0; c < n_cu; ++c) mx = std::max(mx, fin[(size_t)c]);
        return mx;
    };
    if (separated) {
        for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
        for (int s_ = 0; s_ < n_strips; ++s_) cur[(size_t)s_] = bounds(s_);
        search(true);
    }
    // the unconstrained plan, for comparison: keeping the windows apart must not cost more than the launch it saves
    const std::vector<Item> apart_items = items;
    const double apart_t = separated ? slowest() : 1e30;
    items = start;
    for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    search(false);
    const double launch_saved = env_sep_saved ? std::atof(env_sep_saved) : kLaunchSaved;
    if (separated && apart_t <= slowest() + launch_saved) {
        items = apart_items;
        for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    } else {
        separated = false;
    }
    if (SRCNN_DEBUG_ENV("SRCNN_DEBUG_PLANLOG")) {
        double mx = 0, mn = 1e30;
        for (int c = 0; c < n_cu; ++c) { mx = std::max(mx, fin[(size_t)c]); mn = std::min(mn, fin[(size_t)c]); }
        std::fprintf(stderr, "plan: separated=%d finish %.1f..%.1f\n", (int)separated, mn, mx);
    }
    // positions: the items of a strip in creation order; seams between neighbours
    ItemPlan plan;
    plan.separated = separated;
    std::vector<int> y0(items.size()), up(items.size(), -1), dn(items.size(), -1);
    for (int s = 0; s < n_strips; ++s) {
        int y = row_begin, prev = -1;
        for (int i : in_strip[(size_t)s]) {
            if (items[(size_t)i].rows < 2 * SEAM_ROWS) return none;
            y0[(size_t)i] = y;
            y += items[(size_t)i].rows;
            if (prev >= 0) {
                const int id = plan.n_seams();
                plan.seams.insert(plan.seams.end(), {s, y0[(size_t)i]});
                dn[(size_t)prev] = id;
                up[(size_t)i] = id;
            }
            prev = i;
        }
        if (y != row_end) return none;
    }
    auto emit = [&](int i) {
        plan.items.insert(plan.items.end(), {items[(size_t)i].strip, y0[(size_t)i], y0[(size_t)i] + items[(size_t)i].rows, up[(size_t)i], dn[(size_t)i]});
    };
    for (int c = 0; c < n_cu; ++c) emit(fi[(size_t)c]);
    for (int c = 0; c < n_cu; ++c) emit(si[(size_t)c]);
    return plan;
}

ItemPlan plan_items_raw(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu = 2,
                        bool want_seams = false, int extra_top = 0, int extra_bot = 0)
{
    const ItemPlan none;
    static const char *env_plan = SRCNN_DEBUG_ENV("SRCNN_DEBUG_PLAN");      // experiment knob: 1 = the round-1 planner
    if (wgs_per_cu == 2 && want_seams && skew_pct > 0 && !(env_plan && std::atoi(env_plan) == 1)) {
        const ItemPlan balanced = plan_items_balanced(n_cu, n_strips, row_begin, row_end, skew_pct, extra_top, extra_bot);
        if (balanced.count() > 0) return balanced;
    }
    const int rows = row_end - row_begin;
    int slots = wgs_per_cu * n_cu;
    // shortest useful item: with halo rows to recompute (4 per item) short items do not pay; with seams an item
    // only has to be tall enough for the hand-over (2 * SEAM_ROWS, 10 for some slack in the skewed heights)
    const int min_rows = want_seams ? 10 : 24;
    if (skew_pct <= 0 || n_strips <= 0 || n_strips > n_cu) return none;
    // one item per CU: a small plane may leave CUs idle rather than cut its strips into items shorter than that
    bool underfilled = false;
    if (wgs_per_cu == 1 && want_seams && rows / min_rows < slots / n_strips + 1) {
        slots = n_strips * (rows / min_rows);       // every strip in rows / min_rows items of >= min_rows rows
        underfilled = true;
    }
    if (slots / n_strips < 2 || (!underfilled && rows / (slots / n_strips + 1) < min_rows)) return none;
    if (wgs_per_cu == 1) skew_pct = 0;
    const int kbase = slots / n_strips, kextra = slots % n_strips;      // strips [0,kextra) get kbase+1 items
    std::vector<int> k(n_strips), a(n_strips);
    int fast_total = 0;
    for (int s = 0; s < n_strips; ++s) {
        k[s] = kbase + (s < kextra ? 1 : 0);
        a[s] = wgs_per_cu == 1 ? k[s] : k[s] / 2;
        fast_total += a[s];
    }
    for (int s = 0; fast_total < n_cu && s < n_strips; ++s)              // odd counts first, then any
        if (k[s] % 2 == 1 && a[s] < k[s] - 1) { ++a[s]; ++fast_total; }
    for (int s = 0; fast_total < n_cu && s < n_strips; ++s)
        if (a[s] < k[s] - 1) { ++a[s]; ++fast_total; }
    if (wgs_per_cu == 2 && fast_total != n_cu) return none;
    const double d = skew_pct / 100.0;
    std::vector<std::vector<int>> bounds(n_strips);
    for (int s = 0; s < n_strips; ++s) {
        const int b = k[s] - a[s];
        const double u = rows / (a[s] * (1.0 + d) + b * (1.0 - d));
        bounds[s].resize(k[s] + 1);
        for (int j = 0; j <= k[s]; ++j) {
            const double y = j <= a[s] ? j * (1.0 + d) * u : a[s] * (1.0 + d) * u + (j - a[s]) * (1.0 - d) * u;
            bounds[s][j] = row_begin + std::min(rows, std::max(0, (int)std::lround(y)));
        }
        bounds[s][0] = row_begin;
        bounds[s][k[s]] = row_end;
        for (int j = 1; j <= k[s]; ++j)
            if (bounds[s][j] <= bounds[s][j - 1]) return none;          // degenerate: regular grid instead
    }
    // Block i and block n_cu + i share a CU (measured, tools/diag_stamps.py): pair the tallest fast
    // item with the shortest slow one so that every CU carries the same number of rows.
    struct Item { int strip, y0, y1, up, dn; };
    ItemPlan plan;
    // a seam needs 4 rows of the item below and leaves 2 rows either side to the seam kernel
    for (int s = 0; s < n_strips && want_seams; ++s)
        for (int j = 0; j < k[s]; ++j)
            if (bounds[s][j + 1] - bounds[s][j] < 2 * SEAM_ROWS) want_seams = false;
    std::vector<Item> fast, slow;
    for (int s = 0; s < n_strips; ++s)
        for (int j = 0; j < k[s]; ++j) {
            int up = -1, dn = -1;
            if (want_seams && j > 0) up = plan.n_seams() - 1;                // made by the item above
            if (want_seams && j < k[s] - 1) {
                dn = plan.n_seams();
                plan.seams.insert(plan.seams.end(), {s, bounds[s][j + 1]});
            }
            (j < a[s] ? fast : slow).push_back({s, bounds[s][j], bounds[s][j + 1], up, dn});
        }
    std::stable_sort(fast.begin(), fast.end(), [](const Item &x, const Item &y) { return x.y1 - x.y0 > y.y1 - y.y0; });
    std::stable_sort(slow.begin(), slow.end(), [](const Item &x, const Item &y) { return x.y1 - x.y0 < y.y1 - y.y0; });
    plan.items.reserve(ITEM_INTS * (size_t)slots);
    for (const Item &it : fast) plan.items.insert(plan.items.end(), {it.strip, it.y0, it.y1, it.up, it.dn});
    for (const Item &it : slow) plan.items.insert(plan.items.end(), {it.strip, it.y0, it.y1, it.up, it.dn});
    return plan.count() == slots ? plan : none;
}

int skew_percent()
{
    static const char *env_skew = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SKEW");     // experiment knob; 0 = regular grid
    return env_skew ? std::atoi(env_skew) : 10;
}

// A seam's WINDOW is the four output rows b-2 .. b+1 around its boundary row b, which the seam kernel finishes.  When no
// window of a strip shares a row with a window of a NEIGHBOURING strip, the block that finishes a seam can also finish the
// four column-seam pixels either side of its strip on those rows -- the neighbour's values there are complete exports of the
// strip kernel -- and the row-seam and column-seam kernels no longer depend on each other: one launch instead of two.
// The balanced planner builds such plans (ItemPlan::separated) where that costs no balance; other plans keep two launches.
ItemPlan plan_items(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu, bool want_seams, int extra_top,
                    int extra_bot)
{
    return plan_items_raw(n_cu, n_strips, row_begin, row_end, skew_pct, wgs_per_cu, want_seams, extra_top, extra_bot);
}

// Device copy of plan_items() for this geometry, from the context's table cache.  *n_items = 0: use the
// regular grid.  A table is written once, before its first use, into memory no earlier launch reads
// (a fresh slot, or an evicted one after its last reader has finished), and never modified afterwards.
int build_items(srcnn_ctx *c, int n_strips, int row_begin, int row_end, int height, int wgs_per_cu, bool want_seams,
                const srcnn_ctx::ItemTable **table)
{
    *table = nullptr;
    static const char *env_edge = SRCNN_DEBUG_ENV("SRCNN_DEBUG_PLAN_EDGES");     // experiment knob: 0 = round 4's plans (edge rows not weighed)
    const bool edges = !(env_edge && std::atoi(env_edge) == 0);
    // a launch on rows of a taller plane computes two more feature rows at either open end (plan_items_balanced())
    const int extra_top = edges && row_begin > 0 ? 2 : 0, extra_bot = edges && row_end < height ? 2 : 0;
    const int key[8] = {n_strips, row_begin, row_end, skew_percent(), c->n_cu, wgs_per_cu, want_seams ? 1 : 0, extra_top + 4 * extra_bot};
    srcnn_ctx::ItemTable *victim = &c->item_tables[0];
    for (auto &t : c->item_tables) {
        if (t.stamp && std::memcmp(key, t.key, sizeof(key)) == 0) {
            t.stamp = ++c->item_clock;
            *table = &t;
            return SRCNN_OK;
        }
        if (t.stamp < victim->stamp) victim = &t;
    }
    const ItemPlan plan = plan_items(c->n_cu, n_strips, row_begin, row_end, key[3], wgs_per_cu, want_seams, extra_top, extra_bot);
    if (victim->stamp) {                                                // evicting: its readers must be done -- and queued
        if (int rc = flush_seams(c)) return rc;                         // (a deferred seam launch may refer to the victim's tables)
        HIP_TRY(c, hipDeviceSynchronize());
    }
    if (plan.count() > 0) {
        int rc;
        if ((rc = reserve(c, victim->dev, plan.items.size() * sizeof(int)))) return rc;
        HIP_TRY(c, hipMemcpy(victim->dev.p, plan.items.data(), plan.items.size() * sizeof(int), hipMemcpyHostToDevice));
        if (plan.n_seams() > 0) {
            if ((rc = reserve(c, victim->dev_seams, plan.seams.size() * sizeof(int)))) return rc;
            HIP_TRY(c, hipMemcpy(victim->dev_seams.p, plan.seams.data(), plan.seams.size() * sizeof(int),
                                 hipMemcpyHostToDevice));
            if (plan.separated) {
                const int rows = row_end - row_begin;
                std::vector<unsigned char> win((size_t)n_strips * rows, 0);
                for (int id = 0; id < plan.n_seams(); ++id) {
                    const int s_ = plan.seams[2 * (size_t)id], b = plan.seams[2 * (size_t)id + 1];
                    for (int y = b - 2; y < b + 2; ++y)
                        if (y >= row_begin && y < row_end) win[(size_t)s_ * rows + (y - row_begin)] = 1;
                }
                if ((rc = reserve(c, victim->dev_winmap, win.size()))) return rc;
                HIP_TRY(c, hipMemcpy(victim->dev_winmap.p, win.data(), win.size(), hipMemcpyHostToDevice));
            }
        }
    }
    std::memcpy(victim->key, key, sizeof(key));
    victim->count = plan.count();
    victim->n_seams = plan.n_seams();
    victim->separated = plan.n_seams() > 0 && plan.separated;
    victim->stamp = ++c->item_clock;
    *table = victim;
    return SRCNN_OK;
}

// The split-f16 kernel is software-pipelined inside a wave and runs one workgroup per CU (srcnn_split16.hip).
int split16_wgs_per_cu(bool split16, int /*tune*/) { return split16 ? 1 : 2; }

// Seam scratch of the stream the context launches on (one buffer set per stream: srcnn_ctx::SeamScratch).
int seam_scratch_for_stream(srcnn_ctx *c, srcnn_ctx::SeamScratch **out)
{
    srcnn_ctx::SeamScratch *sc = nullptr;
    for (auto &e : c->seam_scratch)
        if (e.used && e.stream == c->stream) sc = &e;
    for (auto &e : c->seam_scratch)
        if (!sc && !e.used) sc = &e;
    if (!sc) {                  // more streams than slots: wait for everything, start over with slot 0
        HIP_TRY(c, hipDeviceSynchronize());
        for (auto &e : c->seam_scratch) e.used = false;
        sc = &c->seam_scratch[0];
    }
    sc->used = true;
    sc->stream = c->stream;
    *out = sc;
    return SRCNN_OK;
}

// Column seams (strips of FW output columns instead of FW - 4 plus two halo columns each side) pay when they save a strip:
// 3840 = 30 instead of 31, 1920 = 15 instead of 16.  Where the count is the same (576: 5 and 5) they only add the export
// work and the third kernel launch.
bool cseam_pays(int width)
{
    const int ns_cs = (width + FW - 1) / FW, ns_halo = (width + FW - 5) / (FW - 4);
    return ns_cs < ns_halo && (width - (ns_cs - 1) * FW >= 4 || ns_cs == 1);
}

// How many frames of a batch go into ONE launch of the fused kernel (srcnn_forward_y_dev; srcnn_query_plan reports the same).
// * A small batch of LARGE planes runs fastest as one single-plane launch per frame (each with its balanced item plan, back
//   to back on the stream) -- ms per frame, same box: 2 x 3840x2160 0.955 against 0.976 for one launch that repeats the item
//   plan frame after frame, 8 x 0.956 / 0.959, 24 x 0.949 / 0.946; 8 x 5760x3240 2.119 / 2.137; 4 x 1920x1080 0.252 against
//   0.275 on the regular grid, 8 x 0.253 / 0.258, 16 x 0.2525 / 0.252 (profiles/r02/ablation.txt section 11).
// * Other batches below kItemBatchMax frames repeat the plane's item plan frame after frame in one launch, whose row-seam
//   scratch is (2 n_cu - n_strips) seams x 43 KB per frame whatever the plane's size (21 MB at 3840x2160, + 4 MB of column
//   seams): at most kItemBatchChunk frames per launch, 200 MB of context-owned scratch per stream instead of 770 MB at 31.
// * Larger batches use the regular strip x segment x frame grid (column-seam scratch only, 4 MB per 3840x2160 frame),
//   at most 64 frames per launch.
// the modes whose fused pass is the float32 MFMA strip kernel (REFBYTES = the same kernel + flags + fix-up)
bool f32_mfma(const srcnn_ctx *c) { return c->mode == SRCNN_MODE_MFMA || c->mode == SRCNN_MODE_REFBYTES; }
int frames_per_launch(const srcnn_ctx *c, int width, int height, int n_frames)
{
    static const char *env_loop = SRCNN_DEBUG_ENV("SRCNN_DEBUG_FRAMELOOP");      // experiment knob: 0 = never one launch per frame
    const size_t px = (size_t)width * height;
    if (c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) return 1;       // flag planes are compact per frame; the FIX-UP spans up to FIX_BATCH_FRAMES of them (run_strip)
    if (c->mode != SRCNN_MODE_MFMA || n_frames <= 1) return kGridBatchChunk;
    if (!(env_loop && std::atoi(env_loop) == 0) &&
        ((px >= ((size_t)4 << 20) && n_frames < kItemBatchMax) || (px >= ((size_t)3 << 19) && n_frames <= 8)))
        return 1;
    return n_frames < kItemBatchMax ? kItemBatchChunk : kGridBatchChunk;
}

}  // namespace host
}  // namespace srcnn

extern "C" {

int srcnn_query_plan(srcnn_ctx *c, int width, int height, int n_frames, int out[6])
{
    if (!c || !out || width <= 0 || height <= 0 || n_frames <= 0) return SRCNN_ERR_INVALID;
    static const char *env_seams = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEAMS");
    const int wgs_per_cu = split16_wgs_per_cu(c->mode == SRCNN_MODE_SPLIT16 || c->mode == SRCNN_MODE_REFBYTES16, 0);
    // mirrors srcnn_forward_y_dev() and run_strip(): `nl` frames go into one launch (1 = one single-plane launch per frame);
    // the float32 fused kernel uses column seams (strips of FW columns) when the geometry allows
    const int nl = std::min(n_frames, frames_per_launch(c, width, height, n_frames));
    const int seam_knob = env_seams ? std::atoi(env_seams) : 3;
    const int ns_cs = (width + FW - 1) / FW;
    bool col_seams = f32_mfma(c) && (seam_knob & 2) && (nl > 1 || (seam_knob & 1)) && cseam_pays(width);
    int items_per_cu = wgs_per_cu;
    const bool row_seams = f32_mfma(c) && (seam_knob & 1);
    auto fits = [&](int n_strips_, int per_cu) { return !plan_items(c->n_cu, n_strips_, 0, height, skew_percent(), per_cu, row_seams).items.empty(); };
    if (col_seams && nl == 1 && !fits(ns_cs, wgs_per_cu)) {
        if (wgs_per_cu == 2 && fits(ns_cs, 1)) items_per_cu = 1;
        else col_seams = false;
    }
    if (!col_seams && nl == 1 && row_seams && wgs_per_cu == 2) {
        const int ns_halo = (width + FW - 5) / (FW - 4);
        if (!fits(ns_halo, 2) && fits(ns_halo, 1)) items_per_cu = 1;
    }
    const Plan pl = make_plan(c, width, height, nl, 2, wgs_per_cu, col_seams ? 0 : -1);
    out[0] = pl.n_strips * pl.n_segs * n_frames;      // over all launches of the batch
    out[1] = pl.seg_rows;
    out[2] = pl.n_strips;
    out[3] = pl.n_segs;
    if (nl == 1 || (f32_mfma(c) && n_frames < kItemBatchMax)) {   // explicit work items (plan_items), repeated per frame of a small batch
        const std::vector<int> items =
            plan_items(c->n_cu, pl.n_strips, 0, height, skew_percent(), items_per_cu,
                       f32_mfma(c) && (seam_knob & 1)).items;
        if (!items.empty()) {
            out[0] = (int)items.size() / ITEM_INTS * n_frames;
            out[1] = 0;
            for (size_t i = 0; i < items.size(); i += ITEM_INTS) out[1] = std::max(out[1], items[i + 2] - items[i + 1]);
            out[3] = ((int)items.size() / ITEM_INTS + pl.n_strips - 1) / pl.n_strips;
        }
    }
    out[4] = (int)strip_lds_bytes(MODE_FUSED);
    out[5] = NTHREADS;
    return SRCNN_OK;
}

#ifdef SRCNN_TUNING_BUILD
/* Undocumented test hook (not part of the ABI, needs no device): the work-item planner.  Fills `items`
 * (ITEM_INTS ints each) and `seams` (2 ints each) up to the given capacities; returns the item count, or
 * SRCNN_ERR_INVALID when a buffer is too small. */
int srcnn_debug_plan_items(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu,
                                      int want_seams, int *items, int max_items, int *seams, int max_seams,
                                      int *n_seams)
{
    // wgs_per_cu + 16 * extra_top + 256 * extra_bot: the planner's edge weights ride in the upper bits (old callers: 0)
    const int extra_top = (wgs_per_cu >> 4) & 15, extra_bot = (wgs_per_cu >> 8) & 15;
    wgs_per_cu &= 15;
    const ItemPlan plan = plan_items(n_cu, n_strips, row_begin, row_end, skew_pct, wgs_per_cu, want_seams != 0, extra_top, extra_bot);
    if (plan.count() > max_items || plan.n_seams() > max_seams || !items || !seams || !n_seams) return SRCNN_ERR_INVALID;
    // an empty vector's data() may be null, which memcpy must not be given even for 0 bytes (found by UBSan)
    if (!plan.items.empty()) std::memcpy(items, plan.items.data(), plan.items.size() * sizeof(int));
    if (!plan.seams.empty()) std::memcpy(seams, plan.seams.data(), plan.seams.size() * sizeof(int));
    *n_seams = plan.n_seams();
    return plan.count();
}
#endif

}  // extern "C"
