// partition.cpp -- k-way partition of a factor graph's variables for the sharded
// (multi-GPU) sweep.  Host code, no GPU: built as libmxs_partition.so.
//
// The reference's analogue is the distribution of computations on agents
// (pydcop/distribution/*.py).  Here every cut factor is replicated on the shards
// owning one of its variables and its remote variables' V->F messages cross once
// per cycle, so the objective is the classic one: balanced
// parts, few cut factors.  METIS is not installed, so this is a multilevel
// partitioner of its own, after the published scheme (Karypis & Kumar 1998):
//   variable graph   two variables are adjacent when they share a factor
//                    (clique expansion; big scopes are closed into a ring)
//   coarsening       heavy-edge matching until ~a hundred vertices are left
//   initial cut      greedy graph growing from several seeds, best kept
//   uncoarsening     boundary Fiduccia-Mattheyses refinement at every level
//   k parts          recursive bisection with proportional targets
// Deterministic for a given seed: every rank computes the same partition on its own.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/maxsum_partition.h"

namespace {

struct Graph {
    int n = 0;
    std::vector<int64_t> xadj;  // [n+1]
    std::vector<int> adj;       // neighbours
    std::vector<int> ew;        // edge weights
    std::vector<int64_t> vw;    // vertex weights
    int64_t total_vw() const { return std::accumulate(vw.begin(), vw.end(), (int64_t)0); }
};

using Rng = std::mt19937_64;

// rows given as unsorted (u, v, w) triples with duplicates -> CSR with merged weights
Graph build_csr(int n, std::vector<int>& eu, std::vector<int>& evv, std::vector<int>& w,
                std::vector<int64_t> vw) {
    Graph g;
    g.n = n;
    g.vw = std::move(vw);
    g.xadj.assign(n + 1, 0);
    for (int u : eu) g.xadj[u + 1]++;
    for (int i = 0; i < n; ++i) g.xadj[i + 1] += g.xadj[i];
    std::vector<int> a(eu.size()), b(eu.size());
    {
        std::vector<int64_t> pos(g.xadj.begin(), g.xadj.end() - 1);
        for (size_t i = 0; i < eu.size(); ++i) {
            const int64_t p = pos[eu[i]]++;
            a[p] = evv[i];
            b[p] = w[i];
        }
    }
    // merge duplicates inside each row with a marker array
    std::vector<int> where(n, -1);
    std::vector<int64_t> nx(n + 1, 0);
    int64_t out = 0;
    for (int u = 0; u < n; ++u) {
        const int64_t row0 = out;
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) {
            const int v = a[p];
            if (v == u) continue;
            if (where[v] >= row0) {
                b[where[v]] += b[p];
            } else {
                where[v] = (int)out;
                a[out] = v;
                b[out] = b[p];
                ++out;
            }
        }
        nx[u + 1] = out;
        // markers of this row must not match later rows: they compare against row0 of that row
    }
    a.resize(out);
    b.resize(out);
    g.xadj = std::move(nx);
    g.adj = std::move(a);
    g.ew = std::move(b);
    return g;
}

// ---- coarsening -----------------------------------------------------------------
// heavy-edge matching; returns the number of coarse vertices, cmap[v] = coarse id
int match(const Graph& g, Rng& rng, int64_t max_vw, std::vector<int>& cmap) {
    const int n = g.n;
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    std::vector<int> mate(n, -1);
    for (int u : order) {
        if (mate[u] >= 0) continue;
        int best = -1, best_w = -1;
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) {
            const int v = g.adj[p];
            if (mate[v] < 0 && g.ew[p] > best_w && g.vw[u] + g.vw[v] <= max_vw) {
                best = v;
                best_w = g.ew[p];
            }
        }
        if (best >= 0) {
            mate[u] = best;
            mate[best] = u;
        } else {
            mate[u] = u;
        }
    }
    // leftovers: unmatched low-degree neighbours of one vertex pair up among themselves
    // (leaves of a star), isolated vertices pair up with each other -- without this
    // such graphs stop coarsening at once
    for (int c = 0; c < n; ++c) {
        int prev = -1;
        for (int64_t p = g.xadj[c]; p < g.xadj[c + 1]; ++p) {
            const int v = g.adj[p];
            if (mate[v] != v || g.xadj[v + 1] - g.xadj[v] > 2) continue;
            if (prev >= 0 && mate[prev] == prev && g.vw[prev] + g.vw[v] <= max_vw) {
                mate[prev] = v;
                mate[v] = prev;
                prev = -1;
            } else {
                prev = v;
            }
        }
    }
    {
        int prev = -1;
        for (int u = 0; u < n; ++u) {
            if (mate[u] != u || g.xadj[u + 1] != g.xadj[u]) continue;
            if (prev >= 0 && g.vw[prev] + g.vw[u] <= max_vw) {
                mate[prev] = u;
                mate[u] = prev;
                prev = -1;
            } else {
                prev = u;
            }
        }
    }
    cmap.assign(n, -1);
    int nc = 0;
    for (int u = 0; u < n; ++u) {  // in vertex order: neighbours stay near each other
        if (cmap[u] >= 0) continue;
        cmap[u] = nc;
        if (mate[u] != u) cmap[mate[u]] = nc;
        ++nc;
    }
    return nc;
}

Graph contract(const Graph& g, const std::vector<int>& cmap, int nc) {
    Graph c;
    c.n = nc;
    c.vw.assign(nc, 0);
    for (int u = 0; u < g.n; ++u) c.vw[cmap[u]] += g.vw[u];
    // members of every coarse vertex
    std::vector<int> first(nc, -1), second(nc, -1);
    for (int u = 0; u < g.n; ++u) {
        const int cu = cmap[u];
        if (first[cu] < 0) first[cu] = u;
        else second[cu] = u;
    }
    c.xadj.assign(nc + 1, 0);
    c.adj.reserve(g.adj.size());
    c.ew.reserve(g.adj.size());
    std::vector<int64_t> where(nc, -1);
    for (int cu = 0; cu < nc; ++cu) {
        const int64_t row0 = (int64_t)c.adj.size();
        for (int m = 0; m < 2; ++m) {
            const int u = m == 0 ? first[cu] : second[cu];
            if (u < 0) continue;
            for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) {
                const int cv = cmap[g.adj[p]];
                if (cv == cu) continue;
                if (where[cv] >= row0) {
                    c.ew[where[cv]] += g.ew[p];
                } else {
                    where[cv] = (int64_t)c.adj.size();
                    c.adj.push_back(cv);
                    c.ew.push_back(g.ew[p]);
                }
            }
        }
        c.xadj[cu + 1] = (int64_t)c.adj.size();
    }
    return c;
}

// ---- 2-way refinement (boundary FM) ---------------------------------------------------
// indexed max-heap on (gain, vertex)
struct Heap {
    std::vector<int> heap, pos;
    std::vector<int64_t> key;
    explicit Heap(int n) : pos(n, -1), key(n, 0) {}
    bool has(int v) const { return pos[v] >= 0; }
    bool empty() const { return heap.empty(); }
    int top() const { return heap[0]; }
    bool less(int a, int b) const { return key[a] < key[b] || (key[a] == key[b] && a > b); }
    void swap_at(int i, int j) {
        std::swap(heap[i], heap[j]);
        pos[heap[i]] = i;
        pos[heap[j]] = j;
    }
    void up(int i) {
        while (i > 0) {
            const int p = (i - 1) / 2;
            if (!less(heap[p], heap[i])) break;
            swap_at(i, p);
            i = p;
        }
    }
    void down(int i) {
        const int n = (int)heap.size();
        for (;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            if (l < n && less(heap[m], heap[l])) m = l;
            if (r < n && less(heap[m], heap[r])) m = r;
            if (m == i) break;
            swap_at(i, m);
            i = m;
        }
    }
    void push(int v, int64_t k) {
        key[v] = k;
        pos[v] = (int)heap.size();
        heap.push_back(v);
        up(pos[v]);
    }
    void update(int v, int64_t k) {
        const int64_t old = key[v];
        key[v] = k;
        if (k > old) up(pos[v]);
        else down(pos[v]);
    }
    void remove(int v) {
        const int i = pos[v];
        const int last = heap.back();
        heap.pop_back();
        pos[v] = -1;
        if (last != v) {
            heap[i] = last;
            pos[last] = i;
            up(i);
            down(pos[last]);
        }
    }
    void clear() {
        for (int v : heap) pos[v] = -1;
        heap.clear();
    }
};

int64_t cut_of(const Graph& g, const std::vector<uint8_t>& side) {
    int64_t c = 0;
    for (int u = 0; u < g.n; ++u)
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p)
            if (side[u] != side[g.adj[p]]) c += g.ew[p];
    return c / 2;
}

// tw[s] = target weight of side s, cap[s] = the most it may hold
void fm_refine(const Graph& g, std::vector<uint8_t>& side, const int64_t tw[2], const int64_t cap[2],
               int passes) {
    const int n = g.n;
    std::vector<int64_t> idw(n), edw(n);
    std::vector<uint8_t> locked(n);
    Heap heaps[2] = {Heap(n), Heap(n)};
    std::vector<int> moved;
    for (int pass = 0; pass < passes; ++pass) {
        int64_t w[2] = {0, 0};
        for (int u = 0; u < n; ++u) {
            w[side[u]] += g.vw[u];
            int64_t i = 0, e = 0;
            for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p)
                (side[g.adj[p]] == side[u] ? i : e) += g.ew[p];
            idw[u] = i;
            edw[u] = e;
        }
        heaps[0].clear();
        heaps[1].clear();
        std::fill(locked.begin(), locked.end(), 0);
        const bool start_feasible = w[0] <= cap[0] && w[1] <= cap[1];
        for (int u = 0; u < n; ++u)
            if (edw[u] > 0 || !start_feasible) heaps[side[u]].push(u, edw[u] - idw[u]);
        moved.clear();
        int64_t cur = 0, best = 0;  // cut change so far (negative = better)
        int64_t best_over = std::max<int64_t>(0, w[0] - cap[0]) + std::max<int64_t>(0, w[1] - cap[1]);
        size_t best_len = 0;
        const int limit = std::max(64, std::min(2000, n / 50));
        int since_best = 0;
        while (since_best < limit) {
            // which side gives a vertex: an overweight side must; otherwise the better gain
            int from = -1;
            if (w[0] > cap[0]) from = 0;
            else if (w[1] > cap[1]) from = 1;
            else {
                const bool a = !heaps[0].empty() && w[1] + g.vw[heaps[0].top()] <= cap[1];
                const bool b = !heaps[1].empty() && w[0] + g.vw[heaps[1].top()] <= cap[0];
                if (a && b) {
                    const int64_t ka = heaps[0].key[heaps[0].top()], kb = heaps[1].key[heaps[1].top()];
                    from = ka > kb ? 0 : kb > ka ? 1 : (w[0] - tw[0] >= w[1] - tw[1] ? 0 : 1);
                } else if (a) from = 0;
                else if (b) from = 1;
            }
            if (from < 0 || heaps[from].empty()) break;
            const int u = heaps[from].top();
            heaps[from].remove(u);
            const int to = from ^ 1;
            locked[u] = 1;
            cur -= edw[u] - idw[u];
            w[from] -= g.vw[u];
            w[to] += g.vw[u];
            side[u] = (uint8_t)to;
            std::swap(idw[u], edw[u]);
            moved.push_back(u);
            for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) {
                const int v = g.adj[p];
                if (side[v] == to) {  // was external to v, now internal
                    idw[v] += g.ew[p];
                    edw[v] -= g.ew[p];
                } else {
                    idw[v] -= g.ew[p];
                    edw[v] += g.ew[p];
                }
                if (locked[v]) continue;
                Heap& h = heaps[side[v]];
                if (h.has(v)) {
                    if (edw[v] > 0 || !start_feasible) h.update(v, edw[v] - idw[v]);
                    else h.remove(v);
                } else if (edw[v] > 0) {
                    h.push(v, edw[v] - idw[v]);
                }
            }
            const int64_t over = std::max<int64_t>(0, w[0] - cap[0]) + std::max<int64_t>(0, w[1] - cap[1]);
            if (over < best_over || (over == best_over && cur < best)) {
                best = cur;
                best_over = over;
                best_len = moved.size();
                since_best = 0;
            } else {
                ++since_best;
            }
        }
        for (size_t i = moved.size(); i > best_len; --i) side[moved[i - 1]] ^= 1;  // roll back
        if (best_len == 0) break;
    }
}

// greedy graph growing: side 0 grows from a random seed, always taking the frontier
// vertex whose move cuts least, until it holds tw[0]
void grow(const Graph& g, Rng& rng, const int64_t tw[2], std::vector<uint8_t>& side) {
    const int n = g.n;
    side.assign(n, 1);
    std::vector<int64_t> deg(n, 0);
    for (int u = 0; u < n; ++u)
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) deg[u] += g.ew[p];
    Heap front(n);  // key = 2 * (weight towards side 0) - degree
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    size_t next_seed = 0;
    int64_t w0 = 0;
    while (w0 < tw[0]) {
        int u = -1;
        if (!front.empty()) {
            u = front.top();
            front.remove(u);
        } else {  // start, or the component is exhausted
            while (next_seed < order.size() && side[order[next_seed]] == 0) ++next_seed;
            if (next_seed == order.size()) break;
            u = order[next_seed++];
        }
        if (w0 > 0 && w0 + g.vw[u] - tw[0] > tw[0] - w0) break;  // closer to the target without it
        side[u] = 0;
        w0 += g.vw[u];
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p) {
            const int v = g.adj[p];
            if (side[v] == 0) continue;
            if (front.has(v)) front.update(v, front.key[v] + 2 * g.ew[p]);
            else front.push(v, 2 * (int64_t)g.ew[p] - deg[v]);
        }
    }
}

// multilevel bisection of g: side[v] in {0, 1}, side 0 gets the fraction f0 of the weight
void bisect(const Graph& g0, double f0, double ub, Rng& rng, std::vector<uint8_t>& side) {
    const int64_t total = g0.total_vw();
    int64_t tw[2] = {(int64_t)(f0 * (double)total), 0};
    tw[1] = total - tw[0];
    std::vector<Graph> levels;
    std::vector<std::vector<int>> cmaps;
    const Graph* g = &g0;
    const int coarsen_to = 160;
    while (g->n > coarsen_to) {
        std::vector<int> cmap;
        const int64_t max_vw = std::max<int64_t>(1, (int64_t)(1.5 * (double)total / coarsen_to));
        const int nc = match(*g, rng, max_vw, cmap);
        if (nc > 0.95 * g->n) break;  // nothing left to match (e.g. stars)
        levels.push_back(contract(*g, cmap, nc));
        cmaps.push_back(std::move(cmap));
        g = &levels.back();
    }
    // heavy coarse vertices make exact balance impossible up there: caps per level
    auto caps = [&](const Graph& gr, int64_t cap[2]) {
        int64_t maxv = 0;
        for (int64_t x : gr.vw) maxv = std::max(maxv, x);
        // a vertex heavier than the slack (a hub, a big coarse vertex) still has to fit somewhere
        for (int s = 0; s < 2; ++s) cap[s] = std::max((int64_t)(ub * (double)tw[s]), tw[s] + maxv / 2);
    };
    int64_t cap[2];
    caps(*g, cap);
    std::vector<uint8_t> best_side;
    int64_t best_cut = INT64_MAX, best_over = INT64_MAX;
    const int trials = g->n <= 2 ? 1 : 10;
    for (int t = 0; t < trials; ++t) {
        std::vector<uint8_t> s;
        grow(*g, rng, tw, s);
        fm_refine(*g, s, tw, cap, 6);
        int64_t w[2] = {0, 0};
        for (int u = 0; u < g->n; ++u) w[s[u]] += g->vw[u];
        const int64_t over = std::max<int64_t>(0, w[0] - cap[0]) + std::max<int64_t>(0, w[1] - cap[1]);
        const int64_t c = cut_of(*g, s);
        if (over < best_over || (over == best_over && c < best_cut)) {
            best_over = over;
            best_cut = c;
            best_side = std::move(s);
        }
    }
    side = std::move(best_side);
    for (int l = (int)levels.size() - 1; l >= 0; --l) {  // project + refine
        const Graph& fine = l == 0 ? g0 : levels[l - 1];
        std::vector<uint8_t> fs(fine.n);
        for (int u = 0; u < fine.n; ++u) fs[u] = side[cmaps[l][u]];
        side = std::move(fs);
        caps(fine, cap);
        fm_refine(fine, side, tw, cap, fine.n > 200000 ? 3 : 5);
    }
}

Graph induced(const Graph& g, const std::vector<uint8_t>& side, int s, std::vector<int>& ids) {
    std::vector<int> local(g.n, -1);
    ids.clear();
    for (int u = 0; u < g.n; ++u)
        if (side[u] == s) {
            local[u] = (int)ids.size();
            ids.push_back(u);
        }
    Graph h;
    h.n = (int)ids.size();
    h.xadj.assign(h.n + 1, 0);
    h.vw.resize(h.n);
    for (int i = 0; i < h.n; ++i) {
        const int u = ids[i];
        h.vw[i] = g.vw[u];
        for (int64_t p = g.xadj[u]; p < g.xadj[u + 1]; ++p)
            if (local[g.adj[p]] >= 0) {
                h.adj.push_back(local[g.adj[p]]);
                h.ew.push_back(g.ew[p]);
            }
        h.xadj[i + 1] = (int64_t)h.adj.size();
    }
    return h;
}

void recurse(const Graph& g, int k, int first_part, double ub, Rng& rng, const std::vector<int>& ids,
             int32_t* part) {
    if (k <= 1 || g.n == 0) {
        for (int u = 0; u < g.n; ++u) part[ids[u]] = first_part;
        return;
    }
    const int k0 = (k + 1) / 2, k1 = k - k0;
    std::vector<uint8_t> side;
    bisect(g, (double)k0 / (double)k, ub, rng, side);
    for (int s = 0; s < 2; ++s) {
        std::vector<int> sub_ids;
        Graph h = induced(g, side, s, sub_ids);
        std::vector<int> global(sub_ids.size());
        for (size_t i = 0; i < sub_ids.size(); ++i) global[i] = ids[sub_ids[i]];
        recurse(h, s == 0 ? k0 : k1, s == 0 ? first_part : first_part + k0, ub, rng, global, part);
    }
}

thread_local std::string g_perr;

}  // namespace

extern "C" {

int mxp_partition(int32_t n_vars, int32_t n_factors, const int32_t* factor_rowptr, const int32_t* edge_var,
                  int32_t k, double imbalance, uint64_t seed, int32_t* part) {
    if (n_vars < 0 || n_factors < 0 || k < 1 || !part || (n_factors && (!factor_rowptr || !edge_var))) {
        g_perr = "mxp_partition: bad argument";
        return -1;
    }
    if (imbalance < 1.0) imbalance = 1.0;
    try {
        if (k == 1 || n_vars == 0) {
            for (int v = 0; v < n_vars; ++v) part[v] = 0;
            return 0;
        }
        // variable graph: clique expansion of small scopes, a ring for big ones
        std::vector<int> eu, ev, w;
        std::vector<int64_t> vw(n_vars, 1);
        for (int f = 0; f < n_factors; ++f) {
            const int a = factor_rowptr[f], b = factor_rowptr[f + 1], ar = b - a;
            for (int i = a; i < b; ++i) {
                if (edge_var[i] < 0 || edge_var[i] >= n_vars) {
                    g_perr = "mxp_partition: variable index out of range";
                    return -1;
                }
                vw[edge_var[i]] += 1;  // weight = 1 + degree: balances variables and edges
            }
            if (ar <= 6) {
                for (int i = a; i < b; ++i)
                    for (int j = a; j < b; ++j)
                        if (i != j) {
                            eu.push_back(edge_var[i]);
                            ev.push_back(edge_var[j]);
                            w.push_back(1);
                        }
            } else {
                for (int i = a; i < b; ++i) {
                    const int j = i + 1 < b ? i + 1 : a;
                    eu.push_back(edge_var[i]);
                    ev.push_back(edge_var[j]);
                    w.push_back(1);
                    eu.push_back(edge_var[j]);
                    ev.push_back(edge_var[i]);
                    w.push_back(1);
                }
            }
        }
        Graph g = build_csr(n_vars, eu, ev, w, std::move(vw));
        eu = std::vector<int>();
        ev = std::vector<int>();
        w = std::vector<int>();
        int depth = 0;
        for (int x = 1; x < k; x *= 2) ++depth;
        const double ub = 1.0 + (imbalance - 1.0) / std::max(1, depth);
        Rng rng(seed);
        std::vector<int> ids(n_vars);
        std::iota(ids.begin(), ids.end(), 0);
        recurse(g, k, 0, ub, rng, ids, part);
        // no rank may end up without a variable (possible when one vertex outweighs whole parts)
        std::vector<int> count(k, 0);
        for (int v = 0; v < n_vars; ++v) count[part[v]]++;
        for (int p = 0; p < k && n_vars >= k; ++p) {
            if (count[p]) continue;
            const int donor = (int)(std::max_element(count.begin(), count.end()) - count.begin());
            int pick = -1;  // the donor's lightest variable
            for (int v = 0; v < n_vars; ++v)
                if (part[v] == donor && (pick < 0 || g.vw[v] < g.vw[pick])) pick = v;
            part[pick] = p;
            count[donor]--;
            count[p]++;
        }
        return 0;
    } catch (const std::bad_alloc&) {
        g_perr = "mxp_partition: out of memory";
        return -4;
    } catch (const std::exception& ex) {
        g_perr = ex.what();
        return -1;
    }
}

const char* mxp_last_error(void) { return g_perr.c_str(); }

}  // extern "C"
