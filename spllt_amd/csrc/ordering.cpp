// Built-in fill-reducing ordering: recursive nested dissection with
// level-structure (BFS) vertex separators, greedy minimum degree on the leaves.
//
// This replaces the Metis call SpLLT makes through SSIDS
// (reference src/spllt_analyse_mod.F90:109-131); neither SPRAL nor Metis is
// available to this build.  The reference pins no ordering, so
// any valid permutation is admissible; quality only changes nnz(L)/flops.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "symbolic.hpp"

namespace spx {
namespace {

struct NDWork {
  const std::vector<int64_t>& xadj;
  const std::vector<int>& adj;
  std::vector<int> region;  // region id per vertex (-1 = already numbered)
  std::vector<int> dist;    // BFS scratch
  std::vector<int> queue;
  std::vector<int>& order;
  int next_region = 1;
  NDWork(const std::vector<int64_t>& x, const std::vector<int>& a, std::vector<int>& o)
      : xadj(x), adj(a), region(o.size(), 0), dist(o.size(), -1), order(o) {}
};

// BFS restricted to `reg`; fills w.queue with the visit order and w.dist with
// levels.  Returns the number of levels.
int bfs(NDWork& w, int root, int reg) {
  w.queue.clear();
  w.queue.push_back(root);
  w.dist[root] = 0;
  int nlev = 1;
  for (size_t h = 0; h < w.queue.size(); ++h) {
    int v = w.queue[h];
    for (int64_t e = w.xadj[v]; e < w.xadj[v + 1]; ++e) {
      int u = w.adj[e];
      if (w.region[u] == reg && w.dist[u] < 0) {
        w.dist[u] = w.dist[v] + 1;
        nlev = w.dist[u] + 1;
        w.queue.push_back(u);
      }
    }
  }
  return nlev;
}

void clear_dist(NDWork& w) {
  for (int v : w.queue) w.dist[v] = -1;
}

// Greedy minimum degree on a small induced subgraph (explicit elimination
// graph; only used for leaves of at most a few dozen vertices).
void leaf_order(NDWork& w, const std::vector<int>& verts, int reg, int lo) {
  int k = (int)verts.size();
  std::vector<int> loc(k);
  // local ids via dist scratch
  for (int i = 0; i < k; ++i) w.dist[verts[i]] = i;
  std::vector<std::vector<char>> a(k, std::vector<char>(k, 0));
  for (int i = 0; i < k; ++i) {
    int v = verts[i];
    for (int64_t e = w.xadj[v]; e < w.xadj[v + 1]; ++e) {
      int u = w.adj[e];
      if (w.region[u] == reg) a[i][w.dist[u]] = 1;
    }
  }
  for (int i = 0; i < k; ++i) w.dist[verts[i]] = -1;
  std::vector<char> done(k, 0);
  for (int step = 0; step < k; ++step) {
    int best = -1, bdeg = 1 << 30;
    for (int i = 0; i < k; ++i) {
      if (done[i]) continue;
      int d = 0;
      for (int j = 0; j < k; ++j) d += (!done[j] && a[i][j]);
      if (d < bdeg) { bdeg = d; best = i; }
    }
    done[best] = 1;
    w.order[verts[best]] = lo + step;
    w.region[verts[best]] = -1;
    // form the clique of its remaining neighbours
    std::vector<int> nb;
    for (int j = 0; j < k; ++j)
      if (!done[j] && a[best][j]) nb.push_back(j);
    for (int x : nb)
      for (int y : nb)
        if (x != y) a[x][y] = 1;
  }
}

struct Item {
  std::vector<int> verts;
  int lo;
};

}  // namespace

void nested_dissection(int n, const std::vector<int64_t>& xadj, const std::vector<int>& adj,
                       int leaf, std::vector<int>& order) {
  order.assign(n, -1);
  if (n == 0) return;
  if (leaf < 1) leaf = 1;
  NDWork w(xadj, adj, order);
  std::vector<Item> stack;
  {
    Item all;
    all.verts.resize(n);
    std::iota(all.verts.begin(), all.verts.end(), 0);
    all.lo = 0;
    stack.push_back(std::move(all));
  }
  while (!stack.empty()) {
    Item it = std::move(stack.back());
    stack.pop_back();
    const int nv = (int)it.verts.size();
    if (nv == 0) continue;
    const int reg = w.next_region++;
    for (int v : it.verts) w.region[v] = reg;
    if (nv <= leaf) {
      leaf_order(w, it.verts, reg, it.lo);
      continue;
    }
    // Connected component of the first vertex.
    int root = it.verts[0];
    int nlev = bfs(w, root, reg);
    if ((int)w.queue.size() < nv) {
      // disconnected: split off this component, handle the rest separately
      Item comp, rest;
      comp.verts = w.queue;
      comp.lo = it.lo;
      clear_dist(w);
      const int creg = w.next_region++;
      for (int v : comp.verts) w.region[v] = creg;
      rest.lo = it.lo + (int)comp.verts.size();
      for (int v : it.verts)
        if (w.region[v] == reg) rest.verts.push_back(v);
      stack.push_back(std::move(rest));
      stack.push_back(std::move(comp));
      continue;
    }
    // pseudo-peripheral root: restart from a minimum-degree vertex of the last
    // level while the eccentricity grows
    for (int iter = 0; iter < 4; ++iter) {
      int far = w.queue.back(), best = far;
      int64_t bdeg = INT64_MAX;
      for (size_t q = w.queue.size(); q-- > 0;) {
        int v = w.queue[q];
        if (w.dist[v] != w.dist[far]) break;
        int64_t d = w.xadj[v + 1] - w.xadj[v];
        if (d < bdeg) { bdeg = d; best = v; }
      }
      clear_dist(w);
      int nl2 = bfs(w, best, reg);
      bool grew = nl2 > nlev;
      nlev = nl2;
      root = best;
      if (!grew) break;
    }
    if (nlev < 3) {
      // (near-)clique: no useful separator
      clear_dist(w);
      leaf_order(w, it.verts, reg, it.lo);
      continue;
    }
    // level sizes and the separator level: smallest level with a 30/70 balance,
    // falling back to the weighted median
    std::vector<int> lsz(nlev, 0);
    for (int v : w.queue) lsz[w.dist[v]]++;
    int best = -1, best_med = 1;
    int bsz = 1 << 30, bmin = -1;
    int below = 0;
    for (int l = 0; l < nlev; ++l) {
      int above = nv - below - lsz[l];
      if (l > 0 && l < nlev - 1) {
        int mn = std::min(below, above);
        double bal = (double)mn / std::max(1, nv - lsz[l]);
        if (bal >= 0.3 && lsz[l] < bsz) { bsz = lsz[l]; best = l; }
        if (mn > bmin) { bmin = mn; best_med = l; }
      }
      below += lsz[l];
    }
    if (best < 0) best = best_med;
    // Partition: A = levels < best, S = level best, B = levels > best.
    Item A, B;
    std::vector<int> S;
    for (int v : w.queue) {
      int d = w.dist[v];
      if (d < best) A.verts.push_back(v);
      else if (d > best) B.verts.push_back(v);
      else S.push_back(v);
    }
    // thin the separator: a separator vertex with no neighbour in B joins A
    {
      std::vector<int> S2;
      for (int v : S) {
        bool touchB = false;
        for (int64_t e = w.xadj[v]; e < w.xadj[v + 1] && !touchB; ++e) {
          int u = w.adj[e];
          touchB = (w.region[u] == reg && w.dist[u] > best);
        }
        if (touchB) S2.push_back(v);
        else A.verts.push_back(v);
      }
      S.swap(S2);
    }
    clear_dist(w);
    A.lo = it.lo;
    B.lo = it.lo + (int)A.verts.size();
    int slo = B.lo + (int)B.verts.size();
    for (size_t i = 0; i < S.size(); ++i) {
      w.order[S[i]] = slo + (int)i;
      w.region[S[i]] = -1;
    }
    stack.push_back(std::move(B));
    stack.push_back(std::move(A));
  }
}

}  // namespace spx
