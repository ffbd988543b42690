// Symbolic analysis + SpLLT tile layout (see symbolic.hpp for the contract).
#include "symbolic.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>

namespace spx {
namespace {

// Symmetrised adjacency (no diagonal, both triangles) of a CSC-lower pattern.
void build_adjacency(int n, const int64_t* ptr, const int* row, std::vector<int64_t>& xadj,
                     std::vector<int>& adj) {
  xadj.assign(n + 1, 0);
  for (int j = 0; j < n; ++j)
    for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
      int i = row[e];
      if (i == j) continue;
      xadj[i + 1]++;
      xadj[j + 1]++;
    }
  for (int j = 0; j < n; ++j) xadj[j + 1] += xadj[j];
  adj.resize(xadj[n]);
  std::vector<int64_t> pos(xadj.begin(), xadj.end() - 1);
  for (int j = 0; j < n; ++j)
    for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
      int i = row[e];
      if (i == j) continue;
      adj[pos[i]++] = j;
      adj[pos[j]++] = i;
    }
}

// Elimination tree of the matrix permuted by `order` (Liu's algorithm with
// path compression).  parent is indexed by pivot position.
void etree(int n, const std::vector<int64_t>& xadj, const std::vector<int>& adj,
           const std::vector<int>& order, const std::vector<int>& porder,
           std::vector<int>& parent) {
  parent.assign(n, -1);
  std::vector<int> anc(n, -1);
  for (int j = 0; j < n; ++j) {
    int v = porder[j];
    for (int64_t e = xadj[v]; e < xadj[v + 1]; ++e) {
      int i = order[adj[e]];
      if (i >= j) continue;
      int r = i;
      while (anc[r] != -1 && anc[r] != j) {
        int nx = anc[r];
        anc[r] = j;
        r = nx;
      }
      if (anc[r] == -1) {
        anc[r] = j;
        parent[r] = j;
      }
    }
  }
}

// Postorder of a forest given by parent[] (children visited in increasing
// index order).  post[k] = k-th vertex visited.
void postorder(int n, const std::vector<int>& parent, std::vector<int>& post) {
  std::vector<int> head(n + 1, -1), next(n, -1);
  for (int j = n - 1; j >= 0; --j) {
    int p = parent[j] < 0 ? n : parent[j];
    next[j] = head[p];
    head[p] = j;
  }
  post.clear();
  post.reserve(n);
  std::vector<int> stack;
  for (int r = head[n]; r != -1; r = next[r]) {
    stack.push_back(r);
    while (!stack.empty()) {
      int v = stack.back();
      int c = head[v];
      if (c != -1) {
        head[v] = next[c];
        stack.push_back(c);
      } else {
        post.push_back(v);
        stack.pop_back();
      }
    }
  }
}

// Column counts of L (diagonal included) for a matrix whose elimination order
// is already a postorder of its etree (Gilbert-Ng-Peyton skeleton counting).
void colcounts(int n, const std::vector<int64_t>& xadj, const std::vector<int>& adj,
               const std::vector<int>& order, const std::vector<int>& porder,
               const std::vector<int>& parent, std::vector<int64_t>& cc) {
  std::vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), uf(n);
  std::vector<int64_t> delta(n, 0);
  std::iota(uf.begin(), uf.end(), 0);
  // first descendant of every vertex; leaves get delta = 1
  for (int j = 0; j < n; ++j) {
    delta[j] = (first[j] == -1) ? 1 : 0;
    int f = (first[j] == -1) ? j : first[j];
    first[j] = f;
    int p = parent[j];
    if (p != -1 && first[p] == -1) first[p] = f;
  }
  auto find = [&](int x) {
    int r = x;
    while (uf[r] != r) r = uf[r];
    while (uf[x] != r) {
      int nx = uf[x];
      uf[x] = r;
      x = nx;
    }
    return r;
  };
  for (int j = 0; j < n; ++j) {
    if (parent[j] != -1) delta[parent[j]]--;
    int v = porder[j];
    for (int64_t e = xadj[v]; e < xadj[v + 1]; ++e) {
      int i = order[adj[e]];
      if (i <= j) continue;
      if (first[j] > maxfirst[i]) {  // j is a leaf of the row subtree of i
        maxfirst[i] = first[j];
        int jprev = prevleaf[i];
        prevleaf[i] = j;
        delta[j]++;
        if (jprev != -1) delta[find(jprev)]--;
      }
    }
    if (parent[j] != -1) uf[j] = parent[j];
  }
  cc = delta;
  for (int j = 0; j < n; ++j)
    if (parent[j] != -1) cc[parent[j]] += cc[j];
}

}  // namespace

// ---------------------------------------------------------------------------
// spllt_prune_tree restated (reference src/spllt_analyse_mod.F90:806-987).
// Starting from the virtual root, the layer l0 is refined by replacing its
// heaviest non-leaf entry with its children until a greedy mapping of the
// layer's subtrees onto nth workers is >= 90 % balanced (or the layer is too
// large); the children of the final layer become pruned-subtree roots.
// small[node] = 0 normal, 1 subtree root, -(root+1) inside subtree of `root`.
// ---------------------------------------------------------------------------
void prune_tree(Symbolic& S, int nth) {
  const int nn = S.nnodes;
  if (nth < 1) nth = 1;
  auto nchild = [&](int v) { return S.child_ptr[v + 1] - S.child_ptr[v]; };
  auto mark = [&](int c) {
    for (int v = S.least_desc[c]; v <= c; ++v) S.small[v] = -(c + 1);
    S.small[c] = 1;
  };
  const double lim = nth * std::max(2.0, std::pow(std::log((double)nth) / std::log(2.0), 2));
  double smallth = 0.01;
  std::vector<int> lzero;
  std::vector<int64_t> lw;
  int nlz = 0;
  for (;;) {  // label 10 in the reference: restart with a smaller threshold
    bool restart = false;
    std::fill(S.small.begin(), S.small.end(), 0);
    const int64_t tot = S.weight[nn];
    lzero.assign(1, nn);
    lw.assign(1, -S.weight[nn]);
    nlz = 1;
    int totleaves = 0;
    for (int v = 0; v <= nn; ++v) totleaves += (nchild(v) == 0);
    int leaves = 0;
    for (;;) {  // godown
      if (nlz <= 0) break;
      if (nlz > lim) break;
      // ascending sort of (negated) weights, carrying the node ids
      std::vector<int> idx(nlz);
      std::iota(idx.begin(), idx.end(), 0);
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lw[a] < lw[b]; });
      {
        std::vector<int> z(nlz);
        std::vector<int64_t> zw(nlz);
        for (int i = 0; i < nlz; ++i) { z[i] = lzero[idx[i]]; zw[i] = lw[idx[i]]; }
        lzero.swap(z);
        lw.swap(zw);
      }
      std::vector<int64_t> pw(nth, 0);
      for (int i = 0; i < nlz; ++i) {
        int p = (int)(std::min_element(pw.begin(), pw.end()) - pw.begin());
        pw[p] += std::llabs(lw[i]);
      }
      int64_t mx = *std::max_element(pw.begin(), pw.end());
      int64_t mn = *std::min_element(pw.begin(), pw.end());
      double rm = mx > 0 ? (double)mn / (double)mx : 1.0;
      if (rm > 0.9 && nlz >= nth) break;
      bool found = false, bottom = false;
      int nrep = -1;
      for (;;) {  // findn
        if (leaves == totleaves) { bottom = true; break; }
        if (leaves == nlz) {
          if (nlz >= lim) { bottom = true; break; }
          smallth /= 2.0;
          if (smallth < 1e-4) { bottom = true; break; }
          restart = true;
          break;
        }
        nrep = lzero[leaves];
        for (int e = S.child_ptr[nrep]; e < S.child_ptr[nrep + 1]; ++e) {
          int c = S.child_idx[e];
          if ((double)S.weight[c] > smallth * (double)tot) {
            found = true;
            lzero.push_back(c);
            lw.push_back(-S.weight[c]);
            nlz++;
          } else {
            mark(c);
          }
        }
        if (found) break;
        leaves++;
      }
      if (restart || bottom) break;
      // drop the replaced node: overwrite it with the last entry
      lzero[leaves] = lzero[nlz - 1];
      lw[leaves] = lw[nlz - 1];
      lzero.pop_back();
      lw.pop_back();
      nlz--;
    }
    if (!restart) break;
  }
  for (int i = 0; i < nlz; ++i) {
    int v = lzero[i];
    for (int e = S.child_ptr[v]; e < S.child_ptr[v + 1]; ++e) mark(S.child_idx[e]);
  }
}

// Modelled time of one node on one MI355X (seconds): its flops at the rate the update kernels reach
// at its K (= its width; sustained 64-tile figures, DESIGN section 4: 28 / 40 / 50 / 58 / 62 TFLOP/s
// at K = 64 ... 1024, i.e. ~68 K / (K + 90)) plus the scatter-add of its generated element at the
// chip's rate for fp64 atomics (262 G adds/s).  The leaves of a nested-dissection
// tree run at a tenth of the rate of the separators near the top: mapping subtrees by flops alone
// gave the rank with the bushier half 30 % more time (flan_like, w = 2: 409 vs 316 ms).
static double node_time(const Symbolic& S, int s) {
  const double m = S.nrow(s), n = S.ncol(s);
  double fl = 0;
  for (int j = 1; j <= (int)n; ++j) fl += (m - n + j) * (m - n + j);
  const double rate = 68e12 * n / (n + 90.0);
  return fl / rate + 0.5 * (m - n) * (m - n) / 262e9;
}

// Mapping of the assembly tree onto `nranks` ranks: whole subtrees per rank, the nodes above them
// are the top tree (-1).  The subtree roots start as the children of the virtual root; the largest
// root of the most loaded rank is replaced by its children (it joins the top tree) and the roots are
// dealt again, largest first to the least loaded rank, until the loads are balanced to 0.9 -- what
// spllt_prune_tree does with its subtrees (reference src/spllt_analyse_mod.F90:806-987, the 0.9 at
// :895-897) -- or the roots get too small to matter.  Loads are MODELLED TIMES (node_time), not
// flops (SPLLT_OWNER_MODEL=flops: the symbolic flop count, the round-3 behaviour of the weights).
void assign_owners(const Symbolic& S, int nranks, std::vector<int>& owner) {
  const int nn = S.nnodes;
  owner.assign(nn, -1);
  if (nranks < 1) nranks = 1;
  std::vector<std::vector<int>> kids(nn + 1);
  for (int s = 0; s < nn; ++s) kids[std::min(S.sparent[s], nn)].push_back(s);
  const char* model = std::getenv("SPLLT_OWNER_MODEL");
  const bool by_flops = model && std::string(model) == "flops";
  // subtree loads (postorder: children before parents)
  std::vector<double> load(nn + 1, 0.0);
  for (int s = 0; s < nn; ++s) {
    if (by_flops) {
      load[s] = (double)S.weight[s];     // (already the sum over the subtree)
    } else {
      load[s] += node_time(S, s);
      load[std::min(S.sparent[s], nn)] += load[s];
    }
  }
  double total = 0;
  for (int r : kids[nn]) total += load[r];
  std::vector<int> roots = kids[nn];
  std::vector<int> where;
  std::vector<double> rl;
  auto deal = [&]() {
    std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return load[a] > load[b]; });
    where.assign(roots.size(), 0);
    rl.assign((size_t)nranks, 0.0);
    for (size_t i = 0; i < roots.size(); ++i) {
      int best = 0;
      for (int r = 1; r < nranks; ++r)
        if (rl[(size_t)r] < rl[(size_t)best]) best = r;
      where[i] = best;
      rl[(size_t)best] += load[roots[i]];
    }
  };
  const double min_piece = total / (64.0 * nranks);      // roots below this stay whole
  for (int it = 0; it < 64 * nranks; ++it) {
    deal();
    if (nranks == 1) break;
    int rmax = 0;
    double lo = rl[0];
    for (int r = 0; r < nranks; ++r) {
      if (rl[(size_t)r] > rl[(size_t)rmax]) rmax = r;
      lo = std::min(lo, rl[(size_t)r]);
    }
    if ((int)roots.size() >= nranks && lo >= 0.9 * rl[(size_t)rmax]) break;
    // the largest root of the most loaded rank that can be split (with fewer roots than ranks: the largest of all)
    int pick = -1;
    for (size_t i = 0; i < roots.size(); ++i) {
      if ((int)roots.size() >= nranks && where[i] != rmax) continue;
      if (kids[roots[i]].empty() || load[roots[i]] < min_piece) continue;
      if (pick < 0 || load[roots[i]] > load[roots[(size_t)pick]]) pick = (int)i;
    }
    if (pick < 0) break;
    const int top = roots[(size_t)pick];
    roots.erase(roots.begin() + pick);
    for (int c : kids[top]) roots.push_back(c);             // `top` joins the top tree (owner stays -1)
  }
  if (nranks == 1) {
    std::fill(owner.begin(), owner.end(), 0);
    return;
  }
  for (size_t i = 0; i < roots.size(); ++i)
    for (int v = S.least_desc[roots[i]]; v <= roots[i]; ++v) owner[v] = where[i];
}

int analyse(int n, const int64_t* ptr, const int* row, const int* user_order,
            const SymOptions& opt, Symbolic& S) {
  S = Symbolic();
  S.n = n;
  S.nb = opt.nb < 1 ? 256 : opt.nb;
  if (n <= 0) return n == 0 ? 0 : -10;
  S.nnzA = ptr[n];
  for (int j = 0; j < n; ++j)
    for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e)
      if (row[e] < 0 || row[e] >= n) return -10;

  std::vector<int64_t> xadj;
  std::vector<int> adj;
  build_adjacency(n, ptr, row, xadj, adj);

  // ---- pivot order -------------------------------------------------------
  std::vector<int> order(n), porder(n);
  if (user_order) {
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) {
      int p = user_order[i];
      if (p < 0 || p >= n || seen[p]) return -10;
      seen[p] = 1;
      order[i] = p;
    }
    S.ordering = "user";
  } else {
    nested_dissection(n, xadj, adj, opt.nd_leaf, order);
    S.ordering = "nd-bfs";
  }
  for (int i = 0; i < n; ++i) porder[order[i]] = i;

  // ---- etree, postorder relabel -----------------------------------------
  std::vector<int> parent, post;
  etree(n, xadj, adj, order, porder, parent);
  postorder(n, parent, post);
  {
    std::vector<int> newpos(n);  // old position -> new position
    for (int k = 0; k < n; ++k) newpos[post[k]] = k;
    std::vector<int> np(n, -1);
    for (int j = 0; j < n; ++j) np[newpos[j]] = parent[j] < 0 ? -1 : newpos[parent[j]];
    parent.swap(np);
    for (int i = 0; i < n; ++i) order[i] = newpos[order[i]];
    for (int i = 0; i < n; ++i) porder[order[i]] = i;
  }
  std::vector<int64_t> cc;
  colcounts(n, xadj, adj, order, porder, parent, cc);

  // ---- maximal exact supernodes -----------------------------------------
  std::vector<int> sn_first;  // first column of each supernode
  std::vector<int> sn_of(n);
  for (int j = 0; j < n; ++j) {
    bool join = j > 0 && parent[j - 1] == j && cc[j] == cc[j - 1] - 1;
    if (!join) sn_first.push_back(j);
    sn_of[j] = (int)sn_first.size() - 1;
  }
  int ns = (int)sn_first.size();
  sn_first.push_back(n);
  std::vector<int> sp(ns, -1);
  std::vector<int64_t> sm(ns), sncol(ns);
  for (int s = 0; s < ns; ++s) {
    int last = sn_first[s + 1] - 1;
    sp[s] = parent[last] < 0 ? -1 : sn_of[parent[last]];
    sncol[s] = sn_first[s + 1] - sn_first[s];
    sm[s] = cc[sn_first[s]];
  }

  // ---- relaxed amalgamation ------------------------------------------------
  // A node is merged into its parent when
  //   * that creates no fill, or
  //   * both have fewer than nemin columns (the rule SSIDS applies for SpLLT,
  //     reference src/spllt_analyse_mod.F90:112-116), or
  //   * the explicit zeros the merge introduces stay a small fraction of the
  //     merged trapezoid (separators produced by nested dissection split into
  //     chains of supernodes whose row structures differ by a few rows; gluing
  //     them back costs a few per cent of storage and removes most of the
  //     sequential panel steps of the factorization).
  const int nemin = opt.nemin < 1 ? 32 : opt.nemin;
  std::vector<int> rep(ns);
  std::iota(rep.begin(), rep.end(), 0);
  std::vector<int64_t> sm0(sm);            // original row counts
  std::vector<double> nz_true(ns);         // structural nonzeros held by the (merged) node
  for (int s = 0; s < ns; ++s) nz_true[s] = (double)sncol[s] * sm[s] - 0.5 * sncol[s] * (sncol[s] - 1);
  for (int s = 0; s < ns; ++s) {  // supernodes are already in postorder
    int p = sp[s];
    if (p < 0) continue;
    const bool exact = (sm[s] - sncol[s] == sm0[p]) && sncol[p] == (sn_first[p + 1] - sn_first[p]);
    const bool tiny = (sncol[s] < nemin && sncol[p] < nemin);
    // merged trapezoid: columns of both, rows = own columns + rows below the parent
    const int64_t below = sm[p] - sncol[p];
    const int64_t nc = sncol[p] + sncol[s];
    const double nz_merged = (double)nc * (nc + below) - 0.5 * nc * (nc - 1);
    const double zeros = nz_merged - (nz_true[s] + nz_true[p]);
    const bool relaxed = opt.relax > 0.0 && zeros <= opt.relax * nz_merged && sm[s] - sncol[s] <= nc - sncol[s] + below;
    if (exact || tiny || relaxed) {
      rep[s] = p;
      sncol[p] = nc;
      sm[p] = nc + below;
      nz_true[p] += nz_true[s];
    }
  }
  auto findrep = [&](int s) {
    int r = s;
    while (rep[r] != r) r = rep[r];
    while (rep[s] != r) {  // path compression
      int nx = rep[s];
      rep[s] = r;
      s = nx;
    }
    return r;
  };
  // final nodes, their tree, and a postorder of it
  std::vector<int> fin_id(ns, -1);
  int nf = 0;
  for (int s = 0; s < ns; ++s)
    if (rep[s] == s) fin_id[s] = nf++;
  std::vector<int> fparent(nf, -1);
  std::vector<std::vector<int>> members(nf);
  for (int s = 0; s < ns; ++s) {
    int r = findrep(s);
    members[fin_id[r]].push_back(s);
    if (r == s) {
      int p = sp[s];
      fparent[fin_id[s]] = p < 0 ? -1 : fin_id[findrep(p)];
    }
  }
  std::vector<int> fpost;
  postorder(nf, fparent, fpost);
  // new pivot positions: nodes in postorder, member supernodes in old order
  {
    std::vector<int> newpos(n);
    int pos = 0;
    S.nnodes = nf;
    S.sptr.assign(nf + 1, 0);
    S.sparent.assign(nf, nf);
    std::vector<int> fnew(nf);
    for (int k = 0; k < nf; ++k) fnew[fpost[k]] = k;
    for (int k = 0; k < nf; ++k) {
      int f = fpost[k];
      S.sptr[k] = pos;
      for (int s : members[f])
        for (int j = sn_first[s]; j < sn_first[s + 1]; ++j) newpos[j] = pos++;
      S.sparent[k] = fparent[f] < 0 ? nf : fnew[fparent[f]];
    }
    S.sptr[nf] = pos;
    for (int i = 0; i < n; ++i) order[i] = newpos[order[i]];
    for (int i = 0; i < n; ++i) porder[order[i]] = i;
  }
  S.order = order;
  S.porder = porder;
  return finish_symbolic(n, ptr, row, &xadj, &adj, opt, S);
}

// Everything SpLLT derives from the symbolic quintuple (order, sptr, sparent [, rptr, rlist]):
// tree arrays, schedule levels, row lists (computed here unless S already carries them),
// flop weights, pruning, tile layout, the val -> L map.  Shared by the built-in analyse and
// by analyse_symbolic (the quintuple SpLLT takes from SSIDS, analyse_mod:155-158).
int finish_symbolic(int n, const int64_t* ptr, const int* row, const std::vector<int64_t>* xadj_p,
                    const std::vector<int>* adj_p, const SymOptions& opt, Symbolic& S) {
  const std::vector<int>& order = S.order;
  const std::vector<int>& porder = S.porder;
  const int nn = S.nnodes;
  S.snode_of.assign(n, 0);
  for (int s = 0; s < nn; ++s)
    for (int j = S.sptr[s]; j < S.sptr[s + 1]; ++j) S.snode_of[j] = s;

  // ---- tree arrays -------------------------------------------------------
  S.child_ptr.assign(nn + 2, 0);
  for (int s = 0; s < nn; ++s) S.child_ptr[S.sparent[s] + 1]++;
  for (int s = 0; s <= nn; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
  S.child_idx.resize(nn);
  {
    std::vector<int> pos(S.child_ptr.begin(), S.child_ptr.end() - 1);
    for (int s = 0; s < nn; ++s) S.child_idx[pos[S.sparent[s]]++] = s;
  }
  S.least_desc.resize(nn);
  S.level.assign(nn, 0);
  for (int s = 0; s < nn; ++s) S.least_desc[s] = s;
  for (int s = 0; s < nn; ++s) {
    int p = S.sparent[s];
    if (p < nn) {
      S.least_desc[p] = std::min(S.least_desc[p], S.least_desc[s]);
      S.level[p] = std::max(S.level[p], S.level[s] + 1);
    }
  }
  S.maxdepth = 0;
  for (int s = 0; s < nn; ++s) S.maxdepth = std::max(S.maxdepth, S.level[s] + 1);
  // Schedule levels are ALAP: a node sits one level below its parent (roots on
  // the top level), so that siblings -- subtrees of similar size in a nested
  // dissection tree -- share batched launches.  Children always have a strictly
  // smaller level than their parent.
  for (int s = nn - 1; s >= 0; --s) {
    int p = S.sparent[s];
    S.level[s] = (p < nn) ? S.level[p] - 1 : S.maxdepth - 1;
  }
  {
    int mn = 0;
    for (int s = 0; s < nn; ++s) mn = std::min(mn, S.level[s]);
    for (int s = 0; s < nn; ++s) S.level[s] -= mn;
  }

  if (S.rlist.empty()) {
    const std::vector<int64_t>& xadj = *xadj_p;
    const std::vector<int>& adj = *adj_p;
  // ---- row lists (supernodal symbolic factorisation) -------------------
  S.rptr.assign(nn + 1, 0);
  {
    std::vector<int> mark(n, -1);
    std::vector<std::vector<int>> below(nn);  // rows strictly below the node's columns
    std::vector<int> tmp;
    for (int s = 0; s < nn; ++s) {
      const int c0 = S.sptr[s], c1 = S.sptr[s + 1];
      tmp.clear();
      for (int j = c0; j < c1; ++j) {
        int v = porder[j];
        for (int64_t e = xadj[v]; e < xadj[v + 1]; ++e) {
          int i = order[adj[e]];
          if (i >= c1 && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
        }
      }
      for (int e = S.child_ptr[s]; e < S.child_ptr[s + 1]; ++e) {
        int c = S.child_idx[e];
        for (int i : below[c])
          if (i >= c1 && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
        std::vector<int>().swap(below[c]);
      }
      std::sort(tmp.begin(), tmp.end());
      below[s] = tmp;
      S.rptr[s + 1] = S.rptr[s] + (c1 - c0) + (int64_t)tmp.size();
      for (int j = c0; j < c1; ++j) S.rlist.push_back(j);
      S.rlist.insert(S.rlist.end(), tmp.begin(), tmp.end());
    }
  }

  }

  // ---- flop weights (spllt_symbolic) -----------------------------------
  S.weight.assign(nn + 1, 0);
  S.nnzL = 0;
  for (int s = 0; s < nn; ++s) {
    int64_t m = S.nrow(s), nc = S.ncol(s), mm = m - nc, fl = 0;
    for (int64_t j = 1; j <= nc; ++j) {
      fl += (mm + j) * (mm + j);
      S.nnzL += mm + j;
    }
    S.weight[s] += fl;
    S.weight[S.sparent[s]] += S.weight[s];
  }
  S.flops = S.weight[nn];

  // ---- pruning ------------------------------------------------------------
  S.small.assign(nn, 0);
  if (opt.prune_tree) prune_tree(S, opt.ncpu);

  // ---- tile layout --------------------------------------------------------
  const int nb = S.nb;
  S.node_bcol0.assign(nn + 1, 0);
  S.maxmn = 0;
  int64_t off = 0, blk = 0;
  for (int s = 0; s < nn; ++s) {
    S.node_bcol0[s] = (int)S.bcols.size();
    const int m = S.nrow(s), nc = S.ncol(s);
    for (int c0 = 0; c0 < nc; c0 += nb) {
      BlockCol b;
      b.node = s;
      b.width = std::min(nb, nc - c0);
      b.r0 = c0;
      b.nrow = m - c0;
      b.off = off;
      b.blk0 = blk;
      off += (int64_t)b.nrow * b.width;
      blk += (b.nrow - 1) / nb + 1;
      S.maxmn = std::max(S.maxmn, std::max(b.width, std::min(nb, b.nrow)));
      S.bcols.push_back(b);
    }
  }
  S.node_bcol0[nn] = (int)S.bcols.size();
  S.arena = off;
  S.nblk = blk;

  // ---- val -> arena map ---------------------------------------------------
  // Entry e=(i,j) of the user's lower triangle lands in pivot column
  // min(order[i],order[j]), row max(...) (spllt_make_map), then in that
  // column's block column at (localrow - r0)*width + (col - c0) (spllt_lcol_map).
  {
    const int nbc = S.nbcol();
    S.lmap_ptr.assign(nbc + 1, 0);
    auto bcol_of = [&](int col) {
      int s = S.snode_of[col];
      return S.node_bcol0[s] + (col - S.sptr[s]) / nb;
    };
    for (int j = 0; j < n; ++j)
      for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
        int a = order[row[e]], b = order[j];
        S.lmap_ptr[bcol_of(std::min(a, b)) + 1]++;
      }
    for (int b = 0; b < nbc; ++b) S.lmap_ptr[b + 1] += S.lmap_ptr[b];
    S.map_dst.resize(S.nnzA);
    S.map_src.resize(S.nnzA);
    std::vector<int64_t> pos(S.lmap_ptr.begin(), S.lmap_ptr.end() - 1);
    for (int j = 0; j < n; ++j)
      for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
        int a = order[row[e]], b = order[j];
        int col = std::min(a, b), r = std::max(a, b);
        int s = S.snode_of[col];
        int bc = S.node_bcol0[s] + (col - S.sptr[s]) / nb;
        const BlockCol& B = S.bcols[bc];
        const int* rows = S.rows(s);
        int lr = (int)(std::lower_bound(rows, rows + S.nrow(s), r) - rows);
        if (lr >= S.nrow(s) || rows[lr] != r) return S.ordering == "symbolic" ? -10 : -99;  // row lists do not cover A
        int64_t k = pos[bc]++;
        S.map_dst[k] = B.off + (int64_t)(lr - B.r0) * B.width + (col - S.sptr[s] - B.r0);
        S.map_src[k] = e;
      }
  }
  return 0;
}

int analyse_symbolic(int n, const int64_t* ptr, const int* row, int nnodes, const int* sptr,
                     const int* sparent, const int64_t* rptr, const int* rlist, const int* order,
                     const SymOptions& opt, Symbolic& S) {
  S = Symbolic();
  S.n = n;
  S.nb = opt.nb < 1 ? 256 : opt.nb;
  if (n <= 0) return n == 0 ? 0 : -10;
  if (nnodes < 1 || !sptr || !sparent || !rptr || !rlist || !order) return -10;
  S.nnzA = ptr[n];
  for (int j = 0; j < n; ++j)
    for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e)
      if (row[e] < 0 || row[e] >= n) return -10;
  // order: a permutation; nodes: postordered (parent > child), contiguous column ranges
  S.order.assign(order, order + n);
  S.porder.assign(n, -1);
  for (int i = 0; i < n; ++i) {
    const int p = order[i];
    if (p < 0 || p >= n || S.porder[p] >= 0) return -10;
    S.porder[p] = i;
  }
  if (sptr[0] != 0 || sptr[nnodes] != n || rptr[0] != 0) return -10;
  for (int s = 0; s < nnodes; ++s) {
    if (sptr[s + 1] <= sptr[s]) return -10;
    if (sparent[s] <= s || sparent[s] > nnodes) return -10;
    const int nc = sptr[s + 1] - sptr[s];
    const int64_t m = rptr[s + 1] - rptr[s];
    if (m < nc) return -10;
    const int* r = rlist + rptr[s];
    for (int64_t k = 0; k < m; ++k) {
      if (r[k] < 0 || r[k] >= n) return -10;
      if (k < nc ? r[k] != sptr[s] + (int)k : r[k] <= r[k - 1]) return -10;   // own columns first, sorted
    }
  }
  // every row below a node's own columns must also be a row of its parent (the fill of the
  // child lands there): the factorization indexes the parent's row list with the child's rows
  for (int s = 0; s < nnodes; ++s) {
    const int p = sparent[s];
    const int nc = sptr[s + 1] - sptr[s];
    const int64_t m = rptr[s + 1] - rptr[s];
    if (p >= nnodes) {
      if (m != nc) return -10;   // a root has no rows below its columns
      continue;
    }
    const int* pr = rlist + rptr[p];
    const int64_t pm = rptr[p + 1] - rptr[p];
    int64_t q = 0;
    for (int64_t k = nc; k < m; ++k) {
      const int r = rlist[rptr[s] + k];
      while (q < pm && pr[q] < r) ++q;
      if (q >= pm || pr[q] != r) return -10;
    }
  }
  S.ordering = "symbolic";
  S.nnodes = nnodes;
  S.sptr.assign(sptr, sptr + nnodes + 1);
  S.sparent.assign(sparent, sparent + nnodes);
  S.rptr.assign(rptr, rptr + nnodes + 1);
  S.rlist.assign(rlist, rlist + rptr[nnodes]);
  return finish_symbolic(n, ptr, row, nullptr, nullptr, opt, S);
}

}  // namespace spx
