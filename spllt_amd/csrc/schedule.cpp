// Host-side construction of the batched stream-DAG program (see schedule.hpp).
#include "schedule.hpp"
#include <climits>
#include <cstdlib>

#include <algorithm>
#include <cstdio>

namespace spx {
namespace {

struct Edge {
  int stream = 0, wait0 = -1, wait1 = -1, record = -1, overlap = 0;
};

static int64_t env_int(const char* name, int64_t dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoll(v) : dflt;
}

struct Builder {
  bool level_atomic = false;  // DIRECT units of the current level must use atomics
  const Symbolic& S;
  const ScheduleOptions& opt;
  Program& P;
  int nb, pw;

  Builder(const Symbolic& s, const ScheduleOptions& o, Program& p)
      : S(s), opt(o), P(p), nb(s.nb), pw(std::min(o.pw, kPanelMax)) {}

  static int cdiv(int a, int b) { return (a + b - 1) / b; }

  int pick_tile(int M, int N) const {
    if (opt.tile <= 64) return 64;
    return (M >= 96 && N >= 96) ? 128 : 64;
  }

  // Append the tiles of unit `u` covering columns [jbeg, jend) with tiles of
  // edge T (tile indices are in units of T; jbeg is a multiple of T).
  void add_tiles(std::vector<UpdTile>& out, int uid, const UpdUnit& u, int T, bool lower,
                 int jbeg, int jend) {
    int nti = cdiv(u.M, T);
    for (int tj = jbeg / T; tj * T < jend; ++tj)
      for (int ti = 0; ti < nti; ++ti) {
        if (lower && u.src_r0 + (ti + 1) * T - 1 < u.src_c0 + tj * T) continue;
        UpdTile t;
        t.unit = uid;
        t.ti = (short)ti;
        t.tj = (short)tj;
        out.push_back(t);
      }
  }

  // Emit one GEMM phase: the units in `us` are split by tile size into at most
  // two launches.
  // last launch index emitted per call (to attach `record` to the final one)
  void emit_gemm(int level, std::vector<UpdUnit>& us, double flops, bool lower = true,
                 Edge e = Edge()) {
    if (us.empty()) {
      // keep the event graph consistent: an empty phase still has to forward
      // its record event; emit a marker launch with no work
      if (e.record >= 0 || e.wait0 >= 0 || e.wait1 >= 0) {
        Launch L;
        L.kind = L_GEMM; L.level = level; L.first = 0; L.count = 0; L.tile = 64; L.flops = 0;
        L.stream = e.stream; L.wait0 = e.wait0; L.wait1 = e.wait1; L.record = e.record;
        P.launches.push_back(L);
      }
      return;
    }
    std::vector<UpdTile> t128, t64, t32;
    // Tile size per launch (measured, scripts/update_bench.hip + gpu_tilesweep.sh):
    // the 128-tile (8 waves, 2 workgroups/CU) only pays when the launch fills the
    // chip for many rounds (>= 4096 tiles); the 64-tile is as fast per flop at every
    // K and has the shorter tail; latency-bound launches (<= 2048 64-tiles: one
    // round) use 32-tiles so that every CU gets work and a lone tile takes ~10 us.
    int64_t n128 = 0, n64 = 0;
    for (auto& u : us) {
      if (u.mode == MODE_TRSM) continue;
      if (pick_tile(u.M, u.N) == 128) n128 += (int64_t)cdiv(u.M, 128) * cdiv(u.N, 128);
      n64 += (int64_t)cdiv(u.M, 64) * cdiv(u.N, 64);
    }
    static const int64_t small_max = env_int("SPLLT_TILE_SMALL", 4096);
    static const int64_t tiny_max = env_int("SPLLT_TILE_TINY", 2048);
    const bool small_launch = n128 > 0 && n128 < small_max;
    const bool tiny_launch = n64 > 0 && n64 <= tiny_max;
    for (auto& u : us) {
      int uid = (int)P.units.size();
      if (u.mode == MODE_DIRECT) u.atomic = level_atomic ? 1 : 0;
      u.a_w = S.bcols[u.src_bcol0].width;
      u.a_off = S.bcols[u.src_bcol0].off;
      P.units.push_back(u);
      int T = (u.mode == MODE_TRSM) ? (u.N > 64 ? 128 : pick_tile(u.M, u.N)) : pick_tile(u.M, u.N);
      if (small_launch && u.mode != MODE_TRSM) T = 64;
      if (tiny_launch && u.mode != MODE_TRSM) T = 32;
      const bool low = lower && u.lower;
      if (T == 128 && u.mode != MODE_TRSM) {
        // 128-wide tile columns, except a trailing remainder of <= 64 columns
        // which is covered by 64-wide tiles (a mostly empty 128-column would
        // waste up to half of its MFMAs)
        int rem = u.N % 128;
        int full = (rem > 0 && rem <= 64) ? u.N - rem : u.N;
        if (full > 0) add_tiles(t128, uid, u, 128, low, 0, full);
        if (full < u.N) add_tiles(t64, uid, u, 64, low, full, u.N);
      } else {
        add_tiles(T == 128 ? t128 : (T == 64 ? t64 : t32), uid, u, T, low, 0, u.N);
      }
    }
    // longest tiles first: the K extent of a tile is its unit's source width, and a
    // launch mixes units of very different K (inter-node updates), so dispatching
    // the heavy tiles first shortens the tail of the launch
    {
      const int ubase = (int)P.units.size() - (int)us.size();
      std::vector<int64_t> work(us.size());
      for (size_t i = 0; i < us.size(); ++i) {
        const UpdUnit& u = us[i];
        int64_t k = 0;
        if (u.nseg == 1) k = u.klen >= 0 ? u.klen : S.bcols[u.src_bcol0].width;
        else for (int sg = 0; sg < u.nseg; ++sg) k += S.bcols[u.src_bcol0 + sg].width;
        work[i] = k;
      }
      auto by_work = [&](const UpdTile& a, const UpdTile& b) { return work[a.unit - ubase] > work[b.unit - ubase]; };
      std::stable_sort(t128.begin(), t128.end(), by_work);
      std::stable_sort(t64.begin(), t64.end(), by_work);
      std::stable_sort(t32.begin(), t32.end(), by_work);
    }
    double ntot = (double)t128.size() * 16 + (double)t64.size() * 4 + (double)t32.size();
    std::vector<UpdTile>* lists[3] = {&t128, &t64, &t32};
    const int edges[3] = {128, 64, 32};
    int first_nonempty = -1, last_nonempty = -1;
    for (int i = 0; i < 3; ++i)
      if (!lists[i]->empty()) {
        if (first_nonempty < 0) first_nonempty = i;
        last_nonempty = i;
      }
    for (int pass = 0; pass < 3; ++pass) {
      auto& tv = *lists[pass];
      if (tv.empty()) continue;
      Launch L;
      L.kind = L_GEMM;
      L.level = level;
      L.first = (int64_t)P.tiles.size();
      L.count = (int64_t)tv.size();
      L.tile = edges[pass];
      L.flops = flops * ((double)tv.size() * (edges[pass] / 32) * (edges[pass] / 32)) / std::max(1.0, ntot);
      L.stream = e.stream;
      L.wait0 = pass == first_nonempty ? e.wait0 : -1;
      L.wait1 = pass == first_nonempty ? e.wait1 : -1;
      L.record = pass == last_nonempty ? e.record : -1;
      L.overlap = e.overlap;
      P.tiles.insert(P.tiles.end(), tv.begin(), tv.end());
      P.launches.push_back(L);
    }
    us.clear();
  }

  // Inter-node update units of node s, one per touched ancestor block column,
  // covering every K segment (block column) of s.
  void between_templates(int s, std::vector<UpdUnit>& out) {
    const int nn = S.nnodes;
    const int m = S.nrow(s);
    const int* idx = S.rows(s);
    int cptr = S.ncol(s);
    int a = S.sparent[s];
    while (a < nn && cptr < m) {
      const int asa = S.sptr[a], aen = S.sptr[a + 1] - 1;
      while (cptr < m && idx[cptr] < asa) cptr++;
      if (cptr >= m) break;
      if (idx[cptr] <= aen) {
        // positions of rows cptr..m-1 of s inside a's row list
        const int* aidx = S.rows(a);
        const int am = S.nrow(a);
        int64_t base = (int64_t)P.relpos.size();
        {
          int q = 0;
          for (int r = cptr; r < m; ++r) {
            while (q < am && aidx[q] < idx[r]) q++;
            if (q >= am || aidx[q] != idx[r]) {
              std::fprintf(stderr, "spllt-hip: structure inclusion violated (node %d -> %d)\n", s, a);
              q = std::min(q, am - 1);
            }
            P.relpos.push_back(q);
          }
        }
        const int first = cptr;
        while (cptr < m && idx[cptr] <= aen) {
          int cb = (idx[cptr] - asa) / nb;
          int jlast = std::min(asa + (cb + 1) * nb - 1, aen);
          int cptr2 = cptr;
          while (cptr2 + 1 < m && idx[cptr2 + 1] <= jlast) cptr2++;
          const BlockCol& D = S.bcols[S.node_bcol0[a] + cb];
          UpdUnit u{};
          u.b_bcol0 = -1;
          u.lower = 1;
          u.mode = MODE_SCATTER;
          u.d_off = D.off;
          u.d_ld = D.width;
          u.d_row0 = D.r0;
          u.d_col0 = asa + cb * nb;
          u.relrow_off = base + (cptr - first);
          u.gcol_off = S.rptr[s] + cptr;
          u.src_bcol0 = S.node_bcol0[s];
          u.nseg = S.node_bcol0[s + 1] - S.node_bcol0[s];
          u.seg_r0 = 0;
          u.seg_stride = nb;
          u.src_r0 = cptr;
          u.src_c0 = cptr;
          u.M = m - cptr;
          u.N = cptr2 - cptr + 1;
          u.k0 = 0;
          u.klen = -1;
          out.push_back(u);
          cptr = cptr2 + 1;
        }
      }
      a = S.sparent[a];
    }
  }

  // Units for the not yet issued K segments of every node of the level, given
  // that its first `done` block columns are final.  final_pass: take everything
  // that is left.  Otherwise a node contributes when it is complete, or when at
  // least slice_width finished block columns are pending and two or more are
  // still to come (every slice repeats the scatter of the whole update).
  double collect_between(const std::vector<int>& nodes, const std::vector<std::vector<UpdUnit>>& tmpl,
                         std::vector<int>& emitted, int done, bool final_pass,
                         std::vector<UpdUnit>& out) {
    double fl = 0;
    static const int slice_w = (int)env_int("SPLLT_SLICE_WIDTH", opt.slice_width);
    static const int slice_tail = (int)env_int("SPLLT_SLICE_TAIL", 2);
    for (size_t i = 0; i < nodes.size(); ++i) {
      const int s = nodes[i];
      const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
      const int avail = std::min(done, nc);
      int take = 0;
      if (final_pass) take = nc - emitted[i];
      else if (avail == nc) take = nc - emitted[i];
      else if (avail - emitted[i] >= slice_w && nc - avail >= slice_tail) take = avail - emitted[i];
      if (take <= 0 || tmpl[i].empty()) { if (take > 0) emitted[i] += take; continue; }
      const int b0 = S.node_bcol0[s] + emitted[i];
      int kcols = 0;
      for (int b = b0; b < b0 + take; ++b) kcols += S.bcols[b].width;
      for (const UpdUnit& t : tmpl[i]) {
        UpdUnit u = t;
        u.src_bcol0 = b0;
        u.nseg = take;
        u.seg_r0 = S.bcols[b0].r0;
        out.push_back(u);
        fl += 2.0 * kcols * ((double)u.M * u.N - 0.5 * u.N * (u.N - 1));
      }
      emitted[i] += take;
    }
    return fl;
  }

  void run() {
    P.pw = pw;
    const int nn = S.nnodes;
    int maxlevel = -1;
    for (int s = 0; s < nn; ++s) maxlevel = std::max(maxlevel, S.level[s]);

    // dinv slots: one per (block column, panel)
    std::vector<int64_t> dinv_slot(S.nbcol() + 1, 0);
    {
      int64_t o = 0;
      for (int b = 0; b < S.nbcol(); ++b) {
        dinv_slot[b] = o;
        int w = S.bcols[b].width;
        for (int c = 0; c < w; c += pw) {
          int pn = std::min(pw, w - c);
          o += (int64_t)pn * pn;
        }
      }
      dinv_slot[S.nbcol()] = o;
      P.dinv_size = o;
    }

    std::vector<UpdUnit> us;
    int ev_level = -1;  // completion event of the previous level
    // A partitioned program (multi-GPU) runs the rank's own subtrees first, then
    // an EXCHANGE marker (the extend-add of the top-tree block columns across
    // ranks happens there), then the replicated top tree.
    const bool partitioned = opt.node_owner != nullptr && opt.nranks > 1;
    const int nphase = partitioned ? 2 : 1;
    for (int ph = 0; ph < nphase; ++ph) {
    std::vector<std::vector<int>> by_level(maxlevel + 1);
    for (int s = 0; s < nn; ++s) {
      bool take = true;
      if (partitioned) take = (ph == 0) ? (opt.node_owner[s] == opt.rank) : (opt.node_owner[s] < 0);
      if (take) by_level[S.level[s]].push_back(s);
    }
    if (partitioned && ph == 1) {
      Launch X;
      X.kind = L_EXCHANGE;
      X.level = -1;
      X.first = X.count = 0;
      X.tile = 0;
      X.flops = 0;
      X.stream = 0;
      X.wait0 = ev_level;  // everything of phase 1, both streams
      int ev = P.nevents++;
      X.record = ev;
      ev_level = ev;
      P.launches.push_back(X);
    }
    for (int lev = 0; lev <= maxlevel; ++lev) {
      const auto& nodes = by_level[lev];
      if (nodes.empty()) continue;
      int maxnc = 0;
      for (int s : nodes) maxnc = std::max(maxnc, S.node_bcol0[s + 1] - S.node_bcol0[s]);
      const bool la = opt.lookahead;
      bool first_of_level = true;   // first panel-stream launch waits for the previous level
      int evB_prev = -1;            // bulk event of step c-1 (trailing update of c-1 -> c+1..)
      int evP_last = -1;            // panel event of the last finished step
      int evB1_prev = -1;           // bulk event: rest rows of block column c updated by c-1
      // The fused strip kernel shortens the panel chain (1 launch instead of
      // 2*np-1 per block column) but runs one workgroup per CU; it pays where
      // the level is latency-bound (few, large nodes), not where thousands of
      // strips would queue.  Use it when all strips of a step fit in ~2 rounds.
      std::vector<int> evB_hist;    // bulk event of every step of this level
      bool fs = la && opt.fused_strip;
      if (fs) {
        int64_t worst = 0;
        for (int c = 0; c < maxnc; ++c) {
          int64_t strips = 0;
          for (int s : nodes) {
            int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (c >= nc) continue;
            const BlockCol& B = S.bcols[S.node_bcol0[s] + c];
            if (B.nrow > B.width) strips += cdiv(B.nrow - B.width, B.width <= 320 ? 32 : 16);
          }
          worst = std::max(worst, strips);
        }
        fs = worst <= opt.strip_limit;
      }
      // fused panel steps (k_panel_step): TRSM + every missing update of the next
      // 64-wide panel in one launch, on levels whose steps are latency-bound
      bool ps = !fs && opt.panel_step;
      if (ps) {
        int64_t worst = 0;
        for (int c = 0; c < maxnc; ++c) {
          int64_t t = 0;
          for (int s : nodes) {
            int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (c >= nc) continue;
            const BlockCol& B = S.bcols[S.node_bcol0[s] + c];
            t += cdiv(std::max(0, B.nrow - std::min(pw, B.width)), 32);
          }
          worst = std::max(worst, t);
        }
        ps = worst <= opt.panel_step_limit;
      }
      // inter-node update units of every node of the level (all K segments); they
      // are issued in slices as the block columns they read become final
      std::vector<std::vector<UpdUnit>> tmpl(nodes.size());
      std::vector<int> emitted(nodes.size(), 0);
      for (size_t i = 0; i < nodes.size(); ++i) between_templates(nodes[i], tmpl[i]);
      int evF_last = -1;            // last far-stream event (early inter-node slices)
      const bool lazy = la && !fs && !ps && opt.lazy_next;
      // merged / fused panel updates share destinations with concurrently running launches
      level_atomic = lazy || ps;
      for (int c = 0; c < maxnc; ++c) {
        int maxp = 0;
        for (int s : nodes) {
          int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
          if (c < nc) maxp = std::max(maxp, cdiv(S.bcols[S.node_bcol0[s] + c].width, pw));
        }
        // With the fused strip kernel the panel chain only walks the diagonal
        // tile (rows < width); the rows below are solved by one k_trsm_strip
        // launch once every panel of the tile is factored.
        auto chain_rows = [&](const BlockCol& B) {
          return (fs && B.width <= 896) ? std::min(B.nrow, B.width) : B.nrow;
        };
        // (0) single-workgroup panel chain of the diagonal tiles (fused mode, w <= 256)
        bool chained = false;
        // (k_tile_chain walks K in steps of 4 and 16-column tiles: panel widths that are
        // no multiple of 16 keep the per-panel launches)
        if (fs && opt.tile_chain && pw % 16 == 0) {
          bool all_fit = true;
          for (int s : nodes) {
            int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (c < nc && S.bcols[S.node_bcol0[s] + c].width > 256) all_fit = false;
          }
          if (all_fit) {
            Launch L;
            L.kind = L_CHAIN;
            L.level = lev;
            L.first = (int64_t)P.chain_units.size();
            L.tile = 0;
            double fl = 0;
            for (int s : nodes) {
              int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              int b = S.node_bcol0[s] + c;
              const BlockCol& B = S.bcols[b];
              PotrfUnit q{};
              q.off = B.off;
              q.dinv_off = dinv_slot[b];
              q.ld = B.width;
              q.n = std::min(B.nrow, B.width);
              q.gcol = S.sptr[s] + B.r0;
              q.flags = pw;
              P.chain_units.push_back(q);
              fl += (double)q.n * q.n * q.n / 3.0;
            }
            L.count = (int64_t)P.chain_units.size() - L.first;
            L.flops = fl;
            P.flops_potrf += fl;
            if (first_of_level) {
              L.wait0 = ev_level;
              first_of_level = false;
            }
            if (L.count > 0) P.launches.push_back(L);
            chained = true;
          }
        }
        for (int p = 0; p < maxp && !chained; ++p) {
          // (1) left-looking update of panel p by the previous panels of the block
          // column and (lazy_next) by the previous block column of the node
          double fl = 0;
          if (!ps && (p > 0 || (lazy && c > 0))) {
            for (int s : nodes) {
              int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              int b = S.node_bcol0[s] + c;
              const BlockCol& B = S.bcols[b];
              int c0 = p * pw;
              if (c0 >= B.width) continue;
              int pn = std::min(pw, B.width - c0);
              if (lazy && c > 0) {
                const BlockCol& Q = S.bcols[b - 1];
                UpdUnit v{};
                v.b_bcol0 = -1;
                v.lower = 1;
                v.mode = MODE_DIRECT;
                v.d_off = B.off;
                v.d_ld = B.width;
                v.d_row0 = c0;
                v.d_col0 = c0;
                v.src_bcol0 = b - 1;
                v.nseg = 1;
                v.seg_r0 = Q.r0;
                v.seg_stride = nb;
                v.src_r0 = B.r0 + c0;
                v.src_c0 = B.r0 + c0;
                v.M = B.nrow - c0;
                v.N = pn;
                v.k0 = 0;
                v.klen = Q.width;
                us.push_back(v);
                fl += 2.0 * Q.width * ((double)v.M * pn - 0.5 * pn * (pn - 1));
              }
              if (p == 0) continue;
              UpdUnit u{};
              u.b_bcol0 = -1;
              u.lower = 1;
              u.mode = MODE_DIRECT;
              u.d_off = B.off;
              u.d_ld = B.width;
              u.d_row0 = c0;
              u.d_col0 = c0;
              u.src_bcol0 = b;
              u.nseg = 1;
              u.seg_r0 = B.r0;
              u.seg_stride = nb;
              u.src_r0 = B.r0 + c0;
              u.src_c0 = B.r0 + c0;
              u.M = chain_rows(B) - c0;
              u.N = pn;
              u.k0 = 0;
              u.klen = c0;
              us.push_back(u);
              fl += 2.0 * c0 * ((double)u.M * pn - 0.5 * pn * (pn - 1));
            }
            P.flops_update += fl;
            Edge e1;
            // the first launch of step c follows the bulk update (c-2 -> c..)
            if (lazy && p == 0 && c >= 2) e1.wait0 = evB_hist[c - 2];
            emit_gemm(lev, us, fl, true, e1);
          }
          // (2) POTRF of the diagonal panel blocks
          {
            Launch L;
            L.kind = L_POTRF;
            L.level = lev;
            L.first = (int64_t)P.potrf_units.size();
            L.tile = 0;
            fl = 0;
            for (int s : nodes) {
              int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              int b = S.node_bcol0[s] + c;
              const BlockCol& B = S.bcols[b];
              int c0 = p * pw;
              if (c0 >= B.width) continue;
              int pn = std::min(pw, B.width - c0);
              PotrfUnit q{};
              q.off = B.off + (int64_t)c0 * B.width + c0;
              q.ld = B.width;
              q.n = pn;
              q.gcol = S.sptr[s] + B.r0 + c0;
              int64_t slot = dinv_slot[b];
              for (int cc = 0; cc < c0; cc += pw) slot += (int64_t)pw * pw;
              q.dinv_off = slot;
              P.potrf_units.push_back(q);
              fl += (double)pn * pn * pn / 3.0;
            }
            L.count = (int64_t)P.potrf_units.size() - L.first;
            L.flops = fl;
            P.flops_potrf += fl;
            if (la && first_of_level) {
              L.wait0 = ev_level;  // everything of the previous level (incl. its bulk stream)
              first_of_level = false;
            } else if (la && ps && p == 0 && c >= 2) {
              L.wait0 = evB_hist[c - 2];  // bulk update (c-2 -> c..) precedes the first panel of c
            }
            if (L.count > 0) P.launches.push_back(L);
          }
          // (3p) fused panel step: TRSM of the rows below + update of the next panel
          if (ps) {
            Launch L;
            L.kind = L_PANEL;
            L.level = lev;
            L.first = (int64_t)P.tiles.size();
            L.tile = 32;
            fl = 0;
            for (int s : nodes) {
              int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              int b = S.node_bcol0[s] + c;
              const BlockCol& B = S.bcols[b];
              int c0 = p * pw;
              if (c0 >= B.width) continue;
              int pn = std::min(pw, B.width - c0);
              int rows = B.nrow - (c0 + pn);
              if (rows <= 0) continue;
              PanelStepUnit q{};
              q.off = B.off;
              q.ld = B.width;
              q.c0 = c0;
              q.pn = pn;
              q.nrows = rows;
              int64_t slot = dinv_slot[b];
              for (int cc = 0; cc < c0; cc += pw) slot += (int64_t)pw * pw;
              q.dinv_off = slot;
              q.d_off = -1;
              q.s_off = -1;
              double kk = 0;  // K extent of the update (global segments + the panel)
              if (c0 + pn < B.width) {
                // next panel in the same block column: previous block column + own panels
                q.d_off = B.off;
                q.d_ld = B.width;
                q.d_c0 = c0 + pn;
                q.d_pn = std::min(pw, B.width - (c0 + pn));
                q.d_rshift = 0;
                kk = c0 + pn;
                if (c > 0) {
                  const BlockCol& Q = S.bcols[b - 1];
                  q.s_off = Q.off;
                  q.s_ld = Q.width;
                  q.s_k = Q.width;
                  q.s_rshift = Q.width;
                  kk += Q.width;
                }
              } else if (c + 1 < nc) {
                // panel 0 of the next block column: this block column's panels
                const BlockCol& Dn = S.bcols[b + 1];
                q.d_off = Dn.off;
                q.d_ld = Dn.width;
                q.d_c0 = 0;
                q.d_pn = std::min(pw, Dn.width);
                q.d_rshift = B.width;
                kk = c0 + pn;
              }
              int uid = (int)P.panel_units.size();
              P.panel_units.push_back(q);
              for (int t = 0; t < cdiv(rows, 32); ++t) {
                UpdTile tt;
                tt.unit = uid;
                tt.ti = (short)t;
                tt.tj = 0;
                P.tiles.push_back(tt);
              }
              const double ft = (double)rows * pn * pn;
              double fu = 0;
              if (q.d_off >= 0) fu = 2.0 * kk * ((double)rows * q.d_pn - 0.5 * q.d_pn * (q.d_pn - 1));
              P.flops_trsm += ft;
              P.flops_update += fu;
              fl += ft + fu;
            }
            L.count = (int64_t)P.tiles.size() - L.first;
            L.flops = fl;
            if (L.count > 0) P.launches.push_back(L);
            continue;
          }
          // (3) TRSM of the rows below the panel: X = A * inv(Lpp)^T (in place)
          fl = 0;
          for (int s : nodes) {
            int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (c >= nc) continue;
            int b = S.node_bcol0[s] + c;
            const BlockCol& B = S.bcols[b];
            int c0 = p * pw;
            if (c0 >= B.width) continue;
            int pn = std::min(pw, B.width - c0);
            int rows = chain_rows(B) - (c0 + pn);
            if (rows <= 0) continue;
            UpdUnit u{};
            u.b_bcol0 = -1;
            u.mode = MODE_TRSM;
            u.lower = 0;
            u.d_off = B.off;
            u.d_ld = B.width;
            u.d_row0 = c0 + pn;
            u.d_col0 = c0;
            u.src_bcol0 = b;
            u.nseg = 1;
            u.seg_r0 = B.r0;
            u.seg_stride = nb;
            u.src_r0 = B.r0 + c0 + pn;
            u.src_c0 = 0;
            u.M = rows;
            u.N = pn;
            u.k0 = c0;
            u.klen = pn;
            int64_t slot = dinv_slot[b];
            for (int cc = 0; cc < c0; cc += pw) slot += (int64_t)pw * pw;
            u.dinv_off = slot;
            u.dinv_ld = pn;
            us.push_back(u);
            fl += (double)rows * pn * pn;
          }
          P.flops_trsm += fl;
          emit_gemm(lev, us, fl, false);
        }
        // (3s) fused strip TRSM of all rows below the diagonal tile
        if (fs) {
          for (int rs : {32, 16}) {
            Launch L;
            L.kind = L_STRIP;
            L.level = lev;
            L.first = (int64_t)P.tiles.size();
            L.tile = rs;
            double fl = 0;
            for (int s : nodes) {
              int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              int b = S.node_bcol0[s] + c;
              const BlockCol& B = S.bcols[b];
              if (B.width > 896 || B.nrow <= B.width) continue;
              if ((B.width <= 320 ? 32 : 16) != rs) continue;
              StripUnit q{};
              q.off = B.off;
              q.dinv_off = dinv_slot[b];
              q.ld = B.width;
              q.row0 = B.width;
              q.nrows = B.nrow - B.width;
              q.pw = pw;
              int uid = (int)P.strip_units.size();
              P.strip_units.push_back(q);
              for (int t = 0; t < cdiv(q.nrows, rs); ++t) {
                UpdTile tt;
                tt.unit = uid;
                tt.ti = (short)t;
                tt.tj = 0;
                P.tiles.push_back(tt);
              }
              fl += (double)q.nrows * B.width * B.width;
            }
            L.count = (int64_t)P.tiles.size() - L.first;
            L.flops = fl;
            L.stream = 0;
            L.wait0 = evB1_prev;  // the rest rows were last written by the bulk stream
            P.flops_trsm += fl;
            if (L.count > 0) P.launches.push_back(L);
          }
        }
        // (4) right-looking update of the node's later block columns, K = blkn.
        // With lookahead the part that gates the next panel chain stays on the
        // panel stream (block column c+1, or only its diagonal tile when the
        // strip kernel is used) and the rest goes to the bulk stream, where it
        // overlaps the panel chain of block column c+1.
        int evP = -1;
        if (la) {
          evP = P.nevents++;
          P.launches.back().record = evP;  // last launch of the panel chain of step c
          evP_last = evP;
        }
        std::vector<UpdUnit> us_bulk, us_rest;
        double fl = 0, fl_bulk = 0, fl_rest = 0;
        for (int s : nodes) {
          int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
          if (c + 1 >= nc) continue;
          int b = S.node_bcol0[s] + c;
          const BlockCol& B = S.bcols[b];
          for (int jj = c + 1; jj < nc; ++jj) {
            const BlockCol& D = S.bcols[S.node_bcol0[s] + jj];
            UpdUnit u{};
            u.b_bcol0 = -1;
            u.lower = 1;
            u.mode = MODE_DIRECT;
            u.d_off = D.off;
            u.d_ld = D.width;
            u.d_row0 = 0;
            u.d_col0 = 0;
            u.src_bcol0 = b;
            u.nseg = 1;
            u.seg_r0 = B.r0;
            u.seg_stride = nb;
            u.src_r0 = D.r0;
            u.src_c0 = D.r0;
            u.M = D.nrow;
            u.N = D.width;
            u.k0 = 0;
            u.klen = B.width;
            const double f1 = 2.0 * B.width * ((double)u.M * u.N - 0.5 * u.N * (u.N - 1));
            if ((lazy || ps) && jj == c + 1) {
              continue;  // applied panel by panel in step c+1
            } else if (la && jj > c + 1) {
              us_bulk.push_back(u);
              fl_bulk += f1;
            } else if (fs && jj == c + 1 && D.width <= 896 && D.nrow > D.width) {
              // split: diagonal tile of block column c+1 (panel stream) / rows below (bulk)
              UpdUnit ud = u, ur = u;
              ud.M = D.width;
              ur.d_row0 = D.width;
              ur.src_r0 = D.r0 + D.width;
              ur.M = D.nrow - D.width;
              const double fd = 2.0 * B.width * ((double)ud.M * ud.N - 0.5 * ud.N * (ud.N - 1));
              us.push_back(ud);
              fl += fd;
              us_rest.push_back(ur);
              fl_rest += f1 - fd;
            } else {
              us.push_back(u);
              fl += f1;
            }
          }
        }
        P.flops_update += fl + fl_bulk + fl_rest;
        if (!la) {
          emit_gemm(lev, us, fl);
        } else {
          Edge e0;  // c -> c+1 on the panel stream, after the bulk update (c-1 -> c+1..)
          e0.stream = 0;
          e0.wait0 = evB_prev;
          if (!us.empty()) emit_gemm(lev, us, fl, true, e0);
          int evB1 = -1;
          bool waited = false;
          if (!us_rest.empty()) {
            Edge e1;
            e1.stream = 1;
            e1.overlap = 1;
            e1.wait0 = evP;
            waited = true;
            evB1 = P.nevents++;
            e1.record = evB1;
            emit_gemm(lev, us_rest, fl_rest, true, e1);
          }
          int evB = -1;
          if (!us_bulk.empty()) {
            Edge e1;
            e1.stream = 1;
            e1.overlap = 1;
            e1.wait0 = waited ? -1 : evP;
            evB = P.nevents++;
            e1.record = evB;
            emit_gemm(lev, us_bulk, fl_bulk, true, e1);
          }
          evB_prev = evB;
          evB1_prev = evB1;
          evB_hist.push_back(evB);
          // (4b) early inter-node slices: block columns 0..c are final, so the part
          // of update_between that reads them can run beside the remaining panel
          // chains of the level (far stream) instead of after the last one
          if (opt.slice_between && c + 1 < maxnc) {
            std::vector<UpdUnit> sl;
            double fs_ = collect_between(nodes, tmpl, emitted, c + 1, false, sl);
            if (!sl.empty()) {
              P.flops_between += fs_;
              Edge ef;
              ef.stream = 2;
              ef.wait0 = evP;
              static const int slice_pad = (int)env_int("SPLLT_SLICE_PAD", 0);
              ef.overlap = slice_pad;
              evF_last = P.nevents++;
              ef.record = evF_last;
              emit_gemm(lev, sl, fs_, true, ef);
            }
          }
        }
      }

      // (5) inter-node updates (update_between + scatter): whatever the slices
      // issued during the panel chains have not covered yet
      std::vector<UpdUnit> rest;
      double fl = collect_between(nodes, tmpl, emitted, INT_MAX, true, rest);
      us.insert(us.end(), rest.begin(), rest.end());
      P.flops_between += fl;
      if (!la) {
        emit_gemm(lev, us, fl);
      } else {
        // bulk stream, after the last panel chain of the level; its completion
        // event gates the first panel launch of the next level
        Edge e;
        e.stream = 1;
        e.wait0 = evP_last;
        e.wait1 = evF_last;   // early slices on the far stream
        ev_level = P.nevents++;
        e.record = ev_level;
        emit_gemm(lev, us, fl, true, e);
      }
    }
    }  // phases
    P.final_event = ev_level;
  }
};

}  // namespace

void build_program(const Symbolic& S, const ScheduleOptions& opt, Program& P) {
  P = Program();
  Builder b(S, opt, P);
  b.run();
}

void build_solve_program(const Symbolic& S, int pw, SolveProgram& P, const int* node_owner, int rank) {
  P = SolveProgram();
  pw = std::min(pw, kPanelMax);
  const int nn = S.nnodes, nbc = S.nbcol();
  P.units.resize(nbc);
  {
    int64_t o = 0;
    for (int b = 0; b < nbc; ++b) {
      const BlockCol& B = S.bcols[b];
      SolveUnit& u = P.units[b];
      u.off = B.off;
      u.dinv_off = o;
      u.idx_off = S.rptr[B.node] + B.r0;
      u.w = B.width;
      u.nrow = B.nrow;
      u.pw = pw;
      u.pad_ = 0;
      for (int c = 0; c < B.width; c += pw) {
        int pn = std::min(pw, B.width - c);
        o += (int64_t)pn * pn;
      }
    }
  }
  int maxlevel = -1;
  for (int s = 0; s < nn; ++s) maxlevel = std::max(maxlevel, S.level[s]);
  // one (diag, strip) launch pair per level and block-column step; the same
  // pairs are replayed in reverse for the backward substitution
  struct Step { int level; int64_t d0, dn, t0, tn; };
  std::vector<Step> steps;
  size_t nsub_steps = 0;
  const int nphase = node_owner ? 2 : 1;
  for (int ph = 0; ph < nphase; ++ph) {
    std::vector<std::vector<int>> by_level(maxlevel + 1);
    for (int s = 0; s < nn; ++s) {
      bool take = true;
      if (node_owner) take = (ph == 0) ? (node_owner[s] == rank) : (node_owner[s] < 0);
      if (take) by_level[S.level[s]].push_back(s);
    }
    for (int lev = 0; lev <= maxlevel; ++lev) {
      const auto& nodes = by_level[lev];
      int maxnc = 0;
      for (int s : nodes) maxnc = std::max(maxnc, S.node_bcol0[s + 1] - S.node_bcol0[s]);
      for (int c = 0; c < maxnc; ++c) {
        Step st;
        st.level = lev;
        st.d0 = (int64_t)P.diag_list.size();
        st.t0 = (int64_t)P.tiles.size();
        for (int s : nodes) {
          int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
          if (c >= nc) continue;
          int b = S.node_bcol0[s] + c;
          P.diag_list.push_back(b);
          int below = S.bcols[b].nrow - S.bcols[b].width;
          for (int t = 0; t * kSolveStripRows < below; ++t) {
            UpdTile tt;
            tt.unit = b;
            tt.ti = (short)t;
            tt.tj = 0;
            P.tiles.push_back(tt);
          }
        }
        st.dn = (int64_t)P.diag_list.size() - st.d0;
        st.tn = (int64_t)P.tiles.size() - st.t0;
        steps.push_back(st);
      }
    }
    if (ph == 0) nsub_steps = steps.size();
  }
  for (size_t i = 0; i < steps.size(); ++i) {
    const Step& st = steps[i];
    if (st.dn > 0) P.fwd.push_back(SolveLaunch{SV_DIAG_FWD, st.level, st.d0, st.dn});
    if (st.tn > 0) P.fwd.push_back(SolveLaunch{SV_STRIP_FWD, st.level, st.t0, st.tn});
    if (i + 1 == nsub_steps) P.fwd_nsub = P.fwd.size();
  }
  if (nsub_steps == 0) P.fwd_nsub = 0;
  for (size_t i = steps.size(); i-- > 0;) {
    const Step& st = steps[i];
    if (i + 1 == nsub_steps) P.bwd_ntop = P.bwd.size();
    if (st.tn > 0) P.bwd.push_back(SolveLaunch{SV_STRIP_BWD, st.level, st.t0, st.tn});
    if (st.dn > 0) P.bwd.push_back(SolveLaunch{SV_DIAG_BWD, st.level, st.d0, st.dn});
  }
  if (nsub_steps == 0) P.bwd_ntop = P.bwd.size();
}

}  // namespace spx
