// Host-side construction of the batched stream-DAG program (see schedule.hpp).
#include "schedule.hpp"
#include <climits>
#include <cstdlib>

#include <algorithm>
#include <cstdio>

namespace spx {
namespace {

struct Edge {
  int stream = 0, wait0 = -1, wait1 = -1, wait2 = -1, record = -1, overlap = 0, lat = 0;
};

static int64_t env_int(const char* name, int64_t dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoll(v) : dflt;
}

struct Builder {
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
      if (e.record >= 0 || e.wait0 >= 0 || e.wait1 >= 0 || e.wait2 >= 0) {
        Launch L;
        L.kind = L_GEMM; L.level = level; L.first = 0; L.count = 0; L.tile = 64; L.flops = 0;
        L.stream = e.stream; L.add_wait(e.wait0); L.add_wait(e.wait1); L.add_wait(e.wait2); L.record = e.record;
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
    const int64_t small_max = env_int("SPLLT_TILE_SMALL", opt.tile128_min);
    const int64_t tiny_max = env_int("SPLLT_TILE_TINY", 2048);
    const bool small_launch = n128 > 0 && n128 < small_max;
    // 32-tiles are for latency: throughput launches (bulk / far streams: trailing updates,
    // zones of the inter-node updates) keep 64-tiles unless they cannot even fill the chip once
    static const int64_t tiny_bulk = env_int("SPLLT_TILE_TINY_BULK", 512);
    const bool throughput = e.stream == ST_BULK || e.stream == ST_FAR;
    const bool tiny_launch = n64 > 0 && n64 <= (throughput ? tiny_bulk : tiny_max);
    for (auto& u : us) {
      int uid = (int)P.units.size();
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
    // XCD-aware order of a throughput launch.  Workgroup b runs on XCD b % 8 (round-robin
    // dispatch; each XCD has its own 4 MiB L2), and about `cap` workgroups of the launch are
    // resident per XCD at a time.  In the plain order (unit by unit, tile column by tile column)
    // the tiles resident on one XCD at a time are a scattered eighth of ~8 cap consecutive
    // tiles: every operand strip (128 or 64 rows x K) is fetched by nearly every XCD that holds
    // one of its tiles -- 3-5 x the algorithmic bytes on the large configurations
    // (profiles/r02/serena/categories.csv).  Here the tiles of a unit are cut into BANDS of
    // ~cap tiles (a few tile rows x all tile columns of the unit), the bands are dealt to the
    // XCDs in turn (heaviest first: the longest-first order of the launch is kept), and the
    // launch order interleaves the eight sequences -- so the workgroups resident on an XCD form
    // a compact 2-D block of ONE unit that walks K together: its A strips are shared by all
    // tile columns, its B strips by all its rows.
    static const int64_t xcd_order_env = env_int("SPLLT_XCD_ORDER", 1);
    auto xcd_order = [&](std::vector<UpdTile>& tv, int T) {
      const size_t cap = T == 128 ? 64 : (T == 64 ? 160 : 256);
      if (!xcd_order_env || tv.size() < 8 * cap) return;
      std::vector<std::vector<UpdTile>> seq(8);
      size_t band = 0, i = 0;
      while (i < tv.size()) {
        size_t j = i;
        int tjmin = tv[i].tj, tjmax = tv[i].tj;
        while (j < tv.size() && tv[j].unit == tv[i].unit) {
          tjmin = std::min<int>(tjmin, tv[j].tj);
          tjmax = std::max<int>(tjmax, tv[j].tj);
          ++j;
        }
        const size_t ntj = (size_t)(tjmax - tjmin + 1);
        const int R = (int)std::max<size_t>(1, cap / ntj);       // tile rows per band
        std::vector<UpdTile> u(tv.begin() + (long)i, tv.begin() + (long)j);
        std::stable_sort(u.begin(), u.end(), [R](const UpdTile& a, const UpdTile& b) {
          const int ga = a.ti / R, gb = b.ti / R;
          if (ga != gb) return ga < gb;
          if (a.tj != b.tj) return a.tj < b.tj;
          return a.ti < b.ti;
        });
        for (size_t k = 0; k < u.size();) {
          const int g = u[k].ti / R;
          std::vector<UpdTile>& dst = seq[band++ % 8];
          for (; k < u.size() && u[k].ti / R == g; ++k) dst.push_back(u[k]);
        }
        i = j;
      }
      std::vector<UpdTile> out;
      out.reserve(tv.size());
      size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t p = 0; p < tv.size(); ++p) {
        size_t r = p % 8;
        if (pos[r] >= seq[r].size()) {                   // this XCD's share is used up: the fullest one's
          size_t best = 0, left = 0;
          for (size_t q = 0; q < 8; ++q)
            if (seq[q].size() - pos[q] > left) { left = seq[q].size() - pos[q]; best = q; }
          r = best;
        }
        out.push_back(seq[r][pos[r]++]);
      }
      tv.swap(out);
    };
    xcd_order(t128, 128);
    xcd_order(t64, 64);
    // useful flops of ONE tile of a unit, the convention of the whole program (and of the
    // reference's symbolic count): 2 K per entry the tile really computes for the destination
    // (entries above the diagonal of a unit that straddles it do not count); TRSM: the
    // triangular solve (N per entry) + the left-looking part folded into it (2 (K - N))
    const int ubase0 = (int)P.units.size() - (int)us.size();
    auto tile_flops = [&](const UpdTile& t, int T) {
      const UpdUnit& u = us[(size_t)(t.unit - ubase0)];
      const int i0 = t.ti * T, i1 = std::min(u.M, i0 + T), j0 = t.tj * T, j1 = std::min(u.N, j0 + T);
      if (i1 <= i0 || j1 <= j0) return 0.0;
      int64_t k = 0;
      if (u.nseg == 1) k = u.klen >= 0 ? u.klen : S.bcols[u.src_bcol0].width;
      else for (int sg = 0; sg < u.nseg; ++sg) k += S.bcols[u.src_bcol0 + sg].width;
      if (u.mode == MODE_TRSM) return (double)(i1 - i0) * (j1 - j0) * ((double)u.N + 2.0 * (double)(k - u.N));
      double cnt = 0;
      // (the reporting convention of direct_flops: a DIRECT unit is cut at the diagonal only when
      // it starts on it; inter-node units always are)
      const bool cut = lower && u.lower && (u.mode != MODE_DIRECT || u.src_r0 == u.src_c0);
      if (!cut) cnt = (double)(i1 - i0) * (j1 - j0);
      else
        for (int j = j0; j < j1; ++j) {
          const int lo = std::max(i0, u.src_c0 + j - u.src_r0);   // first row with src_r0 + i >= src_c0 + j
          if (lo < i1) cnt += i1 - lo;
        }
      return 2.0 * (double)k * cnt;
    };
    (void)flops;
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
      L.flops = 0;
      for (const UpdTile& t : tv) L.flops += tile_flops(t, edges[pass]);
      L.stream = e.stream;
      if (pass == first_nonempty) {
        L.add_wait(e.wait0);
        L.add_wait(e.wait1);
        L.add_wait(e.wait2);
      }
      L.record = pass == last_nonempty ? e.record : -1;
      L.overlap = e.overlap;
      L.lat = e.lat;
      P.tiles.insert(P.tiles.end(), tv.begin(), tv.end());
      P.launches.push_back(L);
    }
    us.clear();
  }

  // Inter-node update units of node s, one per touched ancestor block column,
  // covering every K segment (block column) of s.
  // dist2 (phase 2 of a program with a distributed top tree): only the destination block columns
  // this rank owns
  bool dist2 = false;
  int64_t xbcast_base = 0;   // offset of the broadcast area in the exchange buffer (behind the reduce regions)
  bool mine(int b) const { return !dist2 || opt.top_owner[b] == opt.rank; }

  // last >= 0: only the ancestors up to node `last` (the root of s's subtree task)
  void between_templates(int s, std::vector<UpdUnit>& out, int last = -1) {
    const int nn = S.nnodes;
    const int m = S.nrow(s);
    const int* idx = S.rows(s);
    int cptr = S.ncol(s);
    int a = S.sparent[s];
    while (a < nn && cptr < m && (last < 0 || a <= last)) {
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
          u.dinv_ld = S.node_bcol0[a] + cb;   // (unused by the kernel in this mode) destination block column
          if (mine(u.dinv_ld)) out.push_back(u);
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

  // A DIRECT update unit: destination = columns [dc0, dc0+N) x stored rows [dr0, dr0+M) of block
  // column D (global id bd), K = columns [k0, k0+klen) of block column bs of the same node.
  UpdUnit direct_unit(int bs, int k0, int klen, int bd, int dr0, int M, int dc0, int N) const {
    const BlockCol& B = S.bcols[bs];
    const BlockCol& D = S.bcols[bd];
    UpdUnit u{};
    u.b_bcol0 = -1;
    u.lower = 1;
    u.mode = MODE_DIRECT;
    u.d_off = D.off;
    u.d_ld = D.width;
    u.d_row0 = dr0;
    u.d_col0 = dc0;
    u.src_bcol0 = bs;
    u.nseg = 1;
    u.seg_r0 = B.r0;
    u.seg_stride = nb;
    u.src_r0 = D.r0 + dr0;   // node-local row of destination row dr0
    u.src_c0 = D.r0 + dc0;   // node-local row that corresponds to destination column dc0
    u.M = M;
    u.N = N;
    u.k0 = k0;
    u.klen = klen;
    return u;
  }
  static double direct_flops(const UpdUnit& u) {
    // lower part only where the unit straddles the diagonal (src_r0 == src_c0)
    const double full = (double)u.M * u.N;
    const double cut = (u.src_r0 == u.src_c0) ? 0.5 * u.N * (u.N - 1) : 0.0;
    return 2.0 * u.klen * (full - cut);
  }

  // Deterministic inter-node updates: every SCATTER unit becomes a MODE_BUFFER unit whose
  // product lands in the scratch buffer (one launch), then one L_GATHER launch subtracts the
  // buffered blocks from their destination tiles, tile by tile, items in unit order.
  void emit_buffered(int level, std::vector<UpdUnit>& us, double flops, Edge e) {
    struct Key { int bd, rt, ct; };
    std::vector<std::pair<Key, GatherItem>> items;
    int64_t off = 0;
    for (UpdUnit& u : us) {
      const int bd = u.dinv_ld;                 // destination block column (between_templates)
      const BlockCol& D = S.bcols[bd];
      const int dcol_base = u.d_col0;           // pivot position of the block column's column 0
      GatherItem g{};
      g.buf_off = off;
      g.relrow_off = u.relrow_off;
      g.gcol_off = u.gcol_off;
      g.ld = u.N;
      g.diag_shift = u.src_r0 - u.src_c0;
      g.lower = u.lower;
      // runs of rows / columns that fall into the same 64-entry tile of the destination
      std::vector<std::pair<int, int>> rruns, cruns;   // (first index, tile)
      for (int i = 0; i < u.M; ++i) {
        const int t = (P.relpos[u.relrow_off + i] - D.r0) / 64;
        if (rruns.empty() || rruns.back().second != t) rruns.push_back({i, t});
      }
      for (int j = 0; j < u.N; ++j) {
        const int t = (S.rlist[u.gcol_off + j] - dcol_base) / 64;
        if (cruns.empty() || cruns.back().second != t) cruns.push_back({j, t});
      }
      for (size_t a = 0; a < rruns.size(); ++a) {
        g.i0 = rruns[a].first;
        g.i1 = a + 1 < rruns.size() ? rruns[a + 1].first : u.M;
        for (size_t b = 0; b < cruns.size(); ++b) {
          g.j0 = cruns[b].first;
          g.j1 = b + 1 < cruns.size() ? cruns[b + 1].first : u.N;
          if (g.lower && g.diag_shift + g.i1 - 1 < g.j0) continue;   // entirely above the diagonal
          items.push_back({Key{bd, rruns[a].second, cruns[b].second}, g});
        }
      }
      // the unit itself: product stored at scratch[off ...], M x N row-major
      u.mode = MODE_BUFFER;
      u.d_off = off;
      u.d_ld = u.N;
      u.d_row0 = 0;
      u.d_col0 = 0;
      off += (int64_t)u.M * u.N;
    }
    P.scratch_size = std::max(P.scratch_size, off);
    std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) {
      if (x.first.bd != y.first.bd) return x.first.bd < y.first.bd;
      if (x.first.rt != y.first.rt) return x.first.rt < y.first.rt;
      return x.first.ct < y.first.ct;
    });
    Launch G;
    G.kind = L_GATHER;
    G.level = level;
    G.first = (int64_t)P.gather_tiles.size();
    G.tile = 64;
    G.flops = 0;
    G.stream = e.stream;
    for (size_t i = 0; i < items.size();) {
      const Key k = items[i].first;
      const BlockCol& D = S.bcols[k.bd];
      GatherTile t{};
      t.d_off = D.off;
      t.d_ld = D.width;
      t.row0 = k.rt * 64;
      t.col0 = k.ct * 64;
      t.rows = std::min(64, D.nrow - t.row0);
      t.cols = std::min(64, D.width - t.col0);
      t.drow_base = D.r0;
      t.dcol_base = S.sptr[D.node] + D.r0;
      t.first = (int)P.gather_items.size();
      for (; i < items.size() && items[i].first.bd == k.bd && items[i].first.rt == k.rt &&
             items[i].first.ct == k.ct; ++i)
        P.gather_items.push_back(items[i].second);
      t.count = (int)P.gather_items.size() - t.first;
      P.gather_tiles.push_back(t);
    }
    G.count = (int64_t)P.gather_tiles.size() - G.first;
    Edge eg = e;
    eg.record = -1;
    emit_gemm(level, us, flops, true, eg);   // (a marker when there is nothing to do)
    P.launches.push_back(G);                 // same stream, in order behind the products
  }

  // ---------------------------------------------------------------------------------------------
  // Subtree tasks (L_SUBTREE).  The reference hands every pruned subtree to ONE task that factorizes
  // its nodes in post-order, keeps what leaves the subtree in a private generated element and adds
  // that to the ancestors once (a20-a25, src/spllt_factorization_mod.F90:39-261).  Here: maximal
  // subtrees whose nodes all have one block column of at most one panel and whose modelled time on
  // ONE CU stays within opt.subtree_us; one workgroup each, all in one launch beside the lowest
  // levels of the rest of the tree.  What it buys: the leaves' updates -- thousands of tiles whose
  // scatter-adds into far, cold ancestors bound the lowest levels -- land in the subtree's own
  // block columns and in its generated element (a few hundred KB, cache-resident while the subtree
  // is being worked on) instead, and the ancestors see one add per entry of the generated element.
  // in_sub[s] = 1 for the nodes the tasks cover; sub_event = the event the launch records (-1: no
  // task, or a single-stream program), sub_lstar = the lowest level an ancestor of a subtree root has.
  // ---------------------------------------------------------------------------------------------
  std::vector<char> in_sub;
  int sub_event = -1, sub_lstar = INT_MAX;
  void subtree_tasks(const std::vector<int64_t>& dinv_slot) {
    const int nn = S.nnodes;
    in_sub.assign((size_t)nn, 0);
    const double budget = (double)env_int("SPLLT_SUBTREE_US", opt.subtree_us);
    if (!env_int("SPLLT_SUBTREES", opt.subtrees ? 1 : 0) || budget <= 0) return;
    for (int s = 0; s < nn; ++s)
      if (S.sparent[s] <= s) return;                        // (children before parents, or no tasks)
    // modelled time of a node inside a task (one CU): Cholesky of the panel, solve of the 64-row
    // blocks below, one 64 x 64 tile of the update per pair of them
    const double kInf = 1e30;
    std::vector<double> cost((size_t)nn);
    for (int s = 0; s < nn; ++s) {
      const int w = S.ncol(s), m = S.nrow(s);
      const bool ok = S.node_bcol0[s + 1] - S.node_bcol0[s] == 1 && w <= pw && w <= kPanelMax && m - w < 32768;
      const double nrb = cdiv(m - w, 64);
      cost[(size_t)s] = ok ? 12.0 + 2.0 * nrb + 2.5 * 0.5 * nrb * (nrb + 1) : kInf;
    }
    for (int s = 0; s < nn; ++s) {
      const int p = S.sparent[s];
      if (p < nn) cost[(size_t)p] = (cost[(size_t)p] < kInf && cost[(size_t)s] < kInf) ? cost[(size_t)p] + cost[(size_t)s] : kInf;
    }
    std::vector<int> root_of((size_t)nn, -1);
    for (int s = nn - 1; s >= 0; --s) {
      if (cost[(size_t)s] > budget) continue;
      const int p = S.sparent[s];
      root_of[(size_t)s] = (p < nn && root_of[(size_t)p] >= 0) ? root_of[(size_t)p] : s;
    }
    std::vector<int> roots;
    for (int s = 0; s < nn; ++s)
      if (root_of[(size_t)s] == s) roots.push_back(s);
    if (roots.empty()) return;
    // longest first: the launch ends with its longest task
    std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<std::vector<int>> members((size_t)nn);
    for (int s = 0; s < nn; ++s)
      if (root_of[(size_t)s] >= 0) {
        members[(size_t)root_of[(size_t)s]].push_back(s);
        in_sub[(size_t)s] = 1;
      }
    Launch L;
    L.kind = L_SUBTREE;
    L.level = 0;
    L.first = (int64_t)P.sub_tasks.size();
    L.tile = 64;
    L.flops = 0;
    auto push_unit = [&](UpdUnit u) {
      u.a_w = S.bcols[u.src_bcol0].width;
      u.a_off = S.bcols[u.src_bcol0].off;
      P.units.push_back(u);
      const double fl = 2.0 * u.a_w * ((double)u.M * u.N - 0.5 * u.N * (u.N - 1));
      P.flops_between += fl;
      L.flops += fl;
    };
    for (int r : roots) {
      const int wr = S.ncol(r), mr = S.nrow(r);
      const int* ridx = S.rows(r);
      const int rend = S.sptr[r + 1] - 1;                   // the root's last column
      SubTask T{};
      T.node_first = (int)P.sub_nodes.size();
      T.g_n = members[(size_t)r].size() > 1 ? mr - wr : 0;
      T.g_off = P.gen_size;
      P.gen_size += (int64_t)T.g_n * (T.g_n + 1) / 2;
      if (S.sparent[r] < nn) sub_lstar = std::min(sub_lstar, S.level[S.sparent[r]]);
      for (int s : members[(size_t)r]) {
        const int b = S.node_bcol0[s];
        const BlockCol& B = S.bcols[b];
        const int w = S.ncol(s), m = S.nrow(s);
        SubNode N{};
        N.off = B.off;
        N.dinv_off = dinv_slot[(size_t)b];
        N.w = w;
        N.nrow = m;
        N.gcol = S.sptr[s];
        N.unit_first = (int)P.units.size();
        N.root = s == r ? 1 : 0;
        std::vector<UpdUnit> us;
        between_templates(s, us, s == r ? -1 : r);          // the root: all its ancestors; the others: up to the root
        for (const UpdUnit& u : us) push_unit(u);
        if (s != r && T.g_n > 0) {
          // what leaves the subtree: the rows of s beyond the root's columns, against themselves
          const int* idx = S.rows(s);
          int cout = w;
          while (cout < m && idx[cout] <= rend) cout++;
          if (cout < m) {
            const int64_t base = (int64_t)P.relpos.size();
            int q = wr;
            for (int i = cout; i < m; ++i) {
              while (q < mr && ridx[q] < idx[i]) q++;
              if (q >= mr || ridx[q] != idx[i]) {
                std::fprintf(stderr, "spllt-hip: structure inclusion violated (node %d -> subtree root %d)\n", s, r);
                q = std::min(q, mr - 1);
              }
              P.relpos.push_back(q - wr);
            }
            UpdUnit u{};
            u.b_bcol0 = -1;
            u.lower = 1;
            u.mode = MODE_GEN;
            u.d_off = T.g_off;
            u.d_ld = 0;
            u.relrow_off = base;
            u.gcol_off = base;
            u.src_bcol0 = b;
            u.nseg = 1;
            u.seg_r0 = 0;
            u.seg_stride = nb;
            u.src_r0 = cout;
            u.src_c0 = cout;
            u.M = m - cout;
            u.N = m - cout;
            u.k0 = 0;
            u.klen = -1;
            u.dinv_ld = -1;
            push_unit(u);
          }
        }
        N.unit_count = (int)P.units.size() - N.unit_first;
        P.sub_nodes.push_back(N);
        const double fp = (double)w * w * w / 3.0, ft = (double)(m - w) * w * w;
        P.flops_potrf += fp;
        P.flops_trsm += ft;
        L.flops += fp + ft;
      }
      T.node_count = (int)P.sub_nodes.size() - T.node_first;
      P.sub_tasks.push_back(T);
    }
    L.count = (int64_t)P.sub_tasks.size() - L.first;
    L.stream = opt.lookahead ? ST_SIDE : ST_CHAIN;
    if (opt.lookahead) {
      sub_event = P.nevents++;
      L.record = sub_event;
    }
    P.launches.push_back(L);
  }

  void run() {
    P.pw = pw;
    // A block-column step whose block columns are wider than one panel runs in CHAIN BLOCKS of up
    // to four panels: the whole diagonal block of the chain block is factored by ONE workgroup
    // (L_CHAIN4, k_chain_block) and the rows below are solved against it by one launch (L_TRSM4,
    // k_trsm_rows) -- two dependent launches per 4 pw columns of the panel chain instead of twelve
    // (POTRF, TRSM, in-panel update per panel).  chain4 off: one panel per chain step.  The layout
    // of the inverses is per panel either way (cb = pw).
    const bool chain4 = env_int("SPLLT_CHAIN4", opt.chain4 ? 1 : 0) != 0;
    const int cb = pw;                       // layout of the dinv slots: one pn x pn inverse per panel
    P.cb = cb;
    const int nn = S.nnodes;
    int maxlevel = -1;
    for (int s = 0; s < nn; ++s) maxlevel = std::max(maxlevel, S.level[s]);

    // dinv slots: one Winv per (block column, panel), see ChainUnit
    std::vector<int64_t> dinv_slot(S.nbcol() + 1, 0);
    {
      int64_t o = 0;
      for (int b = 0; b < S.nbcol(); ++b) {
        dinv_slot[b] = o;
        const int w = S.bcols[b].width;
        o += winv_total(w, cb);
      }
      dinv_slot[S.nbcol()] = o;
      P.dinv_size = o;
    }
    const int64_t fused_max = env_int("SPLLT_FUSED_PANEL_MAX", opt.fused_panel_max);
    int super_panel = (int)env_int("SPLLT_SUPER_PANEL", opt.super_panel);
    if (super_panel % (4 * pw) != 0) super_panel = 0;      // (a multiple of the chain block, or off)

    const bool la = opt.lookahead;
    const bool det_all = opt.deterministic;
    // levels below this one assemble their inter-node updates through the buffer + ordered gather
    // (no atomics) even in the default engine: see ScheduleOptions::buffer_levels
    const int buffer_levels = (int)env_int("SPLLT_BUFFER_LEVELS", opt.buffer_levels);
    const bool pair_sources = env_int("SPLLT_PAIR_TRAILING", opt.pair_sources ? 1 : 0) != 0;
    auto edge = [&](int stream) {
      Edge e;
      e.stream = la ? stream : ST_CHAIN;
      return e;
    };
    std::vector<UpdUnit> us;
    int ev_level = -1;       // event that covers everything of the levels processed so far
    std::vector<std::pair<int, int>> zone_events;  // (zone, event) of the last level's inter-node updates
    bool zoned = false;      // ... and they were really split into zones
    // A partitioned program (multi-GPU) runs the rank's own subtrees first, then
    // an EXCHANGE marker (the extend-add of the top-tree block columns across
    // ranks happens there), then the replicated top tree.
    const bool partitioned = opt.node_owner != nullptr && opt.nranks > 1;
    const int nphase = partitioned ? 2 : 1;
    in_sub.assign((size_t)nn, 0);
    if (!partitioned && !det_all && buffer_levels == 0) subtree_tasks(dinv_slot);
    bool sub_waited = sub_event < 0;
    for (int ph = 0; ph < nphase; ++ph) {
    std::vector<std::vector<int>> by_level(maxlevel + 1);
    for (int s = 0; s < nn; ++s) {
      bool take = true;
      if (partitioned) take = (ph == 0) ? (opt.node_owner[s] == opt.rank) : (opt.node_owner[s] < 0);
      if (take && !in_sub[(size_t)s]) by_level[S.level[s]].push_back(s);
    }
    dist2 = partitioned && ph == 1 && opt.top_owner != nullptr;
    // event of the extend-add that delivers the block columns of a top-tree level (distributed top
    // tree, multi-stream program: one reduce-scatter per level, see below); -1: nothing to wait for
    std::vector<int> xlev_event((size_t)maxlevel + 2, -1);
    int xlev_last = -1;
    bool xpipe = false;
    if (partitioned && ph == 1) {
      // extend-add of the top-tree block columns across the ranks
      std::vector<int> top;
      for (int b = 0; b < S.nbcol(); ++b)
        if (opt.node_owner[S.bcols[b].node] < 0) top.push_back(b);
      auto push_exchange = [&](const Exchange& E, int stream, int wait_ev) {
        Launch X;
        X.kind = L_EXCHANGE;
        X.level = -1;
        X.first = (int64_t)P.exchanges.size();
        X.count = 0;
        P.exchanges.push_back(E);
        X.tile = 0;
        X.flops = 0;
        X.stream = stream;
        X.add_wait(wait_ev);
        const int ev = P.nevents++;
        X.record = ev;
        P.launches.push_back(X);
        return ev;
      };
      if (!dist2) {
        Exchange E{};
        E.first_item = (int)P.xitems.size();
        E.kind = X_REDUCE_ALL;
        int64_t o = 0;
        for (int b : top) {
          const int64_t cnt = (int64_t)S.bcols[b].nrow * S.bcols[b].width;
          P.xitems.push_back(ExchangeItem{b, -1, o, cnt, S.bcols[b].off, 0});
          o += cnt;
        }
        E.elems = o + 1;   // + the "not positive definite" indicator
        E.chunk = 0;
        E.nitems = (int)P.xitems.size() - E.first_item;
        P.xbuf_elems = std::max(P.xbuf_elems, E.elems);
        const int ev = push_exchange(E, ST_CHAIN, ev_level);   // (waits for everything of phase 1, all streams)
        ev_level = ev;
        zone_events.clear();
        zone_events.push_back({0, ev});   // phase 2 starts behind the exchange
      } else {
        // Distributed top tree: reduce-scatter to the owners -- PIPELINED BY LEVEL of the top tree
        // ("pipeline per ancestor node so the reduce overlaps"; the reference's walk is
        // per destination tile, src/spllt_factorization_mod.F90:39-191).  One exchange per level,
        // lowest first, each with its own region of the buffer (rank r's chunk of level l at
        // base_l + r chunk_l; Exchange::elems = the END of the region, i.e. base_l = elems -
        // nranks chunk_l), issued back to back on the SIDE stream of the multi-stream program: a
        // step of top level l waits for chunk l only, so the lowest top level's panel chains run
        // while the (larger) chunks of the levels above are still on the wire.  What lands in a
        // level (the inter-node updates of the levels below it) waits for that level's chunk too:
        // the unpack overwrites.  The single-stream program issues them in a row on its one stream.
        std::vector<int> tlev;
        for (int b : top) tlev.push_back(S.level[S.bcols[b].node]);
        std::sort(tlev.begin(), tlev.end());
        tlev.erase(std::unique(tlev.begin(), tlev.end()), tlev.end());
        xpipe = la;
        int64_t base = 0;
        int wait_ev = ev_level;
        for (int tl : tlev) {
          Exchange E{};
          E.first_item = (int)P.xitems.size();
          E.kind = X_REDUCE_OWNER;
          std::vector<int64_t> fill((size_t)opt.nranks, 0);
          for (int b : top)
            if (S.level[S.bcols[b].node] == tl) fill[(size_t)opt.top_owner[b]] += (int64_t)S.bcols[b].nrow * S.bcols[b].width;
          int64_t chunk = 1;
          for (int64_t v : fill) chunk = std::max(chunk, v);
          std::fill(fill.begin(), fill.end(), 0);
          for (int r = 0; r < opt.nranks; ++r)
            for (int b : top) {
              if (S.level[S.bcols[b].node] != tl || opt.top_owner[b] != r) continue;
              const int64_t cnt = (int64_t)S.bcols[b].nrow * S.bcols[b].width;
              P.xitems.push_back(ExchangeItem{b, r, base + (int64_t)r * chunk + fill[(size_t)r], cnt, S.bcols[b].off, 0});
              fill[(size_t)r] += cnt;
            }
          E.chunk = chunk;
          base += chunk * opt.nranks;
          E.elems = base;
          E.nitems = (int)P.xitems.size() - E.first_item;
          const int ev = push_exchange(E, xpipe ? ST_SIDE : ST_CHAIN, wait_ev);
          wait_ev = -1;                  // (the following ones are behind it in their stream)
          xlev_event[(size_t)tl] = ev;
          xlev_last = ev;
        }
        P.xbuf_elems = std::max(P.xbuf_elems, base);
        xbcast_base = base;              // the broadcasts of phase 2 use the buffer behind the reduce regions
        if (!xpipe) {
          ev_level = xlev_last;
          zone_events.clear();
          zone_events.push_back({0, xlev_last});
        } else {
          // every level of phase 2 waits for its own chunk (below); nothing of phase 1 is left to wait for
          ev_level = xlev_event[(size_t)tlev.front()];
          zone_events.clear();
          zone_events.push_back({0, ev_level});
        }
      }
    }
    for (int lev = 0; lev <= maxlevel; ++lev) {
      const auto& nodes = by_level[lev];
      if (nodes.empty()) continue;
      const bool det = det_all || lev < buffer_levels;
      int maxnc = 0;
      for (int s : nodes) maxnc = std::max(maxnc, S.node_bcol0[s + 1] - S.node_bcol0[s]);
      // Zones: the inter-node updates at the end of the previous level were issued sorted by
      // destination -- first everything that lands in block column 0 of the nodes of THIS level,
      // then block column 1, ... -- with an event per zone (zone_events).  Step c of this level
      // only waits for zone c, so the panel chains of a level run beside the bulk of the
      // inter-node updates of the level below instead of after it.  zev(c): the event that
      // covers every update into block column c of this level's nodes.
      const std::vector<std::pair<int, int>> zones_in = zone_events;
      auto zev = [&](int c) {
        int ev = -1;
        for (const auto& z : zones_in)
          if (z.first <= c) ev = z.second;
        return ev;
      };
      const bool pipelined = la && zoned;   // zones of the level below may still be running
      std::vector<int> evB_hist;    // bulk event of every step of this level (-1: none)
      int evD_last = -1;            // "block column done" event of the last finished chunk
      int evF_last = -1;            // last far-stream event (early inter-node slices)
      // inter-node update units of every node of the level (all K segments); they
      // are issued in slices as the block columns they read become final
      std::vector<std::vector<UpdUnit>> tmpl(nodes.size());
      std::vector<int> emitted(nodes.size(), 0);
      for (size_t i = 0; i < nodes.size(); ++i) between_templates(nodes[i], tmpl[i]);
      if (!sub_waited && lev >= sub_lstar) {
        // the first level that holds ancestors of subtree roots: everything from here on is behind
        // the subtree tasks (this marker in the chain stream's order, the other streams through the
        // chain's events); what runs below this level only shares atomic destinations with them
        std::vector<UpdUnit> none;
        Edge e = edge(ST_CHAIN);
        e.wait0 = sub_event;
        emit_gemm(lev, none, 0.0, true, e);
        sub_waited = true;
      }
      if (xpipe && xlev_event[(size_t)lev] >= 0) {
        // pipelined extend-add: the block columns of this top-tree level are here when their own
        // reduce-scatter is (every launch of the level is behind this marker: in the chain stream's
        // order, or through the chain's events)
        std::vector<UpdUnit> none;
        Edge e = edge(ST_CHAIN);
        e.wait0 = xlev_event[(size_t)lev];
        emit_gemm(lev, none, 0.0, true, e);
      }
      for (int c = 0; c < maxnc; ++c) {
        int maxw = 0;
        for (int s : nodes) {
          const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
          if (c < nc) maxw = std::max(maxw, S.bcols[S.node_bcol0[s] + c].width);
        }
        // chain blocks of up to four panels for a step that has block columns of several panels
        // (a step of one-panel block columns is what it always was: one fused launch, or POTRF + TRSM)
        const bool blk4 = chain4 && maxw > pw;
        const int cstep = blk4 ? 4 * pw : pw;    // columns one chunk of the step covers
        const int ng = cdiv(maxw, cstep);
        // Latency-bound step (few row blocks below the panels): one fused launch per panel
        bool fuse_c = false;
        auto row_blocks_of_step = [&](int cc) {
          int64_t nt = 0;
          for (int s : nodes) {
            const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (cc >= nc) continue;
            if (!mine(S.node_bcol0[s] + cc)) continue;
            const BlockCol& B = S.bcols[S.node_bcol0[s] + cc];
            nt += std::max(1, cdiv(B.nrow - std::min(pw, B.width), 64));
          }
          return nt;
        };
        if (opt.fused_panel && !blk4 && pw <= 64) fuse_c = row_blocks_of_step(c) <= fused_max;
        const int evB_c2 = (la && c >= 2) ? evB_hist[c - 2] : -1;   // bulk (c-2 -> c..)
        const int evB_c1 = (la && c >= 1) ? evB_hist[c - 1] : -1;   // bulk (c-1 -> c+1..)
        for (int g = 0; g < ng; ++g) {
          const int cs = g * cstep;
          // "block column final" event: only after the last chunk of the step (what the bulk / far
          // streams and the level end wait for); it rides on the chunk's last chain-stream launch --
          // a launch of its own per panel (an event record behind every kernel of the chain)
          // costs the chain stream ~5 us each
          int evD = -1;
          if (la && g + 1 == ng) {
            evD = P.nevents++;
            evD_last = evD;
          }
          if (fuse_c) {
            // the whole panel step in one launch (k_panel): POTRF, the rows below, and the
            // left-looking update of the next panel's columns
            Launch L;
            L.kind = L_PANEL;
            L.level = lev;
            L.first = (int64_t)P.tiles.size();
            L.tile = 64;
            double fl = 0;
            for (int s : nodes) {
              const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              const int b = S.node_bcol0[s] + c;
              if (!mine(b)) continue;
              const BlockCol& B = S.bcols[b];
              const int c0 = cs;
              if (c0 >= B.width) continue;
              const int pn = std::min(pw, B.width - c0);
              const int next_pn = std::min(pw, B.width - c0 - pn);
              const int below = B.nrow - c0 - pn;
              PanelUnit u{};
              u.off = B.off;
              u.dinv_off = dinv_slot[b] + winv_offset(B.width, pw, cb, c0 / pw);
              u.ld = B.width;
              u.c0 = c0;
              u.pn = pn;
              u.next_pn = next_pn;
              u.nrow = B.nrow;
              u.gcol = S.sptr[s] + B.r0 + c0;
              const int nt = std::max(1, cdiv(below, 64));
              u.ntile = nt;
              u.pad_ = 0;
              const int ui = (int)P.panel_units.size();
              P.panel_units.push_back(u);
              for (int t = 0; t < nt; ++t) P.tiles.push_back(UpdTile{ui, (short)t, 0});
              const double fp = (double)pn * pn * pn / 3.0, ft = (double)below * pn * pn;
              const double fu = 2.0 * (c0 + pn) * ((double)below * next_pn - 0.5 * next_pn * (next_pn - 1));
              P.flops_potrf += fp;
              P.flops_trsm += ft;
              P.flops_update += fu;
              fl += fp + ft + fu;
            }
            L.count = (int64_t)P.tiles.size() - L.first;
            L.flops = fl;
            L.stream = ST_CHAIN;
            if (la && g == 0) {
              L.add_wait(zev(c));    // every inter-node update into block column c
              L.add_wait(evB_c2);    // bulk update (c-2 -> c..) wrote this tile
            }
            L.record = evD;
            // (a rank that owns nothing of this step still carries its waits and its event: what
            // follows on the other streams is ordered behind the updates into block column c
            // through them)
            if (L.count > 0 || L.record >= 0 || L.wait[0] >= 0) P.launches.push_back(L);
          } else {
            // (1) chain step: the diagonal block of the chunk, one workgroup per node -- one panel
            // (k_chain_potrf) or a chain block of up to four (k_chain_block)
            {
              Launch L;
              L.kind = blk4 ? L_CHAIN4 : L_CHAIN;
              L.level = lev;
              L.first = (int64_t)P.chain_units.size();
              L.tile = 0;
              double fl = 0;
              for (int s : nodes) {
                const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
                if (c >= nc) continue;
                const int b = S.node_bcol0[s] + c;
                if (!mine(b)) continue;
                const BlockCol& B = S.bcols[b];
                const int c0 = cs;
                if (c0 >= B.width) continue;
                const int ce = std::min(B.width, cs + cstep);
                const int pn = ce - c0;
                ChainUnit u{};
                u.off = B.off;
                u.winv_off = dinv_slot[b] + winv_offset(B.width, pw, cb, c0 / pw);
                u.ld = B.width;
                u.c0 = c0;
                u.pn = pn;
                u.cs = cs;
                u.ce = ce;
                u.gcol = S.sptr[s] + B.r0 + c0;
                P.chain_units.push_back(u);
                // the factorization of the pn x pn diagonal block, split as the one-panel steps
                // would count it: Cholesky of the panels, solve and update of the blocks between
                double fp = 0, ft = 0, fu = 0;
                for (int q0 = 0; q0 < pn; q0 += pw) {
                  const int qn = std::min(pw, pn - q0), below = pn - q0 - qn;
                  fp += (double)qn * qn * qn / 3.0;
                  ft += (double)below * qn * qn;
                  fu += (double)qn * ((double)below * (below + 1));
                }
                P.flops_potrf += fp;
                P.flops_trsm += ft;
                P.flops_update += fu;
                fl += fp + ft + fu;
              }
              L.count = (int64_t)P.chain_units.size() - L.first;
              L.flops = fl;
              L.stream = ST_CHAIN;
              if (la && g == 0) {
                L.add_wait(zev(c));    // every inter-node update into block column c
                L.add_wait(evB_c2);    // bulk update (c-2 -> c..) wrote this tile
              }
              if (L.count > 0 || L.wait[0] >= 0) P.launches.push_back(L);   // (no unit of ours: the waits stay)
            }
            // (2) rows below the chunk: X = A(:, chunk) L_dd^-T
            double fl = 0;
            for (int s : nodes) {
              const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
              if (c >= nc) continue;
              const int b = S.node_bcol0[s] + c;
              if (!mine(b)) continue;
              const BlockCol& B = S.bcols[b];
              const int c0 = cs;
              if (c0 >= B.width) continue;
              const int ce = std::min(B.width, cs + cstep);
              const int pn = ce - c0;
              const int rows = B.nrow - ce;
              if (rows <= 0) continue;
              UpdUnit u{};
              u.b_bcol0 = -1;
              u.mode = MODE_TRSM;
              u.lower = 0;
              u.d_off = B.off;
              u.d_ld = B.width;
              u.d_row0 = ce;
              u.d_col0 = c0;
              u.src_bcol0 = b;
              u.nseg = 1;
              u.seg_r0 = B.r0;
              u.seg_stride = nb;
              u.src_r0 = B.r0 + ce;
              u.src_c0 = 0;
              u.M = rows;
              u.N = pn;
              u.k0 = cs;
              u.klen = pn;
              u.dinv_off = dinv_slot[b] + winv_offset(B.width, pw, cb, c0 / pw);
              u.dinv_ld = winv_ld(B.width, cb, c0);
              us.push_back(u);
              // (the count of the one-panel steps: triangular solves of the panels + the left-looking
              // products between them)
              double ft = 0, fu = 0;
              for (int q0 = 0; q0 < pn; q0 += pw) {
                const int qn = std::min(pw, pn - q0);
                ft += (double)rows * qn * qn;
                fu += 2.0 * rows * qn * q0;
              }
              P.flops_trsm += ft;
              P.flops_update += fu;
              fl += ft + fu;
            }
            if (blk4) {
              // one workgroup per 64 rows and all columns of the chain block (k_trsm_rows)
              Launch L;
              L.kind = L_TRSM4;
              L.level = lev;
              L.first = (int64_t)P.tiles.size();
              L.tile = 64;
              L.flops = fl;
              L.stream = ST_CHAIN;
              L.record = evD;
              L.lat = 1;
              for (UpdUnit& u : us) {
                const int uid = (int)P.units.size();
                u.a_w = S.bcols[u.src_bcol0].width;
                u.a_off = S.bcols[u.src_bcol0].off;
                P.units.push_back(u);
                for (int ti = 0; ti < cdiv(u.M, 64); ++ti) P.tiles.push_back(UpdTile{uid, (short)ti, 0});
              }
              L.count = (int64_t)P.tiles.size() - L.first;
              if (L.count > 0 || L.record >= 0) P.launches.push_back(L);   // (an empty launch still forwards the event)
              us.clear();
            } else {
              Edge e = edge(ST_CHAIN);
              e.record = evD;        // (an empty launch still forwards the event)
              e.lat = 1;
              emit_gemm(lev, us, fl, false, e);
            }
          }
          // (3b) distributed top tree: the block columns that are final with this chunk go from
          // their owners to everybody (the exchange sits on the chain stream: whatever reads
          // them waits for its event)
          if (dist2) {
            Exchange E{};
            E.kind = X_BCAST;
            E.first_item = (int)P.xitems.size();
            int64_t o = xbcast_base;
            for (int r = 0; r < opt.nranks; ++r)
              for (int s : nodes) {
                const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
                if (c >= nc) continue;
                const int b = S.node_bcol0[s] + c;
                const BlockCol& B = S.bcols[b];
                if (opt.top_owner[b] != r || cs >= B.width || cs + cstep < B.width) continue;
                const int64_t cnt = (int64_t)B.nrow * B.width;
                P.xitems.push_back(ExchangeItem{b, r, o, cnt, B.off, 0});
                o += cnt;
                const int64_t dcnt = dinv_slot[b + 1] - dinv_slot[b];
                P.xitems.push_back(ExchangeItem{b, r, o, dcnt, dinv_slot[b], 1});
                o += dcnt;
              }
            E.nitems = (int)P.xitems.size() - E.first_item;
            E.elems = o;
            if (E.nitems > 0) {
              P.xbuf_elems = std::max(P.xbuf_elems, E.elems);
              Launch X;
              X.kind = L_EXCHANGE;
              X.level = lev;
              X.first = (int64_t)P.exchanges.size();
              X.count = 0;
              X.tile = 0;
              X.flops = 0;
              X.stream = ST_CHAIN;
              P.exchanges.push_back(E);
              if (la) {
                X.add_wait(evD);
                evD = P.nevents++;      // "block column final AND here"
                X.record = evD;
                evD_last = evD;
              }
              P.launches.push_back(X);
            }
          }
          // (4) updates by the finished chunk, left-looking inside the block column: the next
          // chunk's columns by everything left of them (chain stream: small launches that fit
          // the reserved CUs).  After the last chunk the whole block column updates block
          // column c+1 (chain stream) and c+2.. (bulk stream, beside the chain of c+1).
          std::vector<UpdUnit> us_n1, us_n2, us_bulk;
          double fl_n1 = 0, fl_n2 = 0, fl_bulk = 0;
          bool to_next_bcol = false;
          for (int s : nodes) {
            const int nc = S.node_bcol0[s + 1] - S.node_bcol0[s];
            if (c >= nc) continue;
            const int b = S.node_bcol0[s] + c;
            const BlockCol& B = S.bcols[b];
            if (cs >= B.width) continue;
            const int ce = std::min(B.width, cs + cstep);
            if (ce < B.width) {
              if (fuse_c || !mine(b)) continue;   // part of the panel launch / the owner's business
              const int ce2 = std::min(B.width, ce + cstep);
              // Left-looking inside the block column: the next panel's columns by everything left of
              // them.  Block columns wider than a SUPER-PANEL (256 columns) do that only inside the
              // current super-panel; when one is finished it updates ALL remaining columns of the
              // block column at once (right-looking, K = 256, N up to nb - 256: a 128-tile GEMM).
              // Without it the 12 panels of a 768-wide block column re-read everything left of them
              // (K = 64, 128, ... 704 with N = 64: 5.5 passes over the block column, ~15 TFLOP/s).
              UpdUnit n1;
              if (super_panel > 0 && B.width > super_panel && ce % super_panel == 0)
                n1 = direct_unit(b, ce - super_panel, super_panel, b, ce, B.nrow - ce, ce, B.width - ce);
              else if (super_panel > 0 && B.width > super_panel)
                n1 = direct_unit(b, (ce / super_panel) * super_panel, ce % super_panel, b, ce, B.nrow - ce, ce, ce2 - ce);
              else
                n1 = direct_unit(b, 0, ce, b, ce, B.nrow - ce, ce, ce2 - ce);
              us_n1.push_back(n1);
              fl_n1 += direct_flops(n1);
            } else {
              // last chunk: the whole block column updates the node's later block columns
              for (int jj = c + 1; jj < nc; ++jj) {
                const int bd = S.node_bcol0[s] + jj;
                if (!mine(bd)) continue;
                const BlockCol& D = S.bcols[bd];
                if (jj == c + 1) {
                  to_next_bcol = true;
                  UpdUnit n1 = direct_unit(b, 0, B.width, bd, 0, D.nrow, 0, D.width);
                  us_n1.push_back(n1);
                  fl_n1 += direct_flops(n1);
                } else {
                  // Trailing updates two source block columns at a time (K = 2 nb: the destination
                  // is read and written once for both, and the update kernel runs 10-15 % faster
                  // at twice the K): an odd block column c goes together with c-1 into every
                  // destination from c+2 on; an even one only updates c+2 alone (which cannot
                  // wait for c+1), the rest follows with its partner.
                  UpdUnit u = direct_unit(b, 0, B.width, bd, 0, D.nrow, 0, D.width);
                  double fu = direct_flops(u);
                  if (pair_sources) {
                    if ((c & 1) == 0) {
                      if (jj != c + 2) continue;
                    } else {
                      u.src_bcol0 = b - 1;
                      u.nseg = 2;
                      u.seg_r0 = S.bcols[b - 1].r0;
                      u.k0 = 0;
                      u.klen = -1;
                      u.a_off = 0;
                      fu *= (double)(S.bcols[b - 1].width + B.width) / B.width;
                    }
                  }
                  // zones of the level below may still be adding into this block column
                  if (pipelined) u.atomic = 1;
                  (la ? us_bulk : us_n2).push_back(u);
                  (la ? fl_bulk : fl_n2) += fu;
                }
              }
            }
          }
          P.flops_update += fl_n1 + fl_n2 + fl_bulk;
          if (!us_n1.empty()) {
            Edge e = edge(ST_CHAIN);
            e.lat = 1;
            if (la) {
              if (to_next_bcol) {
                e.wait1 = evB_c1;      // bulk (c-1 -> c+1..) writes the same entries
                e.wait2 = zev(c + 1);  // and so do the inter-node updates into block column c+1
              }
            }
            emit_gemm(lev, us_n1, fl_n1, true, e);
          }
          if (!us_n2.empty()) emit_gemm(lev, us_n2, fl_n2, true, edge(ST_CHAIN));   // single-stream program only
          if (g + 1 == ng) {
            int evB = -1;
            if (!us_bulk.empty()) {
              Edge e = edge(ST_BULK);
              e.overlap = 1;
              e.wait0 = evD;
              evB = P.nevents++;
              e.record = evB;
              emit_gemm(lev, us_bulk, fl_bulk, true, e);
            }
            evB_hist.push_back(evB);
            // (4b) early inter-node slices: block columns 0..c are final, so the part
            // of update_between that reads them can run beside the remaining panel
            // chains of the level (far stream) instead of after the last one
            // (pipelined extend-add: no early slices -- they would add into levels whose chunk may
            // still be on the wire, and a far stream that waits for the last chunk holds up the
            // zones of the next level behind it)
            if (la && opt.slice_between && !det && !xpipe && c + 1 < maxnc) {
              std::vector<UpdUnit> sl;
              double fs_ = collect_between(nodes, tmpl, emitted, c + 1, false, sl);
              if (!sl.empty()) {
                P.flops_between += fs_;
                Edge ef = edge(ST_FAR);
                ef.wait0 = evD;
                ef.overlap = 1;
                evF_last = P.nevents++;
                ef.record = evF_last;
                emit_gemm(lev, sl, fs_, true, ef);
              }
            }
          }
        }
      }

      // (5) inter-node updates (update_between + scatter): whatever the slices
      // issued during the panel chains have not covered yet
      std::vector<UpdUnit> rest;
      double fl = collect_between(nodes, tmpl, emitted, INT_MAX, true, rest);
      P.flops_between += fl;
      zone_events.clear();
      zoned = false;
      if (det) {
        // deterministic engine: products into the scratch buffer, then an ordered gather per
        // destination tile (no atomics anywhere)
        Edge e = edge(ST_FAR);
        if (la) {
          e.wait0 = evD_last;
          e.wait1 = ev_level;
          if (xpipe) e.wait2 = xlev_last;     // (the buffered blocks land in every level above)
        }
        emit_buffered(lev, rest, fl, e);
        if (la) {
          ev_level = P.nevents++;
          P.launches.back().record = ev_level;
          zone_events.push_back({0, ev_level});   // every step of the next level waits for all of it
        }
      } else if (!la) {
        emit_gemm(lev, rest, fl);
      } else {
        // far stream, sorted by destination zone (see above); the event of the last zone
        // covers the whole level
        static const int zones_env = (int)env_int("SPLLT_ZONES", -1);
        const bool use_zones = !det && (zones_env >= 0 ? zones_env != 0 : opt.zones);
        auto zone_of = [&](const UpdUnit& u) {
          const int a = S.bcols[u.dinv_ld].node;
          // one zone: every step of the next level waits for all of it (pipelined extend-add: what
          // goes beyond the next level is still a group of its own -- it waits for the last chunk)
          if (!use_zones) return (xpipe && S.level[a] != lev + 1) ? INT_MAX : 0;
          if (S.level[a] != lev + 1) return INT_MAX;
          return u.dinv_ld - S.node_bcol0[a];
        };
        std::stable_sort(rest.begin(), rest.end(),
                         [&](const UpdUnit& x, const UpdUnit& y) { return zone_of(x) < zone_of(y); });
        {
          // marker: the early slices of this level (they may already have covered a whole
          // zone) and the zones of the level below are done -- what a step of the next level
          // waits for when no zone of this level is left for its block column
          std::vector<UpdUnit> none;
          Edge e = edge(ST_FAR);
          e.wait0 = evD_last;   // every block column of the level is final
          e.wait1 = evF_last;   // early slices
          e.wait2 = ev_level;   // the zones of the level below
          ev_level = P.nevents++;
          e.record = ev_level;
          emit_gemm(lev, none, 0.0, true, e);
          zone_events.push_back({-1, ev_level});
          if (xpipe && lev + 1 <= maxlevel && xlev_event[(size_t)lev + 1] >= 0) {
            // the zones of the next level add into block columns that its own chunk overwrites
            Edge e2 = edge(ST_FAR);
            e2.wait0 = xlev_event[(size_t)lev + 1];
            emit_gemm(lev, none, 0.0, true, e2);
          }
        }
        size_t i = 0;
        bool rest_waited = false;
        while (i < rest.size()) {
          const int z = zone_of(rest[i]);
          if (xpipe && z == INT_MAX && !rest_waited) {
            // what goes beyond the next level may land anywhere above: behind the last chunk
            std::vector<UpdUnit> none;
            Edge e2 = edge(ST_FAR);
            e2.wait0 = xlev_last;
            emit_gemm(lev, none, 0.0, true, e2);
            rest_waited = true;
          }
          std::vector<UpdUnit> grp;
          double fz = 0;
          for (; i < rest.size() && zone_of(rest[i]) == z; ++i) {
            const UpdUnit& u = rest[i];
            grp.push_back(u);
            int kcols = 0;
            for (int sg = 0; sg < u.nseg; ++sg) kcols += S.bcols[u.src_bcol0 + sg].width;
            fz += 2.0 * kcols * ((double)u.M * u.N - 0.5 * u.N * (u.N - 1));
          }
          Edge e = edge(ST_FAR);   // in order behind the marker
          ev_level = P.nevents++;
          e.record = ev_level;
          emit_gemm(lev, grp, fz, true, e);
          zone_events.push_back({z, ev_level});
          if (z != 0 || use_zones) zoned = zoned || use_zones;
        }
      }
    }
    }  // phases
    if (partitioned && opt.top_owner != nullptr) {
      // a failed pivot anywhere: every rank learns it at the end
      Exchange E{};
      E.kind = X_FLAG;
      E.first_item = (int)P.xitems.size();
      E.nitems = 0;
      E.elems = 1;
      P.xbuf_elems = std::max<int64_t>(P.xbuf_elems, 1);
      Launch X;
      X.kind = L_EXCHANGE;
      X.level = -1;
      X.first = (int64_t)P.exchanges.size();
      X.count = 0;
      X.tile = 0;
      X.flops = 0;
      X.stream = ST_CHAIN;
      P.exchanges.push_back(E);
      X.add_wait(ev_level);
      ev_level = P.nevents++;
      X.record = ev_level;
      P.launches.push_back(X);
    }
    if (!sub_waited) {
      // nothing above the subtree tasks (they cover whole trees of the forest): the end waits for them
      std::vector<UpdUnit> none;
      Edge e = edge(ST_FAR);
      e.wait0 = sub_event;
      e.wait1 = ev_level;
      ev_level = P.nevents++;
      e.record = ev_level;
      emit_gemm(maxlevel, none, 0.0, true, e);
    }
    P.final_event = ev_level;
  }
};

}  // namespace

bool latency_bound(const Symbolic& S, int pw) {
  int maxlevel = -1;
  for (int s = 0; s < S.nnodes; ++s) maxlevel = std::max(maxlevel, S.level[s]);
  std::vector<int> widest(maxlevel + 1, 0);
  for (int s = 0; s < S.nnodes; ++s) widest[S.level[s]] = std::max(widest[S.level[s]], S.ncol(s));
  double chain_us = 0;
  for (int w : widest) chain_us += 60.0 * ((w + pw - 1) / pw);
  const double bulk_us = (double)S.flops / 45e6;   // 45 TFLOP/s
  return chain_us > 0.25 * bulk_us;
}

void assign_top_owners(const Symbolic& S, const std::vector<int>& node_owner, int nranks,
                       std::vector<int>& top_owner) {
  top_owner.assign((size_t)S.nbcol(), -1);
  if (nranks < 1) nranks = 1;
  int maxlevel = -1;
  for (int s = 0; s < S.nnodes; ++s) maxlevel = std::max(maxlevel, S.level[s]);
  std::vector<std::vector<int>> by_level((size_t)maxlevel + 1);
  for (int s = 0; s < S.nnodes; ++s)
    if (node_owner[(size_t)s] < 0) by_level[(size_t)S.level[s]].push_back(s);
  int next = 0;
  for (const auto& nodes : by_level) {
    int maxnc = 0;
    for (int s : nodes) maxnc = std::max(maxnc, S.node_bcol0[s + 1] - S.node_bcol0[s]);
    for (int c = 0; c < maxnc; ++c)
      for (int s : nodes)
        if (c < S.node_bcol0[s + 1] - S.node_bcol0[s]) top_owner[(size_t)(S.node_bcol0[s] + c)] = next++ % nranks;
  }
}

bool distribute_top_tree(const Symbolic& S, const std::vector<int>& node_owner, int nranks) {
  if (nranks < 2) return false;
  // Model of the top-tree phase (the same constants as latency_bound): its panel chains are a
  // dependent sequence either way -- one ~60 us step per 64 columns of the widest node of every
  // level -- and its flops (the reference's symbolic count, spllt_analyse_mod:1007-1023) are
  // repeated on every rank when replicated, shared when distributed, which in turn pays a
  // broadcast (~100 us: pack, launch, hand-over, unpack) per block-column step:
  //   replicated   max(chain, flops / rate)
  //   distributed  max(chain + steps * 100 us, flops / (nranks * rate))
  // A chain-bound top tree (small problems) stays replicated.
  int maxlevel = -1;
  for (int s = 0; s < S.nnodes; ++s) maxlevel = std::max(maxlevel, S.level[s]);
  std::vector<int> widest((size_t)maxlevel + 1, 0), nsteps((size_t)maxlevel + 1, 0);
  double top_flops = 0;
  for (int s = 0; s < S.nnodes; ++s) {
    if (node_owner[(size_t)s] >= 0) continue;
    const double m = S.nrow(s), n = S.ncol(s);
    for (int j = 1; j <= (int)n; ++j) top_flops += (m - n + j) * (m - n + j);
    widest[(size_t)S.level[s]] = std::max(widest[(size_t)S.level[s]], S.ncol(s));
    nsteps[(size_t)S.level[s]] = std::max(nsteps[(size_t)S.level[s]], S.node_bcol0[s + 1] - S.node_bcol0[s]);
  }
  double chain_us = 0, steps = 0;
  for (int l = 0; l <= maxlevel; ++l) {
    chain_us += 60.0 * ((widest[(size_t)l] + 63) / 64);
    steps += nsteps[(size_t)l];
  }
  const double flops_us = top_flops / 45e6;   // 45 TFLOP/s
  const double t_rep = std::max(chain_us, flops_us);
  const double t_dist = std::max(chain_us + 100.0 * steps, flops_us / nranks);
  return t_dist < 0.9 * t_rep;
}

void build_program(const Symbolic& S, const ScheduleOptions& opt, Program& P) {
  P = Program();
  Builder b(S, opt, P);
  b.run();
}

void build_solve_program(const Symbolic& S, int pw, int cb, SolveProgram& P, const int* node_owner,
                         int rank) {
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
      u.cb = cb;
      u.gcol0 = S.sptr[B.node] + B.r0;
      u.pad_ = 0;
      o += winv_total(B.width, cb);
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
