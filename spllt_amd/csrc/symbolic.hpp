// Symbolic phase of the spllt-hip engine (host side, plain C++17).
//
// Produces everything the factorize hot path consumes:
//   * a fill-reducing pivot order (built-in nested dissection, or a caller
//     supplied one),
//   * the assembly tree with relaxed (nemin) supernodes and sorted row lists
//     -- the same quintuple (sptr, sparent, rptr, rlist, order) SpLLT takes
//     from SSIDS (reference src/spllt_analyse_mod.F90:129-158),
//   * SpLLT's tile layout of every supernode (reference
//     src/spllt_analyse_mod.F90:305-469), flattened into
//     one arena with a 64-bit offset per block column,
//   * the user-val -> L scatter map (reference spllt_make_map/spllt_lcol_map,
//     src/spllt_analyse_mod.F90:1033-1171),
//   * subtree flop weights (spllt_symbolic, :990-1029) and the pruned-subtree
//     marking (spllt_prune_tree, :806-987) used as the multi-GPU partition.
//
// All indices in this header are 0-based; the C-ABI converts at the edge.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace spx {

struct SymOptions {
  int nb = 256;            // tile size (spllt_options%nb)
  int nemin = 32;          // supernode amalgamation threshold
  bool prune_tree = true;  // spllt_options%prune_tree
  int ncpu = 1;            // pruning target (number of workers / GPUs)
  int nd_leaf = 32;        // nested-dissection leaf size (built-in ordering)
  double relax = 0.05;     // relaxed amalgamation: merge while explicit zeros <= relax * entries
};

// One block column of L: a row-major (nrow x width) matrix, rows r0..m-1 of
// its supernode, whose first `width` rows are the (lower-triangular) diagonal
// square.  Tiles are row ranges of `nb` rows of it.
struct BlockCol {
  int node;        // owning supernode
  int width;       // blkn
  int r0;          // node-local index of the first row (= (c-1)*nb)
  int nrow;        // rows stored (m - r0)
  int64_t off;     // offset of the first entry in the arena
  int64_t blk0;    // 0-based id of the diagonal tile
};

struct Symbolic {
  int n = 0;
  int64_t nnzA = 0;
  int nnodes = 0;
  int nb = 0;
  std::string ordering;  // "nd-bfs" | "user" | "symbolic"

  std::vector<int> order;    // order[var] = pivot position
  std::vector<int> porder;   // porder[pos] = var
  std::vector<int> sptr;     // nnodes+1, first pivot position of each node
  std::vector<int> sparent;  // nnodes, parent id (== nnodes for roots: virtual root)
  std::vector<int64_t> rptr; // nnodes+1
  std::vector<int> rlist;    // sorted pivot positions; first ncol are the node's own
  std::vector<int> snode_of; // n, pivot position -> node

  // tree
  std::vector<int> least_desc;           // nnodes
  std::vector<int> level;                // nnodes, 0 = leaf; parent > max(children)
  std::vector<int> child_ptr, child_idx; // CSR, nnodes+2 (virtual root at nnodes)
  std::vector<int64_t> weight;           // nnodes+1 subtree flops (virtual root last)
  std::vector<int> small;                // nnodes: 0 / 1 / -(root+1)  [0-based root encoded as -(root+1)]
  int maxdepth = 0;

  // tiling
  std::vector<int> node_bcol0;    // nnodes+1, first block column of node
  std::vector<BlockCol> bcols;    // nbcol
  int64_t nblk = 0;               // total number of tiles (final_blk)
  int maxmn = 0;
  int64_t arena = 0;              // total doubles in all lcol arrays

  // user val -> arena scatter (assignment semantics)
  std::vector<int64_t> map_dst;
  std::vector<int64_t> map_src;
  std::vector<int64_t> lmap_ptr;  // nbcol+1 ranges of map_* per block column

  // statistics
  int64_t nnzL = 0;   // sum_nodes sum_j (m-n+j)
  int64_t flops = 0;  // sum_nodes sum_j (m-n+j)^2   (reference's F_sym)

  int ncol(int s) const { return sptr[s + 1] - sptr[s]; }
  int nrow(int s) const { return (int)(rptr[s + 1] - rptr[s]); }
  const int* rows(int s) const { return rlist.data() + rptr[s]; }
  int nbcol() const { return (int)bcols.size(); }
};

// Full analyse.  ptr/row: CSC of the lower triangle (0-based), n columns.
// user_order: optional (size n) pivot position of each variable; nullptr ->
// built-in nested dissection.  Returns 0 or a negative SpLLT error flag.
int analyse(int n, const int64_t* ptr, const int* row, const int* user_order,
            const SymOptions& opt, Symbolic& S);

// Analyse from a caller-supplied symbolic factorization -- the quintuple SpLLT takes from
// SSIDS (reference src/spllt_analyse_mod.F90:129-158): all 0-based here; order[var] = pivot
// position, nodes postordered with contiguous column ranges sptr, sparent[s] > s (== nnodes
// for roots), row lists sorted with the node's own columns first.  No supernode detection,
// amalgamation or reordering happens: the factorization uses exactly this partition.
int analyse_symbolic(int n, const int64_t* ptr, const int* row, int nnodes, const int* sptr,
                     const int* sparent, const int64_t* rptr, const int* rlist, const int* order,
                     const SymOptions& opt, Symbolic& S);
int finish_symbolic(int n, const int64_t* ptr, const int* row, const std::vector<int64_t>* xadj,
                    const std::vector<int>* adj, const SymOptions& opt, Symbolic& S);

// Exposed for tests.
void nested_dissection(int n, const std::vector<int64_t>& xadj,
                       const std::vector<int>& adj, int leaf, std::vector<int>& order);
void prune_tree(Symbolic& S, int nth);
// Multi-GPU partition from the pruning marks: every pruned subtree (small == 1
// root and its members) goes to one rank, heaviest first onto the least loaded
// rank; nodes outside pruned subtrees (the top tree) get owner -1.
void assign_owners(const Symbolic& S, int nranks, std::vector<int>& owner);

}  // namespace spx
