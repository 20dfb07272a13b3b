// Host-side validity projectors of the reverse loop, C++ behind the C ABI of
// include/scgg.h.  Follows /root/reference/src/ConStruct/projector/projector_3d_utils.py:
//   CoWProjector.project :256-377 + check_validity_and_fix_if_needed :395-428 + is_line_graph_valid :431-461
//   CoWVanillaProjector.project :499-590, ATMTreeProjector.project :697-825 (+ :847-879, has_no_cycles :150-154),
//   ATMVanillaTreeProjector.project :952-1043, AbstractProjector.project :57-147 with TreeProjector :157-164.
// No GPU code here: the projector is host work in the reference too (networkx); the device side of the step is the
// compaction of the new edges and the scatter of the edits (reverse_step.hip: new_edges_* / apply_edits kernels).
//
// The reference draws one np.random.shuffle per graph that falls back to one-by-one insertion, from numpy's GLOBAL
// stream.  That stream lives in the Python process, so a step is two calls: scgg_projector_begin does everything up
// to the shuffles and reports which graphs need one (in graph order, with their edge counts); the caller draws the
// permutations from np.random in that order; scgg_projector_finish does the one-by-one insertions.  Graphs are
// independent of each other, so the split does not change any result.
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/scgg.h"
#include "kernels.h"

using scgg::fail;

namespace {

constexpr int MAX_CLS = 32;          // classes 0..31 fit the per-node bit set
constexpr int MAX_TOPK = 16;

const int COW_TABLE[13][13] = {      // projector_3d_utils.py:193-205
    {0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0},
    {0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0}, {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 0},
    {0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0}, {0, 1, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0}, {0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1}, {0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1, 1}, {0, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0, 1},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0}};
const int ATM_TABLE[4][4] = {{0, 1, 0, 0}, {1, 0, 1, 0}, {0, 1, 0, 1}, {0, 0, 1, 0}};   // :645-648

// One sample's graph: lab[u*n+v] = class of edge {u,v} (0 = none; the unlabelled tree projector stores 1),
// cnt[u*MAX_CLS+c] = edges of class c at node u, present[u] = bit set of the classes with cnt > 0.
struct Graph {
  int n = 0, n_edges = 0, n_bad = 0;
  const uint32_t* allowed = nullptr;   // class adjacency bit sets (nullptr: no class rule)
  std::vector<int8_t> lab;
  std::vector<uint16_t> cnt;
  std::vector<uint32_t> present;
  std::vector<uint8_t> bad;            // bad[u]: two classes present at node u that the table does not join

  void init(int n_, const uint32_t* allowed_) {
    n = n_;
    n_edges = n_bad = 0;
    allowed = allowed_;
    lab.assign((size_t)n * n, 0);
    cnt.assign((size_t)n * MAX_CLS, 0);
    present.assign(n, 0);
    bad.assign(n, 0);
  }
  void bump(int u, int c, int d) {
    uint16_t& k = cnt[(size_t)u * MAX_CLS + c];
    k = (uint16_t)(k + d);
    if (k) present[u] |= 1u << c; else present[u] &= ~(1u << c);
  }
  void refresh(int u) {                // the class rule is local to a node: only the two end points can change
    if (!allowed) return;
    const uint32_t p = present[u];
    uint8_t b = 0;
    for (uint32_t q = p; q; q &= q - 1)
      if (p & ~allowed[__builtin_ctz(q)]) { b = 1; break; }
    n_bad += (int)b - (int)bad[u];
    bad[u] = b;
  }
  void add(int u, int v, int c) {          // nx.Graph.add_edge: an existing edge takes the new label
    int8_t old = lab[(size_t)u * n + v];
    if (old) { bump(u, old, -1); bump(v, old, -1); } else ++n_edges;
    lab[(size_t)u * n + v] = lab[(size_t)v * n + u] = (int8_t)c;
    bump(u, c, 1);
    bump(v, c, 1);
    refresh(u);
    refresh(v);
  }
  void remove(int u, int v) {
    int8_t old = lab[(size_t)u * n + v];
    if (!old) return;
    bump(u, old, -1);
    bump(v, old, -1);
    lab[(size_t)u * n + v] = lab[(size_t)v * n + u] = 0;
    --n_edges;
    refresh(u);
    refresh(v);
  }
};

// is_line_graph_valid: two edges are joined in the line graph iff they share a node, so the class pairs it marks are
// the pairs of classes present at a common node; invalid iff two different classes at one node are not adjacent in
// the table (the diagonal is cleared, :457).
// Kept per node and updated by add / remove (Graph::refresh), so the test itself is one comparison.
bool line_valid(const Graph& g) { return g.n_bad == 0; }

bool no_cycles(const Graph& g, std::vector<int>& parent) {      // has_no_cycles :150-154
  if (g.n_edges >= g.n) return false;
  parent.resize(g.n);
  for (int i = 0; i < g.n; ++i) parent[i] = i;
  auto find = [&](int x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  };
  for (int u = 0; u < g.n; ++u)
    for (int v = u + 1; v < g.n; ++v)
      if (g.lab[(size_t)u * g.n + v]) {
        int a = find(u), b = find(v);
        if (a == b) return false;
        parent[a] = b;
      }
  return true;
}

struct NewEdge {
  int u, v, label;
  int8_t cands[MAX_TOPK];
};

// A graph that falls back to one-by-one insertion.  Its stored graph is back in the state before the step (a new
// edge sits on a pair that had no edge, so the reference's "nx_graph = old_nx_graph" is undone by removing the
// step's edges again: no copies of the graph are kept).
struct Fallback {
  int b;
  std::vector<NewEdge> edges;      // the unblocked new edges of graph b, in row-major order
};

struct Edit { int b, u, v, c; };

}  // namespace

struct scgg_projector {
  int kind = 0, B = 0, n_cls = 0, top_k = 0;
  bool fixing = false, forest = false, table = false, counted = false;
  uint32_t allowed[MAX_CLS] = {};
  long long counter = 0;
  std::vector<Graph> graphs;
  std::vector<std::vector<uint8_t>> blocked;
  std::vector<Fallback> pending;
  std::vector<Edit> edits;
  std::vector<int> uf;

  bool valid_fn(const Graph& g) {          // the vanilla / tree validity functions (:593, :1046, :158)
    if (table && !line_valid(g)) return false;
    return forest ? no_cycles(g, uf) : true;
  }

  // check_validity_and_fix_if_needed (:395-428, :847-879) [+ has_no_cycles for ATM, :830]
  bool try_fix(Graph& g, const NewEdge* e, int m) {
    bool all_ok = true;
    for (int k = 0; k < m; ++k) {
      g.add(e[k].u, e[k].v, e[k].label);
      bool ok = true;
      if (!line_valid(g)) {
        ok = false;
        if (counted) --counter;
        for (int c = 0; c < top_k; ++c) {
          if (counted) ++counter;
          int cand = e[k].cands[c];
          if (cand == 0) continue;         // class 0 would delete the edge
          g.remove(e[k].u, e[k].v);
          g.add(e[k].u, e[k].v, cand);
          if (line_valid(g)) { ok = true; break; }
        }
      }
      all_ok = all_ok && ok;
    }
    if (!all_ok) return false;
    return forest ? no_cycles(g, uf) : true;
  }
};

namespace {

int setup_kind(scgg_projector* p, int kind) {
  p->kind = kind;
  switch (kind) {
    case SCGG_PROJ_COW: p->fixing = true; p->table = true; p->counted = true; p->n_cls = 14; break;
    case SCGG_PROJ_COW_VANILLA: p->table = true; p->n_cls = 14; break;
    case SCGG_PROJ_ATM: p->fixing = true; p->table = true; p->forest = true; p->n_cls = 5; break;
    case SCGG_PROJ_ATM_VANILLA: p->table = true; p->forest = true; p->n_cls = 5; break;
    case SCGG_PROJ_TREE: p->forest = true; p->n_cls = 2; break;
    default: return fail("scgg_projector_create: unknown kind %d", kind);
  }
  if (p->table) {
    const int T = p->n_cls - 1;
    for (int a = 0; a < T; ++a) {
      uint32_t m = 1u << (a + 1);
      for (int b = 0; b < T; ++b) {
        int v = T == 13 ? COW_TABLE[a][b] : ATM_TABLE[a][b];
        int w = T == 13 ? COW_TABLE[b][a] : ATM_TABLE[b][a];
        if (v != w) return fail("validity table is not symmetric");      // check_symmetric :207
        if (v) m |= 1u << (b + 1);
      }
      p->allowed[a + 1] = m;
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int scgg_projector_create(int32_t kind, int32_t B, const int32_t* n_nodes, scgg_projector** out) {
  if (!out || !n_nodes || B < 0) return fail("scgg_projector_create: bad arguments");
  auto* p = new scgg_projector();
  if (setup_kind(p, kind)) { delete p; return -1; }
  p->B = B;
  p->graphs.resize(B);
  p->blocked.resize(B);
  for (int b = 0; b < B; ++b) {
    if (n_nodes[b] < 0 || n_nodes[b] > 32767) { delete p; return fail("scgg_projector_create: n_nodes[%d]=%d", b, n_nodes[b]); }
    if (p->forest && n_nodes[b] == 0) {      // the reference's constructor assertion (nx.is_forest on no nodes raises)
      delete p;
      return fail("scgg_projector_create: graph %d has no nodes (the forest check is undefined on it)", b);
    }
    p->graphs[b].init(n_nodes[b], p->table ? p->allowed : nullptr);
    p->blocked[b].assign((size_t)n_nodes[b] * n_nodes[b], 0);
  }
  *out = p;
  return 0;
}

void scgg_projector_destroy(scgg_projector* p) { delete p; }

int scgg_projector_begin(scgg_projector* p, int64_t m, const int32_t* rec, int32_t top_k, int32_t* fb_graph,
                         int32_t* fb_count, int32_t* n_fb) {
  if (!p || m < 0 || (m && !rec) || !fb_graph || !fb_count || !n_fb) return fail("scgg_projector_begin: bad arguments");
  if (top_k < 0 || top_k > MAX_TOPK) return fail("scgg_projector_begin: top_k=%d (max %d)", top_k, MAX_TOPK);
  if (p->fixing && top_k == 0) return fail("scgg_projector_begin: this projector repairs labels and needs the top-k candidates");
  if (!p->pending.empty()) return fail("scgg_projector_begin: the previous step was not finished");
  p->top_k = top_k;
  p->edits.clear();
  const int stride = 4 + top_k;
  *n_fb = 0;
  // validate the whole batch first: nothing is modified when a record is bad
  int prev_b = -1, prev_u = -1, prev_v = -1;
  for (int64_t k = 0; k < m; ++k) {
    const int32_t* r = rec + k * stride;
    int b = r[0], u = r[1], v = r[2], c = r[3];
    if (b < 0 || b >= p->B) return fail("scgg_projector_begin: record %lld: graph %d of %d", (long long)k, b, p->B);
    int n = p->graphs[b].n;
    if (u < 0 || v <= u || v >= n) return fail("scgg_projector_begin: record %lld: pair (%d,%d) in a graph of %d nodes", (long long)k, u, v, n);
    if (b < prev_b || (b == prev_b && (u < prev_u || (u == prev_u && v <= prev_v))))
      return fail("scgg_projector_begin: records must be sorted by (graph, i, j) without repeats");
    prev_b = b; prev_u = u; prev_v = v;
    // a class change on a stored edge (possible only outside the absorbing process; the reference would relabel it)
    // is refused: the undo below relies on new edges sitting on empty pairs
    if (p->graphs[b].lab[(size_t)u * n + v])
      return fail("scgg_projector_begin: record %lld: pair (%d,%d) of graph %d already holds an edge", (long long)k, u, v, b);
    // the reference indexes its table with class-1 and fails on a class outside it
    if (p->table && (c < 1 || c >= p->n_cls)) return fail("edge class %d outside the validity table (1..%d)", c, p->n_cls - 1);
    if (!p->table && c < 1) return fail("scgg_projector_begin: record %lld: class %d is not an edge", (long long)k, c);
    for (int t = 0; t < top_k; ++t)
      if (r[4 + t] < 0 || r[4 + t] >= (p->table ? p->n_cls : MAX_CLS))
        return fail("candidate class %d outside the validity table (0..%d)", r[4 + t], p->n_cls - 1);
  }
  std::vector<NewEdge> keep;
  for (int64_t lo = 0; lo < m;) {
    const int b = rec[lo * stride];
    int64_t hi = lo;
    while (hi < m && rec[hi * stride] == b) ++hi;
    Graph& g = p->graphs[b];
    std::vector<uint8_t>& blocked = p->blocked[b];
    keep.clear();
    for (int64_t k = lo; k < hi; ++k) {      // :282-301: an edge blocked earlier is deleted again at once
      const int32_t* r = rec + k * stride;
      if (blocked[(size_t)r[1] * g.n + r[2]]) {
        p->edits.push_back({b, r[1], r[2], 0});
      } else {
        NewEdge e{r[1], r[2], p->table ? r[3] : 1, {}};
        for (int t = 0; t < top_k; ++t) e.cands[t] = (int8_t)r[4 + t];
        keep.push_back(e);
      }
    }
    lo = hi;
    if (keep.empty()) continue;
    const int mk = (int)keep.size();
    bool fallback = true;                    // one edge: straight to one by one, :327 / :552 / :106 (graph untouched)
    if (mk > 1) {
      if (p->fixing) {
        if (p->try_fix(g, keep.data(), mk))  // :318-324: labels may have been repaired
          for (const NewEdge& e : keep) p->edits.push_back({b, e.u, e.v, g.lab[(size_t)e.u * g.n + e.v]});
        fallback = !p->try_fix(g, keep.data(), mk);      // :327 evaluates the attempt a second time
      } else {
        for (const NewEdge& e : keep) g.add(e.u, e.v, e.label);
        fallback = !p->valid_fn(g);
      }
      if (fallback)
        for (const NewEdge& e : keep) g.remove(e.u, e.v);      // nx_graph = old_nx_graph
    }
    if (fallback) p->pending.push_back({b, keep});
    if (fallback) {
      fb_graph[*n_fb] = b;
      fb_count[*n_fb] = mk;
      ++*n_fb;
    }
  }
  return 0;
}

int scgg_projector_finish(scgg_projector* p, const int32_t* perm, int64_t n_perm, int32_t* edits_out, int64_t edits_cap,
                          int64_t* n_edits) {
  if (!p || !n_edits) return fail("scgg_projector_finish: bad arguments");
  int64_t need = 0;
  for (const Fallback& f : p->pending) need += (int64_t)f.edges.size();
  if (need != n_perm || (need && !perm)) return fail("scgg_projector_finish: %lld permutation entries, expected %lld", (long long)n_perm, (long long)need);
  int64_t off = 0;
  for (const Fallback& f : p->pending) {     // validate before touching anything
    const int mk = (int)f.edges.size();
    std::vector<uint8_t> seen(mk, 0);
    for (int k = 0; k < mk; ++k) {
      int q = perm[off + k];
      if (q < 0 || q >= mk || seen[q]) return fail("scgg_projector_finish: not a permutation for graph %d", f.b);
      seen[q] = 1;
    }
    off += mk;
  }
  off = 0;
  for (Fallback& f : p->pending) {
    const int b = f.b, mk = (int)f.edges.size();
    Graph& g = p->graphs[b];
    std::vector<uint8_t>& blocked = p->blocked[b];
    for (int k = 0; k < mk; ++k) {
      // np.random.shuffle(edges_to_add) moved the edges only: position k holds edge perm[k] but still the k-th
      // label and candidates (:330-331, :555-556)
      NewEdge e = f.edges[k];
      e.u = f.edges[perm[off + k]].u;
      e.v = f.edges[perm[off + k]].v;
      bool ok;
      if (p->fixing) {
        ok = p->try_fix(g, &e, 1);
      } else {
        g.add(e.u, e.v, e.label);
        ok = p->valid_fn(g);
      }
      if (!ok) {
        g.remove(e.u, e.v);                              // nx_graph = old_nx_graph
        p->edits.push_back({b, e.u, e.v, 0});
        blocked[(size_t)e.u * g.n + e.v] = 1;            // :349 blocked for the rest of the chain
      } else if (p->fixing) {
        p->edits.push_back({b, e.u, e.v, g.lab[(size_t)e.u * g.n + e.v]});      // :353-355
      }
    }
    off += mk;
  }
  p->pending.clear();
  // last write per pair wins (the reference writes z_s.E in this order)
  std::unordered_map<uint64_t, size_t> last;
  last.reserve(p->edits.size() * 2);
  for (size_t k = 0; k < p->edits.size(); ++k) {
    const Edit& e = p->edits[k];
    last[((uint64_t)e.b << 32) | ((uint64_t)e.u << 16) | (uint64_t)e.v] = k;
  }
  int64_t ne = 0;
  for (size_t k = 0; k < p->edits.size(); ++k) {
    const Edit& e = p->edits[k];
    if (last[((uint64_t)e.b << 32) | ((uint64_t)e.u << 16) | (uint64_t)e.v] != k) continue;
    if (ne >= edits_cap || !edits_out) { p->edits.clear(); return fail("scgg_projector_finish: more than %lld edits", (long long)edits_cap); }
    int32_t* o = edits_out + ne * 4;
    o[0] = e.b; o[1] = e.u; o[2] = e.v; o[3] = e.c;
    ++ne;
  }
  p->edits.clear();
  *n_edits = ne;
  return 0;
}

int scgg_projector_export(const scgg_projector* p, int32_t b, int8_t* graph_labels, uint8_t* blocked) {
  if (!p || b < 0 || b >= p->B) return fail("scgg_projector_export: graph %d of %d", b, p ? p->B : 0);
  const Graph& g = p->graphs[b];
  if (graph_labels) std::memcpy(graph_labels, g.lab.data(), g.lab.size());
  if (blocked) std::memcpy(blocked, p->blocked[b].data(), p->blocked[b].size());
  return 0;
}

int64_t scgg_projector_counter(const scgg_projector* p) { return p ? p->counter : 0; }

}  // extern "C"
