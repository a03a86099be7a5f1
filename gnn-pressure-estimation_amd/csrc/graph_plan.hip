// K0: graph plan built once per topology on the host (stable counting sorts; microseconds for a C-Town batch).
// Replaces PyG's per-call remove_self_loops / add_self_loops (torch_geometric.utils.loop, invoked inside every
// GATConv.forward: reference call sites GraphModels.py:464-465) and the implicit scatter order of
// GATConv / SimpleConv aggregation (GraphModels.py:464-466).
//
// Edge order matters for fp32 reproducibility: inside a destination row the edges keep PyG's order (original
// edges in edge_index order, the appended self loop last), which is the order index_add_/scatter visits them.
#include <vector>

#include "gatres_common.h"

namespace {

// stable counting sort of `n` items by key[i] in [0, N): perm[pos] = item, ptr[N+1]
static void counting_sort(const std::vector<int32_t>& key, int64_t N, int32_t* ptr, std::vector<int32_t>& perm) {
  const int64_t n = (int64_t)key.size();
  for (int64_t i = 0; i <= N; ++i) ptr[i] = 0;
  for (int64_t e = 0; e < n; ++e) ptr[key[e] + 1]++;
  for (int64_t i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> cur(ptr, ptr + N);
  perm.resize(n);
  for (int64_t e = 0; e < n; ++e) perm[cur[key[e]]++] = (int32_t)e;
}

}  // namespace

extern "C" int gatres_graph_count_host(const int64_t* ei, int64_t E, int64_t N, int64_t* num_edges_gat_out) {
  if ((!ei && E > 0) || !num_edges_gat_out || E < 0 || N <= 0) return GATRES_E_BADARG;
  int64_t keep = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) return GATRES_E_GRAPH;
    keep += (s != d);
  }
  if (keep + N > INT32_MAX || E > INT32_MAX) return GATRES_E_UNSUPPORTED;
  *num_edges_gat_out = keep + N;
  return 0;
}

extern "C" int gatres_graph_build_host(const int64_t* ei, int64_t E, int64_t N, int32_t* rowptr, int32_t* col,
                                       int32_t* t_rowptr, int32_t* t_eid, int32_t* t_dst, int32_t* m_rowptr,
                                       int32_t* m_col, int32_t* mt_rowptr, int32_t* mt_dst) {
  if ((!ei && E > 0) || !rowptr || !col || !t_rowptr || !t_eid || !t_dst || !m_rowptr || !m_col || !mt_rowptr || !mt_dst)
    return GATRES_E_BADARG;
  int64_t Eg = 0;
  const int rc = gatres_graph_count_host(ei, E, N, &Eg);
  if (rc) return rc;

  // GATConv edge list in PyG order: non-self-loop edges, then (i, i) for every node
  std::vector<int32_t> src(Eg), dst(Eg);
  int64_t k = 0;
  for (int64_t e = 0; e < E; ++e)
    if (ei[e] != ei[E + e]) { src[k] = (int32_t)ei[e]; dst[k] = (int32_t)ei[E + e]; ++k; }
  for (int64_t i = 0; i < N; ++i, ++k) { src[k] = (int32_t)i; dst[k] = (int32_t)i; }

  std::vector<int32_t> perm, pos(Eg);
  counting_sort(dst, N, rowptr, perm);                 // destination-major
  for (int64_t p = 0; p < Eg; ++p) { col[p] = src[perm[p]]; pos[perm[p]] = (int32_t)p; }
  counting_sort(src, N, t_rowptr, perm);               // source-major, PyG edge order inside a row
  for (int64_t p = 0; p < Eg; ++p) { t_eid[p] = pos[perm[p]]; t_dst[p] = dst[perm[p]]; }

  // SimpleConv: the original edge list untouched
  std::vector<int32_t> ms(E), md(E);
  for (int64_t e = 0; e < E; ++e) { ms[e] = (int32_t)ei[e]; md[e] = (int32_t)ei[E + e]; }
  counting_sort(md, N, m_rowptr, perm);
  for (int64_t p = 0; p < E; ++p) m_col[p] = ms[perm[p]];
  counting_sort(ms, N, mt_rowptr, perm);
  for (int64_t p = 0; p < E; ++p) mt_dst[p] = md[perm[p]];
  return 0;
}
