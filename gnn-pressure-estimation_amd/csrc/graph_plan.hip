// K0: graph plan built once per topology on the host (stable counting sorts; microseconds for a C-Town batch).
// Replaces PyG's per-call remove_self_loops / add_self_loops (torch_geometric.utils.loop, invoked inside every
// GATConv.forward: reference call sites GraphModels.py:464-465) and the implicit scatter order of
// GATConv / SimpleConv aggregation (GraphModels.py:464-466).
//
// Edge order matters for fp32 reproducibility: inside a destination row the edges keep PyG's order (original
// edges in edge_index order, the appended self loop last), which is the order index_add_/scatter visits them.
#include <algorithm>
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

// GATRES_GRAPH_SYMMETRIC: the set of non-loop edges equals its reverse (multiplicities do not matter: a part needs a
// partner's row if ANY edge joins them in that direction).
extern "C" int gatres_graph_flags_host(const int64_t* ei, int64_t E, int64_t N, int32_t* flags_out) {
  if ((!ei && E > 0) || !flags_out || E < 0 || N <= 0) return GATRES_E_BADARG;
  std::vector<uint64_t> key;
  key.reserve((size_t)E);
  std::vector<int32_t> indeg((size_t)N, 0), outdeg((size_t)N, 0);
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) return GATRES_E_GRAPH;
    if (s != d) key.push_back(((uint64_t)s << 32) | (uint64_t)d);
    ++indeg[(size_t)d]; ++outdeg[(size_t)s];
  }
  // every row of every CSR of the plan (GATConv: the edges without self loops + one; SimpleConv: all of them; and the
  // transposes) has at most 32 entries: the blocked kernels' rows are walked by their own lane group (k_blocked.hip)
  int32_t most = 0;
  for (int64_t i = 0; i < N; ++i) most = std::max(most, std::max(indeg[(size_t)i], outdeg[(size_t)i]));
  const bool le32 = most + 1 <= 32, le6 = most + 1 <= 6;
  std::sort(key.begin(), key.end());
  bool sym = true;
  for (size_t i = 0; i < key.size() && sym; ++i)
    sym = std::binary_search(key.begin(), key.end(), (key[i] << 32) | (key[i] >> 32));
  *flags_out = (sym ? GATRES_GRAPH_SYMMETRIC : 0) | (le32 ? GATRES_GRAPH_DEG_LE32 : 0) | (le6 ? GATRES_GRAPH_DEG_LE6 : 0);
  return 0;
}

// Segments: cut after node i iff no edge joins a node <= i with a node > i.  A PyG Batch yields one segment per
// snapshot (or finer, if a snapshot is disconnected); neighbours smaller than `merge_upto` are then coalesced so a
// workgroup never gets a degenerate handful of nodes.
extern "C" int gatres_graph_segments_host(const int64_t* ei, int64_t E, int64_t N, int32_t merge_upto,
                                          int32_t* seg_ptr, int32_t* num_segments_out, int32_t* max_nodes_out,
                                          int32_t* max_edges_gat_out, int32_t* max_edges_mean_out) {
  if ((!ei && E > 0) || !seg_ptr || !num_segments_out || !max_nodes_out || !max_edges_gat_out || !max_edges_mean_out ||
      E < 0 || N <= 0)
    return GATRES_E_BADARG;
  if (N > INT32_MAX) return GATRES_E_UNSUPPORTED;
  std::vector<int32_t> reach(N);
  for (int64_t i = 0; i < N; ++i) reach[i] = (int32_t)i;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) return GATRES_E_GRAPH;
    const int64_t lo = s < d ? s : d, hi = s < d ? d : s;
    if (reach[lo] < hi) reach[lo] = (int32_t)hi;
  }
  std::vector<int32_t> cuts;                       // fine segment ends (exclusive)
  int32_t far = 0;
  for (int64_t i = 0; i < N; ++i) {
    if (reach[i] > far) far = reach[i];
    if (far <= i) cuts.push_back((int32_t)(i + 1));
  }
  int32_t ns = 0, start = 0, cur_end = 0, mx = 0;
  seg_ptr[0] = 0;
  for (size_t c = 0; c < cuts.size(); ++c) {
    const int32_t end = cuts[c];
    if (cur_end > start && end - start > merge_upto) {      // close the running segment before this piece
      seg_ptr[++ns] = cur_end;
      if (cur_end - start > mx) mx = cur_end - start;
      start = cur_end;
    }
    cur_end = end;
  }
  seg_ptr[++ns] = cur_end;
  if (cur_end - start > mx) mx = cur_end - start;
  // per-segment edge counts (every edge lies inside one segment, so count by destination)
  std::vector<int64_t> in_all(N + 1, 0), in_gat(N + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    in_all[ei[E + e] + 1]++;
    in_gat[ei[E + e] + 1] += (ei[e] != ei[E + e]);
  }
  for (int64_t i = 0; i < N; ++i) { in_all[i + 1] += in_all[i]; in_gat[i + 1] += in_gat[i] + 1; }   // +1: self loop
  int64_t mg = 0, mm = 0;
  for (int32_t sgi = 0; sgi < ns; ++sgi) {
    const int64_t a = seg_ptr[sgi], b = seg_ptr[sgi + 1];
    if (in_gat[b] - in_gat[a] > mg) mg = in_gat[b] - in_gat[a];
    if (in_all[b] - in_all[a] > mm) mm = in_all[b] - in_all[a];
  }
  *num_segments_out = ns;
  *max_nodes_out = mx;
  *max_edges_gat_out = (int32_t)(mg > INT32_MAX ? INT32_MAX : mg);
  *max_edges_mean_out = (int32_t)(mm > INT32_MAX ? INT32_MAX : mm);
  return 0;
}

// Row windows of split segments.  When a segment is carried by M workgroups, part p owns the 16-aligned row range
// [16*(tiles*p/M), 16*(tiles*(p+1)/M)) and only ever touches the rows its own rows are adjacent to; the smallest
// contiguous row range holding both is the part's WINDOW.  With a locality-preserving node order the window is a
// fraction of the segment, and the kernels can size their LDS tables by it.  For M = 2, 4, 8 (out[3*k + ...]):
// the largest window in rows, in GATConv edges (in-edges of the window's rows, self loops included) and in
// SimpleConv edges, over all segments and parts; the same for every part count M = 2 .. 8.
extern "C" int gatres_graph_windows_host(const int64_t* ei, int64_t E, int64_t N, const int32_t* seg_ptr,
                                         int32_t num_segments, int32_t* out9) {
  // (out9 holds 28 entries: three window figures for each M = 2 .. 8, then the seven halo sizes)
  if ((!ei && E > 0) || !seg_ptr || !out9 || num_segments <= 0 || N <= 0 || E < 0) return GATRES_E_BADARG;
  std::vector<int32_t> seg_of(N);
  for (int32_t s = 0; s < num_segments; ++s)
    for (int32_t i = seg_ptr[s]; i < seg_ptr[s + 1]; ++i) seg_of[i] = s;
  std::vector<int64_t> in_all(N + 1, 0), in_gat(N + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    in_all[ei[E + e] + 1]++;
    in_gat[ei[E + e] + 1] += (ei[e] != ei[E + e]);
  }
  for (int64_t i = 0; i < N; ++i) { in_all[i + 1] += in_all[i]; in_gat[i + 1] += in_gat[i] + 1; }
  for (int k = 0; k < 7; ++k) {
    const int M = k + 2;
    std::vector<int32_t> wlo((size_t)num_segments * M), whi((size_t)num_segments * M);
    std::vector<int32_t> hin((size_t)num_segments * M, 0), hout((size_t)num_segments * M, 0);   // remote in- / out-edges
    auto bound = [&](int32_t s, int p) {
      const int32_t n = seg_ptr[s + 1] - seg_ptr[s], tiles = (n + 15) / 16;
      const int64_t b = 16 * ((int64_t)tiles * p / M);
      return (int32_t)(b < n ? b : n);
    };
    auto part_of = [&](int32_t s, int32_t r) {        // r: row inside segment s
      int p = 0;
      while (p + 1 < M && bound(s, p + 1) <= r) ++p;
      return p;
    };
    for (int32_t s = 0; s < num_segments; ++s)
      for (int p = 0; p < M; ++p) { wlo[(size_t)s * M + p] = bound(s, p); whi[(size_t)s * M + p] = bound(s, p + 1); }
    for (int64_t e = 0; e < E; ++e) {
      const int64_t a = ei[e], b = ei[E + e];
      if (a == b) continue;
      const int32_t s = seg_of[a];
      if (seg_of[b] != s) return GATRES_E_GRAPH;
      const int32_t ra = (int32_t)(a - seg_ptr[s]), rb = (int32_t)(b - seg_ptr[s]);
      const int pa = part_of(s, ra), pb = part_of(s, rb);
      if (pa == pb) continue;
      hout[(size_t)s * M + pa]++;                   // edge a -> b: an out-edge of a's part, an in-edge of b's part
      hin[(size_t)s * M + pb]++;
      int32_t& la = wlo[(size_t)s * M + pa]; int32_t& ha = whi[(size_t)s * M + pa];
      int32_t& lb = wlo[(size_t)s * M + pb]; int32_t& hb = whi[(size_t)s * M + pb];
      if (rb < la) la = rb;
      if (rb + 1 > ha) ha = rb + 1;
      if (ra < lb) lb = ra;
      if (ra + 1 > hb) hb = ra + 1;
    }
    int64_t mr = 0, mg = 0, mm = 0;
    for (int32_t s = 0; s < num_segments; ++s)
      for (int p = 0; p < M; ++p) {
        const int64_t lo = seg_ptr[s] + wlo[(size_t)s * M + p], hi = seg_ptr[s] + whi[(size_t)s * M + p];
        if (hi - lo > mr) mr = hi - lo;
        if (in_gat[hi] - in_gat[lo] > mg) mg = in_gat[hi] - in_gat[lo];
        if (in_all[hi] - in_all[lo] > mm) mm = in_all[hi] - in_all[lo];
      }
    out9[3 * k + 0] = (int32_t)mr;
    out9[3 * k + 1] = (int32_t)(mg > INT32_MAX ? INT32_MAX : mg);
    out9[3 * k + 2] = (int32_t)(mm > INT32_MAX ? INT32_MAX : mm);
    int32_t mh = 0;
    for (size_t i = 0; i < hin.size(); ++i) mh = std::max(mh, std::max(hin[i], hout[i]));
    out9[21 + k] = mh;
  }
  return 0;
}

// Part tables of split segments (gatres_graph_t.part_tables): everything the window kernel's two prologues used to derive
// from the CSR arrays in every launch -- the part's row window, its edge ranges, 16-bit local copies of its CSR slices, the
// padded edge descriptors of its rows (k_window_stages.h) and its hand-off lists -- built ONCE per (plan, M) on the host
// and copied into LDS by the kernel (the topology of a WDN batch never changes: train.py:302-303 collates the same
// edge_index every iteration).  Hand-off lists are de-duplicated here (the in-kernel builder kept one entry per edge).
// Layout of one part (int32 words; u16 tables are padded to whole words):
//   [0 .. GATRES_PT_HEADER)   header, see the enum below
//   forward image             rpo | colo | mrpo | mcolo | pad to 16 B | nbin | mbin            (the LDS image, verbatim)
//   backward image            rpo | colo | trpo | teido | tdsto | mrpw | mtrpo | mtdsto | pad | nbin | tout | mout
//   lists                     f_hlist, f_elist, b_hrow, b_hedge, b_erow, b_eedge
namespace {
inline int even_i(int v) { return (v + 1) & ~1; }
struct PartGeom {
  int n0, n, e0, em0, t0, mt0, lo, hi, wlo, whi, elo, oeg, ewlo, weg, melo, oem, tlo, otg, mtlo, otm;
};
constexpr int PT_MAXD = 6;      // k_fused_dev.h: MAXD
}  // namespace

extern "C" int gatres_graph_part_tables_host(const int32_t* rowptr, const int32_t* col, const int32_t* t_rowptr,
                                             const int32_t* t_eid, const int32_t* t_dst, const int32_t* m_rowptr,
                                             const int32_t* m_col, const int32_t* mt_rowptr, const int32_t* mt_dst,
                                             const int32_t* seg_ptr, int32_t num_segments, int32_t M, int32_t* out,
                                             int64_t stride_words, int64_t* stride_words_out) {
  if (!rowptr || !col || !t_rowptr || !t_eid || !t_dst || !m_rowptr || !m_col || !mt_rowptr || !mt_dst || !seg_ptr ||
      num_segments <= 0 || M < 2 || M > 8 || !stride_words_out)
    return GATRES_E_BADARG;
  int64_t need = 0;
  std::vector<uint16_t> img;
  std::vector<int32_t> words;
  for (int32_t s = 0; s < num_segments; ++s) {
    PartGeom g;
    g.n0 = seg_ptr[s]; g.n = seg_ptr[s + 1] - g.n0;
    g.e0 = rowptr[g.n0]; g.em0 = m_rowptr[g.n0]; g.t0 = t_rowptr[g.n0]; g.mt0 = mt_rowptr[g.n0];
    if (g.n > 65535 || rowptr[g.n0 + g.n] - g.e0 > 65535 || m_rowptr[g.n0 + g.n] - g.em0 > 65535) return GATRES_E_UNSUPPORTED;
    const int tiles = (g.n + 15) >> 4;
    for (int p = 0; p < M; ++p) {
      g.lo = 16 * (int)((long long)tiles * p / M);
      g.hi = std::min(g.n, 16 * (int)((long long)tiles * (p + 1) / M));
      if (g.hi < g.lo) g.hi = g.lo;
      const int ow = g.hi - g.lo;
      int vmin = g.lo, vmax = g.hi;
      auto widen = [&](const int32_t* ptr, const int32_t* idx) {
        for (int e = ptr[g.n0 + g.lo]; e < ptr[g.n0 + g.hi]; ++e) {
          const int j = idx[e] - g.n0;
          vmin = std::min(vmin, j); vmax = std::max(vmax, j + 1);
        }
      };
      widen(rowptr, col); widen(t_rowptr, t_dst); widen(m_rowptr, m_col); widen(mt_rowptr, mt_dst);
      g.wlo = vmin; g.whi = vmax;
      const int wr = g.whi - g.wlo;
      g.elo = rowptr[g.n0 + g.lo] - g.e0;           g.oeg = rowptr[g.n0 + g.hi] - g.e0 - g.elo;
      g.ewlo = rowptr[g.n0 + g.wlo] - g.e0;         g.weg = rowptr[g.n0 + g.whi] - g.e0 - g.ewlo;
      g.melo = m_rowptr[g.n0 + g.lo] - g.em0;       g.oem = m_rowptr[g.n0 + g.hi] - g.em0 - g.melo;
      g.tlo = t_rowptr[g.n0 + g.lo] - g.t0;         g.otg = t_rowptr[g.n0 + g.hi] - g.t0 - g.tlo;
      g.mtlo = mt_rowptr[g.n0 + g.lo] - g.mt0;      g.otm = mt_rowptr[g.n0 + g.hi] - g.mt0 - g.mtlo;
      // ---- 16-bit local CSR slices
      std::vector<uint16_t> rpo(ow + 1), colo(g.oeg), mrpo(ow + 1), mcolo(g.oem), trpo(ow + 1), teido(g.otg), tdsto(g.otg),
          mrpw(wr + 1), mtrpo(ow + 1), mtdsto(g.otm);
      for (int r = 0; r <= ow; ++r) {
        rpo[r] = (uint16_t)(rowptr[g.n0 + g.lo + r] - (g.e0 + g.elo));
        mrpo[r] = (uint16_t)(m_rowptr[g.n0 + g.lo + r] - (g.em0 + g.melo));
        trpo[r] = (uint16_t)(t_rowptr[g.n0 + g.lo + r] - (g.t0 + g.tlo));
        mtrpo[r] = (uint16_t)(mt_rowptr[g.n0 + g.lo + r] - (g.mt0 + g.mtlo));
      }
      for (int k = 0; k < g.oeg; ++k) colo[k] = (uint16_t)(col[g.e0 + g.elo + k] - g.n0);
      for (int k = 0; k < g.oem; ++k) mcolo[k] = (uint16_t)(m_col[g.em0 + g.melo + k] - g.n0);
      for (int k = 0; k < g.otg; ++k) {
        teido[k] = (uint16_t)(t_eid[g.t0 + g.tlo + k] - g.e0);
        tdsto[k] = (uint16_t)(t_dst[g.t0 + g.tlo + k] - g.n0);
      }
      for (int r = 0; r <= wr; ++r) mrpw[r] = (uint16_t)(m_rowptr[g.n0 + g.wlo + r] - m_rowptr[g.n0 + g.wlo]);
      for (int k = 0; k < g.otm; ++k) mtdsto[k] = (uint16_t)(mt_dst[g.mt0 + g.mtlo + k] - g.n0);
      // ---- padded edge descriptors (k_window_stages.h: build_nbr_in / build_nbr_out)
      std::vector<uint16_t> nbin(8 * ow), mbin(8 * ow), tout(16 * ow, 0), mout(16 * ow, 0);
      for (int r = 0; r < ow; ++r) {
        const int ra = g.lo + r;
        {
          const int beg = rpo[r], deg = (int)rpo[r + 1] - beg, elast = std::max(g.oeg - 1, 0);
          nbin[8 * r] = (uint16_t)beg; nbin[8 * r + 1] = (uint16_t)deg;
          for (int k = 0; k < PT_MAXD; ++k)
            nbin[8 * r + 2 + k] = g.oeg > 0 ? colo[std::min(beg + std::min(k, std::max(deg - 1, 0)), elast)] : (uint16_t)ra;
        }
        {
          const int beg = mrpo[r], deg = (int)mrpo[r + 1] - beg, elast = std::max(g.oem - 1, 0);
          mbin[8 * r] = (uint16_t)beg; mbin[8 * r + 1] = (uint16_t)deg;
          for (int k = 0; k < PT_MAXD; ++k) mbin[8 * r + 2 + k] = k < deg ? mcolo[std::min(beg + k, elast)] : (uint16_t)ra;
        }
        {
          const int beg = trpo[r], deg = (int)trpo[r + 1] - beg, elast = std::max(g.otg - 1, 0);
          tout[16 * r] = (uint16_t)deg;
          for (int k = 0; k < PT_MAXD; ++k) {
            const int kk = std::min(beg + std::min(k, std::max(deg - 1, 0)), elast);
            tout[16 * r + 2 + k] = g.otg > 0 ? tdsto[kk] : (uint16_t)ra;
            tout[16 * r + 8 + k] = g.otg > 0 ? teido[kk] : (uint16_t)0;
          }
        }
        {
          const int beg = mtrpo[r], deg = (int)mtrpo[r + 1] - beg, elast = std::max(g.otm - 1, 0);
          mout[16 * r] = (uint16_t)deg;
          for (int k = 0; k < PT_MAXD; ++k) {
            const int kk = std::min(beg + std::min(k, std::max(deg - 1, 0)), elast);
            const int ii = k < deg ? (int)mtdsto[kk] : ra;
            mout[16 * r + 2 + k] = (uint16_t)ii;
            mout[16 * r + 8 + k] = (uint16_t)std::max((int)mrpw[ii - g.wlo + 1] - (int)mrpw[ii - g.wlo], 1);
          }
        }
      }
      // ---- hand-off lists (sorted, unique)
      auto remote = [&](int j) { return j < g.lo || j >= g.hi; };
      std::vector<uint16_t> f_hlist, f_elist, b_hrow, b_hedge, b_erow, b_eedge;
      for (int k = 0; k < g.oeg; ++k)
        if (remote(colo[k])) { f_hlist.push_back(colo[k]); b_eedge.push_back((uint16_t)(g.elo + k)); }
      for (int r = 0; r < ow; ++r) {
        bool out_remote = false, in_remote = false;
        for (int t = trpo[r]; t < trpo[r + 1]; ++t) out_remote = out_remote || remote(tdsto[t]);
        for (int e = rpo[r]; e < rpo[r + 1]; ++e) in_remote = in_remote || remote(colo[e]);
        if (out_remote) f_elist.push_back((uint16_t)(g.lo + r));
        if (in_remote) b_erow.push_back((uint16_t)(g.lo + r));
      }
      for (int t = 0; t < g.otg; ++t)
        if (remote(tdsto[t])) { b_hrow.push_back(tdsto[t]); b_hedge.push_back(teido[t]); }
      auto uniq = [](std::vector<uint16_t>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
      uniq(f_hlist); uniq(b_hrow); uniq(b_hedge);
      // ---- assemble
      img.clear();
      auto put = [&](const std::vector<uint16_t>& v, int padded) {
        img.insert(img.end(), v.begin(), v.end());
        img.resize(img.size() + (size_t)(padded - (int)v.size()), 0);
      };
      auto pad16 = [&](size_t from) { while ((img.size() - from) % 8) img.push_back(0); };
      int32_t hdr[GATRES_PT_HEADER] = {0};
      hdr[GATRES_PT_MAGIC] = GATRES_PT_MAGIC_VALUE; hdr[GATRES_PT_M] = M;
      hdr[GATRES_PT_N0] = g.n0; hdr[GATRES_PT_N] = g.n; hdr[GATRES_PT_E0] = g.e0; hdr[GATRES_PT_EM0] = g.em0;
      hdr[GATRES_PT_T0] = g.t0; hdr[GATRES_PT_MT0] = g.mt0; hdr[GATRES_PT_LO] = g.lo; hdr[GATRES_PT_HI] = g.hi;
      hdr[GATRES_PT_WLO] = g.wlo; hdr[GATRES_PT_WHI] = g.whi; hdr[GATRES_PT_ELO] = g.elo; hdr[GATRES_PT_OEG] = g.oeg;
      hdr[GATRES_PT_EWLO] = g.ewlo; hdr[GATRES_PT_WEG] = g.weg; hdr[GATRES_PT_MELO] = g.melo; hdr[GATRES_PT_OEM] = g.oem;
      hdr[GATRES_PT_TLO] = g.tlo; hdr[GATRES_PT_OTG] = g.otg; hdr[GATRES_PT_MTLO] = g.mtlo; hdr[GATRES_PT_OTM] = g.otm;
      size_t from = img.size();
      hdr[GATRES_PT_F_IMG] = GATRES_PT_HEADER + (int)(img.size() / 2);
      put(rpo, even_i(ow + 1)); put(colo, even_i(g.oeg)); put(mrpo, even_i(ow + 1)); put(mcolo, even_i(g.oem));
      pad16(from); put(nbin, 8 * ow); put(mbin, 8 * ow);
      hdr[GATRES_PT_F_IMG_WORDS] = (int)((img.size() - from) / 2);
      from = img.size();
      hdr[GATRES_PT_B_IMG] = GATRES_PT_HEADER + (int)(img.size() / 2);
      put(rpo, even_i(ow + 1)); put(colo, even_i(g.oeg)); put(trpo, even_i(ow + 1)); put(teido, even_i(g.otg));
      put(tdsto, even_i(g.otg)); put(mrpw, even_i(wr + 1)); put(mtrpo, even_i(ow + 1)); put(mtdsto, even_i(g.otm));
      pad16(from); put(nbin, 8 * ow); put(tout, 16 * ow); put(mout, 16 * ow);
      hdr[GATRES_PT_B_IMG_WORDS] = (int)((img.size() - from) / 2);
      auto put_list = [&](const std::vector<uint16_t>& v, int off_field, int cnt_field) {
        hdr[off_field] = GATRES_PT_HEADER + (int)(img.size() / 2);
        hdr[cnt_field] = (int)v.size();
        put(v, even_i((int)v.size()));
      };
      put_list(f_hlist, GATRES_PT_F_HLIST, GATRES_PT_F_HCNT); put_list(f_elist, GATRES_PT_F_ELIST, GATRES_PT_F_ECNT);
      put_list(b_hrow, GATRES_PT_B_HROW, GATRES_PT_B_HRCNT);   put_list(b_hedge, GATRES_PT_B_HEDGE, GATRES_PT_B_HECNT);
      put_list(b_erow, GATRES_PT_B_EROW, GATRES_PT_B_ERCNT);   put_list(b_eedge, GATRES_PT_B_EEDGE, GATRES_PT_B_EECNT);
      const int64_t total_words = GATRES_PT_HEADER + (int64_t)img.size() / 2;
      need = std::max(need, total_words);
      if (out) {
        if (total_words > stride_words) return GATRES_E_BADARG;
        int32_t* dst = out + ((int64_t)s * M + p) * stride_words;
        for (int i = 0; i < GATRES_PT_HEADER; ++i) dst[i] = hdr[i];
        for (size_t i = 0; i + 1 < img.size() + 1 && i < img.size(); i += 2)
          dst[GATRES_PT_HEADER + i / 2] = (int32_t)((uint32_t)img[i] | ((uint32_t)img[i + 1] << 16));
      }
    }
  }
  *stride_words_out = (need + 3) & ~(int64_t)3;      // 16-byte multiples: every part starts 16-byte aligned
  return 0;
}

// Reverse Cuthill-McKee inside every segment.  The fused kernels give part p of a split segment a contiguous row range
// and size its LDS tables by the range of rows those rows are adjacent to (gatres_graph_windows_host); that range is
// part + 2 x bandwidth, so a bandwidth-reducing order is what keeps the window a fraction of the segment.  EPANET files
// list junctions in drawing order, not in a locality-preserving one (utils/DataLoader.py:28-37 keeps the file order).
// Deterministic: ties broken by the caller's node id.
extern "C" int gatres_graph_reorder_host(const int64_t* ei, int64_t E, int64_t N, const int32_t* seg_ptr,
                                         int32_t num_segments, int32_t* perm) {
  if ((!ei && E > 0) || !seg_ptr || !perm || num_segments <= 0 || N <= 0 || E < 0) return GATRES_E_BADARG;
  if (N > INT32_MAX || 2 * E > INT32_MAX) return GATRES_E_UNSUPPORTED;
  // undirected adjacency (self loops dropped; duplicates are harmless)
  std::vector<int32_t> key(2 * E), val(2 * E);
  int64_t k = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) return GATRES_E_GRAPH;
    if (s == d) continue;
    key[k] = (int32_t)s; val[k] = (int32_t)d; ++k;
    key[k] = (int32_t)d; val[k] = (int32_t)s; ++k;
  }
  key.resize(k); val.resize(k);
  std::vector<int32_t> ptr(N + 1), order;
  counting_sort(key, N, ptr.data(), order);
  std::vector<int32_t> adj(k);
  for (int64_t p = 0; p < k; ++p) adj[p] = val[order[p]];
  auto deg = [&](int32_t v) { return ptr[v + 1] - ptr[v]; };

  std::vector<int32_t> level(N, -1), queue;
  std::vector<char> done(N, 0);
  queue.reserve(1024);
  // BFS from `root` over not-yet-numbered nodes; fills `queue` (visit order, neighbours by increasing degree) and
  // returns the eccentricity; `stamp` marks this traversal in level[]
  std::vector<int32_t> nb;
  auto bfs = [&](int32_t root, std::vector<int32_t>& out) {
    out.clear();
    out.push_back(root);
    level[root] = 0;
    int32_t ecc = 0;
    for (size_t head = 0; head < out.size(); ++head) {
      const int32_t v = out[head];
      nb.clear();
      for (int32_t p = ptr[v]; p < ptr[v + 1]; ++p) {
        const int32_t u = adj[p];
        if (!done[u] && level[u] < 0) { level[u] = level[v] + 1; nb.push_back(u); }
      }
      std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return deg(a) != deg(b) ? deg(a) < deg(b) : a < b; });
      for (int32_t u : nb) { out.push_back(u); if (level[u] > ecc) ecc = level[u]; }
    }
    return ecc;
  };
  auto clear_levels = [&](const std::vector<int32_t>& out) { for (int32_t v : out) level[v] = -1; };

  std::vector<int32_t> comp, trial;
  for (int32_t s = 0; s < num_segments; ++s) {
    const int32_t a = seg_ptr[s], b = seg_ptr[s + 1];
    if (a < 0 || b > N || a > b) return GATRES_E_BADARG;
    int32_t next = a;                                 // next new id of this segment
    std::vector<int32_t> seg_order;
    seg_order.reserve(b - a);
    for (int32_t v0 = a; v0 < b; ++v0) {
      if (done[v0]) continue;
      // component of v0: start from its minimum-degree node, then walk to a pseudo-peripheral one
      bfs(v0, comp);
      clear_levels(comp);
      int32_t root = comp[0];
      for (int32_t v : comp) {
        if (v < a || v >= b) return GATRES_E_GRAPH;    // an edge leaves the segment
        if (deg(v) < deg(root) || (deg(v) == deg(root) && v < root)) root = v;
      }
      int32_t ecc = bfs(root, trial);
      for (int it = 0; it < 8; ++it) {
        // candidate: a minimum-degree node of the last level
        int32_t cand = trial.back();
        for (size_t i = trial.size(); i-- > 0 && level[trial[i]] == ecc;)
          if (deg(trial[i]) < deg(cand) || (deg(trial[i]) == deg(cand) && trial[i] < cand)) cand = trial[i];
        clear_levels(trial);
        std::vector<int32_t> t2;
        const int32_t ecc2 = bfs(cand, t2);
        if (ecc2 > ecc) { ecc = ecc2; root = cand; trial.swap(t2); }
        else { clear_levels(t2); bfs(root, trial); break; }
      }
      for (int32_t v : trial) { done[v] = 1; seg_order.push_back(v); }
      clear_levels(trial);
    }
    if ((int32_t)seg_order.size() != b - a) return GATRES_E_GRAPH;
    for (size_t i = seg_order.size(); i-- > 0;) perm[next++] = seg_order[i];      // reversed
  }
  return 0;
}

// LZ4 block decompression (host), for the Blosc-compressed chunks of the reference's zarr stores
// (gnn_pressure_estimation/utils/DataLoader.py:212-242 reads what scenegenv7.py:701-725 writes; zarr's default
// compressor is Blosc(cname='lz4')).  The published LZ4 block format: sequences of [token][literal length ext]
// [literals][offset lo hi][match length ext]; the last sequence ends after its literals.  Returns the number of bytes
// produced, or a negative GATRES_E_* code on malformed input (never reads or writes out of bounds).
extern "C" int64_t gatres_lz4_decompress_host(const uint8_t* src, int64_t src_len, uint8_t* dst, int64_t dst_cap) {
  if (!src || !dst || src_len < 0 || dst_cap < 0) return GATRES_E_BADARG;
  int64_t ip = 0, op = 0;
  while (ip < src_len) {
    const uint8_t token = src[ip++];
    int64_t lit = token >> 4;
    if (lit == 15) {
      uint8_t b;
      do {
        if (ip >= src_len) return GATRES_E_BADARG;
        b = src[ip++];
        lit += b;
      } while (b == 255);
    }
    if (ip + lit > src_len || op + lit > dst_cap) return GATRES_E_BADARG;
    for (int64_t k = 0; k < lit; ++k) dst[op + k] = src[ip + k];
    ip += lit; op += lit;
    if (ip >= src_len) break;                          // last sequence: literals only
    if (ip + 2 > src_len) return GATRES_E_BADARG;
    const int64_t off = src[ip] | ((int64_t)src[ip + 1] << 8);
    ip += 2;
    if (off == 0 || off > op) return GATRES_E_BADARG;
    int64_t ml = (token & 15) + 4;
    if ((token & 15) == 15) {
      uint8_t b;
      do {
        if (ip >= src_len) return GATRES_E_BADARG;
        b = src[ip++];
        ml += b;
      } while (b == 255);
    }
    if (op + ml > dst_cap) return GATRES_E_BADARG;
    for (int64_t k = 0; k < ml; ++k) dst[op + k] = dst[op + k - off];      // byte by byte: matches may overlap themselves
    op += ml;
  }
  return op;
}
