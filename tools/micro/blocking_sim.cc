// blocking_sim.cc -- CPU experiment (no GPU): how many node RECORDS does the order-free set walk fetch per query
// under different ways of packing the k-d tree's inner nodes into records?
//
// The set walk (seed_lookup.hip, kWalkSet) enters a node iff its tight box is within the search radius of the query.
// Boxes are nested, so the nodes a query enters form a connected subtree from the root; a record is fetched for every
// entered node that is the HEAD of a record.  Today's quad records put a head at every even depth (root, grandchildren,
// ...), each with the (up to) four nodes two levels below as slots.  A record can hold ANY fragment of the tree with
// <= K frontier slots (K = 4 for the 128-byte quad record with 16-bit boxes), and which nodes are heads is free to
// choose: this tool evaluates, on real queries against a real index,
//   even    : heads at even depth (the current BuildQuads)
//   odd     : root + heads at odd depth
//   dp-geo  : heads and fragment shapes chosen by a bottom-up DP that minimises sum over heads of w(head), w = volume of
//             the node's box grown by the radius (what an index build can compute without queries)
//   dp-emp  : the same with w = entries counted over a training half of the queries (what is achievable at best)
// for K = 4, 6, 7, 8, on the other half of the queries.  Leaves are scanned identically in every scheme.
//
//   blocking_sim nodes.bin values.bin vind.bin queries.bin radius
//     nodes.bin  : int32 n_nodes, then n_nodes x {int32 feat, uint32 a, uint32 b, uint32 child2}  (pre-order, child1 = self + 1)
//     values.bin : int64 n, float[n]    (point i = values[i .. i + 5])
//     vind.bin   : int64 m, uint32[m]
//     queries.bin: int64 nq, float[nq][6]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Node { int32_t feat; uint32_t a, b, child2; };
static std::vector<Node> nodes;
static std::vector<float> lo, hi;  // [n_nodes][6] tight boxes
static std::vector<float> vals;
static std::vector<uint32_t> vind;
static std::vector<int> depth;

template <class T> static std::vector<T> slurp(const char *path, bool i32_header = false) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(1); }
  int64_t n = 0;
  if (i32_header) { int32_t n32; if (fread(&n32, 4, 1, f) != 1) exit(1); n = n32; }
  else if (fread(&n, 8, 1, f) != 1) exit(1);
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, i32_header ? 4 : 8, SEEK_SET);
  const size_t cnt = (sz - (i32_header ? 4 : 8)) / sizeof(T);
  std::vector<T> v(cnt);
  if (fread(v.data(), sizeof(T), cnt, f) != cnt) exit(1);
  fclose(f);
  (void)n;
  return v;
}

static inline bool is_leaf(uint32_t x) { return nodes[x].feat < 0; }
static inline uint32_t left(uint32_t x) { return x + 1; }
static inline uint32_t right(uint32_t x) { return nodes[x].child2; }

static inline float boxdist(uint32_t x, const float *q) {
  float s = 0;
  for (int d = 0; d < 6; ++d) {
    const float l = lo[(size_t)x * 6 + d], h = hi[(size_t)x * 6 + d];
    const float g = q[d] < l ? l - q[d] : (q[d] > h ? q[d] - h : 0.0f);
    s += g * g;
  }
  return s;
}

// knapsack DP.  C[x] = cost of subtree(x) when x is a head; g(y, k) = cost of subtree(y) when y lies inside a fragment and
// may use up to k frontier slots: g(y, 1) = C[y] (0 for a leaf), g(y, k) = min(g(y, k - 1), min_{k1 + k2 = k} g(L, k1) + g(R, k2)).
struct Plan {
  int K;
  std::vector<double> C;
  std::vector<float> g;        // [n][K + 1]
  std::vector<uint8_t> split;  // [n][K + 1]: k1 given to the left child (0 = y itself is a slot)
  std::vector<uint8_t> head;
};

static void run_dp(Plan &p, const std::vector<double> &w) {
  const size_t n = nodes.size();
  const int K = p.K;
  p.C.assign(n, 0.0);
  p.g.assign(n * (K + 1), 0.0f);
  p.split.assign(n * (K + 1), 0);
  for (size_t xi = n; xi-- > 0;) {
    const uint32_t x = (uint32_t)xi;
    float *gx = &p.g[xi * (K + 1)];
    uint8_t *sx = &p.split[xi * (K + 1)];
    if (is_leaf(x)) continue;  // all zeros: a leaf slot costs nothing here
    const uint32_t L = left(x), R = right(x);
    const float *gl = &p.g[(size_t)L * (K + 1)], *gr = &p.g[(size_t)R * (K + 1)];
    // x as a head: its own record + the best use of K slots below it (at least its two children)
    double best = 1e300;
    int bk = 1;
    for (int k1 = 1; k1 < K; ++k1) {
      const double c = (double)gl[k1] + (double)gr[K - k1];
      if (c < best) { best = c; bk = k1; }
    }
    p.C[xi] = w[xi] + best;
    sx[0] = (uint8_t)bk;  // the head's own split
    gx[1] = (float)p.C[xi];
    sx[1] = 0;
    for (int k = 2; k <= K; ++k) {  // "at most k slots": y itself as a slot, or split into k1 + (k - k1)
      double b = gx[1];
      uint8_t s = 0;
      for (int k1 = 1; k1 < k; ++k1) {
        const double c = (double)gl[k1] + (double)gr[k - k1];
        if (c < b) { b = c; s = (uint8_t)k1; }
      }
      gx[k] = (float)b;
      sx[k] = s;
    }
  }
  // heads, top-down
  p.head.assign(n, 0);
  p.head[0] = is_leaf(0) ? 0 : 1;
  std::vector<std::pair<uint32_t, int>> st;
  for (size_t xi = 0; xi < n; ++xi) {
    if (!p.head[xi]) continue;
    const uint32_t x = (uint32_t)xi;
    const int k1 = p.split[xi * (K + 1)];
    st.clear();
    st.push_back({left(x), k1});
    st.push_back({right(x), K - k1});
    while (!st.empty()) {
      auto [y, k] = st.back();
      st.pop_back();
      if (is_leaf(y)) continue;
      const uint8_t s = k >= 2 ? p.split[(size_t)y * (K + 1) + k] : 0;
      if (s == 0) { p.head[y] = 1; continue; }  // y is a frontier slot: a head of its own
      const int kk = k;
      st.push_back({left(y), (int)s});
      st.push_back({right(y), kk - (int)s});
    }
  }
}

int main(int argc, char **argv) {
  if (argc < 6) { fprintf(stderr, "usage: blocking_sim nodes values vind queries radius\n"); return 2; }
  {
    auto raw = slurp<Node>(argv[1], true);
    nodes.swap(raw);
  }
  vals = slurp<float>(argv[2]);
  vind = slurp<uint32_t>(argv[3]);
  std::vector<float> qs = slurp<float>(argv[4]);
  const float radius = (float)atof(argv[5]);
  const size_t n = nodes.size(), nq = qs.size() / 6;
  lo.assign(n * 6, 0);
  hi.assign(n * 6, 0);
  depth.assign(n, 0);
  size_t n_inner = 0, n_leaves = 0;
  for (size_t xi = n; xi-- > 0;) {
    const uint32_t x = (uint32_t)xi;
    float *l = &lo[xi * 6], *h = &hi[xi * 6];
    if (is_leaf(x)) {
      ++n_leaves;
      for (int d = 0; d < 6; ++d) { l[d] = 1e30f; h[d] = -1e30f; }
      for (uint32_t i = nodes[x].a; i < nodes[x].b; ++i)
        for (int d = 0; d < 6; ++d) {
          const float v = vals[(size_t)vind[i] + d];
          l[d] = std::min(l[d], v);
          h[d] = std::max(h[d], v);
        }
    } else {
      ++n_inner;
      const uint32_t L = left(x), R = right(x);
      for (int d = 0; d < 6; ++d) {
        l[d] = std::min(lo[(size_t)L * 6 + d], lo[(size_t)R * 6 + d]);
        h[d] = std::max(hi[(size_t)L * 6 + d], hi[(size_t)R * 6 + d]);
      }
    }
  }
  int max_depth = 0;
  for (size_t xi = 0; xi < n; ++xi)
    if (!is_leaf((uint32_t)xi)) {
      depth[left((uint32_t)xi)] = depth[right((uint32_t)xi)] = depth[xi] + 1;
      max_depth = std::max(max_depth, depth[xi] + 1);
    }
  fprintf(stderr, "%zu nodes (%zu inner, %zu leaves), depth %d, %zu queries, radius %g\n", n, n_inner, n_leaves, max_depth, nq, radius);
  // entered sets: training half -> empirical weights, test half -> evaluation
  std::vector<double> w_emp(n, 0.0), w_geo(n, 0.0);
  const double r = std::sqrt((double)radius);
  for (size_t xi = 0; xi < n; ++xi) {
    double v = 1.0;
    for (int d = 0; d < 6; ++d) v *= ((double)hi[xi * 6 + d] - (double)lo[xi * 6 + d]) + 2.0 * r;
    w_geo[xi] = v;
  }
  std::vector<std::vector<uint32_t>> entered(nq);
  std::vector<uint32_t> stk;
  double leaves_entered = 0, pts = 0;
  for (size_t qi = 0; qi < nq; ++qi) {
    const float *q = &qs[qi * 6];
    stk.assign(1, 0u);
    while (!stk.empty()) {
      const uint32_t x = stk.back();
      stk.pop_back();
      if (!(boxdist(x, q) < radius)) continue;
      if (is_leaf(x)) { if (qi & 1) { leaves_entered += 1; pts += nodes[x].b - nodes[x].a; } continue; }
      entered[qi].push_back(x);
      stk.push_back(right(x));
      stk.push_back(left(x));
    }
    if (!(qi & 1)) for (uint32_t x : entered[qi]) w_emp[x] += 1.0;
  }
  const double n_test = (double)(nq / 2);
  // smooth the empirical weights with the geometric ones (scaled to the same total over the inner nodes)
  double se = 0, sg = 0;
  for (size_t xi = 0; xi < n; ++xi) if (!is_leaf((uint32_t)xi)) { se += w_emp[xi]; sg += w_geo[xi]; }
  std::vector<double> w_mix(n);
  for (size_t xi = 0; xi < n; ++xi) w_mix[xi] = w_emp[xi] + 0.25 * w_geo[xi] * (se / sg);
  auto eval = [&](const std::vector<uint8_t> &head) {
    double rec = 0;
    for (size_t qi = 1; qi < nq; qi += 2)
      for (uint32_t x : entered[qi]) rec += head[x];
    return rec / n_test;
  };
  double inner_entered = 0;
  for (size_t qi = 1; qi < nq; qi += 2) inner_entered += (double)entered[qi].size();
  printf("per query (test half): inner nodes entered %.1f, leaves entered %.1f, points tested %.1f\n", inner_entered / n_test,
         leaves_entered / n_test, pts / n_test);
  std::vector<uint8_t> head(n, 0);
  size_t nh = 0;
  for (size_t xi = 0; xi < n; ++xi) { head[xi] = !is_leaf((uint32_t)xi) && (depth[xi] % 2 == 0); nh += head[xi]; }
  printf("%-28s records/query %7.2f   records in the index %9zu\n", "even depth (current quads)", eval(head), nh);
  nh = 0;
  for (size_t xi = 0; xi < n; ++xi) { head[xi] = !is_leaf((uint32_t)xi) && (xi == 0 || depth[xi] % 2 == 1); nh += head[xi]; }
  printf("%-28s records/query %7.2f   records in the index %9zu\n", "root + odd depth", eval(head), nh);
  for (int m = 0; m < 3; ++m) {
    nh = 0;
    for (size_t xi = 0; xi < n; ++xi) { head[xi] = !is_leaf((uint32_t)xi) && (xi == 0 || depth[xi] % 3 == m); nh += head[xi]; }
    printf("octs: root + depth %% 3 == %d   records/query %7.2f   records in the index %9zu\n", m, eval(head), nh);
  }
  for (int K : {4, 6, 7, 8}) {
    for (int which = 0; which < 2; ++which) {
      Plan p;
      p.K = K;
      run_dp(p, which ? w_mix : w_geo);
      nh = 0;
      for (auto h : p.head) nh += h;
      char name[64];
      snprintf(name, sizeof name, "dp-%s K=%d", which ? "emp" : "geo", K);
      printf("%-28s records/query %7.2f   records in the index %9zu\n", name, eval(p.head), nh);
    }
  }
  return 0;
}
