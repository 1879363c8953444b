// libbuglab_data: native `*.msgpack.l.gz` reader (include/buglab_data.h).  Host only: g++ + zlib.
//
// A shard is a gzip stream of concatenated msgpack objects, one BugLabData map per code snippet
// (reference buglab/utils/msgpackutils.py:11-21).  The reader inflates incrementally, finds the extent
// of the next object with a bounds-checked skip, and walks only the graph part (node strings, edge
// lists, reference nodes) into flat arrays; every other key is copied verbatim into a small msgpack map
// that the Python side decodes.  Subtoken nodes / HasSubtoken edges are added exactly like
// buglab/representations/data.py:97-121 does.
#include "../../include/buglab_data.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

typedef const uint8_t* P;

inline uint64_t be(P p, int n) {
  uint64_t v = 0;
  for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
  return v;
}

// ---- msgpack primitives (bounds checked: nullptr = truncated, sets *bad on a reserved byte) -------------
// header of a container / string / bin: returns pointer past the header and the element or byte count
enum Kind { K_NIL, K_BOOL, K_INT, K_FLOAT, K_STR, K_BIN, K_ARRAY, K_MAP, K_EXT, K_BAD };

struct Head {
  Kind kind;
  uint64_t n;  // elements (array), pairs (map), bytes (str / bin / ext payload incl. type byte), or the value (int)
  int64_t ival;
  P body;      // first byte after the header
};

bool read_head(P p, P end, Head& h) {
  if (p >= end) return false;
  const uint8_t b = *p;
  auto need = [&](int k) { return end - p >= 1 + k; };
  h.ival = 0;
  if (b <= 0x7f) { h.kind = K_INT; h.ival = b; h.body = p + 1; return true; }
  if (b >= 0xe0) { h.kind = K_INT; h.ival = (int8_t)b; h.body = p + 1; return true; }
  if (b >= 0x80 && b <= 0x8f) { h.kind = K_MAP; h.n = b & 0x0f; h.body = p + 1; return true; }
  if (b >= 0x90 && b <= 0x9f) { h.kind = K_ARRAY; h.n = b & 0x0f; h.body = p + 1; return true; }
  if (b >= 0xa0 && b <= 0xbf) { h.kind = K_STR; h.n = b & 0x1f; h.body = p + 1; return true; }
  switch (b) {
    case 0xc0: h.kind = K_NIL; h.body = p + 1; return true;
    case 0xc2: case 0xc3: h.kind = K_BOOL; h.ival = b & 1; h.body = p + 1; return true;
    case 0xc4: case 0xc5: case 0xc6: {
      const int k = 1 << (b - 0xc4);
      if (!need(k)) return false;
      h.kind = K_BIN; h.n = be(p + 1, k); h.body = p + 1 + k; return true;
    }
    case 0xc7: case 0xc8: case 0xc9: {
      const int k = 1 << (b - 0xc7);
      if (!need(k)) return false;
      h.kind = K_EXT; h.n = be(p + 1, k) + 1; h.body = p + 1 + k; return true;
    }
    case 0xca: if (!need(4)) return false; h.kind = K_FLOAT; h.n = 4; h.body = p + 1; return true;
    case 0xcb: if (!need(8)) return false; h.kind = K_FLOAT; h.n = 8; h.body = p + 1; return true;
    case 0xcc: case 0xcd: case 0xce: case 0xcf: {
      const int k = 1 << (b - 0xcc);
      if (!need(k)) return false;
      h.kind = K_INT; h.ival = (int64_t)be(p + 1, k); h.body = p + 1 + k; return true;
    }
    case 0xd0: if (!need(1)) return false; h.kind = K_INT; h.ival = (int8_t)p[1]; h.body = p + 2; return true;
    case 0xd1: if (!need(2)) return false; h.kind = K_INT; h.ival = (int16_t)be(p + 1, 2); h.body = p + 3; return true;
    case 0xd2: if (!need(4)) return false; h.kind = K_INT; h.ival = (int32_t)be(p + 1, 4); h.body = p + 5; return true;
    case 0xd3: if (!need(8)) return false; h.kind = K_INT; h.ival = (int64_t)be(p + 1, 8); h.body = p + 9; return true;
    case 0xd4: case 0xd5: case 0xd6: case 0xd7: case 0xd8:
      h.kind = K_EXT; h.n = (1u << (b - 0xd4)) + 1; h.body = p + 1; return true;
    case 0xd9: case 0xda: case 0xdb: {
      const int k = 1 << (b - 0xd9);
      if (!need(k)) return false;
      h.kind = K_STR; h.n = be(p + 1, k); h.body = p + 1 + k; return true;
    }
    case 0xdc: case 0xdd: {
      const int k = b == 0xdc ? 2 : 4;
      if (!need(k)) return false;
      h.kind = K_ARRAY; h.n = be(p + 1, k); h.body = p + 1 + k; return true;
    }
    case 0xde: case 0xdf: {
      const int k = b == 0xde ? 2 : 4;
      if (!need(k)) return false;
      h.kind = K_MAP; h.n = be(p + 1, k); h.body = p + 1 + k; return true;
    }
    default: h.kind = K_BAD; h.body = p + 1; return true;  // 0xc1: never used
  }
}

// pointer past one complete object, nullptr if it is truncated; *bad set on a malformed byte.
// Iterative (a work counter instead of recursion): shards hold deeply nested but finite objects.
P skip_object(P p, P end, bool* bad) {
  uint64_t pending = 1;
  while (pending) {
    Head h;
    if (!read_head(p, end, h)) return nullptr;
    --pending;
    switch (h.kind) {
      case K_BAD: *bad = true; return nullptr;
      case K_ARRAY: pending += h.n; p = h.body; break;
      case K_MAP: pending += 2 * h.n; p = h.body; break;
      case K_STR: case K_BIN: case K_EXT: case K_FLOAT:
        if ((uint64_t)(end - h.body) < h.n) return nullptr;
        p = h.body + h.n;
        break;
      default: p = h.body; break;
    }
  }
  return p;
}

// ---- subtoken splitting (buglab/runtime/vocabulary.py::split_identifier_into_parts) -----------------------
inline bool is_up(uint8_t c) { return c >= 'A' && c <= 'Z'; }
inline bool is_lo(uint8_t c) { return c >= 'a' && c <= 'z'; }
inline bool is_dg(uint8_t c) { return c >= '0' && c <= '9'; }
inline bool is_alnum(uint8_t c) { return is_up(c) || is_lo(c) || is_dg(c); }

// regex alternatives, in order:  [A-Z]+(?=[A-Z][a-z]) | [A-Z]?[a-z]+ | [A-Z]+ | [0-9]+ | [^A-Za-z0-9]+
// applied with finditer to every non-empty '_'-separated piece; parts are lower-cased (ASCII here).
template <class Emit>
void split_piece(const uint8_t* s, size_t n, Emit&& emit) {
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    if (is_up(s[i])) {
      size_t u = i;
      while (u < n && is_up(s[u])) ++u;  // maximal run of capitals [i, u)
      // alt 1: greedy [A-Z]+ with backtracking so that "[A-Z][a-z]" follows: the run minus its last capital,
      // when a lower-case letter follows the run (needs >= 2 capitals)
      if (u < n && is_lo(s[u]) && u - i >= 2) {
        j = u - 1;
      } else if (u - i == 1 && u < n && is_lo(s[u])) {  // alt 2: one capital then [a-z]+
        j = u;
        while (j < n && is_lo(s[j])) ++j;
      } else {
        j = u;  // alt 3: [A-Z]+
      }
    } else if (is_lo(s[i])) {
      while (j < n && is_lo(s[j])) ++j;
    } else if (is_dg(s[i])) {
      while (j < n && is_dg(s[j])) ++j;
    } else {
      while (j < n && !is_alnum(s[j])) ++j;
    }
    emit(s + i, j - i);
    i = j;
  }
}

template <class Emit>
int split_identifier(const uint8_t* s, size_t n, Emit&& emit) {
  int parts = 0;
  size_t i = 0;
  while (i <= n) {
    size_t j = i;
    while (j < n && s[j] != '_') ++j;
    if (j > i) split_piece(s + i, j - i, [&](const uint8_t* q, size_t m) { ++parts; emit(q, m); });
    i = j + 1;
  }
  if (parts == 0) { emit(s, n); parts = 1; }  // an identifier with no parts is returned whole
  return parts;
}

inline void lower_into(std::string& out, const uint8_t* q, size_t m) {
  out.assign((const char*)q, m);
  for (auto& c : out)
    if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
}

inline bool has_non_ascii(const uint8_t* s, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (s[i] & 0x80) return true;
  return false;
}

}  // namespace

struct bl_vocab {
  std::unordered_map<std::string, int32_t> ids;
};

struct bl_reader {
  gzFile gz = nullptr;
  std::vector<uint8_t> buf;
  size_t cur = 0, end = 0;
  bool eof = false;
  // storage of the current datapoint
  std::string node_text;
  std::vector<int32_t> node_off;
  std::vector<std::string> kind_names;
  std::vector<const char*> kind_ptr;
  std::vector<std::vector<int32_t>> pairs, feats;
  std::vector<const int32_t*> pairs_ptr, feats_ptr;
  std::vector<int32_t> counts;
  std::string feat_text;
  std::vector<int32_t> feat_off;
  std::vector<int32_t> refs;
  std::vector<uint8_t> rest;
};

namespace {

bool fill(bl_reader* r) {  // read more inflated bytes; false at end of file
  if (r->eof) return false;
  if (r->cur > 0 && r->cur == r->end) r->cur = r->end = 0;
  if (r->cur > (1u << 20)) {  // compact
    memmove(r->buf.data(), r->buf.data() + r->cur, r->end - r->cur);
    r->end -= r->cur;
    r->cur = 0;
  }
  const size_t chunk = 1u << 20;
  if (r->buf.size() < r->end + chunk) r->buf.resize(std::max(r->buf.size() * 2, r->end + chunk));
  const int got = gzread(r->gz, r->buf.data() + r->end, (unsigned)chunk);
  if (got < 0) {
    int e = 0;
    set_err("gzread failed: %s", gzerror(r->gz, &e));
    r->eof = true;
    return false;
  }
  if (got == 0) { r->eof = true; return false; }
  r->end += (size_t)got;
  return true;
}

void put_map_header(std::vector<uint8_t>& o, uint32_t n) {
  if (n < 16) o.push_back((uint8_t)(0x80 | n));
  else if (n < 65536) { o.push_back(0xde); o.push_back((uint8_t)(n >> 8)); o.push_back((uint8_t)n); }
  else { o.push_back(0xdf); for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(n >> s)); }
}

bool key_is(const Head& h, const char* name) {
  const size_t n = strlen(name);
  return (h.kind == K_STR || h.kind == K_BIN) && h.n == n && memcmp(h.body, name, n) == 0;
}

#define FAIL(...) do { set_err(__VA_ARGS__); return false; } while (0)

bool parse_nodes(bl_reader* r, P p, P end) {
  Head h;
  if (!read_head(p, end, h) || h.kind != K_ARRAY) FAIL("graph.nodes is not an array");
  p = h.body;
  r->node_off.push_back(0);
  for (uint64_t i = 0; i < h.n; ++i) {
    Head s;
    if (!read_head(p, end, s) || (s.kind != K_STR && s.kind != K_BIN)) FAIL("graph.nodes[%llu] is not a string", (unsigned long long)i);
    r->node_text.append((const char*)s.body, s.n);
    r->node_off.push_back((int32_t)r->node_text.size());
    p = s.body + s.n;
  }
  return true;
}

bool parse_edges(bl_reader* r, P p, P end) {
  Head m;
  if (!read_head(p, end, m) || m.kind != K_MAP) FAIL("graph.edges is not a map");
  p = m.body;
  r->feat_off.push_back(0);
  bool bad = false;
  for (uint64_t k = 0; k < m.n; ++k) {
    Head key;
    if (!read_head(p, end, key) || (key.kind != K_STR && key.kind != K_BIN)) FAIL("edge kind is not a string");
    r->kind_names.emplace_back((const char*)key.body, key.n);
    p = key.body + key.n;
    Head arr;
    if (!read_head(p, end, arr) || arr.kind != K_ARRAY) FAIL("edge list of '%s' is not an array", r->kind_names.back().c_str());
    p = arr.body;
    std::vector<int32_t> pr, ft;
    pr.reserve(2 * arr.n);
    ft.reserve(arr.n);
    for (uint64_t e = 0; e < arr.n; ++e) {
      Head edge;
      if (!read_head(p, end, edge) || edge.kind != K_ARRAY || edge.n < 2) FAIL("edge %llu of '%s' is not [src, tgt, ...]", (unsigned long long)e, r->kind_names.back().c_str());
      p = edge.body;
      for (int c = 0; c < 2; ++c) {
        Head v;
        if (!read_head(p, end, v) || v.kind != K_INT) FAIL("edge endpoint is not an integer");
        pr.push_back((int32_t)v.ival);
        p = v.body;
      }
      int32_t feat = -1;
      for (uint64_t c = 2; c < edge.n; ++c) {
        Head v;
        if (!read_head(p, end, v)) FAIL("truncated edge");
        if (c == 2 && (v.kind == K_STR || v.kind == K_BIN)) {
          feat = (int32_t)r->feat_off.size() - 1;
          r->feat_text.append((const char*)v.body, v.n);
          r->feat_off.push_back((int32_t)r->feat_text.size());
          p = v.body + v.n;
        } else {
          if (c == 2) feat = -2;  // a non-string third element: the Python path must look at it
          p = skip_object(p, end, &bad);
          if (!p) FAIL("malformed edge element");
        }
      }
      ft.push_back(feat);
    }
    r->counts.push_back((int32_t)arr.n);
    r->pairs.push_back(std::move(pr));
    r->feats.push_back(std::move(ft));
  }
  return true;
}

bool parse_refs(bl_reader* r, P p, P end) {
  Head h;
  if (!read_head(p, end, h) || h.kind != K_ARRAY) FAIL("graph.reference_nodes is not an array");
  p = h.body;
  for (uint64_t i = 0; i < h.n; ++i) {
    Head v;
    if (!read_head(p, end, v) || v.kind != K_INT) FAIL("reference node is not an integer");
    r->refs.push_back((int32_t)v.ival);
    p = v.body;
  }
  return true;
}

// Iteration order of a CPython `set` of non-negative ints after the given insertion sequence.  The reference walks
// `token_nodes` (a set filled with n1, n2 of every NextToken edge, buglab/representations/data.py:98-103) in SET order
// (:109), which numbers the subtoken nodes and orders the HasSubtoken edges -- so the order is part of the data contract.
// This is Objects/setobject.c of CPython 3.7 - 3.12 restated: open addressing, hash(n) = n, LINEAR_PROBES = 9 probes
// after the home slot (only when they do not run past the table), then i = 5 i + 1 + (perturb >>= 5); growth when
// fill * 5 >= mask * 3 to the first power of two above used * 4 (used * 2 beyond 50 000 entries), re-inserting the old
// table in slot order.  Nothing is ever removed here, so there are no dummy entries.
std::vector<int32_t> cpython_int_set_order(const std::vector<int32_t>& inserted) {
  std::vector<int64_t> table(8, -1);  // -1 = unused slot
  size_t mask = 7, used = 0;
  auto insert_clean = [](std::vector<int64_t>& tb, size_t m, int64_t key) {
    size_t perturb = (size_t)key, i = (size_t)key & m;
    for (;;) {
      if (tb[i] < 0) { tb[i] = key; return; }
      if (i + 9 <= m) {
        for (size_t j = 1; j <= 9; ++j)
          if (tb[i + j] < 0) { tb[i + j] = key; return; }
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & m;
    }
  };
  for (int32_t k32 : inserted) {
    const int64_t key = k32;
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    bool found = false, placed = false;
    while (!found && !placed) {
      const size_t probes = (i + 9 <= mask) ? 9 : 0;
      for (size_t j = 0; j <= probes; ++j) {
        if (table[i + j] < 0) { table[i + j] = key; placed = true; break; }
        if (table[i + j] == key) { found = true; break; }
      }
      if (found || placed) break;
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
    if (!placed) continue;
    ++used;
    if (used * 5 < mask * 3) continue;
    const size_t minused = used > 50000 ? used * 2 : used * 4;
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    std::vector<int64_t> nt(newsize, -1);
    for (int64_t v : table)
      if (v >= 0) insert_clean(nt, newsize - 1, v);
    table.swap(nt);
    mask = newsize - 1;
  }
  std::vector<int32_t> order;
  order.reserve(used);
  for (int64_t v : table)
    if (v >= 0) order.push_back((int32_t)v);
  return order;
}

// buglab/representations/data.py:97-121
void add_open_vocab(bl_reader* r, bl_datapoint_t* out) {
  int next_token = -1;
  for (size_t k = 0; k < r->kind_names.size(); ++k) {
    if (r->kind_names[k] == "HasSubtoken") return;  // already present in the file
    if (r->kind_names[k] == "NextToken") next_token = (int)k;
  }
  if (next_token < 0) return;
  // the reference iterates the SET of token nodes: same order here (node ids are non-negative msgpack ints)
  bool negative = false;
  for (int32_t v : r->pairs[next_token]) negative |= v < 0;
  if (negative) { out->non_ascii_identifier = 1; return; }  // hash(-1) = -2 etc.: leave such a file to the Python path
  out->created_has_subtoken = 1;
  const std::vector<int32_t> toks = cpython_int_set_order(r->pairs[next_token]);
  std::unordered_map<std::string, int32_t> vocab_nodes;
  std::vector<int32_t> pr;
  const int32_t n0 = (int32_t)r->node_off.size() - 1;
  std::string part;
  for (int32_t node : toks) {
    if (node < 0 || node >= n0) continue;
    // copy: node_text grows below
    const std::string s(r->node_text.data() + r->node_off[node], (size_t)(r->node_off[node + 1] - r->node_off[node]));
    const uint8_t* u = (const uint8_t*)s.data();
    if (s.empty() || !(is_up(u[0]) || is_lo(u[0]) || u[0] == '_')) continue;  // IS_IDENTIFIER.match = prefix match
    if (has_non_ascii(u, s.size())) { out->non_ascii_identifier = 1; continue; }
    split_identifier(u, s.size(), [&](const uint8_t* q, size_t m) {
      lower_into(part, q, m);
      auto it = vocab_nodes.find(part);
      int32_t idx;
      if (it == vocab_nodes.end()) {
        idx = (int32_t)r->node_off.size() - 1;
        r->node_text.append(part);
        r->node_off.push_back((int32_t)r->node_text.size());
        vocab_nodes.emplace(part, idx);
      } else {
        idx = it->second;
      }
      pr.push_back(node);
      pr.push_back(idx);
    });
  }
  r->kind_names.emplace_back("HasSubtoken");
  r->counts.push_back((int32_t)(pr.size() / 2));
  r->feats.emplace_back(pr.size() / 2, -1);
  r->pairs.push_back(std::move(pr));
}

bool parse_datapoint(bl_reader* r, P p, P end, bl_datapoint_t* out) {
  Head top;
  if (!read_head(p, end, top)) FAIL("truncated datapoint");
  if (top.kind == K_NIL) { out->is_nil = 1; return true; }
  if (top.kind != K_MAP) FAIL("datapoint is not a map");
  bool bad = false;
  p = top.body;
  // rest = { other top-level keys (verbatim) ..., "graph": { other graph keys (verbatim) } }
  std::vector<std::pair<P, P>> keep_top, keep_graph;
  bool saw_graph = false;
  for (uint64_t i = 0; i < top.n; ++i) {
    Head key;
    if (!read_head(p, end, key)) FAIL("truncated key");
    P kbeg = p;
    P vbeg = skip_object(p, end, &bad);
    if (!vbeg) FAIL("malformed key");
    P vend = skip_object(vbeg, end, &bad);
    if (!vend) FAIL("malformed value");
    if (key_is(key, "graph")) {
      saw_graph = true;
      Head g;
      if (!read_head(vbeg, vend, g) || g.kind != K_MAP) FAIL("graph is not a map");
      P q = g.body;
      for (uint64_t j = 0; j < g.n; ++j) {
        Head gk;
        if (!read_head(q, vend, gk)) FAIL("truncated graph key");
        P gkbeg = q;
        P gvbeg = skip_object(q, vend, &bad);
        if (!gvbeg) FAIL("malformed graph key");
        P gvend = skip_object(gvbeg, vend, &bad);
        if (!gvend) FAIL("malformed graph value");
        if (key_is(gk, "nodes")) { if (!parse_nodes(r, gvbeg, gvend)) return false; }
        else if (key_is(gk, "edges")) { if (!parse_edges(r, gvbeg, gvend)) return false; }
        else if (key_is(gk, "reference_nodes")) { if (!parse_refs(r, gvbeg, gvend)) return false; }
        else keep_graph.emplace_back(gkbeg, gvend);
        q = gvend;
      }
    } else {
      keep_top.emplace_back(kbeg, vend);
    }
    p = vend;
  }
  if (!saw_graph) FAIL("datapoint has no 'graph'");
  if (r->node_off.empty()) r->node_off.push_back(0);
  if (r->feat_off.empty()) r->feat_off.push_back(0);
  out->num_file_nodes = (int32_t)r->node_off.size() - 1;
  add_open_vocab(r, out);
  put_map_header(r->rest, (uint32_t)keep_top.size() + 1);
  for (auto& kv : keep_top) r->rest.insert(r->rest.end(), kv.first, kv.second);
  r->rest.push_back(0xa5);
  r->rest.insert(r->rest.end(), (const uint8_t*)"graph", (const uint8_t*)"graph" + 5);
  put_map_header(r->rest, (uint32_t)keep_graph.size());
  for (auto& kv : keep_graph) r->rest.insert(r->rest.end(), kv.first, kv.second);
  return true;
}

}  // namespace

extern "C" const char* bl_data_last_error(void) { return g_err; }
extern "C" int32_t bl_data_version(void) { return 3; }  // 3: bl_graph_in_t.adj_feat / bl_collated_t.msg_feat

extern "C" int32_t bl_pyset_order(const int32_t* inserted, int32_t n, int32_t* out) {
  if (n < 0 || (n > 0 && (!inserted || !out))) { set_err("bl_pyset_order: bad arguments"); return -1; }
  std::vector<int32_t> in(inserted, inserted + n);
  for (int32_t v : in)
    if (v < 0) { set_err("bl_pyset_order: negative key"); return -1; }
  const std::vector<int32_t> order = cpython_int_set_order(in);
  std::copy(order.begin(), order.end(), out);
  return (int32_t)order.size();
}

extern "C" bl_reader* bl_reader_open(const char* path) {
  gzFile gz = gzopen(path, "rb");
  if (!gz) {
    set_err("cannot open %s", path);
    return nullptr;
  }
  gzbuffer(gz, 1u << 18);
  bl_reader* r = new bl_reader();
  r->gz = gz;
  r->buf.resize(1u << 21);
  return r;
}

extern "C" void bl_reader_close(bl_reader* r) {
  if (!r) return;
  if (r->gz) gzclose(r->gz);
  delete r;
}

extern "C" int32_t bl_reader_next(bl_reader* r, bl_datapoint_t* out) {
  if (!r || !out) { set_err("null reader / output"); return -1; }
  memset(out, 0, sizeof(*out));
  r->node_text.clear(); r->node_off.clear(); r->kind_names.clear(); r->kind_ptr.clear(); r->pairs.clear(); r->feats.clear();
  r->pairs_ptr.clear(); r->feats_ptr.clear(); r->counts.clear(); r->feat_text.clear(); r->feat_off.clear(); r->refs.clear();
  r->rest.clear();
  P obj_end = nullptr;
  for (;;) {
    bool bad = false;
    if (r->cur < r->end) {
      obj_end = skip_object(r->buf.data() + r->cur, r->buf.data() + r->end, &bad);
      if (bad) { set_err("malformed msgpack at stream offset of the current object"); return -2; }
      if (obj_end) break;
    }
    if (!fill(r)) {
      if (g_err[0] && r->eof && strstr(g_err, "gzread")) return -3;
      if (r->cur == r->end) return 0;
      set_err("truncated msgpack object at end of stream");
      return -2;
    }
  }
  P beg = r->buf.data() + r->cur;
  r->cur = (size_t)(obj_end - r->buf.data());
  if (!parse_datapoint(r, beg, obj_end, out)) return -2;
  if (out->is_nil) return 1;
  for (size_t k = 0; k < r->kind_names.size(); ++k) {
    r->kind_ptr.push_back(r->kind_names[k].c_str());
    r->pairs_ptr.push_back(r->pairs[k].data());
    r->feats_ptr.push_back(r->feats[k].data());
  }
  out->num_nodes = (int32_t)r->node_off.size() - 1;
  out->node_text = r->node_text.data();
  out->node_text_off = r->node_off.data();
  out->num_edge_kinds = (int32_t)r->kind_names.size();
  out->edge_kind = r->kind_ptr.data();
  out->edge_pairs = r->pairs_ptr.data();
  out->edge_feat = r->feats_ptr.data();
  out->edge_count = r->counts.data();
  out->feat_text = r->feat_text.data();
  out->feat_text_off = r->feat_off.data();
  out->num_feats = (int32_t)r->feat_off.size() - 1;
  out->reference_nodes = r->refs.data();
  out->num_reference_nodes = (int32_t)r->refs.size();
  out->rest = r->rest.data();
  out->rest_len = (int64_t)r->rest.size();
  return 1;
}

extern "C" bl_vocab* bl_vocab_create(const char* text, const int32_t* off, int32_t n) {
  if (!text || !off || n < 0) { set_err("bl_vocab_create: null argument"); return nullptr; }
  bl_vocab* v = new bl_vocab();
  v->ids.reserve((size_t)n * 2);
  for (int32_t i = 0; i < n; ++i) v->ids.emplace(std::string(text + off[i], (size_t)(off[i + 1] - off[i])), i);
  return v;
}

extern "C" void bl_vocab_free(bl_vocab* v) { delete v; }

extern "C" int32_t bl_tensorize_nodes(const bl_vocab* v, int32_t unk_id, const char* text, const int32_t* off, int32_t n,
                                      int32_t S, int32_t* ids, int32_t* lens, uint8_t* needs_python) {
  if (!v || !text || !off || !ids || !lens || !needs_python || S < 1) { set_err("bl_tensorize_nodes: null argument or S < 1"); return -1; }
  std::string part;
  for (int32_t i = 0; i < n; ++i) {
    const uint8_t* s = (const uint8_t*)text + off[i];
    const size_t len = (size_t)(off[i + 1] - off[i]);
    int32_t* row = ids + (size_t)i * S;
    for (int32_t k = 0; k < S; ++k) row[k] = 0;
    needs_python[i] = 0;
    if (has_non_ascii(s, len)) { needs_python[i] = 1; lens[i] = 1; continue; }
    int kept = 0;
    split_identifier(s, len, [&](const uint8_t* q, size_t m) {
      if (kept >= S) return;
      lower_into(part, q, m);
      auto it = v->ids.find(part);
      row[kept++] = it == v->ids.end() ? unk_id : it->second;
    });
    lens[i] = kept > 0 ? kept : 1;
  }
  return 0;
}

// Stable counting sort of `keys` in [0, K): the collator's CSRs (node -> incoming / outgoing messages),
// the target order inside an edge type and the token-sorted subtoken occurrences are all this primitive;
// NumPy's stable argsort on 64-bit keys (timsort) took 150 of the 160 ms a config-c2 minibatch needs.
extern "C" int32_t bl_counting_sort(const int32_t* keys, int64_t E, int32_t K, int32_t* ptr, int32_t* perm) {
  if ((E > 0 && (!keys || !perm)) || !ptr || K < 0 || E < 0 || E > 0x7fffffffLL) { set_err("bl_counting_sort: bad argument"); return -1; }
  for (int32_t k = 0; k <= K; ++k) ptr[k] = 0;
  for (int64_t i = 0; i < E; ++i) {
    const int32_t k = keys[i];
    if (k < 0 || k >= K) { set_err("bl_counting_sort: key %d outside [0, %d)", k, K); return -2; }
    ++ptr[k + 1];
  }
  for (int32_t k = 0; k < K; ++k) ptr[k + 1] += ptr[k];
  std::vector<int32_t> next(ptr, ptr + K);
  for (int64_t i = 0; i < E; ++i) perm[next[keys[i]]++] = (int32_t)i;
  return 0;
}

// ---- native collator for the graph part of a minibatch (buglab/data/collate.py::collate_graphs) ------------------
// Disjoint union of B tensorised graphs: node arrays concatenated, messages merged type-major and target-sorted
// inside a type (ties in (graph, list) order), both CSRs, the hub-first node order and the token-sorted
// subtoken occurrence chunks -- all in one pass-structured function without Python objects, so several
// minibatches collate concurrently in worker threads (ctypes releases the GIL for the call).
// reference: GnnBugLabModel.initialize/extend/finalize_minibatch, gnn.py:431-604 (Python list appends).
extern "C" int32_t bl_collate_graphs(const bl_graph_in_t* graphs, int32_t B, int32_t T, int32_t S, int32_t hub_degree,
                                     int32_t token_chunk, bl_collated_t* out) {
  if (!graphs || !out || B < 0 || T < 1 || S < 1 || token_chunk < 1) { set_err("bl_collate_graphs: bad argument"); return -1; }
  // node offsets, token arrays
  int64_t N = 0, E = 0;
  std::vector<int64_t> node_off((size_t)B + 1, 0);
  for (int32_t b = 0; b < B; ++b) {
    const bl_graph_in_t& g = graphs[b];
    if (g.num_nodes < 0 || g.token_stride < 1 || g.token_stride > S) { set_err("bl_collate_graphs: graph %d: bad node count / token stride", b); return -1; }
    node_off[b + 1] = node_off[b] + g.num_nodes;
    for (int32_t t = 0; t < T; ++t) E += g.adj_count ? g.adj_count[t] : 0;
  }
  N = node_off[B];
  if (N != out->num_nodes || E != out->num_messages) { set_err("bl_collate_graphs: output sized for N=%lld E=%lld, inputs hold N=%lld E=%lld", (long long)out->num_nodes, (long long)out->num_messages, (long long)N, (long long)E); return -1; }
  if (N > 0x7fffffffLL || E > 0x7fffffffLL) { set_err("bl_collate_graphs: more than 2^31 nodes or messages"); return -1; }
  int64_t n_occ = 0;
  for (int32_t b = 0; b < B; ++b) {
    const bl_graph_in_t& g = graphs[b];
    for (int32_t i = 0; i < g.num_nodes; ++i) {
      int32_t* row = out->token_ids + (size_t)(node_off[b] + i) * S;
      const int32_t* in = g.token_ids + (size_t)i * g.token_stride;
      for (int32_t k = 0; k < g.token_stride; ++k) row[k] = in[k];
      for (int32_t k = g.token_stride; k < S; ++k) row[k] = 0;
      const int32_t len = g.token_lens[i];
      out->token_lens[node_off[b] + i] = len;
      n_occ += len < 0 ? 0 : (len > S ? S : len);
    }
  }
  // messages: per type, gather in graph order, then a stable counting sort by target inside the type
  std::vector<int32_t> tsrc, ttgt, tfeat, cnt((size_t)N + 1);
  const bool with_feat = out->msg_feat != nullptr;  // a per-edge payload (edge-feature token ids) travels with the messages
  int64_t pos = 0;
  out->type_ptr[0] = 0;
  for (int32_t t = 0; t < T; ++t) {
    tsrc.clear(); ttgt.clear(); tfeat.clear();
    for (int32_t b = 0; b < B; ++b) {
      const bl_graph_in_t& g = graphs[b];
      const int32_t c = g.adj_count ? g.adj_count[t] : 0;
      const int32_t* a = g.adj ? g.adj[t] : nullptr;
      const int32_t off = (int32_t)node_off[b];
      if (with_feat && c > 0 && (!g.adj_feat || !g.adj_feat[t])) { set_err("bl_collate_graphs: graph %d type %d: msg_feat requested but no adj_feat", b, t); return -1; }
      for (int32_t e = 0; e < c; ++e) {
        const int32_t u = a[2 * e], v = a[2 * e + 1];
        if (u < 0 || u >= g.num_nodes || v < 0 || v >= g.num_nodes) { set_err("bl_collate_graphs: graph %d type %d edge %d: node id out of range", b, t, e); return -2; }
        tsrc.push_back(u + off);
        ttgt.push_back(v + off);
        if (with_feat) tfeat.push_back(g.adj_feat[t][e]);
      }
    }
    const size_t m = ttgt.size();
    if (m) {
      // counting sort over the target ids that occur: min..max window keeps the histogram small per type
      int32_t lo = ttgt[0], hi = ttgt[0];
      for (size_t i = 1; i < m; ++i) { lo = std::min(lo, ttgt[i]); hi = std::max(hi, ttgt[i]); }
      const size_t span = (size_t)(hi - lo) + 2;
      std::fill(cnt.begin(), cnt.begin() + span, 0);
      for (size_t i = 0; i < m; ++i) ++cnt[(size_t)(ttgt[i] - lo) + 1];
      for (size_t k = 1; k < span; ++k) cnt[k] += cnt[k - 1];
      for (size_t i = 0; i < m; ++i) {
        const int32_t p = cnt[(size_t)(ttgt[i] - lo)]++;
        out->msg_src[pos + p] = tsrc[i];
        out->msg_tgt[pos + p] = ttgt[i];
        if (with_feat) out->msg_feat[pos + p] = tfeat[i];
      }
    }
    pos += (int64_t)m;
    out->type_ptr[t + 1] = (int32_t)pos;
  }
  // CSRs node -> incoming / outgoing message ids (ascending)
  auto csr = [&](const int32_t* keys, int32_t* ptr, int32_t* items) {
    for (int64_t k = 0; k <= N; ++k) ptr[k] = 0;
    for (int64_t i = 0; i < E; ++i) ++ptr[keys[i] + 1];
    for (int64_t k = 0; k < N; ++k) ptr[k + 1] += ptr[k];
    std::copy(ptr, ptr + N, cnt.begin());
    for (int64_t i = 0; i < E; ++i) items[cnt[keys[i]]++] = (int32_t)i;
  };
  csr(out->msg_tgt, out->tgt_ptr, out->tgt_msgs);
  csr(out->msg_src, out->src_ptr, out->src_msgs);
  // hub-first processing order of the per-node kernels
  {
    std::vector<int32_t> hubs;
    auto deg = [&](int32_t n) { return (out->tgt_ptr[n + 1] - out->tgt_ptr[n]) + (out->src_ptr[n + 1] - out->src_ptr[n]); };
    for (int32_t n = 0; n < (int32_t)N; ++n)
      if (deg(n) > hub_degree) hubs.push_back(n);
    std::stable_sort(hubs.begin(), hubs.end(), [&](int32_t a, int32_t b) { return deg(a) > deg(b); });
    int64_t w = 0;
    for (int32_t h : hubs) out->node_order[w++] = h;
    for (int32_t n = 0; n < (int32_t)N; ++n)
      if (deg(n) <= hub_degree) out->node_order[w++] = n;
  }
  // token-sorted subtoken occurrences, cut in chunks of one token each
  if (n_occ > out->occ_capacity) { set_err("bl_collate_graphs: occ_capacity %lld < %lld occurrences", (long long)out->occ_capacity, (long long)n_occ); return -1; }
  {
    int32_t vmax = -1;
    for (int64_t n = 0; n < N; ++n) {
      const int32_t len = std::min(std::max(out->token_lens[n], 0), S);
      for (int32_t k = 0; k < len; ++k) {
        const int32_t id = out->token_ids[(size_t)n * S + k];
        if (id < 0) { set_err("bl_collate_graphs: negative token id"); return -2; }
        vmax = std::max(vmax, id);
      }
    }
    std::vector<int32_t> tcnt((size_t)vmax + 2, 0);
    for (int64_t n = 0; n < N; ++n) {
      const int32_t len = std::min(std::max(out->token_lens[n], 0), S);
      for (int32_t k = 0; k < len; ++k) ++tcnt[(size_t)out->token_ids[(size_t)n * S + k] + 1];
    }
    for (int32_t v = 0; v <= vmax; ++v) tcnt[(size_t)v + 1] += tcnt[v];
    std::vector<int32_t> tstart(tcnt.begin(), tcnt.end());
    for (int64_t n = 0; n < N; ++n) {
      const int32_t len = std::min(std::max(out->token_lens[n], 0), S);
      for (int32_t k = 0; k < len; ++k) out->tok_occ[tcnt[out->token_ids[(size_t)n * S + k]]++] = (int32_t)(n * S + k);
    }
    int64_t nchunks = 0;
    for (int32_t v = 0; v <= vmax; ++v) {
      for (int32_t beg = tstart[v]; beg < tstart[(size_t)v + 1]; beg += token_chunk) {
        out->tok_chunk_ptr[nchunks] = beg;
        out->tok_chunk_id[nchunks] = v;
        ++nchunks;
      }
    }
    out->tok_chunk_ptr[nchunks] = (int32_t)n_occ;
    out->num_occ = n_occ;
    out->num_chunks = nchunks;
  }
  return 0;
}
