// graph_json.h -- the on-disk graph format of TRG::saveGraph / TRG::loadPrebuiltGraph
// (reference: cpp/trg_planner/core/trg_planner/src/graph/trg.cpp:66-177): what
// nlohmann::json::dump(4) writes for
//     {"nodes": [{"id", "pos": [x, y, z], "state"}], "edges": [{"source", "target", "weight", "dist"}]}
// (nlohmann orders object keys alphabetically).  Host-only and free of engine state, so the reader,
// its validation and the writer are exercised on the CPU (tests/cpp/graph_json_check.cpp, under
// ASan/UBSan, against the real nlohmann/json the reference uses).
#pragma once
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <ostream>
#include <string>
#include <vector>

namespace trg {

struct GraphJson {
  struct N {
    int id;
    float p[3];
    int state;
  };
  struct Ed {
    int s, t;
    float w, d;
  };
  std::vector<N> nodes;  // file order (= the reference's unordered_map iteration order when saved)
  std::vector<Ed> edges;
};

namespace json_detail {
// minimal reader for the schema above: arrays of flat objects with numeric members
struct Cursor {
  const std::string &s;
  size_t i = 0;
  explicit Cursor(const std::string &str) : s(str) {}
  void ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i;
  }
  bool eat(char c) {
    ws();
    if (i < s.size() && s[i] == c) {
      ++i;
      return true;
    }
    return false;
  }
  bool str(std::string &out) {
    ws();
    if (i >= s.size() || s[i] != '"') return false;
    size_t j = s.find('"', i + 1);
    if (j == std::string::npos) return false;
    out = s.substr(i + 1, j - i - 1);
    i = j + 1;
    return true;
  }
  bool num(double &out) {
    ws();
    if (i >= s.size()) return false;
    const char *b = s.c_str() + i;
    char *end = nullptr;
    out = strtod(b, &end);
    if (end == b) return false;
    i += (size_t)(end - b);
    return true;
  }
};
// out-of-range or non-finite numbers become an id validate_graph_json refuses
inline int to_int(double x) { return (x >= -2147483648.0 && x <= 2147483647.0) ? (int)x : INT_MIN; }
}  // namespace json_detail

inline bool parse_graph_json(const std::string &txt, GraphJson &g, std::string &err) {
  using json_detail::Cursor;
  using json_detail::to_int;
  g.nodes.clear();
  g.edges.clear();
  Cursor c(txt);
  auto fail = [&](const char *what) {
    err = std::string("Failed to load graph: ") + what;
    return false;
  };
  if (!c.eat('{')) return fail("not a JSON object");
  while (true) {
    std::string key;
    if (!c.str(key)) break;
    if (!c.eat(':') || !c.eat('[')) return fail("bad array");
    while (c.eat('{')) {
      GraphJson::N n{0, {0, 0, 0}, 0};
      GraphJson::Ed ed{0, 0, 0, 0};
      while (true) {
        std::string k;
        if (!c.str(k)) break;
        if (!c.eat(':')) return fail("bad member");
        double v = 0;
        if (k == "pos") {
          if (!c.eat('[')) return fail("bad pos");
          for (int q = 0; q < 3; ++q) {
            if (!c.num(v)) return fail("bad pos");
            n.p[q] = (float)v;
            c.eat(',');
          }
          if (!c.eat(']')) return fail("bad pos");
        } else {
          if (!c.num(v)) return fail("bad number");
          if (k == "id") n.id = to_int(v);
          else if (k == "state") n.state = to_int(v);
          else if (k == "source") ed.s = to_int(v);
          else if (k == "target") ed.t = to_int(v);
          else if (k == "weight") ed.w = (float)v;
          else if (k == "dist") ed.d = (float)v;
        }
        if (!c.eat(',')) break;
      }
      if (!c.eat('}')) return fail("unterminated object");
      if (key == "nodes") g.nodes.push_back(n);
      else if (key == "edges") g.edges.push_back(ed);
      if (!c.eat(',')) break;
    }
    if (!c.eat(']')) return fail("unterminated array");
    if (!c.eat(',')) break;
  }
  if (!c.eat('}')) return fail("unterminated document");
  return true;
}

// The reference indexes by id through hash maps and so never touches memory out of range; the
// engine's slot == id layout needs every id exactly once in [0, V), every edge end point in range
// and a NodeState that exists (trg.h:27-31: Valid 0, Invalid -1, Frontier 1).
inline bool validate_graph_json(const GraphJson &g, std::string &err) {
  const size_t V = g.nodes.size();
  if (V >= 0x7FFFFFF0u) {
    err = "Failed to load graph: too many nodes";
    return false;
  }
  std::vector<bool> seen(V, false);
  for (const auto &n : g.nodes) {
    if (n.id < 0 || (size_t)n.id >= V) {
      err = "Failed to load graph: node ids are not dense 0..V-1";
      return false;
    }
    if (seen[(size_t)n.id]) {
      err = "Failed to load graph: duplicate node id";
      return false;
    }
    seen[(size_t)n.id] = true;
    if (n.state != 0 && n.state != -1 && n.state != 1) {
      err = "Failed to load graph: unknown node state";
      return false;
    }
  }
  for (const auto &ed : g.edges)
    if (ed.s < 0 || (size_t)ed.s >= V || ed.t < 0 || (size_t)ed.t >= V) {
      err = "Failed to load graph: edge end point out of range";
      return false;
    }
  return true;
}

// Layout of nlohmann::json::dump(4): keys in alphabetical order, 4-space indent, "[]" for an empty
// array.  %.9g of a float converts back to the same float (nlohmann itself prints the shortest
// representation of the float widened to double; both parse to identical fp32 values).
inline void write_graph_json(std::ostream &f, const GraphJson &g) {
  char buf[256];
  f << "{\n    \"edges\": [";
  bool first = true;
  for (const auto &ed : g.edges) {
    snprintf(buf, sizeof(buf),
             "%s\n        {\n            \"dist\": %.9g,\n            \"source\": %d,\n"
             "            \"target\": %d,\n            \"weight\": %.9g\n        }",
             first ? "" : ",", (double)ed.d, ed.s, ed.t, (double)ed.w);
    f << buf;
    first = false;
  }
  f << (first ? "]" : "\n    ]") << ",\n    \"nodes\": [";
  first = true;
  for (const auto &n : g.nodes) {
    snprintf(buf, sizeof(buf),
             "%s\n        {\n            \"id\": %d,\n            \"pos\": [\n                %.9g,\n"
             "                %.9g,\n                %.9g\n            ],\n            \"state\": %d\n"
             "        }",
             first ? "" : ",", n.id, (double)n.p[0], (double)n.p[1], (double)n.p[2], n.state);
    f << buf;
    first = false;
  }
  f << (first ? "]" : "\n    ]") << "\n}";
}

}  // namespace trg
