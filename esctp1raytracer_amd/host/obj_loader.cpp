// obj_loader.cpp -- OBJ/MTL -> esc_scene, the load-time surface the renderer keeps.
//
// Behavioural restatement of model::loadobj (/root/reference/src/scene/sceneloader.cpp:14-106)
// on top of the vendored tinyobjloader it calls (src/scene/tiny_obj_loader.h, v1.0.x).  Only
// the OBSERVABLE behaviour is reproduced; the code is organised as a small
// recursive-descent reader, not as tinyobj's token macros:
//   * numbers: digits accumulated in double as  m = m*10 + d ; fraction digits added as
//     d * 10^-k (table for k < 8, pow beyond) ; exponent applied as ldexp(m * 5^e, e) ;
//     result narrowed to float -- NOT strtof (tiny_obj_loader.h:465-587).  The loaded vertex
//     bits, and therefore pixels, depend on this.
//   * faces: fan triangulation (i0,i1,i2),(i0,i2,i3).. (:899-926); negative = relative
//     index (:416-420); v, v/t, v//n, v/t/n.
//   * shapes: `g` / `o` flush the pending faces into a shape only when some are pending and
//     ALWAYS start a new empty shape (:1592-1649); a `usemtl` that changes the material
//     moves the pending faces into the CURRENT shape without closing it (:1519-1546) -- so in
//     CornellBox-Original.obj the short box joins "leftWall" (SURVEY.md quirk S11); faces
//     already moved are dropped if a `g` follows with nothing pending.
//   * MTL: defaults 0 / Ns = 1 (:844-864); any warning (missing file, `d` together with
//     `Tr`) is an error for the renderer (sceneloader.cpp:27-30, quirk S10).
//   * per shape: material of the FIRST face (sceneloader.cpp:52); vertices de-indexed three
//     per face (:73-98); normals pushed only where the face references one, normalised with
//     vec.h's v / sqrt(dot(v,v)) (:84-89); light source iff dot(ke,ke) > 0 (:63-64).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "scene.h"

namespace esc {
namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// decimal text -> double with tinyobj's exact sequence of double operations
bool parse_decimal(const char *s, const char *end, double *out) {
  if (s >= end) return false;
  const char *c = s;
  bool neg = false;
  if (*c == '+' || *c == '-') {
    neg = (*c == '-');
    ++c;
  } else if (!is_digit(*c)) {
    return false;
  }
  double m = 0.0;
  int e10 = 0;
  int ndig = 0;
  while (c != end && is_digit(*c)) {
    m *= 10;
    m += (int)(*c - '0');
    ++c;
    ++ndig;
  }
  if (ndig == 0) return false;
  bool done = (c == end);
  if (!done && *c == '.') {
    static const double neg_pow[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
    ++c;
    int k = 1;
    while (c != end && is_digit(*c)) {
      m += (int)(*c - '0') * (k < 8 ? neg_pow[k] : std::pow(10.0, -k));
      ++k;
      ++c;
    }
    done = (c == end);
  } else if (!done && !(*c == 'e' || *c == 'E')) {
    done = true;
  }
  if (!done && (*c == 'e' || *c == 'E')) {
    ++c;
    bool eneg = false;
    if (c != end && (*c == '+' || *c == '-')) {
      eneg = (*c == '-');
      ++c;
    } else if (!is_digit(*c)) {
      return false; // bare 'e'
    }
    int nd = 0;
    while (c != end && is_digit(*c)) {
      e10 = e10 * 10 + (int)(*c - '0');
      ++c;
      ++nd;
    }
    if (eneg) e10 = -e10;
    if (nd == 0) return false;
  }
  const double mag = e10 ? std::ldexp(m * std::pow(5.0, e10), e10) : m;
  *out = (neg ? -1 : 1) * mag;
  return true;
}

struct Cursor {
  const char *p;
  void skip_blank() { p += std::strspn(p, " \t"); }
  // next whitespace-delimited field as float (default when missing / malformed)
  float number(double dflt = 0.0) {
    skip_blank();
    const char *end = p + std::strcspn(p, " \t\r");
    double v = dflt;
    parse_decimal(p, end, &v);
    p = end;
    return (float)v;
  }
  std::string word() { // sscanf("%s")-like
    p += std::strspn(p, " \t\r\n");
    size_t n = std::strcspn(p, " \t\r\n");
    std::string w(p, n);
    p += n;
    return w;
  }
  bool at_end() const { return *p == '\0' || *p == '\r' || *p == '\n'; }
};

bool keyword(const char *tok, const char *kw) {
  const size_t n = std::strlen(kw);
  return std::strncmp(tok, kw, n) == 0 && is_blank(tok[n]);
}

// istream line reader accepting \n, \r\n and \r (tinyobj's safeGetline)
bool next_line(std::istream &in, std::string &line) {
  line.clear();
  if (in.peek() == EOF) return false;
  for (;;) {
    int c = in.get();
    if (c == '\n') return true;
    if (c == '\r') {
      if (in.peek() == '\n') in.get();
      return true;
    }
    if (c == EOF) return true;
    line.push_back((char)c);
  }
}

struct MtlEntry {
  std::string name;
  Material m;
};

// returns false + message on anything tinyobj would report as a warning
bool read_mtl(const std::string &path, std::vector<MtlEntry> &mats, std::map<std::string, int> &by_name,
              std::string &warn) {
  std::ifstream in(path.c_str());
  if (!in) {
    warn += "WARN: Material file [ " + path + " ] not found.\n";
    return false;
  }
  MtlEntry cur;
  cur.m.Ns = 1.f; // InitMaterial: shininess = 1
  bool has_d = false, has_tr = false;
  auto flush = [&]() {
    by_name.insert(std::make_pair(cur.name, (int)mats.size())); // first definition wins
    mats.push_back(cur);
  };
  std::string line;
  while (next_line(in, line)) {
    const size_t last = line.find_last_not_of(" \t");
    line = (last == std::string::npos) ? std::string() : line.substr(0, last + 1);
    if (line.empty()) continue;
    Cursor c{line.c_str()};
    c.skip_blank();
    const char *t = c.p;
    if (*t == '\0' || *t == '#') continue;
    if (keyword(t, "newmtl")) {
      if (!cur.name.empty()) flush();
      cur = MtlEntry();
      cur.m.Ns = 1.f;
      has_d = has_tr = false;
      c.p = t + 7;
      cur.name = c.word();
      continue;
    }
    auto rgb = [&](float *dst) {
      c.p = t + 2;
      dst[0] = c.number();
      dst[1] = c.number();
      dst[2] = c.number();
    };
    if (t[0] == 'K' && t[1] == 'a' && is_blank(t[2])) { rgb(cur.m.ka); continue; }
    if (t[0] == 'K' && t[1] == 'd' && is_blank(t[2])) { rgb(cur.m.kd); continue; }
    if (t[0] == 'K' && t[1] == 's' && is_blank(t[2])) { rgb(cur.m.ks); continue; }
    if (t[0] == 'K' && t[1] == 'e' && is_blank(t[2])) { rgb(cur.m.ke); continue; }
    if (t[0] == 'N' && t[1] == 's' && is_blank(t[2])) {
      c.p = t + 2;
      cur.m.Ns = c.number();
      continue;
    }
    if (t[0] == 'd' && is_blank(t[1])) {
      if (has_tr) warn += "WARN: Both `d` and `Tr` parameters defined for \"" + cur.name + "\".\n";
      has_d = true;
      continue;
    }
    if (t[0] == 'T' && t[1] == 'r' && is_blank(t[2])) {
      if (has_d) warn += "WARN: Both `d` and `Tr` parameters defined for \"" + cur.name + "\".\n";
      has_tr = true;
      continue;
    }
    // everything else (Ni, illum, Tf, map_*, PBR terms) does not reach the renderer
  }
  flush(); // the last (or the unnamed default) material is always appended
  return true;
}

struct Corner {
  int v = -1, vt = -1, vn = -1;
};

int fix_index(int idx, int n) { return idx > 0 ? idx - 1 : (idx == 0 ? 0 : n + idx); }

Corner read_corner(Cursor &c, int nv, int nvn, int nvt) {
  Corner k;
  k.v = fix_index(std::atoi(c.p), nv);
  c.p += std::strcspn(c.p, "/ \t\r");
  if (*c.p != '/') return k;
  ++c.p;
  if (*c.p == '/') { // v//n
    ++c.p;
    k.vn = fix_index(std::atoi(c.p), nvn);
    c.p += std::strcspn(c.p, "/ \t\r");
    return k;
  }
  k.vt = fix_index(std::atoi(c.p), nvt);
  c.p += std::strcspn(c.p, "/ \t\r");
  if (*c.p != '/') return k;
  ++c.p;
  k.vn = fix_index(std::atoi(c.p), nvn);
  c.p += std::strcspn(c.p, "/ \t\r");
  return k;
}

struct Shape {
  std::string name;
  std::vector<Corner> corners; // 3 per triangle
  std::vector<int> material_ids; // per triangle
};

// tinyobj's exportFaceGroupToShape with triangulate = true
bool move_faces(Shape &shape, std::vector<std::vector<Corner>> &pending, int material,
                const std::string &name) {
  if (pending.empty()) return false;
  for (const auto &poly : pending) {
    // a polygon with < 3 corners contributes nothing (tinyobj reads face[1] regardless;
    // such input is malformed and rejected later by the 3-per-face check anyway)
    for (size_t k = 2; k < poly.size(); k++) {
      shape.corners.push_back(poly[0]);
      shape.corners.push_back(poly[k - 1]);
      shape.corners.push_back(poly[k]);
      shape.material_ids.push_back(material);
    }
  }
  shape.name = name;
  return true;
}

} // namespace

int load_obj(esc_scene &scene, const std::string &path) {
  std::ifstream in(path.c_str());
  if (!in) {
    set_error("TinyOBJ Error loading " + path + " error: Cannot open file [" + path + "]");
    return ESC_ERR_IO;
  }
  const std::string base_dir = path.substr(0, path.rfind('/') + 1); // sceneloader.cpp:20

  std::vector<float> v, vn;
  int n_vt = 0;
  std::vector<MtlEntry> materials;
  std::map<std::string, int> by_name;
  std::string warn;

  std::vector<Shape> shapes;
  Shape shape;
  std::vector<std::vector<Corner>> pending;
  std::string name;
  int material = -1;

  std::string line;
  while (next_line(in, line)) {
    if (line.empty()) continue;
    Cursor c{line.c_str()};
    c.skip_blank();
    const char *t = c.p;
    if (*t == '\0' || *t == '#') continue;
    if (t[0] == 'v' && is_blank(t[1])) {
      c.p = t + 2;
      for (int i = 0; i < 3; i++) v.push_back(c.number());
      continue;
    }
    if (t[0] == 'v' && t[1] == 'n' && is_blank(t[2])) {
      c.p = t + 3;
      for (int i = 0; i < 3; i++) vn.push_back(c.number());
      continue;
    }
    if (t[0] == 'v' && t[1] == 't' && is_blank(t[2])) {
      ++n_vt;
      continue;
    }
    if (t[0] == 'f' && is_blank(t[1])) {
      c.p = t + 2;
      c.skip_blank();
      std::vector<Corner> poly;
      while (!c.at_end()) {
        poly.push_back(read_corner(c, (int)v.size() / 3, (int)vn.size() / 3, n_vt));
        c.p += std::strspn(c.p, " \t\r");
      }
      pending.push_back(std::move(poly));
      continue;
    }
    if (keyword(t, "usemtl")) {
      c.p = t + 7;
      const std::string mname = c.word();
      auto it = by_name.find(mname);
      const int id = (it == by_name.end()) ? -1 : it->second;
      if (id != material) {
        move_faces(shape, pending, material, name); // shape stays open
        pending.clear();
        material = id;
      }
      continue;
    }
    if (keyword(t, "mtllib")) {
      std::stringstream ss(std::string(t + 7));
      std::string fn;
      bool any = false, found = false;
      while (std::getline(ss, fn, ' ')) {
        any = true;
        if (read_mtl(base_dir + fn, materials, by_name, warn)) {
          found = true;
          break;
        }
      }
      if (!any) warn += "WARN: Looks like empty filename for mtllib.\n";
      else if (!found) warn += "WARN: Failed to load material file(s).\n";
      continue;
    }
    if ((t[0] == 'g' || t[0] == 'o') && is_blank(t[1])) {
      if (move_faces(shape, pending, material, name)) shapes.push_back(shape);
      shape = Shape();
      pending.clear();
      c.p = t + 1;
      if (t[0] == 'g') {
        // first name after 'g' (tinyobj splits the whole line; names[0] is "g" itself)
        c.skip_blank();
        name = c.at_end() ? std::string() : c.word();
      } else {
        name = c.word();
      }
      continue;
    }
    // other statements (s, t, l, ...) do not reach the renderer
  }
  if (move_faces(shape, pending, material, name) || !shape.corners.empty()) shapes.push_back(shape);

  if (!warn.empty()) { // sceneloader.cpp:27-30: any tinyobj message is fatal
    set_error("TinyOBJ Error loading " + path + " error: " + warn);
    return ESC_ERR_PARSE;
  }

  // ---- sceneloader.cpp:34-104
  for (const Shape &sh : shapes) {
    if (sh.material_ids.empty()) continue;
    const int mid = sh.material_ids[0]; // :52
    if (mid < 0 || (size_t)mid >= materials.size()) {
      set_error("esc_scene_load_obj: shape '" + sh.name + "' in " + path +
                " has no material (the reference indexes materials[-1] here)");
      return ESC_ERR_PARSE;
    }
    Geometry g;
    g.name = sh.name;
    g.object_material = materials[(size_t)mid].m;
    {
      float m13[ESC_MATERIAL_FLOATS];
      material_to_floats(g.object_material, m13);
      material_from_floats(m13, g.object_material); // sets lightsource, :63-64
    }
    const size_t nf = sh.corners.size() / 3;
    for (size_t f = 0; f < nf; f++) {
      for (int i = 0; i < 3; i++) {
        const Corner &k = sh.corners[3 * f + i];
        if (k.v < 0 || (size_t)k.v * 3 + 2 >= v.size()) {
          set_error("esc_scene_load_obj: vertex index out of range in " + path);
          return ESC_ERR_PARSE;
        }
        const uint32_t vert_idx = (uint32_t)g.n_vertices();
        g.vertex.push_back(v[3 * k.v + 0]);
        g.vertex.push_back(v[3 * k.v + 1]);
        g.vertex.push_back(v[3 * k.v + 2]);
        if (k.vn != -1) { // :84-89
          if (k.vn < 0 || (size_t)k.vn * 3 + 2 >= vn.size()) {
            set_error("esc_scene_load_obj: normal index out of range in " + path);
            return ESC_ERR_PARSE;
          }
          const float *n = &vn[3 * (size_t)k.vn];
          float sum = 0; // vec.h:95-101 then :135-137
          for (int q = 0; q < 3; q++) sum += n[q] * n[q];
          const float len = std::sqrt(sum);
          g.normals.push_back(n[0] / len);
          g.normals.push_back(n[1] / len);
          g.normals.push_back(n[2] / len);
        }
        g.face_index.push_back(vert_idx);
      }
    }
    if (!g.normals.empty() && g.n_normals() != g.n_vertices()) {
      // the reference would read normals[face[k]] past the array here (sceneloader.cpp:84-89
      // pushes normals only for corners that have one); refuse instead of reading garbage
      set_error("esc_scene_load_obj: shape '" + sh.name + "' mixes corners with and without normals");
      return ESC_ERR_PARSE;
    }
    scene.geometry.push_back(std::move(g));
    if (scene.geometry.back().object_material.lightsource) // :102-104
      scene.light_sources.push_back(scene.geometry.size() - 1);
  }
  return ESC_OK;
}

} // namespace esc
