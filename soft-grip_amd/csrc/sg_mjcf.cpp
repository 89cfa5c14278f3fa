// sg_mjcf.cpp -- native MJCF subset compiler: XML file -> model blob (include/softgrip_model.h).
//
// The C-ABI counterpart of mujoco_py.load_model_from_path (reference environment/manenv.py:27,36) for callers that have no
// Python: sg_model_compile() in softgrip.h.  It implements the same MJCF subset as soft-grip_amd/mjcf.py (SURVEY.md App. A.1:
// nested <include>, <compiler angle/settotalmass>, <option>, <size>, nested <default class>, bodies with box / capsule / sphere /
// plane geoms, hinge / slide joints, sites, spatial tendons through sites, fixed tendons, cylinder actuators on tendons,
// accelerometer / gyro sensors, <composite type="box|ellipsoid|cylinder"> shells with their joint-fix, neighbour and tendon-fix
// equalities) in the same processing order, and writes the same tagged-array container, field for field; tests/test_mjcf.py
// compares the two compilers on every scene (integers and names equal, reals to 1e-12 relative: the summation orders of the
// mass-matrix inverse differ).  Everything MuJoCo-specific is restated from MuJoCo's documentation (parity with a
// real mjModel is unpinned).
#include <array>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/softgrip_model.h"
#include "sg_mjcf.h"

namespace {

constexpr double kMinVal = 1e-15;
constexpr double kPi = 3.141592653589793238462643383279502884;

struct Fail {
  std::string msg;
};
[[noreturn]] void fail(const std::string& m) { throw Fail{m}; }

// ------------------------------------------------------------------------------------------------ XML
using Attr = std::vector<std::pair<std::string, std::string>>;
struct Node {
  std::string tag;
  Attr attr;
  std::vector<std::unique_ptr<Node>> kids;
  const std::string* get(const char* k) const {
    for (auto& a : attr)
      if (a.first == k) return &a.second;
    return nullptr;
  }
  std::string gets(const char* k, const char* dflt) const {
    const std::string* v = get(k);
    return v ? *v : std::string(dflt);
  }
};

struct XmlParser {
  const std::string& s;
  size_t p = 0;
  std::string file;
  explicit XmlParser(const std::string& src, const std::string& f) : s(src), file(f) {}
  [[noreturn]] void err(const char* what) { fail(file + ": XML error at byte " + std::to_string(p) + ": " + what); }
  void ws() {
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) p++;
  }
  bool starts(const char* t) const { return s.compare(p, strlen(t), t) == 0; }
  void skip_misc() {  // whitespace, text, comments, declarations
    for (;;) {
      while (p < s.size() && s[p] != '<') p++;
      if (p >= s.size()) return;
      if (starts("<!--")) {
        size_t e = s.find("-->", p + 4);
        if (e == std::string::npos) err("unterminated comment");
        p = e + 3;
      } else if (starts("<?")) {
        size_t e = s.find("?>", p + 2);
        if (e == std::string::npos) err("unterminated declaration");
        p = e + 2;
      } else if (starts("<!")) {
        size_t e = s.find('>', p + 2);
        if (e == std::string::npos) err("unterminated <!...>");
        p = e + 1;
      } else {
        return;
      }
    }
  }
  static bool name_char(char c) { return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
  std::string name() {
    size_t b = p;
    while (p < s.size() && name_char(s[p])) p++;
    if (p == b) err("name expected");
    return s.substr(b, p - b);
  }
  static std::string unescape(const std::string& v) {
    if (v.find('&') == std::string::npos) return v;
    std::string o;
    for (size_t i = 0; i < v.size(); i++) {
      if (v[i] != '&') { o += v[i]; continue; }
      static const struct { const char* e; char c; } ents[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
      bool done = false;
      for (auto& en : ents)
        if (v.compare(i, strlen(en.e), en.e) == 0) { o += en.c; i += strlen(en.e) - 1; done = true; break; }
      if (!done) o += v[i];
    }
    return o;
  }
  std::unique_ptr<Node> element() {  // p at '<' of a start tag
    p++;
    auto n = std::make_unique<Node>();
    n->tag = name();
    for (;;) {
      ws();
      if (p >= s.size()) err("unterminated tag");
      if (s[p] == '/') {
        if (p + 1 >= s.size() || s[p + 1] != '>') err("'/>' expected");
        p += 2;
        return n;
      }
      if (s[p] == '>') { p++; break; }
      std::string k = name();
      ws();
      if (p >= s.size() || s[p] != '=') err("'=' expected");
      p++;
      ws();
      if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) err("quoted value expected");
      char q = s[p++];
      size_t e = s.find(q, p);
      if (e == std::string::npos) err("unterminated attribute value");
      n->attr.emplace_back(k, unescape(s.substr(p, e - p)));
      p = e + 1;
    }
    for (;;) {  // children until the end tag
      skip_misc();
      if (p >= s.size()) err("unterminated element");
      if (starts("</")) {
        p += 2;
        std::string t = name();
        if (t != n->tag) err("mismatched end tag");
        ws();
        if (p >= s.size() || s[p] != '>') err("'>' expected");
        p++;
        return n;
      }
      n->kids.push_back(element());
    }
  }
};

std::string dir_of(const std::string& path) {
  size_t k = path.find_last_of('/');
  return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

constexpr int kMaxIncludeDepth = 16;   // the reference nests three files deep (experiment -> gripper -> scene)
std::unique_ptr<Node> load_xml(const std::string& path, const std::string& base_dir, int depth);

void expand_includes(Node* parent, const std::string& base_dir, int depth) {
  std::vector<std::unique_ptr<Node>> out;
  for (auto& c : parent->kids) {
    if (c->tag == "include") {
      const std::string* f = c->get("file");
      if (!f) fail("<include> without file");
      if (depth >= kMaxIncludeDepth) fail("<include file=\"" + *f + "\"> nests deeper than " + std::to_string(kMaxIncludeDepth) + " files (an include cycle?)");
      auto inc = load_xml(base_dir + "/" + *f, base_dir, depth + 1);
      for (auto& k : inc->kids) out.push_back(std::move(k));
    } else {
      expand_includes(c.get(), base_dir, depth);
      out.push_back(std::move(c));
    }
  }
  parent->kids = std::move(out);
}

std::unique_ptr<Node> load_xml(const std::string& path, const std::string& base_dir, int depth) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) fail("cannot open " + path);
  std::string src;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, n);
  fclose(f);
  XmlParser P(src, path);
  P.skip_misc();
  if (P.p >= src.size()) fail(path + ": no root element");
  auto root = P.element();
  if (root->tag != "mujoco") fail(path + ": root element must be <mujoco>");
  expand_includes(root.get(), base_dir.empty() ? dir_of(path) : base_dir, depth);
  return root;
}

// ------------------------------------------------------------------------------------------------ small math
struct V3 { double v[3] = {0, 0, 0}; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct Q4 { double v[4] = {1, 0, 0, 0}; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
struct M3 { double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; };

std::vector<double> parse_vec(const std::string& s) {
  std::vector<double> o;
  const char* c = s.c_str();
  for (;;) {
    while (*c && isspace((unsigned char)*c)) c++;
    if (!*c) break;
    char* e;
    double v = strtod(c, &e);
    if (e == c) fail("not a number in \"" + s + "\"");
    o.push_back(v);
    c = e;
  }
  return o;
}
std::vector<double> vec_n(const std::string* s, size_t n, std::initializer_list<double> dflt) {
  if (!s) return std::vector<double>(dflt);
  auto v = parse_vec(*s);
  if (v.size() != n) fail("expected " + std::to_string(n) + " numbers, got \"" + *s + "\"");
  return v;
}
double norm3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
Q4 quat_normalize(const std::vector<double>& q) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  Q4 o;
  if (n < kMinVal) return o;
  for (int i = 0; i < 4; i++) o[i] = q[i] / n;
  return o;
}
Q4 quat_normalize(const Q4& q) { return quat_normalize(std::vector<double>(q.v, q.v + 4)); }
Q4 quat_mul(const Q4& a, const Q4& b) {
  Q4 o;
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  return o;
}
M3 quat_to_mat(const Q4& q) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  M3 R;
  R.m[0][0] = w * w + x * x - y * y - z * z; R.m[0][1] = 2 * (x * y - w * z); R.m[0][2] = 2 * (x * z + w * y);
  R.m[1][0] = 2 * (x * y + w * z); R.m[1][1] = w * w - x * x + y * y - z * z; R.m[1][2] = 2 * (y * z - w * x);
  R.m[2][0] = 2 * (x * z - w * y); R.m[2][1] = 2 * (y * z + w * x); R.m[2][2] = w * w - x * x - y * y + z * z;
  return R;
}
V3 mul(const M3& R, const V3& a) {
  V3 o;
  for (int i = 0; i < 3; i++) o[i] = R.m[i][0] * a[0] + R.m[i][1] * a[1] + R.m[i][2] * a[2];
  return o;
}
V3 cross(const V3& a, const V3& b) {
  V3 o;
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  return o;
}
Q4 quat_z2vec(const V3& vin) {  // minimal rotation taking +z to vec (mjcf.py quat_z2vec)
  const double n = norm3(vin.v);
  V3 vec; for (int i = 0; i < 3; i++) vec[i] = vin[i] / n;
  V3 z; z[2] = 1.0;
  V3 axis = cross(z, vec);
  const double s = norm3(axis.v);
  if (s < 1e-10) { axis[0] = 1; axis[1] = 0; axis[2] = 0; }
  else for (int i = 0; i < 3; i++) axis[i] /= s;
  const double ang = std::atan2(s, vec[2]);
  Q4 q;
  q[0] = std::cos(ang / 2);
  for (int i = 0; i < 3; i++) q[1 + i] = axis[i] * std::sin(ang / 2);
  return q;
}

// ------------------------------------------------------------------------------------------------ defaults
using AttrMap = std::map<std::string, std::string>;
struct Defaults {
  std::map<std::string, std::map<std::string, AttrMap>> classes;
  Defaults() { classes["main"]; }
  void read(const Node& elem, const std::string& cls, const std::string* parent) {
    if (!classes.count(cls)) {
      classes[cls];
      if (parent) classes[cls] = classes[*parent];  // inherit a copy of the parent's settings
    }
    for (auto& c : elem.kids)  // own settings first, nested classes (which inherit them) second
      if (c->tag != "default")
        for (auto& a : c->attr) classes[cls][c->tag][a.first] = a.second;
    for (auto& c : elem.kids)
      if (c->tag == "default") {
        const std::string* cn = c->get("class");
        if (!cn) fail("nested <default> without class");
        read(*c, *cn, &cls);
      }
  }
  AttrMap resolve(const char* tag, const Attr& own, const std::string* childclass) const {
    std::string cls = childclass ? *childclass : "main";
    for (auto& a : own)
      if (a.first == "class") cls = a.second;
    auto it = classes.find(cls);
    if (it == classes.end()) fail("unknown default class '" + cls + "'");
    AttrMap out;
    auto jt = it->second.find(tag);
    if (jt != it->second.end()) out = jt->second;
    for (auto& a : own) out[a.first] = a.second;
    return out;
  }
};
const std::string* mget(const AttrMap& m, const char* k) {
  auto it = m.find(k);
  return it == m.end() ? nullptr : &it->second;
}
double mnum(const AttrMap& m, const char* k, double dflt) {
  const std::string* s = mget(m, k);
  if (!s) return dflt;
  auto v = parse_vec(*s);
  if (v.empty()) fail(std::string("empty number for ") + k);
  return v[0];
}
int mint(const AttrMap& m, const char* k, int dflt) {
  const std::string* s = mget(m, k);
  return s ? (int)strtol(s->c_str(), nullptr, 10) : dflt;
}

// ------------------------------------------------------------------------------------------------ spec objects
struct Geom {
  std::string name;
  int type = SG_GEOM_SPHERE;
  double size[3] = {0, 0, 0};
  V3 pos; Q4 quat;
  bool has_mass = false;
  double mass = 0, density = 1000;
  int contype = 1, conaffinity = 1, condim = 3, priority = 0;
  double friction[3] = {1, 0.005, 0.0001}, solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2}, solmix = 1, margin = 0, gap = 0;
};
struct Joint {
  std::string name;
  int type = SG_JNT_HINGE;
  V3 pos, axis;
  bool limited = false;
  double range[2] = {0, 0}, stiffness = 0, damping = 0, armature = 0, margin = 0, ref = 0, springref = 0;
  double solref[2] = {0.02, 1}, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
};
struct Site { std::string name; V3 pos; Q4 quat; };
struct Body {
  std::string name; V3 pos; Q4 quat; int parent = -1;
  std::vector<Geom> geoms; std::vector<Joint> joints; std::vector<Site> sites;
};
struct Tendon {
  std::string name; bool spatial = false;
  std::vector<std::string> sites;                        // spatial
  std::vector<std::pair<std::string, double>> joints;    // fixed: (joint, coef)
  double stiffness = 0, damping = 0;
};
struct Equality {
  int type = SG_EQ_JOINT; std::string name1, name2; bool has2 = false;
  double solref[2], solimp[5], data[5] = {0, 0, 0, 0, 0};
};
struct Actuator { std::string tendon; double timeconst, gain, bias[3], gear; };
struct Sensor { int type; std::string site, name; };

void parse_solimp(const std::string& s, double* out) {
  auto v = parse_vec(s);
  if (v.size() > 5) fail("solimp has more than 5 numbers");
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
}

struct Compiler {
  std::unique_ptr<Node> root;
  bool composite_neighbors;
  Defaults defaults;
  std::vector<Body> bodies;
  std::vector<Tendon> tendons;
  std::vector<Equality> equalities;
  std::vector<Actuator> actuators;
  std::vector<Sensor> sensors;
  double settotalmass = -1;
  double timestep = 0.002, gravity[3] = {0, 0, -9.81}, tolerance = 1e-8, impratio = 1;
  int iterations = 100, nconmax = -1, njmax = -1;
  std::string solver = "Newton", cone = "pyramidal";

  Geom make_geom(const AttrMap& at, const std::string& name) {
    Geom g;
    g.name = name;
    std::string t = mget(at, "type") ? *mget(at, "type") : "sphere";
    if (t == "plane") g.type = SG_GEOM_PLANE; else if (t == "sphere") g.type = SG_GEOM_SPHERE;
    else if (t == "capsule") g.type = SG_GEOM_CAPSULE; else if (t == "box") g.type = SG_GEOM_BOX;
    else fail("unsupported geom type '" + t + "'");
    if (const std::string* s = mget(at, "size")) {
      auto v = parse_vec(*s);
      if (v.size() > 3) fail("geom size has more than 3 numbers");
      for (size_t i = 0; i < v.size(); i++) g.size[i] = v[i];
    }
    if (const std::string* s = mget(at, "friction")) {
      auto v = parse_vec(*s);
      if (v.size() > 3) fail("geom friction has more than 3 numbers");
      for (size_t i = 0; i < v.size(); i++) g.friction[i] = v[i];
    }
    auto p = vec_n(mget(at, "pos"), 3, {0, 0, 0});
    for (int i = 0; i < 3; i++) g.pos[i] = p[i];
    g.quat = quat_normalize(vec_n(mget(at, "quat"), 4, {1, 0, 0, 0}));
    if (mget(at, "mass")) { g.has_mass = true; g.mass = mnum(at, "mass", 0); }
    g.density = mnum(at, "density", 1000.0);
    g.contype = mint(at, "contype", 1); g.conaffinity = mint(at, "conaffinity", 1);
    g.condim = mint(at, "condim", 3); g.priority = mint(at, "priority", 0);
    auto sr = vec_n(mget(at, "solref"), 2, {0.02, 1.0});
    g.solref[0] = sr[0]; g.solref[1] = sr[1];
    if (const std::string* s = mget(at, "solimp")) parse_solimp(*s, g.solimp);
    g.solmix = mnum(at, "solmix", 1.0); g.margin = mnum(at, "margin", 0.0); g.gap = mnum(at, "gap", 0.0);
    return g;
  }
  static bool to_bool(const std::string* s) {
    if (!s) return false;
    std::string t;
    for (char c : *s) if (!isspace((unsigned char)c)) t += (char)tolower((unsigned char)c);
    return t == "true";
  }
  Joint make_joint(const AttrMap& at, const std::string& name) {
    Joint j;
    j.name = name;
    std::string t = mget(at, "type") ? *mget(at, "type") : "hinge";
    if (t == "hinge") j.type = SG_JNT_HINGE; else if (t == "slide") j.type = SG_JNT_SLIDE;
    else fail("unsupported joint type '" + t + "'");
    auto ax = vec_n(mget(at, "axis"), 3, {0, 0, 1});
    const double n = norm3(ax.data());
    for (int i = 0; i < 3; i++) j.axis[i] = ax[i] / n;
    auto p = vec_n(mget(at, "pos"), 3, {0, 0, 0});
    for (int i = 0; i < 3; i++) j.pos[i] = p[i];
    j.limited = to_bool(mget(at, "limited"));
    auto r = vec_n(mget(at, "range"), 2, {0, 0});
    j.range[0] = r[0]; j.range[1] = r[1];
    j.stiffness = mnum(at, "stiffness", 0); j.damping = mnum(at, "damping", 0); j.armature = mnum(at, "armature", 0);
    j.margin = mnum(at, "margin", 0); j.ref = mnum(at, "ref", 0); j.springref = mnum(at, "springref", 0);
    auto sr = vec_n(mget(at, "solreflimit"), 2, {0.02, 1.0});
    j.solref[0] = sr[0]; j.solref[1] = sr[1];
    if (const std::string* s = mget(at, "solimplimit")) parse_solimp(*s, j.solimp);
    return j;
  }

  void body_children(const Node& elem, int bid, const std::string* childclass) {
    for (auto& cp : elem.kids) {
      const Node& c = *cp;
      if (c.tag == "geom") {
        bodies[bid].geoms.push_back(make_geom(defaults.resolve("geom", c.attr, childclass), c.gets("name", "")));
      } else if (c.tag == "joint") {
        bodies[bid].joints.push_back(make_joint(defaults.resolve("joint", c.attr, childclass), c.gets("name", "")));
      } else if (c.tag == "site") {
        AttrMap at = defaults.resolve("site", c.attr, childclass);
        Site s;
        s.name = c.gets("name", "");
        auto p = vec_n(mget(at, "pos"), 3, {0, 0, 0});
        for (int i = 0; i < 3; i++) s.pos[i] = p[i];
        s.quat = quat_normalize(vec_n(mget(at, "quat"), 4, {1, 0, 0, 0}));
        bodies[bid].sites.push_back(s);
      } else if (c.tag == "body") {
        Body nb;
        nb.name = c.gets("name", "");
        auto p = vec_n(c.get("pos"), 3, {0, 0, 0});
        for (int i = 0; i < 3; i++) nb.pos[i] = p[i];
        nb.quat = quat_normalize(vec_n(c.get("quat"), 4, {1, 0, 0, 0}));
        nb.parent = bid;
        bodies.push_back(nb);
        const int id = (int)bodies.size() - 1;
        const std::string* cc = c.get("childclass");
        body_children(c, id, cc ? cc : childclass);
      } else if (c.tag == "composite") {
        composite(c, bid, childclass);
      } else if (c.tag == "light" || c.tag == "camera") {
      } else if (c.tag == "inertial") {
        fail("<inertial> is not supported (inertiafromgeom only)");
      } else if (c.tag == "freejoint") {
        // SURVEY 8(f) rank 4 (reference data/gripper/soft_experiments_softball.xml:8), as mjcf.py: 7 positions (world position +
        // quaternion), 6 dofs; no spring, damper, armature or limit
        if (bodies[bid].parent != 0 || !bodies[bid].joints.empty()) fail("a free joint must be the only joint of a child of the world body");
        Joint j;
        j.name = c.gets("name", "");
        j.type = SG_JNT_FREE;
        j.pos = V3(); j.axis = V3(); j.axis[2] = 1;
        bodies[bid].joints.push_back(j);
      } else {
        fail("unsupported worldbody element <" + c.tag + ">");
      }
    }
  }

  // SURVEY.md App. A.2
  void composite(const Node& elem, int bid, const std::string* childclass) {
    const std::string ctype = elem.gets("type", "");
    if (ctype != "box" && ctype != "ellipsoid" && ctype != "cylinder") fail("unsupported composite type '" + ctype + "'");
    const std::string prefix = elem.gets("prefix", "");
    const std::string* cs = elem.get("count");
    if (!cs) fail("composite without count");
    auto cv = parse_vec(*cs);
    // range-checked before the cast (a huge double -> int is undefined) and before the O(count^3) grid walk: the kernels take <= 256
    // shell elements (sg_model_create), which no axis beyond 64 can stay under
    if (cv.size() != 3) fail("box/ellipsoid composites need a 3-D count");
    for (double c : cv)
      if (!(c >= 2 && c <= 64) || c != std::floor(c)) fail("composite count must be whole numbers in [2, 64] per axis, got \"" + *cs + "\"");
    const int count[3] = {(int)cv[0], (int)cv[1], (int)cv[2]};
    {
      const long inner = (long)(count[0] - 2) * (count[1] - 2) * (count[2] - 2);
      const long shell = (long)count[0] * count[1] * count[2] - (inner > 0 ? inner : 0);
      if (shell > 256) fail("composite with " + std::to_string(shell) + " shell elements: at most 256 are supported");
    }
    const std::string* sp = elem.get("spacing");
    if (!sp) fail("composite without spacing");
    const double spacing = parse_vec(*sp)[0];
    AttrMap gattr = defaults.resolve("geom", Attr(), childclass), jattr = defaults.resolve("joint", Attr(), childclass), tattr;
    double eqj_solref[2] = {0.02, 1}, eqj_solimp[5] = {0.9, 0.95, 0.001, 0.5, 2}, eqt_solref[2] = {0.02, 1}, eqt_solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
    for (auto& cp : elem.kids) {
      const Node& c = *cp;
      if (c.tag == "geom") {
        for (auto& a : c.attr) gattr[a.first] = a.second;
      } else if (c.tag == "joint" || c.tag == "tendon") {
        const bool isj = c.tag == "joint";
        if (isj && c.gets("kind", "main") != "main") fail("only <joint kind='main'> is supported in composites");
        for (auto& a : c.attr) {
          if (a.first == "solreffix") {
            auto v = parse_vec(a.second);
            if (v.size() != 2) fail("solreffix needs 2 numbers");
            double* d = isj ? eqj_solref : eqt_solref;
            d[0] = v[0]; d[1] = v[1];
          } else if (a.first == "solimpfix") {
            parse_solimp(a.second, isj ? eqj_solimp : eqt_solimp);
          } else if (a.first != "kind") {
            (isj ? jattr : tattr)[a.first] = a.second;
          }
        }
      } else if (c.tag == "skin") {  // render-only
      } else {
        fail("unsupported composite child <" + c.tag + ">");
      }
    }
    Geom gc = make_geom(gattr, prefix + "Gcenter");
    gc.type = SG_GEOM_SPHERE;
    gc.pos = V3();
    gc.size[0] = gc.size[0] * 2; gc.size[1] = 0; gc.size[2] = 0;
    bodies[bid].geoms.push_back(gc);

    double half[3];
    for (int k = 0; k < 3; k++) half[k] = 0.5 * spacing * (count[k] - 1);
    Tendon ten;
    ten.name = prefix + "T";
    ten.stiffness = mnum(tattr, "stiffness", 0); ten.damping = mnum(tattr, "damping", 0);
    const size_t ten_index = tendons.size();
    tendons.push_back(ten);
    auto on_shell = [&](const int* q) { return q[0] == 0 || q[0] == count[0] - 1 || q[1] == 0 || q[1] == count[1] - 1 || q[2] == 0 || q[2] == count[2] - 1; };
    for (int ix = 0; ix < count[0]; ix++)
      for (int iy = 0; iy < count[1]; iy++)
        for (int iz = 0; iz < count[2]; iz++) {
          const int idx[3] = {ix, iy, iz};
          if (!on_shell(idx)) continue;
          V3 p;
          for (int k = 0; k < 3; k++) p[k] = 2.0 * idx[k] / (count[k] - 1) - 1;
          if (ctype == "box") {
            for (int k = 0; k < 3; k++) p[k] = p[k] * half[k];
          } else if (ctype == "ellipsoid") {
            const double n = norm3(p.v);
            for (int k = 0; k < 3; k++) p[k] = p[k] / n * half[k];
          } else {  // cylinder
            const double l0 = std::fmax(std::fabs(p[0]), std::fabs(p[1])), n2 = std::sqrt(p[0] * p[0] + p[1] * p[1]);
            V3 q;
            if (n2 >= kMinVal) { q[0] = p[0] / n2 * half[0] * l0; q[1] = p[1] / n2 * half[1] * l0; }  // else: an element on the axis (odd counts) stays there
            q[2] = p[2] * half[2];
            p = q;
          }
          char tag[64];
          snprintf(tag, sizeof tag, "%d_%d_%d", ix, iy, iz);
          Body b;
          b.name = prefix + "B" + tag; b.pos = p; b.quat = quat_z2vec(p); b.parent = bid;
          Geom g = make_geom(gattr, prefix + "G" + tag);
          if (g.type == SG_GEOM_CAPSULE) {
            g.pos = V3(); g.pos[2] = -(g.size[0] + g.size[1]);
          } else {
            g.type = SG_GEOM_SPHERE;
            g.pos = V3(); g.pos[2] = -g.size[0];
          }
          b.geoms.push_back(g);
          AttrMap ja = jattr;
          ja["type"] = "slide"; ja["pos"] = "0 0 0"; ja["axis"] = "0 0 1";
          Joint j = make_joint(ja, prefix + "J" + tag);
          b.joints.push_back(j);
          bodies.push_back(b);
          Equality e;
          e.type = SG_EQ_JOINT; e.name1 = j.name;
          memcpy(e.solref, eqj_solref, sizeof e.solref); memcpy(e.solimp, eqj_solimp, sizeof e.solimp);
          equalities.push_back(e);
          tendons[ten_index].joints.emplace_back(j.name, 1.0);
          if (composite_neighbors) {
            // "each joint is equality-constrained to remain equal to its neighbor joints" (MuJoCo 2.x composite documentation):
            // one two-joint equality towards the next shell element along +x, +y, +z, right after the element's own fix row
            for (int d = 0; d < 3; d++) {
              int q[3] = {ix, iy, iz};
              q[d] = q[d] + 1 < count[d] - 1 ? q[d] + 1 : count[d] - 1;
              if ((q[0] == ix && q[1] == iy && q[2] == iz) || !on_shell(q)) continue;
              Equality n;
              n.type = SG_EQ_JOINT; n.name1 = j.name; n.has2 = true;
              char t2[64];
              snprintf(t2, sizeof t2, "%d_%d_%d", q[0], q[1], q[2]);
              n.name2 = prefix + "J" + t2;
              memcpy(n.solref, eqj_solref, sizeof n.solref); memcpy(n.solimp, eqj_solimp, sizeof n.solimp);
              n.data[1] = 1.0;
              equalities.push_back(n);
            }
          }
        }
    Equality et;
    et.type = SG_EQ_TENDON; et.name1 = tendons[ten_index].name;
    memcpy(et.solref, eqt_solref, sizeof et.solref); memcpy(et.solimp, eqt_solimp, sizeof et.solimp);
    equalities.push_back(et);
  }

  void tendon_section(const Node& elem) {
    for (auto& tp : elem.kids) {
      const Node& t = *tp;
      AttrMap at = defaults.resolve("tendon", t.attr, nullptr);
      Tendon ten;
      ten.name = t.gets("name", "");
      ten.stiffness = mnum(at, "stiffness", 0); ten.damping = mnum(at, "damping", 0);
      if (t.tag == "spatial") {
        ten.spatial = true;
        for (auto& w : t.kids) {
          if (w->tag != "site") fail("only site wraps are supported in spatial tendons");
          const std::string* s = w->get("site");
          if (!s) fail("<site> wrap without site");
          ten.sites.push_back(*s);
        }
        if (ten.sites.size() < 2) fail("spatial tendon needs >= 2 sites");
      } else if (t.tag == "fixed") {
        for (auto& w : t.kids) {
          const std::string *jn = w->get("joint"), *cf = w->get("coef");
          if (!jn || !cf) fail("fixed tendon entries need joint and coef");
          ten.joints.emplace_back(*jn, parse_vec(*cf)[0]);
        }
      } else {
        fail("unsupported tendon <" + t.tag + ">");
      }
      tendons.push_back(ten);
    }
  }

  void run(const std::string& path, bool neighbors) {
    composite_neighbors = neighbors;
    root = load_xml(path, "", 0);
    Body world;
    world.name = "world";
    bodies.push_back(world);
    auto each = [&](const char* tag, auto fn) {
      for (auto& c : root->kids)
        if (c->tag == tag) fn(*c);
    };
    each("compiler", [&](const Node& e) {
      if (e.gets("angle", "degree") != "radian") fail("only angle='radian' is supported");
      if (const std::string* s = e.get("settotalmass")) settotalmass = parse_vec(*s)[0];
    });
    each("option", [&](const Node& e) {
      if (const std::string* s = e.get("timestep")) timestep = parse_vec(*s)[0];
      if (const std::string* s = e.get("gravity")) { auto v = vec_n(s, 3, {}); for (int i = 0; i < 3; i++) gravity[i] = v[i]; }
      if (const std::string* s = e.get("iterations")) iterations = (int)strtol(s->c_str(), nullptr, 10);
      if (const std::string* s = e.get("tolerance")) tolerance = parse_vec(*s)[0];
      if (const std::string* s = e.get("impratio")) impratio = parse_vec(*s)[0];
      if (const std::string* s = e.get("solver")) solver = *s;
      if (const std::string* s = e.get("cone")) cone = *s;
    });
    each("size", [&](const Node& e) {
      if (const std::string* s = e.get("nconmax")) nconmax = (int)strtol(s->c_str(), nullptr, 10);
      if (const std::string* s = e.get("njmax")) njmax = (int)strtol(s->c_str(), nullptr, 10);
    });
    each("default", [&](const Node& e) { defaults.read(e, "main", nullptr); });
    each("worldbody", [&](const Node& e) { body_children(e, 0, nullptr); });
    each("tendon", [&](const Node& e) { tendon_section(e); });
    each("actuator", [&](const Node& e) {
      for (auto& ap : e.kids) {
        if (ap->tag != "cylinder") fail("unsupported actuator <" + ap->tag + ">");
        AttrMap at = defaults.resolve("cylinder", ap->attr, nullptr);
        if (!mget(at, "tendon")) fail("cylinder actuators must act on a tendon");
        Actuator a;
        a.tendon = *mget(at, "tendon");
        double area = mnum(at, "area", 1.0);
        if (mget(at, "diameter")) { const double d = mnum(at, "diameter", 0); area = kPi * (d * d) / 4; }
        auto b = vec_n(mget(at, "bias"), 3, {0, 0, 0});
        a.timeconst = mnum(at, "timeconst", 1.0); a.gain = area;
        for (int i = 0; i < 3; i++) a.bias[i] = b[i];
        a.gear = mnum(at, "gear", 1.0);
        actuators.push_back(a);
      }
    });
    each("sensor", [&](const Node& e) {
      for (auto& sp : e.kids) {
        if (sp->tag != "accelerometer" && sp->tag != "gyro") fail("unsupported sensor <" + sp->tag + ">");
        const std::string* site = sp->get("site");
        if (!site) fail("sensor without site");
        sensors.push_back(Sensor{sp->tag == "accelerometer" ? SG_SENS_ACCELEROMETER : SG_SENS_GYRO, *site, sp->gets("name", "")});
      }
    });
    if (solver != "PGS" || cone != "elliptic") fail("only solver='PGS' cone='elliptic' (reference soft_scene.xml:13) is implemented");
  }
};

void geom_volume_inertia(const Geom& g, double* vol, double* Id) {  // per-unit-mass principal inertia in the geom frame
  if (g.type == SG_GEOM_SPHERE) {
    const double r = g.size[0];
    *vol = 4.0 / 3.0 * kPi * (r * r * r);
    Id[0] = Id[1] = Id[2] = 0.4 * r * r;
  } else if (g.type == SG_GEOM_BOX) {
    const double sx = g.size[0], sy = g.size[1], sz = g.size[2];
    *vol = 8 * sx * sy * sz;
    Id[0] = (sy * sy + sz * sz) / 3.0; Id[1] = (sx * sx + sz * sz) / 3.0; Id[2] = (sx * sx + sy * sy) / 3.0;
  } else if (g.type == SG_GEOM_CAPSULE) {
    const double r = g.size[0], h = 2 * g.size[1];
    *vol = kPi * (r * r * h + 4.0 / 3.0 * (r * r * r));
    const double ms = 4 * r / (4 * r + 3 * h), mc = 1.0 - ms;
    const double ixy = mc * (3 * r * r + h * h) / 12 + 2 * ms * r * r / 5 + ms * h * (3 * r + 2 * h) / 8, iz = mc * r * r / 2 + 2 * ms * r * r / 5;
    Id[0] = Id[1] = ixy; Id[2] = iz;
  } else {
    *vol = 0; Id[0] = Id[1] = Id[2] = 0;
  }
}

// ------------------------------------------------------------------------------------------------ flat model + blob
struct Flat {
  std::vector<std::pair<std::string, std::vector<double>>> f64;
  std::vector<std::pair<std::string, std::vector<int32_t>>> i32;
  std::vector<double>& F(const char* n) { f64.emplace_back(n, std::vector<double>()); return f64.back().second; }
  std::vector<int32_t>& I(const char* n) { i32.emplace_back(n, std::vector<int32_t>()); return i32.back().second; }
};

void put_record(std::string& out, const std::string& name, uint32_t code, int64_t count, const void* data, size_t nbytes) {
  if (name.size() > 23) fail("field name too long: " + name);
  sg_blob_record r;
  memset(&r, 0, sizeof r);
  memcpy(r.name, name.data(), name.size());
  r.dtype = code; r.count = count;
  out.append((const char*)&r, sizeof r);
  out.append((const char*)data, nbytes);
  out.append((8 - nbytes % 8) % 8, '\0');
}

std::string join(const std::vector<std::string>& v) {
  std::string o;
  for (size_t i = 0; i < v.size(); i++) { if (i) o += '|'; o += v[i]; }
  return o;
}

std::string finalize(Compiler& C, bool implicit_tendon_damping) {
  std::vector<Body>& B = C.bodies;
  const int nbody = (int)B.size();
  std::vector<int> body_jntadr(nbody), body_jntnum(nbody), body_geomadr(nbody), body_geomnum(nbody), body_weldid(nbody, 0), body_parentid(nbody);
  struct JRef { int body; const Joint* j; };
  struct GRef { int body; const Geom* g; };
  struct SRef { int body; const Site* s; };
  std::vector<JRef> joints; std::vector<GRef> geoms; std::vector<SRef> sites;
  for (int i = 0; i < nbody; i++) {
    body_parentid[i] = B[i].parent > 0 ? B[i].parent : 0;
    body_jntadr[i] = B[i].joints.empty() ? -1 : (int)joints.size();
    body_jntnum[i] = (int)B[i].joints.size();
    body_geomadr[i] = B[i].geoms.empty() ? -1 : (int)geoms.size();
    body_geomnum[i] = (int)B[i].geoms.size();
    for (auto& j : B[i].joints) joints.push_back({i, &j});
    for (auto& g : B[i].geoms) geoms.push_back({i, &g});
    for (auto& s : B[i].sites) sites.push_back({i, &s});
  }
  const int nj = (int)joints.size(), ng = (int)geoms.size(), ns = (int)sites.size();
  for (int i = 1; i < nbody; i++) body_weldid[i] = B[i].joints.empty() ? body_weldid[B[i].parent] : i;
  // joint -> first position / first dof (a free joint: 7 / 6), dof -> joint
  std::vector<int> jnt_qposadr(nj), jnt_dofadr(nj), dof_jntid;
  int nq = 0;
  bool has_free = false;
  for (int k = 0; k < nj; k++) {
    const bool fr = joints[k].j->type == SG_JNT_FREE;
    has_free = has_free || fr;
    jnt_qposadr[k] = nq; jnt_dofadr[k] = (int)dof_jntid.size();
    nq += fr ? 7 : 1;
    for (int d = 0; d < (fr ? 6 : 1); d++) dof_jntid.push_back(k);
  }
  const int nvd = (int)dof_jntid.size();
  std::vector<int> dof_parentid(nvd, -1), last_dof(nbody, -1);
  for (int i = 1; i < nbody; i++) {
    int prev = last_dof[B[i].parent];
    for (int k = 0; k < body_jntnum[i]; k++) {
      const int jj = body_jntadr[i] + k;
      for (int d = jnt_dofadr[jj]; d < jnt_dofadr[jj] + (joints[jj].j->type == SG_JNT_FREE ? 6 : 1); d++) { dof_parentid[d] = prev; prev = d; }
    }
    last_dof[i] = prev;
  }

  // inertial properties from the geoms
  std::vector<double> mass(nbody, 0.0), ipos(3 * nbody, 0.0), imat(9 * nbody, 0.0);
  for (int i = 0; i < nbody; i++) {
    const auto& G = B[i].geoms;
    std::vector<double> gm(G.size());
    std::vector<std::array<double, 3>> gI(G.size());
    double msum = 0;
    for (size_t k = 0; k < G.size(); k++) {
      double vol, Id[3];
      geom_volume_inertia(G[k], &vol, Id);
      gm[k] = G[k].has_mass ? G[k].mass : G[k].density * vol;
      for (int a = 0; a < 3; a++) gI[k][a] = Id[a] * gm[k];
      msum += gm[k];
    }
    if (G.empty() || msum <= 0) continue;
    mass[i] = msum;
    double cp[3] = {0, 0, 0};
    for (size_t k = 0; k < G.size(); k++)
      for (int a = 0; a < 3; a++) cp[a] += gm[k] * G[k].pos[a];
    for (int a = 0; a < 3; a++) ipos[3 * i + a] = cp[a] / mass[i];
    for (size_t k = 0; k < G.size(); k++) {
      const M3 R = quat_to_mat(G[k].quat);
      double d[3];
      for (int a = 0; a < 3; a++) d[a] = G[k].pos[a] - ipos[3 * i + a];
      const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double rir = 0;  // (R diag(I) R')_ab
          for (int c = 0; c < 3; c++) rir += R.m[a][c] * gI[k][c] * R.m[b][c];
          imat[9 * i + 3 * a + b] += rir + gm[k] * ((a == b ? dd : 0.0) - d[a] * d[b]);
        }
    }
  }
  if (C.settotalmass > 0) {
    double tot = 0;
    for (int i = 1; i < nbody; i++) tot += mass[i];
    const double scale = C.settotalmass / std::fmax(kMinVal, tot);
    for (auto& v : mass) v *= scale;
    for (auto& v : imat) v *= scale;
  }
  for (int i = 1; i < nbody; i++)
    if (body_weldid[i] != 0 && body_jntnum[i] > 0 && mass[i] < kMinVal) fail("moving body " + std::to_string(i) + " (" + B[i].name + ") has no mass");

  // name -> index maps (unnamed objects cannot be referenced)
  std::map<std::string, int> jidx, sidx, tidx;
  for (int k = 0; k < nj; k++) if (!joints[k].j->name.empty()) jidx[joints[k].j->name] = k;
  for (int k = 0; k < ns; k++) if (!sites[k].s->name.empty()) sidx[sites[k].s->name] = k;
  auto need = [&](const std::map<std::string, int>& m, const std::string& n, const char* what) {
    auto it = m.find(n);
    if (it == m.end()) fail(std::string("unknown ") + what + " '" + n + "'");
    return it->second;
  };

  Flat M;
  // ---- fp64 fields in the container's order (mjcf.py Model._FIELDS_F64); filled below where they need kinematics ----
  std::vector<double> body_pos(3 * nbody), body_quat(4 * nbody);
  for (int i = 0; i < nbody; i++) {
    for (int a = 0; a < 3; a++) body_pos[3 * i + a] = B[i].pos[a];
    for (int a = 0; a < 4; a++) body_quat[4 * i + a] = B[i].quat[a];
  }
  std::vector<double> jnt_pos(3 * nj), jnt_axis(3 * nj), jnt_range(2 * nj), jnt_stiffness(nj), jnt_margin(nj), jnt_solref(2 * nj), jnt_solimp(5 * nj),
      qpos0(nq), qpos_spring(nq), dof_damping(nvd), dof_armature(nvd);
  std::vector<int> jnt_type(nj), jnt_bodyid(nj), jnt_limited(nj);
  for (int k = 0; k < nj; k++) {
    const Joint& j = *joints[k].j;
    jnt_type[k] = j.type; jnt_bodyid[k] = joints[k].body; jnt_limited[k] = j.limited ? 1 : 0;
    for (int a = 0; a < 3; a++) { jnt_pos[3 * k + a] = j.pos[a]; jnt_axis[3 * k + a] = j.axis[a]; }
    jnt_range[2 * k] = j.range[0]; jnt_range[2 * k + 1] = j.range[1];
    jnt_stiffness[k] = j.stiffness; jnt_margin[k] = j.margin;
    jnt_solref[2 * k] = j.solref[0]; jnt_solref[2 * k + 1] = j.solref[1];
    for (int a = 0; a < 5; a++) jnt_solimp[5 * k + a] = j.solimp[a];
    const int qa = jnt_qposadr[k], da = jnt_dofadr[k];
    if (j.type == SG_JNT_FREE) {   // the body's pose (a child of the world)
      const Body& bb = B[joints[k].body];
      for (int a = 0; a < 3; a++) qpos0[qa + a] = qpos_spring[qa + a] = bb.pos[a];
      for (int a = 0; a < 4; a++) qpos0[qa + 3 + a] = qpos_spring[qa + 3 + a] = bb.quat[a];
      for (int d = 0; d < 6; d++) { dof_damping[da + d] = j.damping; dof_armature[da + d] = j.armature; }
    } else {
      qpos0[qa] = j.ref; qpos_spring[qa] = j.springref; dof_damping[da] = j.damping; dof_armature[da] = j.armature;
    }
  }
  std::vector<double> geom_size(3 * ng), geom_pos(3 * ng), geom_quat(4 * ng), geom_friction(3 * ng), geom_solref(2 * ng), geom_solimp(5 * ng),
      geom_solmix(ng), geom_margin(ng), geom_gap(ng), geom_rbound(ng, 0.0);
  std::vector<int> geom_type(ng), geom_bodyid(ng), geom_contype(ng), geom_conaffinity(ng), geom_condim(ng), geom_priority(ng);
  for (int k = 0; k < ng; k++) {
    const Geom& g = *geoms[k].g;
    geom_type[k] = g.type; geom_bodyid[k] = geoms[k].body; geom_contype[k] = g.contype; geom_conaffinity[k] = g.conaffinity;
    geom_condim[k] = g.condim; geom_priority[k] = g.priority;
    for (int a = 0; a < 3; a++) { geom_size[3 * k + a] = g.size[a]; geom_pos[3 * k + a] = g.pos[a]; geom_friction[3 * k + a] = g.friction[a]; }
    for (int a = 0; a < 4; a++) geom_quat[4 * k + a] = g.quat[a];
    geom_solref[2 * k] = g.solref[0]; geom_solref[2 * k + 1] = g.solref[1];
    for (int a = 0; a < 5; a++) geom_solimp[5 * k + a] = g.solimp[a];
    geom_solmix[k] = g.solmix; geom_margin[k] = g.margin; geom_gap[k] = g.gap;
    if (g.type == SG_GEOM_SPHERE) geom_rbound[k] = g.size[0];
    else if (g.type == SG_GEOM_CAPSULE) geom_rbound[k] = g.size[0] + g.size[1];
    else if (g.type == SG_GEOM_BOX) geom_rbound[k] = norm3(g.size);
  }
  std::vector<double> site_pos(3 * ns), site_quat(4 * ns);
  std::vector<int> site_bodyid(ns);
  for (int k = 0; k < ns; k++) {
    site_bodyid[k] = sites[k].body;
    for (int a = 0; a < 3; a++) site_pos[3 * k + a] = sites[k].s->pos[a];
    for (int a = 0; a < 4; a++) site_quat[4 * k + a] = sites[k].s->quat[a];
  }
  const int nt = (int)C.tendons.size();
  std::vector<int> tendon_adr(nt), tendon_num(nt), wrap_type, wrap_objid;
  std::vector<double> wrap_prm, tendon_stiffness(nt), tendon_damping(nt);
  for (int t = 0; t < nt; t++) {
    const Tendon& T = C.tendons[t];
    tendon_adr[t] = (int)wrap_type.size();
    if (T.spatial)
      for (auto& s : T.sites) { wrap_type.push_back(SG_WRAP_SITE); wrap_objid.push_back(need(sidx, s, "site")); wrap_prm.push_back(0.0); }
    else
      for (auto& jc : T.joints) { wrap_type.push_back(SG_WRAP_JOINT); wrap_objid.push_back(need(jidx, jc.first, "joint")); wrap_prm.push_back(jc.second); }
    tendon_num[t] = (int)wrap_type.size() - tendon_adr[t];
    tendon_stiffness[t] = T.stiffness; tendon_damping[t] = T.damping;
    if (!T.name.empty()) tidx[T.name] = t;
  }
  const int ne = (int)C.equalities.size();
  std::vector<int> eq_type(ne), eq_obj1id(ne), eq_obj2id(ne);
  std::vector<double> eq_solref(2 * ne), eq_solimp(5 * ne), eq_data(5 * ne);
  for (int e = 0; e < ne; e++) {
    const Equality& E = C.equalities[e];
    eq_type[e] = E.type;
    eq_obj1id[e] = E.type == SG_EQ_JOINT ? need(jidx, E.name1, "joint") : need(tidx, E.name1, "tendon");
    eq_obj2id[e] = E.has2 ? need(jidx, E.name2, "joint") : -1;
    eq_solref[2 * e] = E.solref[0]; eq_solref[2 * e + 1] = E.solref[1];
    for (int a = 0; a < 5; a++) { eq_solimp[5 * e + a] = E.solimp[a]; eq_data[5 * e + a] = E.data[a]; }
  }
  const int nu = (int)C.actuators.size();
  std::vector<int> actuator_trnid(nu);
  std::vector<double> actuator_timeconst(nu), actuator_gain(nu), actuator_bias(3 * nu), actuator_gear(nu);
  for (int u = 0; u < nu; u++) {
    const Actuator& A = C.actuators[u];
    actuator_trnid[u] = need(tidx, A.tendon, "tendon");
    actuator_timeconst[u] = A.timeconst; actuator_gain[u] = A.gain; actuator_gear[u] = A.gear;
    for (int a = 0; a < 3; a++) actuator_bias[3 * u + a] = A.bias[a];
  }
  const int nsens = (int)C.sensors.size();
  std::vector<int> sensor_type(nsens), sensor_objid(nsens), sensor_adr(nsens);
  for (int k = 0; k < nsens; k++) {
    sensor_type[k] = C.sensors[k].type; sensor_objid[k] = need(sidx, C.sensors[k].site, "site"); sensor_adr[k] = 3 * k;
  }

  // ---- qpos0-dependent constants (MuJoCo's mj_setConst; mjcf.py Model._set_const) ----
  const int nv = nvd;
  std::vector<V3> xpos(nbody), xanchor(nv), xaxis(nv);   // anchors / axes by DOF
  std::vector<int> dof_rot(nv, 0);
  std::vector<Q4> xquat(nbody);
  std::vector<M3> xmat(nbody);
  xmat[0] = quat_to_mat(xquat[0]);
  for (int i = 1; i < nbody; i++) {
    const int p = body_parentid[i];
    M3 R = quat_to_mat(xquat[p]);
    V3 bp; for (int a = 0; a < 3; a++) bp[a] = body_pos[3 * i + a];
    V3 t = mul(R, bp), pos;
    for (int a = 0; a < 3; a++) pos[a] = xpos[p][a] + t[a];
    Q4 bq; for (int a = 0; a < 4; a++) bq[a] = body_quat[4 * i + a];
    Q4 quat = quat_mul(xquat[p], bq);
    for (int k = 0; k < body_jntnum[i]; k++) {
      const int jn = body_jntadr[i] + k, j = jnt_dofadr[jn];
      if (jnt_type[jn] == SG_JNT_FREE) {   // at qpos0 the free body sits where the XML puts it; dofs: world translations, body-axis rotations
        R = quat_to_mat(quat_normalize(quat));
        for (int c = 0; c < 3; c++) {
          xanchor[j + c] = pos; xanchor[j + 3 + c] = pos;
          V3 e; e[c] = 1; xaxis[j + c] = e;
          V3 u; for (int a = 0; a < 3; a++) u[a] = R.m[a][c];
          xaxis[j + 3 + c] = u;
          dof_rot[j + 3 + c] = 1;
        }
        continue;
      }
      R = quat_to_mat(quat);
      V3 jp, ja; for (int a = 0; a < 3; a++) { jp[a] = jnt_pos[3 * jn + a]; ja[a] = jnt_axis[3 * jn + a]; }
      V3 rj = mul(R, jp);
      for (int a = 0; a < 3; a++) xanchor[j][a] = pos[a] + rj[a];
      xaxis[j] = mul(R, ja);
      const double dq = 0.0;  // at qpos0
      if (jnt_type[jn] == SG_JNT_SLIDE) {
        for (int a = 0; a < 3; a++) pos[a] = pos[a] + xaxis[j][a] * dq;
      } else {
        dof_rot[j] = 1;
        Q4 ql; ql[0] = std::cos(dq / 2);
        for (int a = 0; a < 3; a++) ql[1 + a] = ja[a] * std::sin(dq / 2);
        quat = quat_mul(quat, ql);
        V3 r2 = mul(quat_to_mat(quat), jp);
        for (int a = 0; a < 3; a++) pos[a] = xanchor[j][a] - r2[a];
      }
    }
    xpos[i] = pos;
    xquat[i] = quat_normalize(quat);
  }
  for (int i = 0; i < nbody; i++) xmat[i] = quat_to_mat(xquat[i]);
  // point Jacobians: the dofs on the way to the root
  struct Jcol { int dof; V3 jp, jr; };
  auto jac_point = [&](int body, const V3& point) {
    std::vector<Jcol> cols;
    for (int b = body; b > 0; b = body_parentid[b])
      for (int k = 0; k < body_jntnum[b]; k++) {
        const int jn = body_jntadr[b] + k;
        for (int j = jnt_dofadr[jn]; j < jnt_dofadr[jn] + (jnt_type[jn] == SG_JNT_FREE ? 6 : 1); j++) {
          Jcol c; c.dof = j;
          if (!dof_rot[j]) { c.jp = xaxis[j]; }
          else {
            c.jr = xaxis[j];
            V3 d; for (int a = 0; a < 3; a++) d[a] = point[a] - xanchor[j][a];
            c.jp = cross(xaxis[j], d);
          }
          cols.push_back(c);
        }
      }
    return cols;
  };
  std::vector<double> Mq((size_t)nv * nv, 0.0);
  for (int j = 0; j < nv; j++) Mq[(size_t)j * nv + j] = dof_armature[j];
  std::vector<V3> com(nbody);
  for (int b = 1; b < nbody; b++) {
    V3 ip; for (int a = 0; a < 3; a++) ip[a] = ipos[3 * b + a];
    V3 t = mul(xmat[b], ip);
    for (int a = 0; a < 3; a++) com[b][a] = xpos[b][a] + t[a];
    if (mass[b] <= 0 || body_weldid[b] == 0) continue;
    M3 Iw;  // R I R'
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int d = 0; d < 3; d++)
          for (int e = 0; e < 3; e++) s += xmat[b].m[a][d] * imat[9 * b + 3 * d + e] * xmat[b].m[c][e];
        Iw.m[a][c] = s;
      }
    auto cols = jac_point(b, com[b]);
    for (auto& c1 : cols)
      for (auto& c2 : cols) {
        const V3 Ij = mul(Iw, c2.jr);
        Mq[(size_t)c1.dof * nv + c2.dof] += mass[b] * (c1.jp[0] * c2.jp[0] + c1.jp[1] * c2.jp[1] + c1.jp[2] * c2.jp[2]) +
                                            (c1.jr[0] * Ij[0] + c1.jr[1] * Ij[1] + c1.jr[2] * Ij[2]);
      }
  }
  // inverse of the symmetric positive definite mass matrix: Cholesky M = L L', M^-1 = L^-T L^-1
  std::vector<double> L(Mq), Minv((size_t)nv * nv, 0.0);
  for (int j = 0; j < nv; j++) {
    double d = L[(size_t)j * nv + j];
    for (int k = 0; k < j; k++) d -= L[(size_t)j * nv + k] * L[(size_t)j * nv + k];
    if (!(d > 0)) fail("mass matrix is not positive definite at dof " + std::to_string(j));
    d = std::sqrt(d);
    L[(size_t)j * nv + j] = d;
    for (int i = j + 1; i < nv; i++) {
      double s = L[(size_t)i * nv + j];
      for (int k = 0; k < j; k++) s -= L[(size_t)i * nv + k] * L[(size_t)j * nv + k];
      L[(size_t)i * nv + j] = s / d;
    }
  }
  {
    std::vector<double> Li((size_t)nv * nv, 0.0);  // L^-1 (lower triangular)
    for (int c = 0; c < nv; c++) {
      Li[(size_t)c * nv + c] = 1.0 / L[(size_t)c * nv + c];
      for (int i = c + 1; i < nv; i++) {
        double s = 0;
        for (int k = c; k < i; k++) s -= L[(size_t)i * nv + k] * Li[(size_t)k * nv + c];
        Li[(size_t)i * nv + c] = s / L[(size_t)i * nv + i];
      }
    }
    for (int a = 0; a < nv; a++)
      for (int b = a; b < nv; b++) {
        double s = 0;
        for (int k = b; k < nv; k++) s += Li[(size_t)k * nv + a] * Li[(size_t)k * nv + b];
        Minv[(size_t)a * nv + b] = Minv[(size_t)b * nv + a] = s;
      }
  }
  double tr = 0;
  for (int j = 0; j < nv; j++) tr += Mq[(size_t)j * nv + j];
  const double meaninertia = tr / (nv > 1 ? nv : 1);
  std::vector<double> dof_invweight0(nv), body_invweight0(2 * nbody, 0.0);
  for (int j = 0; j < nv; j++) dof_invweight0[j] = Minv[(size_t)j * nv + j];
  for (int k = 0; k < nj; k++)   // mj_setConst: one value for a free joint's three translations, one for its three rotations
    if (jnt_type[k] == SG_JNT_FREE)
      for (int h3 = 0; h3 < 2; h3++) {
        const int d0 = jnt_dofadr[k] + 3 * h3;
        const double av = (dof_invweight0[d0] + dof_invweight0[d0 + 1] + dof_invweight0[d0 + 2]) / 3;
        dof_invweight0[d0] = dof_invweight0[d0 + 1] = dof_invweight0[d0 + 2] = av;
      }
  for (int b = 1; b < nbody; b++) {
    if (body_weldid[b] == 0) continue;
    auto cols = jac_point(b, com[b]);
    double tt = 0, rr = 0;
    for (int a = 0; a < 3; a++) {  // diagonal entries of J M^-1 J' (translation rows, rotation rows)
      double st = 0, sr = 0;
      for (auto& c1 : cols)
        for (auto& c2 : cols) {
          const double mi = Minv[(size_t)c1.dof * nv + c2.dof];
          st += c1.jp[a] * mi * c2.jp[a];
          sr += c1.jr[a] * mi * c2.jr[a];
        }
      tt += st; rr += sr;
    }
    body_invweight0[2 * b] = tt / 3; body_invweight0[2 * b + 1] = rr / 3;
  }
  std::vector<double> tendon_length0(nt, 0.0), tendon_invweight0(nt, 0.0);
  {
    std::vector<V3> sx(ns);
    for (int k = 0; k < ns; k++) {
      V3 sp; for (int a = 0; a < 3; a++) sp[a] = site_pos[3 * k + a];
      V3 t = mul(xmat[site_bodyid[k]], sp);
      for (int a = 0; a < 3; a++) sx[k][a] = xpos[site_bodyid[k]][a] + t[a];
    }
    for (int t = 0; t < nt; t++) {
      std::vector<double> J(nv, 0.0);
      const int a0 = tendon_adr[t], n = tendon_num[t];
      if (wrap_type[a0] == SG_WRAP_JOINT) {
        for (int w = a0; w < a0 + n; w++) {   // (wrap_objid: a joint id)
          tendon_length0[t] += wrap_prm[w] * qpos0[jnt_qposadr[wrap_objid[w]]];
          J[jnt_dofadr[wrap_objid[w]]] = wrap_prm[w];
        }
      } else {
        for (int w = a0; w < a0 + n - 1; w++) {
          const int s0 = wrap_objid[w], s1 = wrap_objid[w + 1];
          double d[3];
          for (int a = 0; a < 3; a++) d[a] = sx[s1][a] - sx[s0][a];
          const double ln = norm3(d);
          tendon_length0[t] += ln;
          if (ln > kMinVal) {
            for (auto& c : jac_point(site_bodyid[s1], sx[s1])) J[c.dof] += (d[0] * c.jp[0] + d[1] * c.jp[1] + d[2] * c.jp[2]) / ln;
            for (auto& c : jac_point(site_bodyid[s0], sx[s0])) J[c.dof] -= (d[0] * c.jp[0] + d[1] * c.jp[1] + d[2] * c.jp[2]) / ln;
          }
        }
      }
      double s = 0;
      for (int a = 0; a < nv; a++) {
        if (J[a] == 0) continue;
        double r = 0;
        for (int b = 0; b < nv; b++) r += Minv[(size_t)a * nv + b] * J[b];
        s += J[a] * r;
      }
      tendon_invweight0[t] = s;
    }
  }

  // ---- the container ----
  std::string body;
  int nrec = 0;
  auto addF = [&](const char* n, const std::vector<double>& v) { put_record(body, n, SG_DT_F64, (int64_t)v.size(), v.data(), v.size() * 8); nrec++; };
  auto addI = [&](const char* n, const std::vector<int>& v) { put_record(body, n, SG_DT_I32, (int64_t)v.size(), v.data(), v.size() * 4); nrec++; };
  addF("opt_d", {C.timestep, C.gravity[0], C.gravity[1], C.gravity[2], C.tolerance, C.impratio, meaninertia});
  addI("opt_i", {C.iterations, C.nconmax, C.njmax, implicit_tendon_damping ? 1 : 0});
  addF("body_pos", body_pos); addF("body_quat", body_quat); addF("body_ipos", ipos); addF("body_imat", imat); addF("body_mass", mass);
  addF("body_invweight0", body_invweight0);
  addF("jnt_pos", jnt_pos); addF("jnt_axis", jnt_axis); addF("jnt_range", jnt_range); addF("jnt_stiffness", jnt_stiffness);
  addF("jnt_margin", jnt_margin); addF("jnt_solref", jnt_solref); addF("jnt_solimp", jnt_solimp);
  addF("qpos0", qpos0); addF("qpos_spring", qpos_spring); addF("dof_damping", dof_damping); addF("dof_armature", dof_armature);
  addF("dof_invweight0", dof_invweight0);
  addF("geom_size", geom_size); addF("geom_pos", geom_pos); addF("geom_quat", geom_quat); addF("geom_friction", geom_friction);
  addF("geom_solref", geom_solref); addF("geom_solimp", geom_solimp); addF("geom_solmix", geom_solmix);
  addF("geom_margin", geom_margin); addF("geom_gap", geom_gap); addF("geom_rbound", geom_rbound);
  addF("site_pos", site_pos); addF("site_quat", site_quat);
  addF("tendon_stiffness", tendon_stiffness); addF("tendon_damping", tendon_damping);
  addF("tendon_lengthspring", tendon_length0);  // springlength = -1: the length at qpos0
  addF("tendon_length0", tendon_length0); addF("tendon_invweight0", tendon_invweight0);
  addF("wrap_prm", wrap_prm); addF("eq_solref", eq_solref); addF("eq_solimp", eq_solimp); addF("eq_data", eq_data);
  addF("actuator_timeconst", actuator_timeconst); addF("actuator_gain", actuator_gain); addF("actuator_bias", actuator_bias);
  addF("actuator_gear", actuator_gear);
  addI("body_parentid", body_parentid); addI("body_weldid", body_weldid); addI("body_jntadr", body_jntadr); addI("body_jntnum", body_jntnum);
  addI("body_geomadr", body_geomadr); addI("body_geomnum", body_geomnum);
  addI("jnt_type", jnt_type); addI("jnt_bodyid", jnt_bodyid); addI("jnt_limited", jnt_limited); addI("dof_parentid", dof_parentid);
  addI("geom_type", geom_type); addI("geom_bodyid", geom_bodyid); addI("geom_contype", geom_contype); addI("geom_conaffinity", geom_conaffinity);
  addI("geom_condim", geom_condim); addI("geom_priority", geom_priority);
  addI("site_bodyid", site_bodyid); addI("tendon_adr", tendon_adr); addI("tendon_num", tendon_num); addI("wrap_type", wrap_type);
  addI("wrap_objid", wrap_objid);
  addI("eq_type", eq_type); addI("eq_obj1id", eq_obj1id); addI("eq_obj2id", eq_obj2id); addI("actuator_trnid", actuator_trnid);
  addI("sensor_type", sensor_type); addI("sensor_objid", sensor_objid); addI("sensor_adr", sensor_adr);
  if (has_free) { addI("jnt_qposadr", jnt_qposadr); addI("jnt_dofadr", jnt_dofadr); addI("dof_jntid", dof_jntid); }   // (only then do the three index spaces differ)
  std::vector<std::string> bn, jn, gn, sn, tn, sen;
  for (auto& b : B) bn.push_back(b.name);
  for (auto& j : joints) jn.push_back(j.j->name);
  for (auto& g : geoms) gn.push_back(g.g->name);
  for (auto& s : sites) sn.push_back(s.s->name);
  for (auto& t : C.tendons) tn.push_back(t.name);
  for (auto& s : C.sensors) sen.push_back(s.name);
  const std::string names = join(bn) + "\n" + join(jn) + "\n" + join(gn) + "\n" + join(sn) + "\n" + join(tn) + "\n" + join(sen);
  put_record(body, "names", SG_DT_U8, (int64_t)names.size(), names.data(), names.size());
  nrec++;
  sg_blob_header h;
  h.magic = SG_BLOB_MAGIC; h.version = SG_BLOB_VERSION; h.nrec = (uint32_t)nrec; h.reserved = 0; h.total_bytes = (int64_t)(body.size() + sizeof h);
  std::string out((const char*)&h, sizeof h);
  out += body;
  return out;
}

}  // namespace

bool sg_mjcf_compile_file(const char* xml_path, bool composite_neighbors, bool implicit_tendon_damping, std::string* blob, std::string* err) {
  try {
    Compiler C;
    C.run(xml_path, composite_neighbors);
    *blob = finalize(C, implicit_tendon_damping);
    return true;
  } catch (const Fail& f) {
    *err = f.msg;
    return false;
  } catch (const std::exception& e) {
    *err = e.what();
    return false;
  }
}
