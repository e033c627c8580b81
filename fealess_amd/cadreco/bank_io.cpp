// bank_io.cpp -- reads an existing CadReco data directory without OpenCV:
//   <dir>/linemod_templates.yml   OpenCV FileStorage YAML 1.0 as written by writeLinemod
//                                 (reference: linemod/linemod_if.cpp:36-63, linemod/linemod.cpp:98-129,
//                                  1681-1794)
//   <dir>/depth/<template_id>.png 16-bit single-channel PNG in 0.1 mm (CadReco/obj_reco_lmicp.cpp:156-157)
// The YAML reader handles the subset FileStorage emits for this schema: block mappings, block
// sequences introduced by "-", flow sequences "[ ... ]" (possibly wrapped over several lines),
// plain / quoted scalars.  The PNG reader handles non-interlaced 8/16-bit grayscale (zlib inflate
// + the five scanline filters).
#include <sys/stat.h>
#include "fealess_cadreco.h"

#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

namespace fealess {
namespace {

struct Node {
  enum Kind { SCALAR, SEQ, MAP } kind = SCALAR;
  std::string scalar;
  std::vector<std::unique_ptr<Node> > seq;
  std::vector<std::pair<std::string, std::unique_ptr<Node> > > map;
  const Node *get(const std::string &k) const
  {
    for (auto &kv : map)
      if (kv.first == k) return kv.second.get();
    return nullptr;
  }
  double num(double dflt = 0) const { return kind == SCALAR && !scalar.empty() ? strtod(scalar.c_str(), nullptr) : dflt; }
};

struct Line { int indent; std::string text; };

std::string trim(const std::string &s)
{
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::string unquote(const std::string &s)
{
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
  return s;
}

class Parser {
 public:
  explicit Parser(std::vector<Line> lines) : L(std::move(lines)) {}
  std::unique_ptr<Node> parse() { return block(0 <= (int)L.size() - 1 ? L[0].indent : 0); }
  std::string err;

 private:
  std::vector<Line> L;
  size_t pos = 0;

  // a flow sequence may continue on following lines until brackets balance
  std::string gather_flow(std::string first)
  {
    auto depth = [](const std::string &s) { int d = 0; for (char c : s) d += (c == '[') - (c == ']'); return d; };
    std::string s = first;
    while (depth(s) > 0 && pos < L.size()) s += " " + L[pos++].text;
    return s;
  }
  std::unique_ptr<Node> flow(const std::string &text, size_t &i)
  {
    std::unique_ptr<Node> n(new Node);
    n->kind = Node::SEQ;
    ++i;   // '['
    std::string cur;
    auto flush = [&]() { std::string t = trim(cur); if (!t.empty()) { std::unique_ptr<Node> c(new Node); c->scalar = unquote(t); n->seq.push_back(std::move(c)); } cur.clear(); };
    while (i < text.size()) {
      char c = text[i];
      if (c == '[') { n->seq.push_back(flow(text, i)); cur.clear(); }
      else if (c == ']') { flush(); ++i; return n; }
      else if (c == ',') { flush(); ++i; }
      else { cur += c; ++i; }
    }
    return n;
  }
  std::unique_ptr<Node> value(const std::string &v, int parent_indent)
  {
    std::string t = trim(v);
    if (t.empty()) {                       // nested block on the following lines
      if (pos < L.size() && (L[pos].indent > parent_indent || (L[pos].indent == parent_indent && L[pos].text[0] == '-')))
        return block(L[pos].indent);
      return std::unique_ptr<Node>(new Node);
    }
    if (t[0] == '[') { std::string f = gather_flow(t); size_t i = 0; return flow(f, i); }
    std::unique_ptr<Node> n(new Node);
    n->scalar = unquote(t);
    return n;
  }
  std::unique_ptr<Node> block(int indent)
  {
    std::unique_ptr<Node> n(new Node);
    if (pos >= L.size()) return n;
    bool is_seq = L[pos].text[0] == '-';
    n->kind = is_seq ? Node::SEQ : Node::MAP;
    while (pos < L.size() && L[pos].indent == indent) {
      std::string t = L[pos].text;
      if (is_seq) {
        if (t[0] != '-') break;
        ++pos;
        std::string rest = trim(t.substr(1));
        if (rest.empty()) {                // "-" alone: the item is the block that follows
          if (pos < L.size() && L[pos].indent > indent) n->seq.push_back(block(L[pos].indent));
          else n->seq.push_back(std::unique_ptr<Node>(new Node));
        } else if (rest[0] == '[') {
          std::string f = gather_flow(rest); size_t i = 0; n->seq.push_back(flow(f, i));
        } else if (rest.find(':') != std::string::npos && rest[0] != '"') {
          // "- key: value" opens a mapping whose further keys are indented past the dash
          const int sub = indent + (int)t.find(rest, 1);
          L.insert(L.begin() + pos, Line{sub, rest});
          n->seq.push_back(block(sub));
        } else {
          std::unique_ptr<Node> c(new Node); c->scalar = unquote(rest); n->seq.push_back(std::move(c));
        }
      } else {
        if (t[0] == '-') break;
        size_t c = t.find(':');
        if (c == std::string::npos) { err = "expected 'key:' in '" + t + "'"; ++pos; continue; }
        std::string key = unquote(trim(t.substr(0, c)));
        ++pos;
        n->map.emplace_back(key, value(t.substr(c + 1), indent));
      }
    }
    return n;
  }
};

bool load_lines(const std::string &filename, std::vector<Line> &out)
{
  std::ifstream f(filename.c_str());
  if (!f) return false;
  std::string s;
  while (std::getline(f, s)) {
    if (!s.empty() && s.back() == '\r') s.pop_back();
    std::string t = trim(s);
    if (t.empty() || t[0] == '%' || t == "---" || t == "..." || t[0] == '#') continue;
    int ind = 0;
    while (ind < (int)s.size() && s[ind] == ' ') ++ind;
    out.push_back(Line{ind, t});
  }
  return true;
}

void read_template(const Node *n, Template &t)   // Template::read, linemod.cpp:98-113
{
  auto gi = [&](const char *k) { const Node *c = n->get(k); return c ? (int)c->num() : 0; };
  t.width = gi("width");
  t.height = gi("height");
  t.offset_x = gi("offset_x");
  t.offset_y = gi("offset_y");
  t.pyramid_level = gi("pyramid_level");
  const Node *fs = n->get("features");
  if (fs && fs->kind == Node::SEQ)
    for (auto &f : fs->seq)
      if (f->kind == Node::SEQ && f->seq.size() >= 3)
        t.features.push_back(Feature{(int)f->seq[0]->num(), (int)f->seq[1]->num(), (int)f->seq[2]->num()});
}

}  // namespace

bool ReadLinemod(const std::string &filename, DetectorFile &out, std::string *err)
{
  std::vector<Line> lines;
  if (!load_lines(filename, lines)) { if (err) *err = "cannot open " + filename; return false; }
  if (lines.empty()) { if (err) *err = "empty file"; return false; }
  Parser p(std::move(lines));
  std::unique_ptr<Node> root = p.parse();
  out = DetectorFile();
  // Detector::read (linemod.cpp:1681-1694)
  if (const Node *n = root->get("pyramid_levels")) out.pyramid_levels = (int)n->num();
  if (const Node *n = root->get("T"))
    for (auto &c : n->seq) out.T.push_back((int)c->num());
  if (const Node *n = root->get("modalities"))
    for (auto &m : n->seq)
      if (const Node *t = m->get("type")) out.modalities.push_back(t->scalar);
  // readClass (linemod.cpp:1711-1762)
  if (const Node *cls = root->get("classes"))
    for (auto &c : cls->seq) {
      ObjectClass oc;
      if (const Node *id = c->get("class_id")) oc.class_id = id->scalar;
      if (const Node *pl = c->get("pyramid_levels"))
        if ((int)pl->num() != out.pyramid_levels) { if (err) *err = "class pyramid_levels mismatch (CV_Assert linemod.cpp:1720)"; return false; }
      if (const Node *tps = c->get("template_pyramids")) {
        int expected = 0;
        for (auto &tp : tps->seq) {
          const Node *tid = tp->get("template_id");
          if (!tid || (int)tid->num() != expected) { if (err) *err = "template_id != expected_id (CV_Assert linemod.cpp:1745)"; return false; }
          ++expected;
          std::vector<float> pose;
          if (const Node *ps = tp->get("template_pose"))
            for (auto &v : ps->seq) pose.push_back((float)v->num());
          oc.poses.push_back(pose);
          std::vector<Template> pyr;
          if (const Node *ts = tp->get("templates"))
            for (auto &t : ts->seq) { Template tt; read_template(t.get(), tt); pyr.push_back(tt); }
          oc.template_pyramids.push_back(pyr);
        }
      }
      out.classes.push_back(oc);
    }
  if (!p.err.empty() && err) *err = p.err;
  return true;
}

// ---- packed binary cache of a DetectorFile (parsing a 16000-template YAML takes seconds, the
// cache is read with a handful of fread calls).  Little-endian, versioned, with the YAML's size and mtime so a stale
// cache is ignored.  Layout: magic "FLBANK1\0", u64 yml_size, i64 yml_mtime, i32 levels, i32 nT, T[], i32 nmod,
// {i32 len, bytes}[], i32 nclasses, per class {i32 len, id, i32 npyr, per pyramid {i32 npose, f32[], i32 ntempl,
// per template {5 x i32, i32 nfeat, nfeat x 3 x i32}}}.
namespace {
struct BinW {
  FILE *f;
  bool ok = true;
  void raw(const void *p, size_t n) { ok = ok && fwrite(p, 1, n, f) == n; }
  void i32(int v) { raw(&v, 4); }
  void str(const std::string &s) { i32((int)s.size()); raw(s.data(), s.size()); }
};
struct BinR {
  FILE *f;
  bool ok = true;
  void raw(void *p, size_t n) { ok = ok && fread(p, 1, n, f) == n; }
  int i32() { int v = 0; raw(&v, 4); return v; }
  std::string str() { int n = i32(); std::string s; if (ok && n >= 0 && n < (1 << 20)) { s.resize((size_t)n); raw(&s[0], (size_t)n); } else ok = false; return s; }
};
const char kBankMagic[8] = {'F', 'L', 'B', 'A', 'N', 'K', '1', 0};
}  // namespace

bool WriteBankCache(const DetectorFile &det, const std::string &filename, unsigned long long yml_size, long long yml_mtime)
{
  FILE *f = fopen(filename.c_str(), "wb");
  if (!f) return false;
  BinW w{f};
  w.raw(kBankMagic, 8);
  w.raw(&yml_size, 8);
  w.raw(&yml_mtime, 8);
  w.i32(det.pyramid_levels);
  w.i32((int)det.T.size());
  for (int t : det.T) w.i32(t);
  w.i32((int)det.modalities.size());
  for (auto &m : det.modalities) w.str(m);
  w.i32((int)det.classes.size());
  for (auto &c : det.classes) {
    w.str(c.class_id);
    w.i32((int)c.template_pyramids.size());
    for (size_t p = 0; p < c.template_pyramids.size(); ++p) {
      const std::vector<float> &pose = p < c.poses.size() ? c.poses[p] : std::vector<float>();
      w.i32((int)pose.size());
      if (!pose.empty()) w.raw(pose.data(), pose.size() * 4);
      w.i32((int)c.template_pyramids[p].size());
      for (auto &t : c.template_pyramids[p]) {
        const int hdr[6] = {t.width, t.height, t.offset_x, t.offset_y, t.pyramid_level, (int)t.features.size()};
        w.raw(hdr, sizeof(hdr));
        if (!t.features.empty()) w.raw(t.features.data(), t.features.size() * sizeof(Feature));
      }
    }
  }
  const bool ok = w.ok;
  return fclose(f) == 0 && ok;
}

bool ReadBankCache(const std::string &filename, DetectorFile &out, unsigned long long yml_size, long long yml_mtime)
{
  FILE *f = fopen(filename.c_str(), "rb");
  if (!f) return false;
  BinR r{f};
  char magic[8];
  unsigned long long sz = 0;
  long long mt = 0;
  r.raw(magic, 8);
  r.raw(&sz, 8);
  r.raw(&mt, 8);
  bool ok = r.ok && memcmp(magic, kBankMagic, 8) == 0 && sz == yml_size && mt == yml_mtime;
  out = DetectorFile();
  if (ok) {
    out.pyramid_levels = r.i32();
    int nT = r.i32();
    for (int i = 0; r.ok && i < nT && nT < 64; ++i) out.T.push_back(r.i32());
    int nm = r.i32();
    for (int i = 0; r.ok && i < nm && nm < 64; ++i) out.modalities.push_back(r.str());
    int nc = r.i32();
    for (int ci = 0; r.ok && ci < nc; ++ci) {
      ObjectClass oc;
      oc.class_id = r.str();
      int np = r.i32();
      for (int p = 0; r.ok && p < np; ++p) {
        int npose = r.i32();
        std::vector<float> pose;
        if (r.ok && npose >= 0 && npose < 1024) { pose.resize((size_t)npose); if (npose) r.raw(pose.data(), (size_t)npose * 4); } else r.ok = false;
        oc.poses.push_back(pose);
        int nt = r.i32();
        std::vector<Template> pyr;
        for (int ti = 0; r.ok && ti < nt && nt < 1024; ++ti) {
          int hdr[6] = {0};
          r.raw(hdr, sizeof(hdr));
          Template t;
          t.width = hdr[0]; t.height = hdr[1]; t.offset_x = hdr[2]; t.offset_y = hdr[3]; t.pyramid_level = hdr[4];
          if (r.ok && hdr[5] >= 0 && hdr[5] < (1 << 20)) { t.features.resize((size_t)hdr[5]); if (hdr[5]) r.raw(t.features.data(), (size_t)hdr[5] * sizeof(Feature)); } else r.ok = false;
          pyr.push_back(t);
        }
        oc.template_pyramids.push_back(pyr);
      }
      out.classes.push_back(oc);
    }
    ok = r.ok;
  }
  fclose(f);
  if (!ok) out = DetectorFile();
  return ok;
}

// readLinemod through the cache: <file>.flbank next to the YAML, rebuilt whenever the YAML's size or mtime changes
bool ReadLinemodCached(const std::string &filename, DetectorFile &out, std::string *err, bool *from_cache)
{
  struct stat st;
  if (from_cache) *from_cache = false;
  if (stat(filename.c_str(), &st) != 0) { if (err) *err = "cannot open " + filename; return false; }
  const std::string cache = filename + ".flbank";
  if (ReadBankCache(cache, out, (unsigned long long)st.st_size, (long long)st.st_mtime)) {
    if (from_cache) *from_cache = true;
    return true;
  }
  if (!ReadLinemod(filename, out, err)) return false;
  (void)WriteBankCache(out, cache, (unsigned long long)st.st_size, (long long)st.st_mtime);   // best effort (read-only dirs)
  return true;
}

bool WriteLinemod(const DetectorFile &det, const std::string &filename)
{
  FILE *f = fopen(filename.c_str(), "w");
  if (!f) return false;
  fprintf(f, "%%YAML:1.0\n---\npyramid_levels: %d\nT: [ ", det.pyramid_levels);
  for (size_t i = 0; i < det.T.size(); ++i) fprintf(f, "%s%d", i ? ", " : "", det.T[i]);
  fprintf(f, " ]\nmodalities:\n");
  for (auto &m : det.modalities) {
    // default parameters of linemod.cpp:515-519 / 827-832
    if (m == "ColorGradient")
      fprintf(f, "   -\n      type: ColorGradient\n      weak_threshold: 10.\n      num_features: 63\n      strong_threshold: 55.\n");
    else
      fprintf(f, "   -\n      type: %s\n      distance_threshold: 2000\n      difference_threshold: 50\n      num_features: 63\n      extract_threshold: 2\n", m.c_str());
  }
  fprintf(f, "classes:\n");
  for (auto &c : det.classes) {
    fprintf(f, "   -\n      class_id: \"%s\"\n      modalities: [ ", c.class_id.c_str());
    for (size_t i = 0; i < det.modalities.size(); ++i) fprintf(f, "%s%s", i ? ", " : "", det.modalities[i].c_str());
    fprintf(f, " ]\n      pyramid_levels: %d\n      template_pyramids:\n", det.pyramid_levels);
    for (size_t t = 0; t < c.template_pyramids.size(); ++t) {
      fprintf(f, "         -\n            template_id: %zu\n            template_pose: [ ", t);
      static const std::vector<float> no_pose;
      const std::vector<float> &p = t < c.poses.size() ? c.poses[t] : no_pose;
      for (size_t i = 0; i < p.size(); ++i) fprintf(f, "%s%.9g%s", i ? ", " : "", p[i], (i % 4 == 3 && i + 1 < p.size()) ? "\n               " : "");
      fprintf(f, " ]\n            templates:\n");
      for (auto &tt : c.template_pyramids[t]) {
        fprintf(f, "               -\n                  width: %d\n                  height: %d\n                  offset_x: %d\n"
                   "                  offset_y: %d\n                  pyramid_level: %d\n                  features:\n",
                tt.width, tt.height, tt.offset_x, tt.offset_y, tt.pyramid_level);
        for (auto &ft : tt.features) fprintf(f, "                     - [ %d, %d, %d ]\n", ft.x, ft.y, ft.label);
      }
    }
  }
  fclose(f);
  return true;
}

// ---- PNG (16-bit or 8-bit grayscale, non-interlaced) ---------------------------------------------
bool ReadPng16(const std::string &filename, std::vector<unsigned short> &pixels, int &w, int &h, std::string *err)
{
  auto fail = [&](const char *m) { if (err) *err = std::string(m) + ": " + filename; return false; };
  std::ifstream f(filename.c_str(), std::ios::binary);
  if (!f) return fail("cannot open");
  std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (buf.size() < 8 || memcmp(buf.data(), sig, 8)) return fail("not a PNG");
  auto be32 = [&](size_t o) { return (unsigned)buf[o] << 24 | (unsigned)buf[o + 1] << 16 | (unsigned)buf[o + 2] << 8 | buf[o + 3]; };
  size_t o = 8;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat;
  w = h = 0;
  while (o + 12 <= buf.size()) {
    unsigned len = be32(o);
    std::string type((const char *)&buf[o + 4], 4);
    if (o + 12 + len > buf.size()) return fail("truncated chunk");
    const unsigned char *d = &buf[o + 8];
    if (type == "IHDR") {
      if (len != 13) return fail("bad IHDR");
      w = (int)be32(o + 8); h = (int)be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12];
    }
    else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
    else if (type == "IEND") break;
    o += 12 + len;
  }
  if (w <= 0 || h <= 0 || ctype != 0 || (depth != 16 && depth != 8) || interlace) return fail("unsupported PNG (need non-interlaced 8/16-bit gray)");
  if ((long long)w * h > (64ll << 20)) return fail("PNG larger than 64 Mpixel");
  const int bpp = depth / 8;
  const size_t stride = (size_t)w * bpp;
  std::vector<unsigned char> raw((stride + 1) * (size_t)h);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return fail("inflate failed");
  std::vector<unsigned char> img(stride * (size_t)h);
  for (int y = 0; y < h; ++y) {
    const unsigned char *src = &raw[(stride + 1) * (size_t)y];
    unsigned char *cur = &img[stride * (size_t)y];
    const unsigned char *up = y ? &img[stride * (size_t)(y - 1)] : nullptr;
    const int ft = src[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
      int v = src[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: return fail("bad filter");
      }
      cur[i] = (unsigned char)v;
    }
  }
  pixels.resize((size_t)w * h);
  for (size_t i = 0; i < pixels.size(); ++i)
    pixels[i] = depth == 16 ? (unsigned short)(img[2 * i] << 8 | img[2 * i + 1]) : (unsigned short)img[i];
  return true;
}

}  // namespace fealess
