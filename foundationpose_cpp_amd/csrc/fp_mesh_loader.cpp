// fp_mesh_loader.cpp -- dependency-free mesh loading for the BaseMeshLoader contract.
//
// Replaces AssimpMeshLoader (reference detection_6d_foundationpose/src/mesh_loader/assimp_mesh_loader.cpp:116-295,
// API detection_6d_foundationpose/include/detection_6d_foundationpose/mesh_loader.hpp:15-93) without assimp / OpenCV /
// Eigen: Wavefront OBJ + MTL (map_Kd) + PNG (zlib inflate) in, the twelve getters' data out.
//   * first mesh only, polygons fan-triangulated (aiProcess_Triangulate), face-vertex tuples (v,vt,vn) that are
//     identical share one vertex in order of first use (the documented stand-in for aiProcess_JoinIdenticalVertices,
//     whose exact vertex order is unpinned -- SURVEY.md §8c);
//   * diameter = max pairwise vertex distance (:47-60), centre = AABB centre (:16-45,179-180), PCA OBB (:62-114);
//   * UVs are mandatory (throws in the reference :182-185 -> error here); missing/unreadable texture -> 2x2
//     (100,100,100) (:217-222); texture is returned RGB (imread BGR + cvtColor BGR2RGB, :216,223).
//   * meshes without normals get area-weighted vertex normals (the reference would dereference a null mNormals).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/foundationpose_amd.h"
#include "fp_internal.h"

namespace {

using fp::load_texture_rgb;  // image decoders live in fp_image_io.cpp

// ---------------------------------------------------------------- 3x3 symmetric eigen (Jacobi) ------------------------
void eigen_sym3(const double A_in[9], double evals[3], double evecs[9] /* columns, row-major storage */) {
  double A[9];
  std::memcpy(A, A_in, sizeof(A));
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-30) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (std::fabs(apq) < 1e-300) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A J
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T A
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int a, int b) { return A[a * 3 + a] < A[b * 3 + b]; });  // ascending, like Eigen
  for (int j = 0; j < 3; j++) {
    evals[j] = A[idx[j] * 3 + idx[j]];
    for (int k = 0; k < 3; k++) evecs[k * 3 + j] = V[k * 3 + idx[j]];
  }
}

std::string dirname_of(const std::string &p) {
  size_t s = p.find_last_of("/\\");
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

}  // namespace

namespace {
struct Corner { int v, t, n; };
struct ParsedMesh {
  std::vector<std::array<float, 3>> pos, nrm;
  std::vector<std::array<float, 2>> uv;
  std::vector<std::array<Corner, 3>> tris;
  std::string mtllib, usemtl;     // OBJ: the texture is named by the material library
  std::string texture_file;       // PLY: `comment TextureFile <name>` (what assimp turns into the diffuse texture)
};

// Stanford PLY (ascii / binary_little_endian / binary_big_endian): per-vertex x y z [nx ny nz] [s t | u v | texture_u texture_v],
// faces as `property list <T> <T> vertex_indices|vertex_index` (polygons are fanned), optional per-face `texcoord` list (MeshLab's
// per-wedge UVs), `comment TextureFile <file>`.  The BOP / YCB-V object models come in this form.
bool parse_ply(const std::string &path, ParsedMesh &out, std::string &err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open"; return false; }
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") { err = "not a PLY file"; return false; }
  enum Fmt { ASCII, LE, BE } fmt = ASCII;
  struct Prop { std::string name, type, ctype; bool list = false; };
  struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
  std::vector<Elem> elems;
  bool header_done = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ss(line);
    std::string tag;
    if (!(ss >> tag)) continue;
    if (tag == "format") { std::string v; ss >> v; fmt = v == "ascii" ? ASCII : v == "binary_little_endian" ? LE : BE; }
    else if (tag == "comment") { std::string k; ss >> k; if (k == "TextureFile") std::getline(ss >> std::ws, out.texture_file); }
    else if (tag == "element") { Elem e; ss >> e.name >> e.count; elems.push_back(e); }
    else if (tag == "property" && !elems.empty()) {
      Prop pr; std::string t; ss >> t;
      if (t == "list") { pr.list = true; ss >> pr.ctype >> pr.type >> pr.name; } else { pr.type = t; ss >> pr.name; }
      elems.back().props.push_back(pr);
    } else if (tag == "end_header") { header_done = true; break; }
  }
  if (!header_done) { err = "PLY header without end_header"; return false; }
  auto tsize = [](const std::string &t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
  };
  bool bad = false;
  auto read_num = [&](const std::string &t) -> double {
    if (fmt == ASCII) { double v = 0; if (!(f >> v)) bad = true; return v; }
    const int n = tsize(t);
    unsigned char b[8] = {0};
    if (n == 0 || !f.read((char *)b, n)) { bad = true; return 0; }
    if (fmt == BE) std::reverse(b, b + n);
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, b, 4); return v; }
    if (t == "double" || t == "float64") { double v; std::memcpy(&v, b, 8); return v; }
    if (t == "char" || t == "int8") return (signed char)b[0];
    if (t == "uchar" || t == "uint8") return b[0];
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, b, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, b, 4); return v; }
    uint32_t v; std::memcpy(&v, b, 4); return v;
  };
  bool has_uv = false, has_n = false;
  std::map<std::tuple<int, float, float>, int> wedge_uv;
  for (auto &e : elems) {
    if (e.count > (size_t)1 << 28) { err = "PLY element count out of range"; return false; }
    if (e.name == "vertex") {
      int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
      for (size_t k = 0; k < e.props.size(); k++) {
        const std::string &n = e.props[k].name;
        if (e.props[k].list) { err = "PLY: list property on vertices is not supported"; return false; }
        if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k;
        else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k; else if (n == "nz") inz = (int)k;
        else if (n == "s" || n == "u" || n == "texture_u") iu = (int)k; else if (n == "t" || n == "v" || n == "texture_v") iv = (int)k;
      }
      if (ix < 0 || iy < 0 || iz < 0) { err = "PLY vertices without x / y / z"; return false; }
      has_n = inx >= 0 && iny >= 0 && inz >= 0;
      has_uv = iu >= 0 && iv >= 0;
      std::vector<double> row(e.props.size());
      out.pos.reserve(std::min<size_t>(e.count, (size_t)1 << 20));   // (the count is file-supplied: no allocation by a corrupt header)
      for (size_t i = 0; i < e.count && !bad; i++) {
        for (size_t k = 0; k < e.props.size(); k++) row[k] = read_num(e.props[k].type);
        out.pos.push_back({(float)row[ix], (float)row[iy], (float)row[iz]});
        if (has_n) out.nrm.push_back({(float)row[inx], (float)row[iny], (float)row[inz]});
        if (has_uv) out.uv.push_back({(float)row[iu], (float)row[iv]});
      }
    } else if (e.name == "face") {
      for (size_t i = 0; i < e.count && !bad; i++) {
        std::vector<int> idx;
        std::vector<float> wedge;
        for (auto &pr : e.props) {
          if (!pr.list) { (void)read_num(pr.type); continue; }
          const long n = (long)read_num(pr.ctype);
          if (n < 0 || n > 4096) { bad = true; break; }
          for (long k = 0; k < n; k++) {
            const double v = read_num(pr.type);
            if (pr.name == "vertex_indices" || pr.name == "vertex_index") {
              if (!(v >= 0.0 && v < 2147483647.0)) { bad = true; break; }   // (range-checked as a double: the cast of NaN / out-of-range values is undefined)
              idx.push_back((int)v);
            }
            else if (pr.name == "texcoord") wedge.push_back((float)v);
          }
        }
        if (bad || idx.size() < 3) continue;
        std::vector<Corner> cs;
        for (size_t k = 0; k < idx.size(); k++) {
          Corner c{idx[k], has_uv ? idx[k] : -1, has_n ? idx[k] : -1};
          if (wedge.size() == 2 * idx.size()) {     // per-wedge UVs win over per-vertex ones (MeshLab); equal (vertex, uv) pairs share an entry
            const auto key = std::make_tuple(idx[k], wedge[2 * k], wedge[2 * k + 1]);
            auto it = wedge_uv.find(key);
            if (it == wedge_uv.end()) {
              out.uv.push_back({wedge[2 * k], wedge[2 * k + 1]});
              it = wedge_uv.emplace(key, (int)out.uv.size() - 1).first;
            }
            c.t = it->second;
          }
          cs.push_back(c);
        }
        for (size_t k = 2; k < cs.size(); k++) out.tris.push_back({cs[0], cs[k - 1], cs[k]});
      }
    } else {   // an element this loader does not use: skip its data
      for (size_t i = 0; i < e.count && !bad; i++)
        for (auto &pr : e.props) {
          if (!pr.list) { (void)read_num(pr.type); continue; }
          const long n = (long)read_num(pr.ctype);
          if (n < 0 || n > 1 << 20) { bad = true; break; }
          for (long k = 0; k < n; k++) (void)read_num(pr.type);
        }
    }
  }
  if (bad) { err = "truncated or malformed PLY body"; return false; }
  return true;
}
}  // namespace

struct fp_loaded_mesh {
  std::string name;
  std::vector<float> vertices, normals, texcoords;
  std::vector<uint32_t> faces;
  std::vector<uint8_t> texture;
  int th = 0, tw = 0;
  float diameter = 0;
  float center[3] = {0, 0, 0};
  float obb[16];  // column-major
  float dim[3];
  fp_mesh view;
};

extern "C" {

static fp_loaded_mesh *fp_mesh_load_obj_impl(const char *name, const char *mesh_file_path) {
  if (!mesh_file_path || !*mesh_file_path) { fp::set_error("[AssimpMeshLoader] Got empty mesh_file_path !"); return nullptr; }
  ParsedMesh pm;
  std::vector<std::array<float, 3>> &pos = pm.pos, &nrm = pm.nrm;
  std::vector<std::array<float, 2>> &uv = pm.uv;
  std::vector<std::array<Corner, 3>> &tris = pm.tris;
  std::string &mtllib = pm.mtllib, &usemtl = pm.usemtl;
  const std::string path_s(mesh_file_path);
  const std::string ext = path_s.size() >= 4 ? path_s.substr(path_s.size() - 4) : "";
  const bool is_ply = ext == ".ply" || ext == ".PLY";
  if (is_ply) {
    std::string perr;
    if (!parse_ply(path_s, pm, perr)) { fp::set_error("[AssimpMeshLoader] Failed to read mesh file: " + path_s + " (" + perr + ")"); return nullptr; }
  } else {
  std::ifstream f(mesh_file_path);
  if (!f) { fp::set_error(std::string("[AssimpMeshLoader] Failed to read mesh file: ") + mesh_file_path); return nullptr; }
  std::string line;
  bool first_object_done = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ss(line);
    std::string tag;
    if (!(ss >> tag) || tag[0] == '#') continue;
    if (tag == "v") { std::array<float, 3> p{}; ss >> p[0] >> p[1] >> p[2]; pos.push_back(p); }
    else if (tag == "vn") { std::array<float, 3> p{}; ss >> p[0] >> p[1] >> p[2]; nrm.push_back(p); }
    else if (tag == "vt") { std::array<float, 2> p{}; ss >> p[0] >> p[1]; uv.push_back(p); }
    else if (tag == "mtllib") { std::getline(ss >> std::ws, mtllib); }
    else if (tag == "usemtl") { if (usemtl.empty()) ss >> usemtl; }
    else if (tag == "o" || tag == "g") { if (!tris.empty()) first_object_done = true; }
    else if (tag == "f") {
      if (first_object_done) continue;  // only mMeshes[0] (assimp_mesh_loader.cpp:176)
      std::vector<Corner> cs;
      std::string tok;
      while (ss >> tok) {
        Corner c{0, 0, 0};
        int field = 0;
        size_t start = 0;
        for (size_t i = 0; i <= tok.size(); i++)
          if (i == tok.size() || tok[i] == '/') {
            std::string part = tok.substr(start, i - start);
            int val = part.empty() ? 0 : std::atoi(part.c_str());
            if (field == 0) c.v = val; else if (field == 1) c.t = val; else if (field == 2) c.n = val;
            field++; start = i + 1;
          }
        auto fix = [](int idx, size_t n) { return idx > 0 ? idx - 1 : (idx < 0 ? (int)n + idx : -1); };
        c.v = fix(c.v, pos.size()); c.t = fix(c.t, uv.size()); c.n = fix(c.n, nrm.size());
        cs.push_back(c);
      }
      for (size_t i = 2; i < cs.size(); i++) tris.push_back({cs[0], cs[i - 1], cs[i]});
    }
  }
  }
  if (tris.empty() || pos.empty()) { fp::set_error(std::string("[AssimpMeshLoader] Failed to read mesh file: ") + mesh_file_path); return nullptr; }
  std::unique_ptr<fp_loaded_mesh> owner(new fp_loaded_mesh());   // freed on every early return and if anything below throws
  fp_loaded_mesh *m = owner.get();
  m->name = name ? name : "";
  std::map<std::tuple<int, int, int>, uint32_t> seen;
  bool has_uv = true, has_n = true;
  for (auto &t : tris)
    for (auto &c : t) {
      if (c.v < 0 || c.v >= (int)pos.size()) { fp::set_error("[AssimpMeshLoader] Failed to read mesh file: bad vertex index"); return nullptr; }
      if (c.t < 0 || c.t >= (int)uv.size()) has_uv = false;
      if (c.n < 0 || c.n >= (int)nrm.size()) has_n = false;
    }
  if (!has_uv) { fp::set_error("[AssimpMeshLoader] Got invalid texturecoords!"); return nullptr; }
  for (auto &t : tris)
    for (auto &c : t) {
      auto key = std::make_tuple(c.v, c.t, has_n ? c.n : -1);
      auto it = seen.find(key);
      uint32_t id;
      if (it == seen.end()) {
        id = (uint32_t)(m->vertices.size() / 3);
        seen[key] = id;
        for (int k = 0; k < 3; k++) m->vertices.push_back(pos[c.v][k]);
        for (int k = 0; k < 2; k++) m->texcoords.push_back(uv[c.t][k]);
        for (int k = 0; k < 3; k++) m->normals.push_back(has_n ? nrm[c.n][k] : 0.0f);
      } else {
        id = it->second;
      }
      m->faces.push_back(id);
    }
  const size_t V = m->vertices.size() / 3;
  if (!has_n) {  // area-weighted vertex normals
    for (size_t fi = 0; fi + 2 < m->faces.size(); fi += 3) {
      const float *a = &m->vertices[m->faces[fi] * 3], *b = &m->vertices[m->faces[fi + 1] * 3], *c = &m->vertices[m->faces[fi + 2] * 3];
      float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) m->normals[m->faces[fi + k] * 3 + j] += n[j];
    }
    for (size_t v = 0; v < V; v++) {
      float *n = &m->normals[v * 3];
      float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (l > 0) { n[0] /= l; n[1] /= l; n[2] /= l; }
    }
  }
  // CalcMeshDiameter (:47-60): float, O(V^2)
  {
    float best = 0.0f;
    const float *v = m->vertices.data();
    for (size_t i = 0; i < V; i++)
      for (size_t j = i + 1; j < V; j++) {
        float dx = v[i * 3] - v[j * 3], dy = v[i * 3 + 1] - v[j * 3 + 1], dz = v[i * 3 + 2] - v[j * 3 + 2];
        float d = dx * dx + dy * dy + dz * dz;
        if (d > best) best = d;
      }
    m->diameter = std::sqrt(best);
  }
  // FindMinMaxVertex + centre (:16-45,179-180)
  {
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) mn[k] = mx[k] = m->vertices[k];
    for (size_t i = 0; i < V; i++)
      for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], m->vertices[i * 3 + k]); mx[k] = std::max(mx[k], m->vertices[i * 3 + k]); }
    for (int k = 0; k < 3; k++) m->center[k] = (float)((mx[k] + mn[k]) / 2.0);
  }
  // ComputeOBB (:62-114): mean, covariance / V, eigen-decomposition (ascending), extents of R^T v
  {
    double mean[3] = {0, 0, 0};
    for (size_t i = 0; i < V; i++) for (int k = 0; k < 3; k++) mean[k] += m->vertices[i * 3 + k];
    for (int k = 0; k < 3; k++) mean[k] /= (double)V;
    double cov[9] = {0};
    for (size_t i = 0; i < V; i++) {
      double d[3] = {m->vertices[i * 3] - mean[0], m->vertices[i * 3 + 1] - mean[1], m->vertices[i * 3 + 2] - mean[2]};
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) cov[a * 3 + b] += d[a] * d[b];
    }
    for (double &c : cov) c /= (double)V;
    double ev[3], R[9];
    eigen_sym3(cov, ev, R);
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < V; i++)
      for (int j = 0; j < 3; j++) {
        double pr = R[0 * 3 + j] * m->vertices[i * 3] + R[1 * 3 + j] * m->vertices[i * 3 + 1] + R[2 * 3 + j] * m->vertices[i * 3 + 2];
        mn[j] = std::min(mn[j], pr); mx[j] = std::max(mx[j], pr);
      }
    for (int j = 0; j < 3; j++) m->dim[j] = (float)(mx[j] - mn[j]);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m->obb[c * 4 + r] = (float)R[r * 3 + c];
    for (int r = 0; r < 3; r++) { m->obb[12 + r] = (float)mean[r]; m->obb[r * 4 + 3] = 0; }
    m->obb[15] = 1;
  }
  // texture: MTL map_Kd of the first used material (or the first map_Kd found)
  {
    std::string tex_path;
    if (!mtllib.empty()) {
      std::ifstream mf(dirname_of(mesh_file_path) + "/" + mtllib);
      std::string cur, l2, first_map;
      while (mf && std::getline(mf, l2)) {
        if (!l2.empty() && l2.back() == '\r') l2.pop_back();
        std::istringstream ss(l2);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "newmtl") ss >> cur;
        else if (tag == "map_Kd") {
          std::string p;
          std::getline(ss >> std::ws, p);
          if (first_map.empty()) first_map = p;
          if (tex_path.empty() && (usemtl.empty() || cur == usemtl)) tex_path = p;
        }
      }
      if (tex_path.empty()) tex_path = first_map;
      if (!tex_path.empty()) tex_path = dirname_of(mesh_file_path) + "/" + tex_path;
    }
    if (is_ply && !pm.texture_file.empty()) tex_path = dirname_of(mesh_file_path) + "/" + pm.texture_file;
    const bool named = !tex_path.empty();
    const bool present = named && std::ifstream(tex_path, std::ios::binary).good();
    std::string why;
    if (present && !load_texture_rgb(tex_path, m->texture, m->th, m->tw, &why)) {
      // The reference decodes whatever cv::imread knows; this loader reads PNG (any bit depth, interlaced or not), BMP, PNM and
      // TGA.  A texture file that EXISTS but cannot be decoded here (JPEG, TIFF, WebP ...) must not silently become the grey
      // default: the rendered crops would get the wrong colours and refine / score accuracy would drop without a trace.
      fp::set_error("[MeshLoader] texture '" + tex_path + "' named by map_Kd cannot be decoded: " + why);
      return nullptr;
    }
    if (!present) {  // no map_Kd, or the file is missing: default texture map, like the reference (:217-222)
      m->th = m->tw = 2;
      m->texture.assign(12, 100);
    }
  }
  m->view = fp_mesh{m->name.c_str(), (int)V, (int)(m->faces.size() / 3), m->vertices.data(), m->normals.data(),
                    m->texcoords.data(), m->faces.data(), m->texture.data(), m->th, m->tw, m->diameter,
                    {m->center[0], m->center[1], m->center[2]}};
  return owner.release();
}
fp_loaded_mesh *fp_mesh_load_obj(const char *name, const char *mesh_file_path) {
  try {
    return fp_mesh_load_obj_impl(name, mesh_file_path);
  } catch (const std::exception &e) {  // nothing may unwind through the C ABI
    fp::set_error(std::string("fp_mesh_load_obj: ") + e.what());
    return nullptr;
  }
}


void fp_mesh_free(fp_loaded_mesh *m) { delete m; }

const fp_mesh *fp_mesh_view(const fp_loaded_mesh *m) { return m ? &m->view : nullptr; }

int fp_mesh_orient_bounds(const fp_loaded_mesh *m, float orient_bounds[16], float dimension[3]) {
  if (!m) { fp::set_error("null mesh"); return 1; }
  if (orient_bounds) std::memcpy(orient_bounds, m->obb, sizeof(float) * 16);
  if (dimension) std::memcpy(dimension, m->dim, sizeof(float) * 3);
  return 0;
}

}  // extern "C"
