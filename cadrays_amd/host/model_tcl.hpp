// model_tcl.hpp -- C++ reader of the scene wire format CADRays itself writes and re-reads: model.tcl + meshes/*.ply + textures/
// (exporter: reference src/ImportExport/ImportExport.cxx:164-305, 436-607; re-import = sourcing model.tcl,
// src/Launcher/ImportSettingsEditor.cxx:378-380).  The reference host is C++, so a C++ host of this backend can load a saved
// scene without Python.  What the exporter emits is a straight-line command list; this reader evaluates exactly that subset
// (no control flow, no expr -- cadrays_amd/scene_tcl.py is the full evaluator for hand-written demo scripts):
//
//   variable Root [file dirname ...] | set Root <dir>        $Root = directory of the script
//   rtmeshread <file.ply> <name> [options]                   binary / ascii PLY with normals and s/t (AisMesh.cxx:490 writes 'plyb')
//   vdisplay / verase <name>...   vclear
//   vsetmaterial <name> <stock>                              stand-ins for OCCT's stock materials (every field is overridden by vbsdf)
//   vbsdf <name> -Kc|-Kd|-Ks|-Kt|-Le r g b | -baseRoughness|-coatRoughness x | -absorpColor r g b | -absorpCoeff x |
//                -baseFresnel|-coatFresnel Schlick r g b | Constant f | Conductor n k | Dielectric n | -normalize
//   rttexture <name> <image> | -scale S T | -on | -off       8-bit PNG or baseline JPEG (jpeg_baseline.hpp); texels squared like the environment [OCCT-ext]
//   vlocation <name> -rotation x y z w | -scale s | -location x y z
//   vcamera -orthographic | -perspective | -fovy a | -distance d      vviewparams -proj|-up|-at|-eye x y z | -size s
//   vtextureenv on <image>        vlight clear | add directional direction x y z | add positional position x y z ... smoothness s
//   intensity i [head 1]          rtlight <id> -color r g b           vrenderparams ... -rayDepth n
//   rtdisplay / rterase <node>                               like vdisplay / verase
//   rtmodel / vupdate ...                                    accepted, no effect on the path
//
// Numbers are parsed as double and narrowed to float where the Python reader narrows them, so both hosts hand the same bytes
// to the boundary (tests/test_scene_tcl.py::test_cpp_driver_reads_model_tcl_like_the_python_reader).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/cadrays_hip.h"
#include "jpeg_baseline.hpp"

namespace crh_host {

struct Texture { uint32_t w = 0, h = 0, ch = 0; std::vector<float> texels; };

struct TclScene {
  std::vector<float> pos, nrm, uv;            // per vertex; uv empty when no texture is bound
  std::vector<int32_t> tri;                   // i0 i1 i2 material
  std::vector<crh_bsdf> mats;
  std::vector<crh_light> lights;
  std::vector<float> env; uint32_t envW = 0, envH = 0;
  std::vector<Texture> textures;
  crh_camera cam{}; crh_params par{};
  std::vector<std::string> unsupported;       // commands this reader accepted but could not honour
};

namespace detail {

inline std::string lower(std::string s) { for (char& c : s) c = (char)tolower((unsigned char)c); return s; }
inline bool is_number(const std::string& s)
{
  if (s.empty()) return false;
  char* e = nullptr; strtod(s.c_str(), &e);
  return e && *e == 0;
}

// ---- minimal PNG (8-bit, non-interlaced, colour types 0 / 2 / 3 / 4 / 6) over zlib
inline bool read_png(const std::string& path, uint32_t& w, uint32_t& h, uint32_t& ch, std::vector<uint8_t>& out, std::string& err)
{
  std::vector<uint8_t> d;
  if (!read_file_bytes(path, d, err)) return false;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) { err = path + ": not a PNG file"; return false; }
  auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
  std::vector<uint8_t> idat, plte, trns; uint32_t depth = 0, ctype = 0, interlace = 0; w = h = 0;
  for (size_t o = 8; o + 12 <= d.size();) {
    const uint32_t len = be32(o); const std::string type((const char*)&d[o + 4], 4);
    if (o + 12 + len > d.size()) break;
    const uint8_t* p = &d[o + 8];
    if (type == "IHDR") { if (len < 13) { err = path + ": truncated PNG header"; return false; } w = be32(o + 8); h = be32(o + 12); depth = p[8]; ctype = p[9]; interlace = p[12]; }
    else if (type == "PLTE") plte.assign(p, p + len);
    else if (type == "tRNS") trns.assign(p, p + len);
    else if (type == "IDAT") idat.insert(idat.end(), p, p + len);
    else if (type == "IEND") break;
    o += 12 + len;
  }
  if (!w || !h || depth != 8 || interlace != 0) { err = path + ": only 8-bit non-interlaced PNG images are read"; return false; }
  const uint32_t spp = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!spp) { err = path + ": unknown PNG colour type"; return false; }
  const size_t stride = (size_t)w * spp;
  // a file is untrusted input: the header's size must be one the compressed data can actually fill (deflate expands at most ~1032 : 1)
  if (w > (1u << 16) || h > (1u << 16) || (double)(stride + 1) * h > 1032.0 * (double)idat.size() + 65536.0) { err = path + ": image size does not fit its data"; return false; }
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { err = path + ": corrupt image data"; return false; }
  std::vector<uint8_t> img(stride * h);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* s = &raw[(stride + 1) * y + 1];
    uint8_t* r = &img[stride * y]; const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= spp ? r[x - spp] : 0, b = up ? up[x] : 0, c = (up && x >= spp) ? up[x - spp] : 0;
      int pr = 0;
      switch (ft) {
        case 0: pr = 0; break; case 1: pr = a; break; case 2: pr = b; break; case 3: pr = (a + b) >> 1; break;
        case 4: { const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); } break;
        default: err = path + ": bad PNG filter"; return false;
      }
      r[x] = (uint8_t)(s[x] + pr);
    }
  }
  // to RGB or RGBA (alpha kept when the file has one)
  const bool alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
  ch = alpha ? 4 : 3;
  out.resize((size_t)w * h * ch);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    uint8_t r = 0, g = 0, b = 0, a = 255; const uint8_t* p = &img[i * spp];
    if (ctype == 0) { r = g = b = p[0]; } else if (ctype == 2) { r = p[0]; g = p[1]; b = p[2]; }
    else if (ctype == 3) { const size_t k = p[0]; if (3 * k + 2 < plte.size()) { r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2]; } if (k < trns.size()) a = trns[k]; }
    else if (ctype == 4) { r = g = b = p[0]; a = p[1]; } else { r = p[0]; g = p[1]; b = p[2]; a = p[3]; }
    uint8_t* o = &out[i * ch]; o[0] = r; o[1] = g; o[2] = b; if (alpha) o[3] = a;
  }
  return true;
}

// RGB / RGBA bytes -> 8-bit PNG (filter 0, one zlib stream): the file format of the reference's image dump
// (BufferDump -> Output_<name>_<n>.png, AppViewer.cxx:1259-1261, main.cxx:193-228), which the harness compares pixel by pixel
inline bool write_png(const std::string& path, const uint8_t* px, uint32_t w, uint32_t h, uint32_t ch, std::string& err)
{
  if ((ch != 3 && ch != 4) || !w || !h) { err = path + ": only RGB / RGBA images are written"; return false; }
  const size_t stride = (size_t)w * ch;
  std::vector<uint8_t> raw((stride + 1) * h);
  for (uint32_t y = 0; y < h; ++y) { raw[(stride + 1) * y] = 0; memcpy(&raw[(stride + 1) * y + 1], px + stride * y, stride); }
  uLongf zlen = compressBound((uLong)raw.size()); std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = path + ": compression failed"; return false; }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { err = "cannot write " + path; return false; }
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
    const uint8_t L[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    uLong crc = crc32(0L, (const Bytef*)type, 4); if (len) crc = crc32(crc, data, len);
    const uint8_t Cc[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    fwrite(L, 1, 4, f); fwrite(type, 1, 4, f); if (len) fwrite(data, 1, len, f); fwrite(Cc, 1, 4, f);
  };
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  fwrite(sig, 1, 8, f);
  const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                            8, (uint8_t)(ch == 4 ? 6 : 2), 0, 0, 0};
  chunk("IHDR", ihdr, 13); chunk("IDAT", z.data(), (uint32_t)zlen); chunk("IEND", nullptr, 0);
  const bool ok = !ferror(f); fclose(f);
  if (!ok) err = "cannot write " + path;
  return ok;
}

// 8-bit image file -> RGB(A) bytes; the format is recognised by content, not by extension
inline bool read_image_u8(const std::string& path, uint32_t& w, uint32_t& h, uint32_t& ch, std::vector<uint8_t>& px, std::string& err)
{
  uint8_t sig[2] = {0, 0};
  { std::vector<uint8_t> head; std::string e2; struct stat st; if (stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) if (FILE* f = fopen(path.c_str(), "rb")) { if (fread(sig, 1, 2, f) != 2) sig[0] = sig[1] = 0; fclose(f); } }
  if (sig[0] == 0xFF && sig[1] == 0xD8) { ch = 3; return read_jpeg(path, w, h, px, err); }
  return read_png(path, w, h, ch, px, err);
}

// 8-bit image -> linear float texels: rgb squared ("de-gamma for gamma = 2", like the environment map [OCCT-ext]), alpha kept
inline bool load_texture(const std::string& path, Texture& t, std::string& err)
{
  std::vector<uint8_t> px;
  if (!read_image_u8(path, t.w, t.h, t.ch, px, err)) return false;
  t.texels.resize(px.size());
  for (size_t i = 0; i < px.size(); ++i) {
    const float v = (float)px[i] / 255.0f;
    t.texels[i] = (t.ch == 4 && (i & 3) == 3) ? v : v * v;
  }
  return true;
}

struct Mesh { std::vector<float> pos, nrm, uv; std::vector<int32_t> faces; bool has_uv = false; };

// PLY: vertex element with x y z [nx ny nz] [s t | u v | texture_u texture_v], face element with one index list; binary LE or ascii
inline bool read_ply(const std::string& path, Mesh& m, std::string& err, bool smooth = false)
{
  std::vector<uint8_t> d8;
  if (!read_file_bytes(path, d8, err)) return false;
  std::vector<char> d(d8.begin(), d8.end());
  const std::string all(d.begin(), d.end());
  const size_t eh = all.find("end_header");
  if (eh == std::string::npos) { err = path + ": not a PLY file"; return false; }
  size_t body = eh + 10; if (body < all.size() && all[body] == '\r') ++body; if (body < all.size() && all[body] == '\n') ++body;
  struct Prop { std::string name, type, ltype, itype; bool list = false; };
  struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
  std::vector<Elem> elems; std::string fmt;
  { std::istringstream hs(all.substr(0, eh)); std::string line;
    while (std::getline(hs, line)) {
      std::istringstream ls(line); std::string t0; ls >> t0;
      if (t0 == "format") ls >> fmt;
      else if (t0 == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
      else if (t0 == "property" && !elems.empty()) { Prop p; ls >> p.type; if (p.type == "list") { p.list = true; ls >> p.ltype >> p.itype; } ls >> p.name; elems.back().props.push_back(p); }
    } }
  if (fmt != "binary_little_endian" && fmt != "ascii") { err = path + ": unsupported PLY format " + fmt; return false; }
  auto tsize = [](const std::string& t) -> size_t {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1; if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "double" || t == "float64") return 8; return 4; };
  const bool ascii = fmt == "ascii";
  std::istringstream as(ascii ? all.substr(body) : std::string());
  size_t off = body;
  auto get = [&](const std::string& t, double& v) -> bool {
    if (ascii) return (bool)(as >> v);
    const size_t n = tsize(t); if (off + n > d.size()) return false;
    const char* p = &d[off]; off += n;
    if (t == "float" || t == "float32") { float x; memcpy(&x, p, 4); v = x; } else if (t == "double" || t == "float64") { memcpy(&v, p, 8); }
    else if (t == "char" || t == "int8") { int8_t x; memcpy(&x, p, 1); v = x; } else if (t == "uchar" || t == "uint8") { uint8_t x; memcpy(&x, p, 1); v = x; }
    else if (t == "short" || t == "int16") { int16_t x; memcpy(&x, p, 2); v = x; } else if (t == "ushort" || t == "uint16") { uint16_t x; memcpy(&x, p, 2); v = x; }
    else if (t == "uint" || t == "uint32") { uint32_t x; memcpy(&x, p, 4); v = x; } else { int32_t x; memcpy(&x, p, 4); v = x; }
    return true; };
  bool has_n = false;
  for (const Elem& e : elems) {
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, is = -1, it = -1;
    for (size_t k = 0; k < e.props.size(); ++k) {
      const std::string& n = e.props[k].name;
      if (n == "x") ix = (int)k; else if (n == "y") iy = (int)k; else if (n == "z") iz = (int)k; else if (n == "nx") inx = (int)k; else if (n == "ny") iny = (int)k;
      else if (n == "nz") inz = (int)k; else if (n == "s" || n == "u" || n == "texture_u") is = (int)k; else if (n == "t" || n == "v" || n == "texture_v") it = (int)k;
    }
    // a file is untrusted input: a vertex needs all of x y z (normals all three, texture coordinates both), an element without
    // properties has no data to walk over, a list cannot be longer than what is left of the file
    const bool is_vertex = e.name == "vertex" && ix >= 0 && iy >= 0 && iz >= 0, with_n = inx >= 0 && iny >= 0 && inz >= 0, with_uv = is >= 0 && it >= 0;
    if (e.name == "vertex" && !is_vertex) { err = path + ": vertex element without x, y, z"; return false; }
    if (e.props.empty()) continue;
    for (size_t i = 0; i < e.count; ++i) {
      std::vector<double> vals(e.props.size(), 0.0);
      for (size_t k = 0; k < e.props.size(); ++k) {
        const Prop& p = e.props[k];
        if (!p.list) { if (!get(p.type, vals[k])) { err = path + ": truncated"; return false; } continue; }
        double cnt; if (!get(p.ltype, cnt)) { err = path + ": truncated"; return false; }
        if (!(cnt >= 0.0) || cnt > (double)d.size()) { err = path + ": bad list length"; return false; }
        std::vector<int32_t> idx((size_t)cnt);
        for (size_t j = 0; j < idx.size(); ++j) { double v; if (!get(p.itype, v)) { err = path + ": truncated"; return false; }
          if (!(v >= -2147483648.0 && v <= 2147483647.0)) { err = path + ": face index out of range"; return false; }
          idx[j] = (int32_t)v; }
        if (e.name == "face") for (size_t j = 1; j + 1 < idx.size(); ++j) { m.faces.push_back(idx[0]); m.faces.push_back(idx[j]); m.faces.push_back(idx[j + 1]); }
      }
      if (is_vertex) {
        m.pos.push_back((float)vals[ix]); m.pos.push_back((float)vals[iy]); m.pos.push_back((float)vals[iz]);
        if (with_n) { has_n = true; m.nrm.push_back((float)vals[inx]); m.nrm.push_back((float)vals[iny]); m.nrm.push_back((float)vals[inz]); }
        if (with_uv) { m.has_uv = true; m.uv.push_back((float)vals[is]); m.uv.push_back((float)vals[it]); }
      }
    }
  }
  const size_t nV = m.pos.size() / 3;
  for (int32_t i : m.faces) if (i < 0 || (size_t)i >= nV) { err = path + ": face index out of range"; return false; }
  if (!has_n) {
    // The file has no normals: the reference asks assimp for them (MeshImporter.cxx:80-87) -- aiProcess_GenSmoothNormals with -gensmooth (area-weighted
    // vertex normals), aiProcess_GenNormals without (ONE normal per face, vertices no longer shared between faces).  Same arithmetic as
    // cadrays_amd/scene_tcl.py read_ply: float edge vectors, double cross products, normalised in double.
    auto face_normal = [&](size_t t, double n[3]) {
      const float* a = &m.pos[3 * m.faces[t]]; const float* b = &m.pos[3 * m.faces[t + 1]]; const float* c = &m.pos[3 * m.faces[t + 2]];
      const double e1[3] = {(double)(b[0] - a[0]), (double)(b[1] - a[1]), (double)(b[2] - a[2])}, e2[3] = {(double)(c[0] - a[0]), (double)(c[1] - a[1]), (double)(c[2] - a[2])};
      n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    };
    if (smooth) {
      std::vector<double> acc(3 * nV, 0.0);
      for (size_t t = 0; t + 2 < m.faces.size(); t += 3) {
        double n[3]; face_normal(t, n);
        for (int k = 0; k < 3; ++k) for (int x = 0; x < 3; ++x) acc[3 * m.faces[t + k] + x] += n[x];
      }
      m.nrm.resize(3 * nV);
      for (size_t v = 0; v < nV; ++v) { const double l = std::max(std::sqrt(acc[3 * v] * acc[3 * v] + acc[3 * v + 1] * acc[3 * v + 1] + acc[3 * v + 2] * acc[3 * v + 2]), 1e-30); for (int x = 0; x < 3; ++x) m.nrm[3 * v + x] = (float)(acc[3 * v + x] / l); }
    } else {
      const size_t nF = m.faces.size() / 3;
      std::vector<float> pos(9 * nF), nrm(9 * nF), uv(m.has_uv ? 6 * nF : 0);
      for (size_t f = 0; f < nF; ++f) {
        double n[3]; face_normal(3 * f, n);
        const double l = std::max(std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-30);
        for (int k = 0; k < 3; ++k) {
          const int32_t v = m.faces[3 * f + k];
          for (int x = 0; x < 3; ++x) { pos[9 * f + 3 * k + x] = m.pos[3 * v + x]; nrm[9 * f + 3 * k + x] = (float)(n[x] / l); }
          if (m.has_uv) { uv[6 * f + 2 * k] = m.uv[2 * v]; uv[6 * f + 2 * k + 1] = m.uv[2 * v + 1]; }
        }
      }
      m.pos.swap(pos); m.nrm.swap(nrm); if (m.has_uv) m.uv.swap(uv);
      for (size_t i = 0; i < 3 * nF; ++i) m.faces[i] = (int32_t)i;
    }
  }
  return true;
}

struct Fresnel { float v[4]; };
inline double clip(double x, double lo, double hi) { return std::min(std::max(x, lo), hi); }
inline Fresnel fr_constant(double f) { return {{-1.0f, 0.f, (float)clip(f, 0, 1), 0.f}}; }                          // MaterialEditor.cxx:209-255
inline Fresnel fr_schlick(double r, double g, double b) { return {{(float)clip(r, 0, 1), (float)clip(g, 0, 1), (float)clip(b, 0, 1), 0.f}}; }
inline Fresnel fr_conductor(double n, double k) { return {{-2.0f, (float)clip(n, 1e-2, 1e3), (float)clip(k, 1e-2, 1e3), 0.f}}; }
inline Fresnel fr_dielectric(double n) { return {{-3.0f, (float)clip(n, 1.0, 1e3), 0.f, 0.f}}; }

struct Bsdf {
  float Kc[4] = {0, 0, 0, 0}, Kd[3] = {0, 0, 0}, Ks[4] = {0, 0, 0, 0}, Kt[3] = {0, 0, 0}, Le[3] = {0, 0, 0}, Ab[4] = {0, 0, 0, 0};
  Fresnel coat = fr_constant(0.0), base = fr_constant(1.0);
  static Bsdf diffuse(float k) { Bsdf b; b.Kd[0] = b.Kd[1] = b.Kd[2] = k; return b; }
  static Bsdf metallic(float w, Fresnel f, float rough) { Bsdf b; b.Ks[0] = b.Ks[1] = b.Ks[2] = w; b.Ks[3] = rough; b.base = f; return b; }
  static Bsdf glass(float w, float ar, float ag, float ab, float coeff, double ior)
  { Bsdf b; b.coat = fr_dielectric(ior); b.Kt[0] = b.Kt[1] = b.Kt[2] = w; b.Kc[0] = b.Kc[1] = b.Kc[2] = 1.f; b.Ab[0] = ar; b.Ab[1] = ag; b.Ab[2] = ab; b.Ab[3] = coeff; return b; }
  void normalize()       // MaterialEditor.cxx:311-329
  {
    float m = 0.f; for (int k = 0; k < 3; ++k) m = std::max(m, Kd[k] + Ks[k] + Kt[k]);
    if (m > 1.0f) for (int k = 0; k < 3; ++k) { Kd[k] /= m; Ks[k] /= m; Kt[k] /= m; }
  }
};

// Stand-ins for OCCT's 24 stock materials [OCCT-ext], fitted in round 6 to the icons OCCT rendered for them (data/materials/*.png of the reference;
// tests/golden/icon_features.json): the same table as cadrays_amd/scene_tcl.py (_STOCK_*), where the evidence for each group is written down.
inline Bsdf stock_material(const std::string& name)
{
  const std::string n = lower(name);
  struct Metal { const char* name; float r, g, b, rough; };
  static const Metal metals[] = {{"brass", 0.63f, 0.46f, 0.2f, 0.02f}, {"bronze", 0.7f, 0.39f, 0.16f, 0.02f}, {"copper", 0.94f, 0.64f, 0.47f, 0.045f}, {"gold", 0.97f, 0.75f, 0.3f, 0.05f}, {"pewter", 0.65f, 0.63f, 0.54f, 0.04f}, {"silver", 0.95f, 0.9f, 0.75f, 0.065f}, {"steel", 0.57f, 0.53f, 0.45f, 0.035f}, {"chrome", 0.59f, 0.57f, 0.49f, 0.04f}, {"aluminium", 0.9f, 0.87f, 0.75f, 0.06f}, {"aluminum", 0.9f, 0.87f, 0.75f, 0.06f}, {"metalized", 0.25f, 0.25f, 0.22f, 0.02f}};
  struct Glass { const char* name; float r, g, b, coeff; double ior; };
  static const Glass glasses[] = {{"glass", 0.75f, 0.95f, 0.9f, 0.05f, 1.62}, {"water", 0.7f, 0.75f, 0.85f, 0.05f, 1.33}, {"diamond", 0.95f, 0.95f, 0.95f, 0.05f, 2.42}};
  struct Matte { const char* name; float r, g, b, ks, rough; };
  static const Matte mattes[] = {{"plaster", 0.53f, 0.52f, 0.49f, 0.0f, 0.0f}, {"stone", 0.28f, 0.27f, 0.26f, 0.0f, 0.0f}, {"charcoal", 0.12f, 0.12f, 0.115f, 0.0f, 0.0f}, {"satin", 0.66f, 0.65f, 0.62f, 0.04f, 0.35f}, {"plastic", 0.22f, 0.22f, 0.21f, 0.04f, 0.2f}, {"shiny_plastic", 0.33f, 0.33f, 0.31f, 0.05f, 0.08f}, {"jade", 0.25f, 0.45f, 0.24f, 0.04f, 0.2f}, {"obsidian", 0.035f, 0.014f, 0.033f, 0.05f, 0.1f}, {"neon_gnc", 0.32f, 0.33f, 0.36f, 0.05f, 0.1f}};
  for (const Glass& g : glasses) if (n == g.name) return Bsdf::glass(1.f, g.r, g.g, g.b, g.coeff, g.ior);
  if (n == "transparent") { Bsdf b = Bsdf::diffuse(0.15f); b.Kt[0] = b.Kt[1] = b.Kt[2] = 0.8f; return b; }      // index-matched: the icon's tile edges run straight through the ball
  for (const Metal& m : metals) if (n == m.name) return Bsdf::metallic(1.f, fr_schlick(m.r, m.g, m.b), m.rough);
  if (n == "neon_phc") { Bsdf b; b.Kd[0] = 0.f; b.Kd[1] = 0.3f; b.Kd[2] = 0.2f; b.Le[0] = 0.f; b.Le[1] = 0.9f; b.Le[2] = 0.55f; return b; }      // the one emissive preset
  for (const Matte& d : mattes) if (n == d.name) {
    Bsdf b; b.Kd[0] = d.r; b.Kd[1] = d.g; b.Kd[2] = d.b;
    if (d.ks > 0.f) { b.Ks[0] = b.Ks[1] = b.Ks[2] = d.ks; b.Ks[3] = d.rough; b.base = fr_constant(1.0); }
    return b;
  }
  return Bsdf::diffuse(0.8f);
}

struct Object {
  Mesh mesh; Bsdf bsdf = Bsdf::diffuse(0.8f); bool displayed = false;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, s = 1.0, t[3] = {0, 0, 0};
  std::string texture; bool tex_on = true;
};

// one command line -> words: "quoted strings", {braced words}, [bracketed] kept as one word, $Root / ${Root} substituted, # comments
inline std::vector<std::string> split_words(const std::string& line, const std::string& root)
{
  std::vector<std::string> w; std::string cur; bool in = false; size_t i = 0;
  auto flush = [&] { if (in) { w.push_back(cur); cur.clear(); in = false; } };
  while (i < line.size()) {
    const char c = line[i];
    if (c == '#' && !in && w.empty()) break;
    if (c == ' ' || c == '\t' || c == '\r' || c == ';') { flush(); ++i; continue; }
    in = true;
    if (c == '"') { ++i; while (i < line.size() && line[i] != '"') { if (line[i] == '$') goto subst; cur += line[i++]; continue; subst: { size_t j = i + 1; bool br = j < line.size() && line[j] == '{'; if (br) ++j; size_t k = j; while (k < line.size() && (isalnum((unsigned char)line[k]) || line[k] == '_')) ++k; const std::string v = line.substr(j, k - j); cur += (v == "Root") ? root : ("$" + v); i = br && k < line.size() && line[k] == '}' ? k + 1 : k; } } ++i; continue; }
    if (c == '{' || c == '[') { const char cl = c == '{' ? '}' : ']'; int depth = 0; size_t j = i; for (; j < line.size(); ++j) { if (line[j] == c) ++depth; else if (line[j] == cl && --depth == 0) break; } cur += line.substr(i + (c == '{' ? 1 : 0), j - i - (c == '{' ? 1 : -1)); i = j + 1; continue; }
    if (c == '$') { size_t j = i + 1; const bool br = j < line.size() && line[j] == '{'; if (br) ++j; size_t k = j; while (k < line.size() && (isalnum((unsigned char)line[k]) || line[k] == '_')) ++k; const std::string v = line.substr(j, k - j); cur += (v == "Root") ? root : ("$" + v); i = br && k < line.size() && line[k] == '}' ? k + 1 : k; continue; }
    cur += c; ++i;
  }
  flush();
  return w;
}

}  // namespace detail

// Evaluate model.tcl at `path` for a width x height target.  false + err on the first command that cannot be evaluated.
inline bool read_model_tcl(const std::string& path, uint32_t width, uint32_t height, TclScene& out, std::string& err)
{
  using namespace detail;
  std::vector<uint8_t> script;
  if (!read_file_bytes(path, script, err)) return false;
  std::istringstream f(std::string(script.begin(), script.end()));
  std::string root = "."; { const size_t sl = path.find_last_of('/'); if (sl != std::string::npos) root = path.substr(0, sl); }
  if (root.empty()) root = "/";
  std::vector<std::string> order; std::map<std::string, Object> objs;
  struct L { std::string kind; double vec[3]; double sm = 0, inten = 1; int head = 0; double color[3] = {1, 1, 1}; bool alive = true; };
  // a fresh viewer owns a directional headlight (0) and an ambient light (1) [OCCT-ext]; exported scenes start with `vlight clear`
  std::vector<L> lights; { L h; h.kind = "directional"; h.vec[0] = 0; h.vec[1] = 0; h.vec[2] = -1; h.head = 1; lights.push_back(h); L a; a.kind = "ambient"; a.vec[0] = a.vec[1] = a.vec[2] = 0; lights.push_back(a); }
  struct { bool has_eye = false, has_at = false, has_proj = false, ortho = false; double eye[3], at[3], up[3] = {0, 0, 1}, proj[3] = {0, -1, 0}, fovy = 45.0, size = 0.0; } cam;
  int depth = 5; std::string env_path;
  auto num = [&](const std::string& s, double& v) { if (!is_number(s)) return false; v = strtod(s.c_str(), nullptr); return true; };
  auto fail = [&](int ln, const std::string& m) { err = path + ":" + std::to_string(ln) + ": " + m; return false; };

  std::string line, pending; int ln = 0;
  while (std::getline(f, line)) {
    ++ln;
    if (!line.empty() && line.back() == '\\') { pending += line.substr(0, line.size() - 1) + " "; continue; }
    line = pending + line; pending.clear();
    std::vector<std::string> a = split_words(line, root);
    if (a.empty()) continue;
    const std::string cmd = a[0]; a.erase(a.begin());
    a.erase(std::remove(a.begin(), a.end(), std::string("-noupdate")), a.end());
    if (cmd == "variable" || cmd == "set") { if (!a.empty() && a[0] == "Root" && a.size() > 1 && a[1].find("info script") == std::string::npos && a[1].find("file") == std::string::npos) root = a[1]; continue; }
    if (cmd == "rtmeshread") {
      if (a.size() < 2) return fail(ln, "usage: rtmeshread <file name> <node name> [options]");
      if (objs.count(a[1])) return fail(ln, "Error: Mesh with the name '" + a[1] + "' already exists");
      const std::string ext = lower(a[0].size() > 4 ? a[0].substr(a[0].size() - 4) : "");
      if (ext != ".ply") return fail(ln, "rtmeshread: this reader loads PLY meshes (what the exporter writes); " + a[0]);
      bool smooth = false;
      for (size_t i = 2; i < a.size(); ++i) { const std::string k = lower(a[i]); if (k == "-up") { if (i + 1 < a.size() && lower(a[i + 1]) != "z") out.unsupported.push_back("rtmeshread -up " + a[i + 1]); ++i; } else if (k == "-gensmooth" || k == "-gs") smooth = true; else if (k == "-fixnorms" || k == "-fn") out.unsupported.push_back("rtmeshread " + a[i]); }
      Object o; std::string e2;
      if (!read_ply(a[0], o.mesh, e2, smooth)) return fail(ln, e2);
      o.displayed = true;
      objs[a[1]] = std::move(o); order.push_back(a[1]);
    } else if (cmd == "vdisplay" || cmd == "verase" || cmd == "rtdisplay" || cmd == "rterase") {      // rtdisplay / rterase: the data model's Show / Hide (ImportExportPlugin.cxx:373-425)
      for (const std::string& n : a) { auto it = objs.find(n); if (it != objs.end()) it->second.displayed = cmd == "vdisplay" || cmd == "rtdisplay"; } }
    else if (cmd == "vclear") { for (auto& kv : objs) kv.second.displayed = false; }
    else if (cmd == "vsetmaterial") { if (a.size() < 2 || !objs.count(a[0])) return fail(ln, "vsetmaterial: unknown object"); objs[a[0]].bsdf = stock_material(a[1]); }
    else if (cmd == "vbsdf") {
      if (a.empty() || !objs.count(a[0])) return fail(ln, "vbsdf: unknown object");
      Bsdf& b = objs[a[0]].bsdf; size_t i = 1;
      auto take = [&](size_t n, double* v) { size_t got = 0; while (got < n && i + 1 < a.size() && is_number(a[i + 1])) { v[got++] = strtod(a[i + 1].c_str(), nullptr); ++i; } if (got == 1 && n == 3) { v[1] = v[2] = v[0]; got = 3; } return got == n; };
      while (i < a.size()) {
        const std::string k = lower(a[i]); double v[3];
        if (k == "-kc") { if (!take(3, v)) return fail(ln, "vbsdf -Kc expects 3 values"); for (int x = 0; x < 3; ++x) b.Kc[x] = (float)v[x]; }
        else if (k == "-kd") { if (!take(3, v)) return fail(ln, "vbsdf -Kd expects 3 values"); for (int x = 0; x < 3; ++x) b.Kd[x] = (float)v[x]; }
        else if (k == "-ks") { if (!take(3, v)) return fail(ln, "vbsdf -Ks expects 3 values"); for (int x = 0; x < 3; ++x) b.Ks[x] = (float)v[x]; }
        else if (k == "-kt") { if (!take(3, v)) return fail(ln, "vbsdf -Kt expects 3 values"); for (int x = 0; x < 3; ++x) b.Kt[x] = (float)v[x]; }
        else if (k == "-le") { if (!take(3, v)) return fail(ln, "vbsdf -Le expects 3 values"); for (int x = 0; x < 3; ++x) b.Le[x] = (float)v[x]; }
        else if (k == "-baseroughness") { if (!take(1, v)) return fail(ln, "vbsdf -baseRoughness expects 1 value"); b.Ks[3] = (float)v[0]; }
        else if (k == "-coatroughness") { if (!take(1, v)) return fail(ln, "vbsdf -coatRoughness expects 1 value"); b.Kc[3] = (float)v[0]; }
        else if (k == "-absorpcolor" || k == "-absorptioncolor") { if (!take(3, v)) return fail(ln, "vbsdf -absorpColor expects 3 values"); for (int x = 0; x < 3; ++x) b.Ab[x] = (float)v[x]; }
        else if (k == "-absorpcoeff" || k == "-absorptioncoeff") { if (!take(1, v)) return fail(ln, "vbsdf -absorpCoeff expects 1 value"); b.Ab[3] = (float)v[0]; }
        else if (k == "-basefresnel" || k == "-coatfresnel") {
          if (i + 1 >= a.size()) return fail(ln, "vbsdf: Fresnel model expected");
          const std::string kind = lower(a[++i]); Fresnel fr;
          if (kind == "constant") { if (!take(1, v)) return fail(ln, "Fresnel Constant expects 1 value"); fr = fr_constant(v[0]); }
          else if (kind == "schlick") { if (!take(3, v)) return fail(ln, "Fresnel Schlick expects 3 values"); fr = fr_schlick(v[0], v[1], v[2]); }
          else if (kind == "conductor") { if (!take(2, v)) return fail(ln, "Fresnel Conductor expects 2 values"); fr = fr_conductor(v[0], v[1]); }
          else if (kind == "dielectric") { if (!take(1, v)) return fail(ln, "Fresnel Dielectric expects 1 value"); fr = fr_dielectric(v[0]); }
          else return fail(ln, "vbsdf: unknown Fresnel model " + kind);
          (k == "-basefresnel" ? b.base : b.coat) = fr;
        }
        else if (k == "-n" || k == "-normalize") b.normalize();
        else return fail(ln, "vbsdf: unknown option " + a[i]);
        ++i;
      }
    } else if (cmd == "rttexture") {
      if (a.size() < 2 || !objs.count(a[0])) return fail(ln, "rttexture: no such object");
      Object& o = objs[a[0]]; size_t i = 1;
      while (i < a.size()) {
        const std::string k = lower(a[i]);
        if (k == "-scale") { i += 3; }                       // meshes keep their own uv; -scale re-parametrises CAD shapes only (DataNode.cxx:219-222)
        else if (k == "-on" || k == "-off") { o.tex_on = k == "-on"; ++i; }
        else { o.texture = a[1]; ++i; }
      }
    } else if (cmd == "vlocation") {
      if (a.empty() || !objs.count(a[0])) return fail(ln, "vlocation: unknown object");
      Object& o = objs[a[0]]; size_t i = 1; double v[4];
      while (i < a.size()) {
        const std::string k = lower(a[i]);
        auto need = [&](size_t n) { if (i + n >= a.size()) return false; for (size_t j = 0; j < n; ++j) if (!num(a[i + 1 + j], v[j])) return false; return true; };
        if (k == "-location" || k == "-setlocation") { if (!need(3)) return fail(ln, "vlocation -location expects 3 values"); for (int x = 0; x < 3; ++x) o.t[x] = v[x]; i += 4; }
        else if (k == "-rotation" || k == "-setrotation") {
          if (!need(4)) return fail(ln, "vlocation -rotation expects a quaternion");
          double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]); if (n == 0) n = 1; const double x = v[0] / n, y = v[1] / n, z = v[2] / n, w = v[3] / n;
          const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
          memcpy(o.R, R, sizeof R); i += 5;
        }
        else if (k == "-scale" || k == "-setscale") { if (!need(1)) return fail(ln, "vlocation -scale expects 1 value"); o.s = v[0]; i += 2; }
        else if (k == "-reset") { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(o.R, I, sizeof I); o.s = 1; o.t[0] = o.t[1] = o.t[2] = 0; ++i; }
        else return fail(ln, "vlocation: option " + a[i] + " is not part of the exported format");
      }
    } else if (cmd == "vcamera") {
      for (size_t i = 0; i < a.size(); ++i) { const std::string k = lower(a[i]); double v;
        if (k == "-persp" || k == "-perspective") cam.ortho = false; else if (k == "-ortho" || k == "-orthographic") cam.ortho = true;
        else if (k == "-fovy" && i + 1 < a.size() && num(a[i + 1], v)) { cam.fovy = v; ++i; } else if (k == "-distance") ++i; }
    } else if (cmd == "vviewparams") {
      for (size_t i = 0; i < a.size();) { const std::string k = lower(a[i]); double v[3];
        if ((k == "-proj" || k == "-up" || k == "-at" || k == "-eye") && i + 3 < a.size() && num(a[i + 1], v[0]) && num(a[i + 2], v[1]) && num(a[i + 3], v[2])) {
          double* dst = k == "-proj" ? cam.proj : k == "-up" ? cam.up : k == "-at" ? cam.at : cam.eye; memcpy(dst, v, sizeof v);
          if (k == "-at") cam.has_at = true; else if (k == "-eye") cam.has_eye = true; else if (k == "-proj") cam.has_proj = true; i += 4; }
        else if ((k == "-size" || k == "-scale") && i + 1 < a.size() && num(a[i + 1], v[0])) { if (k == "-size") cam.size = v[0]; i += 2; }
        else ++i; }
    } else if (cmd == "vtextureenv") { env_path = (a.size() > 1 && lower(a[0]) == "on") ? a[1] : ""; }
    else if (cmd == "vlight") {
      if (a.empty()) continue;
      const std::string op = lower(a[0]); L* l = nullptr; size_t i = 0;
      if (op == "clear") { lights.clear(); continue; }
      if (op == "del" || op == "delete") { if (a.size() > 1) { const int idx = atoi(a[1].c_str()); if (idx >= 0 && (size_t)idx < lights.size()) lights[(size_t)idx].alive = false; } continue; }
      if (op == "add") { if (a.size() < 2) return fail(ln, "vlight add: light type expected"); L n; n.kind = lower(a[1]); n.vec[0] = n.vec[1] = 0; n.vec[2] = n.kind == "directional" ? -1 : 0; lights.push_back(n); l = &lights.back(); i = 2; }
      else if (op == "change") { if (a.size() < 2 || atoi(a[1].c_str()) < 0 || (size_t)atoi(a[1].c_str()) >= lights.size()) return fail(ln, "vlight change: no such light"); l = &lights[(size_t)atoi(a[1].c_str())]; i = 2; }
      else return fail(ln, "vlight: unknown operation " + a[0]);
      while (i < a.size()) { std::string k = lower(a[i]); while (!k.empty() && k[0] == '-') k.erase(0, 1); double v[3];
        if ((k == "direction" || k == "dir" || k == "pos" || k == "position") && i + 3 < a.size() && num(a[i + 1], v[0]) && num(a[i + 2], v[1]) && num(a[i + 3], v[2])) { memcpy(l->vec, v, sizeof v); i += 4; }
        else if ((k == "sm" || k == "smoothness") && i + 1 < a.size() && num(a[i + 1], v[0])) { l->sm = v[0]; i += 2; }
        else if ((k == "int" || k == "intensity") && i + 1 < a.size() && num(a[i + 1], v[0])) { l->inten = v[0]; i += 2; }
        else if ((k == "head" || k == "headlight") && i + 1 < a.size()) { l->head = atoi(a[i + 1].c_str()); i += 2; }
        else if (k == "color" || k == "colour") i += 2;
        else return fail(ln, "vlight: unknown parameter " + a[i]); }
    } else if (cmd == "rtlight") {
      if (a.size() >= 5 && lower(a[1]) == "-color") { const int idx = atoi(a[0].c_str()); double v[3];
        if (idx >= 0 && (size_t)idx < lights.size() && num(a[2], v[0]) && num(a[3], v[1]) && num(a[4], v[2])) memcpy(lights[(size_t)idx].color, v, sizeof v); }
    } else if (cmd == "vrenderparams") { for (size_t i = 0; i + 1 < a.size(); ++i) if (lower(a[i]) == "-raydepth") depth = atoi(a[i + 1].c_str()); }
    else if (cmd == "rtmodel" || cmd == "rtgroup" || cmd == "vupdate" || cmd == "vrepaint" || cmd == "vsetdispmode" || cmd == "vaspects" || cmd == "vvbo" ||
             cmd == "vfit" || cmd == "vselect" || cmd == "vzbufftrihedron" || cmd == "vsetcolor" || cmd == "vinit" || cmd == "pload") { /* no effect on the path */ }
    else out.unsupported.push_back(cmd + " (line " + std::to_string(ln) + ")");
  }

  // ---- snapshot: displayed objects in script order, one material per object
  std::map<std::string, uint32_t> slots; bool any_tex = false;
  for (const std::string& name : order) { const Object& o = objs[name]; if (o.displayed && !o.mesh.faces.empty() && !o.texture.empty() && o.tex_on && o.mesh.has_uv) any_tex = true; }
  uint32_t nv = 0;
  for (const std::string& name : order) {
    const Object& o = objs[name];
    if (!o.displayed || o.mesh.faces.empty()) continue;
    const size_t n = o.mesh.pos.size() / 3; const uint32_t mat = (uint32_t)out.mats.size();
    for (size_t v = 0; v < n; ++v) {
      const double p[3] = {(double)o.mesh.pos[3 * v] * o.s, (double)o.mesh.pos[3 * v + 1] * o.s, (double)o.mesh.pos[3 * v + 2] * o.s};
      const double q[3] = {(double)o.mesh.nrm[3 * v], (double)o.mesh.nrm[3 * v + 1], (double)o.mesh.nrm[3 * v + 2]};
      for (int r = 0; r < 3; ++r) {
        out.pos.push_back((float)((p[0] * o.R[3 * r] + p[1] * o.R[3 * r + 1]) + p[2] * o.R[3 * r + 2] + o.t[r]));
        out.nrm.push_back((float)((q[0] * o.R[3 * r] + q[1] * o.R[3 * r + 1]) + q[2] * o.R[3 * r + 2]));
      }
      if (any_tex) { out.uv.push_back(o.mesh.has_uv ? o.mesh.uv[2 * v] : 0.f); out.uv.push_back(o.mesh.has_uv ? o.mesh.uv[2 * v + 1] : 0.f); }
    }
    for (size_t t = 0; t + 2 < o.mesh.faces.size(); t += 3) { for (int k = 0; k < 3; ++k) out.tri.push_back(o.mesh.faces[t + k] + (int32_t)nv); out.tri.push_back((int32_t)mat); }
    nv += (uint32_t)n;
    crh_bsdf m{}; const Bsdf& b = o.bsdf; int slot = -1;
    if (!o.texture.empty() && o.tex_on) {
      if (!o.mesh.has_uv) out.unsupported.push_back("rttexture " + name + ": object has no texture coordinates");
      else { auto it = slots.find(o.texture);
        if (it == slots.end()) { Texture t; std::string e2; if (!load_texture(o.texture, t, e2)) { err = e2; return false; } slots[o.texture] = (uint32_t)out.textures.size(); out.textures.push_back(std::move(t)); it = slots.find(o.texture); }
        slot = (int)it->second; }
    }
    for (int k = 0; k < 4; ++k) { m.Kc[k] = b.Kc[k]; m.Ks[k] = b.Ks[k]; m.Absorption[k] = b.Ab[k]; m.FresnelCoat[k] = b.coat.v[k]; m.FresnelBase[k] = b.base.v[k]; }
    for (int k = 0; k < 3; ++k) { m.Kd[k] = b.Kd[k]; m.Kt[k] = b.Kt[k]; m.Le[k] = b.Le[k]; }
    m.Kd[3] = (float)(slot + 1); m.Kt[3] = slot >= 0 ? 1.0f : 0.0f; m.Le[3] = slot >= 0 ? 1.0f : 0.0f;
    out.mats.push_back(m);
  }
  if (out.pos.empty()) { err = path + ": no displayed geometry"; return false; }

  double eye[3], at[3];
  if (cam.has_eye && cam.has_at) { memcpy(eye, cam.eye, sizeof eye); memcpy(at, cam.at, sizeof at); }
  else {                                   // vfit: frame the bounding sphere along the projection direction
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t v = 0; v < out.pos.size(); v += 3) for (int x = 0; x < 3; ++x) { lo[x] = std::min(lo[x], (double)out.pos[v + x]); hi[x] = std::max(hi[x], (double)out.pos[v + x]); }
    double r = 0; for (int x = 0; x < 3; ++x) { at[x] = (lo[x] + hi[x]) / 2; r += (hi[x] - lo[x]) * (hi[x] - lo[x]); } r = std::sqrt(r) / 2;
    double half = cam.fovy * M_PI / 180.0 / 2; half = std::min(half, std::atan(std::tan(half) * width / height));
    const double pl = std::sqrt(cam.proj[0] * cam.proj[0] + cam.proj[1] * cam.proj[1] + cam.proj[2] * cam.proj[2]);
    for (int x = 0; x < 3; ++x) eye[x] = at[x] + cam.proj[x] / pl * (r / std::sin(half));
  }
  crh_camera& c = out.cam; memset(&c, 0, sizeof c);
  for (int x = 0; x < 3; ++x) { c.eye[x] = (float)eye[x]; c.dir[x] = (float)(at[x] - eye[x]); c.up[x] = (float)cam.up[x]; }
  c.fovy_deg = (float)cam.fovy; c.aspect = 0.f; c.is_ortho = cam.ortho ? 1 : 0; c.ortho_scale = (float)((cam.size != 0.0 ? cam.size : 2.0) / 2); c.aperture_radius = 0.f; c.focal_dist = 1.0f;

  for (const L& l : lights) {
    if (!l.alive || (l.kind != "directional" && l.kind != "positional")) continue;      // ambient / spot: ignored by the path tracer (LightSourcesEditor.cxx:157-178)
    double vec[3] = {l.vec[0], l.vec[1], l.vec[2]};
    if (l.head) {                          // headlight: given in eye space (x right, y up, z towards the viewer)
      double fw[3], fl = 0; for (int x = 0; x < 3; ++x) { fw[x] = at[x] - eye[x]; fl += fw[x] * fw[x]; } fl = std::sqrt(fl); if (fl == 0) fl = 1; for (double& x : fw) x /= fl;
      double rt[3] = {fw[1] * cam.up[2] - fw[2] * cam.up[1], fw[2] * cam.up[0] - fw[0] * cam.up[2], fw[0] * cam.up[1] - fw[1] * cam.up[0]};
      double rl = std::sqrt(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]); if (rl == 0) rl = 1; for (double& x : rt) x /= rl;
      const double up[3] = {rt[1] * fw[2] - rt[2] * fw[1], rt[2] * fw[0] - rt[0] * fw[2], rt[0] * fw[1] - rt[1] * fw[0]};
      for (int x = 0; x < 3; ++x) vec[x] = l.vec[0] * rt[x] + l.vec[1] * up[x] - l.vec[2] * fw[x] + (l.kind == "positional" ? eye[x] : 0.0);
    }
    crh_light o{}; for (int x = 0; x < 3; ++x) { o.vec[x] = (float)vec[x]; o.emission[x] = (float)l.color[x] * (float)l.inten; }
    o.is_point = l.kind == "positional" ? 1.0f : 0.0f; o.smoothness = (float)l.sm;
    out.lights.push_back(o);
  }
  if (!env_path.empty()) {
    Texture t; std::string e2;
    if (load_texture(env_path, t, e2)) { out.envW = t.w; out.envH = t.h; out.env.resize(3 * (size_t)t.w * t.h); for (size_t i = 0; i < (size_t)t.w * t.h; ++i) for (int k = 0; k < 3; ++k) out.env[3 * i + k] = t.texels[t.ch * i + k]; }
    else out.unsupported.push_back("vtextureenv: " + e2);
  }
  crh_params& p = out.par; memset(&p, 0, sizeof p);      // Graphic3d_RenderingParams defaults of the Python mirror (cadrays_amd/scenes.py Params)
  p.width = width; p.height = height; p.max_depth = (uint32_t)depth; p.two_sided = 1; p.seed = 1; p.tile_size = 32; p.white_point = 1.0f; p.env_as_background = 0 /* OCCT's default; the reference's icons show it (scene_tcl.py snapshot) */; p.russian_roulette = 1;
  return true;
}

}  // namespace crh_host
