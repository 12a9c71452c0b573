// jpeg_baseline.hpp -- 8-bit Huffman JPEG reader (baseline and progressive) for the C++ scene reader.
// Why it exists: the environment map the reference ships and loads by default is a JPEG (data/maps/default.jpg, loaded at
// reference src/Launcher/AppGui.cxx:963; the GUI's file filters are png/jpg, LightSourcesEditor.cxx:348,388), so a scene exported
// by CADRays normally references one.  The Python reader decodes it with Pillow (libjpeg-turbo); this decoder restates the same
// arithmetic -- the "islow" integer inverse DCT (13-bit constants, two passes), fancy (triangle) chroma upsampling for 2x1 / 2x2
// subsampling and the 16-bit fixed-point YCbCr -> RGB tables of the IJG specification -- so that both hosts hand identical
// texels to the boundary (tests/test_scene_tcl.py::test_cpp_jpeg_reader_matches_pillow).
// Read: sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection + successive approximation) Huffman JPEG with 1 or 3
// components, sampling factors 1 or 2 (chroma 1x1), interleaved and per-component scans, restart intervals, Adobe APP14 transform
// flag.  Refused with a message: arithmetic coding, lossless / hierarchical, 12-bit, CMYK.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <cstdio>
#include <sys/stat.h>

namespace crh_host {
namespace detail {

// whole file -> bytes; false (with a message) for anything that is not a readable regular file -- a directory opens fine as an ifstream
// and only throws once it is read
inline bool read_file_bytes(const std::string& path, std::vector<uint8_t>& d, std::string& err)
{
  struct stat st;
  if (path.find('\0') != std::string::npos || stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { err = "cannot open " + std::string(path.c_str()); return false; }
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return false; }
  d.resize((size_t)st.st_size);
  const size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
  fclose(f);
  d.resize(got);
  return true;
}

struct JpegHuff { uint8_t bits[17] = {0}; uint8_t vals[256] = {0}; int mincode[17], maxcode[18], valptr[17]; bool set = false; };

inline void jpeg_build_huff(JpegHuff& h)
{
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k; h.mincode[l] = code;
    code += h.bits[l]; k += h.bits[l];
    h.maxcode[l] = h.bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff; h.set = true;
}

struct JpegBits {
  const uint8_t* p; const uint8_t* end; uint32_t acc = 0; int n = 0; bool hit_marker = false;
  int bit()
  {
    if (n == 0) {
      uint8_t b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) { const uint8_t b2 = p < end ? *p : 0; if (b2 == 0) ++p; else { hit_marker = true; --p; b = 0; } }
      }
      acc = b; n = 8;
    }
    --n; return (acc >> n) & 1;
  }
  int receive(int s) { int v = 0; for (int i = 0; i < s; ++i) v = (v << 1) | bit(); return v; }
  void reset() { n = 0; acc = 0; hit_marker = false; }
};

inline int jpeg_decode_sym(JpegBits& br, const JpegHuff& h)
{
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
inline int jpeg_extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// islow inverse DCT of one dequantised block (natural order) -> 8x8 samples.  64-bit intermediates: a valid stream never leaves the
// 32-bit range the IJG code works in (same results), a corrupt one must not run into signed overflow
inline void jpeg_idct_islow(const long long* in, uint8_t* out, int out_stride)
{
  typedef long long I;
  constexpr int CB = 13, P1 = 2;
  constexpr I F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
                F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
  I ws[64];
  auto descale = [](I x, int n) { return (x + ((I)1 << (n - 1))) >> n; };
  for (int c = 0; c < 8; ++c) {
    const I* i = in + c;
    I z2 = i[16], z3 = i[48];
    I z1 = (z2 + z3) * F_0_541;
    I tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
    z2 = i[0]; z3 = i[32];
    I tmp0 = (z2 + z3) * ((I)1 << CB), tmp1 = (z2 - z3) * ((I)1 << CB);
    const I tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = i[56]; tmp1 = i[40]; tmp2 = i[24]; tmp3 = i[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; I z4 = tmp1 + tmp3;
    const I z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    I* w = ws + c;
    w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
    w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
    w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
    w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
  }
  auto clamp8 = [](I v) { v += 128; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  for (int r = 0; r < 8; ++r) {
    const I* w = ws + 8 * r;
    I z2 = w[2], z3 = w[6];
    I z1 = (z2 + z3) * F_0_541;
    I tmp2 = z1 + z3 * (-F_1_847), tmp3 = z1 + z2 * F_0_765;
    I tmp0 = (w[0] + w[4]) * ((I)1 << CB), tmp1 = (w[0] - w[4]) * ((I)1 << CB);
    const I tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; I z4 = tmp1 + tmp3;
    const I z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    uint8_t* o = out + (size_t)out_stride * r; constexpr int S = CB + P1 + 3;
    o[0] = clamp8(descale(tmp10 + tmp3, S)); o[7] = clamp8(descale(tmp10 - tmp3, S));
    o[1] = clamp8(descale(tmp11 + tmp2, S)); o[6] = clamp8(descale(tmp11 - tmp2, S));
    o[2] = clamp8(descale(tmp12 + tmp1, S)); o[5] = clamp8(descale(tmp12 - tmp1, S));
    o[3] = clamp8(descale(tmp13 + tmp0, S)); o[4] = clamp8(descale(tmp13 - tmp0, S));
  }
}

// out: RGB, 3 bytes per pixel
inline bool read_jpeg(const std::string& path, uint32_t& W, uint32_t& H, std::vector<uint8_t>& out, std::string& err)
{
  std::vector<uint8_t> d;
  if (!read_file_bytes(path, d, err)) return false;
  if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = path + ": not a JPEG file"; return false; }
  static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  uint16_t qt[4][64] = {{0}}; JpegHuff hdc[4], hac[4];
  struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0, rw = 0, rh = 0; std::vector<int16_t> coef; std::vector<uint8_t> plane; };
  Comp comp[3]; int nc = 0, restart = 0, adobe_transform = -1, hmax = 1, vmax = 1, mcux = 0, mcuy = 0; bool have_sof = false, progressive = false, seen_scan = false; W = H = 0;
  size_t o = 2;
  auto be16 = [&](size_t p) { return (int)((d[p] << 8) | d[p + 1]); };
  for (;;) {
    if (o + 4 > d.size()) { if (seen_scan) break; err = path + ": truncated JPEG"; return false; }
    if (d[o] != 0xFF) { err = path + ": bad JPEG marker"; return false; }
    const int m = d[o + 1];
    if (m == 0xFF) { ++o; continue; }
    if (m == 0xD9) { if (seen_scan) break; err = path + ": no image data"; return false; }
    const int L = be16(o + 2);
    if (L < 2 || o + 2 + (size_t)L > d.size()) { err = path + ": truncated JPEG segment"; return false; }      // L counts its own two bytes
    const uint8_t* p = &d[o + 4]; const int n = L - 2;       // payload p[0 .. n): every fixed-size read below is checked against n first
    if (m == 0xDB) { for (int k = 0; k < n;) { const int pq = p[k] >> 4, tq = p[k] & 15; ++k; if (tq > 3 || pq > 1 || k + 64 * (1 + pq) > n) { err = path + ": bad quantisation table"; return false; }
        for (int i = 0; i < 64; ++i) { qt[tq][zz[i]] = pq ? (uint16_t)((p[k] << 8) | p[k + 1]) : p[k]; k += pq ? 2 : 1; } } }
    else if (m == 0xC4) { for (int k = 0; k < n;) { const int tc = p[k] >> 4, th = p[k] & 15; ++k; if (th > 3 || tc > 1 || k + 16 > n) { err = path + ": bad Huffman table"; return false; }
        JpegHuff& h = tc ? hac[th] : hdc[th]; int cnt = 0; for (int l = 1; l <= 16; ++l) { h.bits[l] = p[k + l - 1]; cnt += h.bits[l]; } k += 16;
        if (cnt > 256 || k + cnt > n) { err = path + ": bad Huffman table"; return false; }
        memcpy(h.vals, p + k, (size_t)cnt); k += cnt; jpeg_build_huff(h); } }
    else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (have_sof) { err = path + ": second frame header"; return false; }
      if (n < 6 || n < 6 + 3 * (int)p[5]) { err = path + ": truncated frame header"; return false; }
      if (p[0] != 8) { err = path + ": only 8-bit JPEG images are read"; return false; }
      progressive = m == 0xC2;
      H = (uint32_t)be16(o + 5); W = (uint32_t)be16(o + 7); nc = p[5];
      if ((nc != 1 && nc != 3) || !W || !H) { err = path + ": only grey and 3-component JPEG images are read"; return false; }
      for (int c = 0; c < nc; ++c) { comp[c].id = p[6 + 3 * c]; comp[c].h = p[7 + 3 * c] >> 4; comp[c].v = p[7 + 3 * c] & 15; comp[c].tq = p[8 + 3 * c] & 3; }
      if (nc == 1) comp[0].h = comp[0].v = 1;                                    // a single-component frame is never interleaved
      for (int c = 0; c < nc; ++c) { hmax = std::max(hmax, comp[c].h); vmax = std::max(vmax, comp[c].v); }
      for (int c = 0; c < nc; ++c)
        if (comp[c].h < 1 || comp[c].v < 1 || comp[c].h > 2 || comp[c].v > 2 || (c > 0 && (comp[c].h != 1 || comp[c].v != 1)) || (c == 0 && (comp[c].h != hmax || comp[c].v != vmax)))
          { err = path + ": unsupported JPEG sampling factors"; return false; }
      if (hmax == 1 && vmax == 2) { err = path + ": unsupported JPEG sampling factors (1x2)"; return false; }
      if ((double)W * H > 1073741824.0 || (double)W * H > 4096.0 * (double)d.size() + 65536.0) { err = path + ": image size does not fit its data"; return false; }      // untrusted input: no 65535 x 65535 canvas for a 300-byte file
      mcux = (int)((W + 8 * hmax - 1) / (8 * hmax)); mcuy = (int)((H + 8 * vmax - 1) / (8 * vmax));
      for (int c = 0; c < nc; ++c) { Comp& C = comp[c]; C.bw = mcux * C.h; C.bh = mcuy * C.v;
        const int cw = (int)((W * C.h + hmax - 1) / hmax), chh = (int)((H * C.v + vmax - 1) / vmax); C.rw = (cw + 7) / 8; C.rh = (chh + 7) / 8;   // blocks a non-interleaved scan covers
        C.coef.assign((size_t)C.bw * C.bh * 64, 0); }
      have_sof = true;
    }
    else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) { err = path + ": lossless / hierarchical / arithmetic-coded JPEG is not read"; return false; }
    else if (m == 0xDD) { if (n < 2) { err = path + ": truncated restart interval"; return false; } restart = be16(o + 4); }
    else if (m == 0xEE && n >= 12 && memcmp(p, "Adobe", 5) == 0) adobe_transform = p[11];
    else if (m == 0xDA) {
      if (!have_sof) { err = path + ": scan before frame header"; return false; }
      if (n < 1) { err = path + ": bad scan header"; return false; }
      const int ns = p[0]; if (ns < 1 || ns > nc || n < 4 + 2 * ns) { err = path + ": bad scan header"; return false; }
      int sc[3];
      for (int k = 0; k < ns; ++k) { int c = 0; while (c < nc && comp[c].id != p[1 + 2 * k]) ++c; if (c == nc) { err = path + ": scan of an unknown component"; return false; }
        sc[k] = c; comp[c].td = p[2 + 2 * k] >> 4; comp[c].ta = p[2 + 2 * k] & 15; if (comp[c].td > 3 || comp[c].ta > 3) { err = path + ": bad table selector"; return false; } }
      const int Ss = p[1 + 2 * ns], Se = p[2 + 2 * ns], Ah = p[3 + 2 * ns] >> 4, Al = p[3 + 2 * ns] & 15;
      if (!progressive && (Ss != 0 || Se != 63 || Ah != 0 || Al != 0)) { err = path + ": bad sequential scan parameters"; return false; }
      if (progressive && (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13)) { err = path + ": bad progressive scan parameters"; return false; }
      for (int k = 0; k < ns; ++k) { const Comp& C = comp[sc[k]];
        if ((Ss == 0 && Ah == 0 && !hdc[C.td].set) || (Se > 0 && !hac[C.ta].set)) { err = path + ": missing Huffman table"; return false; } }
      o += 2 + (size_t)L; seen_scan = true;
      JpegBits br{&d[o], d.data() + d.size()};
      for (int c = 0; c < nc; ++c) comp[c].pred = 0;
      int eobrun = 0, todo = restart, rst = 0;
      const bool inter = ns > 1;
      const int ux = inter ? mcux : comp[sc[0]].rw, uy = inter ? mcuy : comp[sc[0]].rh;       // scan units: MCUs, or the component's own blocks
      bool bad = false;
      // one block of this scan
      auto decode_block = [&](Comp& C, int16_t* blk) {
        if (!progressive) {
          const int s = jpeg_decode_sym(br, hdc[C.td]); if (s < 0 || s > 11) { bad = true; return; }
          C.pred = std::max(-(1 << 20), std::min(1 << 20, C.pred + jpeg_extend(br.receive(s), s))); blk[0] = (int16_t)C.pred;      // bounded: a corrupt stream must not overflow
          for (int k = 1; k < 64;) { const int rs = jpeg_decode_sym(br, hac[C.ta]); if (rs < 0) { bad = true; return; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r; if (k > 63) { bad = true; return; }
            blk[zz[k]] = (int16_t)jpeg_extend(br.receive(sz), sz); ++k; }
          return;
        }
        if (Ss == 0) {
          if (Ah == 0) { const int s = jpeg_decode_sym(br, hdc[C.td]); if (s < 0 || s > 11) { bad = true; return; }
            C.pred = std::max(-(1 << 20), std::min(1 << 20, C.pred + jpeg_extend(br.receive(s), s))); blk[0] = (int16_t)((long long)C.pred * (1 << Al)); }
          else if (br.bit()) blk[0] |= (int16_t)(1 << Al);
          return;
        }
        if (Ah == 0) {
          if (eobrun > 0) { --eobrun; return; }
          for (int k = Ss; k <= Se; ++k) { const int rs = jpeg_decode_sym(br, hac[C.ta]); if (rs < 0) { bad = true; return; }
            const int r = rs >> 4, sz = rs & 15;
            if (sz) { k += r; if (k > 63) { bad = true; return; } blk[zz[k]] = (int16_t)(jpeg_extend(br.receive(sz), sz) * (1 << Al)); }
            else if (r == 15) k += 15;
            else { eobrun = (1 << r); if (r) eobrun += br.receive(r); --eobrun; break; } }
          return;
        }
        const int p1 = 1 << Al, m1 = -(1 << Al);
        auto refine = [&](int16_t& cf) { if (br.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1); };
        int k = Ss;
        if (eobrun == 0) {
          for (; k <= Se; ++k) { const int rs = jpeg_decode_sym(br, hac[C.ta]); if (rs < 0) { bad = true; return; }
            int r = rs >> 4, sv = rs & 15;
            if (sv) sv = br.bit() ? p1 : m1;
            else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.receive(r); break; }
            for (; k <= Se; ++k) { int16_t& cf = blk[zz[k]];
              if (cf != 0) refine(cf); else if (--r < 0) break; }
            if (sv && k <= Se) blk[zz[k]] = (int16_t)sv; }
        }
        if (eobrun > 0) { for (; k <= Se; ++k) { int16_t& cf = blk[zz[k]]; if (cf != 0) refine(cf); } --eobrun; }
      };
      for (int my = 0; my < uy && !bad; ++my) for (int mx = 0; mx < ux && !bad; ++mx) {
        if (restart && todo == 0) {
          br.reset();
          while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
          if (br.p + 1 >= br.end || br.p[1] != 0xD0 + (rst & 7)) { err = path + ": restart marker missing"; return false; }
          br.p += 2; ++rst; todo = restart; eobrun = 0;
          for (int c = 0; c < nc; ++c) comp[c].pred = 0;
        }
        if (inter) { for (int k = 0; k < ns && !bad; ++k) { Comp& C = comp[sc[k]];
            for (int by = 0; by < C.v; ++by) for (int bx = 0; bx < C.h; ++bx) decode_block(C, &C.coef[((size_t)(my * C.v + by) * C.bw + (size_t)(mx * C.h + bx)) * 64]); } }
        else { Comp& C = comp[sc[0]]; decode_block(C, &C.coef[((size_t)my * C.bw + (size_t)mx) * 64]); }
        if (restart) --todo;
      }
      if (bad) { err = path + ": corrupt JPEG data"; return false; }
      // continue after the entropy-coded segment: the next marker that is not a restart marker or a stuffed byte
      const uint8_t* q = br.p;
      while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
      o = (size_t)(q - d.data());
      continue;
    }
    o += 2 + (size_t)L;
  }
  if (!have_sof || !seen_scan) { err = path + ": no image data"; return false; }
  // dequantise + inverse DCT
  { long long coef[64];
    for (int c = 0; c < nc; ++c) { Comp& C = comp[c]; const int stride = C.bw * 8; C.plane.assign((size_t)stride * C.bh * 8, 0);
      for (int by = 0; by < C.bh; ++by) for (int bx = 0; bx < C.bw; ++bx) { const int16_t* blk = &C.coef[((size_t)by * C.bw + bx) * 64];
        for (int i = 0; i < 64; ++i) coef[i] = (long long)blk[i] * qt[C.tq][i];
        jpeg_idct_islow(coef, &C.plane[(size_t)by * 8 * stride + (size_t)bx * 8], stride); }
      std::vector<int16_t>().swap(C.coef); } }
  // chroma -> full resolution (fancy upsampling), over the real (not MCU-padded) extent
  auto sample_at = [](const Comp& c, int x, int y) { return (int)c.plane[(size_t)y * c.bw * 8 + x]; };
  std::vector<uint8_t> full[3];
  for (int c = 0; c < nc; ++c) {
    const int hs = hmax / comp[c].h, vs = vmax / comp[c].v;
    full[c].resize((size_t)W * H);
    if (hs == 1 && vs == 1) { for (uint32_t y = 0; y < H; ++y) memcpy(&full[c][(size_t)y * W], &comp[c].plane[(size_t)y * comp[c].bw * 8], W); continue; }
    const int dw = (int)((W + hs - 1) / hs), dh = (int)((H + vs - 1) / vs);      // downsampled extent
    std::vector<int> colsum((size_t)dw);
    std::vector<uint8_t> row((size_t)dw * 2);
    for (uint32_t y = 0; y < H; ++y) {
      if (dw <= 2) {                                                   // the IJG decoder filters only planes wider than two samples; narrower ones are replicated
        const int sy = (int)(y / (uint32_t)vs);
        for (int x = 0; x < dw; ++x) row[2 * x] = row[2 * x + 1] = (uint8_t)sample_at(comp[c], x, sy);
      } else if (vs == 1) {                                            // h2v1: 3/4 nearer + 1/4 further, rounding alternates
        const int sy = (int)y;
        int v = sample_at(comp[c], 0, sy); row[0] = (uint8_t)v; row[1] = (uint8_t)((v * 3 + sample_at(comp[c], 1, sy) + 2) >> 2);
        for (int x = 1; x < dw - 1; ++x) { v = sample_at(comp[c], x, sy) * 3; row[2 * x] = (uint8_t)((v + sample_at(comp[c], x - 1, sy) + 1) >> 2); row[2 * x + 1] = (uint8_t)((v + sample_at(comp[c], x + 1, sy) + 2) >> 2); }
        v = sample_at(comp[c], dw - 1, sy); row[2 * dw - 2] = (uint8_t)((v * 3 + sample_at(comp[c], dw - 2, sy) + 1) >> 2); row[2 * dw - 1] = (uint8_t)v;
      } else {                                                         // h2v2: vertical 3:1 blend with the nearer neighbour row, then the same horizontally on 16ths
        const int sy = (int)(y >> 1); int oy = (y & 1) ? sy + 1 : sy - 1;
        oy = oy < 0 ? 0 : (oy > dh - 1 ? dh - 1 : oy);
        for (int x = 0; x < dw; ++x) colsum[x] = sample_at(comp[c], x, sy) * 3 + sample_at(comp[c], x, oy);
        row[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4); row[1] = (uint8_t)((colsum[0] * 3 + colsum[1] + 7) >> 4);
        for (int x = 1; x < dw - 1; ++x) { row[2 * x] = (uint8_t)((colsum[x] * 3 + colsum[x - 1] + 8) >> 4); row[2 * x + 1] = (uint8_t)((colsum[x] * 3 + colsum[x + 1] + 7) >> 4); }
        row[2 * dw - 2] = (uint8_t)((colsum[dw - 1] * 3 + colsum[dw - 2] + 8) >> 4); row[2 * dw - 1] = (uint8_t)((colsum[dw - 1] * 4 + 7) >> 4);
      }
      memcpy(&full[c][(size_t)y * W], row.data(), W);
    }
  }
  out.resize((size_t)W * H * 3);
  auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  if (nc == 1) { for (size_t i = 0; i < (size_t)W * H; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = full[0][i]; return true; }
  if (adobe_transform == 0) { for (size_t i = 0; i < (size_t)W * H; ++i) for (int c = 0; c < 3; ++c) out[3 * i + c] = full[c][i]; return true; }
  int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) { const int x = i - 128;
    cr_r[i] = (91881 * x + 32768) >> 16; cb_b[i] = (116130 * x + 32768) >> 16; cr_g[i] = -46802 * x; cb_g[i] = -22554 * x + 32768; }
  for (size_t i = 0; i < (size_t)W * H; ++i) {
    const int y = full[0][i], cb = full[1][i], cr = full[2][i];
    out[3 * i] = clamp8(y + cr_r[cr]); out[3 * i + 1] = clamp8(y + ((cb_g[cb] + cr_g[cr]) >> 16)); out[3 * i + 2] = clamp8(y + cb_b[cb]);
  }
  return true;
}

}  // namespace detail
}  // namespace crh_host
