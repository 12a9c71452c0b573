// model_tcl_dump.cpp -- host-only: read a model.tcl with host/model_tcl.hpp and write what it understood as a .crhscene v2 file
// (cadrays_amd/scene_io.py layout), so that tests can compare the C++ reader with the Python reader byte for byte without a GPU.
//   model_tcl_dump <model.tcl> <out.crhscene> [WxH]
//   model_tcl_dump --image <file.png|jpg> <out.raw>      decoded image as three uint32 (w, h, channels) + bytes
//   model_tcl_dump --to-png <file.png|jpg> <out.png>     decoded image written back by the host's PNG writer
#include <cstdio>

#include "model_tcl.hpp"

int main(int argc, char** argv)
{
  if (argc == 4 && std::string(argv[1]) == "--image") {
    uint32_t d[3]; std::vector<uint8_t> px; std::string e;
    if (!crh_host::detail::read_image_u8(argv[2], d[0], d[1], d[2], px, e)) { fprintf(stderr, "%s\n", e.c_str()); return 1; }
    FILE* f = fopen(argv[3], "wb"); if (!f) { perror(argv[3]); return 1; }
    fwrite(d, 4, 3, f); fwrite(px.data(), 1, px.size(), f); fclose(f); return 0;
  }
  if (argc == 4 && std::string(argv[1]) == "--to-png") {
    uint32_t w, h, ch; std::vector<uint8_t> px; std::string e;
    if (!crh_host::detail::read_image_u8(argv[2], w, h, ch, px, e) || !crh_host::detail::write_png(argv[3], px.data(), w, h, ch, e)) { fprintf(stderr, "%s\n", e.c_str()); return 1; }
    return 0;
  }
  if (argc < 3) { fprintf(stderr, "usage: %s <model.tcl> <out.crhscene> [WxH]\n", argv[0]); return 2; }
  uint32_t w = 512, h = 512;
  if (argc > 3 && sscanf(argv[3], "%ux%u", &w, &h) != 2) { fprintf(stderr, "bad size\n"); return 2; }
  crh_host::TclScene sc; std::string err;
  if (!crh_host::read_model_tcl(argv[1], w, h, sc, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
  for (const std::string& u : sc.unsupported) fprintf(stderr, "not honoured: %s\n", u.c_str());
  FILE* f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 1; }
  const uint32_t hdr[7] = {2, (uint32_t)(sc.pos.size() / 3), (uint32_t)(sc.tri.size() / 4), (uint32_t)sc.mats.size(), (uint32_t)sc.lights.size(), sc.envW, sc.envH};
  fwrite("CRHS", 1, 4, f); fwrite(hdr, 4, 7, f); fwrite(&sc.cam, sizeof sc.cam, 1, f); fwrite(&sc.par, sizeof sc.par, 1, f);
  fwrite(sc.pos.data(), 4, sc.pos.size(), f); fwrite(sc.nrm.data(), 4, sc.nrm.size(), f); fwrite(sc.tri.data(), 4, sc.tri.size(), f);
  fwrite(sc.mats.data(), sizeof(crh_bsdf), sc.mats.size(), f); if (!sc.lights.empty()) fwrite(sc.lights.data(), sizeof(crh_light), sc.lights.size(), f);
  if (!sc.env.empty()) fwrite(sc.env.data(), 4, sc.env.size(), f);
  const uint32_t ext[3] = {sc.uv.empty() ? 0u : 1u, 0u, (uint32_t)sc.textures.size()};
  fwrite(ext, 4, 3, f);
  if (!sc.uv.empty()) fwrite(sc.uv.data(), 4, sc.uv.size(), f);
  for (const crh_host::Texture& t : sc.textures) { const uint32_t d[3] = {t.w, t.h, t.ch}; fwrite(d, 4, 3, f); fwrite(t.texels.data(), 4, t.texels.size(), f); }
  fclose(f);
  return 0;
}
