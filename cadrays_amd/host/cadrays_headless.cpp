// cadrays_headless.cpp -- headless C++ host driver over the C ABI of libcadrays_hip.so.
//
// Restates, without window / ImGui / Tcl, the part of the reference application that drives the hot
// path in test mode:
//   CADRays <script.tcl> <nFrames>                         reference src/Launcher/main.cxx:164-189
//   loop: View->Redraw() once per frame, count frames      src/Launcher/AppViewer.cxx:1045-1071
//   BufferDump(Graphic3d_BT_RGB) after the last frame      src/Launcher/AppViewer.cxx:1255-1264
//   write Output_<name>_<n>.png and Output_<name>_<n>.txt (average frame rate)   main.cxx:193-228
// Here: cadrays_headless <scene.crhscene | model.tcl> <nFrames> [device] [lookahead] [gpus] [WxH] writes Output_<name>_<n>.png (+ .ppm: the same LDR pixels),
// Output_<name>_<n>.pfm (linear HDR, the parity buffer of AppGui.cxx:345-349) and Output_<name>_<n>.txt.
// gpus > 1: one context per GPU (devices device .. device+gpus-1; CRH_HEADLESS_SHARE_DEVICE=1 keeps them all on `device`),
// screen tiles interleaved across the contexts, one host thread per context, crh_reduce (RCCL over xGMI) assembles the
// frame on context 0 -- bit-identical to the one-GPU image.
// `--loop lone|drag|display` anywhere on the command line (one GPU): the application's own call patterns instead of nFrames back-to-back Redraw()s, timed from
// this C++ host (the reference host is C++: no interpreter between the calls) --
//   lone     nFrames times: crh_reset, wait, then crh_render(1) + crh_sync timed alone: the frame after a restart (AppViewer.cxx:979-984)
//   drag     every frame: crh_set_camera (the eye orbits the scene centre) + crh_reset + crh_render(1) + crh_read_ldr_begin, the frame of two frames ago collected
//            with crh_read_ldr_end: orbiting the model with the mouse held down, every frame shown (AppViewer.cxx:1099)
//   display  every frame: crh_render(1) + the same asynchronous read-back: a still camera, every frame shown
// (eight untimed warm-up frames come first in every loop mode)
// The outputs are written from the state after the last frame as ever; the JSON line gains "loop", "loop_frames_per_s" / "lone_frame_ms_median".
// A path ending in .tcl is the reference's own saved-scene format (model.tcl + meshes/ + textures/, what File > Export writes and
// ImportSettingsEditor.cxx:378-380 sources back in): read by host/model_tcl.hpp at WxH (default 512x512).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cadrays_hip.h"
#include "model_tcl.hpp"

namespace {

template <class T> bool read_vec(FILE* f, std::vector<T>& v, size_t n)
{
  v.resize(n);
  return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

}  // namespace

int main(int argc, char** argv)
{
  setenv("GPU_MAX_HW_QUEUES", "16", 0);      // before the first HIP call: eight frames in flight need more hardware queues than the runtime's default four (crh_set_pipeline_depth)

  std::string loop;
  for (int i = 1; i + 1 < argc; ++i)
    if (std::string(argv[i]) == "--loop") { loop = argv[i + 1]; for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2]; argc -= 2; break; }
  if (!loop.empty() && loop != "lone" && loop != "drag" && loop != "display") { fprintf(stderr, "--loop must be lone, drag or display\n"); return 2; }
  if (argc < 3) { fprintf(stderr, "usage: %s <scene.crhscene | model.tcl> <nFrames> [device] [lookahead] [gpus] [WxH] [--loop lone|drag|display]\n", argv[0]); return 2; }
  const std::string path = argv[1];
  const int n_frames = atoi(argv[2]);
  const int device = argc > 3 ? atoi(argv[3]) : 0;
  const int lookahead = argc > 4 ? atoi(argv[4]) : 1;      // k > 1: crh_set_lookahead(k), frames traced ahead per wide batch; -k: crh_set_lookahead_auto(k)
  const int n_gpus = argc > 5 ? atoi(argv[5]) : 1;
  if (n_frames <= 0 || n_gpus <= 0) { fprintf(stderr, "nFrames and gpus must be > 0\n"); return 2; }

  crh_camera cam; crh_params par;
  std::vector<float> pos, nrm, env; std::vector<int32_t> tri; std::vector<crh_bsdf> mats; std::vector<crh_light> lights;
  struct Tex { uint32_t w = 0, h = 0, ch = 0; std::vector<float> texels; };
  std::vector<float> uv, xform; std::vector<int32_t> tri_obj; std::vector<Tex> textures; uint32_t nO = 0;
  uint32_t nV = 0, nT = 0, nM = 0, nL = 0, eW = 0, eH = 0;
  if (path.size() > 4 && path.substr(path.size() - 4) == ".tcl") {
    uint32_t w = 512, h = 512;
    if (argc > 6 && sscanf(argv[6], "%ux%u", &w, &h) != 2) { fprintf(stderr, "bad size %s (WxH)\n", argv[6]); return 2; }
    crh_host::TclScene sc; std::string err;
    if (!crh_host::read_model_tcl(path, w, h, sc, err)) { fprintf(stderr, "cadrays_headless: %s\n", err.c_str()); return 1; }
    for (const std::string& u : sc.unsupported) fprintf(stderr, "cadrays_headless: not honoured: %s\n", u.c_str());
    cam = sc.cam; par = sc.par; pos.swap(sc.pos); nrm.swap(sc.nrm); uv.swap(sc.uv); tri.swap(sc.tri); mats.swap(sc.mats); lights.swap(sc.lights);
    env.swap(sc.env); eW = sc.envW; eH = sc.envH;
    textures.resize(sc.textures.size());
    for (size_t i = 0; i < textures.size(); ++i) { textures[i].w = sc.textures[i].w; textures[i].h = sc.textures[i].h; textures[i].ch = sc.textures[i].ch; textures[i].texels.swap(sc.textures[i].texels); }
    nV = (uint32_t)(pos.size() / 3); nT = (uint32_t)(tri.size() / 4); nM = (uint32_t)mats.size(); nL = (uint32_t)lights.size();
  } else {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { perror(path.c_str()); return 1; }
  char magic[4]; uint32_t hdr[7];
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "CRHS", 4) != 0 || fread(hdr, 4, 7, f) != 7 || (hdr[0] != 1 && hdr[0] != 2)) { fprintf(stderr, "not a .crhscene v1/v2 file\n"); return 1; }
  nV = hdr[1]; nT = hdr[2]; nM = hdr[3]; nL = hdr[4]; eW = hdr[5]; eH = hdr[6];
  bool ok = fread(&cam, sizeof cam, 1, f) == 1 && fread(&par, sizeof par, 1, f) == 1 && read_vec(f, pos, 3 * (size_t)nV) && read_vec(f, nrm, 3 * (size_t)nV) &&
            read_vec(f, tri, 4 * (size_t)nT) && read_vec(f, mats, nM) && read_vec(f, lights, nL) && read_vec(f, env, 3 * (size_t)eW * eH);
  // version 2: texture coordinates, the two-level (per-object transform) description and the Kd textures
  if (ok && hdr[0] >= 2) {
    uint32_t ext[3];
    ok = fread(ext, 4, 3, f) == 3;
    if (ok && ext[0]) ok = read_vec(f, uv, 2 * (size_t)nV);
    if (ok && ext[1]) { nO = ext[1]; ok = read_vec(f, tri_obj, nT) && read_vec(f, xform, 12 * (size_t)nO); }
    if (ok) textures.resize(ext[2]);
    for (size_t i = 0; ok && i < textures.size(); ++i) {
      uint32_t d[3];
      ok = fread(d, 4, 3, f) == 3;
      if (ok) { textures[i].w = d[0]; textures[i].h = d[1]; textures[i].ch = d[2]; ok = read_vec(f, textures[i].texels, (size_t)d[0] * d[1] * d[2]); }
    }
  }
  fclose(f);
  if (!ok) { fprintf(stderr, "truncated scene file\n"); return 1; }
  }

  // one context per GPU (== driver + viewer + view + FBO, AppViewer.cxx:601-638), each holding the whole scene
  const bool share = getenv("CRH_HEADLESS_SHARE_DEVICE") != nullptr;
  std::vector<crh_ctx*> ctx((size_t)n_gpus, nullptr);
  auto die_all = [&](crh_ctx* c, const char* what, int rc) {
    fprintf(stderr, "cadrays_headless: %s failed (%d): %s\n", what, rc, c ? crh_last_error(c) : "");
    for (crh_ctx* x : ctx) if (x) crh_destroy(x);
    return 1;
  };
  int rc;
  for (int g = 0; g < n_gpus; ++g) {
    crh_ctx* c = ctx[(size_t)g] = crh_create(share ? device : device + g);
    if (!c) return die_all(nullptr, "crh_create", CRH_E_DEVICE);
    if ((rc = crh_set_geometry(c, pos.data(), nrm.data(), uv.empty() ? nullptr : uv.data(), nV, tri.data(), nT,
                               nO ? tri_obj.data() : nullptr, nO ? xform.data() : nullptr, nO))) return die_all(c, "crh_set_geometry", rc);
    for (size_t i = 0; i < textures.size(); ++i)
      if (textures[i].w && (rc = crh_set_texture(c, (uint32_t)i, textures[i].texels.data(), textures[i].w, textures[i].h, textures[i].ch)))
        return die_all(c, "crh_set_texture", rc);
    if ((rc = crh_set_materials(c, mats.data(), nM))) return die_all(c, "crh_set_materials", rc);
    if ((rc = crh_set_lights(c, lights.data(), nL))) return die_all(c, "crh_set_lights", rc);
    if ((rc = crh_set_envmap(c, env.empty() ? nullptr : env.data(), eW, eH))) return die_all(c, "crh_set_envmap", rc);
    if ((rc = crh_set_camera(c, &cam))) return die_all(c, "crh_set_camera", rc);
    if ((rc = crh_set_params(c, &par))) return die_all(c, "crh_set_params", rc);
    if ((rc = crh_build(c))) return die_all(c, "crh_build", rc);
    { uint32_t frames = 3; crh_query_pipeline_capacity(&frames, nullptr);      // what the hardware queues of this process carry (main exported GPU_MAX_HW_QUEUES before the first HIP call)
      if (frames > 3 && !getenv("CRH_PIPE_DEPTH") && (rc = crh_set_pipeline_depth(c, frames))) return die_all(c, "crh_set_pipeline_depth", rc); }
    if (n_gpus == 1 && lookahead > 1 && (rc = crh_set_lookahead(c, (uint32_t)lookahead))) return die_all(c, "crh_set_lookahead", rc);
    if (n_gpus == 1 && lookahead < -1 && (rc = crh_set_lookahead_auto(c, (uint32_t)-lookahead))) return die_all(c, "crh_set_lookahead_auto", rc);
  }
  crh_ctx* c = ctx[0];

  double lone_median_ms = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  if (n_gpus == 1 && !loop.empty()) {
    // eight untimed frames first: the path state is allocated, every pipeline stream has launched once (the first launch on a hardware queue sets its scratch up)
    for (int k = 0; k < 8; ++k) if ((rc = crh_render(c, 1))) return die_all(c, "crh_render", rc);
    // ... and up to 64 lone frames until the library has settled on its feeder count for this scene (crh_get_frame_tuning: the first 30 frame-kernel frames after a
    // build are measurements; a viewer passes them in the first tenth of a second)
    for (int k = 0; k < 96; ++k) {
      uint32_t tune[5] = {0, 0, 0, 0, 0}; uint64_t order[7] = {0, 0, 0, 0, 0, 0, 0};
      const bool feeders_settled = crh_get_frame_tuning(c, tune) || !tune[0] || tune[1];
      const bool order_settled = crh_get_tile_order(c, nullptr, nullptr, order) || order[4] != 0;      // (whether the sorted tile list pays on this scene: ~ 16 lone frames more)
      if (feeders_settled && order_settled) break;
      if ((rc = crh_reset(c)) || (rc = crh_render(c, 1)) || (rc = crh_sync(c))) return die_all(c, "crh_render", rc);
    }
    if ((rc = crh_reset(c)) || (rc = crh_sync(c))) return die_all(c, "crh_reset", rc);
    t0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> shown(3 * (size_t)par.width * par.height);
    const crh_camera cam0 = cam;
    auto orbit = [&](int i) {                                  // the eye turns about the z axis through the scene's origin, looking at it
      crh_camera k = cam0;
      const double a = 0.002 * i, r = std::sqrt((double)cam0.eye[0] * cam0.eye[0] + (double)cam0.eye[1] * cam0.eye[1]);
      const double a0 = std::atan2((double)cam0.eye[1], (double)cam0.eye[0]);
      k.eye[0] = (float)(r * std::cos(a0 + a)); k.eye[1] = (float)(r * std::sin(a0 + a));
      const double dx = -k.eye[0], dy = -k.eye[1], dz = -k.eye[2], n = std::sqrt(dx * dx + dy * dy + dz * dz);
      if (n > 0) { k.dir[0] = (float)(dx / n); k.dir[1] = (float)(dy / n); k.dir[2] = (float)(dz / n); }
      return k;
    };
    if (loop == "lone") {
      std::vector<double> ms;
      for (int frame = 0; frame < n_frames; ++frame) {
        if ((rc = crh_reset(c)) || (rc = crh_sync(c))) return die_all(c, "crh_reset", rc);
        const auto a = std::chrono::steady_clock::now();
        if ((rc = crh_render(c, 1)) || (rc = crh_sync(c))) return die_all(c, "crh_render", rc);
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
      }
      std::sort(ms.begin(), ms.end()); lone_median_ms = ms[ms.size() / 2];
    } else {
      int begun = 0;
      for (int frame = 0; frame < n_frames; ++frame) {
        if (loop == "drag") { const crh_camera k = orbit(frame); if ((rc = crh_set_camera(c, &k)) || (rc = crh_reset(c))) return die_all(c, "crh_set_camera / crh_reset", rc); }
        if ((rc = crh_render(c, 1))) return die_all(c, "crh_render", rc);
        if (begun >= 2 && (rc = crh_read_ldr_end(c, shown.data()))) return die_all(c, "crh_read_ldr_end", rc);
        if ((rc = crh_read_ldr_begin(c))) return die_all(c, "crh_read_ldr_begin", rc);
        ++begun;
      }
      for (int k = 0; k < std::min(begun, 2); ++k) if ((rc = crh_read_ldr_end(c, shown.data()))) return die_all(c, "crh_read_ldr_end", rc);
      if ((rc = crh_sync(c))) return die_all(c, "crh_sync", rc);
    }
  } else if (n_gpus == 1) {
    // the render loop of AppViewer::Run in test mode: one Redraw per frame until MaxFramesCount
    for (int frame = 0; frame < n_frames; ++frame)
      if ((rc = crh_render(c, 1))) return die_all(c, "crh_render", rc);
    if ((rc = crh_sync(c))) return die_all(c, "crh_sync", rc);
  } else {
    // k-th tile along the Z-order curve -> context k mod gpus (the RT tile entry point, SettingsWidget.cxx:451-476; the same interleave as
    // cadrays_amd/sharding.py: neighbouring tiles cost about the same and land on different GPUs); every context renders all
    // frames of its own tiles with the one-GPU RNG, so the assembled image does not depend on the GPU count
    const uint32_t ts = par.tile_size, tiles_x = (par.width + ts - 1) / ts, n_tiles = tiles_x * ((par.height + ts - 1) / ts);
    std::vector<uint32_t> zorder(n_tiles);
    {
      auto spread = [](uint64_t v) { v = (v | (v << 16)) & 0x0000FFFF0000FFFFull; v = (v | (v << 8)) & 0x00FF00FF00FF00FFull; v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
                                     v = (v | (v << 2)) & 0x3333333333333333ull; return (v | (v << 1)) & 0x5555555555555555ull; };
      for (uint32_t t = 0; t < n_tiles; ++t) zorder[t] = t;
      std::stable_sort(zorder.begin(), zorder.end(), [&](uint32_t a, uint32_t b) {
        return (spread(a % tiles_x) | (spread(a / tiles_x) << 1)) < (spread(b % tiles_x) | (spread(b / tiles_x) << 1)); });
    }
    std::vector<int> rcs((size_t)n_gpus, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpus; ++g)
      th.emplace_back([&, g] {
        std::vector<uint32_t> mine;
        for (uint32_t k = (uint32_t)g; k < n_tiles; k += (uint32_t)n_gpus) mine.push_back(zorder[k]);
        std::sort(mine.begin(), mine.end());
        int r = mine.empty() ? 0 : crh_render_tiles(ctx[(size_t)g], mine.data(), (uint32_t)mine.size(), 0, (uint32_t)n_frames);
        if (!r) r = crh_sync(ctx[(size_t)g]);
        rcs[(size_t)g] = r;
      });
    for (auto& t : th) t.join();
    for (int g = 0; g < n_gpus; ++g) if (rcs[(size_t)g]) return die_all(ctx[(size_t)g], "crh_render_tiles", rcs[(size_t)g]);
    if ((rc = crh_reduce(ctx.data(), (uint32_t)n_gpus, 0))) return die_all(c, "crh_reduce", rc);
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double fps = n_frames / secs;

  std::vector<uint8_t> ldr(3 * (size_t)par.width * par.height);
  std::vector<float> hdr_img(3 * (size_t)par.width * par.height);
  if ((rc = crh_read_ldr(c, ldr.data()))) return die_all(c, "crh_read_ldr", rc);   // BufferDump(Graphic3d_BT_RGB)
  if ((rc = crh_read_hdr(c, hdr_img.data()))) return die_all(c, "crh_read_hdr", rc);   // Graphic3d_BT_RGB_RayTraceHdrLeft
  crh_stats st{}; 
  for (crh_ctx* x : ctx) { crh_stats s1; crh_get_stats(x, &s1); st.rays_nearest += s1.rays_nearest; st.rays_any += s1.rays_any; st.samples += s1.samples; }

  std::string stem = path; const size_t sl = stem.find_last_of('/'); std::string dir = sl == std::string::npos ? "." : stem.substr(0, sl);
  std::string name = sl == std::string::npos ? stem : stem.substr(sl + 1); const size_t dot = name.find_last_of('.'); if (dot != std::string::npos) name = name.substr(0, dot);
  const std::string base = dir + "/Output_" + name + "_" + std::to_string(n_frames);
  { std::string e; if (!crh_host::detail::write_png(base + ".png", ldr.data(), par.width, par.height, 3, e)) fprintf(stderr, "%s\n", e.c_str()); }   // the reference's dump format
  if (FILE* o = fopen((base + ".ppm").c_str(), "wb")) { fprintf(o, "P6\n%u %u\n255\n", par.width, par.height); fwrite(ldr.data(), 1, ldr.size(), o); fclose(o); }
  if (FILE* o = fopen((base + ".pfm").c_str(), "wb")) {
    fprintf(o, "PF\n%u %u\n-1.0\n", par.width, par.height);
    for (uint32_t y = par.height; y-- > 0;) fwrite(&hdr_img[3 * (size_t)y * par.width], sizeof(float), 3 * (size_t)par.width, o);   // PFM is bottom-up
    fclose(o);
  }
  if (FILE* o = fopen((base + ".txt").c_str(), "w")) { fprintf(o, "%g", fps); fclose(o); }
  printf("{\"scene\": \"%s\", \"gpus\": %d, \"frames\": %d, \"fps\": %.4f, \"seconds\": %.6f, \"rays_nearest\": %llu, \"rays_any\": %llu, \"samples\": %llu, \"mrays_per_s\": %.3f",
         name.c_str(), n_gpus, n_frames, fps, secs, (unsigned long long)st.rays_nearest, (unsigned long long)st.rays_any, (unsigned long long)st.samples,
         (double)(st.rays_nearest + st.rays_any) / secs / 1e6);
  if (!loop.empty()) printf(", \"loop\": \"%s\", \"loop_frames_per_s\": %.2f, \"lone_frame_ms_median\": %.4f", loop.c_str(), loop == "lone" ? 0.0 : fps, lone_median_ms);
  if (!loop.empty()) { uint32_t tune[5] = {0, 0, 0, 0, 0}; if (!crh_get_frame_tuning(c, tune)) printf(", \"frame_feeders\": %u, \"frame_us_3_4_feeders\": [%u, %u]", tune[1], tune[3], tune[4]);
                       uint64_t order[7] = {0, 0, 0, 0, 0, 0, 0}; if (!crh_get_tile_order(c, nullptr, nullptr, order)) printf(", \"tile_order_verdict\": %llu, \"frame_us_sorted_row_major\": [%llu, %llu]", (unsigned long long)order[4], (unsigned long long)order[5], (unsigned long long)order[6]); }
  printf("}\n");
  for (crh_ctx* x : ctx) crh_destroy(x);
  return 0;
}
