/* crh_spec.h -- the places where this project's frozen algorithm spec DELIBERATELY departs from what the
 * OCCT 7.3-era GLSL path tracer is recollected to do ([OCCT-ext]: no OCCT source, header or shader is on
 * disk, so neither the recollection nor the departure can be verified here -- parity against the real renderer is UNPINNED).
 *
 * Every departure is a named switch, so that somebody with an OCCT build can flip them one at a time and measure which
 * setting the real renderer agrees with (tools/occt_pin/ is the kit for that).  The DEFAULT of every switch is the behaviour
 * all committed golden vectors were generated with; product (cadrays_amd/csrc) and CPU oracle (oracle/) honour the same
 * switches through the same struct, and tests/test_spec_switches.py keeps them bit-identical with every switch flipped.
 *
 *  #  switch                         default (this spec)                          alternative (OCCT as recollected)
 *  1  crh_spec.uniform_32bit         u = (state >> 8) * 2^-24  in [0, 1)          u = float(state) * 2^-32, rounds to 1.0 for the top 128 states
 *                                     -- never 1.0, so sqrt(1 - u), the light pick and the lobe pick need no guard
 *  2  crh_spec.texel_gamma2          texels handed over are LINEAR float data      the lookup squares the FILTERED texel (pow(rgb, 2): the shader's
 *                                     (the file readers square 8-bit images on       stand-in for sRGB decoding of 8-bit env maps / diffuse maps);
 *                                     the host, before filtering)                    callers then hand over the raw image values in [0, 1]
 *  3  crh_spec.mis_single_lobe       MIS weight of a BSDF-sampled ray uses the     pdf of the SAMPLED lobe times its selection probability only
 *                                     mixture pdf over all non-delta lobes           (what SampleBsdfLayered returns); NEE still evaluates the mixture,
 *                                     (weights of the two strategies sum to 1)       so the two weights no longer sum to one where lobes overlap
 *  4  CRH_SPEC_ORDER_EXACT (build)   child order key = entry-distance bits with    exact entry distance, ties by slot (a full float compare per
 *                                     the slot index in the two low mantissa bits    comparator); changes visit counters and the winner among hits at
 *                                     (one 32-bit integer sort key per child)        EQUAL t only -- compile-time because it sits in the traversal loop
 *  5  CRH_BVH_LEAF_SIZE (format)     one triangle per leaf (crh_bvh_format.h: a    OCCT's builder stops at ~5 triangles (Appendix A "(?)").  NOT a
 *                                     leaf reference IS a triangle index; measured   switch: fixed by the node format, and image-neutral -- the nearest
 *                                     +14 % on the 1 M-triangle soup) hit does not depend on the tree, only the winner among equal t
 *  6  crh_spec.eps_rule              eps = max(1e-6, 1e-5 * |scene diagonal|)      eps = max(1e-6, 1e-4 * scene radius), radius = |diagonal| / 2
 *                                     (crh_params.scene_epsilon > 0 overrides both)
 *  7  crh_spec.eta_no_dielectric     specular transmission under a coat that is    any other index for that case (e.g. 1.5, the material editor's
 *                                     NOT a dielectric is index-matched: eta = 1     default glass index, MaterialEditor.cxx:789-806)
 *
 * Reference evidence that these are the knobs that matter: the only numbers CADRays itself pins are the BSDF / light / camera /
 * parameter vectors of its input contract (cadrays_hip.h cites them line by line); everything in the table is arithmetic inside
 * OCCT's shaders, reached through V3d_View::Redraw() (src/Launcher/AppViewer.cxx:1047).
 */
#ifndef CRH_SPEC_H
#define CRH_SPEC_H

#include <stdint.h>

typedef struct crh_spec {
  uint32_t size;                /* sizeof(crh_spec) of the caller: lets the struct grow without breaking the ABI */
  int32_t  uniform_32bit;       /* #1 */
  int32_t  texel_gamma2;        /* #2 */
  int32_t  mis_single_lobe;     /* #3 */
  int32_t  eps_rule;            /* #6: 0 = 1e-5 * diagonal, 1 = 1e-4 * radius */
  float    eta_no_dielectric;   /* #7: >= 1e-2; default 1 */
} crh_spec;

#define CRH_SPEC_DEFAULTS {(uint32_t)sizeof(crh_spec), 0, 0, 0, 0, 1.0f}

/* #4: build-time (traversal inner loop).  0 = quantised key (default), 1 = exact distance order, ties by slot. */
#ifndef CRH_SPEC_ORDER_EXACT
#define CRH_SPEC_ORDER_EXACT 0
#endif

/* #8: build-time.  Child order of ANY-HIT (shadow-ray) traversal: 0 = near to far like nearest-hit rays (default; what OCCT's
 * SceneAnyHit is recollected to do, it shares the ordered walk), 1 = slot order -- an occlusion query needs no order, and the walk
 * then needs no sort (round-2 verdict item 3).  Changes the any-hit visit counters only: visibility is order-independent. */
#ifndef CRH_SPEC_ANYHIT_SLOT_ORDER
#define CRH_SPEC_ANYHIT_SLOT_ORDER 0
#endif

#endif /* CRH_SPEC_H */
