"""ctypes mirror of include/cadrays_hip.h (the C ABI of libcadrays_hip.so).

The struct layouts are the boundary's input contract; the CPU oracle takes the same structs,
so tests hand identical bytes to both sides.
"""
import ctypes as C

OK, E_INVALID, E_DEVICE, E_NOMEM, E_NOTBUILT = 0, -1, -2, -3, -4

FRESNEL_CONSTANT, FRESNEL_CONDUCTOR, FRESNEL_DIELECTRIC = -1.0, -2.0, -3.0


class crh_bsdf(C.Structure):
    _fields_ = [(n, C.c_float * 4) for n in
                ("Kc", "Kd", "Ks", "Kt", "Le", "Absorption", "FresnelCoat", "FresnelBase")]


class crh_light(C.Structure):
    _fields_ = [("vec", C.c_float * 3), ("is_point", C.c_float),
                ("emission", C.c_float * 3), ("smoothness", C.c_float)]


class crh_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("fovy_deg", C.c_float), ("aspect", C.c_float), ("is_ortho", C.c_int32),
                ("ortho_scale", C.c_float), ("aperture_radius", C.c_float), ("focal_dist", C.c_float)]


class crh_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32),
                ("radiance_clamp", C.c_float), ("two_sided", C.c_int32), ("coherent_rng", C.c_int32),
                ("seed", C.c_uint32), ("tile_size", C.c_uint32), ("tonemap_mode", C.c_int32),
                ("exposure", C.c_float), ("white_point", C.c_float), ("background", C.c_float * 3),
                ("env_as_background", C.c_int32), ("scene_epsilon", C.c_float),
                ("russian_roulette", C.c_int32)]


class crh_stats(C.Structure):
    _fields_ = [("rays_nearest", C.c_uint64), ("rays_any", C.c_uint64), ("nodes_nearest", C.c_uint64),
                ("tris_nearest", C.c_uint64), ("nodes_any", C.c_uint64), ("tris_any", C.c_uint64), ("shaded_hits", C.c_uint64), ("samples", C.c_uint64),
                ("seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class crh_spec(C.Structure):
    """include/crh_spec.h: the switchable departures from the recollected OCCT behaviour; defaults = the frozen spec"""
    _fields_ = [("size", C.c_uint32), ("uniform_32bit", C.c_int32), ("texel_gamma2", C.c_int32), ("mis_single_lobe", C.c_int32),
                ("eps_rule", C.c_int32), ("eta_no_dielectric", C.c_float),
                ("rr_start_bounce", C.c_int32), ("rr_survival_cap", C.c_float), ("min_contribution", C.c_float), ("min_throughput", C.c_float),
                ("raygen_bilinear", C.c_int32), ("env_orientation", C.c_int32), ("display_gamma22", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "size"}


_f32 = lambda x: C.c_float(x).value      # the defaults as the float32 fields hold them, so that get_spec() == SPEC_DEFAULTS
SPEC_DEFAULTS = dict(uniform_32bit=0, texel_gamma2=0, mis_single_lobe=0, eps_rule=0, eta_no_dielectric=1.0,
                     rr_start_bounce=3, rr_survival_cap=_f32(0.95), min_contribution=_f32(1.0e-2), min_throughput=_f32(1.0e-3), raygen_bilinear=0, env_orientation=0, display_gamma22=0)

assert C.sizeof(crh_bsdf) == 128 and C.sizeof(crh_light) == 32 and C.sizeof(crh_spec) == 52

SCHEDULE_AUTO, SCHEDULE_WIDE, SCHEDULE_SMALL, SCHEDULE_STAGED = 0, 1, 2, 3      # crh_set_schedule

NODE_DWORDS = 16          # CRH_NODE_DWORDS of include/crh_bvh_format.h: 4-wide BVH node, 12 dwords used on a 64-B stride

# every symbol include/cadrays_hip.h declares (tests check the built library exports them all)
EXPORTS = [
    "crh_create", "crh_destroy", "crh_last_error", "crh_set_geometry", "crh_set_transforms", "crh_set_visibility", "crh_add_object", "crh_set_materials",
    "crh_set_lights", "crh_set_envmap", "crh_set_texture", "crh_set_camera", "crh_set_params", "crh_set_spec", "crh_get_spec", "crh_spec_order_exact", "crh_spec_anyhit_slot_order", "crh_build", "crh_reset",
    "crh_render", "crh_render_tiles", "crh_set_adaptive", "crh_set_show_tiles", "crh_set_lookahead", "crh_set_lookahead_auto", "crh_set_schedule", "crh_set_pipeline_depth", "crh_set_path_budget", "crh_get_tile_stats", "crh_sync", "crh_read_hdr", "crh_read_ldr", "crh_read_ldr_begin", "crh_read_ldr_end", "crh_read_hdr_begin", "crh_read_hdr_end",
    "crh_save_accum", "crh_load_accum", "crh_accum_device_ptr", "crh_reduce", "crh_enable_counters", "crh_get_stats", "crh_trace_nearest",
    "crh_trace_any", "crh_get_bvh", "crh_get_tlas", "crh_build_bvh_host", "crh_bench_trace", "crh_debug_math", "crh_debug_bsdf", "crh_enable_kernel_timing",
    "crh_get_kernel_timing", "crh_get_packet_stats", "crh_debug_reduce_fake_devices", "crh_get_path_budget", "crh_get_frame_tuning", "crh_get_tile_order", "crh_build_prebuilt", "crh_query_pipeline_capacity", "crh_env_table",
]
