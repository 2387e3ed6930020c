"""ctypes mirror of include/firework_hip.h (the C ABI).  Field order and types must match the header."""
import ctypes as C

FW_ABI_VERSION = 7
FW_INIT_NO_ARENA = 0xFFFFFFFFFFFFFFFF   # fw_init: no path arena (the first render sizes its own)
FW_MAX_SEGMENTS = 11

# fw_status
FW_OK = 0
FW_ERR_BAD_ARG = -1
FW_ERR_EMPTY_SCENE = -2
FW_ERR_NAN_BBOX = -3
FW_ERR_MESH_NORMALS = -4
FW_ERR_MESH_UVS = -5
FW_ERR_UNSUPPORTED = -6
FW_ERR_HIP = -7
FW_ERR_NO_DEVICE = -8
FW_ERR_BVH_DEPTH = -9
FW_ERR_OOM = -10

# kinds
FW_TEX_CONSTANT, FW_TEX_CHECKER, FW_TEX_PERLIN, FW_TEX_TURBULENCE, FW_TEX_MARBLE, FW_TEX_IMAGE = range(6)
FW_MAT_LAMBERTIAN, FW_MAT_METAL, FW_MAT_DIELECTRIC, FW_MAT_EMISSIVE, FW_MAT_ISOTROPIC = range(5)
(FW_SHAPE_SPHERE, FW_SHAPE_XYRECT, FW_SHAPE_XZRECT, FW_SHAPE_YZRECT, FW_SHAPE_RECT3D,
 FW_SHAPE_TRIANGLE_MESH, FW_SHAPE_CONSTANT_MEDIUM, FW_SHAPE_CONE, FW_SHAPE_CYLINDER, FW_SHAPE_DISK) = range(10)
FW_ENV_COLOR, FW_ENV_SKY, FW_ENV_HDR = range(3)
FW_RNG_CTR, FW_RNG_LCG = 0, 1
FW_FLAG_TIME_KERNELS = 1
FW_FLAG_COUNT_DEPOSITS = 2

f32, i32, u32, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_uint64


class fw_vec3(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32)]


class fw_rotor3(C.Structure):
    _fields_ = [("s", f32), ("xy", f32), ("xz", f32), ("yz", f32)]


class fw_texture(C.Structure):
    _fields_ = [("kind", i32), ("color", fw_vec3), ("scale", f32), ("depth", u32), ("odd", i32), ("even", i32),
                ("img_w", u32), ("img_h", u32), ("img_rgb8", C.POINTER(C.c_uint8))]


class fw_material(C.Structure):
    _fields_ = [("kind", i32), ("texture", i32), ("albedo", fw_vec3), ("roughness", f32), ("ref_idx", f32)]


class fw_shape(C.Structure):
    _fields_ = [("kind", i32), ("material", i32), ("radius", f32), ("height", f32), ("phi_max", f32),
                ("inner_radius", f32), ("a_min", f32), ("a_max", f32), ("b_min", f32), ("b_max", f32), ("k", f32), ("flip_normal", i32),
                ("pos", fw_vec3), ("size", fw_vec3),
                ("verts", C.POINTER(f32)), ("n_verts", u32), ("indices", C.POINTER(u32)), ("n_indices", u32),
                ("normals", C.POINTER(f32)), ("uvs", C.POINTER(f32)),
                ("inner", i32), ("density", f32)]


class fw_object(C.Structure):
    _fields_ = [("shape", i32), ("position", fw_vec3), ("rotation", fw_rotor3), ("flip_normals", i32)]


class fw_environment(C.Structure):
    _fields_ = [("kind", i32), ("color", fw_vec3), ("zenith", fw_vec3), ("horizon", fw_vec3),
                ("hdr_w", u32), ("hdr_h", u32), ("hdr_rgb", C.POINTER(f32))]


class fw_scene_desc(C.Structure):
    _fields_ = [("objects", C.POINTER(fw_object)), ("n_objects", u32),
                ("shapes", C.POINTER(fw_shape)), ("n_shapes", u32),
                ("materials", C.POINTER(fw_material)), ("n_materials", u32),
                ("textures", C.POINTER(fw_texture)), ("n_textures", u32),
                ("environment", fw_environment)]


class fw_camera_settings(C.Structure):
    _fields_ = [("cam_pos", fw_vec3), ("look_at", fw_vec3), ("vfov", f32), ("aperture", f32), ("focus_dist", f32)]


class fw_render_params(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("samples", u32), ("gamma", f32), ("use_bvh", i32),
                ("multithreaded", i32), ("camera", fw_camera_settings), ("seed", u64), ("rng_mode", i32),
                ("pixel_ids", C.POINTER(u32)), ("n_pixels", u32), ("paths_per_batch", u32), ("flags", u32),
                ("outputs_on_device", i32), ("stream", C.c_void_p)]


class fw_stats(C.Structure):
    _fields_ = [("samples", u64), ("rays", u64), ("rays_per_depth", u64 * FW_MAX_SEGMENTS),
                ("algorithmic_bytes", u64), ("ms_scene", C.c_double), ("ms_render", C.c_double),
                ("ms_raygen", C.c_double), ("ms_extend", C.c_double), ("ms_shade", C.c_double),
                ("ms_accumulate", C.c_double),
                ("n_extend_launches", u32), ("n_shade_launches", u32), ("n_batches", u32),
                ("tlas_nodes", u32), ("blas_nodes", u32), ("reserved", u32),
                ("bytes_raygen", u64), ("bytes_extend", u64), ("bytes_shade", u64), ("bytes_accumulate", u64),
                ("deposits", u64), ("parked_rays", u64), ("ms_wall", C.c_double), ("ms_d2h", C.c_double)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


def vec3(v):
    return fw_vec3(float(v[0]), float(v[1]), float(v[2]))
