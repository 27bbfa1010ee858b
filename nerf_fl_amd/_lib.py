"""ctypes binding of libnerf_fl_amd.so (the C ABI in include/nerf_fl_amd.h).

There is no fallback: if the shared library is missing or does not export the
ABI this package was written against, importing a renderer entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NFL_LIB selects a diagnostic / variant build of the library (make diag, make variant) -- honoured only in a
# developer session (NERF_FL_AMD_DEV=1), so that a stray environment variable cannot swap the library under the product
LIB_PATH = ((os.environ.get("NFL_LIB") if os.environ.get("NERF_FL_AMD_DEV") == "1" else None)
            or os.path.join(_HERE, "libnerf_fl_amd.so"))

NFL_ABI_VERSION = 9
NFL_GMAX_SLOTS = 1024
NFL_PREC_F16X3 = 0
NFL_PREC_F16 = 1
NFL_PREC_F16W = 2      # backward only: one-product arithmetic, but the gradient chain reads hi + lo weight fragments
NFL_STATUS_NONFINITE = 1
NFL_STATUS_RANGE = 2
NFL_NUM_LAYERS = 19

# layer slot -> state_dict prefix (reference models/nerf.py:121-151)
LAYER_NAMES = (
    [f"xyz_encoding_{i + 1}.0" for i in range(8)]
    + ["xyz_encoding_final", "dir_encoding.0", "static_sigma.0", "static_rgb.0"]
    + [f"transient_encoding.{j}" for j in (0, 2, 4, 6)]
    + ["transient_sigma.0", "transient_rgb.0", "transient_beta.0"]
)
assert len(LAYER_NAMES) == NFL_NUM_LAYERS


class FieldDesc(C.Structure):
    _fields_ = [
        ("n_emb_xyz", C.c_int32), ("n_emb_dir", C.c_int32),
        ("encode_appearance", C.c_int32), ("n_a", C.c_int32),
        ("encode_transient", C.c_int32), ("n_tau", C.c_int32),
        ("beta_min", C.c_float), ("reserved", C.c_int32),
    ]


class FieldParams(C.Structure):
    _fields_ = [("weight", C.c_void_p * NFL_NUM_LAYERS), ("bias", C.c_void_p * NFL_NUM_LAYERS)]


class PackJob(C.Structure):
    _fields_ = [("h_plan", C.c_void_p), ("d_plan", C.c_void_p), ("params", C.POINTER(FieldParams)), ("d_packed", C.c_void_p),
                ("packed_bytes", C.c_size_t), ("d_status", C.c_void_p)]


NFL_PACK_MAX_JOBS = 4


class Camera(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("reserved", C.c_int32), ("pix0", C.c_int64), ("near", C.c_float), ("far", C.c_float)]


class PassArgs(C.Structure):
    _fields_ = [
        ("d_rays", C.c_void_p), ("h_cam", C.POINTER(Camera)), ("d_view_dir", C.c_void_p),
        ("n_rays", C.c_int32), ("n_samples", C.c_int32),
        ("d_z", C.c_void_p), ("d_lin", C.c_void_p), ("d_perturb_rand", C.c_void_p),
        ("perturb", C.c_float), ("use_disp", C.c_int32),
        ("d_z_out", C.c_void_p),
        ("d_noise", C.c_void_p), ("noise_std", C.c_float),
        ("d_a_emb", C.c_void_p), ("d_t_emb", C.c_void_p),
        ("sigma_only", C.c_int32), ("white_back", C.c_int32),
        ("test_extras", C.c_int32), ("stash_split", C.c_int32),
        ("d_weights", C.c_void_p), ("d_opacity", C.c_void_p), ("d_rgb", C.c_void_p), ("d_depth", C.c_void_p),
        ("d_transient_sigmas", C.c_void_p), ("d_beta", C.c_void_p),
        ("d_rgb_static", C.c_void_p), ("d_rgb_transient", C.c_void_p),
        ("d_rgb_static_only", C.c_void_p), ("d_depth_static_only", C.c_void_p),
        ("d_rgb_transient_only", C.c_void_p), ("d_depth_transient_only", C.c_void_p),
        ("d_field_raw", C.c_void_p), ("d_act_stash", C.c_void_p),
        ("d_pe_w_xyz", C.c_void_p), ("d_pe_w_dir", C.c_void_p),
        ("d_loss_target", C.c_void_p), ("d_losses", C.c_void_p), ("d_seed_rgb", C.c_void_p), ("d_seed_beta", C.c_void_p),
        ("loss_coef", C.c_float), ("lambda_u", C.c_float), ("loss_slot", C.c_int32), ("reserved2", C.c_int32),
        ("d_status", C.c_void_p),
        ("d_embedded", C.c_void_p), ("n_points", C.c_int32), ("embedded_stride", C.c_int32),
        ("d_cam", C.c_void_p),
    ]


class CompBwdArgs(C.Structure):
    _fields_ = [
        ("d_field_raw", C.c_void_p), ("d_z", C.c_void_p), ("d_noise", C.c_void_p), ("noise_std", C.c_float),
        ("n_rays", C.c_int32), ("n_samples", C.c_int32), ("use_transient", C.c_int32), ("white_back", C.c_int32),
        ("reserved", C.c_int32),
        ("g_weights", C.c_void_p), ("g_opacity", C.c_void_p), ("g_rgb", C.c_void_p), ("g_depth", C.c_void_p),
        ("g_transient_sigmas", C.c_void_p), ("g_beta", C.c_void_p), ("g_rgb_static", C.c_void_p),
        ("g_rgb_transient", C.c_void_p), ("g_tsig_const", C.c_float), ("reserved3", C.c_int32), ("d_go", C.c_void_p),
        ("d_head_grads", C.c_void_p), ("d_gmax", C.c_void_p),
    ]


class DgradArgs(C.Structure):
    _fields_ = [
        ("d_head_grads", C.c_void_p), ("d_act_stash", C.c_void_p), ("d_grad_stash", C.c_void_p),
        ("n_rays", C.c_int32), ("n_samples", C.c_int32), ("use_transient", C.c_int32), ("dir_is_data", C.c_int32),
        ("d_g_a_emb", C.c_void_p), ("d_g_t_emb", C.c_void_p), ("d_latent_row", C.c_void_p),
        ("d_g_rays", C.c_void_p), ("d_rays", C.c_void_p), ("d_z", C.c_void_p),
        ("d_pe_w_xyz", C.c_void_p), ("d_pe_w_dir", C.c_void_p), ("d_gmax", C.c_void_p), ("rounding_seed", C.c_uint32),
    ]


NFL_ADAM_MAX_TENSORS = 64


class AdamTensors(C.Structure):
    _fields_ = [("param", C.c_void_p * NFL_ADAM_MAX_TENSORS), ("grad", C.c_void_p * NFL_ADAM_MAX_TENSORS),
                ("exp_avg", C.c_void_p * NFL_ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * NFL_ADAM_MAX_TENSORS),
                ("numel", C.c_int32 * NFL_ADAM_MAX_TENSORS)]


class LossArgs(C.Structure):
    _fields_ = [("d_rgb_coarse", C.c_void_p), ("d_rgb_fine", C.c_void_p), ("d_beta", C.c_void_p),
                ("d_transient_sigmas", C.c_void_p), ("d_target", C.c_void_p), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("coef", C.c_float), ("lambda_u", C.c_float), ("d_losses", C.c_void_p), ("d_grad_loss", C.c_void_p * 4),
                ("d_g_rgb_coarse", C.c_void_p), ("d_g_rgb_fine", C.c_void_p), ("d_g_beta", C.c_void_p),
                ("d_g_transient_sigmas", C.c_void_p)]


class FieldGrads(C.Structure):
    _fields_ = [("weight", C.c_void_p * NFL_NUM_LAYERS), ("bias", C.c_void_p * NFL_NUM_LAYERS)]


# every symbol include/nerf_fl_amd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("nfl_plan_bytes", C.c_size_t, [C.POINTER(FieldDesc)]),
    ("nfl_plan_build", C.c_int, [C.POINTER(FieldDesc), C.c_int, C.c_void_p, C.c_size_t]),
    ("nfl_packed_bytes", C.c_size_t, [C.POINTER(FieldDesc), C.c_int]),
    ("nfl_param_count", C.c_size_t, [C.POINTER(FieldDesc)]),
    ("nfl_pack_field", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(FieldParams), C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_void_p]),
    ("nfl_pack_fields", C.c_int, [C.c_int32, C.POINTER(PackJob), C.c_void_p]),
    ("nfl_compose_forward", C.c_int, [C.POINTER(FieldParams), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    ("nfl_render_pass", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PassArgs), C.c_void_p]),
    ("nfl_sample_pdf", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nfl_field_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p]),
    ("nfl_posenc", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nfl_gen_rays", C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int64,
                               C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    ("nfl_act_stash_bytes", C.c_size_t, [C.POINTER(FieldDesc), C.c_int32, C.c_int32, C.c_int32]),
    ("nfl_grad_stash_bytes", C.c_size_t, [C.POINTER(FieldDesc), C.c_int32, C.c_int32, C.c_int32]),
    ("nfl_bwd_plan_build", C.c_int, [C.POINTER(FieldDesc), C.c_int32, C.c_int32, C.c_void_p, C.c_size_t]),
    ("nfl_bwd_packed_bytes", C.c_size_t, [C.POINTER(FieldDesc), C.c_int32, C.c_int32]),
    ("nfl_composite_backward", C.c_int, [C.POINTER(CompBwdArgs), C.c_void_p]),
    ("nfl_mlp_dgrad", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DgradArgs), C.c_void_p]),
    ("nfl_wgrad_plan_bytes", C.c_size_t, []),
    ("nfl_wgrad_plan_build", C.c_int, [C.POINTER(FieldDesc), C.c_int32, C.c_void_p, C.c_size_t]),
    ("nfl_wgrad_scratch_bytes", C.c_size_t, []),
    ("nfl_mlp_wgrad", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(FieldParams), C.c_void_p, C.POINTER(FieldGrads), C.c_void_p]),
    ("nfl_adam_step", C.c_int, [C.POINTER(AdamTensors), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                C.c_void_p]),
    ("nfl_adam_step_dev", C.c_int, [C.POINTER(AdamTensors), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("nfl_loss_forward", C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    ("nfl_loss_backward", C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    ("nfl_abi_version", C.c_int, []),
    ("nfl_version", C.c_char_p, []),
    ("nfl_strerror", C.c_char_p, [C.c_int]),
    ("nfl_render_kernel_name", C.c_char_p, [C.c_int, C.c_int]),
]

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"nerf_fl_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nerf_fl_amd/csrc`. There is no CPU or eager-PyTorch fallback for render_rays.")
    h = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(h, name)            # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if h.nfl_abi_version() != NFL_ABI_VERSION:
        raise RuntimeError(f"nerf_fl_amd: ABI mismatch (library {h.nfl_abi_version()}, python {NFL_ABI_VERSION})")
    _lib = h
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"nerf_fl_amd: {what} failed: {lib().nfl_strerror(code).decode()} ({code})")
