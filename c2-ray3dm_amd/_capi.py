"""ctypes view of the C ABI in include/c2ray_hip.h (libc2ray_hip.so, built by csrc/Makefile).

There is no CPU fallback: if the HIP library is missing or no GPU is present, loading or
creating a context fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libc2ray_hip.so")
MAX_ITER_LOG = 128

GRID_NDENS, GRID_XH, GRID_XH_AV, GRID_XH_INTERMED, GRID_PHIH, GRID_PHIHEAT, GRID_TEMPER, GRID_XH0, GRID_XH_AV0, GRID_XH_INTERMED0 = range(10)


class Params(C.Structure):
    _fields_ = [("mesh", C.c_int32 * 3), ("device", C.c_int32), ("subboxsize", C.c_int32),
                ("max_subbox", C.c_int32), ("numtau", C.c_int32), ("max_outer_iter", C.c_int32),
                ("max_chem_iter", C.c_int32), ("deterministic_rates", C.c_int32),
                ("sweep_mode", C.c_int32), ("allfrac", C.c_int32),
                ("epsilon", C.c_double), ("convergence_fraction", C.c_double),
                ("minimum_fractional_change", C.c_double), ("minimum_fraction_of_atoms", C.c_double),
                ("loss_fraction", C.c_double), ("max_coldensh", C.c_double),
                ("tau_photo_limit", C.c_double), ("sigma_HI", C.c_double),
                ("minlogtau", C.c_double), ("dlogtau", C.c_double), ("weight_floor", C.c_double),
                ("sqrt2", C.c_double), ("sqrt3", C.c_double), ("pi", C.c_double),
                ("abu_c", C.c_double), ("bh00", C.c_double), ("albpow", C.c_double),
                ("colh0", C.c_double), ("temph0", C.c_double), ("S_star", C.c_double),
                ("scratch_bytes", C.c_size_t)]


class Report(C.Structure):
    _fields_ = [("niter", C.c_int32), ("converged", C.c_int32), ("conv_flag", C.c_int64),
                ("conv_criterion", C.c_int64), ("sum_nbox_all", C.c_int64), ("visited", C.c_int64),
                ("photon_loss_all", C.c_double), ("seconds_sweep", C.c_double),
                ("seconds_chem", C.c_double), ("chem_not_converged", C.c_int32),
                ("timing_split", C.c_int32),
                ("it_conv_flag", C.c_int64 * MAX_ITER_LOG), ("it_sum_nbox", C.c_int64 * MAX_ITER_LOG),
                ("it_rel_change_xh1", C.c_double * MAX_ITER_LOG),
                ("it_rel_change_xh0", C.c_double * MAX_ITER_LOG),
                ("it_sum_xh1", C.c_double * MAX_ITER_LOG),
                ("h0_before", C.c_double), ("h1_before", C.c_double), ("h0_after", C.c_double),
                ("h1_after", C.c_double), ("totrec", C.c_double), ("totcollisions", C.c_double),
                ("dh0", C.c_double), ("total_ion", C.c_double), ("totalsrc", C.c_double),
                ("photcons", C.c_double), ("it_photcons", C.c_double * MAX_ITER_LOG),
                ("seconds_upload", C.c_double), ("seconds_download", C.c_double), ("seconds_total", C.c_double)]


class SedParams(C.Structure):
    _fields_ = [("T_eff", C.c_double), ("S_star", C.c_double), ("min_freq", C.c_double),
                ("max_freq", C.c_double), ("pl_index_cross_section", C.c_double), ("hplanck", C.c_double),
                ("k_B", C.c_double), ("two_pi_over_c_square", C.c_double), ("R_solar", C.c_double),
                ("pi", C.c_double), ("minlogtau", C.c_double), ("maxlogtau", C.c_double),
                ("numtau", C.c_int32), ("sed_type", C.c_int32), ("pl_index", C.c_double),
                ("grey", C.c_int32), ("reserved1", C.c_int32)]


class ThermalParams(C.Structure):
    _fields_ = [("tau_heat_limit", C.c_double), ("k_B", C.c_double), ("gamma1", C.c_double),
                ("minitemp", C.c_double), ("relative_denergy", C.c_double),
                ("thermal_rate_floor", C.c_double), ("thermal_time_tol", C.c_double),
                ("temp_conv_rel", C.c_double), ("temp_conv_abs", C.c_double),
                ("H0", C.c_double), ("Omega0", C.c_double),
                ("cool_mintemp", C.c_double), ("cool_dtemp", C.c_double),
                ("cool_points", C.c_int32), ("thermal_max_steps", C.c_int32),
                ("cosmological", C.c_int32), ("reserved0", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
# c2r_reduce_scatter_fn / c2r_allgather_fn: (user, dev_buf, offsets[nranks], counts[nranks], nranks, hip_stream)
SLAB_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int32, C.c_void_p)
ITERATION_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_double)
# c2r_next_sources_fn: (user, pass, want, *first, *count)
NEXT_SOURCES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32))

# every symbol include/c2ray_hip.h declares: (name, restype, argtypes)
_P, _I32, _I64, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = [
    ("c2r_default_params", C.c_int, [C.POINTER(Params)]),
    ("c2r_create", C.c_int, [C.POINTER(_P), C.POINTER(Params)]),
    ("c2r_destroy", None, [_P]),
    ("c2r_last_error", C.c_char_p, [_P]),
    ("c2r_info", C.c_char_p, [_P]),
    ("c2r_set_stream", C.c_int, [_P, _P]),
    ("c2r_set_option", C.c_int, [_P, C.c_char_p, _D]),
    ("c2r_set_tables", C.c_int, [_P, _P, _P, _I32]),
    ("c2r_set_step", C.c_int, [_P, C.POINTER(_D * 3), _D, _D, C.c_float, _D]),
    ("c2r_set_lls", C.c_int, [_P, _I32, _P, _D]),
    ("c2r_set_clumping_grid", C.c_int, [_P, _P]),
    ("c2r_default_thermal", C.c_int, [C.POINTER(ThermalParams)]),
    ("c2r_set_thermal", C.c_int, [_P, C.POINTER(ThermalParams), _P, _P, _I32, _P]),
    ("c2r_set_redshift", C.c_int, [_P, _D]),
    ("c2r_set_final_temperature", C.c_int, [_P]),
    ("c2r_set_sources", C.c_int, [_P, _P, _P, _I32]),
    ("c2r_set_exchange_overlap", C.c_int, [_P, _I32]),
    ("c2r_set_xray_tables", C.c_int, [_P, _P, _P, _I32]),
    ("c2r_set_xray_heat_tables", C.c_int, [_P, _P, _P, _I32]),
    ("c2r_set_xray_sources", C.c_int, [_P, _P, _I32]),
    ("c2r_set_rank", C.c_int, [_P, _I32, _I32, ALLREDUCE_FN, _P]),
    ("c2r_set_slab_chemistry", C.c_int, [_P, _P, _P, _P]),
    ("c2r_slab", C.c_int, [_P, _I32, _I32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("c2r_set_source_share", C.c_int, [_P, _P, _I32]),
    ("c2r_last_nbox", C.c_int, [_P, _P, _I32]),
    ("c2r_set_balance", C.c_int, [_P, _I32]),
    ("c2r_source_share", C.c_int, [_P, _P, _I32, C.POINTER(_I32)]),
    ("c2r_balanced_shares", C.c_int, [_P, _I32, _I32, _I32, _P, C.POINTER(_I32)]),
    ("c2r_set_source_queue", C.c_int, [_P, _P, _P, _I32]),
    ("c2r_get_device", C.c_int, [_P, C.POINTER(_I32)]),
    ("c2r_bind_device_buffers", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("c2r_device_ptr", C.c_int, [_P, _I32, C.POINTER(_P)]),
    ("c2r_upload", C.c_int, [_P, _I32, _P]),
    ("c2r_download", C.c_int, [_P, _I32, _P]),
    ("c2r_zero_rates", C.c_int, [_P]),
    ("c2r_pass_sources", C.c_int, [_P, C.POINTER(_D), C.POINTER(_I64), C.POINTER(_I64)]),
    ("c2r_allreduce_rates", C.c_int, [_P]),
    ("c2r_exchange_stats", C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    ("c2r_evolve0d_host", C.c_int, [_P, _I32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.POINTER(_D)]),
    ("c2r_global_pass_cell_host", C.c_int, [_P, _D, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(_I32)]),
    ("c2r_do_source", C.c_int, [_P, _I32, _P, C.POINTER(_D), C.POINTER(_I32), C.POINTER(_I64)]),
    ("c2r_global_pass", C.c_int, [_P, _D, C.POINTER(_I64), C.POINTER(_D)]),
    ("c2r_iterate", C.c_int, [_P, _D, C.POINTER(_D), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_D)]),
    ("c2r_evolve3d_restart", C.c_int, [_P, _D, _I32, _D, _P, _P, _P, _P, _P, C.POINTER(Report)]),
    ("c2r_set_iteration_hook", C.c_int, [_P, _P, _P]),
    ("c2r_do_source_host", C.c_int, [_P, _I32, _P, _P, _P, _P, C.POINTER(_D), C.POINTER(_I32)]),
    ("c2r_do_grid_host", C.c_int, [_P, _P, _P, _P, _P, C.POINTER(_D), C.POINTER(_I64)]),
    ("c2r_global_pass_host", C.c_int, [_P, _D, _P, _P, _P, _P, _P, C.POINTER(_I64)]),
    ("c2r_photon_sums", C.c_int, [_P, _I32, _I32, C.POINTER(_D * 4)]),
    ("c2r_sum", C.c_int, [_P, _I32, C.POINTER(_D)]),
    ("c2r_evolve3d_dev", C.c_int, [_P, _D, C.POINTER(Report)]),
    ("c2r_evolve3d_restart_dev", C.c_int, [_P, _D, _I32, _D, C.POINTER(Report)]),
    ("c2r_evolve3d", C.c_int, [_P, _D, _P, _P, _P, _P, _P, C.POINTER(Report)]),
    ("c2r_evolve3d_thermal", C.c_int, [_P, _D, _I32, _D, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Report)]),
    ("c2r_default_sed", C.c_int, [C.POINTER(SedParams)]),
    ("c2r_default_sed_power_law", C.c_int, [C.POINTER(SedParams)]),
    ("c2r_build_tables", C.c_int, [C.POINTER(SedParams), _P, _P, _I32, C.POINTER(_D)]),
    ("c2r_build_heat_tables", C.c_int, [C.POINTER(SedParams), _D, _P, _P, _I32]),
    ("c2r_selftest", C.c_int, [_P, C.POINTER(_I64)]),
    ("c2r_profile", C.c_int, [_P, _I32]),
    ("c2r_profile_read", C.c_int, [_P, C.POINTER(_D), C.POINTER(_I64), C.POINTER(_D), C.POINTER(_I64)]),
]

_lib = None


class C2RayHipError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libc2ray_hip.so and type every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("C2RAY_HIP_LIB") or LIB_PATH     # env override: A/B builds of the library
    # One HIP runtime per process: torch bundles its own libamdhip64; importing it first makes
    # this library bind to the same copy (loading the system copy first leaves torch unable to
    # see the GPU).  A Fortran/C host without torch simply uses the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise C2RayHipError(
            "%s not found: build it with `make -C c2-ray3dm_amd/csrc` (or __graft_entry__.build()); "
            "this package has no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError = a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_params(mesh, device=0):
    lib = load_library()
    p = Params()
    rc = lib.c2r_default_params(C.byref(p))
    if rc:
        raise C2RayHipError("c2r_default_params -> %d" % rc)
    mesh = (mesh,) * 3 if isinstance(mesh, int) else tuple(mesh)
    p.mesh[:] = mesh
    p.device = device
    return p


def build_tables(sed=None):
    """rad_ini on the host (c2r_build_tables): returns (thick, thin, R_star)."""
    import numpy as np
    lib = load_library()
    if sed is None:
        sed = SedParams()
        lib.c2r_default_sed(C.byref(sed))
    n = sed.numtau + 1
    thick, thin, r = np.empty(n), np.empty(n), C.c_double()
    rc = lib.c2r_build_tables(C.byref(sed), thick.ctypes.data, thin.ctypes.data, n, C.byref(r))
    if rc:
        raise C2RayHipError("c2r_build_tables -> %d" % rc)
    return thick, thin, r.value


ION_FREQ_HI = 3.28851300169676800e+15      # cgsphotoconstants.f90: ion_freq_HI = ev2fr * eth0


def build_heat_tables(sed=None, ion_freq_HI=ION_FREQ_HI):
    """The heating tables of a non-isothermal run (c2r_build_heat_tables): returns (heat_thick, heat_thin)."""
    import numpy as np
    lib = load_library()
    if sed is None:
        sed = SedParams()
        lib.c2r_default_sed(C.byref(sed))
    n = sed.numtau + 1
    hk, hn = np.empty(n), np.empty(n)
    rc = lib.c2r_build_heat_tables(C.byref(sed), ion_freq_HI, hk.ctypes.data, hn.ctypes.data, n)
    if rc:
        raise C2RayHipError("c2r_build_heat_tables -> %d" % rc)
    return hk, hn
