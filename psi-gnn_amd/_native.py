"""ctypes binding of libpsignn_hip.so (C ABI declared in include/psignn_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``) on the
current HIP stream; torch only supplies memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpsignn_hip.so")

D = 10


class NativeError(RuntimeError):
    pass


class SolveInfo(C.Structure):
    _fields_ = [("nstep", C.c_int32), ("n_iter", C.c_int32), ("prot_break", C.c_int32),
                ("stop_reason", C.c_int32), ("lowest", C.c_double), ("lowest_abs", C.c_double)]


_P = C.c_void_p
_I64 = C.c_int64
_INT = C.c_int

# name -> (restype, argtypes); mirrors include/psignn_hip.h one to one
SIGNATURES = {
    "psignn_last_error": (C.c_char_p, []),
    "psignn_version": (_INT, []),
    "psignn_plan_create": (_INT, [C.POINTER(_P), _I64, _I64, _P, _P, _P, _P, _INT, _P, _INT, _P]),
    "psignn_plan_is_tiled": (_INT, [_P]),
    "psignn_plan_num_tiles": (_I64, [_P]),
    "psignn_plan_ell_rows": (_I64, [_P]),
    "psignn_plan_max_tile_rows": (_INT, [_P]),
    "psignn_plan_permute": (_INT, [_P, _P, _INT, _P, _INT, _P]),
    "psignn_f_forward_p": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_picard_p": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _INT, _P]),
    "psignn_plan_destroy": (None, [_P]),
    "psignn_plan_num_nodes": (_I64, [_P]),
    "psignn_plan_num_edges": (_I64, [_P]),
    "psignn_plan_num_nonself_edges": (_I64, [_P]),
    "psignn_plan_export": (_INT, [_P, _INT, _P, C.c_size_t]),
    "psignn_weights_size": (_I64, [_INT, _INT]),
    "psignn_f_workspace_floats": (_I64, [_P]),
    "psignn_f_forward": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_phi": (_INT, [_P, _P, _INT, _INT, _INT, _P, _P, _P, _P]),
    "psignn_f_jvp": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_f_jvp_p": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P]),
    "psignn_lin_create": (_INT, [_P, _P]),
    "psignn_lin_destroy": (None, [_P]),
    "psignn_lin_bytes": (C.c_size_t, [_P]),
    "psignn_lin_build": (_INT, [_P, _P, _INT, _P, _P, _P, _P]),
    "psignn_lin_jvp": (_INT, [_P, _P, _INT, _P, _P, _P]),
    "psignn_f_vjp": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_f_vjp_p": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_param_grad_size": (_I64, [_INT, _INT]),
    "psignn_f_param_vjp_workspace_floats": (_I64, [_P]),
    "psignn_f_param_vjp": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_f_param_vjp_p": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_f_vjp_backward_workspace_floats": (_I64, [_P]),
    "psignn_f_vjp_backward": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_mlp2_backward_workspace_floats": (_I64, [_I64]),
    "psignn_mlp2_backward": (_INT, [_P, _P, _I64, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_residual_t": (_INT, [_P, _P, _P, _P, _P]),
    "psignn_dsgps_weights_size": (_I64, [_INT]),
    "psignn_dsgps_forward": (_INT, [_P, _P, _INT, _P, _P, _P, _P, _P, _P]),
    "psignn_dsgps_step_p": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_dsgps_grad_size": (_I64, [_INT]),
    "psignn_dsgps_step_backward_workspace_floats": (_I64, [_P]),
    "psignn_dsgps_step_backward": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_dss_weights_size": (_I64, [_INT]),
    "psignn_dss_forward": (_INT, [_P, _P, _INT, C.c_float, _P, _P, _P, _P]),
    "psignn_dss_step_p": (_INT, [_P, _P, _INT, C.c_float, _P, _P, _P, _P]),
    "psignn_dss_grad_size": (_I64, []),
    "psignn_dss_step_backward_workspace_floats": (_I64, [_P]),
    "psignn_dss_step_backward": (_INT, [_P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "psignn_mlp2": (_INT, [_P, _I64, _INT, _INT, _INT, _P, _P, _P, _P, _P, _P]),
    "psignn_residual": (_INT, [_P, _P, _P, _P, _P]),
    "psignn_broyden_create": (_INT, [C.POINTER(_P), _P, _INT, _INT]),
    "psignn_broyden_create_for_batch": (_INT, [C.POINTER(_P), _P, _INT, _INT, _I64]),
    "psignn_broyden_create_n": (_INT, [C.POINTER(_P), _I64, _INT, _INT, _INT]),
    "psignn_broyden_destroy": (None, [_P]),
    "psignn_broyden_bytes": (C.c_size_t, [_P]),
    "psignn_broyden_set_stop_mode": (_INT, [_P, _INT]),
    "psignn_broyden_solve": (_INT, [_P, _P, _INT, _P, _P, _P, C.c_double, _INT, _P, C.POINTER(SolveInfo),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "psignn_broyden_solve_adjoint": (_INT, [_P, _P, _INT, _P, _P, _P, _P, C.c_double, _INT, _P, C.POINTER(SolveInfo),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "psignn_broyden_solve_batch": (_INT, [_INT, C.POINTER(_P), _P, _INT, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.c_double, _INT,
                                          C.POINTER(_P), C.POINTER(SolveInfo), C.POINTER(C.POINTER(C.c_double)),
                                          C.POINTER(C.POINTER(C.c_double)), _P]),
    "psignn_broyden_batchable": (_INT, [_INT, C.POINTER(_P)]),
    "psignn_broyden_get_iterate": (_INT, [_P, _INT, _P, _P]),
    "psignn_broyden_get_pair": (_INT, [_P, _INT, _INT, _P, _P]),
    "psignn_broyden_ext_begin": (_INT, [_P, _P, _P, _P]),
    "psignn_broyden_ext_next_x": (_INT, [_P, _P, _P]),
    "psignn_broyden_ext_trial_x": (_INT, [_P, C.c_double, _P, _P]),
    "psignn_broyden_ext_scale_step": (_INT, [_P, C.c_double, _P]),
    "psignn_broyden_ext_update": (_INT, [_P, _P, C.c_double, C.POINTER(_INT), _P]),
    "psignn_broyden_ext_finish": (_INT, [_P, _P, C.POINTER(SolveInfo), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), _P]),
    "psignn_fpiter_create": (_INT, [C.POINTER(_P), _I64, _INT, _INT, _INT]),
    "psignn_fpiter_destroy": (None, [_P]),
    "psignn_fpiter_bytes": (C.c_size_t, [_P]),
    "psignn_fpiter_poll": (_INT, [_P, C.POINTER(_INT), _P]),
    "psignn_picard_begin": (_INT, [_P, _P, _P]),
    "psignn_picard_current_x": (_INT, [_P, _P, _P]),
    "psignn_picard_update": (_INT, [_P, _P, C.c_double, C.POINTER(_INT), _P]),
    "psignn_anderson_begin": (_INT, [_P, _P, _P, _P, C.c_double, C.c_double, _INT, _P]),
    "psignn_anderson_next_x": (_INT, [_P, _P, _P]),
    "psignn_anderson_update": (_INT, [_P, _P, C.c_double, C.POINTER(_INT), _P]),
    "psignn_fpiter_finish": (_INT, [_P, _P, C.POINTER(SolveInfo), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32), _P]),
    "psignn_fpiter_get_iterate": (_INT, [_P, _INT, _P, _P]),
    "psignn_gmres_create": (_INT, [C.POINTER(_P), _I64, _I64, _INT, _P]),
    "psignn_gmres_destroy": (None, [_P]),
    "psignn_residual_norms": (_INT, [_P, _P, _P, _P, _P, C.POINTER(C.c_double), _P]),
    "psignn_gmres_begin": (_INT, [_P, _P, _P]),
    "psignn_gmres_step": (_INT, [_P, _INT, C.c_double, C.c_double, C.POINTER(_INT), _P]),
    "psignn_gmres_solution": (_INT, [_P, _INT, _P, C.c_double, _P, C.POINTER(C.c_double), _P]),
    "psignn_gmres_history": (_INT, [_P, C.POINTER(C.c_double), _P]),
    "psignn_gmres_reorth_count": (_INT, [_P, C.POINTER(C.c_int), _P]),
    "psignn_prof_enable": (None, [_INT]),
    "psignn_reload_knobs": (None, []),
    "psignn_prof_tile_stamps": (None, [_P]),
    "psignn_prof_collect": (_INT, []),
    "psignn_prof_get": (_INT, [_INT, C.c_char_p, _INT, C.POINTER(_I64), C.POINTER(C.c_double)]),
    "psignn_prof_get2": (_INT, [_INT, C.c_char_p, _INT, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(_I64)]),
    "psignn_prof_launch": (_INT, [_INT, C.c_char_p, _INT, C.POINTER(C.c_double), C.POINTER(_I64)]),
}


def prof_enable(on: bool):
    lib().psignn_prof_enable(int(on))


def prof_collect(with_bytes=False):
    """{kernel name: (calls, total_ms)} of everything launched since the last collect (HIP events); ``with_bytes``:
    (calls, total_ms, algorithmic bytes as stated at the launch sites)."""
    l = lib()
    out = {}
    for i in range(l.psignn_prof_collect()):
        name = C.create_string_buffer(64)
        calls, ms, byts = _I64(0), C.c_double(0.0), _I64(0)
        check(l.psignn_prof_get2(i, name, 64, C.byref(calls), C.byref(ms), C.byref(byts)), "psignn_prof_get2")
        out[name.value.decode()] = (int(calls.value), float(ms.value), int(byts.value)) if with_bytes else (int(calls.value), float(ms.value))
    return out

def prof_launch_log():
    """[(kernel name, ms, algorithmic bytes)] of the launches of the last ``prof_collect``, in launch order."""
    l = lib()
    out = []
    for i in range(l.psignn_prof_launch(0, None, 0, None, None)):
        name = C.create_string_buffer(64)
        ms, byts = C.c_double(0.0), _I64(0)
        check(l.psignn_prof_launch(i, name, 64, C.byref(ms), C.byref(byts)), "psignn_prof_launch")
        out.append((name.value.decode(), float(ms.value), int(byts.value)))
    return out


_lib = None


def _build_once():
    """A source checkout without the built library (the .so is kept out of git): compile it in place when hipcc is there.
    Nothing else is attempted -- without the library every entry point raises."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(os.path.dirname(LIB_PATH), "csrc")
    if not (os.path.exists(hipcc) and os.path.exists(os.path.join(csrc, "Makefile"))):
        return
    import fcntl
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:   # one builder when several ranks start together
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(LIB_PATH):
            print(f"[psi-gnn_amd] {os.path.basename(LIB_PATH)} missing: building it with {hipcc} (make -C {csrc})", flush=True)
            r = subprocess.run(["make", "-C", csrc, "-j8", f"HIPCC={hipcc}"], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0 or not os.path.exists(LIB_PATH):
                raise NativeError(f"building {LIB_PATH} failed (make exit code {r.returncode}); last lines of its output:\n"
                                  + "\n".join(r.stdout.splitlines()[-25:]))


def lib():
    """Load the shared library (once).  Raises NativeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            _build_once()
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C psi-gnn_amd/csrc`.  There is no CPU fallback for the HIP path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().psignn_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed (code {rc}): {msg}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
    """Device pointer of a contiguous tensor (or NULL)."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise NativeError("tensor handed to the HIP path must be contiguous")
    return t.data_ptr()


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise NativeError(f"{name} is on {t.device}: the HIP path needs device tensors and has no CPU fallback")
