"""ctypes binding of libdvdgan_hip.so (include/dvdgan_hip.h).  Fails loudly: no fallback."""
import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVD_LIB_PATH") or os.path.join(_HERE, "csrc", "libdvdgan_hip.so")    # (override: A/B builds)
_lib = None


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("frames", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("C", C.c_int), ("ldi", C.c_int), ("Cout", C.c_int), ("ldo", C.c_int),
                ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("up2", C.c_int), ("relu_in", C.c_int),
                ("nsplit", C.c_int), ("act", C.c_int), ("out_f32", C.c_int), ("ldres", C.c_int),
                ("ldmask", C.c_int), ("res_up2", C.c_int),
                ("inp", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("mask", C.c_void_p), ("out", C.c_void_p), ("ws", C.c_void_p), ("wq", C.c_void_p), ("wq_kind", C.c_int),
                ("pool2", C.c_int)]


class WgradDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("frames", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("C", C.c_int), ("ldx", C.c_int), ("Cin_real", C.c_int), ("Cout", C.c_int), ("Cy", C.c_int),
                ("ldy", C.c_int), ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int), ("up2", C.c_int),
                ("relu_in", C.c_int), ("msplit", C.c_int),
                ("s_co", C.c_longlong), ("s_ci", C.c_longlong), ("s_tap", C.c_longlong),
                ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("ws", C.c_void_p),
                ("overwrite", C.c_int)]


def lib():
    """The loaded shared library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(dvd_gan_amd has no CPU / eager fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.dvd_strerror.restype = C.c_char_p
        _lib.dvd_conv_wgrad_ws_floats.restype = C.c_longlong
        _lib.dvd_sepattn_work_floats.restype = C.c_longlong
        _lib.dvd_conv_fragment_major_bytes.restype = C.c_longlong
        _lib.dvd_convgru_ws_floats.restype = C.c_longlong
        _lib.dvd_conv_thin_image_bytes.restype = C.c_longlong
        _lib.dvd_conv_thin_out_image_bytes.restype = C.c_longlong
        _lib.dvd_convgru_stack_ws_floats.restype = C.c_longlong
        _lib.dvd_cbn_backward_ws_floats.restype = C.c_longlong
        if _lib.dvd_abi_version() != ABI_VERSION:
            raise RuntimeError("libdvdgan_hip.so ABI version mismatch: rebuild it")
        # layout handshake: every descriptor mirror below must have the size the library was compiled with
        for which, mirror in STRUCT_MIRRORS.items():
            want = _lib.dvd_struct_size(which)
            if want != C.sizeof(mirror):
                _lib = None
                raise RuntimeError(f"libdvdgan_hip.so: sizeof descriptor {which} is {want}, the ctypes mirror {mirror.__name__} has "
                                   f"{C.sizeof(mirror)} bytes -- lib.py and include/dvdgan_hip.h disagree")
    return _lib


ABI_VERSION = 13
BN_NREP = 16            # DVD_BN_NREP of include/dvdgan_hip.h
SN_SCRATCH = 512        # DVD_SN_SCRATCH


def check(code):
    if code != 0:
        reset_gru_tickets()           # a failed / aborted launch may have left split-K tickets behind: the next one must start from zero
        raise RuntimeError(f"libdvdgan_hip: {lib().dvd_strerror(code).decode()} (code {code})")


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported storage dtype {t.dtype}")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_tickets = {}


def reset_gru_tickets(current_stream_only=False):
    """Zero the cached ticket buffers (all of them, or the current stream's; the fill runs on the current stream).  The in-launch split-K combine assumes zeroed counters and
    restores them itself, but only when a launch runs to completion: called after any failed library call and once per training
    step (one 32 KB fill), so a faulted or killed launch cannot make later ConvGRU passes skip or mis-time their gate epilogues."""
    for (index, st), t in list(_tickets.items()):
        try:
            if current_stream_only and st != torch.cuda.current_stream(t.device).cuda_stream:
                continue
            t.zero_()
        except Exception:
            pass


def gru_tickets(device):
    """The zero-initialised counters of dvd_gru_desc.tickets for the current stream of `device` (every launch leaves them at
    zero, so one buffer per stream serves every ConvGRU layer issued on it)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(GRU_TICKETS, dtype=torch.int32, device=device)
    return t


class GruDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("T", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("hidden", C.c_int), ("k", C.c_int), ("gx_stride", C.c_longlong),
                ("gx", C.c_void_p), ("w_ur", C.c_void_p), ("w_o", C.c_void_p), ("wd_ur", C.c_void_p),
                ("wd_o", C.c_void_p), ("h0", C.c_void_p),
                ("h_all", C.c_void_p), ("u_all", C.c_void_p), ("r_all", C.c_void_p), ("o_all", C.c_void_p),
                ("hr_all", C.c_void_p), ("h32", C.c_void_p), ("ws", C.c_void_p),
                ("dh_out", C.c_void_p), ("dg", C.c_void_p), ("carry", C.c_void_p), ("dh0", C.c_void_p),
                ("infer", C.c_int),
                ("w_ur_q", C.c_void_p), ("w_o_q", C.c_void_p), ("wd_ur_q", C.c_void_p), ("wd_o_q", C.c_void_p),
                ("tickets", C.c_void_p), ("combine_max", C.c_int), ("ns_cap", C.c_int)]


GRU_TICKETS = 8192                      # == DVD_GRU_TICKETS


GRU_STACK_MAX = 4                       # == DVD_GRU_STACK_MAX


class GruStackDesc(C.Structure):        # == dvd_gru_stack_desc
    _fields_ = [("n_layers", C.c_int), ("layer_policy", C.c_int), ("run", C.c_int), ("cin", C.c_int * GRU_STACK_MAX),
                ("layer", GruDesc * GRU_STACK_MAX),
                ("wx", C.c_void_p * GRU_STACK_MAX), ("wx_q", C.c_void_p * GRU_STACK_MAX), ("bx", C.c_void_p * GRU_STACK_MAX),
                ("wdx", C.c_void_p * GRU_STACK_MAX), ("wdx_q", C.c_void_p * GRU_STACK_MAX), ("dh_mid", C.c_void_p * GRU_STACK_MAX),
                ("ws", C.c_void_p)]


class SnItem(C.Structure):              # == dvd_sn_item
    _fields_ = [("W", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p),
                ("wf", C.c_void_p), ("wd", C.c_void_p), ("h", C.c_int), ("w", C.c_int),
                ("dtype", C.c_int), ("cout", C.c_int), ("cin", C.c_int), ("ntaps", C.c_int), ("cip", C.c_int), ("cop", C.c_int),
                ("blk_wtu", C.c_int), ("blk_wv", C.c_int), ("blk_pack", C.c_int), ("pad_", C.c_int)]


class PackItem(C.Structure):            # == dvd_pack_item
    _fields_ = [("w", C.c_void_p), ("sigma", C.c_void_p), ("wf", C.c_void_p), ("wd", C.c_void_p),
                ("Cout", C.c_int), ("Cin", C.c_int), ("ntaps", C.c_int), ("Cip", C.c_int), ("co_off", C.c_int),
                ("co_tot_f", C.c_int), ("co_tot_d", C.c_int), ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("ci_off", C.c_int), ("ci_tot", C.c_int)]


class FragItem(C.Structure):            # == dvd_frag_item
    _fields_ = [("w", C.c_void_p), ("wq", C.c_void_p), ("ntaps", C.c_int), ("Cout", C.c_int), ("C", C.c_int)]


# dvd_struct_size(which) -> ctypes mirror (DVD_STRUCT_* of include/dvdgan_hip.h)
STRUCT_MIRRORS = {0: ConvDesc, 1: WgradDesc, 2: GruDesc, 3: SnItem, 4: GruStackDesc, 5: PackItem, 6: FragItem}
