"""ctypes binding of libdvbs2_fec_hip.so (include/dvbs2_fec_hip.h).

This is the stub a maintainer of the reference would bind (see INTEGRATION.md for the C++ side);
the tests and bench.py drive the C ABI through it. There is deliberately no fallback: if the
shared library is missing, import of this module raises.
"""
import ctypes as C
import os

# One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and
# resolves it by file name, so it must be loaded BEFORE this library, whose NEEDED libamdhip64.so.7
# then binds to the already-loaded copy. Loading in the other order puts two HIP/HSA runtimes in the
# process and the second one finds no GPU. torch is plumbing here (device buffers, streams, RCCL).
try:
    import torch  # noqa: F401
except ImportError:  # standalone use (e.g. from the GNU Radio blocks): the system ROCm runtime is used
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVBS2_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libdvbs2_fec_hip.so"))  # DVBS2_LIB: kernel experiments

OK, EINVAL, EDEVICE, ESIZE = 0, -1, -2, -3
STANDARD_DVBS2, STANDARD_DVBT2 = 0, 1
FECFRAME_SHORT, FECFRAME_NORMAL, FECFRAME_MEDIUM = 0, 1, 2
OM_CODEWORD, OM_MESSAGE = 0, 1
MOD_QPSK, MOD_8PSK = 0, 4


class FecInfo(C.Structure):
    _fields_ = [("bch_k", C.c_uint32), ("bch_n", C.c_uint32), ("bch_t", C.c_uint32),
                ("ldpc_k", C.c_uint32), ("ldpc_n", C.c_uint32), ("table_k", C.c_uint32),
                ("table", C.c_char * 24)]


class BbDeheaderCounters(C.Structure):
    _fields_ = [("packets", C.c_uint64), ("errors", C.c_uint64), ("bbframes", C.c_uint64), ("dropped", C.c_uint64),
                ("gaps", C.c_uint64), ("overruns", C.c_uint64), ("synched", C.c_int32), ("partial_ts_bytes", C.c_int32)]


# every symbol include/dvbs2_fec_hip.h declares: name -> (restype, argtypes)
_vp, _i, _ip = C.c_void_p, C.c_int, C.POINTER(C.c_int)
SYMBOLS = {
    "dvbs2_last_error": (C.c_char_p, []),
    "dvbs2_device_count": (_i, []),
    "dvbs2_host_register": (_i, [_vp, C.c_size_t]),
    "dvbs2_host_unregister": (_i, [_vp]),
    "dvbs2_host_is_page_locked": (_i, [_vp, C.c_size_t]),
    "dvbs2_host_alloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "dvbs2_host_free": (_i, [_vp]),
    "dvbs2_get_fec_info": (_i, [_i, _i, _i, C.POINTER(FecInfo)]),
    "dvbs2_rate_name": (C.c_char_p, [_i]),
    "dvbs2_rate_from_name": (_i, [C.c_char_p]),
    "dvbs2_ldpc_table_info": (_i, [C.c_char_p, _ip, _ip, _ip, _ip, _ip]),
    "dvbs2_ldpc_table_name": (C.c_char_p, [_i]),
    "dvbs2_ldpc_layer_info": (_i, [C.c_char_p, _i, _ip, _vp, _vp, _i]),
    "dvbs2_ldpc_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    "dvbs2_ldpc_create_table": (_i, [C.POINTER(_vp), C.c_char_p, _i, _i, _i, _i]),
    "dvbs2_ldpc_destroy": (None, [_vp]),
    "dvbs2_ldpc_params": (_i, [_vp, _ip, _ip, _ip, _ip, _ip]),
    "dvbs2_ldpc_decode": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "dvbs2_ldpc_decode_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_ldpc_enqueue_device": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_ldpc_finish": (_i, [_vp]),
    "dvbs2_ldpc_profile": (_i, [_vp, _i, C.POINTER(C.c_double), _ip]),
    "dvbs2_ldpc_kernel_name": (C.c_char_p, [_vp]),
    "dvbs2_ldpc_fallback_rounds": (_i, [_vp]),
    "dvbs2_measure_host_copy": (_i, [_i, C.c_size_t, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dvbs2_measure_shader_clock": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dvbs2_debug_cu_slot_table": (_i, [_i, _i, C.POINTER(C.c_ulonglong), _ip]),
    "dvbs2_bch_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dvbs2_bch_create_raw": (_i, [C.POINTER(_vp), _i, C.c_uint32, _i, _i, _i, _i]),
    "dvbs2_bch_destroy": (None, [_vp]),
    "dvbs2_bch_params": (_i, [_vp, _ip, _ip, _ip]),
    "dvbs2_bch_genpoly": (_i, [_vp, _vp, _i]),
    "dvbs2_bch_decode": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dvbs2_bch_set_descramble": (_i, [_vp, _i]),
    "dvbs2_bb_descramble_sequence": (_i, [_vp, _i]),
    "dvbs2_bch_decode_device": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "dvbs2_demap_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dvbs2_demap_destroy": (None, [_vp]),
    "dvbs2_demap_params": (_i, [_vp, _ip, _ip, _ip, _ip]),
    "dvbs2_demap_soft": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "dvbs2_demap_soft_device": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "dvbs2_demap_estimate_snr": (_i, [_vp, _vp, _i, _vp]),
    "dvbs2_demap_estimate_snr_device": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dvbs2_demap_refine_snr": (_i, [_vp, _vp, _vp, _i, _vp]),
    "dvbs2_demap_refine_snr_device": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "dvbs2_plpayload_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dvbs2_plpayload_destroy": (None, [_vp]),
    "dvbs2_plpayload_params": (_i, [_vp, _ip, _ip, _ip]),
    "dvbs2_plpayload_process": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "dvbs2_plpayload_process_device": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dvbs2_pl_scrambling_rn": (_i, [_i, _vp, _i]),
    "dvbs2_bbdeheader_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dvbs2_bbdeheader_create_raw": (_i, [C.POINTER(_vp), _i, _i, _i]),
    "dvbs2_bbdeheader_destroy": (None, [_vp]),
    "dvbs2_bbdeheader_params": (_i, [_vp, _ip, _ip, _ip]),
    "dvbs2_bbdeheader_process": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_int64)]),
    "dvbs2_bbdeheader_process_device": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dvbs2_bbdeheader_finish": (_i, [_vp, C.POINTER(C.c_int64), _vp]),
    "dvbs2_bbdeheader_counters": (_i, [_vp, _vp, _vp]),
    "dvbs2_bbdeheader_reset": (_i, [_vp, _vp]),
    "dvbs2_chain_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i]),
    "dvbs2_chain_destroy": (None, [_vp]),
    "dvbs2_chain_params": (_i, [_vp, _ip, _ip]),
    "dvbs2_chain_set_descramble": (_i, [_vp, _i]),
    "dvbs2_chain_decode_device": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_chain_create_llr": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _i]),
    "dvbs2_chain_llr_params": (_i, [_vp, _ip, _ip, _ip]),
    "dvbs2_chain_decode_llr_device": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_chain_enqueue_device": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_chain_enqueue_llr_device": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dvbs2_chain_finish": (_i, [_vp]),
    "dvbs2_chain_decode": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "dvbs2_chain_decode_llr": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "dvbs2_chain_ldpc_profile": (_i, [_vp, _i, C.POINTER(C.c_double), _ip]),
    "dvbs2_chain_ldpc_kernel_name": (C.c_char_p, [_vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} not found: build it with `make -C gr-dvbs2rx_amd` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SYMBOLS.items():
    if os.environ.get("DVBS2_LIB") and not hasattr(lib, _name):
        continue  # kernel experiments against a library built from an older tree (tools/ab.sh); the product library must export all
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


class Dvbs2Error(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__(f"libdvbs2_fec_hip error {code}: {lib.dvbs2_last_error().decode()}")


def check(code):
    if code < 0:
        raise Dvbs2Error(code)
    return code
